"""GPU (-m gpu): backward passes (trl_gae_bwd / trl_td_lambda_bwd / trl_td1_bwd / trl_vtrace_bwd) against gradient
fixtures produced by torch autograd THROUGH THE REFERENCE's loops (tools/gen_golden_next.py -> tests/golden/grads.npz):
the scalar loss is sum(out_k * W_k) with the stored random weights, gradients w.r.t. reward, values, log-probabilities
and gamma / lmbda tensors.  Tolerance: the backward kernels run the adjoint scan in the forward kernels' arithmetic
(re-associated affine maps): fp32 rtol = atol = 2e-5, fp64 1e-11 -- the forward bar with the headroom of one more
multiply per element (d_next_state_value = G * gamma * nt).  Both kernel families (SCAN for I == 1, SEQ) are checked."""
import pytest
import torch

import golden_util as G
from torchrl_amd.objectives.value import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tol(dtype):
    return dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1e-11, atol=1e-11)


def leaf(t):
    return None if t is None else t.to(DEV).requires_grad_(True)


def coef(meta, a, name):
    m = meta.get(name)
    if m is None or m["kind"] == "float":
        return (m or {}).get("value"), None
    t = leaf(a[name])
    return t, t


def compare(got, a, names):
    for name, g in zip(names, got):
        want = a["d_" + name]
        assert g.shape == want.shape and g.dtype == want.dtype, name
        torch.testing.assert_close(g.cpu(), want, msg=lambda m, n=name: f"d_{n}: {m}", **tol(want.dtype))


@pytest.fixture(params=[F.ALGO_AUTO, F.ALGO_SEQ], ids=["auto", "seq"])
def bwd_algo(request, monkeypatch):
    monkeypatch.setattr(F, "_BWD_ALGO", request.param)
    return request.param


def kw(meta):
    return {} if meta["time_dim"] == -2 else {"time_dim": meta["time_dim"]}


@pytest.mark.parametrize("meta", G.cases("grads", "gae"), ids=G.ids("grads", "gae"))
def test_gae_grads_golden(meta, bwd_algo):
    a = G.arrays("grads", meta)
    r, v, nv = leaf(a["reward"]), leaf(a["state_value"]), leaf(a["next_state_value"])
    g, gl = coef(meta, a, "gamma")
    l, ll = coef(meta, a, "lmbda")
    for fn in (F.generalized_advantage_estimate, F.vec_generalized_advantage_estimate):
        adv, vt = fn(g, l, v, nv, r, a["done"].to(DEV), a["terminated"].to(DEV), **kw(meta))
        loss = (adv * a["w_advantage"].to(DEV)).sum() + (vt * a["w_value_target"].to(DEV)).sum()
        ins = [r, v, nv] + [x for x in (gl, ll) if x is not None]
        names = ["reward", "state_value", "next_state_value"] + [n for n, x in (("gamma", gl), ("lmbda", ll)) if x is not None]
        compare(torch.autograd.grad(loss, ins), a, names)
        assert F.last_bwd_kernel() == ("gae_bwd_seq" if bwd_algo == F.ALGO_SEQ or meta["shape"][-1] != 1 else "gae_bwd_scan")
    # a single requested gradient / a single incoming gradient
    adv, vt = F.generalized_advantage_estimate(g, l, v, nv, r, a["done"].to(DEV), a["terminated"].to(DEV), **kw(meta))
    (dr,) = torch.autograd.grad((adv * a["w_advantage"].to(DEV)).sum() + (vt * a["w_value_target"].to(DEV)).sum(), [r])
    torch.testing.assert_close(dr.cpu(), a["d_reward"], **tol(dr.dtype))


@pytest.mark.parametrize("meta", G.cases("grads", "td_lambda"), ids=G.ids("grads", "td_lambda"))
def test_td_lambda_grads_golden(meta, bwd_algo):
    a = G.arrays("grads", meta)
    r, nv = leaf(a["reward"]), leaf(a["next_state_value"])
    g, gl = coef(meta, a, "gamma")
    l, ll = coef(meta, a, "lmbda")
    for fn in (F.td_lambda_return_estimate, F.vec_td_lambda_return_estimate):
        ret = fn(g, l, nv, r, a["done"].to(DEV), a["terminated"].to(DEV), **kw(meta))
        ins = [r, nv] + [x for x in (gl, ll) if x is not None]
        names = ["reward", "next_state_value"] + [n for n, x in (("gamma", gl), ("lmbda", ll)) if x is not None]
        compare(torch.autograd.grad((ret * a["w_returns"].to(DEV)).sum(), ins), a, names)


@pytest.mark.parametrize("meta", G.cases("grads", "td1"), ids=G.ids("grads", "td1"))
def test_td1_grads_golden(meta, bwd_algo):
    a = G.arrays("grads", meta)
    r, nv = leaf(a["reward"]), leaf(a["next_state_value"])
    g, gl = coef(meta, a, "gamma")
    ret = F.td1_return_estimate(g, nv, r, a["done"].to(DEV), a["terminated"].to(DEV), **kw(meta))
    ins = [r, nv] + ([gl] if gl is not None else [])
    names = ["reward", "next_state_value"] + (["gamma"] if gl is not None else [])
    compare(torch.autograd.grad((ret * a["w_returns"].to(DEV)).sum(), ins), a, names)


@pytest.mark.parametrize("meta", G.cases("grads", "vtrace"), ids=G.ids("grads", "vtrace"))
def test_vtrace_grads_golden(meta, bwd_algo):
    a = G.arrays("grads", meta)
    lp, lmu = leaf(a["log_pi"]), leaf(a["log_mu"])
    v, nv, r = leaf(a["state_value"]), leaf(a["next_state_value"]), leaf(a["reward"])
    g, gl = coef(meta, a, "gamma")
    adv, vs = F.vtrace_advantage_estimate(g, lp, lmu, v, nv, r, a["done"].to(DEV), a["terminated"].to(DEV), meta["rho_thresh"],
                                          meta["c_thresh"], time_dim=meta["time_dim"])
    loss = (adv * a["w_advantage"].to(DEV)).sum() + (vs * a["w_vs"].to(DEV)).sum()
    ins = [lp, lmu, v, nv, r] + ([gl] if gl is not None else [])
    names = ["log_pi", "log_mu", "state_value", "next_state_value", "reward"] + (["gamma"] if gl is not None else [])
    compare(torch.autograd.grad(loss, ins), a, names)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape", [(64, 32, 1), (300, 32, 1), (37, 100, 1), (5, 7, 1), (3, 2500, 1), (2, 4099, 1), (7, 1025, 1),
                                   (1, 40000, 1)], ids=lambda s: "x".join(map(str, s)))
def test_scan_and_seq_backward_agree(shape, dtype, monkeypatch):
    """Every SCAN geometry (short / ragged / long / look-back) of the four backward kernels against the SEQ kernels
    (pinned by the fixtures above) on seeded inputs, with per-element gamma."""
    g = torch.Generator().manual_seed(77)
    mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32).to(dtype).to(DEV)
    term = (torch.rand(shape, generator=g) < 0.02).to(DEV)
    done = term | (torch.rand(shape, generator=g) < 0.02).to(DEV)
    gam = (0.9 + 0.1 * torch.rand(shape, generator=g)).to(dtype).to(DEV)
    lp, lmu = (0.3 * mk()), (0.3 * mk())
    wa, wv = mk(), mk()
    results = {}
    for algo in (F.ALGO_SCAN, F.ALGO_SEQ):
        monkeypatch.setattr(F, "_BWD_ALGO", algo)
        g2 = torch.Generator().manual_seed(1)
        r, v, nv = (torch.randn(shape, generator=g2, dtype=torch.float32).to(dtype).to(DEV).requires_grad_(True) for _ in range(3))
        lpl, lml = lp.clone().requires_grad_(True), lmu.clone().requires_grad_(True)
        out = []
        adv, vt = F.generalized_advantage_estimate(gam, 0.95, v, nv, r, done, term)
        out += list(torch.autograd.grad((adv * wa).sum() + (vt * wv).sum(), [r, v, nv]))
        assert ("scan" in F.last_bwd_kernel()) == (algo == F.ALGO_SCAN), F.last_bwd_kernel()
        ret = F.td_lambda_return_estimate(gam, 0.95, nv, r, done, term)
        out += list(torch.autograd.grad((ret * wa).sum(), [r, nv]))
        ret = F.td1_return_estimate(gam, nv, r, done, term)
        out += list(torch.autograd.grad((ret * wa).sum(), [r, nv]))
        adv, vs = F.vtrace_advantage_estimate(gam, lpl, lml, v, nv, r, done, term, 0.9, 0.8)
        out += list(torch.autograd.grad((adv * wa).sum() + (vs * wv).sum(), [lpl, lml, v, nv, r]))
        results[algo] = out
    for a, b in zip(results[F.ALGO_SCAN], results[F.ALGO_SEQ]):
        torch.testing.assert_close(a, b, **tol(dtype))


def test_gae_backward_adjoint_identity_full_size():
    """Size-independent property at the north-star size [65536 x 32]: GAE is linear in (reward, state_value,
    next_state_value), so <W, GAE(x)> = <GAE^T W, x> -- the backward kernel is the transpose of the forward one."""
    shape = (65536, 32, 1)
    g = torch.Generator(device=DEV).manual_seed(0)
    r, v, nv = (torch.randn(shape, generator=g, device=DEV, dtype=torch.float64).requires_grad_(True) for _ in range(3))
    term = torch.rand(shape, generator=g, device=DEV) < 0.01
    done = term | (torch.rand(shape, generator=g, device=DEV) < 0.01)
    wa, wv = (torch.randn(shape, generator=g, device=DEV, dtype=torch.float64) for _ in range(2))
    adv, vt = F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term)
    lhs = (adv * wa).sum() + (vt * wv).sum()
    dr, dv, dnv = torch.autograd.grad(lhs, [r, v, nv])
    assert F.last_bwd_kernel() == "gae_bwd_scan"
    rhs = (dr * r).sum() + (dv * v).sum() + (dnv * nv).sum()
    torch.testing.assert_close(lhs.detach(), rhs.detach(), rtol=1e-10, atol=1e-6)


# ---- rolling_gamma=False (trl_td_fixed_bwd): tests/golden/grads_nr.npz, tools/gen_golden_grads_nr.py ----------------------------
@pytest.mark.parametrize("meta", G.cases("grads_nr", "td_lambda"), ids=G.ids("grads_nr", "td_lambda"))
def test_td_lambda_nonrolling_grads_golden(meta):
    a = G.arrays("grads_nr", meta)
    r, nv = leaf(a["reward"]), leaf(a["next_state_value"])
    g, gl = coef(meta, a, "gamma")
    l, ll = coef(meta, a, "lmbda")
    ret = F.td_lambda_return_estimate(g, l, nv, r, a["done"].to(DEV), a["terminated"].to(DEV), False, **kw(meta))
    torch.testing.assert_close(ret.detach().cpu(), a["returns"], **tol(a["returns"].dtype))
    ins = [r, nv] + [x for x in (gl, ll) if x is not None]
    names = ["reward", "next_state_value"] + [n for n, x in (("gamma", gl), ("lmbda", ll)) if x is not None]
    compare(torch.autograd.grad((ret * a["w_returns"].to(DEV)).sum(), ins), a, names)
    assert "fixed_bwd" in F.last_bwd_kernel()


@pytest.mark.parametrize("meta", G.cases("grads_nr", "td1"), ids=G.ids("grads_nr", "td1"))
def test_td1_nonrolling_grads_golden(meta):
    a = G.arrays("grads_nr", meta)
    r, nv = leaf(a["reward"]), leaf(a["next_state_value"])
    g, gl = coef(meta, a, "gamma")
    ret = F.td1_return_estimate(g, nv, r, a["done"].to(DEV), a["terminated"].to(DEV), False, **kw(meta))
    torch.testing.assert_close(ret.detach().cpu(), a["returns"], **tol(a["returns"].dtype))
    compare(torch.autograd.grad((ret * a["w_returns"].to(DEV)).sum(), [r, nv, gl]), a, ["reward", "next_state_value", "gamma"])
    assert "fixed_bwd" in F.last_bwd_kernel()


def test_nonrolling_backward_value_gradients_only_and_expanded_masks():
    """Only the value-like inputs require grad (no coefficient kernel), masks expanded over I > 1 (compact form)."""
    g = torch.Generator().manual_seed(4)
    shape = (3, 150, 2)
    r, nv = (torch.randn(shape, generator=g, dtype=torch.float64).to(DEV).requires_grad_(True) for _ in range(2))
    gam = (0.9 + 0.1 * torch.rand(shape, generator=g, dtype=torch.float64)).to(DEV)
    term = (torch.rand(3, 150, 1, generator=g) < 0.03).to(DEV)
    done = (term | (torch.rand(3, 150, 1, generator=g) < 0.03).to(DEV)).expand(shape)
    term = term.expand(shape)
    w = torch.randn(shape, generator=g, dtype=torch.float64).to(DEV)
    ret = F.td_lambda_return_estimate(gam, 0.9, nv, r, done, term, False)
    d_nv, d_r = torch.autograd.grad((ret * w).sum(), (nv, r))
    # the same through torch autograd over a python restatement of functional.py:877-887
    r2, nv2 = r.detach().clone().requires_grad_(True), nv.detach().clone().requires_grad_(True)
    nvm = nv2 * (~term).to(nv2.dtype)
    dn = done.to(nv2.dtype)
    outs = []
    for k in range(shape[1]):
        gk = nvm[:, -1]
        for i in reversed(range(k, shape[1])):
            gk = gk * (1 - dn[:, i]) + nvm[:, i] * dn[:, i]
            gk = r2[:, i] + gam[:, k] * ((1 - 0.9) * nvm[:, i] + 0.9 * gk)
        outs.append(gk)
    want = torch.autograd.grad((torch.stack(outs, 1) * w).sum(), (nv2, r2))
    torch.testing.assert_close(d_nv, want[0], rtol=1e-10, atol=1e-10)
    torch.testing.assert_close(d_r, want[1], rtol=1e-10, atol=1e-10)
