"""CPU (-m "not gpu"): the product's own host path (csrc/cpu_host.hip, trl_*_cpu -- the SEQ family's row functions of
csrc/seq_rows.h run on host threads) against the reference's golden fixtures and the oracle.  The reference runs on whatever
device its tensors live on (functional.py:163-170; BASELINE config 1 is a CPU call): CPU tensors go through the same public
functions as HIP tensors.  GAE, TD(lambda), TD(1) are mul / add / sub only: BIT-IDENTICAL to the reference loop; V-trace goes
through libm's exp (2e-6 / 1e-14 against the fixtures, as the oracle).  The oracle is the checker here, never the path: the
package does not import it (test_package_does_not_touch_the_oracle)."""
import os
import subprocess
import sys

import pytest
import torch

import golden_util as G
from oracle import value_oracle as O
from torchrl_amd import _extension
from torchrl_amd.objectives.value import functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _default_dtype():
    old = torch.get_default_dtype()
    yield
    torch.set_default_dtype(old)


def vtol(dtype):
    return dict(rtol=2e-6, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-14, atol=1e-14)


# ---- forward: the reference's fixtures -------------------------------------------------------------------------------------
@pytest.mark.parametrize("meta", G.cases("gae"), ids=G.ids("gae"))
def test_gae_golden_bit_exact(meta):
    torch.set_default_dtype(G.DT[meta["default_dtype"]])
    a = G.arrays("gae", meta)
    done, term = G.expand_masks(meta, a)
    for fn in (F.generalized_advantage_estimate, F.vec_generalized_advantage_estimate):
        adv, vt = fn(G.coef(meta, a, "gamma"), G.coef(meta, a, "lmbda"), a["state_value"], a["next_state_value"], a["reward"], done, term,
                     **G.time_kw(meta))
        assert adv.shape == a["advantage"].shape and adv.device.type == "cpu"
        if fn is F.generalized_advantage_estimate:
            assert adv.dtype == a["advantage"].dtype
        assert torch.equal(adv.to(a["advantage"].dtype), a["advantage"])
        assert torch.equal(vt.to(a["value_target"].dtype), a["value_target"])


@pytest.mark.parametrize("meta", G.cases("td_lambda"), ids=G.ids("td_lambda"))
def test_td_lambda_golden_bit_exact(meta):
    a = G.arrays("td_lambda", meta)
    ret = F.td_lambda_return_estimate(G.coef(meta, a, "gamma"), G.coef(meta, a, "lmbda"), a["next_state_value"], a["reward"], a["done"],
                                      a["terminated"], meta["rolling_gamma"], **G.time_kw(meta))
    assert ret.dtype == a["returns"].dtype
    assert torch.equal(ret, a["returns"])


@pytest.mark.parametrize("meta", G.cases("vtrace"), ids=G.ids("vtrace"))
def test_vtrace_golden(meta):
    a = G.arrays("vtrace", meta)
    rho = torch.tensor(meta["rho_thresh"]) if meta["thresh_as_tensor"] else meta["rho_thresh"]
    c = torch.tensor(meta["c_thresh"]) if meta["thresh_as_tensor"] else meta["c_thresh"]
    adv, vs = F.vtrace_advantage_estimate(G.coef(meta, a, "gamma"), a["log_pi"], a["log_mu"], a["state_value"], a["next_state_value"],
                                          a["reward"], a["done"], a.get("terminated"), rho, c, time_dim=meta["time_dim"])
    assert adv.dtype == a["advantage"].dtype
    torch.testing.assert_close(adv, a["advantage"], **vtol(adv.dtype))
    torch.testing.assert_close(vs, a["vs"], **vtol(adv.dtype))


@pytest.mark.parametrize("meta", G.cases("vtrace_thresh"), ids=G.ids("vtrace_thresh"))
def test_vtrace_threshold_tensors_golden(meta):
    a = G.arrays("vtrace_thresh", meta)
    adv, vs = F.vtrace_advantage_estimate(meta["gamma"], a["log_pi"], a["log_mu"], a["state_value"], a["next_state_value"], a["reward"],
                                          a["done"], a["terminated"], a["rho_thresh"], a["c_thresh"], time_dim=meta["time_dim"])
    torch.testing.assert_close(adv, a["advantage"], **vtol(adv.dtype))
    torch.testing.assert_close(vs, a["vs"], **vtol(adv.dtype))


@pytest.mark.parametrize("meta", G.cases("td", "td1"), ids=G.ids("td", "td1"))
def test_td1_golden_bit_exact(meta):
    a = G.arrays("td", meta)
    ret = F.td1_return_estimate(G.coef(meta, a, "gamma"), a["next_state_value"], a["reward"], a["done"], a["terminated"],
                                meta["rolling_gamma"], **G.time_kw(meta))
    assert ret.dtype == a["returns"].dtype and torch.equal(ret, a["returns"])


@pytest.mark.parametrize("meta", G.cases("td", "reward2go"), ids=G.ids("td", "reward2go"))
def test_reward2go_golden(meta):
    a = G.arrays("td", meta)
    out = F.reward2go(a["reward"], a["done"], meta["gamma"], **({} if meta["time_dim"] == -2 else {"time_dim": meta["time_dim"]}))
    assert out.shape == a["out"].shape and out.dtype == a["out"].dtype
    tol = dict(rtol=2e-6, atol=2e-6) if out.dtype == torch.float32 else dict(rtol=1e-13, atol=1e-13)
    torch.testing.assert_close(out, a["out"], **tol)


@pytest.mark.parametrize("meta", G.cases("normalise"), ids=G.ids("normalise"))
def test_normalise_golden(meta):
    a = G.arrays("normalise", meta)
    out = F.standardize(a["x"], meta["eps"])
    tol = dict(rtol=2e-6, atol=2e-6) if out.dtype == torch.float32 else dict(rtol=1e-13, atol=1e-13)
    torch.testing.assert_close(out, a["out"], **tol)


# ---- BASELINE config 1 and friends, bit for bit against the oracle -----------------------------------------------------------
def _inputs(shape, dtype, seed, p=0.02):
    g = torch.Generator().manual_seed(seed)
    mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32).to(dtype)
    r, v, nv = mk(), mk(), mk()
    term = torch.rand(shape, generator=g) < p
    done = term | (torch.rand(shape, generator=g) < p)
    gam = (0.9 + 0.1 * torch.rand(shape, generator=g)).to(dtype)
    lp, lm = (torch.rand(shape, generator=g, dtype=torch.float64).clamp_min(1e-3).log().to(dtype) for _ in range(2))
    return r, v, nv, done, term, gam, lp, lm


def test_baseline_config_1_cartpole_on_cpu():
    """BASELINE config 1: GAE gamma 0.99 lmbda 0.95 on [1 env x 2048 steps] fp32 via vec_generalized_advantage_estimate ON CPU."""
    g = torch.Generator().manual_seed(0)
    shape = (1, 2048, 1)
    r, v, nv = (torch.randn(shape, generator=g) for _ in range(3))
    term = torch.rand(shape, generator=g) < 1 / 200
    adv, vt = F.vec_generalized_advantage_estimate(0.99, 0.95, v, nv, r, term, term)
    oadv, ovt = O.gae(0.99, 0.95, v, nv, r, term, term)
    assert adv.device.type == "cpu" and torch.equal(adv, oadv) and torch.equal(vt, ovt)
    cs = G.checksums().get("C1_gae_f32")
    if cs is not None:   # the reference's own output at this seed, when the checksum fixture holds it
        assert abs(float(adv.double().sum()) - cs["advantage"]["sum"]) <= 1e-6 * abs(cs["advantage"]["sum"]) + 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape,time_dim", [((8, 128, 1), -2), ((64, 100, 5, 1), -3), ((3000, 32, 1), -2), ((5, 7, 2, 2), 1), ((40,), -1),
                                            ((2, 5000, 1), -2)], ids=lambda s: str(s))
def test_every_estimator_vs_oracle(shape, time_dim, dtype):
    r, v, nv, done, term, gam, lp, lm = _inputs(shape, dtype, 3)
    kw = {"time_dim": time_dim}
    T = shape[time_dim]
    for gamma in (0.99, gam):
        adv, vt = F.generalized_advantage_estimate(gamma, 0.95, v, nv, r, done, term, **kw)
        oadv, ovt = O.gae(gamma, 0.95, v, nv, r, done, term, **kw)
        assert torch.equal(adv, oadv) and torch.equal(vt, ovt)
        assert torch.equal(F.td_lambda_return_estimate(gamma, 0.9, nv, r, done, term, **kw), O.td_lambda(gamma, 0.9, nv, r, done, term, **kw))
        assert torch.equal(F.td1_return_estimate(gamma, nv, r, done, term, **kw), O.td1(gamma, nv, r, done, term, **kw))
        a, vs = F.vtrace_advantage_estimate(gamma, lp, lm, v, nv, r, done, term, 1.0, 0.9, time_dim=time_dim)
        oa, ovs = O.vtrace(gamma, lp, lm, v, nv, r, done, term, 1.0, 0.9, time_dim=time_dim)
        assert torch.equal(a, oa) and torch.equal(vs, ovs)   # same libm on both sides
    if T <= 200:
        assert torch.equal(F.td_lambda_return_estimate(gam, 0.9, nv, r, done, term, False, **kw),
                           O.td_lambda(gam, 0.9, nv, r, done, term, False, **kw))
        assert torch.equal(F.td1_return_estimate(gam, nv, r, done, term, False, **kw), O.td1(gam, nv, r, done, term, False, **kw))
    # normalised GAE: the moments are fp64 sums in another order than the oracle's -> rounding of mean / std only
    if r.numel() > 1:
        nadv, vt = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, **kw)
        want = O.normalize(O.gae(0.99, 0.95, v, nv, r, done, term, **kw)[0].contiguous(), 1e-4).reshape(nadv.shape)
        torch.testing.assert_close(nadv, want, **(dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-13, atol=1e-13)))


def test_expanded_masks_and_shifted_value_views_on_cpu():
    shape = (16, 10, 5, 1)
    r, v, nv, done, term, gam, lp, lm = _inputs(shape, torch.float32, 4)
    d = done[:, :, :1].expand(shape)
    t = term[:, :, :1].expand(shape)
    adv, vt = F.generalized_advantage_estimate(0.9, 0.9, v, nv, r, d, t, time_dim=-3)
    oadv, ovt = O.gae(0.9, 0.9, v, nv, r, d.contiguous(), t.contiguous(), time_dim=-3)
    assert torch.equal(adv, oadv) and torch.equal(vt, ovt)
    # the value network evaluated once on T + 1 steps: both windows of one buffer, read in place
    buf = torch.randn(12, 33, 1)
    sv, nsv = buf[:, :-1], buf[:, 1:]
    rr = torch.randn(12, 32, 1)
    dd = torch.rand(12, 32, 1) < 0.05
    adv, vt = F.generalized_advantage_estimate(0.99, 0.95, sv, nsv, rr, dd)
    oadv, ovt = O.gae(0.99, 0.95, sv.contiguous(), nsv.contiguous(), rr, dd)
    assert torch.equal(adv, oadv) and torch.equal(vt, ovt)


def test_thread_count_does_not_change_results():
    lib = _extension.get_lib()
    r, v, nv, done, term, gam, lp, lm = _inputs((4096, 64, 1), torch.float32, 5)
    outs = []
    try:
        for n in (1, 3, 8):
            assert lib.trl_set_cpu_threads(n) == 0 and lib.trl_cpu_threads() == n
            adv, vt, st = F.gae_with_stats(0.99, 0.95, v, nv, r, done, term)
            outs.append((adv, vt, st, F.gae_normalized(0.99, 0.95, v, nv, r, done, term)[0]))
    finally:
        lib.trl_set_cpu_threads(0)
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
    oadv, _ = O.gae(0.99, 0.95, v, nv, r, done, term)
    a64 = oadv.double()
    torch.testing.assert_close(outs[0][2], torch.tensor([float(a64.numel()), float(a64.sum()), float((a64 * a64).sum())], dtype=torch.float64),
                               rtol=1e-12, atol=1e-9)


def test_empty_and_errors_on_cpu():
    r = torch.randn(3, 8, 1)
    d = torch.zeros(3, 8, 1, dtype=torch.bool)
    adv, vt = F.generalized_advantage_estimate(0.9, 0.9, r[:0], r[:0], r[:0], d[:0])
    assert adv.shape == (0, 8, 1)
    with pytest.raises(RuntimeError, match="share a unique shape"):
        F.generalized_advantage_estimate(0.9, 0.9, r, r, r[:, :4], d)
    with pytest.raises(RuntimeError, match="rolling_gamma=False is expected only"):
        F.td_lambda_return_estimate(0.9, 0.9, r, r, d, rolling_gamma=False)
    with pytest.raises(RuntimeError, match="HIP"):
        F.prepare_gae_step(F._prepare_gae(0.9, 0.9, r, r, r, d, d, -2), 1e-4)   # bound steps / workspaces are GPU-only


# ---- backward: the reference's autograd fixtures ---------------------------------------------------------------------------
def gtol(dtype):
    return dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1e-11, atol=1e-11)


def leaf(t):
    return None if t is None else t.clone().requires_grad_(True)


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
        torch.testing.assert_close(g, want, msg=lambda m, n=name: f"d_{n}: {m}", **gtol(want.dtype))


def kw(meta):
    return {} if meta["time_dim"] == -2 else {"time_dim": meta["time_dim"]}


@pytest.mark.parametrize("meta", G.cases("grads", "gae"), ids=G.ids("grads", "gae"))
def test_gae_grads_golden(meta):
    a = G.arrays("grads", meta)
    r, v, nv = leaf(a["reward"]), leaf(a["state_value"]), leaf(a["next_state_value"])
    g, gl = coef(meta, a, "gamma")
    l, ll = coef(meta, a, "lmbda")
    adv, vt = F.generalized_advantage_estimate(g, l, v, nv, r, a["done"], a["terminated"], **kw(meta))
    loss = (adv * a["w_advantage"]).sum() + (vt * a["w_value_target"]).sum()
    ins = [r, v, nv] + [x for x in (gl, ll) if x is not None]
    names = ["reward", "state_value", "next_state_value"] + [n for n, x in (("gamma", gl), ("lmbda", ll)) if x is not None]
    compare(torch.autograd.grad(loss, ins), a, names)
    assert F.last_bwd_kernel() == "gae_bwd_cpu"


@pytest.mark.parametrize("meta", G.cases("grads", "td_lambda"), ids=G.ids("grads", "td_lambda"))
def test_td_lambda_grads_golden(meta):
    a = G.arrays("grads", meta)
    r, nv = leaf(a["reward"]), leaf(a["next_state_value"])
    g, gl = coef(meta, a, "gamma")
    l, ll = coef(meta, a, "lmbda")
    ret = F.td_lambda_return_estimate(g, l, nv, r, a["done"], a["terminated"], **kw(meta))
    ins = [r, nv] + [x for x in (gl, ll) if x is not None]
    names = ["reward", "next_state_value"] + [n for n, x in (("gamma", gl), ("lmbda", ll)) if x is not None]
    compare(torch.autograd.grad((ret * a["w_returns"]).sum(), ins), a, names)


@pytest.mark.parametrize("meta", G.cases("grads", "td1"), ids=G.ids("grads", "td1"))
def test_td1_grads_golden(meta):
    a = G.arrays("grads", meta)
    r, nv = leaf(a["reward"]), leaf(a["next_state_value"])
    g, gl = coef(meta, a, "gamma")
    ret = F.td1_return_estimate(g, nv, r, a["done"], a["terminated"], **kw(meta))
    ins = [r, nv] + ([gl] if gl is not None else [])
    names = ["reward", "next_state_value"] + (["gamma"] if gl is not None else [])
    compare(torch.autograd.grad((ret * a["w_returns"]).sum(), ins), a, names)


@pytest.mark.parametrize("meta", G.cases("grads", "vtrace"), ids=G.ids("grads", "vtrace"))
def test_vtrace_grads_golden(meta):
    a = G.arrays("grads", meta)
    lp, lmu = leaf(a["log_pi"]), leaf(a["log_mu"])
    v, nv, r = leaf(a["state_value"]), leaf(a["next_state_value"]), leaf(a["reward"])
    g, gl = coef(meta, a, "gamma")
    adv, vs = F.vtrace_advantage_estimate(g, lp, lmu, v, nv, r, a["done"], a["terminated"], meta["rho_thresh"], meta["c_thresh"],
                                          time_dim=meta["time_dim"])
    loss = (adv * a["w_advantage"]).sum() + (vs * a["w_vs"]).sum()
    ins = [lp, lmu, v, nv, r] + ([gl] if gl is not None else [])
    names = ["log_pi", "log_mu", "state_value", "next_state_value", "reward"] + (["gamma"] if gl is not None else [])
    compare(torch.autograd.grad(loss, ins), a, names)


def test_nonrolling_backward_is_gpu_only():
    r, v, nv, done, term, gam, lp, lm = _inputs((3, 12, 1), torch.float64, 6)
    rr = r.clone().requires_grad_(True)
    ret = F.td_lambda_return_estimate(gam, 0.9, nv, rr, done, term, False)
    with pytest.raises(NotImplementedError, match="GPU only"):
        ret.sum().backward()


# ---- the modules on CPU tensors --------------------------------------------------------------------------------------------
def test_gae_module_on_cpu():
    from torchrl_amd.objectives.value import GAE, RolloutData

    r, v, nv, done, term, *_ = _inputs((8, 128, 1), torch.float32, 7)
    mod = GAE(gamma=0.99, lmbda=0.95, value_network=None, average_gae=True)
    batch = RolloutData({"state_value": v, "next": {"state_value": nv, "reward": r, "done": done, "terminated": term}}, batch_size=(8, 128))
    out = mod(batch)
    oadv, ovt = O.gae(0.99, 0.95, v, nv, r, done, term)
    assert torch.equal(out["value_target"], ovt)
    torch.testing.assert_close(out["advantage"], O.normalize(oadv, 1e-4), rtol=1e-6, atol=1e-6)


def test_package_does_not_touch_the_oracle():
    """The host path is the product's own code: importing and using the package on CPU tensors loads nothing from oracle/."""
    code = ("import sys, torch\n"
            "from torchrl_amd.objectives.value import functional as F\n"
            "r = torch.randn(4, 16, 1); d = torch.rand(4, 16, 1) < 0.1\n"
            "F.generalized_advantage_estimate(0.99, 0.95, r, r, r, d)\n"
            "F.td_lambda_return_estimate(0.99, 0.95, r, r, d)\n"
            "F.vtrace_advantage_estimate(0.99, r, r, r, r, r, d)\n"
            "bad = [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]\n"
            "maps = open('/proc/self/maps').read()\n"
            "assert not bad and 'value_oracle' not in maps, (bad, 'oracle library mapped')\n"
            "print('clean')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=ROOT))
    assert out.returncode == 0 and "clean" in out.stdout, out.stderr[-2000:]


def test_in_place_update_between_forward_and_backward_raises():
    """The backward kernels re-read the forward call's inputs: a rollout buffer modified in place in between must raise, as
    torch's autograd does for its saved tensors, not give silently wrong gradients."""
    g = torch.Generator().manual_seed(0)
    shape = (4, 12, 1)
    r, nv = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    v = torch.randn(shape, generator=g, requires_grad=True)
    lp = torch.randn(shape, generator=g, requires_grad=True)
    lm = torch.randn(shape, generator=g)
    done = torch.rand(shape, generator=g) < 0.1
    for fwd, victim in ((lambda: F.generalized_advantage_estimate(0.9, 0.8, v, nv, r, done)[0], done),
                        (lambda: F.vtrace_advantage_estimate(0.9, lp, lm, v, nv, r, done)[0], nv)):
        out = fwd()
        out.sum().backward()  # untouched inputs: fine
        out = fwd()
        victim.mul_(0) if victim.dtype != torch.bool else victim.logical_not_()
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            out.sum().backward()
