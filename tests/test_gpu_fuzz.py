"""GPU (-m gpu): seeded random sweep of the public estimators against the CPU oracle.

Shapes (2 to 4 dims, T from 1 to a few thousand, feature sizes that hit every kernel family: I = 1 short / long / look-back
rows, interleaved, lock-step, SEQ), time_dim anywhere, float32 / float64, coefficient forms (python float, 0-dim tensor,
per-element tensor), mask regimes (none, sparse, dense, all done, terminated not a subset of done, per-step masks expanded
over the feature dims), the forward functions the path names.  Every case goes through the C ABI under AUTO dispatch and is
compared with oracle/value_oracle.c at the SCAN bar (fp32 rtol = atol = 1e-5, fp64 1e-12; V-trace fp64 1e-11 for exp)."""
import os
import random

import pytest
import torch

from oracle import value_oracle as O
from torchrl_amd.objectives.value import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# TRL_FUZZ_SCALE=k (an occasional long run on the GPU box): k times as many cases of every sweep, on seeds beyond the default ones
SCALE = max(1, int(os.environ.get("TRL_FUZZ_SCALE", "1")))
N_CASES = 500 * SCALE


def _case(seed):
    rng = random.Random(seed)
    nd = rng.choice([2, 3, 3, 3, 4])
    T = rng.choice([1, 2, 3, 7, 16, 31, 32, 33, 64, 80, 100, 128, 200, 255, 256, 257, 300, 511, 1024, 1025, 2500, 4100])
    feat_choices = [1, 1, 1, 2, 3, 4, 5, 8, 9, 16, 33, 70]
    budget = 60000  # elements: the oracle side stays fast
    if nd == 2:
        td = rng.choice([0, 1])
        other = max(1, min(rng.choice([1, 2, 5, 17, 64, 300]), budget // T))
        shape = [0, 0]
        shape[td] = T
        shape[1 - td] = other
    else:
        td = rng.randrange(nd)
        shape = []
        for k in range(nd):
            if k == td:
                shape.append(T)
            elif k > td:
                shape.append(rng.choice(feat_choices))
            else:
                shape.append(rng.choice([1, 2, 3, 6, 20, 130]))
        while True:  # trim the non-time dims to the element budget
            n = 1
            for s in shape:
                n *= s
            if n <= budget:
                break
            k = max((k for k in range(nd) if k != td), key=lambda k: shape[k])
            shape[k] = max(1, shape[k] // 2)
    time_dim = td - nd if rng.random() < 0.7 else td
    return dict(seed=seed, shape=tuple(shape), td=td, time_dim=time_dim, dtype=rng.choice([torch.float32, torch.float32, torch.float64]),
                gform=rng.choice(["float", "float", "t0", "full"]), lform=rng.choice(["float", "float", "t0", "full"]),
                regime=rng.choice(["none", "sparse", "sparse", "dense", "all_done", "term_free", "expanded"]),
                fn=rng.choice(["gae", "gae", "vec_gae", "gae_normalized", "td_lambda", "vec_td_lambda", "td1", "vtrace", "vtrace"]))


def _coef(form, value, shape, dtype, gen):
    if form == "float":
        return value
    if form == "t0":
        return torch.tensor(value, dtype=dtype)
    return (value - 0.1 * torch.rand(shape, generator=gen, dtype=torch.float64)).to(dtype)


def _masks(c, gen):
    shape, td = c["shape"], c["td"]
    mshape = shape
    if c["regime"] == "expanded":
        mshape = shape[:td + 1] + (1,) * (len(shape) - td - 1)
    p = {"none": 0.0, "sparse": 0.02, "dense": 0.4, "all_done": 1.1, "term_free": 0.1, "expanded": 0.05}[c["regime"]]
    term = torch.rand(mshape, generator=gen) < p
    done = term | (torch.rand(mshape, generator=gen) < p)
    if c["regime"] == "term_free":
        term = torch.rand(mshape, generator=gen) < 0.1   # drawn independently of done
    return done.expand(shape), term.expand(shape)


@pytest.mark.parametrize("seed", range(N_CASES))
def test_random_case_against_oracle(seed):
    c = _case(seed)
    shape, dt, kw = c["shape"], c["dtype"], {"time_dim": c["time_dim"]}
    gen = torch.Generator().manual_seed(1000 + seed)
    mk = lambda: torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
    r, v, nv = mk(), mk(), mk()
    done, term = _masks(c, gen)
    gamma = _coef(c["gform"], 0.97, shape, dt, gen)
    lmbda = _coef(c["lform"], 0.93, shape, dt, gen)
    cu = lambda x: x.to(DEV) if isinstance(x, torch.Tensor) else x
    dd, dtm = cu(done), cu(term)   # expanded masks stay stride-0 views on the device
    tol = dict(rtol=1e-5, atol=1e-5) if dt == torch.float32 else dict(rtol=1e-12, atol=1e-12)
    dc, tc = done.contiguous(), term.contiguous()
    fn = c["fn"]
    info = f"{c}"
    if fn in ("gae", "vec_gae", "gae_normalized"):
        ea, evt = O.gae(gamma, lmbda, v, nv, r, dc, tc, **kw)
        if fn == "gae_normalized":
            if v.numel() < 2:
                pytest.skip("std of one element")
            a, vt = F.gae_normalized(cu(gamma), cu(lmbda), cu(v), cu(nv), cu(r), dd, dtm, eps=1e-4, **kw)
            torch.testing.assert_close(vt.cpu(), evt, msg=lambda m: f"{info}: {m}", **tol)
            ref = O.normalize(ea.contiguous(), 1e-4).reshape(ea.shape)
            # the bar (1e-5 / 1e-12) holds for the advantage A; the normalised advantage is (A - mean) / max(std, eps), so an
            # error e in A arrives as e / std: the bar is scaled by 1 / std where std < 1 (random tiny batches of this sweep
            # can have any std) and is the plain bar otherwise (measured at the BASELINE shapes, std(A) ~ 3 - 5: <= 7e-7 / 3e-14,
            # profiles/r03_parity_errors.txt).  fp64: the mean / std come from fp64 sums in another order, 1e-12 * amplification
            amp = max(1.0, 1.0 / max(float(ea.double().std()), 1e-4))
            base = 1e-5 if dt == torch.float32 else 1e-12
            ntol = dict(rtol=base, atol=base * amp)
            torch.testing.assert_close(a.cpu(), ref, msg=lambda m: f"{info}: {m}", **ntol)
            return
        f = F.generalized_advantage_estimate if fn == "gae" else F.vec_generalized_advantage_estimate
        a, vt = f(cu(gamma), cu(lmbda), cu(v), cu(nv), cu(r), dd, dtm, **kw)
        assert a.shape == ea.shape
        torch.testing.assert_close(a.cpu().to(ea.dtype), ea, msg=lambda m: f"{info}: {m}", **tol)
        torch.testing.assert_close(vt.cpu().to(ea.dtype), evt, msg=lambda m: f"{info}: {m}", **tol)
        if c["regime"] != "term_free":
            assert torch.equal(a.cpu().to(ea.dtype)[dc], ea[dc]), f"{info}: done boundaries must be bit-exact"
    elif fn in ("td_lambda", "vec_td_lambda"):
        e = O.td_lambda(gamma, lmbda, nv, r, dc, tc, None, **kw)
        f = F.td_lambda_return_estimate if fn == "td_lambda" else F.vec_td_lambda_return_estimate
        out = f(cu(gamma), cu(lmbda), cu(nv), cu(r), dd, dtm, **kw)
        torch.testing.assert_close(out.cpu().to(e.dtype), e, msg=lambda m: f"{info}: {m}", **tol)
    elif fn == "td1":
        e = O.td1(gamma, nv, r, dc, tc, None, **kw)
        out = F.td1_return_estimate(cu(gamma), cu(nv), cu(r), dd, dtm, **kw)
        torch.testing.assert_close(out.cpu().to(e.dtype), e, msg=lambda m: f"{info}: {m}", **tol)
    else:
        lp = torch.rand(shape, generator=gen, dtype=torch.float64).clamp_min(1e-3).log().to(dt)
        lm = torch.rand(shape, generator=gen, dtype=torch.float64).clamp_min(1e-3).log().to(dt)
        ea, evs = O.vtrace(gamma, lp, lm, v, nv, r, dc, tc, 1.0, 0.9, **kw)
        a, vs = F.vtrace_advantage_estimate(cu(gamma), cu(lp), cu(lm), cu(v), cu(nv), cu(r), dd, dtm, 1.0, 0.9, **kw)
        # device exp vs the host's: an ulp on every importance ratio (measured at C4 / C5: 1.5e-6 mixed error at fp32 on the
        # scan kernels, 9e-7 on SEQ; 5e-15 at fp64): the north_star bar holds at fp32; fp64 carries the ulp of exp through
        # products of up to T ratios, 1e-11
        vtol = tol if dt == torch.float32 else dict(rtol=1e-11, atol=1e-11)
        torch.testing.assert_close(a.cpu().to(ea.dtype), ea, msg=lambda m: f"{info}: {m}", **vtol)
        torch.testing.assert_close(vs.cpu().to(ea.dtype), evs, msg=lambda m: f"{info}: {m}", **vtol)


N_BWD = 240 * SCALE


@pytest.mark.parametrize("seed", range(N_BWD))
def test_random_backward_adjoint_identity(seed):
    """Backward kernels on random shapes / time dims / coefficient forms / masks: every estimator is LINEAR in its value-like
    inputs (reward, state_value, next_state_value) for fixed masks, coefficients and log-probabilities, so for any direction u
    and any output cotangent g:  <g, F(u)> == <F^T g, u>.  F(u) is the forward path (pinned against the oracle above), F^T g
    what autograd returns through the backward entry points (trl_*_bwd)."""
    c = _case(10_000 + seed)
    fn = ["gae", "vec_gae", "td_lambda", "td1", "vtrace"][seed % 5]
    shape, dt, kw = c["shape"], c["dtype"], {"time_dim": c["time_dim"]}
    gen = torch.Generator().manual_seed(5000 + seed)
    mk = lambda: torch.randn(shape, generator=gen, dtype=torch.float64).to(dt).to(DEV)
    done, term = _masks(c, gen)
    done, term = done.to(DEV), term.to(DEV)
    gamma = _coef(c["gform"], 0.97, shape, dt, gen)
    lmbda = _coef(c["lform"], 0.93, shape, dt, gen)
    gamma = gamma.to(DEV) if isinstance(gamma, torch.Tensor) else gamma
    lmbda = lmbda.to(DEV) if isinstance(lmbda, torch.Tensor) else lmbda
    lp = torch.rand(shape, generator=gen, dtype=torch.float64).clamp_min(1e-3).log().to(dt).to(DEV)
    lm = torch.rand(shape, generator=gen, dtype=torch.float64).clamp_min(1e-3).log().to(dt).to(DEV)

    def forward(r, v, nv):
        if fn == "gae":
            return F.generalized_advantage_estimate(gamma, lmbda, v, nv, r, done, term, **kw)
        if fn == "vec_gae":
            return F.vec_generalized_advantage_estimate(gamma, lmbda, v, nv, r, done, term, **kw)
        if fn == "td_lambda":
            return (F.td_lambda_return_estimate(gamma, lmbda, nv, r, done, term, **kw),)
        if fn == "td1":
            return (F.td1_return_estimate(gamma, nv, r, done, term, **kw),)
        return F.vtrace_advantage_estimate(gamma, lp, lm, v, nv, r, done, term, 1.0, 0.9, **kw)

    x = [mk().requires_grad_(True) for _ in range(3)]          # r, v, nv: the point does not matter for a linear map
    outs = forward(*x)
    gs = [torch.randn(o.shape, generator=gen, dtype=torch.float64).to(o.dtype).to(DEV) for o in outs]
    used = [t for t in x if fn in ("gae", "vec_gae", "vtrace") or t is not x[1]]   # TD returns do not read state_value
    grads = torch.autograd.grad(outs, used, grad_outputs=gs)
    assert F.last_bwd_kernel() is not None
    u = [mk() for _ in range(3)]
    with torch.no_grad():
        fu = forward(*u)
    lhs = sum((g.double() * o.double()).sum() for g, o in zip(gs, fu))
    uu = [t for t, xx in zip(u, x) if any(xx is k for k in used)]
    rhs = sum((d.double() * t.double()).sum() for d, t in zip(grads, uu))
    scale = sum((g.double().abs() * o.double().abs()).sum() for g, o in zip(gs, fu)) + 1.0
    rel = (lhs - rhs).abs() / scale
    assert rel < (2e-5 if dt == torch.float32 else 1e-12), f"{fn} {c}: <g,F(u)>={lhs.item()} <F^T g,u>={rhs.item()} rel={rel.item()} [{F.last_bwd_kernel()}]"


N_SEQ = 300 * SCALE


@pytest.mark.parametrize("seed", range(N_SEQ))
def test_random_seq_family_bit_exact(seed):
    """The SEQ family walks time exactly as the reference loops do: GAE, TD(lambda) (rolling and rolling_gamma=False) and TD(1)
    must equal the oracle -- itself pinned bit for bit to the reference -- BIT FOR BIT on random shapes, time dims, coefficient
    forms (same dtype as the values) and mask regimes."""
    c = _case(20_000 + seed)
    fn = ["gae", "td_lambda", "td1", "td_lambda_fixed", "td1_fixed"][seed % 5]
    shape, dt, kw = c["shape"], c["dtype"], {"time_dim": c["time_dim"]}
    if fn.endswith("_fixed"):   # O(T^2): keep the rows short
        td = c["td"]
        if shape[td] > 300:
            shape = shape[:td] + (37,) + shape[td + 1:]
            c = dict(c, shape=shape)
    gen = torch.Generator().manual_seed(7000 + seed)
    mk = lambda: torch.randn(shape, generator=gen, dtype=torch.float64).to(dt)
    r, v, nv = mk(), mk(), mk()
    done, term = _masks(c, gen)
    gform = "full" if fn.endswith("_fixed") else c["gform"]
    gamma = _coef(gform, 0.97, shape, dt, gen)
    lmbda = _coef(c["lform"], 0.93, shape, dt, gen)
    cu = lambda x: x.to(DEV) if isinstance(x, torch.Tensor) else x
    dd, dtm, dc, tc = cu(done), cu(term), done.contiguous(), term.contiguous()
    td_arg = c["time_dim"]
    if fn == "gae":
        ea, evt = O.gae(gamma, lmbda, v, nv, r, dc, tc, **kw)
        call = F._prepare_gae(cu(gamma), cu(lmbda), cu(v), cu(nv), cu(r), dd, dtm, td_arg)
        a, vt, _ = F._launch_gae(call, algo=F.ALGO_SEQ)
        assert "seq" in F.last_kernel()
        assert torch.equal(a.cpu(), ea) and torch.equal(vt.cpu(), evt), f"{c}"
    elif fn.startswith("td_lambda"):
        rolling = None if fn == "td_lambda" else False
        e = O.td_lambda(gamma, lmbda, nv, r, dc, tc, rolling, **kw)
        call = F._prepare_td_lambda(cu(gamma), cu(lmbda), cu(nv), cu(r), dd, dtm, rolling, td_arg)
        out = F._launch_td_lambda(call, nv.dtype, algo=F.ALGO_SEQ)
        assert "seq" in F.last_kernel()
        assert torch.equal(out.cpu(), e), f"{c}: max abs diff {(out.cpu() - e).abs().max().item()}"
    else:
        rolling = None if fn == "td1" else False
        e = O.td1(gamma, nv, r, dc, tc, rolling, **kw)
        call = F._prepare_td1(cu(gamma), cu(nv), cu(r), dd, dtm, rolling, td_arg)
        out = F._launch_td1(call, nv.dtype, algo=F.ALGO_SEQ)
        assert "seq" in F.last_kernel()
        assert torch.equal(out.cpu(), e), f"{c}: max abs diff {(out.cpu() - e).abs().max().item()}"
