"""GPU (-m gpu): the HIP path, called through the C ABI (ctypes), against
  (a) the golden fixtures generated from the reference (tests/golden), and
  (b) the CPU oracle on seeded inputs.
Bars: SEQ family bit-identical to the reference loop; SCAN family fp32 rtol=atol=1e-5, fp64
rtol=atol=1e-12, and bit-exact wherever `done` is set (the carry is annihilated there)."""
import pytest
import torch

import golden_util as G
from oracle import value_oracle as O
from torchrl_amd.objectives.value import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SEQ, SCAN, LOCKSTEP = F.ALGO_SEQ, F.ALGO_SCAN, F.ALGO_LOCKSTEP


def tol(dtype):
    return dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-12, atol=1e-12)


def cu(x):
    return x.to(DEV) if isinstance(x, torch.Tensor) else x


def check(out, ref, exact, done=None):
    out = out.cpu()
    assert out.dtype == ref.dtype and out.shape == ref.shape
    if exact:
        assert torch.equal(out, ref), f"max abs diff {(out - ref).abs().max().item()}"
    else:
        torch.testing.assert_close(out, ref, **tol(ref.dtype))
        if done is not None:
            assert torch.equal(out[done], ref[done]), "done boundaries must be bit-exact"


def algos(call):
    """SEQ always; SCAN when the family supports the shape (I == 1, or whole outer rows fit a chunk)."""
    chunk = 256 * (4 if call.vdt == torch.float32 else 2)
    out = [SEQ, SCAN] if (call.I == 1 or call.T * call.I <= chunk) else [SEQ]
    if 2 <= call.I <= 31:
        out.append(LOCKSTEP)  # a lane per time step, the I recurrences in lock step: any T
    return out


@pytest.mark.parametrize("meta", G.cases("gae"), ids=G.ids("gae"))
def test_gae_golden(meta):
    torch.set_default_dtype(G.DT[meta["default_dtype"]])
    a = G.arrays("gae", meta)
    done, term = G.expand_masks(meta, a)
    call = F._prepare_gae(cu(G.coef(meta, a, "gamma")), cu(G.coef(meta, a, "lmbda")), cu(a["state_value"]),
                          cu(a["next_state_value"]), cu(a["reward"]), cu(done), cu(term), meta["time_dim"])
    for algo in algos(call):
        adv, vt, _ = F._launch_gae(call, algo=algo)
        check(adv, a["advantage"], algo == SEQ, a["done"])
        check(vt, a["value_target"], algo == SEQ, a["done"])
    # public entry points (AUTO dispatch), loop and vec names
    for fn in (F.generalized_advantage_estimate, F.vec_generalized_advantage_estimate):
        adv, vt = fn(cu(G.coef(meta, a, "gamma")), cu(G.coef(meta, a, "lmbda")), cu(a["state_value"]),
                     cu(a["next_state_value"]), cu(a["reward"]), cu(done), cu(term), **G.time_kw(meta))
        if fn is F.vec_generalized_advantage_estimate:
            # the reference's vec variant returns promote(value dtype, dtype of a gamma*lmbda tensor) (functional.py:317-359;
            # pinned by tests/golden/vec.npz); the values are the loop's
            want = F._vec_gae_out_dtype(a["state_value"].dtype, G.coef(meta, a, "gamma"), G.coef(meta, a, "lmbda"))
            assert adv.dtype == want and vt.dtype == want
            adv, vt = adv.to(a["advantage"].dtype), vt.to(a["advantage"].dtype)
        check(adv, a["advantage"], False, a["done"])
        check(vt, a["value_target"], False, a["done"])


@pytest.mark.parametrize("meta", G.cases("td_lambda"), ids=G.ids("td_lambda"))
def test_td_lambda_golden(meta):
    a = G.arrays("td_lambda", meta)
    call = F._prepare_td_lambda(cu(G.coef(meta, a, "gamma")), cu(G.coef(meta, a, "lmbda")), cu(a["next_state_value"]),
                                cu(a["reward"]), cu(a["done"]), cu(a["terminated"]), meta["rolling_gamma"], meta["time_dim"])
    for algo in algos(call):
        if algo == SCAN and not call.rolling:
            continue
        ret = F._launch_td_lambda(call, a["returns"].dtype, algo=algo)
        check(ret, a["returns"], algo == SEQ)
    ret = F.td_lambda_return_estimate(cu(G.coef(meta, a, "gamma")), cu(G.coef(meta, a, "lmbda")), cu(a["next_state_value"]),
                                      cu(a["reward"]), cu(a["done"]), cu(a["terminated"]), meta["rolling_gamma"], **G.time_kw(meta))
    check(ret, a["returns"], False)


@pytest.mark.parametrize("meta", G.cases("vtrace"), ids=G.ids("vtrace"))
def test_vtrace_golden(meta):
    a = G.arrays("vtrace", meta)
    rho = torch.tensor(meta["rho_thresh"]) if meta["thresh_as_tensor"] else meta["rho_thresh"]
    c = torch.tensor(meta["c_thresh"]) if meta["thresh_as_tensor"] else meta["c_thresh"]
    call = F._prepare_vtrace(cu(G.coef(meta, a, "gamma")), cu(a["log_pi"]), cu(a["log_mu"]), cu(a["state_value"]),
                             cu(a["next_state_value"]), cu(a["reward"]), cu(a["done"]), cu(a.get("terminated")),
                             rho, c, meta["time_dim"])
    for algo in algos(call):
        adv, vs = F._launch_vtrace(call, algo=algo)
        check(adv, a["advantage"], False)
        check(vs, a["vs"], False)
    adv, vs = F.vtrace_advantage_estimate(cu(G.coef(meta, a, "gamma")), cu(a["log_pi"]), cu(a["log_mu"]),
                                          cu(a["state_value"]), cu(a["next_state_value"]), cu(a["reward"]), cu(a["done"]),
                                          cu(a.get("terminated")), rho, c, time_dim=meta["time_dim"])
    check(adv, a["advantage"], False)
    check(vs, a["vs"], False)


@pytest.mark.parametrize("meta", G.cases("normalise"), ids=G.ids("normalise"))
def test_normalise_golden(meta):
    a = G.arrays("normalise", meta)
    x = cu(a["x"])
    st = F.moments(x)
    ost = O.normalize(a["x"], meta["eps"], return_stats=True)[1]
    torch.testing.assert_close(st.cpu(), ost, rtol=1e-13, atol=0)
    out = F.normalize_(x.clone(), st, meta["eps"])
    t = dict(rtol=1e-5, atol=1e-5) if a["out"].dtype == torch.float32 else dict(rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(out.cpu(), a["out"], **t)
    torch.testing.assert_close(F.standardize(x, meta["eps"]).cpu(), a["out"], **t)


def _rand_inputs(shape, dtype, seed, p_term=0.02, p_trunc=0.02):
    g = torch.Generator().manual_seed(seed)
    r, v, nv = (torch.randn(shape, generator=g, dtype=torch.float32).to(dtype) for _ in range(3))
    term = torch.rand(shape, generator=g) < p_term
    done = term | (torch.rand(shape, generator=g) < p_trunc)
    gam = (0.9 + 0.1 * torch.rand(shape, generator=g)).to(dtype)
    lp = torch.rand(shape, generator=g).clamp_min(1e-6).log().to(dtype)
    lm = torch.rand(shape, generator=g).clamp_min(1e-6).log().to(dtype)
    return r, v, nv, done, term, gam, lp, lm


# shapes chosen to hit: SHORT wave-independent (T | 256), SHORT with cross-wave carry, ragged rows
# (T % 4 != 0 -> unaligned path / row tails), single block, LONG rows (several chunks, partial chunk)
SCAN_SHAPES = [(64, 32, 1), (300, 32, 1), (37, 100, 1), (5, 7, 1), (33, 3, 1), (9, 1, 1), (1, 2048, 1), (3, 2500, 1),
               (2, 4099, 1), (130, 128, 1), (16, 1024, 1), (7, 1025, 1)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape", SCAN_SHAPES, ids=[f"{s[0]}x{s[1]}" for s in SCAN_SHAPES])
@pytest.mark.parametrize("tensor_gamma", [False, True], ids=["g_scalar", "g_tensor"])
def test_all_estimators_vs_oracle(shape, dtype, tensor_gamma):
    r, v, nv, done, term, gam, lp, lm = _rand_inputs(shape, dtype, seed=hash(shape) % 1000)
    gamma = gam if tensor_gamma else 0.99
    oadv, ovt = O.gae(gamma, 0.95, v, nv, r, done, term)
    oret = O.td_lambda(gamma, 0.95, nv, r, done, term)
    ova, ovs = O.vtrace(gamma, lp, lm, v, nv, r, done, term)
    gcall = F._prepare_gae(cu(gamma), 0.95, cu(v), cu(nv), cu(r), cu(done), cu(term), -2)
    tcall = F._prepare_td_lambda(cu(gamma), 0.95, cu(nv), cu(r), cu(done), cu(term), None, -2)
    vcall = F._prepare_vtrace(cu(gamma), cu(lp), cu(lm), cu(v), cu(nv), cu(r), cu(done), cu(term), 1.0, 1.0, -2)
    for algo in (SEQ, SCAN):
        adv, vt, st = F._launch_gae(gcall, want_stats=True, algo=algo)
        assert ("scan" in F.last_kernel()) == (algo == SCAN)
        check(adv, oadv, algo == SEQ, done)
        check(vt, ovt, algo == SEQ, done)
        a64 = adv.double()
        ref_st = torch.stack([torch.tensor(float(a64.numel()), device=DEV), a64.sum(), (a64 * a64).sum()])
        torch.testing.assert_close(st, ref_st, rtol=1e-12, atol=1e-9)
        check(F._launch_td_lambda(tcall, dtype, algo=algo), oret, algo == SEQ)
        va, vs = F._launch_vtrace(vcall, algo=algo)
        check(va, ova, False)
        check(vs, ovs, False)


MULTI = [((40, 50, 5, 1), -3), ((6, 9, 2, 5), 1), ((33, 17, 64), -2), ((4, 300, 130), -2), ((4096, 100, 5, 1), -3),
         ((7, 3, 2), -2), ((50, 16, 4), -2), ((19, 101, 3), -2), ((64, 32, 8), 1),
         # 64 < T <= 128: the GAE kernel's row phase gives every (o, i) row a 16-lane DPP row, 5 - 8 steps per lane
         # (gae_interleaved.h, VR > 0); last chunk short of rows, rows short of 16 * VR steps, T * I just under a chunk
         ((33, 80, 5, 1), -3), ((21, 96, 4), -2), ((37, 65, 7), -2), ((9, 113, 9), -2), ((5, 66, 15), -2), ((11, 120, 8), -2),
         ((70, 100, 2), -2), ((3, 128, 8), -2), ((1, 97, 1, 3), 1),
         # 128 < T <= 256: two DPP rows (32 lanes) per (o, i) row
         ((13, 200, 5), -2), ((7, 160, 3), -2), ((9, 250, 4), -2), ((5, 136, 7), -2), ((3, 255, 4), -2), ((6, 129, 2), -2), ((40, 200, 2, 1), -3)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape,time_dim", MULTI, ids=["x".join(map(str, s)) for s, _ in MULTI])
def test_multifeature_vs_oracle(shape, time_dim, dtype):
    """Feature dims after time (I > 1): SEQ bit-exact, the LDS-transposing SCAN kernel within tolerance,
    all three estimators, scalar and per-element gamma."""
    r, v, nv, done, term, gam, lp, lm = _rand_inputs(shape, dtype, seed=11)
    for gamma in (0.9, gam):
        oadv, ovt = O.gae(gamma, 0.9, v, nv, r, done, term, time_dim=time_dim)
        oret = O.td_lambda(gamma, 0.9, nv, r, done, term, time_dim=time_dim)
        ova, ovs = O.vtrace(gamma, lp, lm, v, nv, r, done, term, time_dim=time_dim)
        gcall = F._prepare_gae(cu(gamma), 0.9, cu(v), cu(nv), cu(r), cu(done), cu(term), time_dim)
        tcall = F._prepare_td_lambda(cu(gamma), 0.9, cu(nv), cu(r), cu(done), cu(term), None, time_dim)
        vcall = F._prepare_vtrace(cu(gamma), cu(lp), cu(lm), cu(v), cu(nv), cu(r), cu(done), cu(term), 1.0, 1.0, time_dim)
        for algo in algos(gcall):
            adv, vt, st = F._launch_gae(gcall, want_stats=True, algo=algo)
            if algo == SCAN:
                assert "interleaved" in F.last_kernel(), F.last_kernel()
            check(adv, oadv, algo == SEQ, done)
            check(vt, ovt, algo == SEQ, done)
            a64 = adv.double()
            ref_st = torch.stack([torch.tensor(float(a64.numel()), device=DEV), a64.sum(), (a64 * a64).sum()])
            torch.testing.assert_close(st, ref_st, rtol=1e-12, atol=1e-9)
            check(F._launch_td_lambda(tcall, dtype, algo=algo), oret, algo == SEQ)
            va, vs = F._launch_vtrace(vcall, algo=algo)
            check(va, ova, False)
            check(vs, ovs, False)
    adv, vt = F.generalized_advantage_estimate(0.9, 0.9, cu(v), cu(nv), cu(r), cu(done), cu(term), time_dim=time_dim)
    check(adv, O.gae(0.9, 0.9, v, nv, r, done, term, time_dim=time_dim)[0], False, done)


WIDE = [((5, 100, 20), -2), ((3, 100, 30), -2), ((7, 120, 13), -2), ((4, 200, 20), -2), ((9, 100, 11), -2), ((2, 100, 40), -2),
        ((6, 150, 8), -2), ((11, 65, 33, 1), -3), ((3, 256, 9), -2), ((4, 129, 7, 2), 1), ((300, 100, 12), -2)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape,time_dim", WIDE, ids=["x".join(map(str, s)) for s, _ in WIDE])
def test_many_feature_rows_on_wide_blocks(shape, time_dim, dtype):
    """Outer rows of more than a team's chunk (256 * V elements) and at most 1024 * V, 64 < T <= 256 (MAPPO-shaped rollouts with
    11 - 40 agents): GAE, TD(lambda), TD(1) and V-trace on the 1024-thread transposing kernel (gae_interleaved.h, NT = 1024), scalar and
    per-element gamma, compact (expanded) masks, per-block moments; shapes it does not take fall through to LOCKSTEP / SEQ."""
    r, v, nv, done, term, gam, lp, lm = _rand_inputs(shape, dtype, seed=31)
    td = time_dim if time_dim < 0 else time_dim - len(shape)
    T = shape[len(shape) + td]
    I = 1
    for k in shape[len(shape) + td + 1:]:
        I *= k
    V = 4 if dtype == torch.float32 else 2
    lg = 16 if T <= 128 else 32
    wide = 256 * V < T * I <= 1024 * V and 64 < T <= 256 and I != 16 and I <= 1024 // lg
    mshape = tuple(shape[:len(shape) + td + 1]) + (1,) * (-td - 1)
    g = torch.Generator().manual_seed(7)
    term_c = torch.rand(mshape, generator=g) < 0.03
    done_c = term_c | (torch.rand(mshape, generator=g) < 0.03)
    for gamma in (0.9, gam):
        for dn, tm, dd, tt in ((done, term, cu(done), cu(term)), (done_c.expand(shape), term_c.expand(shape), cu(done_c).expand(shape), cu(term_c).expand(shape))):
            oadv, ovt = O.gae(gamma, 0.9, v, nv, r, dn, tm, time_dim=time_dim)
            oret = O.td_lambda(gamma, 0.9, nv, r, dn, tm, time_dim=time_dim)
            oret1 = O.td1(gamma, nv, r, dn, tm, time_dim=time_dim)
            gcall = F._prepare_gae(cu(gamma), 0.9, cu(v), cu(nv), cu(r), dd, tt, time_dim)
            tcall = F._prepare_td_lambda(cu(gamma), 0.9, cu(nv), cu(r), dd, tt, None, time_dim)
            t1call = F._prepare_td1(cu(gamma), cu(nv), cu(r), dd, tt, True, time_dim)
            ova, ovs = O.vtrace(gamma, lp, lm, v, nv, r, dn, tm, time_dim=time_dim)
            vcall = F._prepare_vtrace(cu(gamma), cu(lp), cu(lm), cu(v), cu(nv), cu(r), dd, tt, 1.0, 1.0, time_dim)
            for algo in ((F.ALGO_AUTO, SCAN) if wide else (F.ALGO_AUTO,)):
                adv, vt, st = F._launch_gae(gcall, want_stats=True, algo=algo)
                if wide:
                    assert F.last_kernel() == "gae_scan_interleaved", F.last_kernel()
                check(adv, oadv, False, dn)
                check(vt, ovt, False, dn)
                a64 = adv.double()
                ref_st = torch.stack([torch.tensor(float(a64.numel()), device=DEV), a64.sum(), (a64 * a64).sum()])
                torch.testing.assert_close(st, ref_st, rtol=1e-12, atol=1e-9)
                ret = F._launch_td_lambda(tcall, dtype, algo=algo)
                if wide:
                    assert F.last_kernel() == "td_lambda_scan_interleaved", F.last_kernel()
                check(ret, oret, False)
                ret1 = F._launch_td1(t1call, dtype, algo=algo)
                if wide:
                    assert F.last_kernel() == "td1_scan_interleaved", F.last_kernel()
                check(ret1, oret1, False)
                va, vs = F._launch_vtrace(vcall, algo=algo)
                if wide:
                    assert F.last_kernel() == "vtrace_scan_interleaved", F.last_kernel()
                check(va, ova, False)
                check(vs, ovs, False)
    nadv, _ = F.gae_normalized(0.9, 0.9, cu(v), cu(nv), cu(r), cu(done), cu(term), time_dim=time_dim)
    torch.testing.assert_close(nadv.cpu(), O.normalize(O.gae(0.9, 0.9, v, nv, r, done, term, time_dim=time_dim)[0], 1e-4), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(37, 100, 30), (9, 200, 20), (21, 100, 5)], ids=lambda s: "x".join(map(str, s)))
def test_feature_rows_fp32_values_fp64_coefficients(shape):
    """python-float coefficients under default dtype float64 beside fp32 values: the transposing kernels keep their maps (and
    V-trace its outputs) in double -- up to 101 KB of LDS per 1024-thread block.  Every estimator against the oracle."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        r, v, nv, done, term, gam, lp, lm = _rand_inputs(shape, torch.float32, seed=41)
        oa, ovt = O.gae(0.97, 0.9, v, nv, r, done, term)
        a, vt = F.generalized_advantage_estimate(0.97, 0.9, cu(v), cu(nv), cu(r), cu(done), cu(term))
        assert F.last_kernel() == "gae_scan_interleaved"
        check(a, oa, False, done)
        check(vt, ovt, False, done)
        ot = O.td_lambda(0.97, 0.9, nv, r, done, term)
        t = F.td_lambda_return_estimate(0.97, 0.9, cu(nv), cu(r), cu(done), cu(term))
        assert F.last_kernel() == "td_lambda_scan_interleaved" and t.dtype == ot.dtype
        torch.testing.assert_close(t.cpu(), ot, rtol=1e-5, atol=1e-5)
        ova, ovs = O.vtrace(0.97, lp, lm, v, nv, r, done, term)
        va, vs = F.vtrace_advantage_estimate(0.97, cu(lp), cu(lm), cu(v), cu(nv), cu(r), cu(done), cu(term))
        assert F.last_kernel() == "vtrace_scan_interleaved" and va.dtype == ova.dtype == torch.float64
        torch.testing.assert_close(va.cpu(), ova, rtol=1e-5, atol=1e-5)   # fp32 inputs (expf): the fp32 bar
        torch.testing.assert_close(vs.cpu(), ovs, rtol=1e-5, atol=1e-5)
    finally:
        torch.set_default_dtype(old)


def test_successive_trajectories_equal_concatenation():
    """test_cost.py:14984-15135: a batch with a `terminated` at T//2-1 equals the two halves computed apart."""
    T = 64
    r, v, nv, done, term, gam, lp, lm = _rand_inputs((8, T, 1), torch.float32, seed=3, p_term=0.0, p_trunc=0.0)
    term[:, T // 2 - 1] = True
    done = term.clone()
    h = T // 2
    for algo in (SEQ, SCAN):
        full = F._launch_gae(F._prepare_gae(0.99, 0.95, cu(v), cu(nv), cu(r), cu(done), cu(term), -2), algo=algo)[0]
        a = F._launch_gae(F._prepare_gae(0.99, 0.95, cu(v[:, :h]), cu(nv[:, :h]), cu(r[:, :h]), cu(done[:, :h]), cu(term[:, :h]), -2), algo=algo)[0]
        b = F._launch_gae(F._prepare_gae(0.99, 0.95, cu(v[:, h:]), cu(nv[:, h:]), cu(r[:, h:]), cu(done[:, h:]), cu(term[:, h:]), -2), algo=algo)[0]
        torch.testing.assert_close(full, torch.cat([a, b], 1), rtol=1e-6, atol=1e-6)
        fr = F._launch_td_lambda(F._prepare_td_lambda(0.99, 0.95, cu(nv), cu(r), cu(done), cu(term), None, -2), torch.float32, algo=algo)
        ar = F._launch_td_lambda(F._prepare_td_lambda(0.99, 0.95, cu(nv[:, :h]), cu(r[:, :h]), cu(done[:, :h]), cu(term[:, :h]), None, -2), torch.float32, algo=algo)
        br = F._launch_td_lambda(F._prepare_td_lambda(0.99, 0.95, cu(nv[:, h:]), cu(r[:, h:]), cu(done[:, h:]), cu(term[:, h:]), None, -2), torch.float32, algo=algo)
        torch.testing.assert_close(fr, torch.cat([ar, br], 1), rtol=1e-6, atol=1e-6)


def test_tdlambda_equals_gae_when_next_value_is_shifted_value():
    """test_cost.py:13753-13822: TD(lambda) advantage == GAE advantage when next_value[t] = value[t+1]."""
    g = torch.Generator().manual_seed(0)
    v_all = torch.randn(6, 101, 1, generator=g)
    v, nv = v_all[:, :-1].contiguous(), v_all[:, 1:].contiguous()
    r = torch.randn(6, 100, 1, generator=g)
    done = torch.zeros(6, 100, 1, dtype=torch.bool)
    adv, _ = F.generalized_advantage_estimate(0.98, 0.94, cu(v), cu(nv), cu(r), cu(done))
    tda = F.td_lambda_advantage_estimate(0.98, 0.94, cu(v), cu(nv), cu(r), cu(done))
    vtda = F.vec_td_lambda_advantage_estimate(0.98, 0.94, cu(v), cu(nv), cu(r), cu(done))
    torch.testing.assert_close(adv, tda, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(adv, vtda, rtol=1e-4, atol=1e-4)


def test_reward2go_closed_form():
    """functional.py:1381-1395 docstring known answer and test_cost.py:16287-16309 closed form."""
    reward = torch.ones(1, 10, device=DEV)
    done = torch.zeros(1, 10, dtype=torch.bool, device=DEV)
    done[:, [3, 7]] = True
    out = F.reward2go(reward, done, 0.99, time_dim=-1)
    exp = torch.tensor([3.9404, 2.9701, 1.9900, 1.0, 3.9404, 2.9701, 1.9900, 1.0, 1.9900, 1.0], device=DEV).view(1, 10)
    torch.testing.assert_close(out, exp, rtol=1e-4, atol=1e-4)


def test_errors_on_device():
    r = torch.randn(3, 8, 1, device=DEV)
    d = torch.zeros(3, 8, 1, dtype=torch.bool, device=DEV)
    with pytest.raises(RuntimeError, match="share a unique shape"):
        F.generalized_advantage_estimate(0.9, 0.9, r, r, r[:, :4], d)
    with pytest.raises(TypeError):
        F.generalized_advantage_estimate(0.9, 0.9, r, r, r, d, d, -2)
    with pytest.raises(RuntimeError, match="rolling_gamma=False is expected only"):
        F.td_lambda_return_estimate(0.9, 0.9, r, r, d, rolling_gamma=False)
    with pytest.raises(NotImplementedError, match="The vectorized version of TD"):
        F.vec_td_lambda_return_estimate(torch.full_like(r, 0.9), 0.9, r, r, d, d, rolling_gamma=False)
    tm = d.clone()
    tm[:, 3] = True
    with pytest.raises(NotImplementedError, match="When using rolling_gamma=False"):
        F.vec_td_lambda_return_estimate(torch.full_like(r, 0.9), 0.9, r, r, tm, tm, rolling_gamma=False)


FUSED_SHAPES = [(64, 32, 1), (65536, 32, 1), (3000, 32, 1), (37, 100, 1), (5000, 100, 1), (33, 3, 1), (9, 1, 1),
                (130, 128, 1), (2000, 1024, 1), (5, 7, 1), (100000, 32, 1), (200000, 32, 1),  # the last two: 5..8 chunks per team
                (32768, 32, 1), (16384, 32, 1), (8192, 32, 1)]  # the per-GPU shards of the [65536 x 32] batch on 2 / 4 / 8 GPUs


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape", FUSED_SHAPES, ids=[f"{s[0]}x{s[1]}" for s in FUSED_SHAPES])
def test_fused_gae_normalise_vs_oracle(shape, dtype):
    """Single-launch GAE + normalise against oracle GAE followed by oracle normalise; value_target and
    the returned moments against the unfused path.  Called twice: the in-launch exchange is epoch-tagged."""
    if shape[1] > 256 * (4 if dtype == torch.float32 else 2):
        pytest.skip("row longer than one chunk: not eligible for the fused kernel (covered by the fallback test)")
    chunk = 256 * (4 if dtype == torch.float32 else 2)
    if -(-shape[0] // (chunk // shape[1])) > 8 * 4 * torch.cuda.get_device_properties(0).multi_processor_count:
        pytest.skip("more than 8 chunks per team: the advantages do not fit in registers (unfused route)")
    r, v, nv, done, term, gam, lp, lm = _rand_inputs(shape, dtype, seed=7)
    oadv, ovt = O.gae(0.99, 0.95, v, nv, r, done, term)
    onorm = O.normalize(oadv, 1e-4)
    call = F._prepare_gae(0.99, 0.95, cu(v), cu(nv), cu(r), cu(done), cu(term), -2)
    # north_star's bar.  Measured (profiles/r03_parity_errors.txt, tests/report_parity_errors.py): normalised advantage
    # 2e-7 .. 7e-7 mixed error at fp32, <= 3e-14 at fp64 (the scan's error divided by std(A) ~ 4, plus the rounding of mean / std)
    t = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-12, atol=1e-12)
    n_el = shape[0] * shape[1]
    a64 = F._launch_gae(call, algo=F.ALGO_SCAN)[0].double()  # same scan, un-normalised
    ref_st = torch.stack([torch.tensor(float(n_el), device=DEV, dtype=torch.float64), a64.sum(), (a64 * a64).sum()])
    for rep in range(3):
        adv, vt, st = F._launch_gae_normalize(call, 1e-4, algo=F.ALGO_FUSED, want_stats=True)
        assert F.last_kernel() == "gae_fused"
        check(vt, ovt, False, done)
        if n_el > 1:
            torch.testing.assert_close(adv.cpu(), onorm, **t)
        torch.testing.assert_close(st, ref_st, rtol=1e-10, atol=1e-7)  # 8 mantissa bits carry the epoch tag
    # AUTO picks the fused kernel; the unfused pipeline gives the same result
    adv_auto, vt_auto = F.gae_normalized(0.99, 0.95, cu(v), cu(nv), cu(r), cu(done), cu(term))
    adv_unf, vt_unf, _ = F._launch_gae_normalize(call, 1e-4, algo=F.ALGO_SCAN)
    if n_el > 1:
        torch.testing.assert_close(adv_auto, adv_unf, **t)
    assert torch.equal(vt_auto, vt_unf)


def test_fused_falls_back_when_not_eligible():
    r, v, nv, done, term, gam, lp, lm = _rand_inputs((3, 2500, 1), torch.float32, seed=9)
    adv, vt = F.gae_normalized(0.99, 0.95, cu(v), cu(nv), cu(r), cu(done), cu(term))
    assert F.last_kernel() == "normalize"
    oadv, ovt = O.gae(0.99, 0.95, v, nv, r, done, term)
    torch.testing.assert_close(adv.cpu(), O.normalize(oadv, 1e-4), rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError, match="FUSED family needs"):
        F._launch_gae_normalize(F._prepare_gae(0.99, 0.95, cu(v), cu(nv), cu(r), cu(done), cu(term), -2), 1e-4, algo=F.ALGO_FUSED)


# few, very long rows: one chunk per block with decoupled look-back between the chunks of a row
LOOKBACK_SHAPES = [(1, 100000, 1), (4, 50000, 1), (3, 8193, 1), (2, 16384, 1), (100, 8192, 1), (1, 9000, 1)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape", LOOKBACK_SHAPES, ids=[f"{s[0]}x{s[1]}" for s in LOOKBACK_SHAPES])
def test_lookback_long_rows_vs_oracle(shape, dtype):
    """Rows of 8 team chunks or more with too few rows to fill the chip take the look-back kernel (all three
    estimators, scalar and per-element gamma); same bar as the other SCAN kernels, and the result equals
    the in-block chunk loop's (the path taken when no workspace is passed) to the same bar."""
    r, v, nv, done, term, gam, lp, lm = _rand_inputs(shape, dtype, seed=5, p_term=0.001, p_trunc=0.001)
    chunk = 256 * (4 if dtype == torch.float32 else 2)
    eligible = shape[1] >= 8 * chunk
    for gamma in (0.99, gam):
        oadv, ovt = O.gae(gamma, 0.95, v, nv, r, done, term)
        oret = O.td_lambda(gamma, 0.95, nv, r, done, term)
        ova, ovs = O.vtrace(gamma, lp, lm, v, nv, r, done, term)
        gcall = F._prepare_gae(cu(gamma), 0.95, cu(v), cu(nv), cu(r), cu(done), cu(term), -2)
        tcall = F._prepare_td_lambda(cu(gamma), 0.95, cu(nv), cu(r), cu(done), cu(term), None, -2)
        vcall = F._prepare_vtrace(cu(gamma), cu(lp), cu(lm), cu(v), cu(nv), cu(r), cu(done), cu(term), 1.0, 1.0, -2)
        for rep in range(2):  # the second call re-uses the (re-zeroed) records of the first
            adv, vt, _ = F._launch_gae(gcall, algo=SCAN)
            assert ("lookback" in F.last_kernel()) == eligible, F.last_kernel()
            check(adv, oadv, False, done)
            check(vt, ovt, False, done)
            ret = F._launch_td_lambda(tcall, dtype, algo=SCAN)
            assert ("lookback" in F.last_kernel()) == eligible, F.last_kernel()
            check(ret, oret, False)
            va, vs = F._launch_vtrace(vcall, algo=SCAN)
            assert ("lookback" in F.last_kernel()) == eligible, F.last_kernel()
            check(va, ova, False)
            check(vs, ovs, False)
        # stats requested -> the in-block chunk loop (partials per block), same values
        adv2, vt2, st = F._launch_gae(gcall, want_stats=True, algo=SCAN)
        assert "long" in F.last_kernel()
        check(adv2, oadv, False, done)


def test_lookback_undiscounted_carry_crosses_every_chunk():
    """gamma = lmbda = 1 and no done: the carry out of the last chunk reaches step 0 through every
    aggregate, so a wrong composition order or a stale record would show at full magnitude."""
    shape = (2, 40000, 1)
    g = torch.Generator().manual_seed(9)
    r = torch.randn(shape, generator=g, dtype=torch.float64)
    v = torch.randn(shape, generator=g, dtype=torch.float64)
    nv = torch.randn(shape, generator=g, dtype=torch.float64)
    done = torch.zeros(shape, dtype=torch.bool)
    oadv, ovt = O.gae(1.0, 1.0, v, nv, r, done, done)
    gcall = F._prepare_gae(1.0, 1.0, cu(v), cu(nv), cu(r), cu(done), cu(done), -2)
    adv, vt, _ = F._launch_gae(gcall, algo=SCAN)
    assert "lookback" in F.last_kernel()
    torch.testing.assert_close(adv.cpu(), oadv, rtol=1e-11, atol=1e-9)
    oret = O.td_lambda(1.0, 1.0, nv, r, done, done)
    tcall = F._prepare_td_lambda(1.0, 1.0, cu(nv), cu(r), cu(done), cu(done), None, -2)
    torch.testing.assert_close(F._launch_td_lambda(tcall, torch.float64, algo=SCAN).cpu(), oret, rtol=1e-11, atol=1e-9)


def test_entry_points_inside_a_compiled_region():
    """torch.compile (dynamo, eager backend: no code generator needed): the estimators are opaque
    calls inside a compiled function and give the eager result."""
    r, v, nv, done, term, gam, lp, lm = (cu(t) for t in _rand_inputs((32, 64, 1), torch.float32, seed=2))

    def fn(v, nv, r):
        adv, vt = F.generalized_advantage_estimate(0.99, 0.95, v * 1.0, nv, r, done, term)
        ret = F.td_lambda_return_estimate(0.99, 0.95, nv, r, done, term)
        return adv * 2.0 + vt, ret + 1.0

    want = fn(v, nv, r)
    got = torch.compile(fn, backend="eager")(v, nv, r)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(0, 8, 1), (4, 0, 1), (3, 5, 0)], ids=["no_rows", "no_steps", "no_features"])
def test_empty_inputs_give_empty_outputs(shape):
    """The reference's loops run zero steps on empty tensors and return empty tensors of the input shape."""
    z = torch.zeros(shape, device=DEV)
    d = torch.zeros(shape, dtype=torch.bool, device=DEV)
    adv, vt = F.generalized_advantage_estimate(0.9, 0.9, z, z, z, d, d)
    assert adv.shape == vt.shape == torch.Size(shape) and adv.dtype == torch.float32
    assert F.td_lambda_return_estimate(0.9, 0.9, z, z, d, d).shape == torch.Size(shape)
    va, vs = F.vtrace_advantage_estimate(0.9, z, z, z, z, z, d, d)
    assert va.shape == vs.shape == torch.Size(shape)
    adv, vt = F.gae_normalized(0.9, 0.9, z, z, z, d, d)
    assert adv.shape == torch.Size(shape)


def test_workspace_error_word_is_clear_after_fused_calls():
    r, v, nv, done, term, gam, lp, lm = (cu(t) for t in _rand_inputs((2048, 32, 1), torch.float32, seed=4))
    adv, vt = F.gae_normalized(0.99, 0.95, v, nv, r, done, term)
    assert F.last_kernel() == "gae_fused" and torch.isfinite(adv).all()
    assert F.workspace_error() is False


BCAST = [((40, 50, 5, 1), -3), ((6, 9, 2, 5), 1), ((33, 17, 64), -2), ((4096, 100, 5, 1), -3), ((7, 3, 2), -2), ((50, 16, 4), -2),
         ((19, 101, 3), -2), ((64, 25, 4), 1), ((4, 300, 130), -2), ((33, 80, 5, 1), -3), ((21, 96, 4), -2), ((9, 113, 9), -2), ((13, 200, 5), -2), ((7, 160, 3, 1), -3)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape,time_dim", BCAST, ids=["x".join(map(str, s)) for s, _ in BCAST])
def test_broadcast_masks_read_in_place(shape, time_dim, dtype):
    """Per-env masks `expand`ed over the feature / agent dims (stride 0, as DoneTransform produces) go to the
    kernels as compact [*B, T] arrays (mask_bcast = 1): SEQ bit-exact, the interleaved SCAN kernel within
    tolerance, all three estimators; equal to what the materialised masks give."""
    r, v, nv, _, _, gam, lp, lm = _rand_inputs(shape, dtype, seed=21)
    td = time_dim if time_dim < 0 else time_dim - len(shape)
    mshape = tuple(shape[:len(shape) + td + 1]) + (1,) * (-td - 1)
    g = torch.Generator().manual_seed(5)
    term_c = torch.rand(mshape, generator=g) < 0.03
    done_c = term_c | (torch.rand(mshape, generator=g) < 0.03)
    term, done = term_c.expand(shape), done_c.expand(shape)
    oadv, ovt = O.gae(0.9, 0.9, v, nv, r, done, term, time_dim=time_dim)
    oret = O.td_lambda(gam, 0.9, nv, r, done, term, time_dim=time_dim)
    ova, ovs = O.vtrace(0.9, lp, lm, v, nv, r, done, term, time_dim=time_dim)
    dd, tt = cu(done_c).expand(shape), cu(term_c).expand(shape)
    gcall = F._prepare_gae(0.9, 0.9, cu(v), cu(nv), cu(r), dd, tt, time_dim)
    tcall = F._prepare_td_lambda(cu(gam), 0.9, cu(nv), cu(r), dd, tt, None, time_dim)
    vcall = F._prepare_vtrace(0.9, cu(lp), cu(lm), cu(v), cu(nv), cu(r), dd, tt, 1.0, 1.0, time_dim)
    assert gcall.mask_bcast == tcall.mask_bcast == vcall.mask_bcast == 1
    for algo in algos(gcall):
        adv, vt, st = F._launch_gae(gcall, want_stats=True, algo=algo)
        check(adv, oadv, algo == SEQ, done)
        check(vt, ovt, algo == SEQ, done)
        check(F._launch_td_lambda(tcall, dtype, algo=algo), oret, algo == SEQ)
        va, vs = F._launch_vtrace(vcall, algo=algo)
        check(va, ova, False)
        check(vs, ovs, False)
    # public entry points, and the normalised form (not fused for I > 1: scan + normalise)
    adv, vt = F.generalized_advantage_estimate(0.9, 0.9, cu(v), cu(nv), cu(r), dd, tt, time_dim=time_dim)
    check(adv, oadv, False, done)
    nadv, _ = F.gae_normalized(0.9, 0.9, cu(v), cu(nv), cu(r), dd, tt, time_dim=time_dim)
    torch.testing.assert_close(nadv.cpu(), O.normalize(oadv, 1e-4), rtol=1e-5, atol=1e-5)
    # rolling_gamma=False (O(T^2) SEQ kernel) reads the compact masks too
    if shape[len(shape) + td] <= 50:
        got = F.td_lambda_return_estimate(cu(gam), 0.9, cu(nv), cu(r), dd, tt, rolling_gamma=False, time_dim=time_dim)
        check(got, O.td_lambda(gam, 0.9, nv, r, done, term, rolling_gamma=False, time_dim=time_dim), True)


LONG_IL = [((16, 400, 5), -2), ((3, 2000, 3, 2), 1), ((64, 1000, 5, 1), -3), ((5, 300, 4), -2), ((7, 257, 2), -2), ((2, 5000, 8), -2),
           ((300, 256, 7), -2), ((40, 700, 12), -2), ((6, 300, 9), -2), ((3, 130, 4, 4), 1), ((5, 260, 17), -2), ((2, 600, 31), -2),
           ((70, 40, 31), -2), ((1100, 300, 3), -2),
           # I = G * NI with the G groups of a step in neighbouring lanes (round 3): short rows (T <= 256 / G) and long ones
           ((50, 40, 16), -2), ((20, 300, 24), -2), ((64, 20, 10), -2), ((9, 70, 20), -2), ((5, 33, 28), -2), ((33, 129, 14), -2),
           ((300, 64, 16), -2), ((3, 1000, 16), -2)]


@pytest.mark.parametrize("shape", [(16, 400, 5), (40, 700, 12)], ids=["16x400x5", "40x700x12"])
def test_long_rows_few_features_fp32_values_fp64_gamma(shape):
    """fp32 values with an fp64 gamma tensor (arithmetic in fp64, outputs fp32) on the lock-step kernels."""
    r, v, nv, done, term, gam, lp, lm = _rand_inputs(shape, torch.float32, seed=32, p_term=0.01, p_trunc=0.01)
    gam = gam.double()
    oadv, ovt = O.gae(gam, 0.9, v, nv, r, done, term)
    adv, vt = F.generalized_advantage_estimate(cu(gam), 0.9, cu(v), cu(nv), cu(r), cu(done), cu(term))
    assert F.last_kernel() == "gae_scan_lockstep" and adv.is_contiguous()
    torch.testing.assert_close(adv.cpu(), oadv, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(vt.cpu(), ovt, rtol=1e-5, atol=1e-5)
    ova, ovs = O.vtrace(gam, lp, lm, v, nv, r, done, term)
    va, vs = F.vtrace_advantage_estimate(cu(gam), cu(lp), cu(lm), cu(v), cu(nv), cu(r), cu(done), cu(term))
    assert F.last_kernel() == "vtrace_scan_lockstep" and va.dtype == torch.float64
    torch.testing.assert_close(va.cpu(), ova, rtol=1e-5, atol=1e-5)  # rho = exp(fp32): expf on the device, fp32 tolerance
    torch.testing.assert_close(vs.cpu(), ovs, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape,time_dim", LONG_IL, ids=["x".join(map(str, s)) for s, _ in LONG_IL])
@pytest.mark.parametrize("bcast", [False, True], ids=["full_masks", "bcast_masks"])
def test_long_rows_with_few_features(shape, time_dim, dtype, bcast):
    """T * I beyond one chunk of the LDS-transposing kernel: 2 <= I <= 31 takes the lock-step kernels (a lane per
    time step, chunk loop with I carries; groups of <= 8 feature elements beyond 8) -- never the uncoalesced SEQ walk
    on few rows; all three estimators and the normalised GAE, SCAN-family tolerance."""
    r, v, nv, done, term, gam, lp, lm = _rand_inputs(shape, dtype, seed=31, p_term=0.01, p_trunc=0.01)
    if bcast:
        td = time_dim if time_dim < 0 else time_dim - len(shape)
        idx = (Ellipsis,) + (slice(0, 1),) * (-td - 1)
        done, term = done[idx].expand(shape), term[idx].expand(shape)
    oadv, ovt = O.gae(gam, 0.9, v, nv, r, done, term, time_dim=time_dim)
    oret = O.td_lambda(0.97, 0.9, nv, r, done, term, time_dim=time_dim)
    ova, ovs = O.vtrace(gam, lp, lm, v, nv, r, done, term, time_dim=time_dim)
    dd, tt = (cu(done), cu(term)) if not bcast else (cu(done[idx]).expand(shape), cu(term[idx]).expand(shape))
    adv, vt = F.generalized_advantage_estimate(cu(gam), 0.9, cu(v), cu(nv), cu(r), dd, tt, time_dim=time_dim)
    assert "scan" in F.last_kernel() and adv.is_contiguous()
    check(adv, oadv, False, done)
    check(vt, ovt, False, done)
    check(F.td_lambda_return_estimate(0.97, 0.9, cu(nv), cu(r), dd, tt, time_dim=time_dim), oret, False)
    assert "scan" in F.last_kernel()
    va, vs = F.vtrace_advantage_estimate(cu(gam), cu(lp), cu(lm), cu(v), cu(nv), cu(r), dd, tt, time_dim=time_dim)
    assert "scan" in F.last_kernel()
    check(va, ova, False)
    check(vs, ovs, False)
    nadv, vt2 = F.gae_normalized(cu(gam), 0.9, cu(v), cu(nv), cu(r), dd, tt, time_dim=time_dim)
    torch.testing.assert_close(nadv.cpu(), O.normalize(oadv, 1e-4), rtol=1e-5, atol=1e-5)
    check(vt2, ovt, False, done)
    # the exact family is still there on request
    gcall = F._prepare_gae(cu(gam), 0.9, cu(v), cu(nv), cu(r), dd, tt, time_dim)
    adv_seq, _, _ = F._launch_gae(gcall, algo=SEQ)
    assert F.last_kernel() == "gae_seq"
    check(adv_seq, oadv, True)
    # and the lock-step family on request, whatever AUTO picked (I <= 31: groups of <= 8 feature elements)
    if gcall.I <= 31:
        adv_ls, vt_ls, st = F._launch_gae(gcall, want_stats=True, algo=LOCKSTEP)
        assert F.last_kernel() == "gae_scan_lockstep"
        check(adv_ls, oadv, False, done)
        check(vt_ls, ovt, False, done)
        a64 = adv_ls.double()
        torch.testing.assert_close(st, torch.stack([torch.tensor(float(a64.numel()), device=DEV), a64.sum(), (a64 * a64).sum()]),
                                   rtol=1e-12, atol=1e-9)
        tcall = F._prepare_td_lambda(0.97, 0.9, cu(nv), cu(r), dd, tt, None, time_dim)
        check(F._launch_td_lambda(tcall, dtype, algo=LOCKSTEP), oret, False)
        vcall = F._prepare_vtrace(cu(gam), cu(lp), cu(lm), cu(v), cu(nv), cu(r), dd, tt, 1.0, 1.0, time_dim)
        va, vs = F._launch_vtrace(vcall, algo=LOCKSTEP)
        assert F.last_kernel() == "vtrace_scan_lockstep"
        check(va, ova, False)
        check(vs, ovs, False)
