"""GPU (-m gpu): the "next" rows of SURVEY.md 8(f).2 -- td0_*, td1_*, vec_td1_*, reward2go -- and V-trace with threshold
tensors, through the HIP path, against golden fixtures generated from the imported reference (tools/gen_golden_next.py)
and against the CPU oracle.  Bars: SEQ family bit-identical to the reference loops; SCAN / LOCKSTEP fp32 rtol=atol=1e-5,
fp64 1e-12; the reference's own vec variants (truncated filters) at the reference's test tolerance 1e-4."""
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


def check(out, ref, exact):
    out = out.cpu()
    assert out.dtype == ref.dtype and out.shape == ref.shape
    if exact:
        assert torch.equal(out, ref), f"max abs diff {(out - ref).abs().max().item()}"
    else:
        torch.testing.assert_close(out, ref, **tol(ref.dtype))


@pytest.mark.parametrize("meta", G.cases("td", "td0"), ids=G.ids("td", "td0"))
def test_td0_golden(meta):
    a = G.arrays("td", meta)
    ret = F.td0_return_estimate(meta["gamma"], cu(a["next_state_value"]), cu(a["reward"]), cu(a["terminated"]))
    check(ret, a["returns"], True)
    adv = F.td0_advantage_estimate(meta["gamma"], cu(a["state_value"]), cu(a["next_state_value"]), cu(a["reward"]), cu(a["done"]),
                                   cu(a["terminated"]))
    check(adv, a["advantage"], True)


def td1_algos(call):
    chunk = 256 * (4 if call.vdt == torch.float32 else 2)
    if not call.rolling:
        return [SEQ]
    out = [SEQ, SCAN] if (call.I == 1 or call.T * call.I <= chunk) else [SEQ]
    if 2 <= call.I <= 31:
        out.append(LOCKSTEP)
    return out


@pytest.mark.parametrize("meta", G.cases("td", "td1"), ids=G.ids("td", "td1"))
def test_td1_golden(meta):
    """td1_return_estimate: rolling and rolling_gamma=False (which starts from g = 0, functional.py:550-561), masks where
    terminated is not a subset of done, every kernel family that takes the shape."""
    a = G.arrays("td", meta)
    gamma = cu(G.coef(meta, a, "gamma"))
    args = (cu(a["next_state_value"]), cu(a["reward"]), cu(a["done"]), cu(a["terminated"]))
    call = F._prepare_td1(gamma, *args, meta["rolling_gamma"], meta["time_dim"])
    for algo in td1_algos(call):
        ret = F._launch_td1(call, a["returns"].dtype, algo=algo)
        fam = F.last_kernel()
        assert ("seq" in fam) == (algo == SEQ), fam
        check(ret, a["returns"], algo == SEQ)
    ret = F.td1_return_estimate(gamma, *args, meta["rolling_gamma"], **G.time_kw(meta))
    check(ret, a["returns"], False)
    if "advantage" in a:
        adv = F.td1_advantage_estimate(gamma, cu(a["state_value"]), *args, meta["rolling_gamma"], time_dim=meta["time_dim"])
        check(adv, a["advantage"], False)
    if "vec_returns" in a:  # the reference's vec variant is filter-truncated: its own tests compare at 1e-4
        vec = F.vec_td1_return_estimate(gamma, *args, meta["rolling_gamma"], time_dim=meta["time_dim"])
        torch.testing.assert_close(vec.cpu(), a["vec_returns"], rtol=1e-4, atol=1e-4)
        vadv = F.vec_td1_advantage_estimate(gamma, cu(a["state_value"]), *args, meta["rolling_gamma"], time_dim=meta["time_dim"])
        torch.testing.assert_close(vadv.cpu(), a["vec_returns"] - a["state_value"], rtol=1e-4, atol=1e-4)


def test_td1_not_rolling_needs_gamma_tensor():
    r = torch.randn(3, 8, 1, device=DEV)
    d = torch.zeros(3, 8, 1, dtype=torch.bool, device=DEV)
    with pytest.raises(RuntimeError, match="rolling_gamma=False is expected only with time-sensitive gamma values"):
        F.td1_return_estimate(0.9, r, r, d, rolling_gamma=False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape", [(300, 32, 1), (37, 100, 1), (3, 2500, 1), (2, 4099, 1), (40, 50, 5), (19, 101, 3), (4, 300, 130)],
                         ids=lambda s: "x".join(map(str, s)))
def test_td1_vs_oracle(shape, dtype):
    g = torch.Generator().manual_seed(31)
    r, nv = (torch.randn(shape, generator=g, dtype=torch.float32).to(dtype) for _ in range(2))
    term = torch.rand(shape, generator=g) < 0.02
    done = term | (torch.rand(shape, generator=g) < 0.02)
    gam = (0.9 + 0.1 * torch.rand(shape, generator=g)).to(dtype)
    for gamma in (0.99, gam):
        want = O.td1(gamma, nv, r, done, term)
        call = F._prepare_td1(cu(gamma), cu(nv), cu(r), cu(done), cu(term), None, -2)
        for algo in td1_algos(call):
            check(F._launch_td1(call, dtype, algo=algo), want, algo == SEQ)
        check(F.td1_return_estimate(cu(gamma), cu(nv), cu(r), cu(done), cu(term)), want, False)
    if shape[1] <= 101:  # O(T^2)
        want = O.td1(gam, nv, r, done, term, False)
        check(F.td1_return_estimate(cu(gam), cu(nv), cu(r), cu(done), cu(term), False), want, True)


@pytest.mark.parametrize("meta", G.cases("td", "reward2go"), ids=G.ids("td", "reward2go"))
def test_reward2go_golden(meta):
    a = G.arrays("td", meta)
    kw = {} if meta["time_dim"] == -2 else {"time_dim": meta["time_dim"]}
    out = F.reward2go(cu(a["reward"]), cu(a["done"]), meta["gamma"], **kw)
    check(out, a["out"], False)


@pytest.mark.parametrize("meta", G.cases("vtrace_thresh"), ids=G.ids("vtrace_thresh"))
def test_vtrace_threshold_tensors_golden(meta):
    """rho.clamp_max(tensor) with broadcastable threshold tensors (functional.py:1311-1330)."""
    a = G.arrays("vtrace_thresh", meta)
    adv, vs = F.vtrace_advantage_estimate(meta["gamma"], cu(a["log_pi"]), cu(a["log_mu"]), cu(a["state_value"]),
                                          cu(a["next_state_value"]), cu(a["reward"]), cu(a["done"]), cu(a["terminated"]),
                                          cu(a["rho_thresh"]), cu(a["c_thresh"]), time_dim=meta["time_dim"])
    assert F.last_kernel() == "vtrace_seq"
    check(adv, a["advantage"], False)
    check(vs, a["vs"], False)


@pytest.mark.parametrize("shape,time_dim", [((7, 33, 3, 2), -3), ((5, 4, 9), -1), ((3, 50, 4), 1), ((2, 3, 40, 5, 1), 2),
                                            ((4096, 100, 5, 1), -3)])
@pytest.mark.parametrize("masks", ["full", "expanded"])
@pytest.mark.parametrize("eps", [None, 1e-4])
@pytest.mark.parametrize("host", ["cxx", "python"])
def test_gae_fast_path_any_time_dim(shape, time_dim, masks, eps, host, monkeypatch):
    """Both hosts of the fast path (csrc/host_fast.cpp's module, and the python route a tree without it takes): the fast path (float gamma / lmbda, dense tensors) with time anywhere and per-step masks expanded over the feature
    dims, against the CPU oracle and against the general path (0-dim tensor gamma routes around the fast path)."""
    if host == "cxx":
        assert F.host_fast_path(), "lib/_trl_host.so is not built (make -C torchrl_amd/csrc)"
    else:
        F.host_fast_path()
        monkeypatch.setattr(F, "_HOST", None)
    g = torch.Generator().manual_seed(3)
    r, v, nv = (torch.randn(shape, generator=g) for _ in range(3))
    td = time_dim % len(shape)
    mshape = shape if masks == "full" else shape[:td + 1] + (1,) * (len(shape) - td - 1)
    term_c = torch.rand(mshape, generator=g) < 0.05
    done_c = term_c | (torch.rand(mshape, generator=g) < 0.05)
    ra, rvt = O.gae(0.99, 0.95, v, nv, r, done_c.expand(shape).contiguous(), term_c.expand(shape).contiguous(), time_dim=time_dim)
    dv, dnv, dr = cu(v), cu(nv), cu(r)
    dd, dt_ = cu(done_c).expand(shape), cu(term_c).expand(shape)   # stride-0 views over the feature dims when "expanded"
    if eps is None:
        adv, vt = F.generalized_advantage_estimate(0.99, 0.95, dv, dnv, dr, dd, dt_, time_dim=time_dim)
        gadv, gvt = F.generalized_advantage_estimate(torch.tensor(0.99), 0.95, dv, dnv, dr, dd, dt_, time_dim=time_dim)
        torch.testing.assert_close(adv.cpu(), ra, **tol(ra.dtype))
        torch.testing.assert_close(vt.cpu(), rvt, **tol(ra.dtype))
        torch.testing.assert_close(adv, gadv, **tol(ra.dtype))
        torch.testing.assert_close(vt, gvt, **tol(ra.dtype))
    else:
        adv, vt = F.gae_normalized(0.99, 0.95, dv, dnv, dr, dd, dt_, eps=eps, time_dim=time_dim)
        ref = O.normalize(ra, eps)
        torch.testing.assert_close(adv.cpu(), ref, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(vt.cpu(), rvt, **tol(ra.dtype))


@pytest.fixture
def default_dtype(request):
    old = torch.get_default_dtype()
    torch.set_default_dtype(request.param)
    yield request.param
    torch.set_default_dtype(old)


@pytest.mark.parametrize("default_dtype", [torch.float32, torch.float64], ids=["def32", "def64"], indirect=True)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape,time_dim", [((6, 40, 1), -2), ((3, 2500, 1), -2), ((9, 20, 3, 1), -3), ((5, 33), -1)])
@pytest.mark.parametrize("masks", ["full", "expanded"])
def test_host_module_and_python_routes_agree(shape, time_dim, masks, dtype, default_dtype, monkeypatch):
    """csrc/host_fast.cpp makes the same C-ABI call the python host code makes: every estimator it serves, bit for bit and
    dtype for dtype, under both torch default dtypes (they decide how python-float coefficients are rounded)."""
    assert F.host_fast_path(), "lib/_trl_host.so is not built (make -C torchrl_amd/csrc)"
    g = torch.Generator().manual_seed(11)
    mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32).to(dtype).to(DEV)
    r, v, nv = mk(), mk(), mk()
    lp = torch.rand(shape, generator=g).clamp_min(1e-3).log().to(dtype).to(DEV)
    lm = torch.rand(shape, generator=g).clamp_min(1e-3).log().to(dtype).to(DEV)
    td = time_dim % len(shape)
    mshape = shape if masks == "full" else shape[:td + 1] + (1,) * (len(shape) - td - 1)
    term = (torch.rand(mshape, generator=g) < 0.05).to(DEV)
    done = (term | (torch.rand(mshape, generator=g) < 0.05).to(DEV)).expand(shape)
    term = term.expand(shape)
    calls = {
        "gae": lambda: F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term, time_dim=time_dim),
        "vec_gae": lambda: F.vec_generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term, time_dim=time_dim),
        "gae_normalized": lambda: F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4, time_dim=time_dim),
        "td_lambda": lambda: (F.td_lambda_return_estimate(0.99, 0.95, nv, r, done, term, time_dim=time_dim),),
        "td_lambda_rolling": lambda: (F.td_lambda_return_estimate(0.99, 0.95, nv, r, done, term, True, time_dim=time_dim),),
        "vec_td_lambda": lambda: (F.vec_td_lambda_return_estimate(0.99, 0.95, nv, r, done, term, False, time_dim=time_dim),),
        "td_lambda_adv": lambda: (F.td_lambda_advantage_estimate(0.99, 0.95, v, nv, r, done, term, time_dim=time_dim),),
        "vtrace": lambda: F.vtrace_advantage_estimate(0.99, lp, lm, v, nv, r, done, term, time_dim=time_dim),
        "vtrace_thresh": lambda: F.vtrace_advantage_estimate(0.97, lp, lm, v, nv, r, done, None, 0.8, 1.3, time_dim=time_dim),
    }
    fast = {k: fn() for k, fn in calls.items()}
    monkeypatch.setattr(F, "_HOST", None)
    for k, fn in calls.items():
        slow = fn()
        for a, b in zip(fast[k], slow):
            assert a.dtype == b.dtype and a.shape == b.shape, k
            assert torch.equal(a, b), f"{k}: max abs diff {(a - b).abs().max().item()}"


def _coef_forms(shape, dtype, g, l):
    """gamma / lmbda the ways the reference's tests pass them (test_cost.py:14147-14197) and a few more."""
    full = lambda x, dt=dtype: torch.full(shape, x, dtype=dt, device=DEV)
    ramp = lambda x: (x - 0.05 * torch.rand(shape, generator=torch.Generator().manual_seed(3))).to(dtype).to(DEV)
    other = torch.float64 if dtype == torch.float32 else torch.float32
    return {
        "zero_dim_host": (torch.tensor(g, dtype=dtype), torch.tensor(l, dtype=dtype), True),
        "zero_dim_host_other_dtype": (torch.tensor(g, dtype=other), l, True),
        "one_element_device": (torch.as_tensor([g], device=DEV, dtype=dtype), torch.as_tensor([l], device=DEV, dtype=dtype), True),
        "one_element_gamma_float_lmbda": (torch.as_tensor([g], dtype=dtype), l, True),
        "full_gamma_float_lmbda": (ramp(g), l, True),
        "full_gamma_one_element_lmbda": (full(g), torch.tensor(l, dtype=dtype), True),
        "full_both": (ramp(g), ramp(l), True),
        "float_gamma_full_lmbda": (g, ramp(l), None),         # eligible only when the default dtype is the value dtype
        "full_gamma_other_dtype": (full(g, other), l, False),  # coefficient dtype != value dtype: python's general path
    }


@pytest.mark.parametrize("default_dtype", [torch.float32, torch.float64], ids=["def32", "def64"], indirect=True)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("shape,time_dim", [((6, 40, 1), -2), ((9, 20, 3, 1), -3)])
def test_host_module_takes_tensor_coefficients(shape, time_dim, dtype, default_dtype, monkeypatch):
    """gamma / lmbda as 0-dim or 1-element tensors (host or device) and as tensors of the inputs' shape go through the C++
    host module too: same C-ABI call as python's general path -- bit for bit, dtype for dtype -- and the
    module really takes them (it returns outputs, not None)."""
    assert F.host_fast_path(), "lib/_trl_host.so is not built (make -C torchrl_amd/csrc)"
    host = F._HOST
    g = torch.Generator().manual_seed(12)
    mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32).to(dtype).to(DEV)
    r, v, nv = mk(), mk(), mk()
    lp = torch.rand(shape, generator=g).clamp_min(1e-3).log().to(dtype).to(DEV)
    lm = torch.rand(shape, generator=g).clamp_min(1e-3).log().to(dtype).to(DEV)
    term = (torch.rand(shape, generator=g) < 0.05).to(DEV)
    done = term | (torch.rand(shape, generator=g) < 0.05).to(DEV)
    forms = _coef_forms(shape, dtype, 0.97, 0.9)
    fast, taken = {}, {}
    for name, (gam, lam, _) in forms.items():
        taken[name] = (
            host.gae(gam, lam, v, nv, r, done, term, None, F.ALGO_AUTO, time_dim) is not None,
            host.td_lambda(gam, lam, nv, r, done, term, F.ALGO_AUTO, time_dim, True) is not None,
            host.vtrace(gam, lp, lm, v, nv, r, done, term, 1.0, 1.0, F.ALGO_AUTO, time_dim) is not None)
        fast[name] = (F.generalized_advantage_estimate(gam, lam, v, nv, r, done, term, time_dim=time_dim)
                      + F.vec_generalized_advantage_estimate(gam, lam, v, nv, r, done, term, time_dim=time_dim)
                      + F.gae_normalized(gam, lam, v, nv, r, done, term, eps=1e-4, time_dim=time_dim)
                      + (F.td_lambda_return_estimate(gam, lam, nv, r, done, term, time_dim=time_dim),
                         F.vec_td_lambda_return_estimate(gam, lam, nv, r, done, term, time_dim=time_dim))
                      + F.vtrace_advantage_estimate(gam, lp, lm, v, nv, r, done, term, time_dim=time_dim))
    monkeypatch.setattr(F, "_HOST", None)
    for name, (gam, lam, eligible) in forms.items():
        if eligible is None:
            eligible = default_dtype == dtype
        assert taken[name][0] == eligible, f"gae {name}: host module {'took' if taken[name][0] else 'declined'} the call"
        if eligible:  # (a coefficient tensor of another dtype changes vec TD(lambda)'s output dtype: python's path)
            assert taken[name][1] == (name != "zero_dim_host_other_dtype") and taken[name][2], f"{name}: td_lambda / vtrace"
        slow = (F.generalized_advantage_estimate(gam, lam, v, nv, r, done, term, time_dim=time_dim)
                + F.vec_generalized_advantage_estimate(gam, lam, v, nv, r, done, term, time_dim=time_dim)
                + F.gae_normalized(gam, lam, v, nv, r, done, term, eps=1e-4, time_dim=time_dim)
                + (F.td_lambda_return_estimate(gam, lam, nv, r, done, term, time_dim=time_dim),
                   F.vec_td_lambda_return_estimate(gam, lam, nv, r, done, term, time_dim=time_dim))
                + F.vtrace_advantage_estimate(gam, lp, lm, v, nv, r, done, term, time_dim=time_dim))
        for k, (a, b) in enumerate(zip(fast[name], slow)):
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k, a.dtype, b.dtype)
            assert torch.equal(a, b), f"{name} output {k}: max abs diff {(a - b).abs().max().item()}"


def test_host_module_leaves_reference_errors_to_python():
    x = torch.zeros(2, 5, 1, device=DEV)
    m = torch.zeros(2, 5, 1, dtype=torch.bool, device=DEV)
    with pytest.raises(RuntimeError, match="rolling_gamma=False is expected only with time-sensitive"):
        F.td_lambda_return_estimate(0.99, 0.95, x, x, m, m, False)
    with pytest.raises(RuntimeError):
        F.vtrace_advantage_estimate(0.99, x, x, x, x[:1], x, m, m)
    with pytest.raises(RuntimeError):
        F.td_lambda_return_estimate(0.99, 0.95, x, x[:, :3], m, m)


@pytest.mark.parametrize("meta", G.cases("vec"), ids=G.ids("vec"))
def test_vec_variants_against_reference_vec_outputs(meta):
    """The vec_* names run the same exact kernels: against the reference's LOOP outputs at the SCAN bar, against the
    reference's own VEC outputs (filter truncation and all) at the reference's 1e-4, same dtypes (tools/gen_golden_vec.py).
    Where the reference's vec path raises on inputs its loop accepts, only the loop outputs pin the case."""
    a = G.arrays("vec", meta)
    done, term = G.expand_masks(meta, a)
    g, l = cu(G.coef(meta, a, "gamma")), cu(G.coef(meta, a, "lmbda"))
    v, nv, r, done, term = cu(a["state_value"]), cu(a["next_state_value"]), cu(a["reward"]), cu(done), cu(term)
    kw = G.time_kw(meta)
    adv, vt = F.generalized_advantage_estimate(g, l, v, nv, r, done, term, **kw)
    check(adv, a["advantage"], False)
    check(vt, a["value_target"], False)
    adv, vt = F.vec_generalized_advantage_estimate(g, l, v, nv, r, done, term, **kw)   # its dtype follows the reference's VEC variant
    torch.testing.assert_close(adv.cpu().to(a["advantage"].dtype), a["advantage"], **tol(a["advantage"].dtype))
    torch.testing.assert_close(vt.cpu().to(a["advantage"].dtype), a["value_target"], **tol(a["advantage"].dtype))
    if "vec_advantage" in a:
        assert adv.dtype == a["vec_advantage"].dtype
        torch.testing.assert_close(adv.cpu(), a["vec_advantage"], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(vt.cpu(), a["vec_value_target"], rtol=1e-4, atol=1e-4)
    ret = F.td_lambda_return_estimate(g, l, nv, r, done, term, **kw)
    check(ret, a["returns"], False)
    vret = F.vec_td_lambda_return_estimate(g, l, nv, r, done, term, **kw)
    if "vec_returns" in a:
        assert vret.dtype == a["vec_returns"].dtype
        torch.testing.assert_close(vret.cpu(), a["vec_returns"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(vret.cpu().to(a["returns"].dtype), a["returns"], **tol(a["returns"].dtype))
