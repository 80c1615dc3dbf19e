"""`_standardize` (reference: torchrl/_utils.py:875-936; reference test: test_cost.py:16166-16176).
CPU: the numpy oracle against the golden vectors generated from the reference
(tools/gen_golden_standardize.py), and the host-side validation.  GPU (-m gpu): the HIP path
(trl_moments_grouped / trl_normalize_grouped through the C ABI) against the same goldens and the oracle.
Tolerance: torch reduces mean / std in the storage dtype with an internal order, the HIP path and the
oracle in fp64, so fp32 agrees to ~1e-6 relative (stated below), fp64 to 1e-12."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import value_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
MAN = json.load(open(os.path.join(HERE, "golden", "standardize_manifest.json")))
Z = np.load(os.path.join(HERE, "golden", "standardize.npz"))
CASES = MAN["cases"]
IDS = [f"{c['id']}-{'x'.join(map(str, c['shape']))}-{c['dtype']}-ex{c['exclude_dims']}" for c in CASES]


def tol(dtype):
    return dict(rtol=3e-6, atol=3e-6) if dtype == torch.float32 else dict(rtol=1e-12, atol=1e-12)


def arrays(c):
    g = lambda k: torch.from_numpy(Z[f"{c['id']}.{k}"]) if f"{c['id']}.{k}" in Z.files else None
    return g("x"), g("out"), g("mean"), g("std")


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_oracle_matches_reference_goldens(c):
    x, ref, mean, std = arrays(c)
    if mean is not None or std is not None or "mean" in c:
        pytest.skip("caller-provided statistics: no reduction to restate")
    torch.testing.assert_close(O.standardize(x, c["exclude_dims"], c["eps"]), ref, **tol(x.dtype))


def test_reference_test_case_closed_form():
    """test_cost.py:16166-16176 restated on the oracle."""
    t = torch.arange(3 * 4 * 5 * 6, dtype=torch.float32).view(3, 4, 5, 6)
    want = (t - t.mean((0, 2), keepdim=True)) / t.std((0, 2), keepdim=True).clamp(1e-6)
    torch.testing.assert_close(O.standardize(t, (1, 3)), want, **tol(torch.float32))
    torch.testing.assert_close(O.standardize(t, ()), (t - t.mean()) / t.std(), **tol(torch.float32))


def test_validation_matches_reference():
    from torchrl_amd._utils import _standardize

    x = torch.randn(16, 5, 2)
    table = {e["name"]: e for e in MAN["errors"]}
    for name, ex in (("repeated", (1, 1)), ("out_of_bounds", (3,)), ("negative_out_of_bounds", (-4,))):
        with pytest.raises(ValueError) as ei:
            _standardize(x, ex)
        assert str(ei.value) == table[name]["message"]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = _standardize(x, (0, 1, 2))
    assert out is x and str(w[0].message) == table["all_excluded"]["message"]
    # caller-provided statistics never reach the library (test_cost.py:16173)
    t = torch.arange(24, dtype=torch.float32).view(2, 3, 4)
    torch.testing.assert_close(_standardize(t, (), -1, 2), (t + 1) / 2)


def test_grouped_pattern():
    from torchrl_amd._utils import _grouped_pattern

    assert _grouped_pattern([32, 10, 5, 1], [2]) == (320, 5, 1, None)
    assert _grouped_pattern([32, 10, 5, 1], [2, 3]) == (320, 5, 1, None)
    assert _grouped_pattern([32, 10, 5, 1], [0]) == (1, 32, 50, None)
    assert _grouped_pattern([6, 4, 3, 2, 5], [1, 3]) == (1, 8, 90, [1, 3, 0, 2, 4])


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_hip_matches_reference_goldens(c):
    from torchrl_amd._utils import _standardize

    x, ref, mean, std = arrays(c)
    dev = "cuda:0"
    kw = {}
    if mean is not None:
        kw["mean"] = mean.to(dev)
    elif "mean" in c:
        kw["mean"] = c["mean"]
    if std is not None:
        kw["std"] = std.to(dev)
    elif "std" in c:
        kw["std"] = c["std"]
    out = _standardize(x.to(dev), tuple(c["exclude_dims"]), eps=c["eps"], **kw)
    assert out.shape == ref.shape and out.dtype == ref.dtype and out.is_contiguous()
    torch.testing.assert_close(out.cpu(), ref, **tol(ref.dtype))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,ex", [((4096, 100, 5, 1), (-2,)), ((65536, 32, 1), (1,)), ((8, 3000, 7), (0, 2)), ((3, 70000), (0,)),
                                      ((5000, 3), (0,)), ((1, 1, 9), (2,))])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_hip_matches_oracle_larger_shapes(shape, ex, dtype):
    """Multi-agent rollout shapes (agent dim kept independent), many groups (> LDS table), permuted
    (non-adjacent) excluded dims, single-element groups (std = NaN as torch gives)."""
    from torchrl_amd._utils import _standardize

    g = torch.Generator().manual_seed(3)
    x = (torch.randn(shape, generator=g, dtype=torch.float64) * 3 + 1).to(dtype)
    want = O.standardize(x, ex, 1e-6)
    out = _standardize(x.to("cuda:0"), ex, eps=1e-6)
    torch.testing.assert_close(out.cpu(), want, equal_nan=True, **tol(dtype))
    # the device statistics are reproducible run to run (fixed reduction order)
    assert torch.equal(out, _standardize(x.to("cuda:0"), ex, eps=1e-6)) or torch.isnan(out).any()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 1000, 65536, 65537, 300000], ids=lambda n: f"n{n}")
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_hip_ungrouped_small_and_large(n, dtype):
    """No excluded dims: one single-block launch up to 65536 elements (a PPO minibatch), moments -> reduce ->
    scale pass above; n = 1 gives NaN as torch's unbiased std does."""
    from torchrl_amd._utils import _standardize
    from torchrl_amd.objectives.value import functional as F

    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n, generator=g, dtype=torch.float64) * 2 - 0.3).to(dtype)
    out = _standardize(x.to("cuda:0"), eps=1e-6)
    assert F.last_kernel() == ("standardize_small" if n <= 65536 else "standardize_persistent")
    torch.testing.assert_close(out.cpu(), O.standardize(x, (), 1e-6), equal_nan=True, **tol(dtype))


# ---- the one-launch route (persistent grid, values kept in registers across an in-launch gather of the block moments) --------
def _big(shape, dtype, seed=5):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32).to(torch.float64) * 3 + 1).to(dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,ex,kernel", [
    ((4096, 100, 5, 1), (-2,), "standardize_grouped_persistent"),   # BASELINE config 3: MAPPO, agent dim kept independent
    ((1000, 7, 3), (1,), "standardize_grouped_persistent"),         # inner > 1: period 21, 1008 of 1024 threads per tile
    ((300000, 16), (1,), "standardize_grouped_persistent"),         # 16 groups: every wave of a block gathers one
    ((64, 2, 512), (1,), "standardize_grouped_persistent"),         # period exactly 1024
    ((1800000, 4), (1,), "standardize_grouped_persistent"),         # 28 values per thread
    ((2200000, 4), (1,), "normalize_grouped"),                      # more than a thread keeps: moments -> reduce -> scale pass
    ((50, 17, 2), (1,), "normalize_grouped"),                       # more than 16 groups
    ((3, 2, 513), (1,), "normalize_grouped"),                       # period above 1024
], ids=lambda v: "x".join(map(str, v)) if isinstance(v, tuple) else None)
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_hip_grouped_one_launch(shape, ex, kernel, dtype):
    from torchrl_amd._utils import _standardize
    from torchrl_amd.objectives.value import functional as F

    x = _big(shape, dtype)
    xd = x.to("cuda:0")
    out = _standardize(xd, ex, eps=1e-6)
    assert F.last_kernel() == kernel
    torch.testing.assert_close(out.cpu(), O.standardize(x, ex, 1e-6), **tol(dtype))
    for _ in range(3):  # fixed reduction order: bitwise reproducible; the epoch protocol holds call after call
        assert torch.equal(out, _standardize(xd, ex, eps=1e-6))


@pytest.mark.gpu
@pytest.mark.parametrize("n,offset,kernel", [(300000, 0, "standardize_persistent"), (3000000, 0, "standardize_persistent"),
                                             (12000000, 0, "standardize_persistent"), (20000000, 0, "normalize"),
                                             (2000001, 1, "standardize_persistent"), (7000003, 3, "standardize_persistent"),
                                             (9000001, 1, "normalize")], ids=lambda v: str(v))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_hip_ungrouped_one_launch(n, offset, kernel, dtype):
    """2 / 8 / 14 sixteen-byte vectors per thread, unaligned inputs (4- / 8-byte accesses, 8 or 28 values per thread), and the
    sizes beyond what a thread keeps (two launches)."""
    from torchrl_amd._utils import _standardize
    from torchrl_amd.objectives.value import functional as F

    if dtype == torch.float64 and n == 12000000:
        kernel = "normalize"  # 14 vectors of 2: 7.3 M elements
    x = _big(n + offset, dtype)[offset:]
    xd = _big(n + offset, dtype).to("cuda:0")[offset:]
    out = _standardize(xd, eps=1e-6)
    assert F.last_kernel() == kernel
    torch.testing.assert_close(out.cpu(), O.standardize(x, (), 1e-6), **tol(dtype))
    assert torch.equal(out, _standardize(xd, eps=1e-6))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_c_abi_standardize_in_place(dtype):
    """include/trl_value.h: `out may alias x` for trl_standardize / trl_standardize_grouped -- on the one-launch routes (every
    load precedes the gather), the small kernel and the two-launch route alike: bit-equal to the out-of-place call."""
    from torchrl_amd import _extension as E
    from torchrl_amd import _utils as U
    from torchrl_amd.objectives.value import functional as F

    lib = E.get_lib()
    dev = torch.device("cuda:0")
    code = 0 if dtype == torch.float32 else 1
    for n, kernel in ((1000, "standardize_small"), (700001, "standardize_persistent"), (30000000, "normalize")):
        x = _big((n,), dtype).to(dev)
        ws = F._workspace(lib, dev, n, 1, 1)
        out = torch.empty_like(x)
        E.check(lib.trl_standardize(F._stream(dev), code, x.data_ptr(), n, 1e-6, out.data_ptr(), None, ws.data_ptr(), ws.numel()), lib)
        assert F.last_kernel() == kernel
        y = x.clone()
        E.check(lib.trl_standardize(F._stream(dev), code, y.data_ptr(), n, 1e-6, y.data_ptr(), None, ws.data_ptr(), ws.numel()), lib)
        assert torch.equal(y, out), (n, kernel)
    for (outer, G, inner), kernel in (((40960, 5, 1), "standardize_grouped_persistent"), ((3000, 7, 11), "standardize_grouped_persistent"),
                                      ((2000, 40, 3), "normalize_grouped")):
        x = _big((outer, G, inner), dtype).to(dev)
        st, wsp, wsn = U._scratch(lib, dev, F._stream(dev), outer, G, inner)
        out = torch.empty_like(x)
        args = lambda src, dst: (F._stream(dev), code, src.data_ptr(), outer, G, inner, 1e-6, dst.data_ptr(), st.data_ptr(), wsp, wsn)
        E.check(lib.trl_standardize_grouped(*args(x, out)), lib)
        assert F.last_kernel() == kernel
        y = x.clone()
        E.check(lib.trl_standardize_grouped(*args(y, y)), lib)
        assert torch.equal(y, out), (outer, G, inner)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_one_launch_routes_share_the_workspace_with_the_fused_gae_step():
    """`_standardize(x)` of a large tensor and the single-launch GAE + normalise use the same {epoch, slots} header of the
    stream's workspace: 300 interleaved calls (more than the 8-bit tag's period) stay bit-identical to the first."""
    from torchrl_amd._utils import _standardize
    from torchrl_amd.objectives.value import functional as F

    g = torch.Generator().manual_seed(11)
    shape = (65536, 32, 1)
    r, v, nv = (torch.randn(shape, generator=g).to("cuda:0") for _ in range(3))
    term = (torch.rand(shape, generator=g) < 0.01).to("cuda:0")
    done = term | (torch.rand(shape, generator=g) < 0.01).to("cuda:0")
    x = _big((700001,), torch.float32).to("cuda:0")
    y = _big((4096, 100, 5, 1), torch.float32, seed=6).to("cuda:0")
    adv0, vt0 = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)
    assert F.last_kernel() == "gae_fused"
    x0, y0 = _standardize(x), _standardize(y, (-2,))
    for k in range(300):
        for _ in range(k % 3):
            xs = _standardize(x)
        ys = _standardize(y, (-2,))
        adv, vt = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)
        if k % 25 == 0 or k > 250:
            assert torch.equal(adv, adv0) and torch.equal(vt, vt0) and torch.equal(ys, y0)
            assert torch.equal(_standardize(x), x0)
    torch.cuda.synchronize()
    assert not F.workspace_error() and not F.fused_degraded()


@pytest.mark.gpu
def test_small_fused_grids_keep_every_slot_fresh():
    """Small batches run the single-launch GAE + normalise on a smaller persistent grid (one chunk per team); the slots of
    the absent blocks are refreshed by every launch, so a full-grid launch exactly 256 launches (the 8-bit tag's period)
    after the previous full-grid one cannot meet a stale granule whose tag matches: 255 small launches between two big ones,
    twice, with a mid-sized grid mixed in; everything bit-identical to the first results."""
    from torchrl_amd.objectives.value import functional as F

    def batch(envs, seed):
        g = torch.Generator().manual_seed(seed)
        shape = (envs, 32, 1)
        r, v, nv = (torch.randn(shape, generator=g).to("cuda:0") for _ in range(3))
        term = (torch.rand(shape, generator=g) < 0.01).to("cuda:0")
        done = term | (torch.rand(shape, generator=g) < 0.01).to("cuda:0")
        return lambda: F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)

    big, mid, small = batch(65536, 1), batch(8192, 2), batch(256, 3)   # grids of 256, 64 and 32 blocks on 256 CUs
    want = {}
    for name, fn in (("big", big), ("mid", mid), ("small", small)):
        want[name] = fn()
        assert F.last_kernel() == "gae_fused"
    for _ in range(2):
        for k in range(255):
            out = small() if k % 5 else mid()
        assert torch.equal(out[0], want["small"][0])
        got = big()
        assert torch.equal(got[0], want["big"][0]) and torch.equal(got[1], want["big"][1])
    for name, fn in (("mid", mid), ("small", small), ("big", big)):
        got = fn()
        assert torch.equal(got[0], want[name][0]) and torch.equal(got[1], want[name][1])
    torch.cuda.synchronize()
    assert not F.workspace_error() and not F.fused_degraded()
