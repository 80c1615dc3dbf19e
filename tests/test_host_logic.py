"""CPU: host-side logic of the drop-in (shape / time-dim errors, canonical (O,T,I) view,
coefficient rounding) and the C-ABI library's exported symbols.  No compute calls here."""
import ctypes
import os
import re

import pytest
import torch

import golden_util as G
from torchrl_amd import _extension
from torchrl_amd.objectives.value import functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "trl_value.h")).read()
    declared = set(re.findall(r"\b(trl_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_extension._PROTOTYPES), (declared ^ set(_extension._PROTOTYPES))
    lib = ctypes.CDLL(_extension.lib_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert _extension.get_lib().trl_abi_version() == _extension.ABI_VERSION


def test_cpu_tensors_take_the_host_entry_points():
    """CPU tensors run through the library's own host path (trl_*_cpu; parity: tests/test_cpu_path.py); other devices raise."""
    r = torch.randn(3, 8, 1)
    d = torch.zeros(3, 8, 1, dtype=torch.bool)
    adv, vt = F.generalized_advantage_estimate(0.9, 0.9, r, r, r, d)
    assert adv.device.type == "cpu" and adv.shape == r.shape
    with pytest.raises(RuntimeError, match="HIP .* and CPU tensors"):
        F.generalized_advantage_estimate(0.9, 0.9, *(t.to("meta") for t in (r, r, r, d)))


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_extension, "_lib", None)
    monkeypatch.setattr(_extension, "_LIB_PATH", "/nonexistent/libtrl_value.so")
    with pytest.raises(_extension.ExtensionMissingError):
        _extension.get_lib()


def test_error_table_matches_reference():
    """Exception types / messages recorded from the reference (tests/golden/manifest.json: errors)."""
    table = {e["name"]: e for e in G.MANIFEST["errors"]}
    r = torch.randn(3, 8, 1)
    d = torch.zeros(3, 8, 1, dtype=torch.bool)
    g = torch.full_like(r, 0.9)
    tm = d.clone()
    tm[:, 3] = True
    calls = {
        "gae_shape_mismatch": lambda: F.generalized_advantage_estimate(0.9, 0.9, r, r, r[:, :4], d),
        "gae_positional_time_dim": lambda: F.generalized_advantage_estimate(0.9, 0.9, r, r, r, d, d, -2),
        "vec_gae_shape_mismatch": lambda: F.vec_generalized_advantage_estimate(0.9, 0.9, r, r, r[:, :4], d),
        "gae_1d_time_dim_m3": lambda: F.generalized_advantage_estimate(0.9, 0.9, r[0, :, 0], r[0, :, 0], r[0, :, 0], d[0, :, 0], time_dim=-3),
        "gae_bad_time_dim": lambda: F.generalized_advantage_estimate(0.9, 0.9, r, r, r, d, time_dim=-4),
        "tdl_shape_mismatch": lambda: F.td_lambda_return_estimate(0.9, 0.9, r, r[:, :4], d),
        "tdl_positional_time_dim": lambda: F.td_lambda_return_estimate(0.9, 0.9, r, r, d, d, None, -2),
        "tdl_not_rolling_scalar": lambda: F.td_lambda_return_estimate(0.9, 0.9, r, r, d, rolling_gamma=False),
        "vec_tdl_not_rolling_after_term": lambda: F.vec_td_lambda_return_estimate(g, 0.9, r, r, tm, tm, rolling_gamma=False),
        "vec_tdl_not_rolling": lambda: F.vec_td_lambda_return_estimate(g, 0.9, r, r, d, d, rolling_gamma=False),
        "vtrace_shape_mismatch": lambda: F.vtrace_advantage_estimate(0.9, r, r, r, r, r[:, :4], d),
        "tdl_adv_shape_mismatch": lambda: F.td_lambda_advantage_estimate(0.9, 0.9, r[:, :4], r, r, d),
    }
    for name, fn in calls.items():
        exp = table[name]
        with pytest.raises(Exception) as ei:
            fn()
        assert type(ei.value).__name__ == exp["type"], (name, ei.value)
        if exp["type"] != "TypeError":
            assert str(ei.value) == exp["message"], name
    # the reference fails on 1-D input with the default time_dim (ValueError from tuple unpacking); we raise too
    with pytest.raises((RuntimeError, ValueError)):
        F.generalized_advantage_estimate(0.9, 0.9, r[0, :, 0], r[0, :, 0], r[0, :, 0], d[0, :, 0])


@pytest.mark.parametrize("shape,time_dim,exp", [((3, 8, 1), -2, (3, 8, 1)), ((4, 10, 5, 1), -3, (4, 10, 5)),
                                                ((4, 10, 5, 1), 1, (4, 10, 5)), ((6, 9), -1, (6, 9, 1)),
                                                ((17,), -1, (1, 17, 1)), ((2, 3, 4, 5), -2, (6, 4, 5)),
                                                ((2, 3, 4, 5), 0, (1, 2, 60))])
def test_canonical_view(shape, time_dim, exp):
    assert F._canonical_dims(shape, time_dim)[:3] == exp


def test_gae_coefficient_rounding_follows_reference():
    """python-float gamma is rounded to the default dtype even for float64 values."""
    r = torch.randn(3, 8, 1, dtype=torch.float64)
    d = torch.zeros(3, 8, 1, dtype=torch.bool)
    c = F._prepare_gae(0.99, 0.95, r, r, r, d, None, -2)
    assert c.vdt == torch.float64 and c.cdt == torch.float32
    assert c.gamma == float(torch.tensor(0.99, dtype=torch.float32)) and c.lg == float(torch.tensor(0.99 * 0.95, dtype=torch.float32))
    c = F._prepare_gae(torch.tensor(0.99, dtype=torch.float64), 0.95, r, r, r, d, None, -2)
    assert c.cdt == torch.float64 and c.gamma == 0.99 and c.lg == 0.99 * 0.95
    torch.set_default_dtype(torch.float64)
    c = F._prepare_gae(0.99, 0.95, r, r, r, d, None, -2)
    assert c.cdt == torch.float64 and c.gamma == 0.99
    torch.set_default_dtype(torch.float32)
    # two 0-dim f32 buffers (the GAE module case) multiply in f32
    g32, l32 = torch.tensor(0.97), torch.tensor(0.93)
    c = F._prepare_gae(g32, l32, r, r, r, d, None, -2)
    assert c.lg == float(g32 * l32)
    # full tensor gamma, python lmbda: lmbda*gamma is formed in-kernel in gamma's dtype
    gt = torch.rand(3, 8, 1)
    c = F._prepare_gae(gt, 0.95, r, r, r, d, None, -2)
    assert c.gamma_t is not None and c.lg_t is None and c.cdt == torch.float32 and c.lmbda == 0.95
    # tensor lmbda: the product is materialised with torch's own rounding
    c = F._prepare_gae(gt, torch.rand(3, 8, 1), r, r, r, d, None, -2)
    assert c.lg_t is not None and c.gamma_t is not None


def test_td_lambda_coefficients_keep_value_dtype():
    """torch.full_like(next_state_value, gamma) keeps the value dtype."""
    r = torch.randn(3, 8, 1, dtype=torch.float64)
    d = torch.zeros(3, 8, 1, dtype=torch.bool)
    c = F._prepare_td_lambda(0.99, 0.95, r, r, d, None, None, -2)
    assert c.cdt == torch.float64 and c.gamma == 0.99 and c.lmbda == 0.95 and c.rolling
    c = F._prepare_td_lambda(0.99, 0.95, r.float(), r.float(), d, None, None, -2)
    assert c.gamma == float(torch.tensor(0.99, dtype=torch.float32))
    c = F._prepare_td_lambda(torch.tensor(0.99), 0.95, r, r, d, None, None, -2)  # 0-dim f32 buffer: rounding comes from the buffer
    assert c.gamma == float(torch.tensor(0.99, dtype=torch.float32))


def test_expanded_masks_are_handed_over_compact():
    """Stride-0 masks over the agent dim (DoneTransform) are read in place by the kernels (mask_bcast = 1):
    no materialisation; anything else -- or an input that needs gradients -- takes the full layout."""
    r = torch.randn(4, 10, 5, 1)
    base = torch.rand(4, 10, 1, 1) < 0.2
    d = base.expand(4, 10, 5, 1)
    c = F._prepare_gae(0.9, 0.9, r, r, r, d, d, -3)
    assert (c.O, c.T, c.I, c.mask_bcast) == (4, 10, 5, 1)
    assert c.done.is_contiguous() and c.done.dtype == torch.uint8 and c.done.shape == (4, 10)
    assert c.done.data_ptr() == base.data_ptr()  # a view of the caller's mask, not a copy
    t = F._prepare_td_lambda(0.9, 0.9, r, r, d, d, None, -3)
    v = F._prepare_vtrace(0.9, r, r, r, r, r, d, d, 1.0, 1.0, -3)
    assert t.mask_bcast == 1 and v.mask_bcast == 1 and v.done.shape == (4, 10)
    # one full mask, or a dense mask: materialised in the canonical layout
    full = torch.rand(4, 10, 5, 1) < 0.2
    for dd, tt in ((d, full), (full, full)):
        c = F._prepare_gae(0.9, 0.9, r, r, r, dd, tt, -3)
        assert c.mask_bcast == 0 and c.done.shape == r.shape and c.done.is_contiguous() and c.done.dtype == torch.uint8
    rg = r.clone().requires_grad_(True)  # differentiable calls keep the compact masks: the backward kernels read them in place
    assert F._prepare_gae(0.9, 0.9, rg, r, r, d, d, -3).mask_bcast == 1
    # I == 1: nothing to broadcast over
    assert F._prepare_gae(0.9, 0.9, r[:, :, :1], r[:, :, :1], r[:, :, :1], base, base, -3).mask_bcast == 0


def test_vtrace_thresholds_round_like_torch_tensor():
    r = torch.randn(3, 8, 1, dtype=torch.float64)
    d = torch.zeros(3, 8, 1, dtype=torch.bool)
    c = F._prepare_vtrace(0.99, r, r, r, r, r, d, None, 0.7, 0.3, -2)
    assert c.rho_thresh == float(torch.tensor(0.7)) and c.c_thresh == float(torch.tensor(0.3))
    assert c.terminated.data_ptr() == c.done.data_ptr() or torch.equal(c.terminated, c.done)


def test_library_binds_to_torchs_hip_runtime_whatever_the_import_order():
    """Importing the package before torch must not bring a second HIP runtime into the process
    (the library's RUNPATH names the system copy; torch ships its own under the same SONAME)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torchrl_amd, torch\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "print(len(libs), libs)\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split()[0] == "1", out.stdout


def test_shifted_pair_detection():
    """state_value / next_state_value as the two T-step windows of one [*B, T + 1, 1] buffer (advantages.py:440-490,
    `shifted=True`) are recognised from strides and storage offsets alone; anything else takes the dense path."""
    sp = lambda v, nv, td=-2, grads=False: F._shifted_pair(v, nv, tuple(v.shape), td, v.dtype, grads)
    buf = torch.randn(6, 11, 1)
    assert sp(buf[:, :-1], buf[:, 1:])
    assert not sp(buf[:, 1:], buf[:, :-1])
    assert not sp(buf[:, :-1], buf[:, 1:].clone())
    assert not sp(buf[:, :-1], buf[:, 1:], grads=True)
    assert not sp(buf[:, :9], buf[:, 1:10])                       # rows T + 2 apart
    assert not sp(buf[:, :-1], buf[:, 1:].double())
    flat = torch.randn(6, 11)
    assert sp(flat[:, :-1], flat[:, 1:], td=-1)                   # no feature dim, time last
    four = torch.randn(2, 3, 11, 1, 1)
    assert sp(four[:, :, :-1], four[:, :, 1:], td=-3) and sp(four[:, :, :-1], four[:, :, 1:], td=2)
    part = torch.randn(9, 11, 1)
    assert sp(part[2:8, :-1], part[2:8, 1:])                      # a batch slice of a larger buffer
    assert not sp(part[::2, :-1], part[::2, 1:])                  # rows not T + 1 apart
    multi = torch.randn(6, 11, 3)
    assert not sp(multi[:, :-1], multi[:, 1:])                    # I = 3: the kernels' shifted path needs I == 1
    one = torch.randn(1, 11, 1)
    assert sp(one[:, :-1], one[:, 1:])                            # a single row: both windows are contiguous views
    tm = torch.randn(11, 6, 1)                                    # time-major storage seen batch-first
    assert not sp(tm[:-1].transpose(0, 1), tm[1:].transpose(0, 1))
    assert not sp(torch.empty(0, 11, 1)[:, :-1], torch.empty(0, 11, 1)[:, 1:])


def test_host_module_declines_what_the_fast_path_does_not_cover():
    """csrc/host_fast.cpp: loads beside the library, returns None (-> python's general path, which raises the reference's
    errors) for CPU tensors, tensor coefficients, mismatched shapes; no compute call is made without a GPU."""
    from torchrl_amd import _extension
    from torchrl_amd.objectives.value import functional as F
    if _extension.get_host() is None:
        pytest.skip("lib/_trl_host.so not built")
    assert F.host_fast_path()
    x = torch.zeros(2, 5, 1)
    m = torch.zeros(2, 5, 1, dtype=torch.bool)
    assert F._HOST.gae(0.99, 0.95, x, x, x, m, m, None, 0, -2) is None            # not on a HIP device
    assert F._HOST.gae(torch.tensor(0.99), 0.95, x, x, x, m, m, None, 0, -2) is None
    assert F._HOST.gae(0.99, 0.95, x, x[:1], x, m, m, None, 0, -2) is None
    assert F._HOST.gae(0.99, 0.95, x, x, x, m.float(), m, None, 0, -2) is None
    with pytest.raises(TypeError):
        F._HOST.gae(0.99, 0.95, x)
