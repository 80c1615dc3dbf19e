"""GPU (-m gpu): trl_gae_normalize_fwd_noalloc -- the entry point that allocates nothing (static device scratch) --
through the C ABI against trl_gae_normalize_fwd on a caller workspace: bit-identical outputs and moments, on two calls
in a row, the second with a smaller input (the scratch keeps the first call's contents); sizes beyond the static
scratch are rejected with a clear error.  Collected last (file name) so that a failure here hides nothing."""
import ctypes

import pytest
import torch

from torchrl_amd import _extension
from torchrl_amd.objectives.value import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(shape, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    r, v, nv = (torch.randn(shape, generator=g, dtype=torch.float32).to(dtype).to(DEV) for _ in range(3))
    term = (torch.rand(shape, generator=g) < 0.02).to(DEV)
    done = term | (torch.rand(shape, generator=g) < 0.02).to(DEV)
    return r, v, nv, done, term


def _both(lib, shape, dtype, seed, algo=F.ALGO_AUTO):
    r, v, nv, done, term = _inputs(shape, dtype, seed)
    c = F._prepare_gae(0.99, 0.95, v, nv, r, done, term, -2)
    dev = r.device
    stream = F._stream(dev)
    outs = []
    for noalloc in (False, True):
        adv = torch.full(shape, float("nan"), dtype=dtype, device=dev)
        vt = torch.full(shape, float("nan"), dtype=dtype, device=dev)
        stats = torch.zeros(3, dtype=torch.float64, device=dev)
        head = (stream, F._DT[c.vdt], F._DT[c.cdt], c.reward.data_ptr(), c.value.data_ptr(), c.next_value.data_ptr(),
                c.done.data_ptr(), c.terminated.data_ptr(), c.mask_bcast, None, None, c.gamma, c.lmbda, c.lg,
                c.O, c.T, c.I, adv.data_ptr(), vt.data_ptr(), 1e-4, stats.data_ptr())
        if noalloc:
            rc = lib.trl_gae_normalize_fwd_noalloc(*head, algo)
        else:
            ws = torch.zeros(lib.trl_workspace_bytes(c.O, c.T, c.I), dtype=torch.uint8, device=dev)
            rc = lib.trl_gae_normalize_fwd(*head, ws.data_ptr(), ws.numel(), algo)
        assert rc == 0, lib.trl_last_error().decode()
        kernel = lib.trl_last_kernel().decode()
        torch.cuda.synchronize()
        outs.append((adv, vt, stats, kernel))
    (a0, v0, s0, k0), (a1, v1, s1, k1) = outs
    assert k0 == k1
    assert not torch.isnan(a1).any() and not torch.isnan(v1).any()
    assert torch.equal(a0, a1) and torch.equal(v0, v1) and torch.equal(s0, s1), (k0, (a0 - a1).abs().max().item())
    return k1


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_noalloc_matches_workspace_entry_point_twice(dtype):
    lib = _extension.get_lib()
    assert lib.trl_static_scratch_bytes() == 2 << 20
    # two calls in a row on the shared static scratch, the second with a smaller input
    assert _both(lib, (65536, 32, 1), dtype, 1) == "gae_fused"
    assert _both(lib, (1000, 32, 1), dtype, 2) == "gae_fused"
    # shapes that take the two-launch route (scan + partials -> scale pass), then a fused one again
    assert _both(lib, (37, 1500, 1), dtype, 3) == "normalize"
    assert _both(lib, (40, 50, 5), dtype, 4) == "normalize"
    assert _both(lib, (4096, 32, 1), dtype, 5) == "gae_fused"


def test_noalloc_rejects_sizes_beyond_the_static_scratch():
    lib = _extension.get_lib()
    O, T, I = 1 << 27, 4, 1  # SEQ-family partials alone exceed 2 MiB
    assert lib.trl_workspace_bytes(O, T, I) > lib.trl_static_scratch_bytes()
    x = torch.zeros(8, device=DEV)
    rc = lib.trl_gae_normalize_fwd_noalloc(F._stream(x.device), 0, 0, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                           x.data_ptr(), 0, None, None, 0.99, 0.95, 0.94, O, T, I, x.data_ptr(), x.data_ptr(),
                                           1e-4, None, F.ALGO_AUTO)
    assert rc == 3  # TRL_ERR_WORKSPACE
    assert "static scratch" in lib.trl_last_error().decode()


def _bound_noalloc_call(lib, shape, dtype, seed):
    r, v, nv, done, term = _inputs(shape, dtype, seed)
    c = F._prepare_gae(0.99, 0.95, v, nv, r, done, term, -2)
    adv, vt = torch.empty(shape, dtype=dtype, device=DEV), torch.empty(shape, dtype=dtype, device=DEV)
    stats = torch.zeros(3, dtype=torch.float64, device=DEV)
    keep = (c, adv, vt, stats)

    def call(stream=None):
        return lib.trl_gae_normalize_fwd_noalloc(
            F._stream(r.device) if stream is None else stream, F._DT[c.vdt], F._DT[c.cdt], c.reward.data_ptr(), c.value.data_ptr(),
            c.next_value.data_ptr(), c.done.data_ptr(), c.terminated.data_ptr(), c.mask_bcast, None, None, c.gamma, c.lmbda, c.lg,
            c.O, c.T, c.I, adv.data_ptr(), vt.data_ptr(), 1e-4, stats.data_ptr(), F.ALGO_AUTO)

    return call, keep


def test_noalloc_allocates_nothing():
    """The property the entry point exists for: after ONE warm call per device (hipGetSymbolAddress, the occupancy
    query: outside graph capture, include/trl_value.h), calls on the fused, the two-launch and the look-back routes leave the
    device's free memory exactly where it was."""
    lib = _extension.get_lib()
    calls = [_bound_noalloc_call(lib, shape, torch.float32, 7 + k)
             for k, shape in enumerate([(65536, 32, 1), (37, 1500, 1), (40, 50, 5), (2, 40000, 1)])]
    for call, _ in calls:  # warm: first use per device, first launch of every kernel involved
        assert call() == 0, lib.trl_last_error().decode()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    kernels = set()
    for _ in range(20):
        for call, _ in calls:
            assert call() == 0
            kernels.add(lib.trl_last_kernel().decode())
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, f"device free memory moved by {free0 - free1} bytes across the no-alloc calls"
    assert {"gae_fused", "normalize"} <= kernels


def test_noalloc_call_is_graph_capturable():
    """One of the callers the entry point is for: HIP graph capture (no allocation is allowed while capturing).  A captured
    fused step and a captured two-launch step replay bit-equal to the eager calls."""
    lib = _extension.get_lib()
    for shape, seed in (((65536, 32, 1), 11), ((37, 1500, 1), 12)):
        call, (c, adv, vt, stats) = _bound_noalloc_call(lib, shape, torch.float32, seed)
        assert call() == 0  # warm, outside capture
        torch.cuda.synchronize()
        want = (adv.clone(), vt.clone(), stats.clone())
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            graph.capture_begin()
            rc = call(side.cuda_stream)
            graph.capture_end()
        assert rc == 0, lib.trl_last_error().decode()
        for _ in range(3):
            adv.fill_(float("nan"))
            vt.fill_(float("nan"))
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(adv, want[0]) and torch.equal(vt, want[1]) and torch.equal(stats, want[2])
