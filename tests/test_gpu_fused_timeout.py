"""GPU (-m gpu): the single-launch GAE + normalise kernel must never fail silently, and the process must recover by itself
(ADVICE r1 / r2).  Its persistent grid needs every compute unit; here half of them are held by a test kernel
(tests/native/hold_cus.hip: one block per CU that takes the CU's whole LDS and sleeps), so the resident half of the grid
waits its bounded 1 s, gives up, writes the UN-NORMALISED advantage and raises the host-visible flag.  Then:
  * the NEXT library call -- on DIFFERENT inputs -- notices the flag without any synchronisation, emits one RuntimeWarning,
    marks the GPU degraded and runs the step through the two-launch route: its result is right;
  * later calls stay on the two-launch route without warnings; an explicit TRL_ALGO_FUSED request that notices a give-up
    raises;
  * after fused_reset() the single launch runs again and -- on different inputs once more -- uses none of the moments the
    failed launch left behind in the workspace (stale block moments);
  * the entry point that allocates nothing has no host flag: its caller learns of the give-up from trl_workspace_error(NULL)."""
import ctypes
import os
import time
import warnings

import pytest
import torch

from torchrl_amd import _extension
from torchrl_amd.objectives.value import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libhold_cus.so")
SHAPE = (65536, 32, 1)


def _inputs(seed):
    g = torch.Generator().manual_seed(seed)
    r, v, nv = (torch.randn(SHAPE, generator=g).to(DEV) for _ in range(3))
    r = r * (1.0 + seed) + 0.25 * seed   # different moments per seed: stale statistics would show
    term = (torch.rand(SHAPE, generator=g) < 0.01).to(DEV)
    done = term | (torch.rand(SHAPE, generator=g) < 0.01).to(DEV)
    return r, v, nv, done, term


def _holder():
    if not os.path.exists(_SO):
        pytest.skip("tests/native/libhold_cus.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    torch.cuda.init()
    hold = ctypes.CDLL(_SO)
    hold.hold_cus.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lds = hold.hold_max_lds()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if lds < 150 * 1024:
        pytest.skip(f"a block can take only {lds} bytes of LDS: it cannot hold a CU on its own")
    side = torch.cuda.Stream()

    def hold_half(seconds=1.6):
        assert hold.hold_cus(ctypes.c_void_p(side.cuda_stream), cus // 2, lds, seconds) == 0
        time.sleep(0.05)  # the holders are resident before the fused grid is dispatched

    return hold_half


def _clean_workspaces():
    for ws in F._WORKSPACES.values():
        ws[4:8].zero_()
    torch.cuda.synchronize()


def test_fused_gives_up_loudly_and_the_process_degrades_by_itself():
    hold_half = _holder()
    F.fused_reset()
    r, v, nv, done, term = _inputs(0)
    want_adv, want_vt = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)
    assert F.last_kernel() == "gae_fused"
    raw_adv, _ = F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term)
    torch.cuda.synchronize()
    assert not F.workspace_error() and not F.fused_degraded()

    hold_half()
    adv, vt = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)
    assert F.last_kernel() == "gae_fused"
    torch.cuda.synchronize()
    # never NaN: every element is either normalised (blocks that started after the holders left) or left un-normalised
    assert torch.isfinite(adv).all()
    assert torch.equal(vt, want_vt)
    given_up = adv == raw_adv
    assert bool((given_up | (adv == want_adv)).all())
    assert given_up.float().mean() > 0.2, "the resident half of the grid must have given up"
    assert F.workspace_error()

    # the next call -- other data -- warns once, completes through the two-launch route and is right
    r1, v1, nv1, done1, term1 = _inputs(1)
    with pytest.warns(RuntimeWarning, match="gave up waiting"):
        adv1, vt1 = F.gae_normalized(0.99, 0.95, v1, nv1, r1, done1, term1, eps=1e-4)
    assert F.last_kernel() == "normalize"
    assert F.fused_degraded()
    a_scan, vt_scan = F.gae_normalized(0.99, 0.95, v1, nv1, r1, done1, term1, eps=1e-4, algo=F.ALGO_SCAN)
    torch.cuda.synchronize()
    assert torch.equal(adv1, a_scan) and torch.equal(vt1, vt_scan)
    # ... against torch's own statistics of the un-normalised advantage
    raw1, _ = F.generalized_advantage_estimate(0.99, 0.95, v1, nv1, r1, done1, term1)
    torch.testing.assert_close(adv1, (raw1 - raw1.mean()) / raw1.std().clamp_min(1e-4), rtol=1e-5, atol=1e-5)
    # later calls: no warning, still two launches
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        F.gae_normalized(0.99, 0.95, v1, nv1, r1, done1, term1, eps=1e-4)
        assert F.last_kernel() == "normalize"
        F.td_lambda_return_estimate(0.99, 0.95, nv1, r1, done1, term1)

    # re-armed: the single launch again, on data the failed launch never saw -- none of its block moments may be reused
    F.fused_reset()
    assert not F.fused_degraded()
    r2, v2, nv2, done2, term2 = _inputs(2)
    adv2, vt2 = F.gae_normalized(0.99, 0.95, v2, nv2, r2, done2, term2, eps=1e-4)
    assert F.last_kernel() == "gae_fused"
    a2_scan, vt2_scan = F.gae_normalized(0.99, 0.95, v2, nv2, r2, done2, term2, eps=1e-4, algo=F.ALGO_SCAN)
    torch.cuda.synchronize()
    assert torch.equal(vt2, vt2_scan)
    torch.testing.assert_close(adv2, a2_scan, rtol=1e-6, atol=1e-6)
    _clean_workspaces()
    assert not F.workspace_error()


def test_explicit_fused_request_raises_after_a_give_up():
    hold_half = _holder()
    F.fused_reset()
    r, v, nv, done, term = _inputs(3)
    hold_half()
    F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4, algo=F.ALGO_FUSED)
    torch.cuda.synchronize()
    with pytest.raises(_extension.TrlError, match="gave up waiting"):
        F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4, algo=F.ALGO_FUSED)
    assert F.fused_degraded()
    F.fused_reset()
    want = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4, algo=F.ALGO_SCAN)
    got = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4, algo=F.ALGO_FUSED)
    torch.cuda.synchronize()
    assert torch.equal(got[1], want[1])
    torch.testing.assert_close(got[0], want[0], rtol=1e-6, atol=1e-6)
    _clean_workspaces()


def test_noalloc_give_up_is_reported_through_the_static_scratch():
    hold_half = _holder()
    lib = _extension.get_lib()
    F.fused_reset()
    r, v, nv, done, term = _inputs(4)
    c = F._prepare_gae(0.99, 0.95, v, nv, r, done, term, -2)
    adv, vt = torch.empty_like(r), torch.empty_like(r)
    stats = torch.zeros(3, dtype=torch.float64, device=DEV)

    def call():
        rc = lib.trl_gae_normalize_fwd_noalloc(
            F._stream(r.device), F._DT[c.vdt], F._DT[c.cdt], c.reward.data_ptr(), c.value.data_ptr(), c.next_value.data_ptr(),
            c.done.data_ptr(), c.terminated.data_ptr(), 0, None, None, c.gamma, c.lmbda, c.lg, c.O, c.T, c.I, adv.data_ptr(),
            vt.data_ptr(), 1e-4, stats.data_ptr(), F.ALGO_AUTO)
        torch.cuda.synchronize()
        return rc

    assert call() == 0
    assert lib.trl_workspace_error(None) == 0
    good = adv.clone()
    hold_half()
    rc = call()
    assert rc == 0
    assert torch.isfinite(adv).all() and not torch.equal(adv, good)
    assert torch.isnan(stats[1:]).all(), "a give-up leaves NaN statistics"
    assert lib.trl_workspace_error(None) == 1      # read ...
    assert lib.trl_workspace_error(None) == 0      # ... and cleared
    assert F.fused_degraded()                      # the device is degraded like after the host flag
    # whatever raised the host flag meanwhile (an earlier fused call of this process allocated it) is consumed here
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        F.td_lambda_return_estimate(0.99, 0.95, nv, r, done, term)
    assert call() in (0,)
    assert lib.trl_last_kernel().decode() == "normalize"
    torch.testing.assert_close(adv, good, rtol=1e-6, atol=1e-6)
    F.fused_reset()
    assert call() == 0
    assert lib.trl_last_kernel().decode() == "gae_fused"
    assert torch.equal(adv, good)


def test_one_launch_standardize_gives_up_loudly_and_degrades():
    """The persistent `_standardize` kernels (ungrouped and per-agent) have the same co-residency requirement and the same
    failure semantics: un-normalised output, never NaN, one warning on the next call, multi-launch route from then on."""
    from torchrl_amd._utils import _standardize

    hold_half = _holder()
    F.fused_reset()
    g = torch.Generator().manual_seed(9)
    x = (torch.randn((4096, 100, 5, 1), generator=g) * 2 + 3).to(DEV)
    for ex, kernel, slow in (((-2,), "standardize_grouped_persistent", "normalize_grouped"), ((), "standardize_persistent", "normalize")):
        want = _standardize(x, ex)
        assert F.last_kernel() == kernel
        torch.cuda.synchronize()
        hold_half()
        got = _standardize(x, ex)
        assert F.last_kernel() == kernel
        torch.cuda.synchronize()
        assert torch.isfinite(got).all()
        raw = got == x
        assert bool((raw | (got == want)).all()) and raw.float().mean() > 0.2
        with pytest.warns(RuntimeWarning, match="gave up waiting"):
            again = _standardize(x, ex)
        assert F.last_kernel() == slow and F.fused_degraded()
        torch.testing.assert_close(again, want, rtol=3e-6, atol=3e-6)
        F.fused_reset()
        assert torch.equal(_standardize(x, ex), want) and F.last_kernel() == kernel
    _clean_workspaces()
