"""GPU (-m gpu): the single-launch GAE + normalise kernel must never fail silently (VERDICT r1 / ADVICE r1).
Its persistent grid needs every compute unit; here half of them are held by a test kernel (tests/native/hold_cus.hip:
one block per CU that takes the CU's whole LDS and sleeps), so the resident half of the grid waits its bounded 1 s, gives
up, writes the UN-NORMALISED advantage and raises the host-visible flag; the NEXT library call fails loudly with
TRL_ERR_LAUNCH (no host synchronisation involved), and the call after that works again."""
import ctypes
import os

import pytest
import torch

from torchrl_amd import _extension
from torchrl_amd.objectives.value import functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "libhold_cus.so")


def test_fused_gives_up_loudly_when_cus_are_held():
    if not os.path.exists(_SO):
        pytest.skip("tests/native/libhold_cus.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    torch.cuda.init()
    hold = ctypes.CDLL(_SO)
    hold.hold_cus.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lds = hold.hold_max_lds()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if lds < 150 * 1024:
        pytest.skip(f"a block can take only {lds} bytes of LDS: it cannot hold a CU on its own")

    shape = (65536, 32, 1)
    g = torch.Generator().manual_seed(0)
    r, v, nv = (torch.randn(shape, generator=g).to(DEV) for _ in range(3))
    term = (torch.rand(shape, generator=g) < 0.01).to(DEV)
    done = term | (torch.rand(shape, generator=g) < 0.01).to(DEV)
    want_adv, want_vt = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)
    raw_adv, _ = F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term)
    assert F.last_kernel() != "gae_fused"
    torch.cuda.synchronize()
    assert not F.workspace_error()

    side = torch.cuda.Stream()
    assert hold.hold_cus(ctypes.c_void_p(side.cuda_stream), cus // 2, lds, 1.6) == 0   # half of the CUs, for 1.6 s
    import time
    time.sleep(0.05)  # the holders are resident before the fused grid is dispatched
    adv, vt = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)
    assert F.last_kernel() == "gae_fused"
    torch.cuda.synchronize()

    # never NaN: every element is either normalised (blocks that started after the holders left) or left un-normalised
    assert torch.isfinite(adv).all()
    assert torch.equal(vt, want_vt)
    given_up = adv == raw_adv
    normalised = adv == want_adv
    assert bool((given_up | normalised).all())
    assert given_up.float().mean() > 0.2, "the resident half of the grid must have given up"
    assert F.workspace_error()

    # the next call on this GPU fails loudly, without anyone having to poll or synchronise ...
    with pytest.raises(_extension.TrlError, match="gave up waiting"):
        F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)
    # ... and the one after that works again
    adv2, vt2 = F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)
    torch.cuda.synchronize()
    assert torch.equal(adv2, want_adv) and torch.equal(vt2, want_vt)
    # leave the shared workspace clean for the tests that follow (the error word is sticky by design)
    for ws in F._WORKSPACES.values():
        ws[4:8].zero_()
    torch.cuda.synchronize()
    assert not F.workspace_error()
