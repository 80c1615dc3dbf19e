"""GPU (-m gpu): the cross-GPU moment exchange (trl_xchg_*, trl_gae_normalize_xchg_fwd) rehearsed with
several ranks that share cuda:0 -- the IPC mailboxes, the epoch protocol, the in-launch cross-GPU
stage of the fused kernel and its three-launch fall-back, all through the C ABI, checked against the
process group's own all-reduce (gloo here; RCCL refuses ranks that share a device).  What a 1-GPU box
cannot show is xGMI itself: bench.py re-validates the exchange against RCCL on the real topology
before it uses it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_exchange_ranks_sharing_one_gpu(world):
    env = dict(os.environ, TRL_BENCH_BACKEND="gloo", TRL_FUSED_GRID_CAP=str(128 // world), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29520 + world), os.path.join(ROOT, "tools", "xchg_check.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "XCHG_CHECK PASS" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "gae_fused_xchg" in out.stdout
