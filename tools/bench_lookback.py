"""Few-long-rows timing: look-back kernel (workspace passed) vs the in-block chunk loop (no workspace).
    python tools/bench_lookback.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd import _extension  # noqa: E402
from torchrl_amd.objectives.value import functional as F  # noqa: E402

dev = torch.device("cuda:0")
lib = _extension.get_lib()


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for dtype in (torch.float32, torch.float64):
    for O, T in [(int(a), int(b)) for a, b in (x.split('x') for x in sys.argv[1:])] if len(sys.argv) > 1 else [(1, 100000), (4, 50000), (1, 1 << 20), (8, 1 << 20), (64, 1 << 17), (300, 8192), (512, 8192), (1, 1 << 24)]:
        shape = (O, T, 1)
        r, v, nv = (torch.randn(shape, device=dev, dtype=dtype) for _ in range(3))
        term = torch.rand(shape, device=dev) < 0.001
        done = term | (torch.rand(shape, device=dev) < 0.001)
        c = F._prepare_gae(0.99, 0.95, v, nv, r, done, term, -2)
        adv, vt = torch.empty_like(v), torch.empty_like(v)
        ws = F._workspace(lib, dev, O, T, 1)
        head = (F._stream(dev), F._DT[c.vdt], F._DT[c.cdt], F._ptr(c.reward), F._ptr(c.value), F._ptr(c.next_value),
                F._ptr(c.done), F._ptr(c.terminated), 0, None, None, c.gamma, c.lmbda, c.lg, O, T, 1, F._ptr(adv), F._ptr(vt), None)
        res = {}
        for name, tail in (("lookback", (F._ptr(ws), ws.numel(), F.ALGO_SCAN)), ("chunk_loop", (None, 0, F.ALGO_SCAN))):
            us = timed(lambda: lib.trl_gae_fwd(*head, *tail))
            res[name] = (us, F.last_kernel())
        es = 4 if dtype == torch.float32 else 8
        nbytes = O * T * (5 * es + 2)
        print(f"{str(dtype):14s} [{O} x {T}]  " + "  ".join(f"{k}: {u:9.1f} us ({kn}, {nbytes / u / 1e3:7.1f} GB/s)" for k, (u, kn) in res.items()), flush=True)
