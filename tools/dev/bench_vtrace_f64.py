"""V-trace / GAE / TD(lambda) forward at rows of 10000 steps: throughput against the batch's footprint (Infinity Cache = 256 MiB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402
from tools.bench_configs import inputs, timeit  # noqa: E402

for dt, b in ((torch.float64, 8), (torch.float32, 4)):
    for O in (128, 256, 512, 1024, 2048):
        shape = (O, 10000, 1)
        r, v, nv, d, t, gam, lp, lm = inputs(shape, dt, 0.001)
        n = r.numel()
        us = min(timeit(lambda: F.generalized_advantage_estimate(gam, 0.95, v, nv, r, d, t), 30, warm=30) for _ in range(3))
        usv = min(timeit(lambda: F.vtrace_advantage_estimate(gam, lp, lm, v, nv, r, d, t), 30, warm=30) for _ in range(3))
        print(f"{str(dt)[6:]:8s} {str(shape):18s} GAE {us:8.2f} us {n*(6*b+2)/1e6:7.1f} MB {n*(6*b+2)/us*1e-3:7.1f} GB/s | V-trace {usv:8.2f} us "
              f"{n*(8*b+2)/1e6:7.1f} MB {n*(8*b+2)/usv*1e-3:7.1f} GB/s", flush=True)
