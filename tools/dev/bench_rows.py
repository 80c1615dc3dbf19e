"""Does the long-row kernel scale with the number of rows at fixed bytes?  (C5-like shapes, GAE / V-trace f64 + f32)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import functional as F
from tools.bench_configs import inputs, timeit
for dt, b in ((torch.float64, 8), (torch.float32, 4)):
    for shape in ((512, 20000, 1), (1024, 10000, 1), (2048, 5000, 1), (4096, 2500, 1)):
        r, v, nv, d, t, gam, lp, lm = inputs(shape, dt, 0.001)
        n = r.numel()
        us = min(timeit(lambda: F.generalized_advantage_estimate(gam, 0.95, v, nv, r, d, t), 30, warm=30) for _ in range(2))
        usv = min(timeit(lambda: F.vtrace_advantage_estimate(gam, lp, lm, v, nv, r, d, t), 30, warm=30) for _ in range(2))
        print(f"{str(dt)[6:]:8s} {str(shape):18s} GAE {us:8.2f} us {n*(6*b+2)/us*1e-3:7.1f} GB/s | V-trace {usv:8.2f} us {n*(8*b+2)/usv*1e-3:7.1f} GB/s [{F.last_kernel()}]", flush=True)
