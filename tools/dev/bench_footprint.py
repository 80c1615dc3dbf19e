"""Is the fp32 / fp64 gap at C5 a cache-residency effect?  Same kernels, footprint swept across the 256 MiB Infinity Cache."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import functional as F
from tools.bench_configs import inputs, timeit
for dt, b in ((torch.float32, 4), (torch.float64, 8)):
    for rows in (256, 512, 1024, 2048, 4096):
        shape = (rows, 10000, 1)
        r, v, nv, d, t, gam, lp, lm = inputs(shape, dt, 0.001)
        n = r.numel()
        us = min(timeit(lambda: F.generalized_advantage_estimate(gam, 0.95, v, nv, r, d, t), 20, warm=20) for _ in range(2))
        print(f"{str(dt)[6:]:8s} {str(shape):18s} {n*(6*b+2)/1e6:8.1f} MB  GAE {us:8.2f} us {n*(6*b+2)/us*1e-3:7.1f} GB/s", flush=True)
        del r, v, nv, d, t, gam, lp, lm
