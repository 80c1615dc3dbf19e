import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import functional as F
from tools.bench_configs import inputs
for dt in (torch.float32, torch.float64):
    r, v, nv, d, t, gam, lp, lm = inputs((1024, 10000, 1), dt, 0.001)
    for _ in range(10):
        F.generalized_advantage_estimate(gam, 0.95, v, nv, r, d, t)
    torch.cuda.synchronize()
