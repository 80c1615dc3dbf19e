"""Host-side cost of the drop-in entry point on a small problem (C2): cProfile on the GPU box."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402

dev = "cuda:0"
shape = (8, 128, 1)
r, v, nv = (torch.randn(shape, device=dev) for _ in range(3))
d = torch.rand(shape, device=dev) < 0.01
f = lambda: F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, d, d)
for _ in range(200):
    f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5000):
    f()
torch.cuda.synchronize()
print("us per call", (time.perf_counter() - t0) / 5000 * 1e6)
pr = cProfile.Profile()
pr.enable()
for _ in range(5000):
    f()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
