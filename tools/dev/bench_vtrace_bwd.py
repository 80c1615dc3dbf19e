"""V-trace backward alone at the C4 / C5 shapes (development: kernel variants through TRL_VALUE_LIB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402
from tools.bench_configs import inputs, timeit  # noqa: E402

f32, f64 = torch.float32, torch.float64
for name, shape, dt, b in (("C4 [65536x32]", (65536, 32, 1), f32, 4), ("C4 [65536x32]", (65536, 32, 1), f64, 8),
                           ("C5 [1024x10000]", (1024, 10000, 1), f32, 4), ("C5 [1024x10000]", (1024, 10000, 1), f64, 8)):
    r, v, nv, d, t, gam, lp, lm = inputs(shape, dt, 0.01)
    n = r.numel()
    ga, gv = torch.randn_like(r), torch.randn_like(r)
    cv = F._prepare_vtrace(0.99, lp, lm, v, nv, r, d, t, 1.0, 1.0, -2)
    _, vs = F._launch_vtrace(cv)
    us = min(timeit(lambda: F._vtrace_bwd(cv, ga, gv, vs, (True,) * 5, False), 50) for _ in range(3))
    bv = 13 * b + 2
    print(f"{name} V-trace backward {'f32' if dt == f32 else 'f64'}: {us:8.2f} us  {n * bv / us * 1e-3:8.1f} GB/s  {n * bv / us * 1e-3 / 8000:6.1%}  [{F.last_bwd_kernel()}]", flush=True)
