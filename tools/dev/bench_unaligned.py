"""Row lengths that break the 16-byte alignment of chunk / row starts (scalar accesses) against their aligned neighbours."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402
from tools.bench_configs import inputs, timeit  # noqa: E402

for dt, b in ((torch.float32, 4), (torch.float64, 8)):
    for shape in ((1024, 10000, 1), (1024, 9999, 1), (1024, 9998, 1), (8192, 1000, 1), (8192, 999, 1), (8192, 998, 1), (65536, 32, 1), (65536, 31, 1),
                  (16384, 300, 1), (16384, 301, 1), (4096, 128, 3, 1), (4096, 127, 3, 1)):
        r, v, nv, d, t, gam, lp, lm = inputs(shape, dt, 0.01)
        n = r.numel()
        td = -2 if len(shape) == 3 else -3
        call = F._prepare_gae(0.9, 0.9, v, nv, r, d, t, td)
        st = F.prepare_gae_step(call, normalize=False)
        us = min(timeit(st.run, 100) for _ in range(3))
        print(f"{str(dt)[6:]:8s} {str(shape):20s} GAE {us:8.2f} us {n*(5*b+2)/us*1e-3:7.1f} GB/s [{F.last_kernel()}]", flush=True)
