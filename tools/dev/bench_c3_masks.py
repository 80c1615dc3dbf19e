"""C3 [4096x100x5]: per-step masks expanded over the agents (read in place, byte gathers) against materialised masks."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402
from tools.bench_configs import inputs, timeit  # noqa: E402

shape = (4096, 100, 5, 1)
r, v, nv, d, t, *_ = inputs(shape, torch.float32, 0.01, mask_shape=(4096, 100, 1, 1))
for name, dd, tt in (("expanded", d, t), ("materialised", d.contiguous(), t.contiguous())):
    call = F._prepare_gae(0.9, 0.9, v, nv, r, dd, tt, -3)
    st = F.prepare_gae_step(call, normalize=False)
    us = min(timeit(st.run, 400) for _ in range(5))
    print(f"C3 masks {name:13s} {us:7.2f} us  [{F.last_kernel()}] bcast={call.mask_bcast}", flush=True)
