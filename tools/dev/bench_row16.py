"""Interleaved GAE, rows of 64 < T <= 128 steps: (o, i) rows as 16-lane DPP rows against the straddling V = 4 mapping."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402
from tools.bench_configs import inputs, timeit  # noqa: E402

SHAPES32 = [(torch.float32, s_) for s_ in ((4096, 127, 3, 1), (4096, 99, 3, 1), (4096, 65, 7, 1), (4096, 33, 7, 1), (8192, 25, 5, 1), (4096, 201, 5, 1), (4096, 100, 5, 1))]
for dt, shape in SHAPES32:
    r, v, nv, d, t, *_ = inputs(shape, dt, 0.01, mask_shape=shape[:2] + (1, 1))
    call = F._prepare_gae(0.9, 0.9, v, nv, r, d, t, -3)
    for algo in (F.ALGO_AUTO, F.ALGO_SCAN, F.ALGO_LOCKSTEP):
        st = F.prepare_gae_step(call, normalize=False, algo=algo)
        us = min(timeit(st.run, 400) for _ in range(5))
        print(f"{str(dt)[6:]} {str(shape):22s} algo {algo} {us:7.2f} us  [{F.last_kernel()}] row32={'off' if os.environ.get('TRL_GIL_ROW32_OFF') else 'on'}", flush=True)
