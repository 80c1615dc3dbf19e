"""TD(lambda) / V-trace with feature dims: the generic transposing kernel (algo 2) against the lock-step kernels (algo 4) and AUTO."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402
from tools.bench_configs import inputs, timeit  # noqa: E402

BIG = [(torch.float32, s_) for s_ in ((4096, 100, 20, 1), (4096, 100, 30, 1), (4096, 100, 11, 1), (4096, 200, 10, 1), (4096, 150, 8, 1), (4096, 100, 16, 1))] \
    + [(torch.float64, s_) for s_ in ((4096, 100, 10, 1), (4096, 100, 8, 1))]
SHAPES = BIG if os.environ.get("TRL_BENCH_BIG") else [(torch.float32, s_) for s_ in ((4096, 100, 5, 1), (4096, 80, 5, 1), (8192, 64, 4, 1), (8192, 40, 3, 1), (8192, 60, 6, 1), (4096, 136, 7, 1),
                                         (4096, 127, 3, 1), (8192, 32, 8, 1), (4096, 200, 4, 1), (8192, 128, 4, 1), (8192, 100, 3, 1))] \
    + [(torch.float64, s_) for s_ in ((4096, 100, 5, 1), (8192, 40, 3, 1), (8192, 64, 4, 1), (4096, 99, 3, 1))]
for dt, shape in SHAPES:
    r, v, nv, d, t, gam, lp, lm = inputs(shape, dt, 0.01, mask_shape=shape[:2] + (1, 1))
    tcall = F._prepare_td_lambda(0.9, 0.9, nv, r, d, t, None, -3)
    vcall = F._prepare_vtrace(0.9, lp, lm, v, nv, r, d, t, 1.0, 1.0, -3)
    row = []
    for algo in (F.ALGO_AUTO, F.ALGO_SCAN, F.ALGO_LOCKSTEP):
        try:
            us_t = min(timeit(lambda: F._launch_td_lambda(tcall, dt, algo=algo), 200) for _ in range(3))
        except Exception:
            us_t = float("nan")
        kt = F.last_kernel()
        try:
            us_v = min(timeit(lambda: F._launch_vtrace(vcall, algo=algo), 200) for _ in range(3))
        except Exception:  # SCAN does not take V-trace rows longer than a team's chunk
            us_v = float("nan")
        row.append(f"algo {algo}: TD(l) {us_t:6.2f} [{kt.split('_')[-1]}] V-trace {us_v:6.2f} [{F.last_kernel().split('_')[-1]}]")
    print(f"{str(dt)[6:]} {str(shape):20s} " + " | ".join(row), flush=True)
