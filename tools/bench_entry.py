"""Host cost of the public entry points (per call, back-to-back, one sync at the end) beside the device time of the same
launch bound once (PreparedStep): SURVEY.md 8(b) drop-in API overhead.
    python tools/bench_entry.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402

DEV = "cuda:0"


def per_call(fn, n=2000):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name, shape, td in (("C1 [1x2048]", (1, 2048, 1), -2), ("C2 [8x128]", (8, 128, 1), -2), ("C3 [4096x100x5]", (4096, 100, 5, 1), -3),
                        ("C4 [65536x32]", (65536, 32, 1), -2)):
    g = torch.Generator().manual_seed(0)
    r, v, nv = (torch.randn(shape, generator=g).to(DEV) for _ in range(3))
    mshape = shape[:len(shape) + td + 1] + (1,) * (-td - 1)
    term = (torch.rand(mshape, generator=g) < 0.01).to(DEV)
    done = (term | (torch.rand(mshape, generator=g) < 0.01).to(DEV)).expand(shape)   # per-step masks, expanded over the agents
    term = term.expand(shape)
    call = F._prepare_gae(0.99, 0.95, v, nv, r, done, term, td)
    k_gae = per_call(F.prepare_gae_step(call, normalize=False).run)
    k_norm = per_call(F.prepare_gae_step(call, 1e-4, normalize=True).run)
    e_gae = per_call(lambda: F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term, time_dim=td))
    e_vec = per_call(lambda: F.vec_generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term, time_dim=td))
    e_norm = per_call(lambda: F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4, time_dim=td))
    g0, gfull = torch.tensor(0.99), torch.full(shape, 0.99, device=DEV)
    e_gen = per_call(lambda: F.generalized_advantage_estimate(g0, 0.95, v, nv, r, done, term, time_dim=td))
    e_full = per_call(lambda: F.generalized_advantage_estimate(gfull, 0.95, v, nv, r, done, term, time_dim=td))
    e_td = per_call(lambda: F.td_lambda_return_estimate(gfull, 0.95, nv, r, done, term, time_dim=td))
    print(f"{name:16s} bound call: gae {k_gae:6.2f} us, gae+normalise {k_norm:6.2f} us | public API: generalized_advantage_estimate "
          f"{e_gae:6.2f} ({e_gae / k_gae:4.2f}x), vec {e_vec:6.2f}, gae_normalized {e_norm:6.2f} ({e_norm / k_norm:4.2f}x), "
          f"0-dim gamma tensor {e_gen:6.2f}, full-shape gamma tensor {e_full:6.2f} (TD(lambda) {e_td:6.2f})", flush=True)
