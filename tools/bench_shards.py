"""GPU time of the fused GAE + normalise step on the per-GPU shards of the strong-scaled BASELINE config 4 batch
([65536 / N x 32] on N GPUs), measured alone on one GPU: eager back-to-back launches (host- or launch-rate-bound for the small
shards) and hipGraph replays of 200 steps (GPU-bound: launch boundary + kernel).  What an 8-GPU node adds on top is the
cross-GPU exchange inside the launch (trl_gae_normalize_xchg_fwd).
    python tools/bench_shards.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from torchrl_amd.objectives.value import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)


def time_it(fn, reps, settle_s=0.03):
    t_end = time.perf_counter() + settle_s
    while time.perf_counter() < t_end:
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / reps)
    return sorted(best)[2]


K = 200
side = torch.cuda.Stream()
for n_gpus in (1, 2, 4, 8, 16):
    envs = bench.O_ENVS // n_gpus
    r, v, nv, done, term = bench.make_inputs(dev, 0, envs=envs)
    call = F._prepare_gae(bench.GAMMA, bench.LMBDA, v, nv, r, done, term, -2)
    eager = F.prepare_gae_step(call, bench.EPS, normalize=True)
    eager.run()
    kernel = F.last_kernel()
    us_eager = time_it(eager.run, 1000) * 1e3
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        step = F.prepare_gae_step(call, bench.EPS, normalize=True)
        step.run()
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(K):
                step.run()
    torch.cuda.synchronize()
    us_graph = time_it(g.replay, 10) * 1e3 / K
    n = envs * bench.T_STEPS
    print(f"shard [{envs:6d} x 32] (N = {n_gpus:2d})  {kernel:10s} eager {us_eager:6.2f} us/step   graph of {K} steps {us_graph:6.2f} us/step   "
          f"{n * 22 / us_graph * 1e-3:7.1f} GB/s per GPU   N x shard rate {n_gpus * n / us_graph * 1e6:.3e} transitions/s (excludes the cross-GPU exchange)",
          flush=True)
