"""Per-call time of the estimator modules (host + device) at C4 and C2: python tools/bench_module.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.objectives.value import GAE, RolloutData, TDLambdaEstimator, VTrace  # noqa: E402

DEV = "cuda:0"


def data(shape):
    full = tuple(shape) + (1,)
    mk = lambda: torch.randn(full, device=DEV)
    term = torch.rand(full, device=DEV) < 0.01
    done = term | (torch.rand(full, device=DEV) < 0.01)
    return RolloutData({"state_value": mk(), "sample_log_prob": torch.rand(full, device=DEV).log(),
                        "next": {"state_value": mk(), "reward": mk(), "done": done, "terminated": term}}, batch_size=shape)


def timeit(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6, (t1 - t0) / n * 1e6


for shape in ((65536, 32), (8, 128)):
    d = data(shape)
    for name, m in (("GAE", GAE(gamma=0.99, lmbda=0.95, value_network=None)),
                    ("GAE average_gae", GAE(gamma=0.99, lmbda=0.95, value_network=None, average_gae=True)),
                    ("TDLambdaEstimator", TDLambdaEstimator(gamma=0.99, lmbda=0.95, value_network=None))):
        tot, enq = timeit(lambda: m(d), 500)
        print(f"{shape}: {name:20s} {tot:8.2f} us / call (host enqueue {enq:.2f})", flush=True)

if os.environ.get("TRL_PROFILE"):
    import cProfile
    import pstats

    d = data((65536, 32))
    m = GAE(gamma=0.99, lmbda=0.95, value_network=None)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(1000):
        m(d)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
