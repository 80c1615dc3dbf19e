"""Per-call cost of the module entry points (GAE / TDLambdaEstimator / VTrace .forward) beside the functional ones."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import functional as F
from torchrl_amd.objectives.value import GAE, TDLambdaEstimator, VTrace, RolloutData
DEV = "cuda:0"

def per(fn, n=2000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for shape in ((8, 128, 1), (65536, 32, 1)):
    g = torch.Generator().manual_seed(0)
    mk = lambda: torch.randn(shape, generator=g).to(DEV)
    v, nv, r = mk(), mk(), mk()
    term = (torch.rand(shape, generator=g) < 0.01).to(DEV)
    done = term | (torch.rand(shape, generator=g) < 0.01).to(DEV)
    lp = torch.rand(shape, generator=g).clamp_min(1e-3).log().to(DEV)
    data = RolloutData({"state_value": v, "sample_log_prob": lp, "next": {"state_value": nv, "reward": r, "done": done, "terminated": term}}, batch_size=shape[:2])
    gae = GAE(gamma=0.99, lmbda=0.95, value_network=None, average_gae=True, device=DEV)
    gae0 = GAE(gamma=0.99, lmbda=0.95, value_network=None, average_gae=False, device=DEV)
    tdl = TDLambdaEstimator(gamma=0.99, lmbda=0.95, value_network=None, device=DEV)
    print(f"{shape}: GAE(average_gae) {per(lambda: gae(data)):.2f} us | GAE {per(lambda: gae0(data)):.2f} | TDLambda {per(lambda: tdl(data)):.2f} | "
          f"F.gae_normalized {per(lambda: F.gae_normalized(0.99, 0.95, v, nv, r, done, term, eps=1e-4)):.2f} | F.gae {per(lambda: F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, done, term)):.2f}", flush=True)
