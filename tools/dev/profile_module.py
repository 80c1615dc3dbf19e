import cProfile, pstats, os, sys, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchrl_amd.objectives.value import GAE, RolloutData
DEV = "cuda:0"
shape = (65536, 32, 1)
g = torch.Generator().manual_seed(0)
mk = lambda: torch.randn(shape, generator=g).to(DEV)
v, nv, r = mk(), mk(), mk()
term = (torch.rand(shape, generator=g) < 0.01).to(DEV)
done = term | (torch.rand(shape, generator=g) < 0.01).to(DEV)
data = RolloutData({"state_value": v, "next": {"state_value": nv, "reward": r, "done": done, "terminated": term}}, batch_size=shape[:2])
gae = GAE(gamma=0.99, lmbda=0.95, value_network=None, average_gae=True, device=DEV)
for _ in range(200): gae(data)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): gae(data)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
