"""Wide (1024-thread) transposing kernel with fp32 values and fp64 coefficients (python floats under default dtype float64):
double maps in LDS (up to 101 KB per block for V-trace).  Checks every estimator against the oracle."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import value_oracle as O  # noqa: E402
from torchrl_amd.objectives.value import functional as F  # noqa: E402

torch.set_default_dtype(torch.float64)
DEV = "cuda:0"
for shape in ((37, 100, 30), (9, 200, 20), (300, 100, 11)):
    g = torch.Generator().manual_seed(3)
    mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32)
    r, v, nv = mk(), mk(), mk()
    lp = torch.rand(shape, generator=g).clamp_min(1e-3).log().float()
    lm = torch.rand(shape, generator=g).clamp_min(1e-3).log().float()
    term = torch.rand(shape, generator=g) < 0.03
    done = term | (torch.rand(shape, generator=g) < 0.03)
    cu = lambda t: t.to(DEV)
    oa, ovt = O.gae(0.97, 0.9, v, nv, r, done, term)
    a, vt = F.generalized_advantage_estimate(0.97, 0.9, cu(v), cu(nv), cu(r), cu(done), cu(term))
    k1 = F.last_kernel()
    ot = O.td_lambda(0.97, 0.9, nv, r, done, term)
    t = F.td_lambda_return_estimate(0.97, 0.9, cu(nv), cu(r), cu(done), cu(term))
    k2 = F.last_kernel()
    ova, ovs = O.vtrace(0.97, lp, lm, v, nv, r, done, term)
    va, vs = F.vtrace_advantage_estimate(0.97, cu(lp), cu(lm), cu(v), cu(nv), cu(r), cu(done), cu(term))
    k3 = F.last_kernel()
    torch.cuda.synchronize()
    err = lambda x, y: float(((x.cpu().double() - y.double()).abs() / (1 + y.double().abs())).max())
    print(shape, k1, a.dtype, err(a, oa), err(vt, ovt), "|", k2, t.dtype, err(t, ot), "|", k3, va.dtype, err(va, ova), err(vs, ovs), flush=True)
