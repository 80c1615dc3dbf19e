"""hipGraph replay of the C4 GAE + normalise step ('and a hipGraph-replay variant').
    python tools/bench_graph.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from torchrl_amd.objectives.value import functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
r, v, nv, done, term = bench.make_inputs(dev, 0)
call = F._prepare_gae(bench.GAMMA, bench.LMBDA, v, nv, r, done, term, -2)
K = 200

eager = F.prepare_gae_step(call, bench.EPS, normalize=True)
ref_adv, ref_vt = (t.clone() for t in eager.run())
torch.cuda.synchronize()


def time_it(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


us_eager = time_it(eager.run, 2000) * 1e3
side = torch.cuda.Stream()
for steps_in_graph in (1, 10, K):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        step = F.prepare_gae_step(call, bench.EPS, normalize=True)  # binds the capturing stream + its workspace
        step.run()
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(steps_in_graph):
                step.run()
    torch.cuda.synchronize()
    step.outputs[0].zero_()
    g.replay()
    torch.cuda.synchronize()
    ok = torch.equal(step.outputs[0], ref_adv) and torch.equal(step.outputs[1], ref_vt)
    us = time_it(g.replay, max(4, 2000 // steps_in_graph)) * 1e3 / steps_in_graph
    print(f"graph of {steps_in_graph:4d} steps: {us:7.2f} us/step  (bit-equal to eager: {ok})   eager back-to-back: {us_eager:.2f} us/step", flush=True)
