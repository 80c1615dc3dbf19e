"""Shifted value layout (one (O, T + 1) buffer read in place) against the dense layout (two (O, T) tensors):
GAE + normalise as bound steps, device time per launch from hipGraph replays of 100 launches (host launch rate excluded).
Usage: python tools/bench_shifted.py [steps]"""
import sys

import torch

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402

DEV = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


side = torch.cuda.Stream()
K = 100  # launches per graph: device time, whatever the host's launch rate is


def timed(make_step):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        step = make_step()  # binds the capturing stream + its workspace
        step.run()
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(K):
                step.run()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(2, steps // K)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * K)


for O, T, dt in ((65536, 32, torch.float32), (65536, 32, torch.float64), (8192, 128, torch.float32), (262144, 32, torch.float32),
                 (1024, 10000, torch.float32)):
    g = torch.Generator(device=DEV).manual_seed(0)
    buf = torch.randn(O, T + 1, 1, device=DEV, generator=g).to(dt)
    r = torch.randn(O, T, 1, device=DEV, generator=g).to(dt)
    done = torch.rand(O, T, 1, device=DEV, generator=g) < 0.02
    v, nv = buf[:, :-1], buf[:, 1:]
    row = {}
    for name, (a, b) in (("dense", (v.clone(), nv.clone())), ("shifted", (v, nv))):
        call = F._prepare_gae(0.99, 0.95, a, b, r, done, done, -2)
        assert (call.next_value is None) == (name == "shifted")
        row[name] = (timed(lambda: F.prepare_gae_step(call, 1e-4)), timed(lambda: F.prepare_gae_step(call, normalize=False)))
        F.prepare_gae_step(call, 1e-4).run()
        kern = F.last_kernel()
    it = 4 if dt == torch.float32 else 8
    by = lambda per: O * T * per / 1e3  # bytes per us = MB/s *1e-3 -> GB/s = B / us / 1e3
    d, s = row["dense"], row["shifted"]
    print(f"[{O} x {T}] {str(dt)[6:]}: gae+normalise dense {d[0]:7.2f} us, shifted {s[0]:7.2f} us ({d[0] / s[0]:.3f}x); "
          f"gae dense {d[1]:7.2f} us, shifted {s[1]:7.2f} us ({d[1] / s[1]:.3f}x)   [{kern}]", flush=True)
