"""Kernel-only GAE timing for a list of (O, T, I) shapes: python tools/bench_shapes.py 20480x100x1 4096x100x5 ..."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402

dev = torch.device("cuda:0")
ALGO = int(os.environ.get("TRL_ALGO", "0"))  # 0 auto, 1 seq, 2 scan, 4 lockstep
for spec in sys.argv[1:]:
    dt = torch.float64 if spec.endswith("d") else torch.float32
    O, T, I = (int(x) for x in spec.rstrip("d").split("x"))
    shape = (O, T, I)
    r, v, nv = (torch.randn(shape, device=dev, dtype=dt) for _ in range(3))
    term = torch.rand(shape, device=dev) < 0.01
    done = term | (torch.rand(shape, device=dev) < 0.01)
    call = F._prepare_gae(0.9, 0.9, v, nv, r, done, term, -2)
    st = F.prepare_gae_step(call, normalize=False, algo=ALGO)
    for _ in range(10):
        st.run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        st.run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 200 * 1e3
    nb = O * T * I * (5 * r.element_size() + 2)
    kern = F.last_kernel()
    fn = lambda: F.generalized_advantage_estimate(0.9, 0.9, v, nv, r, done, term)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        fn()
    b.record()
    torch.cuda.synchronize()
    us_entry = a.elapsed_time(b) / 50 * 1e3
    print(f"{spec:18s} bound call {us:8.2f} us  {nb / us / 1e3:8.1f} GB/s  [{kern}]   entry point {us_entry:8.2f} us  [{F.last_kernel()}]", flush=True)
