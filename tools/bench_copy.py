"""Achievable HBM bandwidth on this box with plain torch kernels (copy: 1 read + 1 write; add: 2 reads + 1 write), sizes
beyond the 256 MiB Infinity Cache -- the practical ceiling the streaming kernels are compared with in DESIGN.md."""
import torch

dev = torch.device("cuda:0")
for mb in (64, 256, 1024, 4096):
    n = mb * (1 << 20) // 4
    a = torch.randn(n, device=dev)
    b = torch.empty_like(a)
    c = torch.randn(n, device=dev)
    for name, fn, nbytes in (("copy", lambda: b.copy_(a), 8 * n), ("add", lambda: torch.add(a, c, out=b), 12 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:5s} {mb:5d} MiB per tensor: {us:9.1f} us  {nbytes / us / 1e3:8.1f} GB/s", flush=True)
