"""Time every BASELINE.json config on one GPU (development / documentation tool).

Prints one line per (config, function): device time per call from HIP events over back-to-back
calls with fixed buffers, transitions/s, algorithmic bytes and the fraction of the 8 TB/s HBM peak.
    python tools/bench_configs.py [--iters 50]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.objectives.value import functional as F  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters, warm=300, settle_s=0.03):
    # clock-settle pre-roll: a fresh box runs the first milliseconds of work at a lower clock (round 3: 300 launches of a
    # 5 us kernel were not enough -- C2 read 7.2 us instead of 4.5): at least `warm` calls AND `settle_s` of wall clock
    import time
    t_end = time.perf_counter() + settle_s
    k = 0
    while k < warm or time.perf_counter() < t_end:
        fn()
        k += 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def inputs(shape, dtype, p, mask_shape=None):
    g = torch.Generator().manual_seed(0)
    mk = lambda: torch.randn(shape, generator=g, dtype=torch.float32).to(dtype).to(DEV)
    r, v, nv = mk(), mk(), mk()
    ms = mask_shape or shape
    term = (torch.rand(ms, generator=g) < p).to(DEV)
    done = term | (torch.rand(ms, generator=g) < p).to(DEV)
    if mask_shape:
        term, done = term.expand(shape), done.expand(shape)
    gam = (0.9 + 0.1 * torch.rand(shape, generator=g)).to(dtype).to(DEV)
    lp = torch.rand(shape, generator=g).clamp_min(1e-6).log().to(dtype).to(DEV)
    lm = torch.rand(shape, generator=g).clamp_min(1e-6).log().to(dtype).to(DEV)
    return r, v, nv, done, term, gam, lp, lm


def report(name, us, n, bpt):
    print(f"{name:58s} {us:9.2f} us  {n / us * 1e6:10.3e} trans/s  {n * bpt / 1e6:8.1f} MB  {n * bpt / us * 1e-3:8.1f} GB/s  {n * bpt / us * 1e-3 / 8000:6.1%}  [{F.last_kernel()}]")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    it = a.iters
    f32, f64 = torch.float32, torch.float64
    # C1 / C2: latency-bound
    for name, shape in (("C1 [1x2048]", (1, 2048, 1)), ("C2 [8x128]", (8, 128, 1))):
        r, v, nv, d, t, gam, lp, lm = inputs(shape, f32, 0.005)
        n = r.numel()
        call = F._prepare_gae(0.99, 0.95, v, nv, r, d, t, -2)
        st = F.prepare_gae_step(call, normalize=False, want_stats=False)
        report(f"{name} GAE f32", timeit(st.run, it), n, 22)
        report(f"{name} GAE f32 (python entry point)", timeit(lambda: F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, d, t), it), n, 22)
    # C3: VMAS, I = 5, expanded masks
    shape = (4096, 100, 5, 1)
    r, v, nv, d, t, gam, lp, lm = inputs(shape, f32, 0.01, mask_shape=(4096, 100, 1, 1))
    n = r.numel()
    report("C3 [4096x100x5] GAE f32 time_dim=-3 (python entry point)", timeit(lambda: F.generalized_advantage_estimate(0.9, 0.9, v, nv, r, d, t, time_dim=-3), it), n, 22)
    call = F._prepare_gae(0.9, 0.9, v, nv, r, d, t, -3)
    st = F.prepare_gae_step(call, normalize=False)
    report("C3 [4096x100x5] GAE f32 (kernel only)", timeit(st.run, it), n, 22)
    # PPO-side _standardize on the C3 advantage, agent dim kept independent (one launch: 4 B read, 4 B written per element)
    from torchrl_amd._utils import _standardize
    adv = st.run()[0].view(shape)
    report("C3 [4096x100x5] _standardize(exclude agent dim) f32", timeit(lambda: _standardize(adv, (-2,)), 4 * it), n, 8)
    report("C3 [4096x100x5] _standardize(all dims) f32", timeit(lambda: _standardize(adv, ()), 4 * it), n, 8)
    # the same two through the C ABI alone (fixed buffers): what the kernels cost without the Python host side
    from torchrl_amd import _extension as _ext
    from torchrl_amd import _utils as _u
    lib_ = _ext.get_lib()
    flat, o_ = adv.contiguous().view(-1), torch.empty_like(adv).view(-1)
    st_, wsp_, wsn_ = _u._scratch(lib_, flat.device, F._stream(flat.device), 409600, 5, 1)
    grouped = (F._stream(flat.device), 0, flat.data_ptr(), 409600, 5, 1, 1e-6, o_.data_ptr(), st_.data_ptr(), wsp_, wsn_)
    report("C3 [4096x100x5] trl_standardize_grouped (C call only)", timeit(lambda: lib_.trl_standardize_grouped(*grouped), 4 * it), n, 8)
    ws_ = F._workspace(lib_, flat.device, n, 1, 1)
    plain = (F._stream(flat.device), 0, flat.data_ptr(), n, 1e-6, o_.data_ptr(), None, ws_.data_ptr(), ws_.numel())
    report("C3 [4096x100x5] trl_standardize (C call only)", timeit(lambda: lib_.trl_standardize(*plain), 4 * it), n, 8)
    # C4
    shape = (65536, 32, 1)
    r, v, nv, d, t, gam, lp, lm = inputs(shape, f32, 0.01)
    n = r.numel()
    call = F._prepare_gae(0.99, 0.95, v, nv, r, d, t, -2)
    report("C4 [65536x32] GAE f32", timeit(F.prepare_gae_step(call, normalize=False).run, 4 * it), n, 22)
    report("C4 [65536x32] GAE+normalise f32 (fused)", timeit(F.prepare_gae_step(call, 1e-4, normalize=True).run, 4 * it), n, 22)
    report("C4 [65536x32] GAE+normalise f32 (scan+normalize kernels)", timeit(F.prepare_gae_step(call, 1e-4, normalize=True, algo=F.ALGO_SCAN).run, 4 * it), n, 22)
    # the per-GPU shards of the strong-scaled batch (BASELINE config 4: [65536 / N x 32] on N GPUs), alone on one GPU
    for n_gpus in (2, 4, 8):
        sh = (65536 // n_gpus, 32, 1)
        rs, vs_, nvs, ds, ts, *_ = inputs(sh, f32, 0.01)
        cs = F._prepare_gae(0.99, 0.95, vs_, nvs, rs, ds, ts, -2)
        report(f"C4 shard [{sh[0]}x32] (N = {n_gpus}) GAE+normalise f32 (fused)", timeit(F.prepare_gae_step(cs, 1e-4, normalize=True).run, 4 * it), rs.numel(), 22)
    report("C4 [65536x32] TD(lambda) f32", timeit(lambda: F.td_lambda_return_estimate(0.99, 0.95, nv, r, d, t), 4 * it), n, 14)
    report("C4 [65536x32] V-trace f32", timeit(lambda: F.vtrace_advantage_estimate(0.99, lp, lm, v, nv, r, d, t), 4 * it), n, 30)
    report("C4 [65536x32] GAE f32 SEQ family", timeit(F.prepare_gae_step(call, normalize=False, algo=F.ALGO_SEQ).run, it), n, 22)
    # C5
    shape = (1024, 10000, 1)
    for dt, nm, b in ((f32, "f32", 4), (f64, "f64", 8)):
        r, v, nv, d, t, gam, lp, lm = inputs(shape, dt, 0.001)
        n = r.numel()
        report(f"C5 [1024x10000] TD(lambda) {nm} tensor gamma", timeit(lambda: F.td_lambda_return_estimate(gam, 0.95, nv, r, d, t), it), n, 3 * b + 2 + b)
        report(f"C5 [1024x10000] V-trace {nm} tensor gamma", timeit(lambda: F.vtrace_advantage_estimate(gam, lp, lm, v, nv, r, d, t), it), n, 6 * b + 2 + 2 * b)
        report(f"C5 [1024x10000] GAE {nm} tensor gamma", timeit(lambda: F.generalized_advantage_estimate(gam, 0.95, v, nv, r, d, t), it), n, 4 * b + 2 + 2 * b)
        report(f"C5 [1024x10000] GAE {nm} scalar gamma", timeit(lambda: F.generalized_advantage_estimate(0.99, 0.95, v, nv, r, d, t), it), n, 3 * b + 2 + 2 * b)
    # backward kernels: device time of the backward launch alone, C4 and C5 shapes
    import ctypes  # noqa: F401
    from torchrl_amd import _extension
    lib = _extension.get_lib()
    for name, shape, dt, b in (("C4 [65536x32]", (65536, 32, 1), f32, 4), ("C5 [1024x10000]", (1024, 10000, 1), f32, 4),
                               ("C5 [1024x10000]", (1024, 10000, 1), f64, 8)):
        nm = "f32" if dt == f32 else "f64"
        r, v, nv, d, t, gam, lp, lm = inputs(shape, dt, 0.01)
        n = r.numel()
        c = F._prepare_gae(0.99, 0.95, v, nv, r, d, t, -2)
        ga, gv = torch.randn_like(r), torch.randn_like(r)
        F._gae_bwd(c, ga, gv, True, True, True)
        us = timeit(lambda: F._gae_bwd(c, ga, gv, True, True, True), it)
        print(f"{name + ' GAE backward ' + nm:58s} {us:9.2f} us  {n / us * 1e6:10.3e} trans/s  {n * (5 * b + 2) / 1e6:8.1f} MB  "
              f"{n * (5 * b + 2) / us * 1e-3:8.1f} GB/s  {n * (5 * b + 2) / us * 1e-3 / 8000:6.1%}  [{F.last_bwd_kernel()}]")
        ct = F._prepare_td_lambda(gam, 0.95, nv, r, d, t, None, -2)
        F._td_bwd(ct, ga, False, True, True)
        us = timeit(lambda: F._td_bwd(ct, ga, False, True, True), it)
        bt = 4 * b + 2  # g_ret + gamma read, 2 written
        print(f"{name + ' TD(lambda) backward ' + nm + ' tensor gamma':58s} {us:9.2f} us  {n / us * 1e6:10.3e} trans/s  {n * bt / 1e6:8.1f} MB  "
              f"{n * bt / us * 1e-3:8.1f} GB/s  {n * bt / us * 1e-3 / 8000:6.1%}  [{F.last_bwd_kernel()}]")
        cv = F._prepare_vtrace(0.99, lp, lm, v, nv, r, d, t, 1.0, 1.0, -2)
        _, vs = F._launch_vtrace(cv)
        F._vtrace_bwd(cv, ga, gv, vs, (True,) * 5, False)
        us = timeit(lambda: F._vtrace_bwd(cv, ga, gv, vs, (True,) * 5, False), it)
        bv = 13 * b + 2  # 8 tensors read, 5 written
        print(f"{name + ' V-trace backward ' + nm:58s} {us:9.2f} us  {n / us * 1e6:10.3e} trans/s  {n * bv / 1e6:8.1f} MB  "
              f"{n * bv / us * 1e-3:8.1f} GB/s  {n * bv / us * 1e-3 / 8000:6.1%}  [{F.last_bwd_kernel()}]")


if __name__ == "__main__":
    main()
