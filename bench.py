"""Benchmark of the north-star step: GAE + advantage normalisation at [65536 envs x 32 steps] fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one pass of the hot path over one batch of synthetic rollouts already resident in
HBM.  N = 1: trl_gae_normalize_fwd -- ONE launch (fused delta + scan + value_target + fp64 moments
+ in-launch gather + normalisation).  N > 1: every rank owns a full [65536 x 32] batch (weak
scaling: the env dimension is sharded) and the only exchange is the sum of the 3 fp64 moments.
Preferred route: trl_gae_normalize_xchg_fwd -- still ONE launch per GPU, the moments cross the GPUs
inside it (peer-to-peer mailbox stores over xGMI); it is validated against the RCCL route on the
live topology first and used only if every rank agrees.  Fallback: trl_gae_fwd (scan + moments) ->
RCCL all-reduce -> trl_normalize.  Rank 0 prints ONE JSON line
(schema in the task contract) with `roofline` for the dominant kernel and `cpu_baseline` (the CPU
oracle timed on this host, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from torchrl_amd.objectives.value import functional as F  # noqa: E402

O_ENVS, T_STEPS = 65536, 32
GAMMA, LMBDA, EPS = 0.99, 0.95, 1e-4
BYTES_PER_TRANSITION = 22  # fp32 GAE: 3x4 B + 2x1 B read, 2x4 B written; normalise adds no compulsory traffic
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
# HBM bytes per launch of the fused kernel from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
# passes, FETCH_SIZE x2 as the guide prescribes for gfx950): profiles/r01_h_bench_pmc_hbm_traffic.json.  A counter
# pass cannot run inside the timed bench, so the measured figure is carried here; null for the unprofiled N > 1 path.
PMC_TRAFFIC_BYTES_FUSED = 46802129


def make_inputs(dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    shape = (O_ENVS, T_STEPS, 1)
    r, v, nv = (torch.randn(shape, generator=g) for _ in range(3))
    term = torch.rand(shape, generator=g) < 0.01
    done = term | (torch.rand(shape, generator=g) < 0.01)
    return tuple(t.to(dev) for t in (r, v, nv, done, term))


def cpu_baseline(r, v, nv, done, term):
    """CPU oracle (scalar C restatement of the reference loop + normalise), 1 thread."""
    from oracle import value_oracle as O

    O.build()
    best = float("inf")
    reps = 0
    t_end = time.perf_counter() + 12.0
    while reps < 5 or (time.perf_counter() < t_end and reps < 200):
        t0 = time.perf_counter()
        adv, _ = O.gae(GAMMA, LMBDA, v, nv, r, done, term)
        O.normalize(adv, EPS)
        best = min(best, time.perf_counter() - t0)
        reps += 1
    return {
        "value": O_ENVS * T_STEPS / best,
        "unit": "transitions/s",
        "cores": 1,
        "kind": "port",
        "sample": f"full [{O_ENVS}x{T_STEPS}] fp32 GAE+normalise workload, min of {reps} runs of oracle/value_oracle.c "
                  f"(bit-identical restatement of the reference loop), host has {os.cpu_count()} cpus",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)  # ~20 ms of GPU time: long enough for the clocks to settle
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    probe_ok = False
    if dist and os.environ.get("TRL_BENCH_EXCHANGE", "1") != "0":
        # The exchange route stores into peers' HBM through IPC mappings: it is tried first in a child
        # process per rank (own gloo group over a file store), BEFORE this process touches the GPU, so
        # that an interconnect that misbehaves costs a probe, not the benchmark.
        import subprocess

        store = f"/tmp/trl_xchg_probe_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        try:
            pr = subprocess.run([sys.executable, "-m", "torchrl_amd.objectives.value.stats_exchange", store], cwd=ROOT,
                                env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")),
                                capture_output=True, text=True, timeout=120)
            probe_ok = pr.returncode == 0
            if not probe_ok and rank == 0:
                print("bench: exchange probe failed:\n" + pr.stdout[-2000:] + pr.stderr[-2000:], file=sys.stderr)
        except Exception as e:  # noqa: BLE001
            print(f"bench: exchange probe did not finish: {e}", file=sys.stderr)
    if dist:
        import torch.distributed as td

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # TRL_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks
        # (ranks share devices; gloo stages the 24-byte all-reduce through the host).  Default: RCCL.
        # (set TRL_FUSED_GRID_CAP = 256 / ranks-per-GPU there so that the ranks' kernels are co-resident)
        backend = os.environ.get("TRL_BENCH_BACKEND", "nccl")
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            td.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    r, v, nv, done, term = make_inputs(dev, seed=rank)
    call = F._prepare_gae(GAMMA, LMBDA, v, nv, r, done, term, -2)  # host canonicalisation: views only, no copies

    # The step is bound once to fixed buffers (a training loop keeps its rollout buffers in place):
    # the timed region contains the C-ABI calls (kernels) and, on the fallback route, the collective.
    collective = "none"
    norm = None
    if dist:
        def all_reduce_stats(st):
            if backend == "gloo":  # rehearsal: gloo has no device tensors
                h = st.cpu()
                td.all_reduce(h)
                st.copy_(h)
            else:
                td.all_reduce(st)

        gae = F.prepare_gae_step(call, normalize=False, want_stats=True)
        norm = F.prepare_normalize(gae.outputs[0], gae.stats, EPS)
        gae.run()
        all_reduce_stats(gae.stats)
        norm.run()
        ref_adv = gae.outputs[0].clone()
        collective = "all_reduce(3 x f64) over " + ("RCCL" if backend == "nccl" else backend)
        probes = [None] * world
        td.all_gather_object(probes, probe_ok)
        if all(probes):
            from torchrl_amd.objectives.value.stats_exchange import StatsExchange

            ex = StatsExchange.create(timeout_s=5.0)  # self-tested against the process group's all-reduce
            good = False
            if ex is not None:
                fused = F.prepare_gae_step(call, EPS, normalize=True, exchange=ex)
                for _ in range(3):
                    fused.run()
                torch.cuda.synchronize()
                good = (not ex.error()) and bool(torch.allclose(fused.outputs[0], ref_adv, rtol=2e-6, atol=2e-6))
            votes = [None] * world
            td.all_gather_object(votes, good)
            if all(votes):
                ex.set_timeout(20.0)
                gae, norm = fused, None
                collective = "in-launch peer-to-peer mailbox exchange (3 x f64 per rank over xGMI), validated against " + collective
            elif rank == 0:
                print(f"bench: exchange route not used (votes {votes}); falling back to {collective}", file=sys.stderr)
        elif rank == 0:
            print(f"bench: exchange route not used (probe results {probes}); using {collective}", file=sys.stderr)
        if rank == 0 and os.path.exists(f"/tmp/trl_xchg_probe_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"):
            os.remove(f"/tmp/trl_xchg_probe_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
    else:
        gae = F.prepare_gae_step(call, EPS, normalize=True)  # one launch (FUSED family)

    # HIP events on the launch stream: two events around the whole timed region.  With one kernel per
    # step (N = 1, and N > 1 on the exchange route) that is the kernel's average launch-to-launch
    # duration without per-step host work; on the RCCL fallback route it is the whole step.
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def step():
        gae.run()
        if norm is not None:
            all_reduce_stats(gae.stats)
            norm.run()
        return gae.outputs

    gae.run()
    kernel_name = F.last_kernel()  # the dominant kernel: the GAE entry point's
    gae_only_ms = None
    if norm is not None:  # fallback route: time the GAE kernel on its own (untimed region)
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(50):
            gae.run()
        ev[1].record()
        torch.cuda.synchronize()
        gae_only_ms = ev[0].elapsed_time(ev[1]) / 50
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        td.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(args.steps):
        step()
    ev[1].record()
    t_host = time.perf_counter() - t0  # host time to enqueue K steps
    torch.cuda.synchronize()
    if dist:
        td.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        n_trans = O_ENVS * T_STEPS
        gae_ms = gae_only_ms if gae_only_ms is not None else ev[0].elapsed_time(ev[1]) / args.steps  # HIP events on the launch stream
        achieved = n_trans * BYTES_PER_TRANSITION / (gae_ms * 1e-3) / 1e9
        out = {
            "metric": "GAE transitions/sec + % HBM roofline at 65536x32",
            "value": world * n_trans * args.steps / dt,
            "unit": "transitions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"GAE(gamma={GAMMA},lmbda={LMBDA})+normalise(eps={EPS}) [{O_ENVS} envs x {T_STEPS} steps] fp32 per GPU",
                       "per_gpu_transitions": n_trans, "parallelism": f"env-sharded x{world}",
                       "collective": collective},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": PMC_TRAFFIC_BYTES_FUSED if kernel_name == "gae_fused" else None, "kernel": kernel_name,
                         "kernel_ms": gae_ms, "bytes_per_launch": n_trans * BYTES_PER_TRANSITION,
                         "host_enqueue_ms_per_step": t_host / args.steps * 1e3},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(*(t.cpu() for t in (r, v, nv, done, term)))
        print(json.dumps(out))
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
