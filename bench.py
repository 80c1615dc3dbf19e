"""Benchmark of the north-star step: GAE + advantage normalisation at [65536 envs x 32 steps] fp32.

    python bench.py [--gpus N] [--steps K] [--warmup W]

    python bench.py --gpus N [--scaling strong|weak]

One "step" is one pass of the hot path over one batch of synthetic rollouts already resident in
HBM.  N = 1: trl_gae_normalize_fwd -- ONE launch (fused delta + scan + value_target + fp64 moments
+ in-launch gather + normalisation).  N > 1, `--scaling strong` (the default; BASELINE config 4): ONE
[65536 x 32] batch sharded over the ranks along the env dimension, [65536 / N x 32] per GPU.
`--scaling weak`: every rank owns a full [65536 x 32] batch.  Either way the only exchange is the
sum of the 3 fp64 moments.
Preferred route: trl_gae_normalize_xchg_fwd -- still ONE launch per GPU, the moments cross the GPUs
inside it (peer-to-peer mailbox stores over xGMI); it is validated against the RCCL route on the
live topology first and used only if every rank agrees.  Other route, always measured beside it:
trl_gae_fwd (scan + moments) -> RCCL all-reduce(3 x f64) -> trl_normalize.

`--gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself, as fresh
child processes under torch.distributed.run, BEFORE this process touches the GPU, and relays rank 0's
line; on a box with fewer than N GPUs the ranks share devices over gloo (a rehearsal of the control
flow, flagged in `config.rehearsal`).  Rank 0 prints ONE JSON line (schema in the task contract) with
`roofline` for the dominant kernel and `cpu_baseline` (the CPU oracle timed on this host, N = 1 only).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

O_ENVS, T_STEPS = 65536, 32
GAMMA, LMBDA, EPS = 0.99, 0.95, 1e-4
BYTES_PER_TRANSITION = 22  # fp32 GAE: 3x4 B + 2x1 B read, 2x4 B written; normalise adds no compulsory traffic
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
PREROLL_STEPS = 1500       # untimed clock-settle pre-roll (~12 ms of GPU time), outside `warmup`, reported in the line
KERNEL_BATCHES, KERNEL_BATCH = 5, 200   # kernel_ms = median over 5 batches of 200 back-to-back launches (all batches reported)
COLD_SETS = 10             # cold leg: rotate over 10 distinct input / output buffer sets, 461 MB > the 256 MiB Infinity Cache


def pmc_traffic(kernel_name, cold=False):
    """HBM bytes per launch of the fused kernel from the PMC counters: the newest profiles/r*_pmc_hbm_traffic.json
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x2 as the guide prescribes for gfx950;
    tools/profile_bench.sh + tools/summarize_pmc.py).  A counter pass cannot run inside the timed bench, so the measured
    figure of the committed profile is carried; None when there is none for this kernel."""
    if kernel_name != "gae_fused":
        return None, None
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json"))
                   if ("_cold_" in os.path.basename(f)) == cold)   # the cold leg has its own counter passes (--cold-only)
    for f in reversed(files):
        try:
            with open(f) as fh:
                d = json.load(fh)
            return int(round(d["hbm_bytes_per_launch"]["total"])), os.path.relpath(f, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def make_inputs(dev, seed, envs=O_ENVS):
    import torch

    g = torch.Generator(device="cpu").manual_seed(seed)
    shape = (envs, T_STEPS, 1)
    r, v, nv = (torch.randn(shape, generator=g) for _ in range(3))
    term = torch.rand(shape, generator=g) < 0.01
    done = term | (torch.rand(shape, generator=g) < 0.01)
    return tuple(t.to(dev) for t in (r, v, nv, done, term))


def cpu_baseline(r, v, nv, done, term):
    """CPU oracle (C restatement of the reference loop + normalise) on the host: 1 thread, and all host cores with the env
    dimension split across OpenMP threads (per-shard GAE and fp64 moments, one combine, parallel scale pass -- the same
    decomposition the GPUs use)."""

    from oracle import value_oracle as O

    O.build()
    O.lib()

    def one_core():
        adv, _ = O.gae(GAMMA, LMBDA, v, nv, r, done, term)
        O.normalize(adv, EPS)

    used = [1]

    def all_cores():
        used[0] = O.gae_normalize_mt(GAMMA, LMBDA, v, nv, r, done, term, EPS)[2]  # OpenMP: OMP_NUM_THREADS (16 on the box) or all cpus

    def best_of(fn, budget_s, max_reps):
        best, reps = float("inf"), 0
        t_end = time.perf_counter() + budget_s
        while reps < 3 or (time.perf_counter() < t_end and reps < max_reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
            reps += 1
        return best, reps

    t1, n1 = best_of(one_core, 8.0, 100)
    tc, nc = best_of(all_cores, 8.0, 400)
    n_trans = O_ENVS * T_STEPS
    return {
        "value": n_trans / tc,
        "unit": "transitions/s",
        "cores": used[0],
        "kind": "port",
        "sample": f"full [{O_ENVS}x{T_STEPS}] fp32 GAE+normalise workload, oracle/value_oracle.c (bit-identical restatement of the "
                  f"reference loop): env dimension split over {used[0]} OpenMP threads, min of {nc} runs; host reports {os.cpu_count()} cpus",
        "one_core": {"value": n_trans / t1, "cores": 1, "sample": f"same workload, 1 thread, min of {n1} runs"},
    }


def self_launch(args):
    """`--gpus N` with no launcher around us: start the N ranks as fresh child processes (torch.distributed.run),
    before this process makes any GPU call, relay rank 0's JSON line, fail if any rank fails."""
    import torch  # importing torch and counting devices does not initialise the GPU

    ndev = torch.cuda.device_count()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    if ndev < args.gpus:
        # fewer GPUs than ranks: the ranks share devices -- RCCL cannot do that, gloo carries the 24-byte all-reduce, and the
        # ranks' persistent grids must be co-resident on the shared GPU
        per_gpu = -(-args.gpus // max(ndev, 1))
        env.setdefault("TRL_BENCH_BACKEND", "gloo")
        env.setdefault("TRL_FUSED_GRID_CAP", str(max(1, 256 // per_gpu)))
        env["TRL_BENCH_REHEARSAL"] = f"{args.gpus} ranks on {ndev} GPU(s)"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--scaling", args.scaling] + (["--no-cpu-baseline"] if args.no_cpu_baseline else [])
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    sys.stderr.write(p.stderr[-6000:])
    if p.returncode != 0 or line is None:
        sys.stderr.write(f"bench: the {args.gpus}-rank run failed (rc {p.returncode})\n" + p.stdout[-2000:])
        sys.exit(p.returncode or 1)
    print(line)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = one [65536 x 32] batch sharded over the ranks (BASELINE config 4); weak = a full batch per rank")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold (beyond-Infinity-Cache) leg")
    ap.add_argument("--cold-only", action="store_true",
                    help="profiling aid (tools/profile_bench.sh): run ONLY the cold leg's launches, so that rocprofv3's kernel "
                         "statistics and PMC counters of this process describe the DRAM-bound launches alone; prints a short line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return

    import torch

    from torchrl_amd.objectives.value import functional as F

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    probe_ok = False
    probe_store = f"/tmp/trl_xchg_probe_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    if dist and os.environ.get("TRL_BENCH_EXCHANGE", "1") != "0":
        # The exchange route stores into peers' HBM through IPC mappings: it is tried first in a child
        # process per rank (own gloo group over a file store), BEFORE this process touches the GPU, so
        # that an interconnect that misbehaves costs a probe, not the benchmark.
        try:
            pr = subprocess.run([sys.executable, "-m", "torchrl_amd.objectives.value.stats_exchange", probe_store], cwd=ROOT,
                                env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")),
                                capture_output=True, text=True, timeout=120)
            probe_ok = pr.returncode == 0
            if not probe_ok and rank == 0:
                print("bench: exchange probe failed:\n" + pr.stdout[-2000:] + pr.stderr[-2000:], file=sys.stderr)
        except Exception as e:  # noqa: BLE001
            print(f"bench: exchange probe did not finish: {e}", file=sys.stderr)
    backend = "none"
    if dist:
        import torch.distributed as td

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # TRL_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks
        # (ranks share devices; gloo stages the 24-byte all-reduce through the host).  Default: RCCL.
        backend = os.environ.get("TRL_BENCH_BACKEND", "nccl")
        local_rank = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            td.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    strong = dist and args.scaling == "strong"
    if dist and O_ENVS % world:
        raise SystemExit(f"bench: the world size must divide {O_ENVS}")

    # The step is bound once to fixed buffers (a training loop keeps its rollout buffers in place):
    # the timed region contains the C-ABI calls (kernels) and, on the RCCL route, the collective.
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def device_ms(fn, n):
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(n):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / n

    if args.cold_only:
        if dist:
            raise SystemExit("bench: --cold-only is a single-GPU profiling aid")
        sets = []
        for k in range(COLD_SETS):
            ins = make_inputs(dev, seed=1000 + k)
            ck = F._prepare_gae(GAMMA, LMBDA, ins[1], ins[2], ins[0], ins[3], ins[4], -2)
            sets.append(F.prepare_gae_step(ck, EPS, normalize=True))
        rounds = max(1, args.steps // COLD_SETS)

        def cold_round():
            for st in sets:
                st.run()

        for _ in range(max(1, args.warmup // COLD_SETS)):
            cold_round()
        ms = device_ms(cold_round, rounds) / COLD_SETS
        print(json.dumps({"cold_only": True, "kernel": F.last_kernel(), "launches": rounds * COLD_SETS, "buffer_sets": COLD_SETS,
                          "kernel_ms": ms, "achieved_gbs": O_ENVS * T_STEPS * BYTES_PER_TRANSITION / (ms * 1e-3) / 1e9}))
        return

    exchange = {"tried": False, "ex": None}

    def get_exchange():
        """The in-launch mailbox exchange, created once (collective call: every rank) if every rank's probe succeeded."""
        if not exchange["tried"]:
            exchange["tried"] = True
            probes = [None] * world
            td.all_gather_object(probes, probe_ok)
            if all(probes):
                from torchrl_amd.objectives.value.stats_exchange import StatsExchange

                exchange["ex"] = StatsExchange.create(timeout_s=5.0)  # self-tested against the process group's all-reduce
            elif rank == 0:
                print(f"bench: exchange route not used (probe results {probes})", file=sys.stderr)
            if rank == 0 and os.path.exists(probe_store):
                os.remove(probe_store)
        return exchange["ex"]

    def run_mode(envs, steps, warmup):
        """Build the step for `envs` env rows on this rank, measure the dominant kernel and time `steps` steps."""
        r, v, nv, done, term = make_inputs(dev, seed=rank, envs=envs)
        call = F._prepare_gae(GAMMA, LMBDA, v, nv, r, done, term, -2)  # host canonicalisation: views only, no copies
        collective = "none"
        routes = {}
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

            def rccl_step():
                gae.run()
                all_reduce_stats(gae.stats)
                norm.run()
                return gae.outputs

            rccl_step()
            ref_adv = gae.outputs[0].clone()
            rccl_name = "all_reduce(3 x f64) over " + ("RCCL" if backend == "nccl" else backend)
            collective = rccl_name
            step = rccl_step
            kernel_step = gae.run
            mailbox_step = None
            ex = get_exchange()
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
                mailbox_step = fused.run
            elif rank == 0:
                print(f"bench: exchange route not used (votes {votes}); falling back to {collective}", file=sys.stderr)
            # both routes, measured beside each other (untimed region; every rank runs the same sequence)
            for _ in range(20):
                rccl_step()
            td.barrier()
            routes["rccl_all_reduce_ms"] = device_ms(rccl_step, 100)
            if mailbox_step is not None:
                for _ in range(20):
                    mailbox_step()
                td.barrier()
                routes["in_launch_mailbox_ms"] = device_ms(mailbox_step, 200)
                step = kernel_step = mailbox_step
                collective = ("in-launch peer-to-peer mailbox exchange (3 x f64 per rank over xGMI), validated against " + rccl_name)
        else:
            one = F.prepare_gae_step(call, EPS, normalize=True)  # one launch (FUSED family)
            step = kernel_step = one.run

        step()
        kernel_name = F.last_kernel()  # the dominant kernel: the GAE entry point's
        # untimed clock-settle pre-roll: a short run would otherwise time the kernel while the clocks still ramp
        for _ in range(PREROLL_STEPS):
            kernel_step()
        torch.cuda.synchronize()
        # HIP events on the launch stream around back-to-back launches of the dominant kernel: median of 5 batches of 200
        kernel_batches = sorted(device_ms(kernel_step, KERNEL_BATCH) for _ in range(KERNEL_BATCHES))
        kernel_ms = kernel_batches[KERNEL_BATCHES // 2]
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()
        # HIP events around the timed steps are informational (timed_region_device_ms_per_step); each record is a marker packet
        # on the stream and a few microseconds of host time, which a short run would carry as per-step cost: long runs only
        with_events = steps >= 200
        t0 = time.perf_counter()
        if with_events:
            ev[0].record()
        for k in range(steps):
            step()
        if with_events:
            ev[1].record()
        t_host = time.perf_counter() - t0  # host time to enqueue K steps
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timed_ms = ev[0].elapsed_time(ev[1]) / steps if with_events else None
        if dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            if backend == "gloo":
                h = tmax.cpu()
                td.all_reduce(h, op=td.ReduceOp.MAX)
                tmax = h
            else:
                td.all_reduce(tmax, op=td.ReduceOp.MAX)
            dt = float(tmax.item())
        return {"envs": envs, "dt": dt, "kernel_ms": kernel_ms, "kernel_batches": kernel_batches, "kernel_name": kernel_name,
                "routes": routes, "collective": collective, "t_host": t_host, "timed_ms": timed_ms, "inputs": (r, v, nv, done, term)}

    envs = O_ENVS // world if strong else O_ENVS   # env rows this rank owns
    m = run_mode(envs, args.steps, args.warmup)
    r, v, nv, done, term = m["inputs"]
    dt, kernel_ms, kernel_batches, kernel_name = m["dt"], m["kernel_ms"], m["kernel_batches"], m["kernel_name"]
    routes, collective, t_host, timed_ms = m["routes"], m["collective"], m["t_host"], m["timed_ms"]
    # N > 1: the other scaling mode beside it (same step count), so that one run documents both curves
    other = None
    if dist:
        o_envs = O_ENVS if strong else O_ENVS // world
        mo = run_mode(o_envs, args.steps, args.warmup)
        other = {"scaling": "weak" if strong else "strong", "per_gpu_envs": o_envs, "value": o_envs * T_STEPS * world * args.steps / mo["dt"],
                 "ms_per_step": mo["dt"] / args.steps * 1e3, "kernel_ms": mo["kernel_ms"], "route_ms_per_step": mo["routes"],
                 "collective": mo["collective"]}
        del mo

    # the public entry point (host canonicalisation + allocation + the same launch), per call
    entry_point_ms = module_entry_ms = None
    if not dist:
        for _ in range(50):
            F.gae_normalized(GAMMA, LMBDA, v, nv, r, done, term, eps=EPS)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(300):
            F.gae_normalized(GAMMA, LMBDA, v, nv, r, done, term, eps=EPS)
        torch.cuda.synchronize()
        entry_point_ms = (time.perf_counter() - t1) / 300 * 1e3
        # and the module the reference's PPO recipes call: GAE(average_gae=True).forward on a tensordict-like batch
        from torchrl_amd.objectives.value import GAE, RolloutData
        mod = GAE(gamma=GAMMA, lmbda=LMBDA, value_network=None, average_gae=True, device=dev)
        batch = RolloutData({"state_value": v, "next": {"state_value": nv, "reward": r, "done": done, "terminated": term}},
                            batch_size=(O_ENVS, T_STEPS))
        for _ in range(50):
            mod(batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(300):
            mod(batch)
        torch.cuda.synchronize()
        module_entry_ms = (time.perf_counter() - t1) / 300 * 1e3

    # cold leg (N = 1): the same launch rotating over COLD_SETS distinct input / output buffer sets, so that no byte of a
    # launch is left in the 256 MiB Infinity Cache by the previous ones -- the DRAM-bound figure beside the warm one --
    # and, for scale, what a plain device copy (1 read + 1 write, torch) reaches on buffers beyond the cache on this box
    cold = None
    if not dist and not args.no_cold:
        sets = []
        for k in range(COLD_SETS):
            ins = make_inputs(dev, seed=1000 + k)
            ck = F._prepare_gae(GAMMA, LMBDA, ins[1], ins[2], ins[0], ins[3], ins[4], -2)
            sets.append(F.prepare_gae_step(ck, EPS, normalize=True))

        def cold_round():
            for st in sets:
                st.run()

        for _ in range(20):
            cold_round()
        cold_batches = sorted(device_ms(cold_round, 20) / COLD_SETS for _ in range(KERNEL_BATCHES))
        cold_ms = cold_batches[KERNEL_BATCHES // 2]
        del sets
        n_copy = 512 * (1 << 20) // 4
        src, dst = torch.empty(n_copy, device=dev).normal_(), torch.empty(n_copy, device=dev)
        for _ in range(3):
            dst.copy_(src)
        copy_gbs = 8 * n_copy / (device_ms(lambda: dst.copy_(src), 10) * 1e-3) / 1e9
        del src, dst
        cold_bytes = O_ENVS * T_STEPS * BYTES_PER_TRANSITION
        cold_traffic, cold_traffic_src = pmc_traffic("gae_fused", cold=True)
        cold = {"traffic": cold_traffic, "traffic_source": cold_traffic_src, "kernel_ms": cold_ms, "kernel_ms_batches": cold_batches, "achieved": cold_bytes / (cold_ms * 1e-3) / 1e9,
                "frac": cold_bytes / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "buffer_sets": COLD_SETS,
                "footprint_bytes": COLD_SETS * O_ENVS * T_STEPS * BYTES_PER_TRANSITION,
                "copy_ceiling_gbs": copy_gbs, "frac_of_copy_ceiling": cold_bytes / (cold_ms * 1e-3) / 1e9 / copy_gbs,
                "note": f"same launch rotating over {COLD_SETS} distinct buffer sets (461 MB > 256 MiB Infinity Cache): DRAM-bound; "
                        "copy_ceiling_gbs = torch device copy of 512 MiB (1 read + 1 write) measured in this run"}

    if rank == 0:
        n_trans = envs * T_STEPS               # transitions one GPU processes per step
        n_job = n_trans * world                # transitions the whole job processes per step
        achieved = n_trans * BYTES_PER_TRANSITION / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(kernel_name if world == 1 else None)
        out = {
            "metric": "GAE transitions/sec + % HBM roofline at 65536x32",
            "value": n_job * args.steps / dt,
            "unit": "transitions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": (f"GAE(gamma={GAMMA},lmbda={LMBDA})+normalise(eps={EPS}) ONE [{O_ENVS} envs x {T_STEPS} steps] fp32 batch sharded "
                                    f"over {world} GPUs as {world} x [{envs} x {T_STEPS}] (strong scaling, BASELINE config 4)" if strong else
                                    f"GAE(gamma={GAMMA},lmbda={LMBDA})+normalise(eps={EPS}) [{O_ENVS} envs x {T_STEPS} steps] fp32 per GPU"
                                    + (" (weak scaling)" if dist else ""))
                                   + "; the working set is Infinity-Cache-resident across back-to-back steps (roofline.cold: the DRAM-bound leg)",
                       "per_gpu_transitions": n_trans, "job_transitions": n_job, "parallelism": f"env-sharded x{world}",
                       "collective": collective, "preroll_steps": PREROLL_STEPS},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel_name,
                         "kernel_ms": kernel_ms, "kernel_ms_batches": kernel_batches,
                         "bytes_per_launch": n_trans * BYTES_PER_TRANSITION,
                         "timed_region_device_ms_per_step": timed_ms,
                         "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
                         "note": "fraction of the 8 TB/s HBM peak; kernel_ms = median of 5 batches of 200 back-to-back launches; with the "
                                 "working set in the 256 MiB Infinity Cache this is a fabric / cache roofline, not a DRAM one -- `cold` is"},
        }
        if cold is not None:
            out["roofline"]["cold"] = cold
        if entry_point_ms is not None:
            out["entry_point_ms"] = entry_point_ms
            out["entry_point_host"] = "cxx" if F.host_fast_path() else "python"
            out["module_entry_ms"] = module_entry_ms
        if routes:
            out["config"]["route_ms_per_step"] = routes
        if other is not None:
            out["config"]["other_scaling_mode"] = other
        if os.environ.get("TRL_BENCH_REHEARSAL"):
            out["config"]["rehearsal"] = os.environ["TRL_BENCH_REHEARSAL"] + f" over {backend}: control-flow rehearsal, not a scaling number"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(*(t.cpu() for t in (r, v, nv, done, term)))
        print(json.dumps(out))
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
