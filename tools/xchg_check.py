"""Multi-rank check of StatsExchange (peer-to-peer mailbox all-reduce of {N, sum, sumsq}).
Rehearsal on a 1-GPU box (ranks share cuda:0, gloo carries the handles and the reference sums):
    TRL_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 \
        --master-addr 127.0.0.1 --master-port 29517 tools/xchg_check.py
"""
import os
import sys
import time

import torch
import torch.distributed as td

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchrl_amd.objectives.value.stats_exchange import StatsExchange  # noqa: E402

backend = os.environ.get("TRL_BENCH_BACKEND", "nccl")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if backend == "nccl":
    td.init_process_group("nccl", device_id=dev)
else:
    td.init_process_group(backend)

ex = StatsExchange.create()
assert ex is not None, "StatsExchange.create() failed its self-test"
g = torch.Generator().manual_seed(100 + rank)
bad = 0
for k in range(200):
    st = torch.cat([torch.tensor([float(1000 + 7 * rank + k)], dtype=torch.float64), torch.randn(2, generator=g, dtype=torch.float64) * 10 ** (k % 7)])
    mine = st.to(dev)
    ex.all_reduce_(mine)
    ref = st.clone()
    if backend == "gloo":
        td.all_reduce(ref)
    else:
        r = ref.to(dev)
        td.all_reduce(r)
        ref = r.cpu()
    out = mine.cpu()
    mag = st.abs()  # a differently-ordered fp64 sum may differ by ~1e-16 of the summed magnitudes
    td.all_reduce(mag) if backend == "gloo" else None
    scale = float(mag.max()) * world if backend != "gloo" else float(mag.max())
    if not (out[0] == ref[0] and float((out - ref).abs().max()) <= 1e-14 * scale):
        bad += 1
# all ranks bit-identical?
mine = torch.tensor([3.0, 0.1 * (rank + 1), 0.7 / (rank + 1)], dtype=torch.float64, device=dev)
ex.all_reduce_(mine)
allv = [None] * world
td.all_gather_object(allv, mine.cpu().tolist())
same = all(v == allv[0] for v in allv)
# latency: back-to-back calls on one buffer
buf = torch.zeros(3, dtype=torch.float64, device=dev)  # zeros stay zeros under repeated summing
td.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 2000
for _ in range(n):
    ex.all_reduce_(buf)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e6
# GAE + normalise on a batch sharded over the ranks: single launch with the in-launch cross-GPU stage
# (fused) and the three-launch fall-back, against GAE(stats) -> process-group all-reduce -> normalise
from torchrl_amd.objectives.value import functional as F  # noqa: E402

gae_bad = []
for shape, dtype in [((8192 + 64 * rank, 32, 1), torch.float32), ((1024, 32, 1), torch.float64), ((300, 1500, 1), torch.float32), ((64, 20, 3), torch.float32)]:
    gg = torch.Generator().manual_seed(7 + rank)
    r, v, nv = (torch.randn(shape, generator=gg).to(dtype).to(dev) for _ in range(3))
    term = (torch.rand(shape, generator=gg) < 0.02).to(dev)
    done = term | (torch.rand(shape, generator=gg) < 0.02).to(dev)
    call = F._prepare_gae(0.99, 0.95, v, nv, r, done, term, -2)
    adv0, vt0, st = F._launch_gae(call, want_stats=True)
    if backend == "gloo":
        h = st.cpu()
        td.all_reduce(h)
        st = h.to(dev)
    else:
        td.all_reduce(st)
    ref = F.normalize_(adv0.clone(), st, 1e-4)
    for rep in range(3):
        adv, vt, gst = F._launch_gae_normalize(call, 1e-4, want_stats=True, exchange=ex)
        kern = F.last_kernel()
        tol = dict(rtol=2e-6, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-12, atol=1e-12)
        ok = torch.allclose(adv, ref, **tol) and torch.equal(vt, vt0) and torch.allclose(gst, st, rtol=1e-12, atol=1e-9) and gst[0].item() == st[0].item()
        if not ok:
            gae_bad.append((shape, str(dtype), kern, rep, float((adv - ref).abs().max())))
    if rank == 0:
        print("gae+normalise sharded", shape, dtype, "->", kern, flush=True)
bad += len(gae_bad)
if gae_bad:
    print("rank", rank, "GAE mismatches:", gae_bad, flush=True)
err = ex.error()
res = [None] * world
td.all_gather_object(res, (bad, same, err, dt))
if rank == 0:
    print("per-rank (mismatches, identical, timeout_error, us/call):", res, flush=True)
    ok = all(r[0] == 0 and r[1] and not r[2] for r in res)
    print("XCHG_CHECK", "PASS" if ok else "FAIL", flush=True)
ex.close()
td.destroy_process_group()
