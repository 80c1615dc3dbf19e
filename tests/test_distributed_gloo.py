"""CPU, world size 2 over gloo: the multi-GPU protocol of the hot path.  Each rank owns a shard of the
env dimension; the only exchange is one all-reduce of the fp64 moments {N, sum, sumsq}; every rank then
normalises with the global statistics.  The kernels themselves need a GPU, so here the two device calls
inside `functional.gae_normalized` are replaced by the CPU oracle (test-only stand-ins): what is tested is
the host control flow, the collective, and that sharded == single-process."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    g = torch.Generator().manual_seed(0)
    shape = (64, 32, 1)
    r, v, nv = (torch.randn(shape, generator=g) for _ in range(3))
    term = torch.rand(shape, generator=g) < 0.05
    done = term | (torch.rand(shape, generator=g) < 0.05)
    return r, v, nv, done, term


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import value_oracle as O
        from torchrl_amd.objectives.value import functional as F

        def fake_launch_gae(call, want_stats=False, algo=0, out=None):  # stand-in for trl_gae_fwd (+ stats)
            assert call.gamma == float(torch.tensor(0.99)) and call.lg == float(torch.tensor(0.99 * 0.95))
            adv, vt = O.gae(0.99, 0.95, call.value, call.next_value, call.reward, call.done.bool(), call.terminated.bool())
            a64 = adv.double()
            return adv, vt, torch.stack([torch.tensor(float(a64.numel()), dtype=torch.float64), a64.sum(), (a64 * a64).sum()])

        def fake_normalize(x, stats, eps, out=None):  # stand-in for trl_normalize
            mean, std = F.stats_to_mean_std(stats, x.dtype, eps)
            return (x - torch.tensor(mean, dtype=x.dtype)) / torch.tensor(std, dtype=x.dtype)

        F._launch_gae, F.normalize_ = fake_launch_gae, fake_normalize
        r, v, nv, done, term = _inputs()
        lo, hi = rank * 32, (rank + 1) * 32  # env dimension sharded
        adv, vt = F.gae_normalized(0.99, 0.95, v[lo:hi], nv[lo:hi], r[lo:hi], done[lo:hi], term[lo:hi], eps=1e-4, distributed=True)
        q.put((rank, adv.numpy(), vt.numpy()))  # plain arrays: no fd passing that would race with the worker's exit
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_normalisation_equals_single_process():
    from oracle import value_oracle as O

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, q)) for rk in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        rk, adv, vt = q.get(timeout=90)
        got[rk] = (torch.from_numpy(adv), torch.from_numpy(vt))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    r, v, nv, done, term = _inputs()
    oadv, ovt = O.gae(0.99, 0.95, v, nv, r, done, term)
    exp = O.normalize(oadv, 1e-4)
    adv = torch.cat([got[0][0], got[1][0]], 0)
    vt = torch.cat([got[0][1], got[1][1]], 0)
    torch.testing.assert_close(adv, exp, rtol=1e-6, atol=1e-6)
    assert torch.equal(vt, ovt)
    # a rank normalising with its LOCAL statistics would differ: the collective matters
    local = O.normalize(oadv[:32], 1e-4)
    assert (local - adv[:32]).abs().max() > 1e-4


def test_stats_to_mean_std_matches_torch():
    from torchrl_amd.objectives.value import functional as F

    x = torch.randn(1000) * 3 + 0.5
    st = torch.stack([torch.tensor(1000.0, dtype=torch.float64), x.double().sum(), (x.double() ** 2).sum()])
    mean, std = F.stats_to_mean_std(st, torch.float32, 1e-4)
    assert abs(mean - float(x.mean())) < 1e-6 and abs(std - float(x.std())) < 1e-6
    assert F.stats_to_mean_std(torch.tensor([5.0, 0.0, 0.0], dtype=torch.float64), torch.float32, 1e-4)[1] == float(torch.tensor(1e-4))
