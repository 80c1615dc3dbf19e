"""Cross-GPU sum of the advantage moments ``{N, sum, sumsq}`` (include/trl_value.h: trl_xchg_*).

The reference normalises with ``adv.mean()`` / ``adv.std()`` over all elements
(torchrl/objectives/value/advantages.py:1459-1463); with the env dimension sharded over GPUs that
is an all-reduce of three doubles.  ``torch.distributed.all_reduce`` (RCCL) is the portable route
(``GAE(process_group=...)`` uses it by default); :class:`StatsExchange` is the intra-node
low-latency route: peer-to-peer mailbox stores over xGMI, one tiny kernel, no host work beyond one
ctypes call.  It is validated against the process group's own all-reduce when it is created and
refuses to be used (``StatsExchange.create`` returns ``None``) if anything disagrees.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from ... import _extension

__all__ = ["StatsExchange"]


class StatsExchange:
    """All ranks of ``group`` (one process per GPU, one node) must create it together."""

    def __init__(self, group=None, device: Optional[torch.device] = None, timeout_s: float = 20.0):
        import torch.distributed as td

        self._lib = lib = _extension.get_lib()
        self._comm = self._mailbox = None
        torch.cuda.init()  # the device context must exist before the library's first HIP call
        self.group = group
        self.rank, self.world = td.get_rank(group), td.get_world_size(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        hb = lib.trl_xchg_handle_bytes()
        handle = ctypes.create_string_buffer(hb)
        mailbox = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _extension.check(lib.trl_xchg_local_init(ctypes.byref(mailbox), handle), lib)
        self._mailbox = mailbox
        mine = (bytes(handle.raw), int(self.device.index))
        everyone = [None] * self.world
        td.all_gather_object(everyone, mine, group=group)
        blob = b"".join(h for h, _ in everyone)
        devs = (ctypes.c_int * self.world)(*[d for _, d in everyone])
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = lib.trl_xchg_open(mailbox, blob, devs, self.rank, self.world, float(timeout_s), ctypes.byref(comm))
        # every rank must learn whether every rank opened its peers before anybody sends
        flags = [None] * self.world
        td.all_gather_object(flags, int(rc), group=group)
        if any(flags):
            if rc == 0:
                lib.trl_xchg_destroy(comm, mailbox)
                self._mailbox = None
                raise RuntimeError("StatsExchange: a peer rank could not map the mailboxes")
            lib.trl_xchg_destroy(None, mailbox)
            self._mailbox = None
            _extension.check(rc, lib)
        self._comm = comm

    @classmethod
    def create(cls, group=None, device=None, timeout_s: float = 20.0, verify: bool = True) -> Optional["StatsExchange"]:
        """Build and (``verify``) check the exchange against ``all_reduce`` of the same process group on
        three probe calls; returns ``None`` on every rank if any rank saw a failure."""
        import torch.distributed as td

        try:
            ex = cls(group, device, timeout_s=timeout_s)
            good = True
        except Exception as e:  # noqa: BLE001 -- every rank must reach the vote below
            import warnings

            warnings.warn(f"StatsExchange could not be set up: {type(e).__name__}: {e}")
            ex, good = None, False
        votes = [None] * td.get_world_size(group)
        td.all_gather_object(votes, good, group=group)
        if not all(votes):
            if ex is not None:
                ex.close()
            return None
        if verify:
            good = ex._self_test()
            td.all_gather_object(votes, good, group=group)
            if not all(votes):
                ex.close()
                return None
        return ex

    def _self_test(self) -> bool:
        import torch.distributed as td

        good = True
        for k in range(3):  # both parities of the double-buffered lines, and a re-use
            probe = torch.tensor([1000.0 + self.rank + k, (self.rank + 1) * 0.3711 + k, (self.rank + 2) ** 2 * 1.0e-3 + k],
                                 dtype=torch.float64, device=self.device)
            want = probe.clone()
            self.all_reduce_(probe)
            backend = td.get_backend(self.group)
            if backend == "gloo":
                w = want.cpu()
                td.all_reduce(w, group=self.group)
                want = w.to(self.device)
            else:
                td.all_reduce(want, group=self.group)
            torch.cuda.synchronize(self.device)
            # rank-ordered fp64 sums: equal to the collective's up to the last bits of a differently-ordered sum
            good = good and bool(torch.allclose(probe, want, rtol=1e-13, atol=0.0)) and probe[0].item() == want[0].item()
        return good and self._lib.trl_xchg_error(self._comm) == 0

    @property
    def comm(self) -> ctypes.c_void_p:
        """The C-ABI communicator (``void *comm`` of trl_xchg_* / trl_gae_normalize_xchg_fwd)."""
        if self._comm is None:
            raise RuntimeError("StatsExchange is closed")
        return self._comm

    def set_timeout(self, seconds: float):
        _extension.check(self._lib.trl_xchg_set_timeout(self.comm, float(seconds)), self._lib)

    def all_reduce_(self, stats: torch.Tensor) -> torch.Tensor:
        """In-place sum over ranks of a device ``float64[3]`` ``{N, sum, sumsq}`` on the current stream."""
        if self._comm is None:
            raise RuntimeError("StatsExchange is closed")
        if stats.dtype != torch.float64 or stats.numel() != 3 or not stats.is_contiguous() or stats.device != self.device:
            raise ValueError(f"stats must be a contiguous float64[3] tensor on {self.device}")
        rc = self._lib.trl_xchg_allreduce_stats(
            ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), self._comm, ctypes.c_void_p(stats.data_ptr()))
        _extension.check(rc, self._lib)
        return stats

    def prepared_call(self, stats: torch.Tensor):
        """``(fn, args)`` for :class:`functional.PreparedStep` (bound to the current stream)."""
        return (self._lib.trl_xchg_allreduce_stats,
                (ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream), self._comm, ctypes.c_void_p(stats.data_ptr())))

    def error(self) -> bool:
        """True if any call timed out waiting for a peer (synchronises the device)."""
        return self._comm is not None and self._lib.trl_xchg_error(self._comm) != 0

    def close(self):
        if self._comm is not None or self._mailbox is not None:
            torch.cuda.synchronize(self.device)
            self._lib.trl_xchg_destroy(self._comm, self._mailbox)
            self._comm = self._mailbox = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def _probe_main(store_path: str) -> int:
    """Crash-isolated check of the exchange on the live topology (run as a child process per rank,
    ``python -m torchrl_amd.objectives.value.stats_exchange <file-store path>`` with RANK / WORLD_SIZE /
    LOCAL_RANK in the environment): builds a gloo group over a file store, creates the exchange
    (self-tested against gloo's all-reduce) and runs a small sharded GAE + normalise through the
    in-launch cross-GPU stage against the three-launch route.  Exit code 0 = every rank passed.  A
    caller that must not die with a misbehaving interconnect (bench.py) looks at the exit code only."""
    import os

    import torch.distributed as td

    from . import functional as F

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.zeros(1, device=dev)  # device context up before anything else
    store = td.FileStore(store_path, world)
    import datetime

    td.init_process_group("gloo", store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=45))
    good = False
    try:
        ex = StatsExchange.create(device=dev, timeout_s=5.0)
        if ex is not None:
            g = torch.Generator().manual_seed(rank)
            shape = (4096, 32, 1)
            r, v, nv = (torch.randn(shape, generator=g).to(dev) for _ in range(3))
            done = (torch.rand(shape, generator=g) < 0.02).to(dev)
            call = F._prepare_gae(0.99, 0.95, v, nv, r, done, done, -2)
            adv0, _, st = F._launch_gae(call, want_stats=True)
            h = st.cpu()
            td.all_reduce(h)
            ref = F.normalize_(adv0.clone(), h.to(dev), 1e-4)
            good = True
            for _ in range(3):
                adv, _, _ = F._launch_gae_normalize(call, 1e-4, exchange=ex)
                good = good and bool(torch.allclose(adv, ref, rtol=2e-6, atol=2e-6))
            err = ex.error()
            if not good or err:
                print(f"exchange probe rank {rank}: mismatch (max abs diff {float((adv - ref).abs().max())}, timeout {err})", flush=True)
            good = good and not err
            ex.close()
        else:
            print(f"exchange probe rank {rank}: StatsExchange.create() declined", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"exchange probe rank {rank}: {type(e).__name__}: {e}", flush=True)
        good = False
    votes = [None] * world
    td.all_gather_object(votes, good)
    td.destroy_process_group()
    return 0 if all(votes) else 3


if __name__ == "__main__":
    import sys

    sys.exit(_probe_main(sys.argv[1]))
