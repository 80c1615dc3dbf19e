// Cross-GPU exchange of the advantage moments {N, sum, sumsq}: the one collective of the path
// (reference semantics: adv.mean() / adv.std() over ALL elements,
// torchrl/objectives/value/advantages.py:1459-1463, when the env dimension is sharded over GPUs).
//
// 24 bytes per rank is far below the size where a ring or tree pays: every rank WRITES its three
// moments straight into a mailbox in each peer's HBM over xGMI (point-to-point stores, one hop)
// and then polls its OWN mailbox (local memory) until every rank's line carries this call's epoch.
// One 64-thread block, no fences, no atomics RMW, no host involvement:
//   * mailbox = fine-grained device memory (hipExtMallocWithFlags), shared with the peers through
//     hipIpc handles; one 64-byte line per (parity, source rank);
//   * a value travels as two 8-byte granules {32 value bits | 32-bit epoch}: each granule is one
//     atomic store, so a reader can never combine halves of different calls, and the value arrives
//     bit-exact;
//   * lines are double-buffered by epoch parity: a rank can be at most one call ahead of a peer
//     (it cannot finish call n+1 before that peer has sent n+1, i.e. after the peer finished reading n);
//   * the epoch lives in the local mailbox header and is advanced by the kernel itself, so the call
//     can be captured in a hipGraph;
//   * the totals are summed in rank order by every rank: all ranks get bit-identical results.
// Every spin is bounded (timeout in the comm, default 20 s -- ranks may be far apart in host time):
// on timeout the error word is set and NaNs are written.
#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/trl_value.h"
#include "common.h"
#include "xchg.h"

namespace {

using xchg::kMaxWorld;
typedef xchg::Mailbox XchgMailbox;

struct XchgComm {
    int rank, world, device;
    XchgMailbox *local;
    void *opened[kMaxWorld];         // peers' mailboxes as mapped by hipIpcOpenMemHandle (nullptr for self)
    XchgMailbox **peers_dev;         // device array [world]
    unsigned long long timeout_ticks;  // 100 MHz wall clock
};

__global__ __launch_bounds__(64) void xchg_allreduce_kernel(xchg::Dev x, double *stats) {
    const int lane = threadIdx.x;
    const unsigned epoch = x.local->epoch + 1u;  // written by the previous call: visible across the kernel boundary
    xchg::send(x, epoch, lane, stats[0], stats[1], stats[2]);
    double n, s, ss;
    const bool all_ok = xchg::recv_sum(x, epoch, lane, n, s, ss);
    if (lane == 0) {
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        stats[0] = all_ok ? n : nan;
        stats[1] = all_ok ? s : nan;
        stats[2] = all_ok ? ss : nan;
        if (!all_ok) x.local->error = 1u;
        x.local->epoch = epoch;
    }
}

}  // namespace

int trl_internal_fail(int code, const char *fmt, ...);  // capi.hip

xchg::Dev xchg_dev_of(void *comm) {
    if (!comm) return xchg::Dev{nullptr, nullptr, 0, 1, 0};
    XchgComm *c = (XchgComm *)comm;
    return xchg::Dev{c->local, c->peers_dev, c->rank, c->world, c->timeout_ticks};
}

extern "C" {

size_t trl_xchg_handle_bytes(void) { return sizeof(hipIpcMemHandle_t); }

int trl_xchg_local_init(void **mailbox, void *ipc_handle_out) {
    if (!mailbox || !ipc_handle_out) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "trl_xchg_local_init: null argument");
    void *p = nullptr;
    int dev = 0;
    hipError_t e = hipInit(0);  // this may be the first HIP call of the process
    if (e == hipSuccess) e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipSetDevice(dev);
    if (e != hipSuccess) return trl_internal_fail(TRL_ERR_LAUNCH, "trl_xchg_local_init: HIP runtime: %s", hipGetErrorString(e));
    e = hipExtMallocWithFlags(&p, sizeof(XchgMailbox), hipDeviceMallocFinegrained);
    if (e != hipSuccess) return trl_internal_fail(TRL_ERR_LAUNCH, "trl_xchg_local_init: hipExtMallocWithFlags: %s", hipGetErrorString(e));
    e = hipMemset(p, 0, sizeof(XchgMailbox));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return trl_internal_fail(TRL_ERR_LAUNCH, "trl_xchg_local_init: %s", hipGetErrorString(e));
    }
    memcpy(ipc_handle_out, &h, sizeof(h));
    *mailbox = p;
    return TRL_OK;
}

int trl_xchg_open(void *mailbox, const void *ipc_handles, const int *peer_devices, int rank, int world, double timeout_s,
                  void **comm_out) {
    if (!mailbox || !ipc_handles || !comm_out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world)
        return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "trl_xchg_open: bad argument (world <= %d)", kMaxWorld);
    XchgComm *c = new XchgComm();
    c->rank = rank;
    c->world = world;
    c->local = (XchgMailbox *)mailbox;
    c->timeout_ticks = (unsigned long long)((timeout_s > 0 ? timeout_s : 20.0) * 1e8);
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return trl_internal_fail(TRL_ERR_LAUNCH, "trl_xchg_open: no device"); }
    XchgMailbox *table[kMaxWorld];
    for (int p = 0; p < world; ++p) {
        c->opened[p] = nullptr;
        if (p == rank) { table[p] = c->local; continue; }
        if (peer_devices && peer_devices[p] >= 0 && peer_devices[p] != c->device) {
            hipError_t pe = hipDeviceEnablePeerAccess(peer_devices[p], 0);  // already-enabled is fine
            (void)pe;
            (void)hipGetLastError();
        }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)ipc_handles + (size_t)p * sizeof(h), sizeof(h));
        hipError_t e = hipIpcOpenMemHandle(&c->opened[p], h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int q = 0; q < p; ++q) if (c->opened[q]) (void)hipIpcCloseMemHandle(c->opened[q]);
            delete c;
            return trl_internal_fail(TRL_ERR_LAUNCH, "trl_xchg_open: hipIpcOpenMemHandle(rank %d): %s", p, hipGetErrorString(e));
        }
        table[p] = (XchgMailbox *)c->opened[p];
    }
    hipError_t e = hipMalloc((void **)&c->peers_dev, sizeof(XchgMailbox *) * kMaxWorld);
    if (e == hipSuccess) e = hipMemcpy(c->peers_dev, table, sizeof(XchgMailbox *) * world, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        for (int q = 0; q < world; ++q) if (c->opened[q]) (void)hipIpcCloseMemHandle(c->opened[q]);
        delete c;
        return trl_internal_fail(TRL_ERR_LAUNCH, "trl_xchg_open: %s", hipGetErrorString(e));
    }
    *comm_out = c;
    return TRL_OK;
}

int trl_xchg_allreduce_stats(void *stream, void *comm, double *stats) {
    if (!comm || !stats) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "trl_xchg_allreduce_stats: null argument");
    hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, xchg_dev_of(comm), stats);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return trl_internal_fail(TRL_ERR_LAUNCH, "trl_xchg_allreduce_stats: %s", hipGetErrorString(e));
    return TRL_OK;
}

int trl_xchg_set_timeout(void *comm, double timeout_s) {
    if (!comm || !(timeout_s > 0)) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "trl_xchg_set_timeout: bad argument");
    ((XchgComm *)comm)->timeout_ticks = (unsigned long long)(timeout_s * 1e8);
    return TRL_OK;
}

int trl_xchg_error(void *comm) {
    if (!comm) return -1;
    XchgComm *c = (XchgComm *)comm;
    unsigned err = 0;
    if (hipMemcpy(&err, &c->local->error, sizeof(err), hipMemcpyDeviceToHost) != hipSuccess) return -1;  // synchronises
    return (int)err;
}

int trl_xchg_destroy(void *comm, void *mailbox) {
    XchgComm *c = (XchgComm *)comm;
    if (c) {
        for (int q = 0; q < c->world; ++q) if (c->opened[q]) (void)hipIpcCloseMemHandle(c->opened[q]);
        if (c->peers_dev) (void)hipFree(c->peers_dev);
        delete c;
    }
    if (mailbox) (void)hipFree(mailbox);
    return TRL_OK;
}

}  // extern "C"
