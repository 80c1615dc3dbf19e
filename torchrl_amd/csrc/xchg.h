// Device-side layout and helpers of the cross-GPU moment exchange (see xchg_kernels.hip for the protocol).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace xchg {

constexpr int kMaxWorld = 64;
constexpr int kLineWords = 8;  // 64-byte line: 6 granules used
struct Mailbox {
    unsigned epoch;
    unsigned error;
    unsigned pad[62];
    unsigned long long lines[2][kMaxWorld][kLineWords];
};

// what a kernel needs of a communicator; local == nullptr: single GPU, no exchange
struct Dev {
    Mailbox *local;
    Mailbox *const *peers;
    int rank, world;
    unsigned long long timeout_ticks;  // 100 MHz wall clock
};

__device__ __forceinline__ void sys_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long sys_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// lane p < world stores this rank's {N, sum, sumsq} into rank p's mailbox (line [parity][rank]).
// A value travels as two granules {32 value bits | 32-bit epoch}; N as an integer (exact).
__device__ __forceinline__ void send(const Dev &x, unsigned epoch, int lane, double n, double s, double ss) {
    if (lane < x.world) {
        unsigned long long bits[3];
        bits[0] = (unsigned long long)(long long)n;
        bits[1] = (unsigned long long)__double_as_longlong(s);
        bits[2] = (unsigned long long)__double_as_longlong(ss);
        unsigned long long *dst = x.peers[lane]->lines[epoch & 1u][x.rank];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            sys_store(dst + 2 * g, (bits[g] << 32) | epoch);
            sys_store(dst + 2 * g + 1, (bits[g] & 0xffffffff00000000ull) | epoch);
        }
    }
}

// Whole wave: lane p < world polls line [parity][p] of the local mailbox until every line carries
// `epoch` (bounded), then the totals are formed in rank order (bit-identical on every rank and in
// every wave that calls this).  Returns false on timeout.
__device__ __forceinline__ bool recv_sum(const Dev &x, unsigned epoch, int lane, double &n, double &s, double &ss) {
    unsigned long long got[6];
    bool ok = lane >= x.world;
    const unsigned long long t_start = wall_clock64();
    for (;;) {
        if (!ok) {
            const unsigned long long *src = x.local->lines[epoch & 1u][lane];
            ok = true;
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                got[g] = sys_load(src + g);
                ok = ok && ((unsigned)(got[g] & 0xffffffffull) == epoch);
            }
        }
        if (__all(ok)) break;
        if (wall_clock64() - t_start > x.timeout_ticks) break;
        __builtin_amdgcn_s_sleep(2);
    }
    const bool all_ok = __all(ok);
    n = s = ss = 0.0;
    if (lane < x.world && ok) {
        n = (double)(long long)((got[0] >> 32) | (got[1] & 0xffffffff00000000ull));
        s = __longlong_as_double((long long)((got[2] >> 32) | (got[3] & 0xffffffff00000000ull)));
        ss = __longlong_as_double((long long)((got[4] >> 32) | (got[5] & 0xffffffff00000000ull)));
    }
    n = wave_sum(n);
    s = wave_sum(s);
    ss = wave_sum(ss);
    return all_ok;
}

}  // namespace xchg

// host side (xchg_kernels.hip): device view of a communicator made by trl_xchg_open; comm == nullptr gives {nullptr...}
xchg::Dev xchg_dev_of(void *comm);
