// Row-sequential kernels ("SEQ" family): one lane owns one (o, i) row of the canonical
// (O, T, I) view and walks time from T-1 down to 0 in registers, in exactly the operation
// order of the reference's Python loops, so the result is BIT-IDENTICAL to the reference
// (reference: torchrl/objectives/value/functional.py:165-180 GAE, :841-887 TD(lambda),
// :1319-1348 V-trace).  Lanes are laid out along i (fastest) then o, so loads coalesce when
// I >= 64; for small I the SCAN family (scan_kernels.hip) is the bandwidth path and this
// family is the exact fallback for every shape, dtype and coefficient form.
//
// HBM traffic: every input element is read once and every output written once
// (GAE fp32: 14 B read + 8 B written per transition); no intermediate is materialised.
#include "seq_rows.h"

namespace {

constexpr int kSeqThreads = 64;

template <typename F>
LaunchResult dispatch_types(int vdt, int cdt, F &&f) {
    if (vdt == 0 && cdt == 0) return f(float{}, float{});
    if (vdt == 1 && cdt == 1) return f(double{}, double{});
    if (vdt == 1 && cdt == 0) return f(double{}, float{});
    if (vdt == 0 && cdt == 1) return f(float{}, double{});
    return LaunchResult{1, 0, nullptr};
}

// The per-row loops live in seq_rows.h (shared with the host entry points of cpu_host.hip); a kernel maps a lane to its row.
template <typename TS, typename TG>
__global__ __launch_bounds__(kSeqThreads) void gae_seq_kernel(GaeArgs a) {
    const i64 idx = (i64)blockIdx.x * kSeqThreads + threadIdx.x;
    double s = 0.0, ss = 0.0;
    if (idx < a.O * a.I) {
        const i64 o = idx / a.I;
        seqrow::gae_row<TS, TG>(a, o, idx - o * a.I, s, ss);
    }
    if (a.partials) {
        block_sum2<kSeqThreads>(s, ss, nullptr);
        if (threadIdx.x == 0) {
            a.partials[2 * (i64)blockIdx.x] = s;
            a.partials[2 * (i64)blockIdx.x + 1] = ss;
        }
    }
}

template <typename TS, typename TG>
__global__ __launch_bounds__(kSeqThreads) void td_lambda_seq_kernel(TdLambdaArgs a) {
    const i64 idx = (i64)blockIdx.x * kSeqThreads + threadIdx.x;
    if (idx >= a.O * a.I) return;
    const i64 o = idx / a.I;
    seqrow::td_lambda_row<TS, TG>(a, o, idx - o * a.I);
}

// rolling_gamma=False: O(T^2) work; one lane per (o, k, i)
template <typename TS, typename TG>
__global__ __launch_bounds__(kSeqThreads) void td_lambda_fixed_seq_kernel(TdLambdaArgs a) {
    const i64 idx = (i64)blockIdx.x * kSeqThreads + threadIdx.x;
    if (idx >= a.O * a.T * a.I) return;
    const i64 o = idx / (a.T * a.I);
    const i64 rem = idx - o * a.T * a.I;
    const i64 k = rem / a.I;
    seqrow::td_lambda_fixed_elem<TS, TG>(a, o, k, rem - k * a.I);
}

template <typename TS, typename TG, bool FIXED>
__global__ __launch_bounds__(kSeqThreads) void td1_seq_kernel(TdLambdaArgs a) {
    const i64 idx = (i64)blockIdx.x * kSeqThreads + threadIdx.x;
    i64 o, i, k = 0;
    if (FIXED) {
        if (idx >= a.O * a.T * a.I) return;
        o = idx / (a.T * a.I);
        const i64 rem = idx - o * a.T * a.I;
        k = rem / a.I;
        i = rem - k * a.I;
    } else {
        if (idx >= a.O * a.I) return;
        o = idx / a.I;
        i = idx - o * a.I;
    }
    seqrow::td1_row<TS, TG, FIXED>(a, o, k, i);
}

template <typename TS, typename TG>
__global__ __launch_bounds__(kSeqThreads) void vtrace_seq_kernel(VtraceArgs a) {
    const i64 idx = (i64)blockIdx.x * kSeqThreads + threadIdx.x;
    if (idx >= a.O * a.I) return;
    const i64 o = idx / a.I;
    seqrow::vtrace_row<TS, TG>(a, o, idx - o * a.I);
}

inline bool grid_ok(i64 nblocks) { return nblocks > 0 && nblocks <= 0x7fffffffLL; }

}  // namespace

i64 seq_max_partials(i64 O, i64 T, i64 I) {
    (void)T;
    return (O * I + kSeqThreads - 1) / kSeqThreads;
}

LaunchResult launch_gae_seq(hipStream_t s, int vdt, int cdt, const GaeArgs &a) {
    const i64 nb = (a.O * a.I + kSeqThreads - 1) / kSeqThreads;
    if (!grid_ok(nb)) return LaunchResult{1, 0, nullptr};
    return dispatch_types(vdt, cdt, [&](auto ts, auto tg) {
        using TS = decltype(ts);
        using TG = decltype(tg);
        hipLaunchKernelGGL((gae_seq_kernel<TS, TG>), dim3((unsigned)nb), dim3(kSeqThreads), 0, s, a);
        return LaunchResult{0, nb, "gae_seq"};
    });
}

LaunchResult launch_td_lambda_seq(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, int rolling) {
    const i64 n = rolling ? a.O * a.I : a.O * a.T * a.I;
    const i64 nb = (n + kSeqThreads - 1) / kSeqThreads;
    if (!grid_ok(nb)) return LaunchResult{1, 0, nullptr};
    return dispatch_types(vdt, cdt, [&](auto ts, auto tg) {
        using TS = decltype(ts);
        using TG = decltype(tg);
        if (rolling) {
            hipLaunchKernelGGL((td_lambda_seq_kernel<TS, TG>), dim3((unsigned)nb), dim3(kSeqThreads), 0, s, a);
            return LaunchResult{0, 0, "td_lambda_seq"};
        }
        hipLaunchKernelGGL((td_lambda_fixed_seq_kernel<TS, TG>), dim3((unsigned)nb), dim3(kSeqThreads), 0, s, a);
        return LaunchResult{0, 0, "td_lambda_fixed_seq"};
    });
}

LaunchResult launch_td1_seq(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, int rolling) {
    const i64 n = rolling ? a.O * a.I : a.O * a.T * a.I;
    const i64 nb = (n + kSeqThreads - 1) / kSeqThreads;
    if (!grid_ok(nb)) return LaunchResult{1, 0, nullptr};
    return dispatch_types(vdt, cdt, [&](auto ts, auto tg) {
        using TS = decltype(ts);
        using TG = decltype(tg);
        if (rolling) {
            hipLaunchKernelGGL((td1_seq_kernel<TS, TG, false>), dim3((unsigned)nb), dim3(kSeqThreads), 0, s, a);
            return LaunchResult{0, 0, "td1_seq"};
        }
        hipLaunchKernelGGL((td1_seq_kernel<TS, TG, true>), dim3((unsigned)nb), dim3(kSeqThreads), 0, s, a);
        return LaunchResult{0, 0, "td1_fixed_seq"};
    });
}

LaunchResult launch_vtrace_seq(hipStream_t s, int vdt, int cdt, const VtraceArgs &a) {
    const i64 nb = (a.O * a.I + kSeqThreads - 1) / kSeqThreads;
    if (!grid_ok(nb)) return LaunchResult{1, 0, nullptr};
    return dispatch_types(vdt, cdt, [&](auto ts, auto tg) {
        using TS = decltype(ts);
        using TG = decltype(tg);
        hipLaunchKernelGGL((vtrace_seq_kernel<TS, TG>), dim3((unsigned)nb), dim3(kSeqThreads), 0, s, a);
        return LaunchResult{0, 0, "vtrace_seq"};
    });
}
