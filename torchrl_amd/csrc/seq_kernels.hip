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
#include "common.h"

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

// ---------------------------------------------------------------------------------------------
template <typename TS, typename TG>
__global__ __launch_bounds__(kSeqThreads) void gae_seq_kernel(GaeArgs a) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ r = (const TS *)a.reward;
    const TS *v = (const TS *)a.value;
    const TS *nv = (const TS *)a.next_value;
    const u8 *__restrict__ d = a.done;
    const u8 *__restrict__ tm = a.terminated;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    const TG *__restrict__ lgt = (const TG *)a.lg_t;
    TS *__restrict__ adv = (TS *)a.advantage;
    TS *__restrict__ vt = (TS *)a.value_target;
    const TG gs = (TG)a.gamma, ls = (TG)a.lmbda, lgs = (TG)a.lg;

    const i64 idx = (i64)blockIdx.x * kSeqThreads + threadIdx.x;
    double s = 0.0, ss = 0.0;
    if (idx < a.O * a.I) {
        const i64 o = idx / a.I, i = idx - o * a.I;
        if (!nv) {  // shifted layout: `value` is the (O, T + 1, I) buffer, the next step's value follows I elements on
            v += o * a.I;
            nv = v + a.I;
        }
        i64 j = (o * a.T + (a.T - 1)) * a.I + i;
        // masks broadcast over the feature dims are read in place: element (o, t) for every i
        const i64 ms = a.mask_bcast ? 1 : a.I;
        i64 mj = a.mask_bcast ? o * a.T + (a.T - 1) : j;
        TC prev = (TC)0;
#pragma unroll 4
        for (i64 t = a.T - 1; t >= 0; --t, j -= a.I, mj -= ms) {
            const TG g = gt ? gt[j] : gs;
            const TG c = lgt ? lgt[j] : (gt ? ls * g : lgs);
            const TG g_nt = g * (TG)(tm[mj] ? 0 : 1);  // multiply, not select: inf*0 -> NaN as in the reference
            const TG disc = c * (TG)(d[mj] ? 0 : 1);
            const TC delta = ((TC)r[j] + (TC)g_nt * (TC)nv[j]) - (TC)v[j];
            prev = delta + prev * (TC)disc;
            const TS as = (TS)prev;
            adv[j] = as;
            vt[j] = as + v[j];
            s += (double)as;
            ss += (double)as * (double)as;
        }
    }
    if (a.partials) {
        block_sum2<kSeqThreads>(s, ss, nullptr);
        if (threadIdx.x == 0) {
            a.partials[2 * (i64)blockIdx.x] = s;
            a.partials[2 * (i64)blockIdx.x + 1] = ss;
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <typename TS, typename TG>
__global__ __launch_bounds__(kSeqThreads) void td_lambda_seq_kernel(TdLambdaArgs a) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ r = (const TS *)a.reward;
    const TS *__restrict__ nv = (const TS *)a.next_value;
    const u8 *__restrict__ d = a.done;
    const u8 *__restrict__ tm = a.terminated;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    const TG *__restrict__ lt = (const TG *)a.lmbda_t;
    TS *__restrict__ ret = (TS *)a.returns;
    const TG gs = (TG)a.gamma, ls = (TG)a.lmbda;

    const i64 idx = (i64)blockIdx.x * kSeqThreads + threadIdx.x;
    if (idx >= a.O * a.I) return;
    const i64 o = idx / a.I, i = idx - o * a.I;
    i64 j = (o * a.T + (a.T - 1)) * a.I + i;
    const i64 ms = a.mask_bcast ? 1 : a.I;
    i64 mj = a.mask_bcast ? o * a.T + (a.T - 1) : j;
    TC g = (TC)(nv[j] * (TS)(tm[mj] ? 0 : 1));
#pragma unroll 4
    for (i64 t = a.T - 1; t >= 0; --t, j -= a.I, mj -= ms) {
        const TS dn = (TS)(d[mj] ? 1 : 0), omd = (TS)(d[mj] ? 0 : 1);
        const TS nvm = nv[j] * (TS)(tm[mj] ? 0 : 1);
        const TG gm = gt ? gt[j] : gs;
        const TG lm = lt ? lt[j] : ls;
        const TG oml = (TG)1 - lm;
        g = g * (TC)omd + (TC)(nvm * dn);
        g = (TC)r[j] + (TC)gm * ((TC)oml * (TC)nvm + (TC)lm * g);
        ret[j] = (TS)g;
    }
}

// rolling_gamma=False (functional.py:877-887): R_k reruns the tail T-1..k with gamma_k, lmbda_k
// fixed.  O(T^2) work; one lane per (o, k, i).
template <typename TS, typename TG>
__global__ __launch_bounds__(kSeqThreads) void td_lambda_fixed_seq_kernel(TdLambdaArgs a) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ r = (const TS *)a.reward;
    const TS *__restrict__ nv = (const TS *)a.next_value;
    const u8 *__restrict__ d = a.done;
    const u8 *__restrict__ tm = a.terminated;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    const TG *__restrict__ lt = (const TG *)a.lmbda_t;
    TS *__restrict__ ret = (TS *)a.returns;

    const i64 idx = (i64)blockIdx.x * kSeqThreads + threadIdx.x;
    if (idx >= a.O * a.T * a.I) return;
    const i64 o = idx / (a.T * a.I);
    const i64 rem = idx - o * a.T * a.I;
    const i64 k = rem / a.I, i = rem - k * a.I;
    const i64 jk = (o * a.T + k) * a.I + i;
    const TG gm = gt ? gt[jk] : (TG)a.gamma;
    const TG lm = lt ? lt[jk] : (TG)a.lmbda;
    const TG oml = (TG)1 - lm;
    i64 j = (o * a.T + (a.T - 1)) * a.I + i;
    const i64 ms = a.mask_bcast ? 1 : a.I;
    i64 mj = a.mask_bcast ? o * a.T + (a.T - 1) : j;
    TC g = (TC)(nv[j] * (TS)(tm[mj] ? 0 : 1));
    for (i64 t = a.T - 1; t >= k; --t, j -= a.I, mj -= ms) {
        const TS dn = (TS)(d[mj] ? 1 : 0), omd = (TS)(d[mj] ? 0 : 1);
        const TS nvm = nv[j] * (TS)(tm[mj] ? 0 : 1);
        g = g * (TC)omd + (TC)(nvm * dn);
        g = (TC)r[j] + (TC)gm * ((TC)oml * (TC)nvm + (TC)lm * g);
    }
    ret[jk] = (TS)g;
}

// TD(1) (functional.py:515-562).  FIXED = rolling_gamma=False: one lane per (o, k, i), the tail T-1..k rerun with
// gamma_k held fixed and g starting from 0 (the reference's python int: no bootstrap from the last next value).
template <typename TS, typename TG, bool FIXED>
__global__ __launch_bounds__(kSeqThreads) void td1_seq_kernel(TdLambdaArgs a) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ r = (const TS *)a.reward;
    const TS *__restrict__ nv = (const TS *)a.next_value;
    const u8 *__restrict__ d = a.done;
    const u8 *__restrict__ tm = a.terminated;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    TS *__restrict__ ret = (TS *)a.returns;

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
    const i64 jk = (o * a.T + k) * a.I + i;
    const TG gk = gt ? gt[jk] : (TG)a.gamma;
    i64 j = (o * a.T + (a.T - 1)) * a.I + i;
    const i64 ms = a.mask_bcast ? 1 : a.I;
    i64 mj = a.mask_bcast ? o * a.T + (a.T - 1) : j;
    TC g = FIXED ? (TC)0 : (TC)nv[j];
#pragma unroll 4
    for (i64 t = a.T - 1; t >= k; --t, j -= a.I, mj -= ms) {
        const TG gm = FIXED ? gk : (gt ? gt[j] : (TG)a.gamma);
        const TG gnt = gm * (TG)(tm[mj] ? 0 : 1);
        const bool b = d[mj] && !tm[mj];
        const TS dnt = (TS)(b ? 1 : 0), omdnt = (TS)(b ? 0 : 1);
        g = (TC)r[j] + (TC)gnt * ((TC)omdnt * g + (TC)(dnt * nv[j]));
        if (!FIXED) ret[j] = (TS)g;
    }
    if (FIXED) ret[jk] = (TS)g;
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T clamp_max_nan(T x, T hi) {  // torch.clamp_max: NaN propagates
    return (x != x) ? x : (x > hi ? hi : x);
}
__device__ __forceinline__ float exp_t(float x) { return expf(x); }
__device__ __forceinline__ double exp_t(double x) { return exp(x); }

template <typename TS, typename TG>
__global__ __launch_bounds__(kSeqThreads) void vtrace_seq_kernel(VtraceArgs a) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ lp = (const TS *)a.log_pi;
    const TS *__restrict__ lmu = (const TS *)a.log_mu;
    const TS *v = (const TS *)a.value;
    const TS *nv = (const TS *)a.next_value;
    const TS *__restrict__ r = (const TS *)a.reward;
    const u8 *__restrict__ d = a.done;
    const u8 *__restrict__ tm = a.terminated;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    TC *__restrict__ adv = (TC *)a.advantage;
    TC *__restrict__ vs_out = (TC *)a.vs;
    const TG gs = (TG)a.gamma;
    const TS rho_s = (TS)a.rho_thresh, c_s = (TS)a.c_thresh;
    const TS *__restrict__ rho_t = (const TS *)a.rho_t;  // optional per-element thresholds (rho.clamp_max(tensor))
    const TS *__restrict__ c_t = (const TS *)a.c_t;

    const i64 idx = (i64)blockIdx.x * kSeqThreads + threadIdx.x;
    if (idx >= a.O * a.I) return;
    const i64 o = idx / a.I, i = idx - o * a.I;
    if (!nv) {  // shifted layout (see gae_seq_kernel)
        v += o * a.I;
        nv = v + a.I;
    }
    i64 j = (o * a.T + (a.T - 1)) * a.I + i;
    const i64 ms = a.mask_bcast ? 1 : a.I;
    i64 mj = a.mask_bcast ? o * a.T + (a.T - 1) : j;
    TC x = (TC)0;
    TC vs_next = (TC)nv[j];
#pragma unroll 2
    for (i64 t = a.T - 1; t >= 0; --t, j -= a.I, mj -= ms) {
        const TG g = gt ? gt[j] : gs;
        const TG dd = g * (TG)(d[mj] ? 0 : 1);
        const TG td = g * (TG)(tm[mj] ? 0 : 1);
        const TS rho = exp_t(lp[j] - lmu[j]);
        const TS crho = clamp_max_nan(rho, rho_t ? rho_t[j] : rho_s);
        const TS cc = clamp_max_nan(rho, c_t ? c_t[j] : c_s);
        const TC delta = (TC)crho * (((TC)r[j] + (TC)td * (TC)nv[j]) - (TC)v[j]);
        x = delta + ((TC)dd * (TC)cc) * x;
        const TC vs = x + (TC)v[j];
        adv[j] = (TC)crho * (((TC)r[j] + (TC)td * vs_next) - (TC)v[j]);
        vs_out[j] = vs;
        vs_next = vs;
    }
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
