// Backward of the NON-rolling forms (rolling_gamma=False) of TD(lambda) and TD(1) -- trl_td_fixed_bwd.
//
// Forward (reference functional.py:877-887 and 550-561): for every start k the tail T-1..k is rerun with the coefficients
// of step k held fixed,
//   TD(lambda): g <- v'm_{T-1};  g <- r_i + gamma_k ((1 - lmbda_k) v'm_i + lmbda_k (g nd_i + v'm_i d_i)),   v'm = v' nt
//   TD(1):      g <- 0;          g <- r_i + gamma_k nt_i ((1 - dnt_i) g + dnt_i v'_i),                       dnt = d & ~term
// i.e. R_k = sum_{i >= k} M[k,i] b_i^(k) (+ M[k,T] v'm_{T-1} for TD(lambda)) with M[k,i] = prod_{j=k}^{i-1} c_k m_j,
// c_k = gamma_k lmbda_k | gamma_k and the step mask m_j = nd_j | nt_j (1 - dnt_j).  O(T^2) work per row, like the forward.
//
// Two kernels, both deterministic (fixed summation order), no atomics:
//  * VALUES: d r_i = sum_k gbar_k M[k,i] and d v'_i are COLUMN sums of M.  One wave per (o, feature) row; lane = start k (in
//    strips of 64), the wave walks i upward: every lane advances its own M[k,i] by one factor per step (M[k,i+1] = M[k,i] c_k
//    m_i) and the column sums are wave reductions (DPP butterfly, fp64).  A strip stops as soon as a step mask has zeroed all
//    of its lanes.
//  * COEFFICIENTS: d gamma_k = gbar_k dR_k/dgamma_k (and lmbda likewise) is a forward-mode derivative along the tail of k:
//    one lane per (o, k, feature), the loop of the forward kernel with the two derivative recurrences beside it.
#include "common.h"

namespace {

constexpr int kRowsPerBlock = 4;  // VALUES: one wave per row, 4 independent rows per 256-thread block (no barriers)

template <typename TS, typename TG, bool TD1>
__global__ __launch_bounds__(kRowsPerBlock *kWave) void td_fixed_bwd_values_kernel(TdFixedBwdArgs a) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ gr = (const TS *)a.g_ret;
    const u8 *__restrict__ d = a.done;
    const u8 *__restrict__ tm = a.terminated;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    const TG *__restrict__ lt = (const TG *)a.lmbda_t;
    TS *__restrict__ d_r = (TS *)a.d_reward;
    TS *__restrict__ d_nv = (TS *)a.d_next_value;
    const int lane = threadIdx.x & (kWave - 1);
    const i64 row = (i64)blockIdx.x * kRowsPerBlock + threadIdx.x / kWave;
    if (row >= a.O * a.I) return;  // whole waves leave together
    const i64 o = row / a.I, f = row - o * a.I;
    const i64 T = a.T;
    auto elem = [&](i64 t) { return (o * T + t) * a.I + f; };
    auto mask = [&](i64 t) { return a.mask_bcast ? o * T + t : elem(t); };

    for (i64 ks = 0; ks < T; ks += kWave) {
        const i64 k = ks + lane;
        const bool valid = k < T;
        const i64 jk = valid ? elem(k) : 0;
        const TC gk = (valid && gr) ? (TC)gr[jk] : (TC)0;
        const TG gam = valid ? (gt ? gt[jk] : (TG)a.gamma) : (TG)0;
        const TG lam = TD1 ? (TG)1 : (valid ? (lt ? lt[jk] : (TG)a.lmbda) : (TG)0);
        const TC c = TD1 ? (TC)gam : (TC)gam * (TC)lam;
        TC P = (TC)1;  // M[k,i], meaningful once i >= k
        for (i64 i = ks; i < T; ++i) {
            const bool started = valid && k <= i;
            const i64 mi = mask(i);
            const bool di = d[mi] != 0, ti = tm[mi] != 0;
            const TC w = started ? gk * P : (TC)0;
            const double s1 = wave_sum((double)w);
            const double s2 = wave_sum((double)(w * (TC)gam));
            double dnv;
            TC m;  // step mask of i: the factor every running tail picks up on its way to i + 1
            if (TD1) {
                const bool dnt = di && !ti;
                m = (TC)((ti || dnt) ? 0 : 1);
                dnv = (dnt && !ti) ? s2 : 0.0;  // gamma_k nt_i dnt_i v'_i
            } else {
                const double s3 = wave_sum((double)(w * (TC)gam * (TC)lam));
                m = (TC)(di ? 0 : 1);
                // gamma_k (1 - lmbda_k nd_i) on v'm_i; at i = T-1 the bootstrap adds gamma_k lmbda_k nd_i back
                dnv = (i == T - 1) ? s2 : s2 - (di ? 0.0 : s3);
                dnv = ti ? 0.0 : dnv;  // v'm = v' nt
            }
            if (lane == 0) {
                const i64 ji = elem(i);
                if (d_r) d_r[ji] = (TS)((ks == 0 ? 0.0 : (double)d_r[ji]) + s1);
                if (d_nv) d_nv[ji] = (TS)((ks == 0 ? 0.0 : (double)d_nv[ji]) + dnv);
            }
            if (started) P = P * (c * m);
            // every tail of this strip has met a zero step mask: the remaining columns get nothing from it
            if (i >= ks + kWave - 1 && !__any(started && P != (TC)0)) {
                if (ks == 0 && lane == 0) {  // the first strip also initialises the outputs
                    for (i64 z = i + 1; z < T; ++z) {
                        if (d_r) d_r[elem(z)] = (TS)0;
                        if (d_nv) d_nv[elem(z)] = (TS)0;
                    }
                }
                break;
            }
        }
    }
}

template <typename TS, typename TG, bool TD1>
__global__ __launch_bounds__(256) void td_fixed_bwd_coef_kernel(TdFixedBwdArgs a) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ gr = (const TS *)a.g_ret;
    const TS *__restrict__ r = (const TS *)a.reward;
    const TS *__restrict__ nv = (const TS *)a.next_value;
    const u8 *__restrict__ d = a.done;
    const u8 *__restrict__ tm = a.terminated;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    const TG *__restrict__ lt = (const TG *)a.lmbda_t;
    TC *__restrict__ d_gamma = (TC *)a.d_gamma;
    TC *__restrict__ d_lmbda = (TC *)a.d_lmbda;

    const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    if (idx >= a.O * a.T * a.I) return;
    const i64 o = idx / (a.T * a.I);
    const i64 rem = idx - o * a.T * a.I;
    const i64 k = rem / a.I, f = rem - k * a.I;
    const i64 jk = (o * a.T + k) * a.I + f;
    const TG gm = gt ? gt[jk] : (TG)a.gamma;
    const TG lm = TD1 ? (TG)1 : (lt ? lt[jk] : (TG)a.lmbda);
    const TG oml = (TG)1 - lm;
    i64 j = (o * a.T + (a.T - 1)) * a.I + f;
    const i64 ms = a.mask_bcast ? 1 : a.I;
    i64 mj = a.mask_bcast ? o * a.T + (a.T - 1) : j;
    TC g = TD1 ? (TC)0 : (TC)(nv[j] * (TS)(tm[mj] ? 0 : 1));
    TC dgg = (TC)0, dgl = (TC)0;  // dg / dgamma_k, dg / dlmbda_k
    for (i64 t = a.T - 1; t >= k; --t, j -= a.I, mj -= ms) {
        if (TD1) {
            const TC nt = (TC)(tm[mj] ? 0 : 1);
            const bool b = d[mj] && !tm[mj];
            const TC dnt = (TC)(b ? 1 : 0), omdnt = (TC)(b ? 0 : 1);
            const TC inner = omdnt * g + dnt * (TC)nv[j];
            dgg = nt * inner + (TC)gm * nt * omdnt * dgg;
            g = (TC)r[j] + (TC)gm * nt * inner;
        } else {
            const TC dn = (TC)(d[mj] ? 1 : 0), omd = (TC)(d[mj] ? 0 : 1);
            const TC nvm = (TC)(nv[j] * (TS)(tm[mj] ? 0 : 1));
            const TC gp = g * omd + nvm * dn;
            const TC inner = (TC)oml * nvm + (TC)lm * gp;
            const TC chain = (TC)gm * (TC)lm * omd;
            dgg = inner + chain * dgg;
            dgl = (TC)gm * (gp - nvm) + chain * dgl;
            g = (TC)r[j] + (TC)gm * inner;
        }
    }
    const TC gk = gr ? (TC)gr[jk] : (TC)0;
    if (d_gamma) d_gamma[jk] = gk * dgg;
    if (d_lmbda) d_lmbda[jk] = gk * dgl;
}

template <typename F>
LaunchResult dispatch_fixed_types(int vdt, int cdt, F &&f) {
    if (vdt == 0 && cdt == 0) return f(float{}, float{});
    if (vdt == 1 && cdt == 1) return f(double{}, double{});
    if (vdt == 1 && cdt == 0) return f(double{}, float{});
    if (vdt == 0 && cdt == 1) return f(float{}, double{});
    return LaunchResult{1, 0, nullptr};
}

}  // namespace

LaunchResult launch_td_fixed_bwd(hipStream_t s, int vdt, int cdt, const TdFixedBwdArgs &a, int td1) {
    const i64 rows = a.O * a.I, n = a.O * a.T * a.I;
    const i64 nbv = (rows + kRowsPerBlock - 1) / kRowsPerBlock, nbc = (n + 255) / 256;
    if (nbv <= 0 || nbv > 0x7fffffffLL || nbc > 0x7fffffffLL) return LaunchResult{1, 0, nullptr};
    return dispatch_fixed_types(vdt, cdt, [&](auto ts, auto tg) {
        using TS = decltype(ts);
        using TG = decltype(tg);
        if (a.d_reward || a.d_next_value) {
            if (td1) hipLaunchKernelGGL((td_fixed_bwd_values_kernel<TS, TG, true>), dim3((unsigned)nbv), dim3(kRowsPerBlock * kWave), 0, s, a);
            else hipLaunchKernelGGL((td_fixed_bwd_values_kernel<TS, TG, false>), dim3((unsigned)nbv), dim3(kRowsPerBlock * kWave), 0, s, a);
        }
        if (a.d_gamma || a.d_lmbda) {
            if (td1) hipLaunchKernelGGL((td_fixed_bwd_coef_kernel<TS, TG, true>), dim3((unsigned)nbc), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((td_fixed_bwd_coef_kernel<TS, TG, false>), dim3((unsigned)nbc), dim3(256), 0, s, a);
        }
        return LaunchResult{0, nbv, td1 ? "td1_fixed_bwd" : "td_lambda_fixed_bwd"};
    });
}
