// One (o, i) row of every estimator, walked sequentially over time in exactly the operation order of the reference's Python
// loops (torchrl/objectives/value/functional.py:165-180 GAE, :841-887 TD(lambda), :515-562 TD(1), :1319-1348 V-trace) --
// the arithmetic of the SEQ kernel family.  __host__ __device__: the SEQ kernels (seq_kernels.hip, bwd_kernels.hip) run one
// row per lane, the host entry points (cpu_host.hip: trl_*_cpu, tensors in host memory) run the same functions over the rows
// of a thread's share -- one statement of the per-step expressions for both, compiled with -ffp-contract=off on both sides.
// Index conventions: canonical (O, T, I) layout, element (o, t, i) at (o*T + t)*I + i; masks either full arrays or (O, T)
// arrays shared by the I feature elements of a step (mask_bcast).
#pragma once
#include "common.h"

#include <math.h>

namespace seqrow {

#define TRL_HD __host__ __device__ __forceinline__

template <typename T>
TRL_HD T clamp_max_nan(T x, T hi) {  // torch.clamp_max: NaN propagates
    return (x != x) ? x : (x > hi ? hi : x);
}
TRL_HD float exp_t(float x) { return expf(x); }
TRL_HD double exp_t(double x) { return exp(x); }

// ---- forward ------------------------------------------------------------------------------------------------------------
// GAE (functional.py:165-180).  s / ss: fp64 moments of the row's advantages, added to.
template <typename TS, typename TG>
TRL_HD void gae_row(const GaeArgs &a, i64 o, i64 i, double &s, double &ss) {
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

// TD(lambda), rolling gamma (functional.py:866-876)
template <typename TS, typename TG>
TRL_HD void td_lambda_row(const TdLambdaArgs &a, i64 o, i64 i) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ r = (const TS *)a.reward;
    const TS *__restrict__ nv = (const TS *)a.next_value;
    const u8 *__restrict__ d = a.done;
    const u8 *__restrict__ tm = a.terminated;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    const TG *__restrict__ lt = (const TG *)a.lmbda_t;
    TS *__restrict__ ret = (TS *)a.returns;
    const TG gs = (TG)a.gamma, ls = (TG)a.lmbda;
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

// TD(lambda), rolling_gamma=False (functional.py:877-887): R_k reruns the tail T-1..k with gamma_k, lmbda_k fixed; one
// call per (o, k, i)
template <typename TS, typename TG>
TRL_HD void td_lambda_fixed_elem(const TdLambdaArgs &a, i64 o, i64 k, i64 i) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ r = (const TS *)a.reward;
    const TS *__restrict__ nv = (const TS *)a.next_value;
    const u8 *__restrict__ d = a.done;
    const u8 *__restrict__ tm = a.terminated;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    const TG *__restrict__ lt = (const TG *)a.lmbda_t;
    TS *__restrict__ ret = (TS *)a.returns;
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

// TD(1) (functional.py:515-562).  FIXED = rolling_gamma=False: one call per (o, k, i), the tail T-1..k rerun with gamma_k
// held fixed and g starting from 0 (the reference's python int: no bootstrap from the last next value); else k = 0.
template <typename TS, typename TG, bool FIXED>
TRL_HD void td1_row(const TdLambdaArgs &a, i64 o, i64 k, i64 i) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ r = (const TS *)a.reward;
    const TS *__restrict__ nv = (const TS *)a.next_value;
    const u8 *__restrict__ d = a.done;
    const u8 *__restrict__ tm = a.terminated;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    TS *__restrict__ ret = (TS *)a.returns;
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

// V-trace (functional.py:1319-1348)
template <typename TS, typename TG>
TRL_HD void vtrace_row(const VtraceArgs &a, i64 o, i64 i) {
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
    if (!nv) {  // shifted layout (see gae_row)
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

// ---- backward (adjoints): time walked forwards; see bwd_kernels.hip for the derivation ----------------------------------------
template <typename TS, typename TG>
TRL_HD void gae_bwd_row(const GaeBwdArgs &a, i64 o, i64 i) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ ga = (const TS *)a.g_adv;
    const TS *__restrict__ gvt = (const TS *)a.g_vt;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    const TG *__restrict__ lgt = (const TG *)a.lg_t;
    TS *__restrict__ dr = (TS *)a.d_reward;
    TS *__restrict__ dv = (TS *)a.d_value;
    TS *__restrict__ dnv = (TS *)a.d_next_value;
    const TG gs = (TG)a.gamma, ls = (TG)a.lmbda, lgs = (TG)a.lg;
    i64 j = o * a.T * a.I + i;
    const i64 ms = a.mask_bcast ? 1 : a.I;
    i64 mj = a.mask_bcast ? o * a.T : j;
    TC y = (TC)0;  // a_{t-1} G_{t-1}
    for (i64 t = 0; t < a.T; ++t, j += a.I, mj += ms) {
        const TG g = gt ? gt[j] : gs;
        const TG c = lgt ? lgt[j] : (gt ? ls * g : lgs);
        const TS gv = gvt ? gvt[j] : (TS)0;
        const TC G = ((ga ? (TC)ga[j] : (TC)0) + (TC)gv) + y;
        if (dr) dr[j] = (TS)G;
        if (dv) dv[j] = (TS)((TC)gv - G);
        if (dnv) dnv[j] = (TS)(G * (TC)(g * (TG)(a.terminated[mj] ? 0 : 1)));
        y = (TC)(c * (TG)(a.done[mj] ? 0 : 1)) * G;
    }
}

template <typename TS, typename TG, bool TD1>
TRL_HD void td_bwd_row(const TdBwdArgs &a, i64 o, i64 i) {
    using TC = typename Promote<TS, TG>::type;
    const TS *__restrict__ gr = (const TS *)a.g_ret;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    const TG *__restrict__ lt = (const TG *)a.lmbda_t;
    TS *__restrict__ dr = (TS *)a.d_reward;
    TS *__restrict__ dnv = (TS *)a.d_next_value;
    i64 j = o * a.T * a.I + i;
    const i64 ms = a.mask_bcast ? 1 : a.I;
    i64 mj = a.mask_bcast ? o * a.T : j;
    TC y = (TC)0;
    for (i64 t = 0; t < a.T; ++t, j += a.I, mj += ms) {
        const TG ge = gt ? gt[j] : (TG)a.gamma;
        const bool tmv = a.terminated[mj] != 0, dv_ = a.done[mj] != 0;
        const TC nt = (TC)(tmv ? 0 : 1), dn = (TC)(dv_ ? 1 : 0);
        TC c, k, tailm;
        if (TD1) {
            const TC gnt = (TC)(ge * (TG)(tmv ? 0 : 1));
            const bool b = dv_ && !tmv;
            c = gnt * (TC)(b ? 0 : 1);
            k = gnt * (TC)(b ? 1 : 0);
            tailm = (TC)1;
        } else {
            const TG le = lt ? lt[j] : (TG)a.lmbda;
            c = ((TC)ge * (TC)le) * ((TC)1 - dn);
            k = ((TC)ge * (((TC)1 - (TC)le) + (TC)le * dn)) * nt;
            tailm = nt;
        }
        const TC G = (TC)gr[j] + y;
        y = c * G;
        TC d = G * k;
        if (t == a.T - 1) d = d + y * tailm;
        if (dr) dr[j] = (TS)G;
        if (dnv) dnv[j] = (TS)d;
    }
}

template <typename TS, typename TG>
TRL_HD void vtrace_bwd_row(const VtraceBwdArgs &a, i64 o, i64 i) {
    using TC = typename Promote<TS, TG>::type;
    const TC *__restrict__ ga = (const TC *)a.g_adv;
    const TC *__restrict__ gs = (const TC *)a.g_vs;
    const TC *__restrict__ vs = (const TC *)a.vs;
    const TS *__restrict__ lp = (const TS *)a.log_pi;
    const TS *__restrict__ lmu = (const TS *)a.log_mu;
    const TS *__restrict__ v = (const TS *)a.value;
    const TS *__restrict__ nv = (const TS *)a.next_value;
    const TS *__restrict__ r = (const TS *)a.reward;
    const TG *__restrict__ gt = (const TG *)a.gamma_t;
    const TS rho_s = (TS)a.rho_thresh, c_s = (TS)a.c_thresh;
    const TS *__restrict__ rho_t = (const TS *)a.rho_t;
    const TS *__restrict__ c_t = (const TS *)a.c_t;
    i64 j = o * a.T * a.I + i;
    const i64 ms = a.mask_bcast ? 1 : a.I;
    i64 mj = a.mask_bcast ? o * a.T : j;
    TC y = (TC)0, dprev = (TC)0;
    for (i64 t = 0; t < a.T; ++t, j += a.I, mj += ms) {
        const bool last_t = t == a.T - 1;
        const TG ge = gt ? gt[j] : (TG)a.gamma;
        const TC nd = (TC)(a.done[mj] ? 0 : 1), nt = (TC)(a.terminated[mj] ? 0 : 1);
        const TG gnd = ge * (TG)(a.done[mj] ? 0 : 1), gnt = ge * (TG)(a.terminated[mj] ? 0 : 1);
        const TS rho = exp_t(lp[j] - lmu[j]);
        const TS rho_hi = rho_t ? rho_t[j] : rho_s, c_hi = c_t ? c_t[j] : c_s;
        const TS rb = clamp_max_nan(rho, rho_hi), cb = clamp_max_nan(rho, c_hi);
        const TC A = ga ? ga[j] : (TC)0, S = gs ? gs[j] : (TC)0;
        const TC dvsp = A * (TC)rb * (TC)gnt;
        const TC coef = (TC)gnd * (TC)cb;
        const TC X = S + y, st = S + dprev;
        const TC vs_n = last_t ? (TC)0 : vs[j + a.I];
        const TC vsp = last_t ? (TC)nv[j] : vs_n;
        const TC x_next = last_t ? (TC)0 : vs_n - (TC)v[j + a.I];
        const TC d_rb = A * (((TC)r[j] + (TC)gnt * vsp) - (TC)v[j]) + X * (((TC)r[j] + (TC)gnt * (TC)nv[j]) - (TC)v[j]);
        const TC d_cb = X * x_next * (TC)gnd;
        const TC d_rho = (rho <= rho_hi ? d_rb : (TC)0) + (rho <= c_hi ? d_cb : (TC)0);
        const TC d_lp = d_rho * (TC)rho;
        const TC w = (A + X) * (TC)rb;
        TC d_nv = X * (TC)rb * (TC)gnt;
        if (last_t) d_nv = d_nv + dvsp;
        if (a.d_log_pi) ((TS *)a.d_log_pi)[j] = (TS)d_lp;
        if (a.d_log_mu) ((TS *)a.d_log_mu)[j] = (TS)(-d_lp);
        if (a.d_value) ((TS *)a.d_value)[j] = (TS)(st - w);
        if (a.d_next_value) ((TS *)a.d_next_value)[j] = (TS)d_nv;
        if (a.d_reward) ((TS *)a.d_reward)[j] = (TS)w;
        if (a.d_gamma_t) ((TC *)a.d_gamma_t)[j] = nt * (TC)rb * (A * vsp + X * (TC)nv[j]) + nd * X * x_next * (TC)cb;
        y = dvsp + coef * X;
        dprev = dvsp;
    }
}

#undef TRL_HD

}  // namespace seqrow
