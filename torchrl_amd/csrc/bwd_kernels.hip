// Backward (adjoint) kernels of GAE, TD(lambda), TD(1) and V-trace.
//
// Every estimator is a reverse affine scan x_t = b_t + a_t x_{t+1}; the adjoint of such a scan is the FORWARD scan
//   G_t = gbar_t + a_{t-1} G_{t-1},  G_{-1} = 0            (gbar: incoming gradient w.r.t. x_t, G = dL/db_t)
// followed by an elementwise epilogue (GAE: d_reward = G, d_state_value = g_vt - G, d_next_state_value = G*gamma*nt).
// The reference gets these by autograd through its Python loop (losses build their estimators with
// differentiable=True, torchrl/objectives/utils.py:74-83; test/test_cost.py:15521-15584 asserts reward.grad).
//
// No flip / shift copies: the scan kernels of scan_impl.h are reused with REVERSED ADDRESSING (scan_core.h:
// load_n_rev / store_n_rev) -- lane l of a chunk owns the V elements that END the chunk's mirror image in memory,
// turned around in registers, so in the kernels' index space time runs backwards and the forward scan is again a
// reverse scan; loads and stores stay 16-byte vectors, a wave still touches 64*V consecutive elements.  The one-step
// coefficient shift (a_{t-1}) disappears by scanning the carried quantity instead of G:
//   y_t = a_t (gbar_t + y_{t-1}),   G_t = gbar_t + y_{t-1}
// so every map is built from the element's own inputs; G comes out of the replay as "gbar + carry-in".
// Algorithmic bytes per transition (fp32): GAE 2*4 (g_adv, g_vt) + 2 read, 3*4 written = 22 B; TD(lambda) 4 + 2 read,
// 8 written = 14 B (+4 per coefficient tensor); V-trace 8*4 + 2 read, 5*4 written = 54 B.
// SCAN family: I == 1, any T (short / long / look-back geometries).  SEQ family: any shape, one lane per (o, i) row
// walking time forwards.
#include "scan_impl.h"
#include "seq_rows.h"

namespace {

using namespace scan;

// time index (in the kernels' reversed space) of the element at reversed flat index `flat`
__device__ __forceinline__ unsigned rev_time(i64 flat, i64 T) { return (unsigned)(flat - row_of(flat, T) * T); }

// ---- GAE ------------------------------------------------------------------------------------------------------------
template <typename TS_, typename TG_, int V_>
struct GaeBwdOp {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef GaeBwdArgs Args;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = false, kShifted = false, kReversed = true, kCustomEmit = true;
    typedef TS TO;
    static constexpr int kNumOut = 1;
    struct Raw {
        TS ga[V], gv[V];
        u8 d[V], tm[V];
        TG g[V], lg[V];
    };
    template <class F>
    static __device__ __forceinline__ void for_each_input(const Args &a, Raw &w, F &&f) {
        f((const TS *)a.g_adv, w.ga);  // optional
        f((const TS *)a.g_vt, w.gv);   // optional
        f(a.done, w.d);
        f(a.terminated, w.tm);
        f((const TG *)a.gamma_t, w.g);  // optional
        f((const TG *)a.lg_t, w.lg);    // optional
    }
    static __device__ __forceinline__ void load_extra(const Args &, i64, i64, int, Raw &) {}
    TC gbar[V], c[V];
    TS gv[V];
    TG gnt[V];
    __device__ __forceinline__ void prepare(const Args &a, const Raw &w) {
        const TG gs = (TG)a.gamma, ls = (TG)a.lmbda, lgs = (TG)a.lg;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const TG ge = a.gamma_t ? w.g[e] : gs;
            const TG ce = a.lg_t ? w.lg[e] : (a.gamma_t ? ls * ge : lgs);
            gnt[e] = ge * (TG)(w.tm[e] ? 0 : 1);
            c[e] = (TC)(ce * (TG)(w.d[e] ? 0 : 1));
            gv[e] = a.g_vt ? w.gv[e] : (TS)0;
            gbar[e] = (a.g_adv ? (TC)w.ga[e] : (TC)0) + (TC)gv[e];
        }
    }
    __device__ __forceinline__ TC init(int) const { return (TC)0; }
    __device__ __forceinline__ TC step(int e, TC x) const { return c[e] * (gbar[e] + x); }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const { a = c[e]; b = c[e] * gbar[e]; }
    __device__ __forceinline__ TS first_value() const { return (TS)0; }
    template <bool VEC>
    __device__ __forceinline__ void emit(const Args &a, i64 flat, int nvalid, const TC (&xs)[V], const bool (&row_end)[V], TC x_in,
                                         TS) const {
        TS dr[V], dv[V], dnv[V];
        TC xp = x_in;
#pragma unroll
        for (int e = V - 1; e >= 0; --e) {
            const TC G = gbar[e] + (row_end[e] ? (TC)0 : xp);
            dr[e] = (TS)G;
            dv[e] = (TS)((TC)gv[e] - G);
            dnv[e] = (TS)(G * (TC)gnt[e]);
            if (e < nvalid) xp = xs[e];
        }
        const i64 n = a.O * a.T;
        if (a.d_reward) store_n_rev<VEC>((TS *)a.d_reward, n, flat, nvalid, dr);
        if (a.d_value) store_n_rev<VEC>((TS *)a.d_value, n, flat, nvalid, dv);
        if (a.d_next_value) store_n_rev<VEC>((TS *)a.d_next_value, n, flat, nvalid, dnv);
    }
};

// ---- TD(lambda) (rolling gamma) and TD(1) ------------------------------------------------------------------------------
// TD(lambda): R_t = r_t + gamma_t ((1 - lmbda_t) v'm_t + lmbda_t (R_{t+1} nd_t + v'm_t d_t)),  v'm = v' nt,  R_T := v'm_{T-1}
//   a_t = gamma_t lmbda_t nd_t;  d v'm_t = G_t gamma_t ((1 - lmbda_t) + lmbda_t d_t)  (+ a_{T-1} G_{T-1} at t = T-1);  d v' = d v'm nt
// TD(1):      R_t = r_t + gnt_t (omdnt_t R_{t+1} + dnt_t v'_t),  gnt = gamma nt,  dnt = done & ~terminated,  R_T := v'_{T-1}
//   a_t = gnt_t omdnt_t;         d v'_t = G_t gnt_t dnt_t  (+ a_{T-1} G_{T-1} at t = T-1)
template <typename TS_, typename TG_, int V_, bool TD1>
struct TdBwdOpT {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef TdBwdArgs Args;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = false, kShifted = false, kReversed = true, kCustomEmit = true;
    typedef TS TO;
    static constexpr int kNumOut = 1;
    struct Raw {
        TS gr[V];
        u8 d[V], tm[V];
        TG g[V], l[V];
        unsigned u0;  // reversed time index of the lane's first element
    };
    template <class F>
    static __device__ __forceinline__ void for_each_input(const Args &a, Raw &w, F &&f) {
        f((const TS *)a.g_ret, w.gr);
        f(a.done, w.d);
        f(a.terminated, w.tm);
        f((const TG *)a.gamma_t, w.g);  // optional
        if (!TD1) f((const TG *)a.lmbda_t, w.l);  // optional
    }
    static __device__ __forceinline__ void load_extra(const Args &a, i64, i64 flat, int, Raw &w) { w.u0 = rev_time(flat, a.T); }
    TC gbar[V], c[V], k[V], tailm[V];  // d v'_t = G_t k_t (+ y_{T-1} tailm at t = T-1); masks included
    unsigned u0;
    __device__ __forceinline__ void prepare(const Args &a, const Raw &w) {
        u0 = w.u0;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const TG ge = a.gamma_t ? w.g[e] : (TG)a.gamma;
            const TC nt = (TC)(w.tm[e] ? 0 : 1), dn = (TC)(w.d[e] ? 1 : 0);
            gbar[e] = (TC)w.gr[e];
            if (TD1) {
                const TC gnt = (TC)(ge * (TG)(w.tm[e] ? 0 : 1));
                const bool b = w.d[e] && !w.tm[e];
                c[e] = gnt * (TC)(b ? 0 : 1);
                k[e] = gnt * (TC)(b ? 1 : 0);
                tailm[e] = (TC)1;  // R_T := v'_{T-1}, unmasked
            } else {
                const TG le = a.lmbda_t ? w.l[e] : (TG)a.lmbda;
                c[e] = ((TC)ge * (TC)le) * ((TC)1 - dn);
                k[e] = ((TC)ge * (((TC)1 - (TC)le) + (TC)le * dn)) * nt;
                tailm[e] = nt;     // R_T := v'_{T-1} nt_{T-1}
            }
        }
    }
    __device__ __forceinline__ TC init(int) const { return (TC)0; }
    __device__ __forceinline__ TC step(int e, TC x) const { return c[e] * (gbar[e] + x); }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const { a = c[e]; b = c[e] * gbar[e]; }
    __device__ __forceinline__ TS first_value() const { return (TS)0; }
    template <bool VEC>
    __device__ __forceinline__ void emit(const Args &a, i64 flat, int nvalid, const TC (&xs)[V], const bool (&row_end)[V], TC x_in,
                                         TS) const {
        TS dr[V], dnv[V];
        TC xp = x_in;
#pragma unroll
        for (int e = V - 1; e >= 0; --e) {
            const TC G = gbar[e] + (row_end[e] ? (TC)0 : xp);
            // t == T-1: the forward recurrence started from this step's next value, so what the step would carry on,
            // y_{T-1} = a_{T-1} G_{T-1} (= xs[e]), lands on it
            const bool last_t = e == 0 ? u0 == 0u : row_end[e - 1];
            TC dn = G * k[e];
            if (last_t) dn = dn + xs[e] * tailm[e];
            dr[e] = (TS)G;
            dnv[e] = (TS)dn;
            if (e < nvalid) xp = xs[e];
        }
        const i64 n = a.O * a.T;
        if (a.d_reward) store_n_rev<VEC>((TS *)a.d_reward, n, flat, nvalid, dr);
        if (a.d_next_value) store_n_rev<VEC>((TS *)a.d_next_value, n, flat, nvalid, dnv);
    }
};
template <typename TS, typename TG, int V>
using TdLambdaBwdOp = TdBwdOpT<TS, TG, V, false>;
template <typename TS, typename TG, int V>
using Td1BwdOp = TdBwdOpT<TS, TG, V, true>;

// ---- V-trace ----------------------------------------------------------------------------------------------------------
// forward (functional.py:1319-1348): rho = exp(lp - lmu), rb = min(rho, rho_thresh), cb = min(rho, c_thresh),
//   x_t = rb_t (r_t + gnt_t v'_t - v_t) + (gnd_t cb_t) x_{t+1},  vs = x + v,  vsp_t = vs_{t+1} | v'_{T-1},
//   adv_t = rb_t (r_t + gnt_t vsp_t - v_t)
// backward, A = g_adv, S = g_vs:  d vsp_t = A_t rb_t gnt_t;  s_t = S_t + d vsp_{t-1};  X_t = s_t + a_{t-1} X_{t-1}, a = gnd cb
//   scanned as y_t = d vsp_t + a_t (S_t + y_{t-1}),  X_t = S_t + y_{t-1};  d vsp of the neighbouring step travels with the
//   carry the way state_value does in the forward kernel (kNeedsNext).
//   d rb = A (r + gnt vsp - v) + X (r + gnt v' - v);  d cb = X x_{t+1} gnd;  d rho = d rb [rho <= rho_thresh] + d cb [rho <= c_thresh]
//   d lp = d rho rho = -d lmu;  d r = (A + X) rb;  d v = s - d r;  d v' = X rb gnt (+ d vsp_{T-1} at t = T-1)
//   d gamma_t = nt rb (A vsp + X v') + nd X x_{t+1} cb
template <typename TS_, typename TG_, int V_>
struct VtraceBwdOp {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef VtraceBwdArgs Args;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = true, kShifted = false, kReversed = true, kCustomEmit = true;
    // scan_impl.h: OpPrefetch*.  Two sets of the 10 input arrays spill in the 1024-thread short kernel; in the long kernel
    // (256-thread blocks) the extra wave per SIMD is worth more than the prefetch (C5 fp32 135 -> 114 us; fp64 at V = 1,
    // dispatch_bwd_scan: 227 with, 225 without; C4 fp32 34 -> 23 us together with V = 2).
    static constexpr bool kPrefetchShort = false, kPrefetchLong = false;
    static constexpr bool kDualScan = false;  // scan_impl.h: OpDualScan
    typedef TS TO;
    static constexpr int kNumOut = 1;
    struct Raw {
        TC ga[V], gs[V], vs[V];
        TS lp[V], lmu[V], v[V], nv[V], r[V];
        u8 d[V], tm[V];
        TG g[V];
        TC vs_x;     // vs / state_value of the step after (in time) the lane's first element
        TS v_x;
        unsigned u0;
    };
    template <class F>
    static __device__ __forceinline__ void for_each_input(const Args &a, Raw &w, F &&f) {
        f((const TC *)a.g_adv, w.ga);  // optional
        f((const TC *)a.g_vs, w.gs);   // optional
        f((const TC *)a.vs, w.vs);
        f((const TS *)a.log_pi, w.lp);
        f((const TS *)a.log_mu, w.lmu);
        f((const TS *)a.value, w.v);
        f((const TS *)a.next_value, w.nv);
        f((const TS *)a.reward, w.r);
        f(a.done, w.d);
        f(a.terminated, w.tm);
        f((const TG *)a.gamma_t, w.g);  // optional
    }
    static __device__ __forceinline__ void load_extra(const Args &a, i64 n_total, i64 flat, int nvalid, Raw &w) {
        w.u0 = rev_time(flat, a.T);
        const bool has = nvalid > 0 && w.u0 != 0u;  // then flat > 0: the element exists
        const i64 m = n_total - flat;               // memory index of the lane's first element, plus one
        w.vs_x = has ? ((const TC *)a.vs)[m] : (TC)0;
        w.v_x = has ? ((const TS *)a.value)[m] : (TS)0;
    }
    // Register diet (10 arrays of V instead of 14: the 1024-thread short kernel has 128 VGPRs per lane): gamma*nt, gamma*nd,
    // d vsp and the scan coefficient are re-formed from g, the mask bits, rb and cb where they are used -- same expressions,
    // same roundings.
    TC A[V], S[V], vs[V];
    TS v[V], nv[V], r[V], rho[V], rb[V], cb[V];
    TG g[V];
    u8 pass;  // bit e: rho_e <= rho_thresh; bit V + e: rho_e <= c_thresh
    u8 ntm, ndm;
    TC vs_x;
    TS v_x;
    unsigned u0;
    __device__ __forceinline__ TG gnt(int e) const { return g[e] * (TG)((ntm >> e) & 1); }
    __device__ __forceinline__ TG gnd(int e) const { return g[e] * (TG)((ndm >> e) & 1); }
    __device__ __forceinline__ TC dvsp(int e) const { return A[e] * (TC)rb[e] * (TC)gnt(e); }
    __device__ __forceinline__ TC coef(int e) const { return (TC)gnd(e) * (TC)cb[e]; }
    __device__ __forceinline__ void prepare(const Args &a, const Raw &w) {
        const TS rho_hi = (TS)a.rho_thresh, c_hi = (TS)a.c_thresh;
        u0 = w.u0;
        vs_x = w.vs_x;
        v_x = w.v_x;
        pass = 0;
        ntm = ndm = 0;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            v[e] = w.v[e];
            nv[e] = w.nv[e];
            r[e] = w.r[e];
            vs[e] = w.vs[e];
            g[e] = a.gamma_t ? w.g[e] : (TG)a.gamma;
            ntm |= (u8)((w.tm[e] ? 0 : 1) << e);
            ndm |= (u8)((w.d[e] ? 0 : 1) << e);
            rho[e] = exp_t(w.lp[e] - w.lmu[e]);
            rb[e] = clamp_max_nan(rho[e], rho_hi);
            cb[e] = clamp_max_nan(rho[e], c_hi);
            pass |= (u8)((rho[e] <= rho_hi ? 1 : 0) << e);
            pass |= (u8)((rho[e] <= c_hi ? 1 : 0) << (V + e));
            A[e] = a.g_adv ? w.ga[e] : (TC)0;
            S[e] = a.g_vs ? w.gs[e] : (TC)0;
        }
    }
    __device__ __forceinline__ TC init(int) const { return (TC)0; }
    __device__ __forceinline__ TC step(int e, TC x) const { return dvsp(e) + coef(e) * (S[e] + x); }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const { a = coef(e); b = dvsp(e) + a * S[e]; }
    // d vsp of the lane's first element: what the step before it (in time) adds to its gradient w.r.t. vs
    __device__ __forceinline__ TS first_value() const { return (TS)dvsp(0); }
    template <bool VEC>
    __device__ __forceinline__ void emit(const Args &a, i64 flat, int nvalid, const TC (&xs)[V], const bool (&row_end)[V], TC x_in,
                                         TS v_next) const {
        TS dlp[V], dlm[V], dv[V], dnv[V], dr[V];
        TC dg[V];
        TC xp = x_in, dprev = (TC)v_next;
#pragma unroll
        for (int e = V - 1; e >= 0; --e) {
            const bool first_t = row_end[e];                          // t == 0
            const bool last_t = e == 0 ? u0 == 0u : row_end[e - 1];   // t == T - 1
            const TC X = S[e] + (first_t ? (TC)0 : xp);
            const TC st = S[e] + (first_t ? (TC)0 : dprev);
            const TC vs_n = e == 0 ? vs_x : vs[e > 0 ? e - 1 : 0];
            const TS v_n = e == 0 ? v_x : v[e > 0 ? e - 1 : 0];
            const TC vsp = last_t ? (TC)nv[e] : vs_n;
            const TC x_next = last_t ? (TC)0 : vs_n - (TC)v_n;
            const TC base_a = ((TC)r[e] + (TC)gnt(e) * vsp) - (TC)v[e];
            const TC base_x = ((TC)r[e] + (TC)gnt(e) * (TC)nv[e]) - (TC)v[e];
            const TC d_rb = A[e] * base_a + X * base_x;
            const TC d_cb = X * x_next * (TC)gnd(e);
            const TC d_rho = (((pass >> e) & 1) ? d_rb : (TC)0) + (((pass >> (V + e)) & 1) ? d_cb : (TC)0);
            const TC d_lp = d_rho * (TC)rho[e];
            const TC w = (A[e] + X) * (TC)rb[e];
            TC d_nv = X * (TC)rb[e] * (TC)gnt(e);
            if (last_t) d_nv = d_nv + dvsp(e);
            dlp[e] = (TS)d_lp;
            dlm[e] = (TS)(-d_lp);
            dr[e] = (TS)w;
            dv[e] = (TS)(st - w);
            dnv[e] = (TS)d_nv;
            dg[e] = (TC)((ntm >> e) & 1) * (TC)rb[e] * (A[e] * vsp + X * (TC)nv[e]) + (TC)((ndm >> e) & 1) * X * x_next * (TC)cb[e];
            if (e < nvalid) { xp = xs[e]; dprev = dvsp(e); }
        }
        const i64 n = a.O * a.T;
        if (a.d_log_pi) store_n_rev<VEC>((TS *)a.d_log_pi, n, flat, nvalid, dlp);
        if (a.d_log_mu) store_n_rev<VEC>((TS *)a.d_log_mu, n, flat, nvalid, dlm);
        if (a.d_value) store_n_rev<VEC>((TS *)a.d_value, n, flat, nvalid, dv);
        if (a.d_next_value) store_n_rev<VEC>((TS *)a.d_next_value, n, flat, nvalid, dnv);
        if (a.d_reward) store_n_rev<VEC>((TS *)a.d_reward, n, flat, nvalid, dr);
        if (a.d_gamma_t) store_n_rev<VEC>((TC *)a.d_gamma_t, n, flat, nvalid, dg);
    }
};

// ---- SEQ family: one lane per (o, i) row, time walked forwards; any shape ----------------------------------------------
constexpr int kBwdSeqThreads = 64;

// the per-row loops: seq_rows.h (shared with the host entry points of cpu_host.hip)
template <typename TS, typename TG>
__global__ __launch_bounds__(kBwdSeqThreads) void gae_bwd_seq_kernel(GaeBwdArgs a) {
    const i64 idx = (i64)blockIdx.x * kBwdSeqThreads + threadIdx.x;
    if (idx >= a.O * a.I) return;
    const i64 o = idx / a.I;
    seqrow::gae_bwd_row<TS, TG>(a, o, idx - o * a.I);
}

template <typename TS, typename TG, bool TD1>
__global__ __launch_bounds__(kBwdSeqThreads) void td_bwd_seq_kernel(TdBwdArgs a) {
    const i64 idx = (i64)blockIdx.x * kBwdSeqThreads + threadIdx.x;
    if (idx >= a.O * a.I) return;
    const i64 o = idx / a.I;
    seqrow::td_bwd_row<TS, TG, TD1>(a, o, idx - o * a.I);
}

template <typename TS, typename TG>
__global__ __launch_bounds__(kBwdSeqThreads) void vtrace_bwd_seq_kernel(VtraceBwdArgs a) {
    const i64 idx = (i64)blockIdx.x * kBwdSeqThreads + threadIdx.x;
    if (idx >= a.O * a.I) return;
    const i64 o = idx / a.I;
    seqrow::vtrace_bwd_row<TS, TG>(a, o, idx - o * a.I);
}

template <typename F>
LaunchResult dispatch_bwd_types(int vdt, int cdt, F &&f) {
    if (vdt == 0 && cdt == 0) return f(float{}, float{});
    if (vdt == 1 && cdt == 1) return f(double{}, double{});
    if (vdt == 1 && cdt == 0) return f(double{}, float{});
    if (vdt == 0 && cdt == 1) return f(float{}, double{});
    return LaunchResult{1, 0, nullptr};
}
inline bool bwd_grid_ok(i64 nb) { return nb > 0 && nb <= 0x7fffffffLL; }

// SCAN launch of a reversed op: I == 1 only; vector accesses also need O*T % V == 0 (mirror image of an aligned vector)
template <template <typename, typename, int> class OpT, class Args, typename F>
LaunchResult dispatch_bwd_scan(int vdt, int cdt, i64 T, F &&f) {
    if constexpr (std::is_same<Args, VtraceBwdArgs>::value) {
        // the register-heavy op: fp32 rows that fit a chunk of 2 elements per lane (T <= 512) take V = 2 -- no spills under the
        // short kernel's 128-VGPR cap, 8-byte accesses (C4 34 -> 23 us); longer rows keep 16-byte accesses
        if (vdt == 0 && T <= 2 * kTeam) {
            if (cdt == 0) return f(OpT<float, float, 2>{});
            if (cdt == 1) return f(OpT<float, double, 2>{});
        }
        // fp64: ONE element per lane (8-byte accesses).  V = 2 needs 164 VGPRs (3 waves per SIMD); V = 1 about half of that.
        // Measured (round 3, tools/dev/bench_vtrace_bwd.py): C4 55.2 -> 36.2 us, C5 261.5 -> 225.0 us (4.15 -> 4.82 TB/s, the
        // DRAM-bound rate of this box for a 1 GB footprint).  fp32 long rows with V = 2: 179 us (123 with the prefetch) against
        // 115 with V = 4 -- there the per-chunk latency chain, not occupancy, is what counts.
        if (vdt == 1 && cdt == 1) return f(OpT<double, double, 1>{});
        if (vdt == 1 && cdt == 0) return f(OpT<double, float, 1>{});
    }
    return dispatch_scan<OpT, Args>(vdt, cdt, f);
}

template <template <typename, typename, int> class OpT, class Args, typename... P>
LaunchResult launch_bwd_scan(hipStream_t s, int vdt, int cdt, const Args &a, void *lb_ws, size_t lb_bytes, const char *base,
                             P... ptrs) {
    if (a.I != 1) return LaunchResult{2, 0, nullptr};
    const bool al = (aligned16(ptrs) && ...);
    return dispatch_bwd_scan<OpT, Args>(vdt, cdt, a.T, [&](auto op) {
        typedef decltype(op) Op;
        const bool vec_ok = al && (a.O * a.T) % Op::V == 0;
        Plan p = make_plan(a.O, a.T, 1, Op::V, vec_ok, device_cus(), lb_ws ? lb_bytes : 0, false);
        if (!p.ok) return LaunchResult{2, 0, nullptr};
        return launch_scan<Op>(s, a, p, a.O, a.T, 1, nullptr, base, lb_ws);
    });
}

}  // namespace

LaunchResult launch_gae_bwd_scan(hipStream_t s, int vdt, int cdt, const GaeBwdArgs &a, void *lb_ws, size_t lb_bytes) {
    return launch_bwd_scan<GaeBwdOp>(s, vdt, cdt, a, lb_ws, lb_bytes, "gae_bwd_scan", a.g_adv, a.g_vt, (const void *)a.done,
                                     (const void *)a.terminated, a.gamma_t, a.lg_t, (const void *)a.d_reward, (const void *)a.d_value,
                                     (const void *)a.d_next_value);
}
LaunchResult launch_gae_bwd_seq(hipStream_t s, int vdt, int cdt, const GaeBwdArgs &a) {
    const i64 nb = (a.O * a.I + kBwdSeqThreads - 1) / kBwdSeqThreads;
    if (!bwd_grid_ok(nb)) return LaunchResult{1, 0, nullptr};
    return dispatch_bwd_types(vdt, cdt, [&](auto ts, auto tg) {
        hipLaunchKernelGGL((gae_bwd_seq_kernel<decltype(ts), decltype(tg)>), dim3((unsigned)nb), dim3(kBwdSeqThreads), 0, s, a);
        return LaunchResult{0, 0, "gae_bwd_seq"};
    });
}

LaunchResult launch_td_bwd_scan(hipStream_t s, int vdt, int cdt, const TdBwdArgs &a, int td1, void *lb_ws, size_t lb_bytes) {
    if (td1)
        return launch_bwd_scan<Td1BwdOp>(s, vdt, cdt, a, lb_ws, lb_bytes, "td1_bwd_scan", a.g_ret, (const void *)a.done,
                                         (const void *)a.terminated, a.gamma_t, (const void *)a.d_reward, (const void *)a.d_next_value);
    return launch_bwd_scan<TdLambdaBwdOp>(s, vdt, cdt, a, lb_ws, lb_bytes, "td_lambda_bwd_scan", a.g_ret, (const void *)a.done,
                                          (const void *)a.terminated, a.gamma_t, a.lmbda_t, (const void *)a.d_reward,
                                          (const void *)a.d_next_value);
}
LaunchResult launch_td_bwd_seq(hipStream_t s, int vdt, int cdt, const TdBwdArgs &a, int td1) {
    const i64 nb = (a.O * a.I + kBwdSeqThreads - 1) / kBwdSeqThreads;
    if (!bwd_grid_ok(nb)) return LaunchResult{1, 0, nullptr};
    return dispatch_bwd_types(vdt, cdt, [&](auto ts, auto tg) {
        using TS = decltype(ts);
        using TG = decltype(tg);
        if (td1) hipLaunchKernelGGL((td_bwd_seq_kernel<TS, TG, true>), dim3((unsigned)nb), dim3(kBwdSeqThreads), 0, s, a);
        else hipLaunchKernelGGL((td_bwd_seq_kernel<TS, TG, false>), dim3((unsigned)nb), dim3(kBwdSeqThreads), 0, s, a);
        return LaunchResult{0, 0, td1 ? "td1_bwd_seq" : "td_lambda_bwd_seq"};
    });
}

LaunchResult launch_vtrace_bwd_scan(hipStream_t s, int vdt, int cdt, const VtraceBwdArgs &a, void *lb_ws, size_t lb_bytes) {
    if (a.rho_t || a.c_t) return LaunchResult{2, 0, nullptr};  // per-element thresholds: SEQ family
    return launch_bwd_scan<VtraceBwdOp>(s, vdt, cdt, a, lb_ws, lb_bytes, "vtrace_bwd_scan", a.g_adv, a.g_vs, a.log_pi, a.log_mu, a.value,
                                        a.next_value, a.reward, a.vs, (const void *)a.done, (const void *)a.terminated, a.gamma_t,
                                        (const void *)a.d_log_pi, (const void *)a.d_log_mu, (const void *)a.d_value,
                                        (const void *)a.d_next_value, (const void *)a.d_reward, (const void *)a.d_gamma_t);
}
LaunchResult launch_vtrace_bwd_seq(hipStream_t s, int vdt, int cdt, const VtraceBwdArgs &a) {
    const i64 nb = (a.O * a.I + kBwdSeqThreads - 1) / kBwdSeqThreads;
    if (!bwd_grid_ok(nb)) return LaunchResult{1, 0, nullptr};
    return dispatch_bwd_types(vdt, cdt, [&](auto ts, auto tg) {
        hipLaunchKernelGGL((vtrace_bwd_seq_kernel<decltype(ts), decltype(tg)>), dim3((unsigned)nb), dim3(kBwdSeqThreads), 0, s, a);
        return LaunchResult{0, 0, "vtrace_bwd_seq"};
    });
}
