// Wavefront-scan kernels ("SCAN" family) for time-contiguous rows (I == 1).
//
// The recurrence x_t = b_t + a_t * x_{t+1} (GAE: a = lmbda*gamma*(1-done), b = delta;
// TD(lambda): a = gamma*lmbda*(1-done), b = r + gamma*v'*(1 - lmbda*(1-done)); V-trace:
// a = gamma*(1-done)*c, b = rho*delta) is a reverse scan of affine maps (a, b).  The flat
// (O*T) array is cut into chunks of kWaves*64*V consecutive elements:
//   1. every lane loads V consecutive time steps with one 16-byte load per tensor
//      (fully coalesced: a wave reads 1 KiB per instruction), fuses the delta / coefficient
//      computation and composes its V maps in registers;
//   2. the 64 lane aggregates are suffix-scanned with wave shuffles (6 steps);
//   3. the wave aggregates of a chunk are exchanged through LDS (one barrier per chunk);
//   4. every lane re-runs its V steps from its carry-in with the reference's exact
//      per-step expression and writes its outputs with 16-byte stores.
// Row ends are *hard* segment boundaries (the carry is discarded by a select, never
// multiplied), `done` boundaries are soft zeros exactly as in the reference
// (x*0, functional.py:174-178), so at a done step the result is delta_t bit-for-bit.
//
// Two geometries, no inter-block communication in either:
//   SHORT rows (T <= chunk): a block owns R whole rows (R*T <= chunk), one chunk, no carry.
//   LONG rows  (T >  chunk): a block owns whole rows and walks each row's chunks from the
//                            end, carrying x (and V-trace's vs) in registers.
//
// HBM traffic = algorithmic bytes: each input element is read once, each output written
// once (GAE fp32 22 B / transition); per-block {sum, sumsq} partials add 16 B per block.
#include "common.h"

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * kWave;

template <typename TC>
struct Aff {
    TC a, b;
    int h;  // hard reset: the incoming value is ignored
};

// apply f2 first, then f1
template <typename TC>
__device__ __forceinline__ Aff<TC> compose(const Aff<TC> &f1, const Aff<TC> &f2) {
    Aff<TC> r;
    const TC t = f1.b + f1.a * f2.b;
    r.a = f1.a * f2.a;
    r.b = f1.h ? f1.b : t;
    r.h = f1.h | f2.h;
    return r;
}
template <typename TC>
__device__ __forceinline__ TC apply(const Aff<TC> &f, TC x) {
    const TC t = f.b + f.a * x;
    return f.h ? f.b : t;
}
template <typename TC>
__device__ __forceinline__ Aff<TC> shfl_down_aff(const Aff<TC> &f, int off) {
    Aff<TC> r;
    r.a = __shfl_down(f.a, off, kWave);
    r.b = __shfl_down(f.b, off, kWave);
    r.h = __shfl_down(f.h, off, kWave);
    return r;
}

// ---- raw vector loads / stores (16 B per lane whenever the element type allows) -------------
template <int BYTES>
struct Raw;
template <>
struct Raw<2> { typedef unsigned short type; };
template <>
struct Raw<4> { typedef unsigned int type; };
template <>
struct Raw<8> { typedef uint2 type; };
template <>
struct Raw<16> { typedef uint4 type; };

template <typename T, int V>
__device__ __forceinline__ void vload(const T *__restrict__ p, T (&out)[V]) {
    constexpr int B = V * (int)sizeof(T);
    if constexpr (B <= 16) {
        typename Raw<B>::type raw = *(const typename Raw<B>::type *)p;
        __builtin_memcpy(out, &raw, B);
    } else {
        static_assert(B % 16 == 0, "vector width");
#pragma unroll
        for (int k = 0; k < B / 16; ++k) {
            uint4 raw = ((const uint4 *)p)[k];
            __builtin_memcpy((char *)out + 16 * k, &raw, 16);
        }
    }
}
template <typename T, int V>
__device__ __forceinline__ void vstore(T *__restrict__ p, const T (&in)[V]) {
    constexpr int B = V * (int)sizeof(T);
    if constexpr (B <= 16) {
        typename Raw<B>::type raw;
        __builtin_memcpy(&raw, in, B);
        *(typename Raw<B>::type *)p = raw;
    } else {
#pragma unroll
        for (int k = 0; k < B / 16; ++k) {
            uint4 raw;
            __builtin_memcpy(&raw, (const char *)in + 16 * k, 16);
            ((uint4 *)p)[k] = raw;
        }
    }
}
template <bool VEC, typename T, int V>
__device__ __forceinline__ void load_n(const T *__restrict__ p, i64 flat, int nvalid, T (&out)[V]) {
    if (VEC && nvalid == V) {
        vload<T, V>(p + flat, out);
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = e < nvalid ? p[flat + e] : (T)0;
    }
}
template <bool VEC, typename T, int V>
__device__ __forceinline__ void store_n(T *__restrict__ p, i64 flat, int nvalid, const T (&in)[V]) {
    if (VEC && nvalid == V) {
        vstore<T, V>(p + flat, in);
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e)
            if (e < nvalid) p[flat + e] = in[e];
    }
}

template <typename T>
__device__ __forceinline__ T clamp_max_nan(T x, T hi) { return (x != x) ? x : (x > hi ? hi : x); }
__device__ __forceinline__ float exp_t(float x) { return expf(x); }
__device__ __forceinline__ double exp_t(double x) { return exp(x); }

// ---- per-estimator element logic -------------------------------------------------------------
// Each Op holds one lane's V consecutive time steps.  Interface used by the scan driver:
//   load<VEC>(args, flat, nvalid)   issue all global loads, then fuse the coefficient math
//   init(e)                         value the recurrence starts from at a row end
//   step(e, x)                      x_t from x_{t+1}: the reference's exact per-step expression
//   affine(e, a, b)                 the same step as an affine map (used for aggregates only)
//   first_value()                   (V-trace) state_value of the lane's first step
//   store<VEC>(...)                 epilogue + global stores (+ fp64 moments for GAE)

template <typename TS_, typename TG_, int V_>
struct GaeOp {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef GaeArgs Args;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = false;
    TS v[V];
    TC delta[V], disc[V];

    template <bool VEC>
    __device__ __forceinline__ void load(const Args &a, i64 flat, int nvalid) {
        TS r[V], nv[V];
        u8 d[V], tm[V];
        TG g[V], lg[V];
        load_n<VEC>((const TS *)a.reward, flat, nvalid, r);
        load_n<VEC>((const TS *)a.value, flat, nvalid, v);
        load_n<VEC>((const TS *)a.next_value, flat, nvalid, nv);
        load_n<VEC>(a.done, flat, nvalid, d);
        load_n<VEC>(a.terminated, flat, nvalid, tm);
        if (a.gamma_t) load_n<VEC>((const TG *)a.gamma_t, flat, nvalid, g);
        if (a.lg_t) load_n<VEC>((const TG *)a.lg_t, flat, nvalid, lg);
        const TG gs = (TG)a.gamma, ls = (TG)a.lmbda, lgs = (TG)a.lg;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const TG ge = a.gamma_t ? g[e] : gs;
            const TG ce = a.lg_t ? lg[e] : (a.gamma_t ? ls * ge : lgs);
            const TG g_nt = ge * (TG)(tm[e] ? 0 : 1);
            disc[e] = (TC)(ce * (TG)(d[e] ? 0 : 1));
            delta[e] = ((TC)r[e] + (TC)g_nt * (TC)nv[e]) - (TC)v[e];
        }
    }
    __device__ __forceinline__ TC init(int) const { return (TC)0; }
    __device__ __forceinline__ TC step(int e, TC x) const { return delta[e] + x * disc[e]; }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const { a = disc[e]; b = delta[e]; }
    __device__ __forceinline__ TS first_value() const { return v[0]; }

    template <bool VEC>
    __device__ __forceinline__ void store(const Args &a, i64 flat, int nvalid, const TC (&x)[V], const bool (&)[V],
                                          TC, TS, double &s, double &ss) const {
        TS adv[V], vt[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            adv[e] = (TS)x[e];
            vt[e] = adv[e] + v[e];
            if (e < nvalid) {
                s += (double)adv[e];
                ss += (double)adv[e] * (double)adv[e];
            }
        }
        store_n<VEC>((TS *)a.advantage, flat, nvalid, adv);
        store_n<VEC>((TS *)a.value_target, flat, nvalid, vt);
    }
};

template <typename TS_, typename TG_, int V_>
struct TdLambdaOp {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef TdLambdaArgs Args;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = false;
    TS r[V], nvm[V], dn[V], omd[V];
    TG gm[V], lm[V];

    template <bool VEC>
    __device__ __forceinline__ void load(const Args &a, i64 flat, int nvalid) {
        TS nv[V];
        u8 d[V], tm[V];
        load_n<VEC>((const TS *)a.reward, flat, nvalid, r);
        load_n<VEC>((const TS *)a.next_value, flat, nvalid, nv);
        load_n<VEC>(a.done, flat, nvalid, d);
        load_n<VEC>(a.terminated, flat, nvalid, tm);
        if (a.gamma_t) load_n<VEC>((const TG *)a.gamma_t, flat, nvalid, gm);
        if (a.lmbda_t) load_n<VEC>((const TG *)a.lmbda_t, flat, nvalid, lm);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            if (!a.gamma_t) gm[e] = (TG)a.gamma;
            if (!a.lmbda_t) lm[e] = (TG)a.lmbda;
            nvm[e] = nv[e] * (TS)(tm[e] ? 0 : 1);
            dn[e] = (TS)(d[e] ? 1 : 0);
            omd[e] = (TS)(d[e] ? 0 : 1);
        }
    }
    // g starts from v'_{T-1} (functional.py:866)
    __device__ __forceinline__ TC init(int e) const { return (TC)nvm[e]; }
    __device__ __forceinline__ TC step(int e, TC x) const {
        const TG oml = (TG)1 - lm[e];
        const TC g1 = x * (TC)omd[e] + (TC)(nvm[e] * dn[e]);
        return (TC)r[e] + (TC)gm[e] * ((TC)oml * (TC)nvm[e] + (TC)lm[e] * g1);
    }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const {
        const TG oml = (TG)1 - lm[e];
        a = ((TC)gm[e] * (TC)lm[e]) * (TC)omd[e];
        b = (TC)r[e] + (TC)gm[e] * ((TC)oml * (TC)nvm[e] + (TC)lm[e] * (TC)(nvm[e] * dn[e]));
    }
    __device__ __forceinline__ TS first_value() const { return (TS)0; }

    template <bool VEC>
    __device__ __forceinline__ void store(const Args &a, i64 flat, int nvalid, const TC (&x)[V], const bool (&)[V],
                                          TC, TS, double &, double &) const {
        TS out[V];
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = (TS)x[e];
        store_n<VEC>((TS *)a.returns, flat, nvalid, out);
    }
};

template <typename TS_, typename TG_, int V_>
struct VtraceOp {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef VtraceArgs Args;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = true;
    TS v[V], nv[V], r[V], crho[V];
    TG td[V];
    TC coef[V], delta[V];

    template <bool VEC>
    __device__ __forceinline__ void load(const Args &a, i64 flat, int nvalid) {
        TS lp[V], lmu[V];
        u8 d[V], tm[V];
        TG g[V];
        load_n<VEC>((const TS *)a.log_pi, flat, nvalid, lp);
        load_n<VEC>((const TS *)a.log_mu, flat, nvalid, lmu);
        load_n<VEC>((const TS *)a.value, flat, nvalid, v);
        load_n<VEC>((const TS *)a.next_value, flat, nvalid, nv);
        load_n<VEC>((const TS *)a.reward, flat, nvalid, r);
        load_n<VEC>(a.done, flat, nvalid, d);
        load_n<VEC>(a.terminated, flat, nvalid, tm);
        if (a.gamma_t) load_n<VEC>((const TG *)a.gamma_t, flat, nvalid, g);
        const TS rho_hi = (TS)a.rho_thresh, c_hi = (TS)a.c_thresh;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const TG ge = a.gamma_t ? g[e] : (TG)a.gamma;
            const TG dd = ge * (TG)(d[e] ? 0 : 1);
            td[e] = ge * (TG)(tm[e] ? 0 : 1);
            const TS rho = exp_t(lp[e] - lmu[e]);
            crho[e] = clamp_max_nan(rho, rho_hi);
            const TS cc = clamp_max_nan(rho, c_hi);
            delta[e] = (TC)crho[e] * (((TC)r[e] + (TC)td[e] * (TC)nv[e]) - (TC)v[e]);
            coef[e] = (TC)dd * (TC)cc;
        }
    }
    __device__ __forceinline__ TC init(int) const { return (TC)0; }
    __device__ __forceinline__ TC step(int e, TC x) const { return delta[e] + coef[e] * x; }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const { a = coef[e]; b = delta[e]; }
    __device__ __forceinline__ TS first_value() const { return v[0]; }

    // x_next / v_next: x and state_value of the step that follows the lane's last step
    // (only read when that step is not a row end).
    template <bool VEC>
    __device__ __forceinline__ void store(const Args &a, i64 flat, int nvalid, const TC (&x)[V], const bool (&row_end)[V],
                                          TC x_next, TS v_next, double &, double &) const {
        TC adv[V], vs[V];
        TC vs_next = x_next + (TC)v_next;
#pragma unroll
        for (int e = V - 1; e >= 0; --e) {
            vs[e] = x[e] + (TC)v[e];
            const TC nxt = row_end[e] ? (TC)nv[e] : vs_next;  // shift ignores done, functional.py:1343-1345
            adv[e] = (TC)crho[e] * (((TC)r[e] + (TC)td[e] * nxt) - (TC)v[e]);
            if (e < nvalid) vs_next = vs[e];
        }
        store_n<VEC>((TC *)a.advantage, flat, nvalid, adv);
        store_n<VEC>((TC *)a.vs, flat, nvalid, vs);
    }
};

// ---- scan driver ------------------------------------------------------------------------------
struct ScanGeom {
    i64 O, T;
    int rows_per_block;  // SHORT geometry
    int long_rows;       // 1: LONG geometry
};

template <class Op>
struct BlockShared {
    typename Op::TC a[2][kWaves], b[2][kWaves];
    int h[2][kWaves];
    typename Op::TS v0[2][kWaves];
};

// Process one chunk: [start, start+count) of the flat array, whose first element is time step
// t0 of its row.  X is the carry: on entry the map's input at the chunk end (x of the element
// that follows the chunk; X.h set if there is none), on exit x of the chunk's first element.
template <class Op, bool VEC, bool WAVE_INDEP>
__device__ __forceinline__ void process_chunk(const typename Op::Args &args, BlockShared<Op> &sh, int buf, i64 start,
                                              int count, unsigned t0, unsigned T, typename Op::TC &Xc,
                                              typename Op::TS &Xv, double &s, double &ss) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int V = Op::V;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const int lane_off = (w * kWave + lane) * V;
    int nvalid = count - lane_off;
    nvalid = nvalid < 0 ? 0 : (nvalid > V ? V : nvalid);
    const i64 flat = start + lane_off;

    Op op;
    op.template load<VEC>(args, flat, nvalid);

    // time index of each element -> row-end flags
    bool row_end[V];
    {
        unsigned t = (t0 + (unsigned)lane_off) % T;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            row_end[e] = (t == T - 1);
            t = row_end[e] ? 0u : t + 1u;
        }
    }

    // 1. lane aggregate
    Aff<TC> agg{(TC)1, (TC)0, 0};
#pragma unroll
    for (int e = V - 1; e >= 0; --e) {
        if (e < nvalid) {
            Aff<TC> el;
            if (row_end[e]) {
                el.a = (TC)0;
                el.b = op.step(e, op.init(e));
                el.h = 1;
            } else {
                op.affine(e, el.a, el.b);
                el.h = 0;
            }
            agg = (e == V - 1) ? el : compose(el, agg);
        }
    }

    // 2. inclusive suffix scan over the wave
    Aff<TC> S = agg;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const Aff<TC> p = shfl_down_aff(S, off);
        if (lane + off < kWave) S = compose(S, p);
    }
    Aff<TC> E = shfl_down_aff(S, 1);  // exclusive: everything after this lane, within the wave
    if (lane == kWave - 1) E = Aff<TC>{(TC)1, (TC)0, 0};
    TS v_next = Op::kNeedsNext ? __shfl_down(op.first_value(), 1, kWave) : (TS)0;

    // 3. carry into this wave
    TC Cw;
    if constexpr (WAVE_INDEP) {
        // every wave's range ends at a row end: S.h is set for every lane, no exchange needed
        Cw = (TC)0;
    } else {
        if (lane == 0) {
            sh.a[buf][w] = S.a;
            sh.b[buf][w] = S.b;
            sh.h[buf][w] = S.h;
            if (Op::kNeedsNext) sh.v0[buf][w] = op.first_value();
        }
        __syncthreads();
        TC x = Xc;
        TS xv = Xv;
        Cw = x;
        TS Cv = xv;
#pragma unroll
        for (int k = kWaves - 1; k >= 0; --k) {
            if (k == w) { Cw = x; Cv = xv; }
            const Aff<TC> g{sh.a[buf][k], sh.b[buf][k], sh.h[buf][k]};
            x = apply(g, x);
            if (Op::kNeedsNext) xv = sh.v0[buf][k];
        }
        Xc = x;  // x of the chunk's first element
        Xv = xv;
        if (Op::kNeedsNext && lane == kWave - 1) v_next = Cv;
    }

    // 4. exact per-step replay from the lane's carry-in
    TC xs[V];
    const TC x_in = apply(E, Cw);
    TC xp = x_in;
#pragma unroll
    for (int e = V - 1; e >= 0; --e) {
        const TC xi = row_end[e] ? op.init(e) : xp;
        xs[e] = op.step(e, xi);
        if (e < nvalid) xp = xs[e];
    }
    op.template store<VEC>(args, flat, nvalid, xs, row_end, x_in, v_next, s, ss);
}

template <class Op, bool VEC, bool WAVE_INDEP, bool LONG>
__global__ __launch_bounds__(kThreads) void scan_kernel(typename Op::Args args, ScanGeom geo, double *partials) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int kChunk = kThreads * Op::V;
    __shared__ BlockShared<Op> sh;
    __shared__ double red[2 * kWaves];
    double s = 0.0, ss = 0.0;
    const unsigned T = (unsigned)geo.T;
    if constexpr (!LONG) {
        const i64 row0 = (i64)blockIdx.x * geo.rows_per_block;
        i64 nrows = geo.O - row0;
        if (nrows > geo.rows_per_block) nrows = geo.rows_per_block;
        TC Xc = (TC)0;
        TS Xv = (TS)0;
        process_chunk<Op, VEC, WAVE_INDEP>(args, sh, 0, row0 * geo.T, (int)(nrows * geo.T), 0u, T, Xc, Xv, s, ss);
    } else {
        const int nchunks = (int)((geo.T + kChunk - 1) / kChunk);
        int buf = 0;
        for (i64 row = blockIdx.x; row < geo.O; row += gridDim.x) {
            TC Xc = (TC)0;
            TS Xv = (TS)0;
            for (int c = nchunks - 1; c >= 0; --c) {
                const i64 off = (i64)c * kChunk;
                i64 cnt = geo.T - off;
                if (cnt > kChunk) cnt = kChunk;
                process_chunk<Op, VEC, false>(args, sh, buf, row * geo.T + off, (int)cnt, (unsigned)off, T, Xc, Xv, s, ss);
                buf ^= 1;
            }
        }
    }
    if (partials) {
        block_sum2<kThreads>(s, ss, red);
        if (threadIdx.x == 0) {
            partials[2 * (i64)blockIdx.x] = s;
            partials[2 * (i64)blockIdx.x + 1] = ss;
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------
struct Plan {
    bool ok;
    bool long_rows, wave_indep, vec;
    int rows_per_block;
    i64 nblocks;
};

constexpr i64 kMaxLongGrid = 256 * 8;  // LONG geometry: blocks grid-stride over rows

i64 gcd_i64(i64 a, i64 b) { while (b) { i64 t = a % b; a = b; b = t; } return a; }

// V: elements per lane; ptr_align_ok: every tensor base pointer is 16-byte aligned.
Plan make_plan(i64 O, i64 T, i64 I, int V, bool ptr_align_ok) {
    Plan p{};
    if (I != 1 || T >= (1LL << 31) - 4096) return p;  // not this family's shape
    const i64 chunk = (i64)kThreads * V;
    p.ok = true;
    if (T <= chunk) {
        // R whole rows per block; prefer R*T multiple of V so every block start stays aligned
        i64 R = chunk / T;
        const i64 q = V / gcd_i64(T, V);
        if (R >= q) R -= R % q;
        if (R > O) R = O;
        p.rows_per_block = (int)R;
        p.nblocks = (O + R - 1) / R;
        p.vec = ptr_align_ok && (p.nblocks == 1 || (R * T) % V == 0);
        p.wave_indep = ((i64)kWave * V) % T == 0;
        p.long_rows = false;
    } else {
        p.long_rows = true;
        p.nblocks = O < kMaxLongGrid ? O : kMaxLongGrid;
        p.vec = ptr_align_ok && (T % V == 0);
        p.wave_indep = false;
    }
    if (p.nblocks > 0x7fffffffLL) p.ok = false;
    return p;
}

template <class Op>
LaunchResult launch_scan(hipStream_t s, const typename Op::Args &a, const Plan &p, i64 O, i64 T, double *partials,
                         const char *name) {
    ScanGeom geo{O, T, p.rows_per_block, p.long_rows ? 1 : 0};
    const dim3 grid((unsigned)p.nblocks), block(kThreads);
#define TRL_LAUNCH(VEC, WI, LG) hipLaunchKernelGGL((scan_kernel<Op, VEC, WI, LG>), grid, block, 0, s, a, geo, partials)
    if (p.long_rows) {
        if (p.vec) TRL_LAUNCH(true, false, true); else TRL_LAUNCH(false, false, true);
    } else if (p.wave_indep) {
        if (p.vec) TRL_LAUNCH(true, true, false); else TRL_LAUNCH(false, true, false);
    } else {
        if (p.vec) TRL_LAUNCH(true, false, false); else TRL_LAUNCH(false, false, false);
    }
#undef TRL_LAUNCH
    return LaunchResult{0, p.nblocks, name};
}

inline bool aligned16(const void *p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; }

template <template <typename, typename, int> class OpT, typename Args, typename F>
LaunchResult dispatch_scan(int vdt, int cdt, F &&f) {
    // V = 16 bytes of the storage dtype per lane
    if (vdt == 0 && cdt == 0) return f(OpT<float, float, 4>{});
    if (vdt == 1 && cdt == 1) return f(OpT<double, double, 2>{});
    if (vdt == 1 && cdt == 0) return f(OpT<double, float, 2>{});
    if (vdt == 0 && cdt == 1) return f(OpT<float, double, 4>{});
    return LaunchResult{1, 0, nullptr};
}

}  // namespace

i64 scan_max_partials(i64 O, i64 T, i64 I) {
    i64 m = 0;
    for (int V : {2, 4}) {
        Plan p = make_plan(O, T, I, V, true);
        if (p.ok && p.nblocks > m) m = p.nblocks;
    }
    return m;
}

LaunchResult launch_gae_scan(hipStream_t s, int vdt, int cdt, const GaeArgs &a) {
    const bool al = aligned16(a.reward) && aligned16(a.value) && aligned16(a.next_value) && aligned16(a.done) &&
                    aligned16(a.terminated) && aligned16(a.gamma_t) && aligned16(a.lg_t) && aligned16(a.advantage) &&
                    aligned16(a.value_target);
    return dispatch_scan<GaeOp, GaeArgs>(vdt, cdt, [&](auto op) {
        typedef decltype(op) Op;
        Plan p = make_plan(a.O, a.T, a.I, Op::V, al);
        if (!p.ok) return LaunchResult{2, 0, nullptr};
        return launch_scan<Op>(s, a, p, a.O, a.T, a.partials, p.long_rows ? "gae_scan_long" : "gae_scan_short");
    });
}

LaunchResult launch_td_lambda_scan(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a) {
    const bool al = aligned16(a.reward) && aligned16(a.next_value) && aligned16(a.done) && aligned16(a.terminated) &&
                    aligned16(a.gamma_t) && aligned16(a.lmbda_t) && aligned16(a.returns);
    return dispatch_scan<TdLambdaOp, TdLambdaArgs>(vdt, cdt, [&](auto op) {
        typedef decltype(op) Op;
        Plan p = make_plan(a.O, a.T, a.I, Op::V, al);
        if (!p.ok) return LaunchResult{2, 0, nullptr};
        return launch_scan<Op>(s, a, p, a.O, a.T, nullptr, p.long_rows ? "td_lambda_scan_long" : "td_lambda_scan_short");
    });
}

LaunchResult launch_vtrace_scan(hipStream_t s, int vdt, int cdt, const VtraceArgs &a) {
    const bool al = aligned16(a.log_pi) && aligned16(a.log_mu) && aligned16(a.value) && aligned16(a.next_value) &&
                    aligned16(a.reward) && aligned16(a.done) && aligned16(a.terminated) && aligned16(a.gamma_t) &&
                    aligned16(a.advantage) && aligned16(a.vs);
    return dispatch_scan<VtraceOp, VtraceArgs>(vdt, cdt, [&](auto op) {
        typedef decltype(op) Op;
        Plan p = make_plan(a.O, a.T, a.I, Op::V, al);
        if (!p.ok) return LaunchResult{2, 0, nullptr};
        return launch_scan<Op>(s, a, p, a.O, a.T, nullptr, p.long_rows ? "vtrace_scan_long" : "vtrace_scan_short");
    });
}
