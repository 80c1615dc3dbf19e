// GAE for rollouts with feature dims after time (I > 1, whole outer rows inside one chunk: T*I <= 256*V) -- the VMAS /
// MAPPO shape [E, T, A, 1] (BASELINE config 3).  Element (o, t, i) sits at (o*T + t)*I + i: the I scans of an outer row are
// interleaved in memory.  As in the generic kernel of scan_impl.h a team loads a chunk of whole outer rows with coalesced
// 16-byte loads and turns it through LDS so that every lane ends up with V consecutive time steps of ONE (o, i) row; the
// difference is WHAT crosses LDS.  Everything elementwise is done in memory order, before and after the turn:
//   in : delta = (r + gamma*nt*v') - v   and   disc = lmbda*gamma*nd     (2 values per element instead of 5 tensors)
//   out: the advantage; value_target = advantage + v is formed in memory order from the v the lane still holds
// -- 16 B of LDS traffic per element and direction instead of 44, a third of the LDS footprint (12.7 KB per block) and far
// fewer live registers, so twice as many blocks are resident.  C3 [4096 x 100 x 5]: 13.1 -> 9.8 us.
// Same arithmetic as every other family: the lane replays its steps with delta + x*disc from its carry-in.
// Round 3: the estimator-specific parts are policies (GaeTurn, TdLambdaTurn, Td1Turn): TD(lambda) / TD(1) returns with rolling
// gamma take the same route -- C3-shaped TD(lambda) 11.1 -> 8.8 us, [4096x136x7] 18.3 -> 13.3, fp64 C3 17.6 -> 12.7 against the
// generic transposing kernel that scan_impl.h held until then; V-trace followed (VtraceTurn: its outputs need the next step's x and
// value, read in memory order I elements further) and that kernel is gone.
#pragma once
#include "scan_impl.h"

namespace {

template <typename TS_, typename TC_, int V_>
struct GaePreparedOp {  // what scan_chunk needs of an estimator whose per-step maps are already formed
    typedef TS_ TS;
    typedef TC_ TC;
    static constexpr int V = V_;
    static constexpr bool kNeedsNext = false;
    TC delta[V], disc[V];
    __device__ __forceinline__ TC init(int) const { return (TC)0; }
    __device__ __forceinline__ TC step(int e, TC x) const { return delta[e] + x * disc[e]; }
    __device__ __forceinline__ void affine(int e, TC &a, TC &b) const { a = disc[e]; b = delta[e]; }
    __device__ __forceinline__ TS first_value() const { return (TS)0; }
};

// ---- what the kernel needs of an estimator -----------------------------------------------------------------------------
// The kernel forms the per-step maps x_t = delta_t + disc_t * x_{t+1} in MEMORY order and only those two arrays cross LDS; a
// policy names the value tensors to load, forms (delta, disc) of one element from them and the element's two mask bytes, and
// says what is written.  GAE: the reference's own per-step expression (the replay delta + x*disc IS its recurrence).
template <typename TS_, typename TG_>
struct GaeTurn {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef TS TO;
    typedef GaeArgs Args;
    static constexpr bool kSecondOutput = true;  // value_target = advantage + the state value the lane still holds
    static constexpr bool kNeedsNext = false, kPrefetch = true;
    template <int V>
    struct Raw {
        TS r[V], v[V], nv[V];
        TG g[V], lg[V];
    };
    const TS *__restrict__ rp, *__restrict__ vp, *__restrict__ nvp;
    const TG *__restrict__ gtp, *__restrict__ lgp;
    TG gs, ls, lgs;
    __device__ __forceinline__ explicit GaeTurn(const Args &a)
        : rp((const TS *)a.reward), vp((const TS *)a.value), nvp((const TS *)a.next_value), gtp((const TG *)a.gamma_t),
          lgp((const TG *)a.lg_t), gs((TG)a.gamma), ls((TG)a.lmbda), lgs((TG)a.lg) {}
    template <bool VEC, int V>
    __device__ __forceinline__ void load(i64 flat, int nvalid, Raw<V> &w) const {
        load_n<VEC>(rp, flat, nvalid, w.r);
        load_n<VEC>(vp, flat, nvalid, w.v);
        load_n<VEC>(nvp, flat, nvalid, w.nv);
        if (gtp) load_n<VEC>(gtp, flat, nvalid, w.g);
        if (lgp) load_n<VEC>(lgp, flat, nvalid, w.lg);
    }
    template <int V>
    struct Keep { TS v[V]; };  // what the memory-order epilogue still needs of the inputs
    template <int V>
    __device__ __forceinline__ void pre(const Raw<V> &w, const u8 (&d)[V], const u8 (&tm)[V], unsigned, TC (&delta)[V], TC (&disc)[V],
                                        Keep<V> &k) const {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const TG ge = gtp ? w.g[e] : gs;
            const TG ce = lgp ? w.lg[e] : (gtp ? ls * ge : lgs);
            const TG g_nt = ge * (TG)(tm[e] ? 0 : 1);  // multiply, not select: as the reference
            disc[e] = (TC)(ce * (TG)(d[e] ? 0 : 1));
            delta[e] = ((TC)w.r[e] + (TC)g_nt * (TC)w.nv[e]) - (TC)w.v[e];
            k.v[e] = w.v[e];
        }
    }
    template <int V>
    __device__ __forceinline__ void post(const Keep<V> &k, int e, TS x, TC, TS, bool, TS &o0, TS &o1) const {
        o0 = x;
        o1 = x + k.v[e];  // value_target = advantage + state value
    }
    __device__ __forceinline__ TS value_at(i64) const { return (TS)0; }
    static __device__ __forceinline__ TS *out0(const Args &a) { return (TS *)a.advantage; }
    static __device__ __forceinline__ TS *out1(const Args &a) { return (TS *)a.value_target; }
};

// TD(lambda) return, rolling gamma (scan_core.h TdLambdaOp: the same sub-expressions).  The maps are TdLambdaOp::affine; the
// step that ends a row starts from the masked next value of that step, which folds into a constant map there: delta =
// step(init), disc = 0.  The replay delta + x*disc is the recurrence re-associated: results differ from the exact-step kernels
// by rounding only (SCAN bar); done positions stay bit-exact (disc = 0 there and delta is the reference's expression).
template <typename TS_, typename TG_>
struct TdLambdaTurn {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef TS TO;
    typedef TdLambdaArgs Args;
    static constexpr bool kSecondOutput = false, kNeedsNext = false, kPrefetch = true;
    template <int V>
    struct Raw {
        TS r[V], nv[V];
        TG g[V], l[V];
    };
    const TS *__restrict__ rp, *__restrict__ nvp;
    const TG *__restrict__ gtp, *__restrict__ ltp;
    TG gs, ls;
    __device__ __forceinline__ explicit TdLambdaTurn(const Args &a)
        : rp((const TS *)a.reward), nvp((const TS *)a.next_value), gtp((const TG *)a.gamma_t), ltp((const TG *)a.lmbda_t),
          gs((TG)a.gamma), ls((TG)a.lmbda) {}
    template <bool VEC, int V>
    __device__ __forceinline__ void load(i64 flat, int nvalid, Raw<V> &w) const {
        load_n<VEC>(rp, flat, nvalid, w.r);
        load_n<VEC>(nvp, flat, nvalid, w.nv);
        if (gtp) load_n<VEC>(gtp, flat, nvalid, w.g);
        if (ltp) load_n<VEC>(ltp, flat, nvalid, w.l);
    }
    template <int V>
    struct Keep {};
    template <int V>
    __device__ __forceinline__ void pre(const Raw<V> &w, const u8 (&d)[V], const u8 (&tm)[V], unsigned ends, TC (&delta)[V], TC (&disc)[V],
                                        Keep<V> &) const {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const bool row_end = (ends >> e) & 1u;
            const TG gm = gtp ? w.g[e] : gs, lm = ltp ? w.l[e] : ls;
            const TS nvm = w.nv[e] * (TS)(tm[e] ? 0 : 1), dn = (TS)(d[e] ? 1 : 0), omd = (TS)(d[e] ? 0 : 1);
            const TG oml = (TG)1 - lm;
            const TC g1 = row_end ? (TC)nvm * (TC)omd + (TC)(nvm * dn) : (TC)(nvm * dn);  // row end: step(init), init = nvm
            delta[e] = (TC)w.r[e] + (TC)gm * ((TC)oml * (TC)nvm + (TC)lm * g1);
            disc[e] = row_end ? (TC)0 : ((TC)gm * (TC)lm) * (TC)omd;
        }
    }
    template <int V>
    __device__ __forceinline__ void post(const Keep<V> &, int, TS x, TC, TS, bool, TS &o0, TS &) const { o0 = x; }
    __device__ __forceinline__ TS value_at(i64) const { return (TS)0; }
    static __device__ __forceinline__ TS *out0(const Args &a) { return (TS *)a.returns; }
    static __device__ __forceinline__ TS *out1(const Args &) { return nullptr; }
};

// TD(1) return, rolling gamma (scan_core.h Td1Op): starts from the RAW next value of the row's last step.
template <typename TS_, typename TG_>
struct Td1Turn {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef TS TO;
    typedef TdLambdaArgs Args;
    static constexpr bool kSecondOutput = false, kNeedsNext = false, kPrefetch = true;
    template <int V>
    struct Raw {
        TS r[V], nv[V];
        TG g[V];
    };
    const TS *__restrict__ rp, *__restrict__ nvp;
    const TG *__restrict__ gtp;
    TG gs;
    __device__ __forceinline__ explicit Td1Turn(const Args &a)
        : rp((const TS *)a.reward), nvp((const TS *)a.next_value), gtp((const TG *)a.gamma_t), gs((TG)a.gamma) {}
    template <bool VEC, int V>
    __device__ __forceinline__ void load(i64 flat, int nvalid, Raw<V> &w) const {
        load_n<VEC>(rp, flat, nvalid, w.r);
        load_n<VEC>(nvp, flat, nvalid, w.nv);
        if (gtp) load_n<VEC>(gtp, flat, nvalid, w.g);
    }
    template <int V>
    struct Keep {};
    template <int V>
    __device__ __forceinline__ void pre(const Raw<V> &w, const u8 (&d)[V], const u8 (&tm)[V], unsigned ends, TC (&delta)[V], TC (&disc)[V],
                                        Keep<V> &) const {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const bool row_end = (ends >> e) & 1u;
            const TG gnt = (gtp ? w.g[e] : gs) * (TG)(tm[e] ? 0 : 1);
            const bool b = d[e] && !tm[e];
            const TS dnt = (TS)(b ? 1 : 0), omdnt = (TS)(b ? 0 : 1);
            const TC tail = row_end ? (TC)omdnt * (TC)w.nv[e] + (TC)(dnt * w.nv[e]) : (TC)(dnt * w.nv[e]);  // row end: step(init), init = nv
            delta[e] = (TC)w.r[e] + (TC)gnt * tail;
            disc[e] = row_end ? (TC)0 : (TC)gnt * (TC)omdnt;
        }
    }
    template <int V>
    __device__ __forceinline__ void post(const Keep<V> &, int, TS x, TC, TS, bool, TS &o0, TS &) const { o0 = x; }
    __device__ __forceinline__ TS value_at(i64) const { return (TS)0; }
    static __device__ __forceinline__ TS *out0(const Args &a) { return (TS *)a.returns; }
    static __device__ __forceinline__ TS *out1(const Args &) { return nullptr; }
};

// V-trace (scan_core.h VtraceOp: the same sub-expressions).  x is the scanned correction vs - v; the outputs need the NEXT step's
// vs = x + v: in memory order that is the element I positions further -- its x from the LDS array the row phase just filled, its
// state value re-read from global memory (a line this block loaded moments ago).  The shift ignores done (functional.py:1343-1345):
// only the step that ends a row takes next_value instead.
template <typename TS_, typename TG_>
struct VtraceTurn {
    typedef TS_ TS;
    typedef TG_ TG;
    typedef typename Promote<TS, TG>::type TC;
    typedef TC TO;  // as VtraceOp: outputs in the promoted dtype
    typedef VtraceArgs Args;
    static constexpr bool kSecondOutput = true, kNeedsNext = true;
    static constexpr bool kPrefetch = false;  // six input arrays: the next chunk's would not fit the register file beside this one's
    template <int V>
    struct Raw {
        TS lp[V], lmu[V], v[V], nv[V], r[V];
        TG g[V];
    };
    template <int V>
    struct Keep {
        TS v[V], nv[V], r[V], crho[V];
        TG td[V];
    };
    const TS *__restrict__ lpp, *__restrict__ lmp, *__restrict__ vp, *__restrict__ nvp, *__restrict__ rp;
    const TG *__restrict__ gtp;
    TG gs;
    TS rho_hi, c_hi;
    __device__ __forceinline__ explicit VtraceTurn(const Args &a)
        : lpp((const TS *)a.log_pi), lmp((const TS *)a.log_mu), vp((const TS *)a.value), nvp((const TS *)a.next_value),
          rp((const TS *)a.reward), gtp((const TG *)a.gamma_t), gs((TG)a.gamma), rho_hi((TS)a.rho_thresh), c_hi((TS)a.c_thresh) {}
    template <bool VEC, int V>
    __device__ __forceinline__ void load(i64 flat, int nvalid, Raw<V> &w) const {
        load_n<VEC>(lpp, flat, nvalid, w.lp);
        load_n<VEC>(lmp, flat, nvalid, w.lmu);
        load_n<VEC>(vp, flat, nvalid, w.v);
        load_n<VEC>(nvp, flat, nvalid, w.nv);
        load_n<VEC>(rp, flat, nvalid, w.r);
        if (gtp) load_n<VEC>(gtp, flat, nvalid, w.g);
    }
    template <int V>
    __device__ __forceinline__ void pre(const Raw<V> &w, const u8 (&d)[V], const u8 (&tm)[V], unsigned, TC (&delta)[V], TC (&disc)[V],
                                        Keep<V> &k) const {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const TG ge = gtp ? w.g[e] : gs;
            const TG dd = ge * (TG)(d[e] ? 0 : 1);
            k.td[e] = ge * (TG)(tm[e] ? 0 : 1);
            const TS rho = exp_t(w.lp[e] - w.lmu[e]);
            k.crho[e] = clamp_max_nan(rho, rho_hi);
            const TS cc = clamp_max_nan(rho, c_hi);
            k.v[e] = w.v[e];
            k.nv[e] = w.nv[e];
            k.r[e] = w.r[e];
            delta[e] = (TC)k.crho[e] * (((TC)w.r[e] + (TC)k.td[e] * (TC)w.nv[e]) - (TC)w.v[e]);
            disc[e] = (TC)dd * (TC)cc;
        }
    }
    template <int V>
    __device__ __forceinline__ void post(const Keep<V> &k, int e, TO x, TC x_next, TS v_next, bool row_end, TO &adv, TO &vs) const {
        const TC nxt = row_end ? (TC)k.nv[e] : x_next + (TC)v_next;
        adv = (TC)k.crho[e] * (((TC)k.r[e] + (TC)k.td[e] * nxt) - (TC)k.v[e]);
        vs = x + (TC)k.v[e];
    }
    __device__ __forceinline__ TS value_at(i64 idx) const { return vp[idx]; }
    static __device__ __forceinline__ TO *out0(const Args &a) { return (TO *)a.advantage; }
    static __device__ __forceinline__ TO *out1(const Args &a) { return (TO *)a.vs; }
};

// 4 blocks of 4 waves per CU and a grid of exactly that many blocks: with the next chunk's loads in flight a lane holds two
// sets of raw inputs, which spills at 5+ blocks per CU (measured 16 us at C3), and a grid above the resident set only queues
// blocks behind the first round (measured: 5/CU 13.9 us, 8/CU 11.1 us, 4/CU 9.8 us).  Two elements per lane (8-byte
// accesses, 64 VGPRs, 6 - 8 blocks per CU, with or without the prefetch) was measured too: 11.2 - 13.5 us at C3, dropped;
// so was eight per lane (half the scans per element, one chunk per block): 11.2 us without, 18 us with the prefetch (spills).
// PMC at C3: 47 % of the LDS-array cycles are bank conflicts (the row gather walks LDS with a lane stride of V*I = 20 words:
// 8 of 32 banks), but the LDS array is busy for only 1.85 us per CU of the 9.6 -- latency, not LDS throughput, bounds it.
// Where the 9.6 us go (variants built with the parts removed): streaming alone (loads, delta, stores; no LDS, no scan) 7.4 us;
// plus both LDS turns 7.8; plus the scan 9.6 -- of which 0.55 us is the scan's cross-wave stage (LDS + barrier: rows of 25
// lanes straddle waves) and the rest VALU issue of the DPP steps (about 150 instructions per wave and chunk).
// A variant with every row in its own aligned power-of-two lane group (no row crosses a wave: no cross-wave stage, no
// hard-reset flag; 5- or 8-wave blocks, 25 of 32 lanes busy in the row phase) was built and measured: 11.4 - 15.7 us at C3
// against 9.5 -- the idle lanes and the fewer resident chunks cost more than the stage it removes.  Dropped.
constexpr int kGilBlocksPerCu = 4;  // resident blocks per CU = the grid's blocks per CU (measured above)

// VR > 0: the row phase gives every (o, i) row LG = 16 lanes (one DPP row; LG = 32: two, chained by one shuffle) and every lane
// VR consecutive steps (T <= LG * VR, at most 256 / LG rows per chunk): a row never leaves its DPP row, so the scan is four row shifts -- no hard-reset flags, no cross-row
// steps, no cross-wave stage and no barrier of its own -- at the price of idle lanes in that phase (C3: 10 rows = 160 of 256
// lanes, 100 of 112 steps).  The memory-order phases (V elements per lane, 16-byte accesses) are the same.  Measured: C3 9.55 ->
// 8.49 us back-to-back, 10.45 -> 9.28 us rocprofv3 average; [4096x112x3] 9.16 -> 8.08; fp64 C3 17.65 -> 16.6; LG = 32 (128 < T <=
// 256): [4096x200x5] 19.4 -> 17.8, [4096x160x3] 9.55 -> 8.2, [4096x136x7] 18.7 -> 16.9, fp64 [4096x160x3] 16.5 -> 15.4.
// NT: threads per block.  256 (a team, 4 blocks per CU), or 1024 with a chunk of 1024 * V elements (round 3, row mapping only):
// whole outer rows of up to 4096 fp32 elements -- MAPPO-shaped rollouts with 11 - 40 agents, which the lock-step kernels served
// at 1.1 - 3.6 TB/s ([4096x100x30] 238 us, [4096x100x11] 59.5 us, [4096x100x20] 57.0 us).
template <class P, int V, int VR, int LG, bool VEC, bool WAVE_INDEP, bool PAD, int NT = kTeam>
__global__ __launch_bounds__(NT, kGilBlocksPerCu) void gae_interleaved_kernel(typename P::Args args, ScanGeom geo, int I, double *partials) {
    typedef typename P::TS TS;
    typedef typename P::TC TC;
    typedef typename P::TO TO;  // what the row phase leaves in LDS and the epilogue writes (V-trace: the promoted dtype)
    typedef GaePreparedOp<TS, TC, V> Op;
    static_assert(NT == kTeam || VR > 0, "blocks wider than a team exist for the row mapping only");
    constexpr int CH = NT * V;
    constexpr int CHP = CH + CH / 32;
    auto pad = [&](int p) __attribute__((always_inline)) { return PAD ? p + ((p >> 7) << 2) : p; };  // scan_impl.h: bank-conflict padding, 16-byte accesses kept
    __shared__ __align__(16) TC ld_d[CHP];
    __shared__ __align__(16) TC ld_c[CHP];
    __shared__ __align__(16) TO ld_o[CHP];
    __shared__ TeamShared<Op> sh;
    __shared__ double red[2 * (NT / kWave)];
    const int lane_off = (int)threadIdx.x * V;
    const unsigned T = (unsigned)geo.T;
    const i64 row_elems = geo.T * I;

    // in-chunk offset of each of the lane's V elements once the chunk is seen as rows (o, i) of T steps; mq: the step
    // (o, t) an element of the lane's MEMORY-order slice belongs to (masks shared by the I feature elements of a step)
    int pe[V], mq[V];
    unsigned ends = 0u;  // bit e: element e of the lane's MEMORY-order slice is the last step of its row
    {
        unsigned rv = (unsigned)lane_off / T, t = (unsigned)lane_off % T;
        unsigned ol = rv / (unsigned)I, i = rv % (unsigned)I;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            pe[e] = pad((int)((ol * T + t) * (unsigned)I + i));
            mq[e] = (int)((unsigned)(lane_off + e) / (unsigned)I);
            if ((unsigned)mq[e] % T == T - 1u) ends |= 1u << e;
            if (++t == T) { t = 0; if (++i == (unsigned)I) { i = 0; ++ol; } }
        }
    }
    const unsigned t0 = (unsigned)lane_off % T;
    constexpr int NR = VR > 0 ? VR : 1;
    int pr[NR];  // VR > 0: LDS offsets of the lane's steps in the row phase (-1: past the row's end)
    if constexpr (VR > 0) {
        const unsigned g = threadIdx.x / LG, j = threadIdx.x % LG;
        const unsigned ol = g / (unsigned)I, i = g % (unsigned)I;
#pragma unroll
        for (int e = 0; e < VR; ++e) {
            const unsigned t = j * VR + e;
            pr[e] = t < T ? pad((int)((ol * T + t) * (unsigned)I + i)) : -1;
        }
    }
    const bool bcast = args.mask_bcast != 0;
    const P pol(args);
    struct Raw {
        typename P::template Raw<V> val;
        u8 d[V], tm[V];
    };
    auto geom_of = [&](i64 it, i64 &row0, i64 &start, int &count) __attribute__((always_inline)) {
        const i64 c = it * gridDim.x + blockIdx.x;
        row0 = c * geo.rows_per_chunk;
        i64 nrows = c < geo.nchunks ? geo.O - row0 : 0;
        if (nrows > geo.rows_per_chunk) nrows = geo.rows_per_chunk;
        if (nrows < 0) nrows = 0;
        start = row0 * row_elems;
        count = (int)(nrows * row_elems);
    };
    // (always_inline: left to the inliner, the two call sites compiled differently from one translation unit to another -- C3 8.7 vs 9.9 us)
    auto load_raw = [&](i64 it, Raw &w) __attribute__((always_inline)) {
        i64 row0, start;
        int count;
        geom_of(it, row0, start, count);
        const int nvalid = clamp_nvalid(count, lane_off, V);
        const i64 flat = start + lane_off;
        pol.template load<VEC>(flat, nvalid, w.val);
        if (bcast) {
            const i64 mstart = row0 * geo.T;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const bool ok = e < nvalid;
                w.d[e] = ok ? args.done[mstart + mq[e]] : (u8)1;
                w.tm[e] = ok ? args.terminated[mstart + mq[e]] : (u8)1;
            }
        } else {
            load_n<VEC>(args.done, flat, nvalid, w.d);
            load_n<VEC>(args.terminated, flat, nvalid, w.tm);
        }
    };

    double s = 0.0, ss = 0.0;
    const i64 niter = (geo.nchunks + gridDim.x - 1) / gridDim.x;
    Raw cur, nxt;
    if constexpr (P::kPrefetch) load_raw(0, cur);
    for (i64 it = 0; it < niter; ++it) {
        if constexpr (P::kPrefetch) {
            if (it + 1 < niter) load_raw(it + 1, nxt);  // the next chunk's loads fly while this one goes through LDS
        } else {
            load_raw(it, cur);
        }
        i64 row0, start;
        int count;
        geom_of(it, row0, start, count);
        const int nvalid = clamp_nvalid(count, lane_off, V);
        const i64 flat = start + lane_off;
        typename P::template Keep<V> keep;  // what the memory-order epilogue still needs of the inputs (GAE: the state value)
        TC delta[V], disc[V];
        pol.pre(cur.val, cur.d, cur.tm, ends, delta, disc, keep);
        {
            TC *dd = (TC *)__builtin_assume_aligned(ld_d + pad(lane_off), 16);
            TC *dc = (TC *)__builtin_assume_aligned(ld_c + pad(lane_off), 16);
#pragma unroll
            for (int e = 0; e < V; ++e) { dd[e] = delta[e]; dc[e] = disc[e]; }
        }
        __syncthreads();
        // ---- row order: gather, scan, scatter the advantage ------------------------------------------------------------
        if constexpr (VR > 0) {
            const bool row_ok = (int)(threadIdx.x / LG) * (int)T < count;  // this lane group holds one of the chunk's nrows * I rows
            const int j = threadIdx.x % LG, jr = threadIdx.x & 15;
            TC dl[VR], dc[VR];
#pragma unroll
            for (int e = 0; e < VR; ++e) {
                const bool ok = row_ok && pr[e] >= 0;
                dl[e] = ok ? ld_d[pr[e]] : (TC)0;  // steps past the row's end map everything to 0 = the value after the last step
                dc[e] = ok ? ld_c[pr[e]] : (TC)0;
            }
            Aff<TC> S{dc[VR - 1], dl[VR - 1], 0};
#pragma unroll
            for (int e = VR - 2; e >= 0; --e) S = compose_t<false>(Aff<TC>{dc[e], dl[e], 0}, S);
#define TRL_ROW_STEP(OFF)                                                                  \
    {                                                                                      \
        const Aff<TC> p{dpp_mov<0x100 + OFF>(S.a), dpp_mov<0x100 + OFF>(S.b), 0};          \
        if (jr + OFF < 16) S = compose_t<false>(S, p);                                     \
    }
            TRL_ROW_STEP(1)
            TRL_ROW_STEP(2)
            TRL_ROW_STEP(4)
            TRL_ROW_STEP(8)
#undef TRL_ROW_STEP
            if constexpr (LG == 32) {  // two DPP rows per (o, i) row: the first absorbs the second's total (its lane 0)
                const int src = (threadIdx.x & 32) | 16;
                const Aff<TC> p{__shfl(S.a, src, kWave), __shfl(S.b, src, kWave), 0};
                if (j < 16) S = compose_t<false>(S, p);
            }
            TC x = dpp_mov<0x130>(S.b);  // everything after this lane applied to 0 (whole-wave shift by one lane) ...
            if (j == LG - 1) x = (TC)0;  // ... the row's last lane: nothing follows
#pragma unroll
            for (int e = VR - 1; e >= 0; --e) {
                x = dl[e] + x * dc[e];
                if (row_ok && pr[e] >= 0) {
                    const TO a = (TO)x;
                    ld_o[pr[e]] = a;
                    s += (double)a;
                    ss += (double)a * (double)a;
                }
            }
        } else if constexpr (NT == kTeam) {
        Op op;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            op.delta[e] = e < nvalid ? ld_d[pe[e]] : (TC)0;
            op.disc[e] = e < nvalid ? ld_c[pe[e]] : (TC)0;
        }
        TC Xc = (TC)0;
        TS Xv = (TS)0;
        scan_chunk<Op, WAVE_INDEP>(op, sh, (int)(it & 1), nvalid, t0, T, geo.scan_lanes, Xc, Xv,
                                   [&](const TC(&xs)[V], const bool(&)[V], TC, TS) {
#pragma unroll
                                       for (int e = 0; e < V; ++e) {
                                           if (e < nvalid) {
                                               const TO a = (TO)xs[e];
                                               ld_o[pe[e]] = a;
                                               s += (double)a;
                                               ss += (double)a * (double)a;
                                           }
                                       }
                                   });
        }
        __syncthreads();
        // ---- memory order again: (GAE) value_target = advantage + v, coalesced stores --------------------------------------
        {
            TO o0[V], o1[V];
            const TO *src = (const TO *)__builtin_assume_aligned(ld_o + pad(lane_off), 16);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const bool row_end = (ends >> e) & 1u;
                TC x_next = (TC)0;
                TS v_next = (TS)0;
                if constexpr (P::kNeedsNext) {  // the same (o, i) row's next step sits I elements further in memory order
                    if (e < nvalid && !row_end) {
                        x_next = (TC)ld_o[pad(lane_off + e + I)];
                        v_next = pol.value_at(flat + e + I);
                    }
                }
                o1[e] = (TO)0;
                pol.post(keep, e, src[e], x_next, v_next, row_end, o0[e], o1[e]);
            }
            store_n<VEC>(P::out0(args), flat, nvalid, o0);
            if constexpr (P::kSecondOutput) store_n<VEC>(P::out1(args), flat, nvalid, o1);
        }
        if constexpr (P::kPrefetch) cur = nxt;
    }
    if (partials) {
        block_sum2<NT>(s, ss, red);
        if (threadIdx.x == 0) {
            partials[2 * (i64)blockIdx.x] = s;
            partials[2 * (i64)blockIdx.x + 1] = ss;
        }
    }
}

template <class P, int V, int NT = kTeam>
LaunchResult launch_turn(hipStream_t s, const typename P::Args &a, const Plan &p, int cus, double *partials, const char *name) {
    typedef typename P::TS TS;
    typedef typename P::TG TG;
    constexpr int kPerCu = NT == kTeam ? kGilBlocksPerCu : 1;  // resident blocks per CU (16 waves either way)
    ScanGeom geo{a.O, a.T, p.rows_per_chunk, scan_lanes_for(a.T, V, p.wave_indep), p.nchunks};
    i64 nb = p.nchunks;
    if (nb > (i64)cus * kPerCu) nb = (i64)cus * kPerCu;
    if (nb > 4096) nb = 4096;  // per-block partials: scan_max_partials
    typedef typename Promote<TS, TG>::type TC;
    // rows as 16- / 32-lane groups (VR steps per lane) when T does not tile a wave anyway: 64 < T <= 128 / 128 < T <= 256
    const i64 rows = (i64)p.rows_per_chunk * a.I;
    int lg = 0;
    if (!p.wave_indep && a.T > 64 && a.T <= 128) lg = 16;
    if (!p.wave_indep && a.T > 128 && a.T <= 256) lg = 32;
    const int vr = lg ? (int)((a.T + lg - 1) / lg) : 0;
    const bool rowmap = lg != 0 && rows <= NT / lg && vr >= 5 && vr <= 8;
    if (NT != kTeam && !rowmap) return LaunchResult{2, 0, nullptr};
    const i64 stride_words = (i64)(rowmap ? vr : V) * a.I * (i64)(sizeof(TC) / 4);
    const bool pad_lds = stride_words % 8 == 0;
    const dim3 grid((unsigned)nb);
#define TRL_GIL(VR, LG, VEC, WI)                                                                                                \
    do {                                                                                                                        \
        if (pad_lds) hipLaunchKernelGGL((gae_interleaved_kernel<P, V, VR, LG, VEC, WI, true, NT>), grid, dim3(NT), 0, s, a, geo, (int)a.I, partials); \
        else hipLaunchKernelGGL((gae_interleaved_kernel<P, V, VR, LG, VEC, WI, false, NT>), grid, dim3(NT), 0, s, a, geo, (int)a.I, partials);         \
    } while (0)
#define TRL_GIL_ROW(VR, LG) { if (p.vec) TRL_GIL(VR, LG, true, false); else TRL_GIL(VR, LG, false, false); }
#define TRL_GIL_ROWS(LG)                          \
    switch (vr) {                                 \
        case 5: TRL_GIL_ROW(5, LG) break;         \
        case 6: TRL_GIL_ROW(6, LG) break;         \
        case 7: TRL_GIL_ROW(7, LG) break;         \
        default: TRL_GIL_ROW(8, LG) break;        \
    }
    if (rowmap && lg == 16) { TRL_GIL_ROWS(16) }
    else if (rowmap) { TRL_GIL_ROWS(32) }
    else if constexpr (NT == kTeam) {
        if (p.wave_indep) { if (p.vec) TRL_GIL(0, 16, true, true); else TRL_GIL(0, 16, false, true); }
        else              { if (p.vec) TRL_GIL(0, 16, true, false); else TRL_GIL(0, 16, false, false); }
    }
#undef TRL_GIL_ROWS
#undef TRL_GIL_ROW
#undef TRL_GIL
    return LaunchResult{0, nb, name};
}

// Outer rows longer than a team's chunk on 1024-thread blocks: T*I <= 1024*V, row mapping only (64 < T <= 256).
constexpr int kTurnBigThreads = 1024;
inline bool turn_big_plan(i64 O, i64 T, i64 I, int V, bool ptr_align_ok, Plan &p) {
    const i64 chunk = (i64)kTurnBigThreads * V;
    if (I < 2 || T * I > chunk || T * I <= (i64)kTeam * V || T <= 64 || T > 256) return false;
    // I = 16 stays with the lock-step kernels, whose lanes read one 16-byte vector per tensor there (G = 4 groups of 4): lock-step /
    // this kernel, us: [4096x100x16] 30.3 / 34.3, [4096x180x16] 45.7 / 59.7, [4096x256x16] 59.7 / 65.6, fp64 [4096x100x16] 56.7 / 60.9.
    // Everything else measured goes the other way: [4096x100x30] 238 / 60.3, [4096x100x11] 59.5 / 26.9, [4096x120x13] 74.9 / 34.0,
    // [4096x200x20] 131.7 / 70.5, [4096x100x20] 57.0 / 37.5, [4096x100x14] 44.1 / 31.4, [4096x100x28] 79.9 / 58.4, [4096x150x8]
    // 33.0 / 27.1, fp64 [4096x100x8] 50.6 / 33.9; ties: [4096x100x24] 57.2 / 56.1, [4096x200x8] 37.0 / 37.4, [4096x250x8] 39.9 / 39.6.
    if (I == 16) return false;
    const i64 groups = kTurnBigThreads / (T <= 128 ? 16 : 32);  // (o, i) rows a block's lanes can take
    i64 R = chunk / (T * I);
    const i64 q = V / gcd_i64(T * I, V);
    if (R >= q) R -= R % q;
    if (R * I > groups) R = groups / I;
    if (R < 1) return false;
    if (R > O) R = O;
    p = Plan{};
    p.ok = p.interleaved = true;
    p.rows_per_chunk = (int)R;
    p.nchunks = (O + R - 1) / R;
    p.nblocks = p.nchunks;
    p.vec = ptr_align_ok && (p.nchunks == 1 || (R * T * I) % V == 0);
    return true;
}

// dtype dispatch of one policy: the team-sized kernel when whole outer rows fit its chunk, else the 1024-thread one; status 2 =
// not this kernel's shape.  Instantiated once per estimator at the end of scan_kernels.hip.
template <template <class, class> class P, class Args>
LaunchResult dispatch_turn(hipStream_t s, int vdt, int cdt, const Args &a, bool aligned, double *partials, const char *name) {
    const int cus = device_cus();
    auto go = [&](auto tsv, auto tgv, auto vc) -> LaunchResult {
        typedef decltype(tsv) TS;
        typedef decltype(tgv) TG;
        constexpr int V = decltype(vc)::value;
        Plan p = make_plan(a.O, a.T, a.I, V, aligned, cus);
        if (p.ok && p.interleaved) return launch_turn<P<TS, TG>, V>(s, a, p, cus, partials, name);
        if (turn_big_plan(a.O, a.T, a.I, V, aligned, p)) return launch_turn<P<TS, TG>, V, kTurnBigThreads>(s, a, p, cus, partials, name);
        return LaunchResult{2, 0, nullptr};
    };
    // V = 16 bytes of the storage dtype per lane (scan_impl.h dispatch_scan)
    if (vdt == 0 && cdt == 0) return go(float{}, float{}, std::integral_constant<int, 4>{});
    if (vdt == 1 && cdt == 1) return go(double{}, double{}, std::integral_constant<int, 2>{});
    if (vdt == 1 && cdt == 0) return go(double{}, float{}, std::integral_constant<int, 2>{});
    if (vdt == 0 && cdt == 1) return go(float{}, double{}, std::integral_constant<int, 4>{});
    return LaunchResult{1, 0, nullptr};
}

}  // namespace
