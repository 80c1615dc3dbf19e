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
//   SHORT rows (T <= chunk): a chunk is R whole rows (R*T <= chunk) and needs no carry.  The
//                            kernel is persistent: ~one block of 4 teams (1024 threads) per CU,
//                            each team walks chunks grid-stride with the next chunk's loads
//                            issued before the current one is scanned, so reads and writes of
//                            neighbouring chunks overlap in time.
//   LONG rows  (T >  chunk): a block (one team) owns whole rows and walks each row's chunks from
//                            the end, carrying x (and V-trace's vs) in registers, again with the
//                            next chunk prefetched.
//
// HBM traffic = algorithmic bytes: each input element is read once, each output written
// once (GAE fp32 22 B / transition); per-block {sum, sumsq} partials add 16 B per block.
#include "scan_core.h"

namespace {

using namespace scan;
constexpr int kTeams = 4;                       // SHORT geometry: teams per block
constexpr int kShortThreads = kTeams * kTeam;   // 1024
constexpr int kLongThreads = kTeam;             // 256

struct ScanGeom {
    i64 O, T;
    int rows_per_chunk;  // SHORT geometry
    int scan_lanes;      // bound on the lanes one row spans (shuffle steps)
    i64 nchunks;         // SHORT geometry
};

__device__ __forceinline__ int clamp_nvalid(int count, int lane_off, int V) {
    const int nv = count - lane_off;
    return nv < 0 ? 0 : (nv > V ? V : nv);
}

// ---- SHORT rows: persistent, 4 teams per block -------------------------------------------------
template <class Op, bool VEC, bool WAVE_INDEP>
__global__ __launch_bounds__(kShortThreads) void scan_short_kernel(typename Op::Args args, ScanGeom geo, double *partials) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int V = Op::V;
    constexpr int NW = kShortThreads / kWave;
    __shared__ TeamShared<Op> sh[kTeams];
    __shared__ double red[2 * NW];
    const int team = threadIdx.x / kTeam;
    const int lane_off = (threadIdx.x & (kTeam - 1)) * V;
    const unsigned T = (unsigned)geo.T;
    const i64 stride = (i64)gridDim.x * kTeams;
    const i64 niter = (geo.nchunks + stride - 1) / stride;  // uniform over the grid (barriers inside scan_chunk)

    auto geom_of = [&](i64 it, i64 &start, int &count) {
        const i64 c = it * stride + (i64)blockIdx.x * kTeams + team;
        const i64 row0 = c * geo.rows_per_chunk;
        i64 nrows = c < geo.nchunks ? geo.O - row0 : 0;
        if (nrows > geo.rows_per_chunk) nrows = geo.rows_per_chunk;
        if (nrows < 0) nrows = 0;
        start = row0 * geo.T;
        count = (int)(nrows * geo.T);
    };

    double s = 0.0, ss = 0.0;
    typename Op::Raw cur, nxt;
    i64 start;
    int count;
    geom_of(0, start, count);
    Op::template load<VEC>(args, start + lane_off, clamp_nvalid(count, lane_off, V), cur);
    for (i64 it = 0; it < niter; ++it) {
        if (it + 1 < niter) {
            i64 nstart;
            int ncount;
            geom_of(it + 1, nstart, ncount);
            Op::template load<VEC>(args, nstart + lane_off, clamp_nvalid(ncount, lane_off, V), nxt);
        }
        geom_of(it, start, count);
        const int nvalid = clamp_nvalid(count, lane_off, V);
        const i64 flat = start + lane_off;
        Op op;
        op.prepare(args, cur);
        TC Xc = (TC)0;
        TS Xv = (TS)0;
        scan_chunk<Op, WAVE_INDEP>(op, sh[team], (int)(it & 1), nvalid, (unsigned)lane_off % T, T, geo.scan_lanes, Xc, Xv,
                                   [&](const TC(&xs)[V], const bool(&row_end)[V], TC x_in, TS v_next) {
                                       op.template store<VEC>(args, flat, nvalid, xs, row_end, x_in, v_next, s, ss);
                                   });
        cur = nxt;
    }
    if (partials) {
        block_sum2<kShortThreads>(s, ss, red);
        if (threadIdx.x == 0) {
            partials[2 * (i64)blockIdx.x] = s;
            partials[2 * (i64)blockIdx.x + 1] = ss;
        }
    }
}

// ---- LONG rows: one team per block walks (row, chunk) pairs from each row's end -----------------
template <class Op, bool VEC>
__global__ __launch_bounds__(kLongThreads) void scan_long_kernel(typename Op::Args args, ScanGeom geo, double *partials) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int V = Op::V;
    constexpr int kChunk = kLongThreads * V;
    __shared__ TeamShared<Op> sh;
    __shared__ double red[2 * kWaves];
    double s = 0.0, ss = 0.0;
    const unsigned T = (unsigned)geo.T;
    const int lane_off = (int)threadIdx.x * V;
    const int nchunks = (int)((geo.T + kChunk - 1) / kChunk);
    auto chunk_count = [&](int cc) {
        i64 cnt = geo.T - (i64)cc * kChunk;
        return (int)(cnt > kChunk ? kChunk : cnt);
    };
    int buf = 0;
    i64 row = blockIdx.x;
    int c = nchunks - 1;
    typename Op::Raw cur, nxt;
    if (row < geo.O)
        Op::template load<VEC>(args, row * geo.T + (i64)c * kChunk + lane_off, clamp_nvalid(chunk_count(c), lane_off, V), cur);
    TC Xc = (TC)0;
    TS Xv = (TS)0;
    while (row < geo.O) {
        i64 nrow = row;
        int nc = c - 1;
        if (nc < 0) { nrow = row + gridDim.x; nc = nchunks - 1; }
        if (nrow < geo.O)
            Op::template load<VEC>(args, nrow * geo.T + (i64)nc * kChunk + lane_off,
                                   clamp_nvalid(chunk_count(nc), lane_off, V), nxt);
        const i64 start = row * geo.T + (i64)c * kChunk;
        const int nvalid = clamp_nvalid(chunk_count(c), lane_off, V);
        const i64 flat = start + lane_off;
        Op op;
        op.prepare(args, cur);
        scan_chunk<Op, false>(op, sh, buf, nvalid, (unsigned)((i64)c * kChunk + lane_off), T, kWave, Xc, Xv,
                              [&](const TC(&xs)[V], const bool(&row_end)[V], TC x_in, TS v_next) {
                                  op.template store<VEC>(args, flat, nvalid, xs, row_end, x_in, v_next, s, ss);
                              });
        buf ^= 1;
        cur = nxt;
        row = nrow;
        c = nc;
    }
    if (partials) {
        block_sum2<kLongThreads>(s, ss, red);
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
    int rows_per_chunk;
    i64 nchunks;
    i64 nblocks;
};

constexpr i64 kMaxLongGrid = 256 * 8;  // LONG geometry: blocks grid-stride over rows
constexpr i64 kMaxShortGrid = 512;     // SHORT geometry: persistent blocks (<= 2 per CU)

i64 gcd_i64(i64 a, i64 b) { while (b) { i64 t = a % b; a = b; b = t; } return a; }

int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// V: elements per lane; ptr_align_ok: every tensor base pointer is 16-byte aligned.
Plan make_plan(i64 O, i64 T, i64 I, int V, bool ptr_align_ok, int cus) {
    Plan p{};
    if (I != 1 || T >= (1LL << 31) - 4096) return p;  // not this family's shape
    const i64 chunk = (i64)kTeam * V;
    p.ok = true;
    if (T <= chunk) {
        // R whole rows per chunk; prefer R*T multiple of V so every chunk start stays aligned
        i64 R = chunk / T;
        const i64 q = V / gcd_i64(T, V);
        if (R >= q) R -= R % q;
        if (R > O) R = O;
        p.rows_per_chunk = (int)R;
        p.nchunks = (O + R - 1) / R;
        i64 nb = (p.nchunks + kTeams - 1) / kTeams;
        if (nb > cus) nb = cus;  // one persistent block per CU; chunks beyond are walked grid-stride
        if (nb > kMaxShortGrid) nb = kMaxShortGrid;
        p.nblocks = nb;
        p.vec = ptr_align_ok && (p.nchunks == 1 || (R * T) % V == 0);
        p.wave_indep = ((i64)kWave * V) % T == 0;
        p.long_rows = false;
    } else {
        p.long_rows = true;
        p.nblocks = O < kMaxLongGrid ? O : kMaxLongGrid;
        p.vec = ptr_align_ok && (T % V == 0);
        p.wave_indep = false;
    }
    return p;
}

template <class Op>
LaunchResult launch_scan(hipStream_t s, const typename Op::Args &a, const Plan &p, i64 O, i64 T, double *partials,
                         const char *name) {
    ScanGeom geo{O, T, p.rows_per_chunk, scan_lanes_for(T, Op::V, p.wave_indep), p.nchunks};
    const dim3 grid((unsigned)p.nblocks);
    if (p.long_rows) {
        if (p.vec) hipLaunchKernelGGL((scan_long_kernel<Op, true>), grid, dim3(kLongThreads), 0, s, a, geo, partials);
        else hipLaunchKernelGGL((scan_long_kernel<Op, false>), grid, dim3(kLongThreads), 0, s, a, geo, partials);
    } else {
#define TRL_LAUNCH(VEC, WI) hipLaunchKernelGGL((scan_short_kernel<Op, VEC, WI>), grid, dim3(kShortThreads), 0, s, a, geo, partials)
        if (p.wave_indep) { if (p.vec) TRL_LAUNCH(true, true); else TRL_LAUNCH(false, true); }
        else              { if (p.vec) TRL_LAUNCH(true, false); else TRL_LAUNCH(false, false); }
#undef TRL_LAUNCH
    }
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
    (void)O; (void)T; (void)I;
    return kMaxLongGrid > kMaxShortGrid ? kMaxLongGrid : kMaxShortGrid;
}

LaunchResult launch_gae_scan(hipStream_t s, int vdt, int cdt, const GaeArgs &a) {
    const bool al = aligned16(a.reward) && aligned16(a.value) && aligned16(a.next_value) && aligned16(a.done) &&
                    aligned16(a.terminated) && aligned16(a.gamma_t) && aligned16(a.lg_t) && aligned16(a.advantage) &&
                    aligned16(a.value_target);
    return dispatch_scan<GaeOp, GaeArgs>(vdt, cdt, [&](auto op) {
        typedef decltype(op) Op;
        Plan p = make_plan(a.O, a.T, a.I, Op::V, al, device_cus());
        if (!p.ok) return LaunchResult{2, 0, nullptr};
        return launch_scan<Op>(s, a, p, a.O, a.T, a.partials, p.long_rows ? "gae_scan_long" : "gae_scan_short");
    });
}

LaunchResult launch_td_lambda_scan(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a) {
    const bool al = aligned16(a.reward) && aligned16(a.next_value) && aligned16(a.done) && aligned16(a.terminated) &&
                    aligned16(a.gamma_t) && aligned16(a.lmbda_t) && aligned16(a.returns);
    return dispatch_scan<TdLambdaOp, TdLambdaArgs>(vdt, cdt, [&](auto op) {
        typedef decltype(op) Op;
        Plan p = make_plan(a.O, a.T, a.I, Op::V, al, device_cus());
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
        Plan p = make_plan(a.O, a.T, a.I, Op::V, al, device_cus());
        if (!p.ok) return LaunchResult{2, 0, nullptr};
        return launch_scan<Op>(s, a, p, a.O, a.T, nullptr, p.long_rows ? "vtrace_scan_long" : "vtrace_scan_short");
    });
}
