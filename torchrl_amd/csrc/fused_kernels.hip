// Single-launch GAE + advantage normalisation ("FUSED" family).
//
// The north-star step -- GAE followed by `(adv - adv.mean()) / adv.std().clamp_min(eps)`
// (reference: torchrl/objectives/value/advantages.py:1436-1463) -- needs the moments of ALL
// advantages before the first normalised value can be written.  Instead of three launches
// (scan + moments, final reduce, scale pass re-reading the advantage from HBM) this kernel keeps
// every advantage in registers across an in-launch all-gather of the per-block moments:
//
//   * one persistent block of 1024 threads per CU (grid = #CUs, all co-resident); a block is
//     4 teams of 4 waves, each team scans K chunks of 256*V time steps exactly like the SCAN
//     family (scan_core.h), with the next chunk's loads issued before the current one is scanned;
//   * value_target is stored as soon as a chunk is done (stores issued before the publish cost
//     nothing; anything issued between publish and gather delays the polls); the advantages stay
//     in registers;
//   * each block publishes its fp64 {sum, sumsq} as two 8-byte granules {value | 8-bit epoch tag}
//     with write-through (agent-scope) stores; wave 0 of every block sweeps all blocks' granules
//     with agent-scope loads until every tag matches (no fences, no atomics, no counters), sums
//     them in a fixed order (bitwise reproducible) and broadcasts mean / std through LDS;
//   * the normalised advantages are written: 22 B of HBM traffic per transition in total, the
//     algorithmic minimum.
//
// Workspace protocol: the first kFusedWsBytes of the workspace hold {epoch, error, slots[]}.
// The caller zero-fills the workspace once; afterwards only this library touches it.  Each
// launch tags its granules with (epoch+1) & 0xff and block 0 bumps the epoch when its own sweep
// has completed (every block has read the epoch by then).  All slots are rewritten by every
// launch (the grid is always #CUs blocks), so a stale tag can never match.  Two launches that
// share a workspace must not overlap (same stream, or external ordering).
// Every spin is bounded (1 s wall clock): on timeout the error word is set and NaNs are written.
#include "scan_core.h"
#include "xchg.h"
#include <cstdlib>

namespace {

using namespace scan;

constexpr int kTeams = 4;
constexpr int kFusedThreads = kTeams * kTeam;  // 1024
constexpr int kMaxBlocks = 512;                // slots available in the workspace header

struct FusedWs {
    unsigned epoch;
    unsigned error;
    unsigned pad[62];
    unsigned long long slots[1];  // kMaxBlocks granule pairs, kSlotStride 8-byte units apart
};
constexpr int kSlotStride = 16;  // one 128-byte line per block: the polls of different blocks do not share a line
constexpr int kSlotReplicas = 4;
static_assert(256 + (size_t)kSlotReplicas * kMaxBlocks * kSlotStride * 8 <= kFusedWsBytes, "workspace header");

struct FusedGeom {
    i64 O, T;
    int rows_per_chunk;
    int scan_lanes;
    i64 nchunks;
    double n_total;
    double eps;
    int slot_stride;  // distance between two blocks' granule pairs, in 8-byte units
};

__device__ __forceinline__ unsigned long long tag_bits(double x, unsigned tag) {
    return ((unsigned long long)__double_as_longlong(x) & ~0xffull) | (unsigned long long)tag;
}
__device__ __forceinline__ double untag(unsigned long long b) { return __longlong_as_double((long long)(b & ~0xffull)); }

// ABL (development only, tools/kbench): 0 = product; 1 = skip the gather (own block's moments);
// 2 = also skip the block reduction; 3 = also skip the wave scan steps; 4 = last chunk's
// value_target stored after the gather; 8 / 9 = value_target stored between publish and gather
// (both slower than storing it at once); 10 = every chunk's loads issued up front.
template <class Op, bool VEC, bool WAVE_INDEP, int K, int ABL = 0, int TEAMS = kTeams>
__global__ __launch_bounds__(TEAMS * kTeam) void gae_fused_kernel(typename Op::Args args, FusedGeom geo, FusedWs *ws,
                                                                  double *stats_out, xchg::Dev xd = xchg::Dev{nullptr, nullptr, 0, 1, 0}) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int V = Op::V;
    constexpr int NW = TEAMS * kTeam / kWave;
    constexpr bool kDeferVt = ABL == 4;  // measured slower (9.6 vs 9.2 us): stores issued before the publish are free
    constexpr bool kVtWindow = ABL == 8 || ABL == 9;  // all value_target stores between publish and gather
    __shared__ TeamShared<Op> sh[TEAMS];
    __shared__ double red[2 * NW];
    __shared__ double bc[3];

    const int team = threadIdx.x / kTeam;
    const int tid = threadIdx.x & (kTeam - 1);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int lane_off = tid * V;
    const unsigned T = (unsigned)geo.T;
    const unsigned tag = (ws->epoch + 1u) & 0xffu;  // written by the previous launch: visible across the kernel boundary
    // batch sharded over GPUs: epoch of the cross-GPU mailbox, read before this block publishes
    const unsigned xepoch = xd.local ? xd.local->epoch + 1u : 0u;
    unsigned long long stamp0 = 0, stamp1 = 0, stamp2 = 0;
    if constexpr (ABL == 7) stamp0 = wall_clock64();

    TS adv_keep[K][V];
    TS vt_last[V];
    TS vt_keep[kVtWindow ? K : 1][V];
    i64 flat_k[K];
    int nvalid_k[K];

    auto chunk_of = [&](int k) { return ((i64)k * gridDim.x + blockIdx.x) * TEAMS + team; };
    auto geom_of = [&](int k, i64 &start, int &count) {
        const i64 c = chunk_of(k);
        const i64 row0 = c * geo.rows_per_chunk;
        i64 nrows = c < geo.nchunks ? geo.O - row0 : 0;
        if (nrows > geo.rows_per_chunk) nrows = geo.rows_per_chunk;
        if (nrows < 0) nrows = 0;
        start = row0 * geo.T;
        count = (int)(nrows * geo.T);
    };
    auto nvalid_of = [&](int count) {
        int nv = count - lane_off;
        return nv < 0 ? 0 : (nv > V ? V : nv);
    };

    double s = 0.0, ss = 0.0;
    constexpr bool kAllLoads = ABL == 10;  // every chunk's loads issued up front
    constexpr int kRaw = kAllLoads ? K : 2;
    typename Op::Raw raw[kRaw];
#pragma unroll
    for (int k = 0; k < (kAllLoads ? K : 1); ++k) {
        i64 start;
        int count;
        geom_of(k, start, count);
        load_inputs<Op, VEC>(args, start + lane_off, nvalid_of(count), raw[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        i64 start;
        int count;
        if (!kAllLoads && k + 1 < K) {
            geom_of(k + 1, start, count);
            load_inputs<Op, VEC>(args, start + lane_off, nvalid_of(count), raw[(k + 1) & 1]);
        }
        geom_of(k, start, count);
        const int nvalid = nvalid_of(count);
        const i64 flat = start + lane_off;
        flat_k[k] = flat;
        nvalid_k[k] = nvalid;
        Op op;
        op.prepare(args, raw[kAllLoads ? k : (k & 1)]);
        TC Xc = (TC)0;
        TS Xv = (TS)0;
        scan_chunk<Op, WAVE_INDEP>(op, sh[team], k & 1, nvalid, (unsigned)lane_off % T, T, (ABL == 3) ? 1 : geo.scan_lanes, Xc, Xv,
                                   [&](const TC(&xs)[V], const bool(&)[V], TC, TS) {
                                       if constexpr (ABL == 5 || ABL == 6) {  // development: plain scan kernel shape
                                           TS vt[V];
                                           op.finish(nvalid, xs, adv_keep[k], vt, s, ss);
                                           store_n<VEC>((TS *)args.value_target, flat, nvalid, vt);
                                           store_n<VEC>((TS *)args.advantage, flat, nvalid, adv_keep[k]);
                                       } else if constexpr (kVtWindow) {
                                           op.finish(nvalid, xs, adv_keep[k], vt_keep[k], s, ss);
                                       } else if (kDeferVt && k == K - 1) {
                                           // the last chunk's value_target is stored after the gather: the
                                           // memory pipes are quiet while the moments are exchanged
                                           op.finish(nvalid, xs, adv_keep[k], vt_last, s, ss);
                                       } else {
                                           TS vt[V];
                                           op.finish(nvalid, xs, adv_keep[k], vt, s, ss);
                                           store_n<VEC>((TS *)args.value_target, flat, nvalid, vt);
                                       }
                                   });
    }

    if constexpr (ABL == 5) return;
    if constexpr (ABL == 6) {  // development: + per-block partials, as the unfused SCAN kernel writes them
        s = wave_sum(s);
        ss = wave_sum(ss);
        if (lane == 0) { red[2 * wv] = s; red[2 * wv + 1] = ss; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double bs = 0.0, bss = 0.0;
            for (int k = 0; k < NW; ++k) { bs += red[2 * k]; bss += red[2 * k + 1]; }
            stats_out[2 * blockIdx.x] = bs;
            stats_out[2 * blockIdx.x + 1] = bss;
        }
        return;
    }
    // ---- block moments -> publish -> gather everybody's --------------------------------------------
    if constexpr (ABL == 0 || ABL == 1 || ABL == 4 || ABL >= 7) {
        s = wave_sum(s);
        ss = wave_sum(ss);
        if (lane == 0) {
            red[2 * wv] = s;
            red[2 * wv + 1] = ss;
        }
        __syncthreads();
    }
    if constexpr (ABL >= 1 && ABL <= 3) {
        if (threadIdx.x == 0) { bc[0] = geo.n_total; bc[1] = s * gridDim.x; bc[2] = ss * gridDim.x + 1.0; }
    } else if (wv == 0) {
        double bs = 0.0, bss = 0.0;  // every lane of wave 0 forms the block totals in the same fixed order
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            bs += red[2 * k];
            bss += red[2 * k + 1];
        }
        if constexpr (ABL == 7) stamp1 = wall_clock64();
        // kSlotReplicas copies of every block's granules (lane r writes copy r): the blocks that poll one line drop
        // from #blocks to #blocks / kSlotReplicas (9.25 -> 9.12 us at C4; 2 / 4 / 8 copies measured, 4 is enough)
        constexpr int kRep = kSlotReplicas;
        const size_t rep_off = (size_t)(blockIdx.x % kRep) * gridDim.x * geo.slot_stride;
        if (lane < kRep) {
            unsigned long long *sl = ws->slots + ((size_t)lane * gridDim.x + blockIdx.x) * geo.slot_stride;
            __hip_atomic_store(sl, tag_bits(bs, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sl + 1, tag_bits(bss, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (ABL == 8) {
#pragma unroll
            for (int k = 0; k < K; ++k) store_n<VEC>((TS *)args.value_target, flat_k[k], nvalid_k[k], vt_keep[k]);
        }
        // lane l gathers blocks l, l+64, ...: only the granules still missing are re-polled
        constexpr int kPer = kMaxBlocks / kWave;
        unsigned long long vs[kPer], vq[kPer];
        bool got[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) got[j] = lane + kWave * j >= (int)gridDim.x;
        const unsigned long long t_start = wall_clock64();
        bool ok;
        for (;;) {
            ok = true;
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                if (!got[j]) {
                    const unsigned long long *sl = ws->slots + rep_off + (size_t)(lane + kWave * j) * geo.slot_stride;
                    vs[j] = __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    vq[j] = __hip_atomic_load(sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
                if (!got[j]) got[j] = ((unsigned)(vs[j] & 0xffull) == tag) && ((unsigned)(vq[j] & 0xffull) == tag);
                ok = ok && got[j];
            }
            if (__all(ok)) break;
            if (wall_clock64() - t_start > 100000000ull) break;  // 1 s at 100 MHz: give up loudly
            __builtin_amdgcn_s_sleep(1);  // sleep 0 / 4 / 10: no gain; two or three poll rounds in flight (also of the missing
                                          // granules only): 16 - 23 us -- extra polls slow the publishes down
        }
        const bool all_ok = __all(ok);
        if constexpr (ABL == 7) stamp2 = wall_clock64();
        double acc = 0.0, acq = 0.0;
#pragma unroll
        for (int j = 0; j < kPer; ++j)
            if (lane + kWave * j < (int)gridDim.x) { acc += untag(vs[j]); acq += untag(vq[j]); }
        acc = wave_sum(acc);
        double tot_ss = wave_sum(acq);
        double tot_n = geo.n_total;
        bool x_ok = true;
        if (xd.local) {
            // second stage, across GPUs: block 0 stores this GPU's totals into every peer's mailbox over
            // xGMI; every block then polls the LOCAL mailbox for all ranks' totals (xchg.h)
            if (blockIdx.x == 0) xchg::send(xd, xepoch, lane, tot_n, acc, tot_ss);
            x_ok = xchg::recv_sum(xd, xepoch, lane, tot_n, acc, tot_ss);
        }
        if (lane == 0) {
            const double nan = __longlong_as_double(0x7ff8000000000000LL);
            const bool good = all_ok && x_ok;
            bc[0] = tot_n;
            bc[1] = good ? acc : nan;
            bc[2] = good ? tot_ss : nan;
            if (!all_ok) ws->error = 1u;
            if (!x_ok) xd.local->error = 1u;
            if (blockIdx.x == 0) {
                ws->epoch = ws->epoch + 1u;  // every block has read the old epoch: all of them have published
                if (xd.local) xd.local->epoch = xepoch;  // likewise: every block read it before publishing
                if (stats_out && ABL != 7) {
                    stats_out[0] = tot_n;
                    stats_out[1] = bc[1];
                    stats_out[2] = bc[2];
                }
            }
        }
    }
    __syncthreads();
    TS mean_s, std_s;
    mean_std<TS>(bc[0], bc[1], bc[2], geo.eps, mean_s, std_s);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        TS out[V];
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = (adv_keep[k][e] - mean_s) / std_s;
        store_n<VEC>((TS *)args.advantage, flat_k[k], nvalid_k[k], out);
    }
    if constexpr (kDeferVt && !kVtWindow) store_n<VEC>((TS *)args.value_target, flat_k[K - 1], nvalid_k[K - 1], vt_last);
    if constexpr (ABL == 7) {
        if (threadIdx.x == 0) {
            unsigned long long *dbg = (unsigned long long *)stats_out + 4 * blockIdx.x;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg[0] = stamp0; dbg[1] = stamp1; dbg[2] = stamp2; dbg[3] = wall_clock64();
        }
    }
}

i64 gcd_i64(i64 a, i64 b) { while (b) { i64 t = a % b; a = b; b = t; } return a; }
inline bool aligned16(const void *p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; }

int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cached[dev] = n;
    }
    return cached[dev];
}

template <class Op>
LaunchResult launch_fused_t(hipStream_t s, const GaeArgs &a, double eps, void *ws, double *stats_out, const xchg::Dev &xd) {
    constexpr int V = Op::V;
    const i64 chunk = (i64)kTeam * V;
    if (a.I != 1 || a.T > chunk) return LaunchResult{2, 0, nullptr};
    int G = device_cus();
    if (G <= 0) return LaunchResult{2, 0, nullptr};
    // rehearsal hook: ranks that share one GPU (tools/xchg_check.py) must all be co-resident
    static const int grid_cap = getenv("TRL_FUSED_GRID_CAP") ? atoi(getenv("TRL_FUSED_GRID_CAP")) : 0;
    if (grid_cap > 0 && G > grid_cap) G = grid_cap;
    if (G > kMaxBlocks) G = kMaxBlocks;
    // R whole rows per chunk, R*T a multiple of V where possible (keeps every chunk start 16-B aligned)
    i64 R = chunk / a.T;
    const i64 q = V / gcd_i64(a.T, V);
    if (R >= q) R -= R % q;
    if (R > a.O) R = a.O;
    const i64 nchunks = (a.O + R - 1) / R;
    const i64 per_iter = (i64)G * kTeams;
    const i64 k_needed = (nchunks + per_iter - 1) / per_iter;
    // the advantages of K chunks stay in registers: K = 8 fits the 128 VGPRs of a 1024-thread block except for fp32 values
    // with fp64 coefficients (spills)
    constexpr bool kMixed = sizeof(typename Op::TS) == 4 && sizeof(typename Op::TC) == 8;
    if (k_needed > (kMixed ? 4 : 8)) return LaunchResult{2, 0, nullptr};
    const bool al = aligned16(a.reward) && (!a.next_value || aligned16(a.value)) && aligned16(a.next_value) && aligned16(a.done) &&
                    aligned16(a.terminated) && aligned16(a.gamma_t) && aligned16(a.lg_t) && aligned16(a.advantage) &&
                    aligned16(a.value_target);
    const bool vec = al && (nchunks == 1 || (R * a.T) % V == 0);
    const bool wi = ((i64)kWave * V) % a.T == 0;
    FusedGeom geo{a.O, a.T, (int)R, scan_lanes_for(a.T, V, wi), nchunks, (double)a.O * (double)a.T, eps, kSlotStride};
    const dim3 grid((unsigned)G), block(kFusedThreads);
    FusedWs *w = (FusedWs *)ws;
#define TRL_FUSED(VEC, WI, KK) hipLaunchKernelGGL((gae_fused_kernel<Op, VEC, WI, KK>), grid, block, 0, s, a, geo, w, stats_out, xd)
#define TRL_FUSED_K(VEC, WI)                                     \
    do {                                                         \
        if (k_needed <= 2) TRL_FUSED(VEC, WI, 2);                \
        else if (k_needed <= 4) TRL_FUSED(VEC, WI, 4);           \
        else if constexpr (!kMixed) TRL_FUSED(VEC, WI, 8);       \
    } while (0)
    if (vec) { if (wi) TRL_FUSED_K(true, true); else TRL_FUSED_K(true, false); }
    else     { if (wi) TRL_FUSED_K(false, true); else TRL_FUSED_K(false, false); }
#undef TRL_FUSED_K
#undef TRL_FUSED
    return LaunchResult{0, 0, "gae_fused"};
}

}  // namespace

LaunchResult launch_gae_fused(hipStream_t s, int vdt, int cdt, const GaeArgs &a, double eps, void *ws, double *stats_out, void *comm) {
    const xchg::Dev xd = xchg_dev_of(comm);
    if (!a.next_value) {  // shifted value layout: its own instantiations (scan_core.h: Shifted<>)
        if (vdt == 0 && cdt == 0) return launch_fused_t<GaeShiftedOp<float, float, 4>>(s, a, eps, ws, stats_out, xd);
        if (vdt == 1 && cdt == 1) return launch_fused_t<GaeShiftedOp<double, double, 2>>(s, a, eps, ws, stats_out, xd);
        if (vdt == 1 && cdt == 0) return launch_fused_t<GaeShiftedOp<double, float, 2>>(s, a, eps, ws, stats_out, xd);
        if (vdt == 0 && cdt == 1) return launch_fused_t<GaeShiftedOp<float, double, 4>>(s, a, eps, ws, stats_out, xd);
        return LaunchResult{1, 0, nullptr};
    }
    if (vdt == 0 && cdt == 0) return launch_fused_t<GaeOp<float, float, 4>>(s, a, eps, ws, stats_out, xd);
    if (vdt == 1 && cdt == 1) return launch_fused_t<GaeOp<double, double, 2>>(s, a, eps, ws, stats_out, xd);
    if (vdt == 1 && cdt == 0) return launch_fused_t<GaeOp<double, float, 2>>(s, a, eps, ws, stats_out, xd);
    if (vdt == 0 && cdt == 1) return launch_fused_t<GaeOp<float, double, 4>>(s, a, eps, ws, stats_out, xd);
    return LaunchResult{1, 0, nullptr};
}
