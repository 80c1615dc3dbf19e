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
//   * each block publishes its fp64 {sum, sumsq} as ONE 16-byte granule {sum | tag, sumsq | tag} (8-bit epoch tag in
//     the low mantissa bits of each half) with one write-through 16-byte store per replica into a PACKED array
//     (16 bytes per block, 8 replicas); wave 0 of every block sweeps its replica -- 4 KB for 256 blocks, four
//     coalesced 16-byte sc1 loads per lane and round -- until every tag matches (no fences, no atomics, no
//     counters), sums them in a fixed order (bitwise reproducible) and broadcasts mean / std through LDS
//     (round 3: last publish -> last gathered 2.1 -> 1.6 us at C4 against one 128-byte line per block polled with
//     8-byte loads, 7.95 -> 7.45 us per launch; tools/kbench/kfused3.hip, kxchg.hip);
//   * the normalised advantages are written: 22 B of HBM traffic per transition in total, the
//     algorithmic minimum.
//   * both outputs leave with WRITE-THROUGH 16-byte stores (sc1): nothing is left dirty in the L2s, so the
//     end-of-kernel write-back is empty (launch-to-launch 8.6 -> 8.25 us at C4 against non-temporal stores, which
//     stay dirty in L2; plain stores: 11.5 us).
//
// Workspace protocol: the first kFusedWsBytes of the workspace hold {epoch, error, slots[]}.
// The caller zero-fills the workspace once; afterwards only this library touches it.  Each
// launch tags its granules with (epoch+1) & 0xff and block 0 bumps the epoch when its own sweep
// has completed (every block has read the epoch by then).  All slots are rewritten by every
// launch (the grid is always #CUs blocks), so a stale tag can never match; a block that GIVES UP re-tags its own
// granules with the previous launch's tag before it exits, so that nothing it published can match the next launch's
// tag either (blocks that start after block 0 has bumped the epoch publish with the tag the next launch will use).
// Two launches that share a workspace must not overlap (same stream, or external ordering).
// Every spin is bounded (1 s wall clock).  A launch that gives up (its blocks were not co-resident: something else
// held CUs) writes the UN-NORMALISED advantage, sets the workspace's error word and raises the host-visible flag
// (capi.hip: the next call on this GPU returns TRL_ERR_LAUNCH) -- never NaNs, never silently.
#include "scan_core.h"
#include "moment_gather.h"
#include "xchg.h"
#include <cstdlib>

namespace {

using namespace scan;

constexpr int kTeams = 4;
constexpr int kFusedThreads = kTeams * kTeam;  // 1024
constexpr int kMaxBlocks = gather::kMaxBlocks;  // slots available in the workspace header

typedef gather::Header FusedWs;  // {epoch, error, slots}: kSlotReplicas packed arrays of gridDim.x 16-byte granules (moment_gather.h)
constexpr int kSlotReplicas = gather::kHeaderReplicas;   // block b sweeps replica b % 8 (replica counts and layouts measured: moment_gather.h)
static_assert(256 + (size_t)kSlotReplicas * kMaxBlocks * 16 <= kFusedWsBytes, "workspace header");

struct FusedGeom {
    i64 O, T;
    int rows_per_chunk;
    int scan_lanes;
    i64 nchunks;
    double n_total;
    double eps;
    unsigned *host_flag;  // device view of the library's pinned host word (may be null): raised when a launch gives up
    int slot_stride;      // granules per replica in the workspace = the largest grid a launch on it uses (>= gridDim.x)
};

// FULL: every team's K chunks are complete (nchunks == K * grid * kTeams, whole chunks of 256*V elements, aligned):
//       no clipping, every lane holds V valid elements -- the C4 shape class; 0.6 us per launch at [65536 x 32].
// XCHG: the batch is sharded over GPUs; the moments are summed over the ranks inside the launch (xchg.h).
template <class Op, bool VEC, bool WAVE_INDEP, int K, bool FULL, bool XCHG>
__global__ __launch_bounds__(kFusedThreads) void gae_fused_kernel(typename Op::Args args, FusedGeom geo, FusedWs *ws, double *stats_out,
                                                                  xchg::Dev xd) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int V = Op::V;
    constexpr int NW = kFusedThreads / kWave;
    static_assert(!FULL || VEC, "full chunks are vector-aligned");
    __shared__ TeamShared<Op> sh[kTeams];
    __shared__ double red[2 * NW];
    __shared__ TS bc_ms[2];

    const int team = threadIdx.x / kTeam;
    const int tid = threadIdx.x & (kTeam - 1);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int lane_off = tid * V;
    const unsigned T = (unsigned)geo.T;
    const unsigned tag = (ws->epoch + 1u) & 0xffu;  // written by the previous launch: visible across the kernel boundary
    // batch sharded over GPUs: epoch of the cross-GPU mailbox, read before this block publishes
    unsigned xepoch = 0u;
    if constexpr (XCHG) xepoch = xd.local->epoch + 1u;

    TS adv_keep[K][V];
    i64 flat_k[K];
    int nvalid_k[K];

    auto geom_of = [&](int k, i64 &start, int &count) {
        // FULL: a block's four teams take four consecutive chunks (one 16 KB run per tensor and round).  Otherwise consecutive
        // chunks go to different BLOCKS first, so that a batch of fewer chunks than team slots -- a [65536 / N x 32] shard of
        // the batch on N GPUs: 256 chunks at N = 8 -- still loads through every CU instead of through the first #chunks / 4
        const i64 c = FULL ? ((i64)k * gridDim.x + blockIdx.x) * kTeams + team : ((i64)k * kTeams + team) * gridDim.x + blockIdx.x;
        if constexpr (FULL) {
            start = c * (i64)(kTeam * V);
            count = kTeam * V;
        } else {
            const i64 row0 = c * geo.rows_per_chunk;
            i64 nrows = c < geo.nchunks ? geo.O - row0 : 0;
            if (nrows > geo.rows_per_chunk) nrows = geo.rows_per_chunk;
            if (nrows < 0) nrows = 0;
            start = row0 * geo.T;
            count = (int)(nrows * geo.T);
        }
    };
    auto nvalid_of = [&](int count) {
        if constexpr (FULL) return V;
        int nv = count - lane_off;
        return nv < 0 ? 0 : (nv > V ? V : nv);
    };

    double s = 0.0, ss = 0.0;
    typename Op::Raw raw[2];
    {
        i64 start;
        int count;
        geom_of(0, start, count);
        load_inputs<Op, VEC>(args, start + lane_off, nvalid_of(count), raw[0]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        i64 start;
        int count;
        if (k + 1 < K) {
            geom_of(k + 1, start, count);
            load_inputs<Op, VEC>(args, start + lane_off, nvalid_of(count), raw[(k + 1) & 1]);
        }
        geom_of(k, start, count);
        const int nvalid = nvalid_of(count);
        const i64 flat = start + lane_off;
        flat_k[k] = flat;
        nvalid_k[k] = nvalid;
        Op op;
        op.prepare(args, raw[k & 1]);
        TC Xc = (TC)0;
        TS Xv = (TS)0;
        scan_chunk<Op, WAVE_INDEP>(op, sh[team], k & 1, nvalid, (unsigned)lane_off % T, T, geo.scan_lanes, Xc, Xv,
                                   [&](const TC(&xs)[V], const bool(&)[V], TC, TS) {
                                       TS vt[V];
                                       op.finish(nvalid, xs, adv_keep[k], vt, s, ss);
                                       store_wt<VEC>((TS *)args.value_target, flat, nvalid, vt);
                                   });
    }

    // ---- block moments -> publish -> gather everybody's --------------------------------------------
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane == 0) {
        red[2 * wv] = s;
        red[2 * wv + 1] = ss;
    }
    __syncthreads();
    if (wv == 0) {
        double bs = 0.0, bss = 0.0;  // every lane of wave 0 forms the block totals in the same fixed order
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            bs += red[2 * k];
            bss += red[2 * k + 1];
        }
        double acc, tot_ss;
        const bool all_ok = gather::publish_and_gather(ws->slots, kSlotReplicas, geo.slot_stride, (int)gridDim.x, (int)blockIdx.x, tag, lane,
                                                       bs, bss, acc, tot_ss);
        double tot_n = geo.n_total;
        bool x_ok = true;
        if constexpr (XCHG) {
            // second stage, across GPUs: block 0 stores this GPU's totals into every peer's mailbox over
            // xGMI; every block then polls the LOCAL mailbox for all ranks' totals (xchg.h).  A rank whose own gather
            // gave up sends NaN, so that every rank sees the failure.
            const double nan = __longlong_as_double(0x7ff8000000000000LL);
            if (blockIdx.x == 0) xchg::send(xd, xepoch, lane, tot_n, all_ok ? acc : nan, all_ok ? tot_ss : nan);
            x_ok = xchg::recv_sum(xd, xepoch, lane, tot_n, acc, tot_ss);
            x_ok = x_ok && acc == acc && tot_ss == tot_ss;
        }
        if (lane == 0) {
            const bool good = all_ok && x_ok;
            TS mean_s = (TS)0, std_s = (TS)1;  // a launch that gave up leaves the advantage un-normalised
            if (good) mean_std<TS>(tot_n, acc, tot_ss, geo.eps, mean_s, std_s);
            bc_ms[0] = mean_s;
            bc_ms[1] = std_s;
            if (!good) {
                if (!all_ok) ws->error = 1u;
                if constexpr (XCHG) { if (!x_ok) xd.local->error = 1u; }
                if (geo.host_flag) __hip_atomic_store(geo.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (blockIdx.x == 0) {
                ws->epoch = ws->epoch + 1u;  // every block has read the old epoch: all of them have published
                if constexpr (XCHG) xd.local->epoch = xepoch;  // likewise: every block read it before publishing
                if (stats_out) {
                    const double nan = __longlong_as_double(0x7ff8000000000000LL);
                    stats_out[0] = tot_n;
                    stats_out[1] = good ? acc : nan;
                    stats_out[2] = good ? tot_ss : nan;
                }
            }
        }
    }
    else if (wv == 1 && geo.slot_stride > (int)gridDim.x) {
        // a grid smaller than the workspace's layout (small batches, below): keep the absent blocks' slots one launch fresh
        gather::refresh_absent_slots(ws->slots, kSlotReplicas, geo.slot_stride, (int)gridDim.x, (int)blockIdx.x, tag, lane);
    }
    __syncthreads();
    const TS mean_s = bc_ms[0], std_s = bc_ms[1];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        TS out[V];
#pragma unroll
        for (int e = 0; e < V; ++e) out[e] = (adv_keep[k][e] - mean_s) / std_s;
        store_wt<VEC>((TS *)args.advantage, flat_k[k], nvalid_k[k], out);
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

// The persistent grid needs one resident block per CU: ask the runtime once per kernel whether a 1024-thread block of
// this instantiation fits a CU at all (registers / LDS).  Whether the CUs are FREE at launch time cannot be asked: a GPU
// that runs other long kernels beside this one starves the gather (bounded wait -> error flag, un-normalised output).
template <class KernelT>
bool fused_block_fits(KernelT kernel) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kFusedThreads, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    return nb >= 1;
}

template <class Op, bool VEC, bool WI, int K, bool FULL, bool XCHG>
int launch_fused_inst(hipStream_t s, const dim3 grid, const GaeArgs &a, const FusedGeom &geo, FusedWs *w, double *stats_out,
                      const xchg::Dev &xd) {
    auto kernel = gae_fused_kernel<Op, VEC, WI, K, FULL, XCHG>;
    static const bool fits = fused_block_fits(kernel);
    if (!fits) return 2;
    hipLaunchKernelGGL(kernel, grid, dim3(kFusedThreads), 0, s, a, geo, w, stats_out, xd);
    return 0;
}

template <class Op>
LaunchResult launch_fused_t(hipStream_t s, const GaeArgs &a, double eps, void *ws, double *stats_out, const xchg::Dev &xd,
                            unsigned *host_flag) {
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
    // Batches of fewer chunks than team slots (the [65536 / N x 32] shards of a batch on N >= 4 GPUs, minibatches): one chunk
    // per team on a SMALLER grid -- fewer blocks to dispatch, fewer granules to gather, and the FULL specialisation applies
    // again.  Measured (graph replay, us per step, grid 256 / 128 / 64 / 32): [16384x32] 6.14 / 5.45 / 6.81 / 9.71, [8192x32]
    // 6.14 / 5.92 / 5.31 / 6.89, [4096x32] 6.22 / 5.93 / 5.79 / 5.40; with two or more chunks per team the full grid wins
    // ([32768x32]: 5.80 / 6.98).  The slot layout keeps the full grid's stride and the absent blocks' slots are refreshed.
    const int slot_stride = G;
    constexpr int kMinGrid = 32;
    const i64 blocks_wanted = (nchunks + kTeams - 1) / kTeams;
    if (blocks_wanted < G) G = (int)(blocks_wanted > kMinGrid ? blocks_wanted : (G < kMinGrid ? G : kMinGrid));
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
    const int KK = k_needed <= 1 ? 1 : k_needed <= 2 ? 2 : k_needed <= 4 ? 4 : 8;
    // every team's KK chunks complete: whole chunks (R*T == 256*V), all rows used, exactly KK chunks per team
    const bool full = vec && R * a.T == chunk && a.O % R == 0 && nchunks == (i64)KK * per_iter;
    FusedGeom geo{a.O, a.T, (int)R, scan_lanes_for(a.T, V, wi), nchunks, (double)a.O * (double)a.T, eps, host_flag, slot_stride};
    const dim3 grid((unsigned)G);
    FusedWs *w = (FusedWs *)ws;
    const bool x = xd.local != nullptr;
    int rc = 2;
#define TRL_FUSED(VEC, WI, KK_, FULL_, X_) rc = launch_fused_inst<Op, VEC, WI, KK_, FULL_, X_>(s, grid, a, geo, w, stats_out, xd)
#define TRL_FUSED_X(VEC, WI, KK_, FULL_) do { if (x) TRL_FUSED(VEC, WI, KK_, FULL_, true); else TRL_FUSED(VEC, WI, KK_, FULL_, false); } while (0)
#define TRL_FUSED_K(VEC, WI, FULL_)                                        \
    do {                                                                   \
        if (KK == 1) TRL_FUSED_X(VEC, WI, 1, FULL_);                       \
        else if (KK == 2) TRL_FUSED_X(VEC, WI, 2, FULL_);                  \
        else if (KK == 4) TRL_FUSED_X(VEC, WI, 4, FULL_);                  \
        else if constexpr (!kMixed) TRL_FUSED_X(VEC, WI, 8, FULL_);        \
    } while (0)
    if (full) { if (wi) TRL_FUSED_K(true, true, true); else TRL_FUSED_K(true, false, true); }
    else if (vec) { if (wi) TRL_FUSED_K(true, true, false); else TRL_FUSED_K(true, false, false); }
    else { if (wi) TRL_FUSED_K(false, true, false); else TRL_FUSED_K(false, false, false); }
#undef TRL_FUSED_K
#undef TRL_FUSED_X
#undef TRL_FUSED
    if (rc != 0) return LaunchResult{rc, 0, nullptr};
    return LaunchResult{0, 0, "gae_fused"};
}

}  // namespace

LaunchResult launch_gae_fused(hipStream_t s, int vdt, int cdt, const GaeArgs &a, double eps, void *ws, double *stats_out, void *comm,
                              unsigned *host_flag) {
    const xchg::Dev xd = xchg_dev_of(comm);
    if (!a.next_value) {  // shifted value layout: its own instantiations (scan_core.h: Shifted<>)
        if (vdt == 0 && cdt == 0) return launch_fused_t<GaeShiftedOp<float, float, 4>>(s, a, eps, ws, stats_out, xd, host_flag);
        if (vdt == 1 && cdt == 1) return launch_fused_t<GaeShiftedOp<double, double, 2>>(s, a, eps, ws, stats_out, xd, host_flag);
        if (vdt == 1 && cdt == 0) return launch_fused_t<GaeShiftedOp<double, float, 2>>(s, a, eps, ws, stats_out, xd, host_flag);
        if (vdt == 0 && cdt == 1) return launch_fused_t<GaeShiftedOp<float, double, 4>>(s, a, eps, ws, stats_out, xd, host_flag);
        return LaunchResult{1, 0, nullptr};
    }
    if (vdt == 0 && cdt == 0) return launch_fused_t<GaeOp<float, float, 4>>(s, a, eps, ws, stats_out, xd, host_flag);
    if (vdt == 1 && cdt == 1) return launch_fused_t<GaeOp<double, double, 2>>(s, a, eps, ws, stats_out, xd, host_flag);
    if (vdt == 1 && cdt == 0) return launch_fused_t<GaeOp<double, float, 2>>(s, a, eps, ws, stats_out, xd, host_flag);
    if (vdt == 0 && cdt == 1) return launch_fused_t<GaeOp<float, double, 4>>(s, a, eps, ws, stats_out, xd, host_flag);
    return LaunchResult{1, 0, nullptr};
}
