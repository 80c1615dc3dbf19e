// (shared by lockstep_gae.hip / lockstep_td_lambda.hip / lockstep_vtrace.hip: one translation unit per estimator, they
// compile in parallel)
// Lock-step scan kernels for a FEW feature elements per time step (2 <= I <= 31), any T.
//
// Element (o, t, i) sits at (o*T + t)*I + i: the I scans of an outer row are interleaved in memory (multi-agent
// rollouts [E, T, A, 1], torchrl's MAPPO / IPPO configs).  Here a LANE owns one time step and its I feature elements
// (I is a template parameter: registers, no indexing), so
//   * a wave reads / writes 64*I consecutive elements per tensor -- HBM traffic is the algorithmic minimum, no LDS
//     transpose (the LDS-transposing kernel of scan_kernels.hip needs T*I <= one chunk and is occupancy-bound);
//   * the I recurrences advance in LOCK STEP: the suffix scan over the lanes (DPP row shifts, two cross-row shuffles,
//     wave aggregates through LDS) composes I independent affine maps per move, and the step that follows (t, i)
//     is always element i of the next lane;
//   * masks that are shared by the I elements of a step (mask_bcast) are one byte per lane;
//   * 9 <= I <= 31: the feature elements are split into ceil(I / 8) groups of NI = ceil(I / groups) (blockIdx.y; the last
//     group may hold fewer), each group an independent lock-step scan that reads its NI-element slice of every step --
//     the groups of a step share cache lines, and with a grid that is a multiple of 8 they run on the same XCD.
// Short outer rows (T <= 256): a chunk is R = 256 / T whole rows, blocks walk chunks grid-stride.  Long rows: a block
// owns outer rows and walks each row's 256-step chunks from the end, the I carries (and V-trace's I next
// state_values) in registers.  Same arithmetic contract as the SCAN family: every lane replays its step with the
// reference's exact expression from its carry-in; row ends are hard resets, `done` a multiplication by zero.
#include <stdlib.h>

#pragma once
#include "scan_core.h"

namespace {

using namespace scan;


struct LsGeom {
    i64 O, T;
    int I;               // feature elements per step in memory (>= Op::V when they are split into groups)
    int rows_per_chunk;  // short rows: whole outer rows per chunk; 0: long rows
    i64 nchunks;         // short rows: chunks in total; long rows: chunks per row
};

// G: groups of a step that sit in NEIGHBOURING LANES (1: a lane owns a whole step, or the groups are on blockIdx.y)
template <class Op, int NW, int G>
struct LsShared {
    typename Op::TC a[2][NW][G][Op::V], b[2][NW][G][Op::V];
    int h[2][NW][G];
    typename Op::TS v0[2][NW][G][Op::V];
};

// One chunk: lane = one time step (valid or not), op prepared with the step's I elements.
// Xc / Xv (in/out, block-uniform): x / state_value of the step that follows the chunk, per feature element; on exit those
// of the chunk's first step.
// G > 1 (a power of two): lane l holds group l % G of step l / G -- the step that follows a lane's is G lanes further, so
// the scan moves by G, 2G, ... lanes and the G groups of a wave keep separate aggregates and carries.
template <class Op, int NW, int G, class Emit>
__device__ __forceinline__ void ls_chunk(const Op &op, LsShared<Op, NW, G> &sh, int buf, bool valid, bool row_end,
                                         typename Op::TC (&Xc)[Op::V], typename Op::TS (&Xv)[Op::V], Emit &&emit) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int NI = Op::V;
    static_assert(G == 1 || G == 2 || G == 4 || G == 8, "groups per step in neighbouring lanes: a power of two below 16");
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const int lir = lane & 15, row = lane >> 4, gl = lane & (G - 1);

    // The hard-reset flag of a map says "a row end lies in the range": it is a property of the LANES covered, the same
    // for the I maps of a lane -- one register and one move per scan step instead of I.
    TC Sa[NI], Sb[NI];
    int Sh = (valid && row_end) ? 1 : 0;
#pragma unroll
    for (int e = 0; e < NI; ++e) {
        if (!valid) { Sa[e] = (TC)1; Sb[e] = (TC)0; }
        else if (row_end) { Sa[e] = (TC)0; Sb[e] = op.step(e, op.init(e)); }
        else op.affine(e, Sa[e], Sb[e]);
    }
    // S <- S o P (P = the map of the lanes that follow): a = S.a P.a, b = S.h ? S.b : S.b + S.a P.b, h = S.h | P.h
#define TRL_LS_COMPOSE(COND, PA, PB, PH)                                  \
    {                                                                     \
        const int ph_ = (PH);                                             \
        _Pragma("unroll") for (int e = 0; e < NI; ++e) {                  \
            const TC pa_ = (PA), pb_ = (PB);                              \
            if (COND) {                                                   \
                const TC t_ = Sb[e] + Sa[e] * pb_;                        \
                Sa[e] = Sa[e] * pa_;                                      \
                Sb[e] = Sh ? Sb[e] : t_;                                  \
            }                                                             \
        }                                                                 \
        if (COND) Sh |= ph_;                                              \
    }
    // inclusive suffix scan over the wave, I maps per move
    if constexpr (G <= 1) TRL_LS_COMPOSE(lir + 1 < 16, dpp_mov<0x101>(Sa[e]), dpp_mov<0x101>(Sb[e]), dpp_mov<0x101>(Sh))
    if constexpr (G <= 2) TRL_LS_COMPOSE(lir + 2 < 16, dpp_mov<0x102>(Sa[e]), dpp_mov<0x102>(Sb[e]), dpp_mov<0x102>(Sh))
    if constexpr (G <= 4) TRL_LS_COMPOSE(lir + 4 < 16, dpp_mov<0x104>(Sa[e]), dpp_mov<0x104>(Sb[e]), dpp_mov<0x104>(Sh))
    TRL_LS_COMPOSE(lir + 8 < 16, dpp_mov<0x108>(Sa[e]), dpp_mov<0x108>(Sb[e]), dpp_mov<0x108>(Sh))
    // the 16-lane rows that follow: their totals sit in their first G lanes (one per group)
    TRL_LS_COMPOSE((row & 1) == 0, __shfl(Sa[e], ((row + 1) << 4) + gl, kWave), __shfl(Sb[e], ((row + 1) << 4) + gl, kWave),
                   __shfl(Sh, ((row + 1) << 4) + gl, kWave))
    TRL_LS_COMPOSE(row < 2, __shfl(Sa[e], 32 + gl, kWave), __shfl(Sb[e], 32 + gl, kWave), __shfl(Sh, 32 + gl, kWave))
#undef TRL_LS_COMPOSE
    if (lane < G) {
        sh.h[buf][w][gl] = Sh;
#pragma unroll
        for (int e = 0; e < NI; ++e) {
            sh.a[buf][w][gl][e] = Sa[e];
            sh.b[buf][w][gl][e] = Sb[e];
            if (Op::kNeedsNext) sh.v0[buf][w][gl][e] = op.value(e);
        }
    }
    __syncthreads();
    TC xs[NI], x_in[NI];
    TS vn[NI];
    // exclusive map (everything after this lane inside the wave) = the inclusive one of the lane that holds the next step
    auto next_step = [&](auto x) {
        if constexpr (G == 1) return dpp_mov<0x130>(x);
        else return __shfl_down(x, G, kWave);
    };
    const bool last_in_wave = lane >= kWave - G;
    int Eh = next_step(Sh);
    if (last_in_wave) Eh = 0;
#pragma unroll
    for (int e = 0; e < NI; ++e) {
        // carry into this wave from the waves that follow it and from the chunk's carry-in
        TC x = Xc[e], Cw = Xc[e];
        TS xv = Xv[e], Cv = Xv[e];
#pragma unroll
        for (int k = NW - 1; k >= 0; --k) {
            if (k == w) { Cw = x; Cv = xv; }
            x = apply(Aff<TC>{sh.a[buf][k][gl][e], sh.b[buf][k][gl][e], sh.h[buf][k][gl]}, x);
            if (Op::kNeedsNext) xv = sh.v0[buf][k][gl][e];
        }
        Xc[e] = x;
        Xv[e] = xv;
        Aff<TC> E{next_step(Sa[e]), next_step(Sb[e]), Eh};
        if (last_in_wave) E = Aff<TC>{(TC)1, (TC)0, 0};
        vn[e] = Op::kNeedsNext ? next_step(op.value(e)) : (TS)0;
        if (Op::kNeedsNext && last_in_wave) vn[e] = Cv;
        x_in[e] = apply(E, Cw);
        xs[e] = op.step(e, row_end ? op.init(e) : x_in[e]);  // the reference's exact step from the carry-in
    }
    emit(xs, x_in, vn);
}

// i0 / ne: first feature element of the block's group and how many the group holds (ne <= Op::V; the rest read as 0,
// which every estimator maps to 0 outputs, and are not stored)
// VEC: every group is full and its slice of a step is naturally aligned -> one vector load / store per tensor
template <class Op, bool VEC>
__device__ __forceinline__ void ls_load(const typename Op::Args &args, i64 step_index, int I, int i0, int ne, bool valid,
                                        typename Op::Raw &raw) {
    constexpr int NI = Op::V;
    Op::for_each_input(args, raw, [&](auto *base, auto &arr) {
        typedef typename std::remove_reference<decltype(arr[0])>::type E;
        if (!base) return;
        if (sizeof(E) == 1 && args.mask_bcast) {  // one mask byte per step, shared by its I elements
            const E m = valid ? base[step_index] : (E)0;
#pragma unroll
            for (int e = 0; e < NI; ++e) arr[e] = m;
        } else {
            load_n<VEC>(base, step_index * I + i0, valid ? ne : 0, arr);
        }
    });
}

// Occupancy target (waves per SIMD) for the fp32 GAE / TD(lambda) kernels: without it the register allocator lands a few
// registers above a threshold for some I (I = 3: 98 VGPRs = 4 waves where 96 give 5; [8192x100x3] 16.6 vs 14.9 us).
// V-trace and the fp64 kernels need more registers than any useful target leaves them (they spill): no target there.
template <class Op>
constexpr int ls_waves() {
    if (Op::kNeedsNext) return 1;
    if (sizeof(typename Op::TC) != 4) return Op::V <= 2 ? 4 : Op::V <= 5 ? 3 : 2;
    return Op::V <= 3 ? 5 : Op::V <= 5 ? 4 : 3;
}

// G > 1: the I = G * NI feature elements of a step are held by G neighbouring lanes (a block covers NT / G steps), so a
// wave reads 64 * NI CONSECUTIVE elements per tensor -- whole cache lines -- instead of NI of every I (blockIdx.y groups:
// every group's blocks fetch the same lines at different times; [4096x300x16] ran at 2 TB/s).
template <class Op, bool VEC, int NT, int G>
__global__ __launch_bounds__(NT, ls_waves<Op>()) void lockstep_kernel(typename Op::Args args, LsGeom geo, double *partials) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    typedef typename Op::TO TO;
    constexpr int NI = Op::V;
    constexpr int NW = NT / kWave;
    constexpr int NS = NT / G;  // steps per chunk
    __shared__ LsShared<Op, NW, G> sh;
    __shared__ double red[2 * NW];
    double s = 0.0, ss = 0.0;
    int buf = 0;
    const int I = geo.I, i0 = (G > 1 ? (int)(threadIdx.x & (G - 1)) : (int)blockIdx.y) * NI;
    const int ne = I - i0 < NI ? I - i0 : NI;
    const unsigned slot = threadIdx.x / G;  // the step of the chunk this lane works on
    auto run = [&](i64 step_index, bool valid, bool row_end, TC (&Xc)[NI], TS (&Xv)[NI]) {
        typename Op::Raw raw;
        ls_load<Op, VEC>(args, step_index, I, i0, ne, valid, raw);
        Op op;
        op.prepare(args, raw);
        ls_chunk<Op, NW, G>(op, sh, buf, valid, row_end, Xc, Xv, [&](const TC(&xs)[NI], const TC(&x_in)[NI], const TS(&vn)[NI]) {
            TO o0[NI], o1[NI];
            op.outputs_ls(valid, xs, row_end, x_in, vn, o0, o1, s, ss);
            store_n<VEC>(Op::out_ptr(args, 0), step_index * I + i0, valid ? ne : 0, o0);
            if (Op::kNumOut > 1) store_n<VEC>(Op::out_ptr(args, 1), step_index * I + i0, valid ? ne : 0, o1);
        });
        buf ^= 1;
    };
    if (geo.rows_per_chunk > 0) {
        // short rows: R whole outer rows per chunk, every chunk ends at a row end (no carry between chunks)
        const unsigned T = (unsigned)geo.T;
        const unsigned lr = slot / T, t = slot % T;
        for (i64 c = blockIdx.x; c < geo.nchunks; c += gridDim.x) {
            const i64 row = c * geo.rows_per_chunk + lr;
            const bool valid = lr < (unsigned)geo.rows_per_chunk && row < geo.O;
            TC Xc[NI];
            TS Xv[NI];
#pragma unroll
            for (int e = 0; e < NI; ++e) { Xc[e] = (TC)0; Xv[e] = (TS)0; }
            run(row * geo.T + t, valid, valid && t == T - 1, Xc, Xv);
        }
    } else {
        for (i64 row = blockIdx.x; row < geo.O; row += gridDim.x) {
            TC Xc[NI];
            TS Xv[NI];
#pragma unroll
            for (int e = 0; e < NI; ++e) { Xc[e] = (TC)0; Xv[e] = (TS)0; }
            for (i64 c = geo.nchunks - 1; c >= 0; --c) {
                const i64 t = c * NS + slot;
                const bool valid = t < geo.T;
                run(row * geo.T + t, valid, valid && t == geo.T - 1, Xc, Xv);
            }
        }
    }
    if (partials) {
        block_sum2<NT>(s, ss, red);
        if (threadIdx.x == 0) {
            const i64 b = (i64)blockIdx.y * gridDim.x + blockIdx.x;
            partials[2 * b] = s;
            partials[2 * b + 1] = ss;
        }
    }
}

int ls_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
}

template <template <typename, typename, int> class OpT, typename TS, typename TG, int NI, int NT, class Args>
LaunchResult launch_ls_nt(hipStream_t s, const Args &a, bool aligned, double *partials, const char *name);

// the groups of a step in neighbouring lanes (I == G * NI): blockIdx.y is not used
template <template <typename, typename, int> class OpT, typename TS, typename TG, int NI, int G, class Args>
LaunchResult launch_ls_lanes(hipStream_t s, const Args &a, bool aligned, double *partials, const char *name) {
    typedef OpT<TS, TG, NI> Op;
    constexpr int NT = 256, NS = NT / G;
    static const int cus = ls_cus();
    LsGeom geo{a.O, a.T, (int)a.I, 0, 0};
    i64 grid;
    if (a.T <= NS) {
        i64 R = NS / a.T;
        if (R > a.O) R = a.O;
        geo.rows_per_chunk = (int)R;
        geo.nchunks = (a.O + R - 1) / R;
        const i64 cap = (i64)cus * 8;
        grid = geo.nchunks < cap ? geo.nchunks : cap;
    } else {
        geo.nchunks = (a.T + NS - 1) / NS;
        grid = a.O < 2048 ? a.O : 2048;
    }
    const dim3 g((unsigned)grid, 1u);
    if constexpr (NI == 2 || NI == 4 || NI == 8) {
        if (aligned) {
            hipLaunchKernelGGL((lockstep_kernel<Op, true, NT, G>), g, dim3(NT), 0, s, a, geo, partials);
            return LaunchResult{0, grid, name};
        }
    }
    hipLaunchKernelGGL((lockstep_kernel<Op, false, NT, G>), g, dim3(NT), 0, s, a, geo, partials);
    return LaunchResult{0, grid, name};
}

template <template <typename, typename, int> class OpT, typename TS, typename TG, int NI, int NT, class Args>
LaunchResult launch_ls_nt(hipStream_t s, const Args &a, bool aligned, double *partials, const char *name) {
    typedef OpT<TS, TG, NI> Op;
    static const int cus = ls_cus();
    LsGeom geo{a.O, a.T, (int)a.I, 0, 0};
    const i64 groups = (a.I + NI - 1) / NI;
    i64 grid;
    if (a.T <= NT) {
        i64 R = NT / a.T;
        if (R > a.O) R = a.O;
        geo.rows_per_chunk = (int)R;
        geo.nchunks = (a.O + R - 1) / R;
        const i64 cap = (i64)cus * 8 * 256 / NT;
        grid = geo.nchunks < cap ? geo.nchunks : cap;
    } else {
        geo.nchunks = (a.T + NT - 1) / NT;
        grid = a.O < 2048 ? a.O : 2048;
    }
    if (groups > 1) {
        // at most 4096 partials in all; a multiple of 8 blocks per group keeps the groups of a step on one XCD
        if (grid > 4096 / groups) grid = 4096 / groups;
        if (grid >= 8) grid -= grid % 8;
    }
    const dim3 g((unsigned)grid, (unsigned)groups);
    if constexpr (NI == 2 || NI == 4 || NI == 8) {
        if (aligned && a.I % NI == 0) {
            hipLaunchKernelGGL((lockstep_kernel<Op, true, NT, 1>), g, dim3(NT), 0, s, a, geo, partials);
            return LaunchResult{0, grid * groups, name};
        }
    }
    hipLaunchKernelGGL((lockstep_kernel<Op, false, NT, 1>), g, dim3(NT), 0, s, a, geo, partials);
    return LaunchResult{0, grid * groups, name};
}

// Block size: 256 lanes.  512-lane blocks pack short rows better (T = 100: 98 % of the lanes hold a step instead of 78 %)
// but measured slower on every shape tried ([4096x100x5] 30.4 vs 20.1 us, [8192x100x8] 55.2 vs 40.2, [8192x32x8] 16.7 vs
// 12.5): half as many resident blocks to overlap one block's barrier phases with another's loads.
template <template <typename, typename, int> class OpT, typename TS, typename TG, int NI, class Args>
LaunchResult launch_ls(hipStream_t s, const Args &a, bool aligned, double *partials, const char *name) {
    // 128-lane blocks when a long row's last chunk would leave a 256-lane block mostly idle and there are rows enough
    // to keep the chip busy (T = 300: 3 chunks of 128 at 78 % instead of 2 of 256 at 59 %: [2048x300x5] 30.7 -> 20.3 us;
    // short rows and few long rows are slower with 128: [8192x100x4] 24.0 vs 19.6, [64x1000x5] 22.9 vs 15.0)
    auto fill = [&](i64 nt) { return (double)a.T / (double)(((a.T + nt - 1) / nt) * nt); };
    if (a.T > 256 && a.O >= 1024 && fill(128) >= fill(256) + 0.1) return launch_ls_nt<OpT, TS, TG, NI, 128>(s, a, aligned, partials, name);
    return launch_ls_nt<OpT, TS, TG, NI, 256>(s, a, aligned, partials, name);
}

template <template <typename, typename, int> class OpT, class Args>
LaunchResult dispatch_ls(hipStream_t s, int vdt, int cdt, const Args &a, bool aligned, double *partials, const char *name) {
    if (a.I < 2 || a.I > 31 || a.T >= (1LL << 31) - 4096) return LaunchResult{2, 0, nullptr};
    // 9 <= I <= 31 = G * NI with G a power of two and NI <= 8: the G groups of a step in neighbouring lanes (whole cache lines
    // per wave).  Measured fp32 GAE (round 3), groups on blockIdx.y -> in lanes: [4096x300x16] 218 -> 82 us (5.3 TB/s),
    // [8192x100x16] 146 -> 55 (SEQ 129), [1024x300x16] 62 -> 22, [256x300x24] 49 -> 22; few small groups beat fewer large
    // ones (registers): I = 12 as 4 x 3 107 us vs 2 x 6 124 at [4096x300x12], I = 24 as 8 x 3 183 vs 4 x 6 225 at
    // [4096x300x24].  A handful of long rows keeps the old mapping (more blocks: [64x1000x16] 20 vs 29 us).
    if (a.I > 8 && (a.T <= 32 || a.O >= 128)) {
#define TRL_LS_LANES(G_, N_)                                                                                              \
    do {                                                                                                                  \
        if (vdt == 0 && cdt == 0) return launch_ls_lanes<OpT, float, float, N_, G_>(s, a, aligned, partials, name);        \
        if (vdt == 0) return launch_ls_lanes<OpT, float, double, N_, G_>(s, a, aligned, partials, name);                   \
        if (cdt == 1) return launch_ls_lanes<OpT, double, double, N_, G_>(s, a, aligned, partials, name);                  \
        return launch_ls_lanes<OpT, double, float, N_, G_>(s, a, aligned, partials, name);                                 \
    } while (0)
        switch ((int)a.I) {
            case 10: TRL_LS_LANES(2, 5);
            case 12: TRL_LS_LANES(4, 3);
            case 14: TRL_LS_LANES(2, 7);
            case 16: TRL_LS_LANES(4, 4);
            case 20: TRL_LS_LANES(4, 5);
            case 24: TRL_LS_LANES(8, 3);
            case 28: TRL_LS_LANES(4, 7);
            default: break;  // no such factorisation: groups on blockIdx.y, below
        }
#undef TRL_LS_LANES
    }
    // groups of 8 or 4 when they divide I (full groups, vector loads), else balanced: 9 -> 5 + 4, 31 -> 8 + 8 + 8 + 7
    const int groups = ((int)a.I + 7) / 8;
    const int ni = a.I <= 8 ? (int)a.I : a.I % 8 == 0 ? 8 : a.I % 4 == 0 ? 4 : ((int)a.I + groups - 1) / groups;
#define TRL_LS_CASE(N)                                                                                  \
    case N:                                                                                             \
        if (vdt == 0 && cdt == 0) return launch_ls<OpT, float, float, N>(s, a, aligned, partials, name);    \
        if (vdt == 0) return launch_ls<OpT, float, double, N>(s, a, aligned, partials, name);                     \
        if (cdt == 1) return launch_ls<OpT, double, double, N>(s, a, aligned, partials, name);                   \
        return launch_ls<OpT, double, float, N>(s, a, aligned, partials, name);
    switch (ni) {
        TRL_LS_CASE(2)
        TRL_LS_CASE(3)
        TRL_LS_CASE(4)
        TRL_LS_CASE(5)
        TRL_LS_CASE(6)
        TRL_LS_CASE(7)
        TRL_LS_CASE(8)
    }
#undef TRL_LS_CASE
    return LaunchResult{2, 0, nullptr};
}

inline bool al16(const void *p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; }

}  // namespace
