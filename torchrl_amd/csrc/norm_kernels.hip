// Advantage statistics and normalisation.
// Replaces `(adv - adv.mean()) / adv.std().clamp_min(eps)` (reference:
// torchrl/objectives/value/advantages.py:1459-1463, 1828-1832; torchrl/_utils.py:901-912).
//
// Statistics travel as {N, sum, sumsq} in fp64 so that a batch sharded over GPUs needs one
// 24-byte all-reduce.  All reductions run in a fixed order (per-lane strided accumulation ->
// wave shuffle tree -> per-block partial -> single-block final tree): bitwise reproducible.
#include "common.h"
#include "moment_gather.h"
#include <cstdlib>

namespace {

constexpr int kRedThreads = 256;

// Final reduction of per-block {sum, sumsq} partials into stats = {N, sum, sumsq}.
__global__ __launch_bounds__(kRedThreads) void reduce_partials_kernel(const double *__restrict__ partials,
                                                                      i64 npartials, double n,
                                                                      double *__restrict__ stats) {
    __shared__ double red[2 * kRedThreads / kWave];
    double s = 0.0, ss = 0.0;
    for (i64 k = threadIdx.x; k < npartials; k += kRedThreads) {
        s += partials[2 * k];
        ss += partials[2 * k + 1];
    }
    block_sum2<kRedThreads>(s, ss, red);
    if (threadIdx.x == 0) {
        stats[0] = n;
        stats[1] = s;
        stats[2] = ss;
    }
}

template <typename TS>
__global__ __launch_bounds__(kRedThreads) void moments_kernel(const TS *__restrict__ x, i64 n,
                                                              double *__restrict__ partials) {
    __shared__ double red[2 * kRedThreads / kWave];
    double s = 0.0, ss = 0.0;
    const i64 stride = (i64)gridDim.x * kRedThreads;
    for (i64 k = (i64)blockIdx.x * kRedThreads + threadIdx.x; k < n; k += stride) {
        const double xv = (double)x[k];
        s += xv;
        ss += xv * xv;
    }
    block_sum2<kRedThreads>(s, ss, red);
    if (threadIdx.x == 0) {
        partials[2 * (i64)blockIdx.x] = s;
        partials[2 * (i64)blockIdx.x + 1] = ss;
    }
}

template <typename TS>
struct Vec4 { TS x, y, z, w; };

template <typename TS>
__global__ __launch_bounds__(kRedThreads) void normalize_kernel(const TS *__restrict__ x, i64 n,
                                                                const double *__restrict__ stats, double eps,
                                                                TS *__restrict__ out, int vec_ok) {
    TS mean_s, std_s;
    mean_std<TS>(stats[0], stats[1], stats[2], eps, mean_s, std_s);
    const i64 stride = (i64)gridDim.x * kRedThreads;
    const i64 tid = (i64)blockIdx.x * kRedThreads + threadIdx.x;
    if (vec_ok) {
        constexpr int V = 16 / sizeof(TS);
        typedef TS vec_t __attribute__((ext_vector_type(V)));
        const i64 nv = n / V;
        const vec_t *__restrict__ xv = (const vec_t *)x;
        vec_t *__restrict__ ov = (vec_t *)out;
        for (i64 k = tid; k < nv; k += stride) {
            vec_t a = xv[k];
#pragma unroll
            for (int e = 0; e < V; ++e) a[e] = (a[e] - mean_s) / std_s;
            __builtin_nontemporal_store(a, ov + k);
        }
        for (i64 k = nv * V + tid; k < n; k += stride) out[k] = (x[k] - mean_s) / std_s;
    } else {
        for (i64 k = tid; k < n; k += stride) out[k] = (x[k] - mean_s) / std_s;
    }
}

// Scale pass that starts from the per-block partials of the producing kernel: every block sums
// the (few hundred) partials itself in a fixed order -- no separate reduction launch -- then
// streams its share of x.  Block 0 also writes {N, sum, sumsq} to stats_out (optional).
template <typename TS>
__global__ __launch_bounds__(kRedThreads) void normalize_from_partials_kernel(const TS *__restrict__ x, i64 n,
                                                                              const double *__restrict__ partials,
                                                                              int npartials, double n_total, double eps,
                                                                              TS *__restrict__ out, double *stats_out,
                                                                              int vec_ok) {
    __shared__ double red[2 * kRedThreads / kWave];
    __shared__ double bc[2];
    double s = 0.0, ss = 0.0;
    for (int k = threadIdx.x; k < npartials; k += kRedThreads) {
        s += partials[2 * k];
        ss += partials[2 * k + 1];
    }
    block_sum2<kRedThreads>(s, ss, red);
    if (threadIdx.x == 0) {
        bc[0] = s;
        bc[1] = ss;
        if (blockIdx.x == 0 && stats_out) { stats_out[0] = n_total; stats_out[1] = s; stats_out[2] = ss; }
    }
    __syncthreads();
    TS mean_s, std_s;
    mean_std<TS>(n_total, bc[0], bc[1], eps, mean_s, std_s);
    const i64 stride = (i64)gridDim.x * kRedThreads;
    const i64 tid = (i64)blockIdx.x * kRedThreads + threadIdx.x;
    if (vec_ok) {
        constexpr int V = 16 / sizeof(TS);
        typedef TS vec_t __attribute__((ext_vector_type(V)));
        const i64 nv = n / V;
        const vec_t *__restrict__ xv = (const vec_t *)x;
        vec_t *__restrict__ ov = (vec_t *)out;
        for (i64 k = tid; k < nv; k += stride) {
            vec_t a = xv[k];
#pragma unroll
            for (int e = 0; e < V; ++e) a[e] = (a[e] - mean_s) / std_s;
            __builtin_nontemporal_store(a, ov + k);
        }
        for (i64 k = nv * V + tid; k < n; k += stride) out[k] = (x[k] - mean_s) / std_s;
    } else {
        for (i64 k = tid; k < n; k += stride) out[k] = (x[k] - mean_s) / std_s;
    }
}

// ---- grouped statistics: `_standardize(x, exclude_dims)` (reference: torchrl/_utils.py:875-936) ---------
// The excluded dims index G independent groups; x is seen as (outer, G, inner) contiguous, element
// f = (o*G + g)*inner + j belongs to group g (the host merges / permutes dims into this pattern).
// Statistics per group: {M, sum, sumsq} in fp64, fixed order, M = outer*inner.
template <typename TS>
__global__ __launch_bounds__(kRedThreads) void moments_grouped_kernel(const TS *__restrict__ x, i64 outer, i64 G, i64 inner,
                                                                      double *__restrict__ partials) {
    __shared__ double red[2 * kRedThreads / kWave];
    const i64 g = blockIdx.y;
    const i64 M = outer * inner;
    double s = 0.0, ss = 0.0;
    const i64 stride = (i64)gridDim.x * kRedThreads;
    if (inner == 1) {
        for (i64 m = (i64)blockIdx.x * kRedThreads + threadIdx.x; m < M; m += stride) {
            const double xv = (double)x[m * G + g];
            s += xv;
            ss += xv * xv;
        }
    } else {
        for (i64 m = (i64)blockIdx.x * kRedThreads + threadIdx.x; m < M; m += stride) {
            const i64 o = m / inner, j = m - o * inner;
            const double xv = (double)x[(o * G + g) * inner + j];
            s += xv;
            ss += xv * xv;
        }
    }
    block_sum2<kRedThreads>(s, ss, red);
    if (threadIdx.x == 0) {
        partials[2 * (g * gridDim.x + blockIdx.x)] = s;
        partials[2 * (g * gridDim.x + blockIdx.x) + 1] = ss;
    }
}

__global__ __launch_bounds__(kRedThreads) void reduce_partials_grouped_kernel(const double *__restrict__ partials, int nper,
                                                                              double m_count, double *__restrict__ stats) {
    __shared__ double red[2 * kRedThreads / kWave];
    const i64 g = blockIdx.x;
    double s = 0.0, ss = 0.0;
    for (int k = threadIdx.x; k < nper; k += kRedThreads) {
        s += partials[2 * (g * nper + k)];
        ss += partials[2 * (g * nper + k) + 1];
    }
    block_sum2<kRedThreads>(s, ss, red);
    if (threadIdx.x == 0) {
        stats[3 * g] = m_count;
        stats[3 * g + 1] = s;
        stats[3 * g + 2] = ss;
    }
}

constexpr int kMaxTableGroups = 2048;  // (mean, std) table in LDS: 2048 * 2 * 8 B = 32 KiB

// Flat, coalesced scale pass; every block first derives the per-group (mean, std) table into LDS.
template <typename TS>
__global__ __launch_bounds__(kRedThreads) void normalize_grouped_kernel(const TS *__restrict__ x, i64 n, i64 G, i64 inner,
                                                                        const double *__restrict__ stats, double eps,
                                                                        TS *__restrict__ out) {
    __shared__ TS tab[2 * kMaxTableGroups];
    for (i64 g = threadIdx.x; g < G; g += kRedThreads) {
        TS mean_s, std_s;
        mean_std<TS>(stats[3 * g], stats[3 * g + 1], stats[3 * g + 2], eps, mean_s, std_s);
        tab[2 * g] = mean_s;
        tab[2 * g + 1] = std_s;
    }
    __syncthreads();
    const i64 stride = (i64)gridDim.x * kRedThreads;
    if (n <= 0x7fffffffLL) {  // 32-bit index math: the 64-bit divisions dominated this kernel (7.4 us at [4096x100x5])
        const unsigned n32 = (unsigned)n, G32 = (unsigned)G, in32 = (unsigned)inner, st32 = (unsigned)stride;
        for (unsigned f = blockIdx.x * kRedThreads + threadIdx.x; f < n32; f += st32) {
            const unsigned g = in32 == 1 ? f % G32 : (f / in32) % G32;
            out[f] = (x[f] - tab[2 * g]) / tab[2 * g + 1];
        }
        return;
    }
    for (i64 f = (i64)blockIdx.x * kRedThreads + threadIdx.x; f < n; f += stride) {
        const i64 g = (f / inner) % G;
        out[f] = (x[f] - tab[2 * g]) / tab[2 * g + 1];
    }
}

// Many groups (G > kMaxTableGroups): one block column per group, as the moments kernel walks them.
template <typename TS>
__global__ __launch_bounds__(kRedThreads) void normalize_grouped_by_group_kernel(const TS *__restrict__ x, i64 outer, i64 G, i64 inner,
                                                                                 const double *__restrict__ stats, double eps,
                                                                                 TS *__restrict__ out) {
    const i64 g = blockIdx.y;
    const i64 M = outer * inner;
    TS mean_s, std_s;
    mean_std<TS>(stats[3 * g], stats[3 * g + 1], stats[3 * g + 2], eps, mean_s, std_s);
    const i64 stride = (i64)gridDim.x * kRedThreads;
    for (i64 m = (i64)blockIdx.x * kRedThreads + threadIdx.x; m < M; m += stride) {
        const i64 o = m / inner, j = m - o * inner;
        const i64 f = (o * G + g) * inner + j;
        out[f] = (x[f] - mean_s) / std_s;
    }
}

// Small tensors (a PPO minibatch of advantages): moments and scale pass in ONE single-block launch;
// the second pass re-reads x from L2.  Same fixed reduction order every run.
constexpr int kSmallThreads = 1024;
template <typename TS>
__global__ __launch_bounds__(kSmallThreads) void standardize_small_kernel(const TS *__restrict__ x, i64 n, double eps,
                                                                          TS *__restrict__ out, double *__restrict__ stats_out) {
    __shared__ double red[2 * kSmallThreads / kWave];
    __shared__ double bc[2];
    double s = 0.0, ss = 0.0;
    for (i64 k = threadIdx.x; k < n; k += kSmallThreads) {
        const double xv = (double)x[k];
        s += xv;
        ss += xv * xv;
    }
    block_sum2<kSmallThreads>(s, ss, red);
    if (threadIdx.x == 0) {
        bc[0] = s;
        bc[1] = ss;
        if (stats_out) { stats_out[0] = (double)n; stats_out[1] = s; stats_out[2] = ss; }
    }
    __syncthreads();
    TS mean_s, std_s;
    mean_std<TS>((double)n, bc[0], bc[1], eps, mean_s, std_s);
    for (i64 k = threadIdx.x; k < n; k += kSmallThreads) out[k] = (x[k] - mean_s) / std_s;
}

// ---- standardize in ONE launch: persistent co-resident grid, values kept in registers across an in-launch gather --------
// `_standardize(x)` / `_standardize(x, exclude_dims)` on a rollout-sized tensor (reference: torchrl/_utils.py:875-936; the MAPPO
// call keeps the agent dim independent, sota-implementations/multiagent/mappo_ippo.py:177-182) is two passes over x with a
// grid-wide dependency between them.  As in the FUSED GAE kernel (fused_kernels.hip) one 1024-thread block per CU loads its
// share once, keeps it in registers, publishes per-block fp64 moments and gathers everybody's (moment_gather.h), then
// writes the normalised values: 2 x sizeof(T) bytes per element, one launch.
//   * x is (outer, G, inner); the group of flat element f is (f / inner) % G, periodic with P = G * inner.  A block walks
//     tiles of S = 1024 - 1024 % P elements (threads t >= S idle): tile bases are multiples of P, so thread t ALWAYS holds
//     group (t / inner) % G -- one fp64 accumulator pair per thread, no per-element group arithmetic, coalesced 4-byte
//     (8-byte) accesses.  G = 1: S = 1024 and a thread takes V elements with one 16-byte access.
//   * per-thread pairs go through LDS; wave g (g < G <= 16) sums the threads of group g in a fixed order, publishes the
//     block's granule of group g and gathers group g over all blocks; its lane 0 leaves (mean_g, std_g) in LDS.
//   * slots: [group][replica][block]; waves g >= G rewrite their slots too (zeros, previous tag) when the workspace is the
//     grouped one, so that every slot of all 16 groups is rewritten by every launch (moment_gather.h: the rule that keeps
//     stale tags from matching).
//   * failure: as the FUSED kernel -- bounded spin, un-normalised output (mean 0, std 1), error word, host flag.
constexpr int kPnThreads = 1024;
constexpr int kPnWaves = kPnThreads / kWave;  // 16 = the most groups one launch takes
constexpr int kPnReplicas = 4;

struct PnGeom {
    i64 n;
    int G, inner, S;      // S: elements (threads) per tile row, a multiple of G * inner
    int pub_groups;       // groups whose slots are (re)written by every launch: 1 (ungrouped workspace) or 16
    int nrep;
    double m_count;       // elements per group
    double eps;
    unsigned *host_flag;
};

// write-through stores (sc1), as the FUSED GAE kernel: nothing is left dirty in the L2s, so the end-of-kernel write-back is
// empty.  The s_nop after the 16-byte form: see scan_core.h store_wt.
__device__ __forceinline__ void store_wt_elem(float *p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_wt_elem(double *p, double v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
template <typename VT>
__device__ __forceinline__ void store_wt_vec16(void *p, const VT &v) {
    static_assert(sizeof(VT) == 16, "16-byte vector");
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

template <typename TS, int K, int V>
__global__ __launch_bounds__(kPnThreads) void standardize_persistent_kernel(const TS *x, TS *out, PnGeom geo,  // out may alias x (every load precedes the gather)
                                                                            gather::Header *ws, double *__restrict__ stats_out) {
    typedef TS vec_t __attribute__((ext_vector_type(V)));
    __shared__ double part[2 * kPnThreads];
    __shared__ TS tab[2 * kPnWaves];
    __shared__ unsigned sh_tag, sh_bad;
    const int t = threadIdx.x, lane = t & (kWave - 1), wv = t / kWave;
    const i64 tile_elems = (i64)geo.S * V;
    if (t == 0) { sh_tag = (ws->epoch + 1u) & 0xffu; sh_bad = 0u; }  // read once per block, before anything is published

    TS keep[K][V];
    double s = 0.0, ss = 0.0;
    const bool active = t < geo.S;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const i64 f0 = ((i64)k * gridDim.x + blockIdx.x) * tile_elems + (i64)t * V;
        if (V > 1 && active && f0 + V <= geo.n) {
            const vec_t a = *(const vec_t *)(x + f0);
#pragma unroll
            for (int e = 0; e < V; ++e) keep[k][e] = a[e];
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) keep[k][e] = (active && f0 + e < geo.n) ? x[f0 + e] : (TS)0;
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) {  // padding elements are zeros: they add nothing
            const double xv = (double)keep[k][e];
            s += xv;
            ss += xv * xv;
        }
    part[2 * t] = s;
    part[2 * t + 1] = ss;
    __syncthreads();
    const unsigned tag = sh_tag;
    if (wv < geo.pub_groups || wv < geo.G) {
        double bs = 0.0, bss = 0.0;
        const bool mine = wv < geo.G;
        if (mine) {  // threads of group wv: (q*G + wv)*inner + r, walked in (q, r) order
            const int cnt = geo.S / geo.G;
            for (int j = lane; j < cnt; j += kWave) {
                const int q = j / geo.inner, r = j - q * geo.inner;
                const int tt = (q * geo.G + wv) * geo.inner + r;
                bs += part[2 * tt];
                bss += part[2 * tt + 1];
            }
            bs = wave_sum(bs);
            bss = wave_sum(bss);
        }
        const int nb = (int)gridDim.x;
        unsigned long long *slots = ws->slots + (size_t)wv * geo.nrep * nb * 2;
        if (mine) {
            double tot, tot2;
            const bool ok = gather::publish_and_gather(slots, geo.nrep, nb, nb, (int)blockIdx.x, tag, lane, bs, bss, tot, tot2);
            if (lane == 0) {
                TS mean_s = (TS)0, std_s = (TS)1;  // a launch that gave up leaves x un-normalised
                if (ok) mean_std<TS>(geo.m_count, tot, tot2, geo.eps, mean_s, std_s);
                tab[2 * wv] = mean_s;
                tab[2 * wv + 1] = std_s;
                if (!ok) {
                    sh_bad = 1u;
                    ws->error = 1u;
                    if (geo.host_flag) __hip_atomic_store(geo.host_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                if (blockIdx.x == 0) {
                    if (stats_out) {
                        const double nan = __longlong_as_double(0x7ff8000000000000LL);
                        stats_out[3 * wv] = geo.m_count;
                        stats_out[3 * wv + 1] = ok ? tot : nan;
                        stats_out[3 * wv + 2] = ok ? tot2 : nan;
                    }
                    // group 0 gathered: wave 0 of every block has published, i.e. every block is past the barrier that
                    // follows its read of the epoch
                    if (wv == 0) ws->epoch = ws->epoch + 1u;
                }
            }
        } else if (lane < geo.nrep) {
            // keep the unused groups' slots one launch fresh -- with the PREVIOUS tag: nobody reads them in this launch, and
            // whatever tag this block computed (a block that starts after block 0 gave up and bumped the epoch computes the
            // next launch's) the granule cannot match the next launch's
            const unsigned old = (tag - 1u) & 0xffu;
            gather::store_granule(slots + ((size_t)lane * nb + blockIdx.x) * 2, gather::tag_bits(0.0, old), gather::tag_bits(0.0, old));
        }
    }
    __syncthreads();
    // one group gave up: the whole output stays un-normalised (recognisable, finite), as documented
    const bool bad = sh_bad != 0u;
    const int g = active ? (t / geo.inner) % geo.G : 0;
    const TS mean_s = bad ? (TS)0 : tab[2 * g], std_s = bad ? (TS)1 : tab[2 * g + 1];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const i64 f0 = ((i64)k * gridDim.x + blockIdx.x) * tile_elems + (i64)t * V;
        if (V > 1 && active && f0 + V <= geo.n) {
            if constexpr (V > 1) {
                vec_t a;
#pragma unroll
                for (int e = 0; e < V; ++e) a[e] = (keep[k][e] - mean_s) / std_s;
                store_wt_vec16(out + f0, a);
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e)
                if (active && f0 + e < geo.n) store_wt_elem(out + f0 + e, (keep[k][e] - mean_s) / std_s);
        }
    }
}

int pn_device_cus() {
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

template <typename TS, int K, int V>
int launch_pn_inst(hipStream_t s, int grid, const void *x, void *out, const PnGeom &geo, void *ws, double *stats_out) {
    auto kernel = standardize_persistent_kernel<TS, K, V>;
    static const bool fits = [&] {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, kPnThreads, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
        return nb >= 1;
    }();
    if (!fits) return 2;
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kPnThreads), 0, s, (const TS *)x, (TS *)out, geo, (gather::Header *)ws, stats_out);
    return 0;
}

template <typename TS>
int launch_pn_t(hipStream_t s, const void *x, void *out, PnGeom geo, void *ws, double *stats_out) {
    int grid = pn_device_cus();
    if (grid <= 0) return 2;
    // rehearsal hook shared with the FUSED GAE kernel: ranks that share one GPU must all be co-resident
    static const int grid_cap = getenv("TRL_FUSED_GRID_CAP") ? atoi(getenv("TRL_FUSED_GRID_CAP")) : 0;
    if (grid_cap > 0 && grid > grid_cap) grid = grid_cap;
    // the grid is the same for every launch on a workspace (moment_gather.h: every slot read is rewritten): #CUs, capped by the
    // slot layout -- the FUSED GAE kernel's for the shared header (ungrouped), kPersistentNormWsBytes for the grouped one
    const int max_grid = geo.pub_groups == 1 ? gather::kMaxBlocks : 256;
    if (grid > max_grid) grid = max_grid;
    constexpr int VMAX = 16 / (int)sizeof(TS);
    const bool vec = geo.G == 1 && geo.inner == 1 && (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
    const i64 per_round = (i64)grid * geo.S * (vec ? VMAX : 1);
    const i64 k_needed = (geo.n + per_round - 1) / per_round;
    // values kept per thread: at most 56 x fp32 / 28 x fp64 (the 128 VGPRs of a 1024-thread block; 64 / 32 spill)
    if (vec) {
        if (k_needed <= 2) return launch_pn_inst<TS, 2, VMAX>(s, grid, x, out, geo, ws, stats_out);
        if (k_needed <= 8) return launch_pn_inst<TS, 8, VMAX>(s, grid, x, out, geo, ws, stats_out);
        if (k_needed <= 14) return launch_pn_inst<TS, 14, VMAX>(s, grid, x, out, geo, ws, stats_out);
        return 2;
    }
    if (k_needed <= 8) return launch_pn_inst<TS, 8, 1>(s, grid, x, out, geo, ws, stats_out);
    if (k_needed <= 28) return launch_pn_inst<TS, 28, 1>(s, grid, x, out, geo, ws, stats_out);
    return 2;
}

inline i64 stream_blocks(i64 n, int per_thread) {
    i64 nb = (n + (i64)kRedThreads * per_thread - 1) / ((i64)kRedThreads * per_thread);
    if (nb < 1) nb = 1;
    if (nb > 2048) nb = 2048;  // 256 CUs x 8 blocks; grid-stride the rest
    return nb;
}

}  // namespace

int launch_reduce_partials(hipStream_t s, const double *partials, i64 npartials, double n, double *stats) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kRedThreads), 0, s, partials, npartials, n, stats);
    return 0;
}

i64 moments_max_partials(i64 n) { return stream_blocks(n, 4); }

int launch_moments(hipStream_t s, int dt, const void *x, i64 n, double *partials, i64 *npartials) {
    const i64 nb = stream_blocks(n, 4);
    if (dt == 0)
        hipLaunchKernelGGL((moments_kernel<float>), dim3((unsigned)nb), dim3(kRedThreads), 0, s, (const float *)x, n, partials);
    else
        hipLaunchKernelGGL((moments_kernel<double>), dim3((unsigned)nb), dim3(kRedThreads), 0, s, (const double *)x, n, partials);
    *npartials = nb;
    return 0;
}

int launch_normalize_from_partials(hipStream_t s, int dt, const void *x, i64 n, const double *partials, i64 npartials,
                                   double n_total, double eps, void *out, double *stats_out, int nblocks_hint) {
    const int vec_ok = (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
    i64 nb = stream_blocks(n, dt == 0 ? 4 : 2);
    if (nblocks_hint > 0 && nb > nblocks_hint) nb = nblocks_hint;
    if (dt == 0)
        hipLaunchKernelGGL((normalize_from_partials_kernel<float>), dim3((unsigned)nb), dim3(kRedThreads), 0, s, (const float *)x, n,
                           partials, (int)npartials, n_total, eps, (float *)out, stats_out, vec_ok);
    else
        hipLaunchKernelGGL((normalize_from_partials_kernel<double>), dim3((unsigned)nb), dim3(kRedThreads), 0, s, (const double *)x, n,
                           partials, (int)npartials, n_total, eps, (double *)out, stats_out, vec_ok);
    return 0;
}

int launch_normalize(hipStream_t s, int dt, const void *x, i64 n, const double *stats, double eps, void *out) {
    const int vec_ok = (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
    const i64 nb = stream_blocks(n, dt == 0 ? 4 : 2);
    if (dt == 0)
        hipLaunchKernelGGL((normalize_kernel<float>), dim3((unsigned)nb), dim3(kRedThreads), 0, s, (const float *)x, n, stats, eps, (float *)out, vec_ok);
    else
        hipLaunchKernelGGL((normalize_kernel<double>), dim3((unsigned)nb), dim3(kRedThreads), 0, s, (const double *)x, n, stats, eps, (double *)out, vec_ok);
    return 0;
}

// ---- grouped -----------------------------------------------------------------------------------
i64 grouped_blocks_per_group(i64 outer, i64 G, i64 inner) {
    const i64 M = outer * inner;
    i64 nb = (M + (i64)kRedThreads * 4 - 1) / ((i64)kRedThreads * 4);
    const i64 cap = 4096 / G > 1 ? 4096 / G : 1;  // ~4096 blocks in total
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return nb;
}

int launch_moments_grouped(hipStream_t s, int dt, const void *x, i64 outer, i64 G, i64 inner, double *partials, double *stats) {
    if (G > 65535) return 2;
    const i64 nb = grouped_blocks_per_group(outer, G, inner);
    const dim3 grid((unsigned)nb, (unsigned)G);
    if (dt == 0)
        hipLaunchKernelGGL((moments_grouped_kernel<float>), grid, dim3(kRedThreads), 0, s, (const float *)x, outer, G, inner, partials);
    else
        hipLaunchKernelGGL((moments_grouped_kernel<double>), grid, dim3(kRedThreads), 0, s, (const double *)x, outer, G, inner, partials);
    hipLaunchKernelGGL(reduce_partials_grouped_kernel, dim3((unsigned)G), dim3(kRedThreads), 0, s, partials, (int)nb,
                       (double)outer * (double)inner, stats);
    return 0;
}

int launch_normalize_grouped(hipStream_t s, int dt, const void *x, i64 outer, i64 G, i64 inner, const double *stats, double eps,
                             void *out) {
    if (G > 65535) return 2;
    const i64 n = outer * G * inner;
    if (G <= kMaxTableGroups) {
        const i64 nb = stream_blocks(n, 4);
        if (dt == 0)
            hipLaunchKernelGGL((normalize_grouped_kernel<float>), dim3((unsigned)nb), dim3(kRedThreads), 0, s, (const float *)x, n, G,
                               inner, stats, eps, (float *)out);
        else
            hipLaunchKernelGGL((normalize_grouped_kernel<double>), dim3((unsigned)nb), dim3(kRedThreads), 0, s, (const double *)x, n, G,
                               inner, stats, eps, (double *)out);
    } else {
        const dim3 grid((unsigned)grouped_blocks_per_group(outer, G, inner), (unsigned)G);
        if (dt == 0)
            hipLaunchKernelGGL((normalize_grouped_by_group_kernel<float>), grid, dim3(kRedThreads), 0, s, (const float *)x, outer, G,
                               inner, stats, eps, (float *)out);
        else
            hipLaunchKernelGGL((normalize_grouped_by_group_kernel<double>), grid, dim3(kRedThreads), 0, s, (const double *)x, outer, G,
                               inner, stats, eps, (double *)out);
    }
    return 0;
}

// ---- one-call standardisation ---------------------------------------------------------------------
bool standardize_small_ok(i64 n) { return n <= 65536; }

int launch_standardize_small(hipStream_t s, int dt, const void *x, i64 n, double eps, void *out, double *stats_out) {
    if (dt == 0)
        hipLaunchKernelGGL((standardize_small_kernel<float>), dim3(1), dim3(kSmallThreads), 0, s, (const float *)x, n, eps, (float *)out, stats_out);
    else
        hipLaunchKernelGGL((standardize_small_kernel<double>), dim3(1), dim3(kSmallThreads), 0, s, (const double *)x, n, eps, (double *)out, stats_out);
    return 0;
}

// ---- one launch ------------------------------------------------------------------------------------
// ws: a {epoch, error, slots} header zero-filled once by the caller and owned by this library afterwards.  ungrouped
// (pub_groups == 1): the head of a trl_workspace_bytes() workspace, shared with the FUSED GAE kernel (same grid, same slots:
// kPnReplicas <= its replica count); grouped (pub_groups == 16): its own region of kPersistentNormWsBytes.
// status 2: not eligible (more than 16 groups, period G * inner > 1024, too many values per thread, block does not fit)
int launch_standardize_persistent(hipStream_t s, int dt, const void *x, i64 outer, i64 G, i64 inner, double eps, void *out, void *ws,
                                  bool grouped_ws, double *stats_out, unsigned *host_flag) {
    if (G < 1 || G > kPnWaves || inner < 1 || G * inner > kPnThreads) return 2;
    const int P = (int)(G * inner);
    // ungrouped: the slot set of the FUSED GAE kernel, which shares the header (8 replicas)
    PnGeom geo{outer * G * inner, (int)G, (int)inner, kPnThreads - kPnThreads % P, grouped_ws ? kPnWaves : 1, grouped_ws ? kPnReplicas : gather::kHeaderReplicas,
               (double)outer * (double)inner, eps, host_flag};
    if (!grouped_ws && G != 1) return 2;
    return dt == 0 ? launch_pn_t<float>(s, x, out, geo, ws, stats_out) : launch_pn_t<double>(s, x, out, geo, ws, stats_out);
}
