// Advantage statistics and normalisation.
// Replaces `(adv - adv.mean()) / adv.std().clamp_min(eps)` (reference:
// torchrl/objectives/value/advantages.py:1459-1463, 1828-1832; torchrl/_utils.py:901-912).
//
// Statistics travel as {N, sum, sumsq} in fp64 so that a batch sharded over GPUs needs one
// 24-byte all-reduce.  All reductions run in a fixed order (per-lane strided accumulation ->
// wave shuffle tree -> per-block partial -> single-block final tree): bitwise reproducible.
#include "common.h"

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
