// Shared device/host helpers for the gfx950 value-estimation kernels.
// Compile with -ffp-contract=off: the reference rounds every product and sum separately
// (torchrl/objectives/value/functional.py:172-178) and parity at done boundaries is bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned char u8;
typedef long long i64;

constexpr int kWave = 64;  // gfx950 wavefront

template <typename A, typename B>
struct Promote { using type = double; };
template <>
struct Promote<float, float> { using type = float; };

// ---- wave / block reductions (fixed order => deterministic) -------------------------------
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, kWave);
    return x;  // valid in lane 0
}

// Reduce (s, ss) over a block of NT threads; result valid in thread 0.
template <int NT>
__device__ __forceinline__ void block_sum2(double &s, double &ss, double *smem /* 2*NT/64 doubles */) {
    constexpr int NW = NT / kWave;
    s = wave_sum(s);
    ss = wave_sum(ss);
    if constexpr (NW > 1) {
        const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
        __syncthreads();  // smem may still be in use by the caller
        if (lane == 0) { smem[2 * w] = s; smem[2 * w + 1] = ss; }
        __syncthreads();
        if (threadIdx.x == 0) {
            s = smem[0]; ss = smem[1];
#pragma unroll
            for (int k = 1; k < NW; ++k) { s += smem[2 * k]; ss += smem[2 * k + 1]; }
        }
    }
}

// Parameter blocks (POD, passed by value to the kernels).
struct GaeArgs {
    const void *reward, *value, *next_value;
    const u8 *done, *terminated;
    const void *gamma_t, *lg_t;
    double gamma, lmbda, lg;
    i64 O, T, I;
    void *advantage, *value_target;
    double *partials;  // optional: per-block {sum, sumsq} of advantage
};

struct TdLambdaArgs {
    const void *reward, *next_value;
    const u8 *done, *terminated;
    const void *gamma_t, *lmbda_t;
    double gamma, lmbda;
    i64 O, T, I;
    void *returns;
};

struct VtraceArgs {
    const void *log_pi, *log_mu, *value, *next_value, *reward;
    const u8 *done, *terminated;
    const void *gamma_t;
    double gamma, rho_thresh, c_thresh;
    i64 O, T, I;
    void *advantage, *vs;
};

// Host-side launchers (defined in the .hip files, called from capi.hip).  Each returns the
// number of per-block partials it wrote (gae only) or a negative value when unsupported.
struct LaunchResult { int status; i64 npartials; const char *kernel; };

LaunchResult launch_gae_seq(hipStream_t s, int vdt, int cdt, const GaeArgs &a);
LaunchResult launch_td_lambda_seq(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, int rolling);
LaunchResult launch_vtrace_seq(hipStream_t s, int vdt, int cdt, const VtraceArgs &a);

LaunchResult launch_gae_scan(hipStream_t s, int vdt, int cdt, const GaeArgs &a);
LaunchResult launch_td_lambda_scan(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a);
LaunchResult launch_vtrace_scan(hipStream_t s, int vdt, int cdt, const VtraceArgs &a);
i64 scan_max_partials(i64 O, i64 T, i64 I);
i64 seq_max_partials(i64 O, i64 T, i64 I);

int launch_reduce_partials(hipStream_t s, const double *partials, i64 npartials, double n, double *stats);
int launch_moments(hipStream_t s, int dt, const void *x, i64 n, double *partials, i64 *npartials);
i64 moments_max_partials(i64 n);
int launch_normalize(hipStream_t s, int dt, const void *x, i64 n, const double *stats, double eps, void *out);
