// Shared device/host helpers for the gfx950 value-estimation kernels.
// Compile with -ffp-contract=off: the reference rounds every product and sum separately
// (torchrl/objectives/value/functional.py:172-178) and parity at done boundaries is bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

typedef unsigned char u8;
typedef long long i64;

constexpr int kWave = 64;  // gfx950 wavefront

template <typename A, typename B>
struct Promote { using type = double; };
template <>
struct Promote<float, float> { using type = float; };

// ---- DPP lane movement (no LDS crossbar, a few cycles instead of a ds_bpermute round trip) ------
// CTRL: quad_perm 0x00-0xff, row_shl:n 0x100+n (lane i reads lane i+n of its 16-lane row),
// row_shr:n 0x110+n, wave_shl:1 0x130 (lane i reads lane i+1 across the wave), row_mirror 0x140,
// row_half_mirror 0x141.  Lanes without a source read 0 (bound_ctrl).
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, dpp_mov<CTRL>(__builtin_bit_cast(int, x)));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = dpp_mov<CTRL>((int)(b & 0xffffffffLL)), hi = dpp_mov<CTRL>((int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
// value of a FIXED lane as a wave-uniform scalar (v_readlane_b32: no LDS crossbar, unlike __shfl / ds_bpermute)
__device__ __forceinline__ int read_lane(int x, int lane) { return __builtin_amdgcn_readlane(x, lane); }
__device__ __forceinline__ float read_lane(float x, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}
__device__ __forceinline__ double readlane_d(double x, int lane) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double read_lane(double x, int lane) { return readlane_d(x, lane); }

// ---- wave / block reductions (fixed order => deterministic) -------------------------------
// Sum over the 64 lanes; the result is wave-uniform.  Butterfly inside each 16-lane row with
// DPP (xor 1, xor 2, half-mirror, mirror), then the four row totals are added in lane order.
__device__ __forceinline__ double wave_sum(double x) {
    x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_mov<0x141>(x);  // row_half_mirror
    x += dpp_mov<0x140>(x);  // row_mirror
    return ((readlane_d(x, 0) + readlane_d(x, 16)) + readlane_d(x, 32)) + readlane_d(x, 48);
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

// mean / clamped std in the storage dtype from {N, sum, sumsq}: what
// `x.mean()` / `x.std().clamp_min(eps)` give (unbiased std; advantages.py:1459-1463).
template <typename TS>
__host__ __device__ __forceinline__ void mean_std(double n, double s, double ss, double eps, TS &mean_s, TS &std_s) {
    const double mean = s / n;
    // unbiased, as torch.std; one element: NaN as torch gives (0 / 0 when the sums are exact -- said explicitly because the
    // in-launch gathers truncate the sums to 45 mantissa bits, moment_gather.h)
    double var = n == 1.0 ? __builtin_nan("") : (ss - n * mean * mean) / (n - 1.0);
    if (var < 0.0) var = 0.0;                         // rounding only; NaN stays NaN
    mean_s = (TS)mean;
    std_s = (TS)sqrt(var);
    const TS eps_s = (TS)eps;
    if (std_s < eps_s) std_s = eps_s;                 // clamp_min keeps NaN
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
    int mask_bcast;    // 1: done / terminated are (O, T) arrays shared by the I feature elements of a step
};

struct TdLambdaArgs {
    const void *reward, *next_value;
    const u8 *done, *terminated;
    const void *gamma_t, *lmbda_t;
    double gamma, lmbda;
    i64 O, T, I;
    void *returns;
    int mask_bcast;
};

struct VtraceArgs {
    const void *log_pi, *log_mu, *value, *next_value, *reward;
    const u8 *done, *terminated;
    const void *gamma_t;
    double gamma, rho_thresh, c_thresh;
    i64 O, T, I;
    void *advantage, *vs;
    int mask_bcast;
    const void *rho_t = nullptr, *c_t = nullptr;  // optional per-element clipping thresholds (value dtype): SEQ family only
};

// ---- backward (adjoint) parameter blocks: bwd_kernels.hip ------------------------------------------------------
// Optional pointers may be null: a null incoming gradient is a tensor of zeros, a null output is not written.
struct GaeBwdArgs {
    const void *g_adv, *g_vt;  // incoming gradients w.r.t. advantage / value_target (value dtype)
    const u8 *done, *terminated;
    const void *gamma_t, *lg_t;
    double gamma, lmbda, lg;
    i64 O, T, I;
    void *d_reward, *d_value, *d_next_value;  // value dtype; d_reward is the adjoint G itself
    int mask_bcast;
};

struct TdBwdArgs {  // TD(lambda) (rolling gamma) and TD(1) (lmbda unused)
    const void *g_ret;
    const u8 *done, *terminated;
    const void *gamma_t, *lmbda_t;
    double gamma, lmbda;
    i64 O, T, I;
    void *d_reward, *d_next_value;
    int mask_bcast;
};

struct TdFixedBwdArgs {  // rolling_gamma=False forms of TD(lambda) / TD(1) (bwd_fixed_kernels.hip)
    const void *g_ret, *reward, *next_value;
    const u8 *done, *terminated;
    const void *gamma_t, *lmbda_t;
    double gamma, lmbda;
    i64 O, T, I;
    void *d_reward, *d_next_value;  // value dtype
    void *d_gamma, *d_lmbda;        // compute dtype, per element (the gradient w.r.t. the coefficient of start k)
    int mask_bcast;
};

struct VtraceBwdArgs {
    const void *g_adv, *g_vs;  // compute dtype (what trl_vtrace_fwd wrote)
    const void *log_pi, *log_mu, *value, *next_value, *reward;
    const void *vs;            // saved forward output (compute dtype)
    const u8 *done, *terminated;
    const void *gamma_t;
    double gamma, rho_thresh, c_thresh;
    i64 O, T, I;
    void *d_log_pi, *d_log_mu, *d_value, *d_next_value, *d_reward;  // value dtype
    void *d_gamma_t;                                                // compute dtype
    int mask_bcast;
    const void *rho_t = nullptr, *c_t = nullptr;  // optional per-element clipping thresholds (value dtype): SEQ family only
};

// Host-side launchers (defined in the .hip files, called from capi.hip).  Each returns the
// number of per-block partials it wrote (gae only) or a negative value when unsupported.
struct LaunchResult { int status; i64 npartials; const char *kernel; };

LaunchResult launch_gae_seq(hipStream_t s, int vdt, int cdt, const GaeArgs &a);
LaunchResult launch_td_lambda_seq(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, int rolling);
LaunchResult launch_vtrace_seq(hipStream_t s, int vdt, int cdt, const VtraceArgs &a);
// TD(1): TdLambdaArgs with lmbda unused
LaunchResult launch_td1_seq(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, int rolling);
LaunchResult launch_td1_scan(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, void *lb_ws, size_t lb_bytes);
LaunchResult launch_td1_lockstep(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a);
// adjoints (bwd_kernels.hip): SCAN family for I == 1 (status 2 otherwise), SEQ for every shape
LaunchResult launch_gae_bwd_scan(hipStream_t s, int vdt, int cdt, const GaeBwdArgs &a, void *lb_ws, size_t lb_bytes);
LaunchResult launch_gae_bwd_seq(hipStream_t s, int vdt, int cdt, const GaeBwdArgs &a);
LaunchResult launch_td_bwd_scan(hipStream_t s, int vdt, int cdt, const TdBwdArgs &a, int td1, void *lb_ws, size_t lb_bytes);
LaunchResult launch_td_bwd_seq(hipStream_t s, int vdt, int cdt, const TdBwdArgs &a, int td1);
LaunchResult launch_td_fixed_bwd(hipStream_t s, int vdt, int cdt, const TdFixedBwdArgs &a, int td1);
LaunchResult launch_vtrace_bwd_scan(hipStream_t s, int vdt, int cdt, const VtraceBwdArgs &a, void *lb_ws, size_t lb_bytes);
LaunchResult launch_vtrace_bwd_seq(hipStream_t s, int vdt, int cdt, const VtraceBwdArgs &a);

// lb_ws / lb_bytes: optional look-back region of the workspace (scan_lookback_bytes), else nullptr / 0
LaunchResult launch_gae_scan(hipStream_t s, int vdt, int cdt, const GaeArgs &a, void *lb_ws, size_t lb_bytes);
LaunchResult launch_td_lambda_scan(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, void *lb_ws, size_t lb_bytes);
LaunchResult launch_vtrace_scan(hipStream_t s, int vdt, int cdt, const VtraceArgs &a, void *lb_ws, size_t lb_bytes);
size_t scan_lookback_bytes(i64 O, i64 T, i64 I);
// lock-step kernels: 2 <= I <= 31 (groups of <= 8 elements), any T (status 2 otherwise); gae writes <= 4096 partials
LaunchResult launch_gae_lockstep(hipStream_t s, int vdt, int cdt, const GaeArgs &a);
LaunchResult launch_td_lambda_lockstep(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a);
LaunchResult launch_vtrace_lockstep(hipStream_t s, int vdt, int cdt, const VtraceArgs &a);
i64 scan_max_partials(i64 O, i64 T, i64 I);
// feature dims after time on the map-forming transposing kernels (gae_interleaved.h; instantiated in scan_kernels.hip); status 2: not their shape
LaunchResult launch_gae_turn(hipStream_t s, int vdt, int cdt, const GaeArgs &a, bool aligned);
LaunchResult launch_td_lambda_turn(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, bool aligned);
LaunchResult launch_td1_turn(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, bool aligned);
LaunchResult launch_vtrace_turn(hipStream_t s, int vdt, int cdt, const VtraceArgs &a, bool aligned);
bool scan_turn_big_ok(int vdt, i64 T, i64 I);  // outer rows of up to 1024 * V elements on the 1024-thread transposing kernel
i64 seq_max_partials(i64 O, i64 T, i64 I);

int launch_reduce_partials(hipStream_t s, const double *partials, i64 npartials, double n, double *stats);
int launch_moments(hipStream_t s, int dt, const void *x, i64 n, double *partials, i64 *npartials);
i64 moments_max_partials(i64 n);
int launch_normalize(hipStream_t s, int dt, const void *x, i64 n, const double *stats, double eps, void *out);
bool standardize_small_ok(i64 n);
int launch_standardize_small(hipStream_t s, int dt, const void *x, i64 n, double eps, void *out, double *stats_out);
// grouped statistics (x seen as (outer, G, inner); group g = (f / inner) % G); status 2 = G too large
i64 grouped_blocks_per_group(i64 outer, i64 G, i64 inner);
int launch_moments_grouped(hipStream_t s, int dt, const void *x, i64 outer, i64 G, i64 inner, double *partials, double *stats);
int launch_normalize_grouped(hipStream_t s, int dt, const void *x, i64 outer, i64 G, i64 inner, const double *stats, double eps,
                             void *out);
int launch_normalize_from_partials(hipStream_t s, int dt, const void *x, i64 n, const double *partials, i64 npartials,
                                   double n_total, double eps, void *out, double *stats_out, int nblocks_hint);

// standardize in one persistent launch (norm_kernels.hip); status 2 = not eligible.  ws: {epoch, error, slots} header
// (zero-filled once): the head of a trl_workspace_bytes() workspace (grouped_ws = false, G == 1) or a region of
// kPersistentNormWsBytes (grouped_ws = true, G <= 16)
constexpr size_t kPersistentNormWsBytes = 256 + (size_t)16 * 4 * 256 * 16;  // header + 16 groups x 4 replicas x 256 blocks x 16 B
int launch_standardize_persistent(hipStream_t s, int dt, const void *x, i64 outer, i64 G, i64 inner, double eps, void *out, void *ws,
                                  bool grouped_ws, double *stats_out, unsigned *host_flag);

// Fused single-launch GAE + normalisation (fused_kernels.hip).  status 2 = shape not eligible.
constexpr size_t kFusedWsBytes = 256 + 4 * 512 * 128;  // header + 4 replicas of one 128-byte slot line per block, at the start of the workspace
// comm: communicator of trl_xchg_open when the batch is sharded over GPUs (moments summed over ranks), else nullptr
// host_flag: device view of a pinned host word the kernel raises when it gives up waiting (may be null)
LaunchResult launch_gae_fused(hipStream_t s, int vdt, int cdt, const GaeArgs &a, double eps, void *ws, double *stats_out, void *comm,
                              unsigned *host_flag);
