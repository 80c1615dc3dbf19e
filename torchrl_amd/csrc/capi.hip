// extern "C" entry points declared in include/trl_value.h: argument validation, kernel-family
// selection, error reporting.  No allocation, no host synchronisation, no torch types.
#include "../../include/trl_value.h"

#include <stdarg.h>
#include <stdio.h>

#include <mutex>

#include "common.h"

namespace {

thread_local char g_err[512] = "";
thread_local const char *g_kernel = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TRL_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return TRL_OK;
}

bool dtype_ok(int d) { return d == TRL_F32 || d == TRL_F64; }

int check_dims(const char *fn, i64 O, i64 T, i64 I) {
    if (O <= 0 || T <= 0 || I <= 0) return fail(TRL_ERR_INVALID_ARGUMENT, "%s: O, T, I must be positive (got %lld, %lld, %lld)", fn, O, T, I);
    // element count must stay far from int64 overflow
    const long double n = (long double)O * (long double)T * (long double)I;
    if (n > 4.0e18L) return fail(TRL_ERR_INVALID_ARGUMENT, "%s: O*T*I too large", fn);
    return TRL_OK;
}

// workspace layout: [0, kFusedWsBytes) fused-kernel header (epoch, error, moment slots);
// [kFusedWsBytes, +64) scratch {N, sum, sumsq}; then the per-block partials.
constexpr size_t kWsStatsOffset = kFusedWsBytes;
constexpr size_t kWsPartialsOffset = kFusedWsBytes + 64;

i64 max_partials(i64 O, i64 T, i64 I) {
    i64 a = seq_max_partials(O, T, I), b = scan_max_partials(O, T, I), c = moments_max_partials(O * T * I);
    i64 m = a > b ? a : b;
    return m > c ? m : c;
}

// the look-back records follow the partials, on a 256-byte boundary
size_t lookback_offset(i64 O, i64 T, i64 I) {
    const size_t end = kWsPartialsOffset + (size_t)max_partials(O, T, I) * 2 * sizeof(double);
    return (end + 255) & ~(size_t)255;
}

// look-back region of a caller workspace (nullptr when the workspace is absent, misaligned or too small)
void *lb_region(void *workspace, size_t workspace_bytes, i64 O, i64 T, i64 I, size_t *bytes) {
    const size_t off = lookback_offset(O, T, I);
    if (!workspace || workspace_bytes <= off || ((uintptr_t)workspace & 255)) { *bytes = 0; return nullptr; }
    *bytes = workspace_bytes - off;
    return (char *)workspace + off;
}



}  // namespace

// ---- host-visible failure flag of the single-launch (FUSED) path ----------------------------------------------------------
// A fused launch whose persistent blocks were not co-resident gives up after a bounded wait, leaves the advantage
// un-normalised and raises a word in pinned host memory.  The next entry-point call on that GPU sees it WITHOUT
// synchronising anything and DEGRADES the device by itself: from then on TRL_ALGO_AUTO takes the two-launch route (scan +
// scale pass: no co-residency requirement) instead of the single launch, and that call -- after doing its own work --
// returns TRL_WARN_FUSED_DEGRADED once (python: one RuntimeWarning).  Only an explicit TRL_ALGO_FUSED request still fails
// with TRL_ERR_LAUNCH.  trl_fused_reset() re-arms the single launch.  One word per device, allocated on the first fused
// launch (never by the no-alloc entry point).
static unsigned *g_flag_host[64] = {nullptr};
static unsigned *g_flag_dev[64] = {nullptr};
static bool g_fused_degraded[64] = {false};
static std::mutex g_flag_mu;

static int current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    return dev;
}

static unsigned *fused_flag(bool may_allocate) {
    const int dev = current_device();
    if (dev < 0) return nullptr;
    std::lock_guard<std::mutex> lock(g_flag_mu);
    if (!g_flag_dev[dev] && may_allocate) {
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || !h) { (void)hipGetLastError(); return nullptr; }
        *(volatile unsigned *)h = 0u;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostFree(h); return nullptr; }
        g_flag_host[dev] = (unsigned *)h;
        g_flag_dev[dev] = (unsigned *)d;
    }
    return g_flag_dev[dev];
}

// true (flag cleared, device marked degraded) if a fused launch on the current device gave up since the last look
static bool observe_fused_failure() {
    const int dev = current_device();
    if (dev < 0) return false;
    volatile unsigned *h = g_flag_host[dev];
    if (!h || *h == 0u) return false;
    *h = 0u;
    g_fused_degraded[dev] = true;
    return true;
}

static bool fused_degraded() {
    const int dev = current_device();
    return dev >= 0 && g_fused_degraded[dev];
}

static const char kGaveUpText[] =
    "an EARLIER single-launch call (GAE + normalise, or standardize) on this GPU gave up waiting for its other blocks (its "
    "persistent grid was not co-resident: another kernel held compute units); that call left its output UN-NORMALISED";

// status of a call that did its work: TRL_OK, or -- once, on the call that noticed the give-up -- the degrade warning
static int done_ok(const char *fn, bool saw_failure) {
    if (!saw_failure) return TRL_OK;
    return fail(TRL_WARN_FUSED_DEGRADED, "%s: completed, but %s.  TRL_ALGO_AUTO now uses the multi-launch routes on this GPU "
                "(trl_fused_reset() re-arms the single launch).", fn, kGaveUpText);
}

// Shifted value layout: next_state_value == NULL, state_value is the (O, T + 1, I) buffer (trl_value.h).
// The wavefront-scan and fused kernels read it for I == 1; the SEQ kernels for any I.
static int check_shifted(const char *fn, const void *next_state_value, int64_t I, int algo) {
    if (next_state_value) return TRL_OK;
    if (I != 1 && algo != TRL_ALGO_SEQ)
        return fail(TRL_ERR_UNSUPPORTED, "%s: the shifted value layout (next_state_value == NULL) needs I == 1 or TRL_ALGO_SEQ (I = %lld)",
                    fn, (long long)I);
    return TRL_OK;
}

// I > 1 with whole outer rows inside one chunk: both the LDS-transposing SCAN kernel and the lock-step kernels
// apply.  Measured (transposing / lock-step us, fp32 unless noted): the transposing kernel when its chunk is full
// (R*T*I >= 95 % of 256*V elements) -- [8192x100x5] 23.1 / 27.8, [8192x85x6] 23.7 / 29.1, [8192x128x8] 36.0 / 39.4,
// [8192x32x8] 10.8 / 12.3, [8192x128x2] 10.6 / 10.8, [8192x32x8] fp64 20.3 / 35.8 -- except I = 4, where a lane of the
// lock-step kernels reads exactly one 16-byte vector per tensor: [8192x64x4] 10.2 / 9.8, [8192x128x4] 20.3 / 18.3,
// [8192x200x4] 35.7 / 33.6; below 95 % the lock-step kernels ([8192x100x3] 19.5 / 15.9).
static bool prefer_lockstep(int val_dtype, int coef_dtype, int64_t T, int64_t I) {
    if (I < 2 || I > 8) return false;
    const int64_t chunk = 256 * (val_dtype == TRL_F32 ? 4 : 2);
    if (T * I > chunk) return true;
    if (val_dtype == TRL_F32 && coef_dtype == TRL_F32 && I == 4) return true;
    const int64_t R = chunk / (T * I);
    return R * T * I * 100 < chunk * 95;
}

// GAE only (round 3): GAE has its own transposing kernel (gae_interleaved.h: 2 arrays instead of 5 cross LDS; rows of 64 < T <= 256
// steps as 16- / 32-lane groups in the row phase) and the rule above, measured on the generic kernel, no longer describes it.
// Transposing / lock-step us at every chunk fill (profiles/r03_gae_rows_as_dpp_rows.txt): [4096x80x5] 7.9 / 12.8, [4096x96x4]
// 7.6 / 9.9, [8192x64x4] 8.1 / 10.1, [8192x40x3] 6.7 / 8.0, [8192x60x6] 15.4 / 22.3, [4096x136x7] 16.9 / 31.7, [4096x160x3]
// 8.2 / 14.4, [4096x200x4] 15.8 / 17.8, [8192x48x4] 9.0 / 9.1; fp64 [8192x40x3] 8.3 / 12.7, [8192x24x4] 8.1 / 10.2, [4096x160x3]
// 15.4 / 25.9, [4096x99x3] 12.4 / 16.9; chunks whose starts are not 16-byte aligned too, since their full vectors became
// single element-aligned accesses (scan_core.h vload_u): [4096x127x3] 8.7 / 9.9, [4096x99x3] 8.4 / 9.4, [4096x65x7] 8.2 / 17.5.
static bool gae_prefers_transposing(int val_dtype, int coef_dtype, int64_t T, int64_t I) {
    (void)coef_dtype;  // python-float coefficients beside fp64 values are fp32: the common fp64 call
    if (I < 2) return false;
    if (T * I <= 256 * (val_dtype == TRL_F32 ? 4 : 2)) return true;
    return scan_turn_big_ok(val_dtype, T, I);  // outer rows of up to 1024 * V elements: 1024-thread blocks (gae_interleaved.h)
}
// TD(lambda) / TD(1) take the same kernel (policies TdLambdaTurn / Td1Turn).  TD(lambda), transposing / lock-step us
// (profiles/r03_interleaved_other_ops.txt): [4096x100x5] 8.8 / 11.4, [4096x80x5] 8.6 / 9.5, [8192x64x4] 8.4 / 9.8, [4096x136x7]
// 13.3 / 26.7, [4096x200x4] 14.0 / 16.5, [8192x128x4] 15.1 / 17.0, [8192x100x3] 10.9 / 12.9, fp64 [4096x100x5] 12.7 / 23.0,
// [8192x40x3] 8.8 / 10.7, [4096x99x3] 12.6 / 13.6 -- but the lock-step kernels below 75 % chunk fill at fp32: [8192x60x6]
// 15.9 / 13.3, [4096x127x3] 9.1 / 8.4.
static bool td_prefers_transposing(int val_dtype, int coef_dtype, int64_t T, int64_t I) {
    (void)coef_dtype;
    if (I < 2) return false;
    const int64_t V = val_dtype == TRL_F32 ? 4 : 2, chunk = 256 * V;
    if (T * I > chunk) return scan_turn_big_ok(val_dtype, T, I);
    if (val_dtype == TRL_F64) return true;
    int64_t R = chunk / (T * I), g = T * I, v = V;  // make_plan's rows per chunk (scan_impl.h)
    while (v) { const int64_t t = g % v; g = v; v = t; }
    const int64_t q = V / g;
    if (R >= q) R -= R % q;
    return R * T * I * 100 >= chunk * 75;
}

// V-trace on the same kernel (policy VtraceTurn), transposing / lock-step us: [4096x100x5] 15.1 / 22.0, [4096x80x5] 12.5 / 17.5,
// [4096x136x7] 26.8 / 44.3, [8192x60x6] 26.8 / 27.2, [4096x150x8] 43.8 / 49.6, [4096x100x30] 92.6 / 256.7, [4096x100x11] 41.6 / 66.8,
// fp64 [8192x40x3] 13.7 / 20.2, [4096x100x10] 50.0 / 109.6; with I <= 4 at fp32 the two are within 5 % of each other and the rule
// above stays ([4096x200x4] 25.4 / 24.4, [8192x100x3] 20.6 / 19.9, [4096x127x3] 15.0 / 12.9).
static bool vtrace_prefers_transposing(int val_dtype, int64_t T, int64_t I) {
    if (I < 2) return false;
    if (T * I > 256 * (val_dtype == TRL_F32 ? 4 : 2)) return scan_turn_big_ok(val_dtype, T, I);
    return val_dtype == TRL_F64 || I >= 5;
}

// Outer rows too long for the transposing kernel, 9 <= I <= 31: lock-step groups or the SEQ walk (coalesced only
// over i, but every (o, i) row is a lane).  Measured fp32 (SEQ / LOCKSTEP us): [64x1000x16] 381 / 20, [1024x300x16]
// 145 / 62, [4096x300x16] 281 / 218, [1024x100x31] 68 / 65, [8192x100x16] 117 / 146, [16384x64x31] 360 / 817:
// SEQ once there are enough rows to hide its per-step latency -- unless I factors as G * NI (G = 2 / 4 / 8 groups of a step
// in neighbouring lanes, lockstep_impl.h, round 3): then a wave reads whole cache lines and LOCKSTEP wins everywhere
// measured ([8192x100x16] 129 / 55, [16384x64x20] 207 / 149, [8192x100x24] 193 / 163, [4096x300x16] 281 / 82).
static bool lockstep_over_seq(int64_t O, int64_t I) {
    const bool lanes = I == 10 || I == 12 || I == 14 || I == 16 || I == 20 || I == 24 || I == 28;
    return I <= 8 || lanes || O * I < 131072;
}

// error reporting for the other translation units' entry points (xchg_kernels.hip)
int trl_internal_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// trl_gae_fwd with the reduction of the per-block moments optional: `want_partials` makes the kernels write them to the
// workspace, `stats` (may be NULL even then) receives their sum from one more single-block launch; *npartials_out tells
// the caller how many there are (trl_gae_normalize_fwd lets the scale pass reduce them itself: one launch fewer).
static int gae_fwd_impl(void *stream, int val_dtype, int coef_dtype, const void *reward, const void *state_value,
                        const void *next_state_value, const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                        const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                        int64_t O, int64_t T, int64_t I, void *advantage, void *value_target,
                        double *stats, void *workspace, size_t workspace_bytes, int algo, bool want_partials,
                        i64 *npartials_out) {
    const bool saw_failure = observe_fused_failure();
    if (!dtype_ok(val_dtype) || !dtype_ok(coef_dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_fwd: bad dtype");
    if (int rc = check_dims("trl_gae_fwd", O, T, I)) return rc;
    if (!reward || !state_value || !done || !terminated || !advantage || !value_target)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_fwd: null tensor pointer");
    if (int rc = check_shifted("trl_gae_fwd", next_state_value, I, algo)) return rc;
    if (lg_t && !gamma_t) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_fwd: lg_t given without gamma_t");
    if (want_partials) {
        if (!workspace || workspace_bytes < trl_workspace_bytes(O, T, I))
            return fail(TRL_ERR_WORKSPACE, "trl_gae_fwd: stats requested but workspace is %zu bytes, need %zu",
                        workspace_bytes, trl_workspace_bytes(O, T, I));
        if (((uintptr_t)workspace & 7) != 0) return fail(TRL_ERR_WORKSPACE, "trl_gae_fwd: workspace must be 8-byte aligned");
    }
    GaeArgs a{reward, state_value, next_state_value, done, terminated, gamma_t, lg_t, gamma, lmbda, lg,
              O, T, I, advantage, value_target, want_partials ? (double *)((char *)workspace + kWsPartialsOffset) : nullptr,
              (mask_bcast && I > 1) ? 1 : 0};
    hipStream_t s = (hipStream_t)stream;
    if (algo != TRL_ALGO_AUTO && algo != TRL_ALGO_SCAN && algo != TRL_ALGO_SEQ && algo != TRL_ALGO_LOCKSTEP)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_fwd: unknown algo %d", algo);
    LaunchResult res{TRL_ERR_UNSUPPORTED, 0, nullptr};
    size_t lb_bytes = 0;
    void *lb = lb_region(workspace, workspace_bytes, O, T, I, &lb_bytes);
    if (algo == TRL_ALGO_AUTO && prefer_lockstep(val_dtype, coef_dtype, T, I) && !gae_prefers_transposing(val_dtype, coef_dtype, T, I))
        res = launch_gae_lockstep(s, val_dtype, coef_dtype, a);
    if (res.status == TRL_ERR_UNSUPPORTED && (algo == TRL_ALGO_AUTO || algo == TRL_ALGO_SCAN))
        res = launch_gae_scan(s, val_dtype, coef_dtype, a, lb, lb_bytes);
    if (res.status == TRL_ERR_UNSUPPORTED && algo == TRL_ALGO_SCAN)
        return fail(TRL_ERR_UNSUPPORTED, "trl_gae_fwd: SCAN family does not support this shape (needs I == 1)");
    if (res.status == TRL_ERR_UNSUPPORTED && ((algo == TRL_ALGO_AUTO && lockstep_over_seq(O, I)) || algo == TRL_ALGO_LOCKSTEP))
        res = launch_gae_lockstep(s, val_dtype, coef_dtype, a);
    if (res.status == TRL_ERR_UNSUPPORTED && algo == TRL_ALGO_LOCKSTEP)
        return fail(TRL_ERR_UNSUPPORTED, "trl_gae_fwd: LOCKSTEP family needs 2 <= I <= 31");
    if (res.status == TRL_ERR_UNSUPPORTED) res = launch_gae_seq(s, val_dtype, coef_dtype, a);
    if (res.status != 0) return fail(res.status, "trl_gae_fwd: launch rejected (grid too large?)");
    if (int rc = check_launch("trl_gae_fwd")) return rc;
    if (stats) {
        launch_reduce_partials(s, (const double *)((char *)workspace + kWsPartialsOffset), res.npartials,
                               (double)O * (double)T * (double)I, stats);
        if (int rc = check_launch("trl_gae_fwd(stats)")) return rc;
    }
    if (npartials_out) *npartials_out = res.npartials;
    g_kernel = res.kernel;
    return done_ok("trl_gae_fwd", saw_failure);
}


// ---- backward entry points (bwd_kernels.hip) ----------------------------------------------------------------------------------
// family selection shared by the four: SCAN (I == 1) unless SEQ is asked for; SEQ for every other shape
template <class ScanFn, class SeqFn>
static int bwd_dispatch(const char *fn, int algo, ScanFn &&scan, SeqFn &&seq) {
    if (algo != TRL_ALGO_AUTO && algo != TRL_ALGO_SCAN && algo != TRL_ALGO_SEQ) return fail(TRL_ERR_INVALID_ARGUMENT, "%s: unknown algo %d", fn, algo);
    LaunchResult res{TRL_ERR_UNSUPPORTED, 0, nullptr};
    if (algo != TRL_ALGO_SEQ) res = scan();
    if (res.status == TRL_ERR_UNSUPPORTED && algo == TRL_ALGO_SCAN) return fail(TRL_ERR_UNSUPPORTED, "%s: SCAN family needs I == 1", fn);
    if (res.status == TRL_ERR_UNSUPPORTED) res = seq();
    if (res.status != 0) return fail(res.status, "%s: launch rejected (grid too large?)", fn);
    if (int rc = check_launch(fn)) return rc;
    g_kernel = res.kernel;
    return TRL_OK;
}

extern "C" int trl_gae_bwd(void *stream, int val_dtype, int coef_dtype, const void *grad_advantage, const void *grad_value_target,
                           const uint8_t *done, const uint8_t *terminated, int mask_bcast, const void *gamma_t, const void *lg_t,
                           double gamma, double lmbda, double lg, int64_t O, int64_t T, int64_t I, void *d_reward,
                           void *d_state_value, void *d_next_state_value, void *workspace, size_t workspace_bytes, int algo) {
    if (!dtype_ok(val_dtype) || !dtype_ok(coef_dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_bwd: bad dtype");
    if (int rc = check_dims("trl_gae_bwd", O, T, I)) return rc;
    if (!done || !terminated || (!grad_advantage && !grad_value_target)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_bwd: null tensor pointer");
    if (!d_reward && !d_state_value && !d_next_state_value) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_bwd: no output requested");
    if (lg_t && !gamma_t) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_bwd: lg_t given without gamma_t");
    GaeBwdArgs a{grad_advantage, grad_value_target, done, terminated, gamma_t, lg_t, gamma, lmbda, lg, O, T, I,
                 d_reward, d_state_value, d_next_state_value, (mask_bcast && I > 1) ? 1 : 0};
    hipStream_t s = (hipStream_t)stream;
    size_t lb_bytes = 0;
    void *lb = lb_region(workspace, workspace_bytes, O, T, I, &lb_bytes);
    return bwd_dispatch("trl_gae_bwd", algo, [&] { return launch_gae_bwd_scan(s, val_dtype, coef_dtype, a, lb, lb_bytes); },
                        [&] { return launch_gae_bwd_seq(s, val_dtype, coef_dtype, a); });
}

static int td_bwd_impl(const char *fn, int td1, void *stream, int val_dtype, int coef_dtype, const void *grad_returns,
                       const uint8_t *done, const uint8_t *terminated, int mask_bcast, const void *gamma_t, const void *lmbda_t,
                       double gamma, double lmbda, int64_t O, int64_t T, int64_t I, void *d_reward, void *d_next_state_value,
                       void *workspace, size_t workspace_bytes, int algo) {
    if (!dtype_ok(val_dtype) || !dtype_ok(coef_dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "%s: bad dtype", fn);
    if (int rc = check_dims(fn, O, T, I)) return rc;
    if (!done || !terminated || !grad_returns) return fail(TRL_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", fn);
    if (!d_reward && !d_next_state_value) return fail(TRL_ERR_INVALID_ARGUMENT, "%s: no output requested", fn);
    TdBwdArgs a{grad_returns, done, terminated, gamma_t, lmbda_t, gamma, lmbda, O, T, I, d_reward, d_next_state_value,
                (mask_bcast && I > 1) ? 1 : 0};
    hipStream_t s = (hipStream_t)stream;
    size_t lb_bytes = 0;
    void *lb = lb_region(workspace, workspace_bytes, O, T, I, &lb_bytes);
    return bwd_dispatch(fn, algo, [&] { return launch_td_bwd_scan(s, val_dtype, coef_dtype, a, td1, lb, lb_bytes); },
                        [&] { return launch_td_bwd_seq(s, val_dtype, coef_dtype, a, td1); });
}

extern "C" int trl_td_lambda_bwd(void *stream, int val_dtype, int coef_dtype, const void *grad_returns, const uint8_t *done,
                                 const uint8_t *terminated, int mask_bcast, const void *gamma_t, const void *lmbda_t, double gamma,
                                 double lmbda, int64_t O, int64_t T, int64_t I, void *d_reward, void *d_next_state_value,
                                 void *workspace, size_t workspace_bytes, int algo) {
    return td_bwd_impl("trl_td_lambda_bwd", 0, stream, val_dtype, coef_dtype, grad_returns, done, terminated, mask_bcast, gamma_t,
                       lmbda_t, gamma, lmbda, O, T, I, d_reward, d_next_state_value, workspace, workspace_bytes, algo);
}

extern "C" int trl_td1_bwd(void *stream, int val_dtype, int coef_dtype, const void *grad_returns, const uint8_t *done,
                           const uint8_t *terminated, int mask_bcast, const void *gamma_t, double gamma, int64_t O, int64_t T,
                           int64_t I, void *d_reward, void *d_next_state_value, void *workspace, size_t workspace_bytes, int algo) {
    return td_bwd_impl("trl_td1_bwd", 1, stream, val_dtype, coef_dtype, grad_returns, done, terminated, mask_bcast, gamma_t, nullptr,
                       gamma, 1.0, O, T, I, d_reward, d_next_state_value, workspace, workspace_bytes, algo);
}

extern "C" int trl_td_fixed_bwd(void *stream, int val_dtype, int coef_dtype, int td1, const void *grad_returns, const void *reward,
                                const void *next_state_value, const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                                const void *gamma_t, const void *lmbda_t, double gamma, double lmbda, int64_t O, int64_t T, int64_t I,
                                void *d_reward, void *d_next_state_value, void *d_gamma_t, void *d_lmbda_t) {
    const char *fn = "trl_td_fixed_bwd";
    if (!dtype_ok(val_dtype) || !dtype_ok(coef_dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "%s: bad dtype", fn);
    if (int rc = check_dims(fn, O, T, I)) return rc;
    if (!grad_returns || !done || !terminated) return fail(TRL_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", fn);
    if (!d_reward && !d_next_state_value && !d_gamma_t && !d_lmbda_t) return fail(TRL_ERR_INVALID_ARGUMENT, "%s: no output requested", fn);
    if ((d_gamma_t || d_lmbda_t) && (!reward || !next_state_value))
        return fail(TRL_ERR_INVALID_ARGUMENT, "%s: coefficient gradients need reward and next_state_value", fn);
    if (td1 && d_lmbda_t) return fail(TRL_ERR_INVALID_ARGUMENT, "%s: TD(1) has no lmbda", fn);
    TdFixedBwdArgs a{grad_returns, reward, next_state_value, done, terminated, gamma_t, td1 ? nullptr : lmbda_t, gamma, td1 ? 1.0 : lmbda,
                     O, T, I, d_reward, d_next_state_value, d_gamma_t, d_lmbda_t, (mask_bcast && I > 1) ? 1 : 0};
    LaunchResult res = launch_td_fixed_bwd((hipStream_t)stream, val_dtype, coef_dtype, a, td1 ? 1 : 0);
    if (res.status != 0) return fail(res.status, "%s: launch rejected (grid too large?)", fn);
    if (int rc = check_launch(fn)) return rc;
    g_kernel = res.kernel;
    return TRL_OK;
}

extern "C" int trl_vtrace_bwd(void *stream, int val_dtype, int coef_dtype, const void *grad_advantage, const void *grad_vs,
                              const void *log_pi, const void *log_mu, const void *state_value, const void *next_state_value,
                              const void *reward, const void *vs, const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                              const void *gamma_t, double gamma, double rho_thresh, double c_thresh, const void *rho_thresh_t,
                              const void *c_thresh_t, int64_t O, int64_t T, int64_t I,
                              void *d_log_pi, void *d_log_mu, void *d_state_value, void *d_next_state_value, void *d_reward,
                              void *d_gamma_t, void *workspace, size_t workspace_bytes, int algo) {
    if (!dtype_ok(val_dtype) || !dtype_ok(coef_dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_vtrace_bwd: bad dtype");
    if (int rc = check_dims("trl_vtrace_bwd", O, T, I)) return rc;
    if (!log_pi || !log_mu || !state_value || !next_state_value || !reward || !vs || !done || !terminated ||
        (!grad_advantage && !grad_vs))
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_vtrace_bwd: null tensor pointer");
    if (!d_log_pi && !d_log_mu && !d_state_value && !d_next_state_value && !d_reward && !d_gamma_t)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_vtrace_bwd: no output requested");
    VtraceBwdArgs a{grad_advantage, grad_vs, log_pi, log_mu, state_value, next_state_value, reward, vs, done, terminated, gamma_t,
                    gamma, rho_thresh, c_thresh, O, T, I, d_log_pi, d_log_mu, d_state_value, d_next_state_value, d_reward,
                    d_gamma_t, (mask_bcast && I > 1) ? 1 : 0, rho_thresh_t, c_thresh_t};
    hipStream_t s = (hipStream_t)stream;
    size_t lb_bytes = 0;
    void *lb = lb_region(workspace, workspace_bytes, O, T, I, &lb_bytes);
    return bwd_dispatch("trl_vtrace_bwd", algo, [&] { return launch_vtrace_bwd_scan(s, val_dtype, coef_dtype, a, lb, lb_bytes); },
                        [&] { return launch_vtrace_bwd_seq(s, val_dtype, coef_dtype, a); });
}


// ---- the entry point that allocates nothing -------------------------------------------------------------------------------
// Callers that cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid it) have no
// workspace to pass: this variant of trl_gae_normalize_fwd brings its own, a file-scope device array.  Every GPU the library
// runs on holds its own copy of the array (at its own address) until the process ends; the array's name is a host-side
// stand-in, so host code only ever uses the per-device address hipGetSymbolAddress returned.
constexpr size_t kStaticScratchBytes = 2u << 20;  // 2 MiB per GPU: C4 needs 0.33 MB, [1024 x 10000] (look-back records) 1.4 MB
static __device__ __align__(256) unsigned char g_static_scratch[kStaticScratchBytes];  // zero-initialised at module load

static void *static_scratch_ptr(int *err) {
    static void *ptrs[64] = {nullptr};  // per device; written once
    static std::mutex mu;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { *err = 1; return nullptr; }
    std::lock_guard<std::mutex> lock(mu);
    if (!ptrs[dev]) {
        void *p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch));  // the address on the CURRENT device
        if (e != hipSuccess || !p) { *err = 2; return nullptr; }
        ptrs[dev] = p;
    }
    *err = 0;
    return ptrs[dev];
}

static void *static_scratch_ptr_or_null() {
    int err = 0;
    return static_scratch_ptr(&err);
}


extern "C" {

int trl_abi_version(void) { return TRL_ABI_VERSION; }
const char *trl_last_error(void) { return g_err; }
const char *trl_last_kernel(void) { return g_kernel; }

size_t trl_workspace_bytes(int64_t O, int64_t T, int64_t I) {
    if (O <= 0 || T <= 0 || I <= 0) return 0;
    return lookback_offset(O, T, I) + scan_lookback_bytes(O, T, I) + 256;
}

int trl_workspace_error(const void *workspace) {
    // fused-kernel header: {epoch, error, ...}; the look-back header sits behind the partials and is
    // re-zeroed by every launch, so only the fused word persists
    unsigned words[2] = {0, 0};
    if (!workspace) {
        // the static scratch of trl_gae_normalize_fwd_noalloc on the current device: its callers have no other way to learn
        // of a give-up (that entry point allocates nothing, the pinned host flag included).  Read AND cleared here; a set
        // word also degrades the device like the host flag does.
        void *ws = static_scratch_ptr_or_null();
        if (!ws) return -1;
        if (hipMemcpy(words, ws, sizeof(words), hipMemcpyDeviceToHost) != hipSuccess) return -1;  // synchronises
        if (words[1] == 0) return 0;
        const unsigned zero = 0;
        if (hipMemcpy((char *)ws + sizeof(unsigned), &zero, sizeof(zero), hipMemcpyHostToDevice) != hipSuccess) return -1;
        const int dev = current_device();
        if (dev >= 0) g_fused_degraded[dev] = true;
        return 1;
    }
    if (hipMemcpy(words, workspace, sizeof(words), hipMemcpyDeviceToHost) != hipSuccess) return -1;  // synchronises
    return words[1] != 0 ? 1 : 0;
}

int trl_fused_degraded(void) { return fused_degraded() ? 1 : 0; }

int trl_fused_reset(void) {
    const int dev = current_device();
    if (dev < 0) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_fused_reset: no current HIP device (or ordinal >= 64)");
    g_fused_degraded[dev] = false;
    if (g_flag_host[dev]) *(volatile unsigned *)g_flag_host[dev] = 0u;
    return TRL_OK;
}

int trl_gae_fwd(void *stream, int val_dtype, int coef_dtype, const void *reward, const void *state_value,
                const void *next_state_value, const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                int64_t O, int64_t T, int64_t I, void *advantage, void *value_target,
                double *stats, void *workspace, size_t workspace_bytes, int algo) {
    return gae_fwd_impl(stream, val_dtype, coef_dtype, reward, state_value, next_state_value, done, terminated, mask_bcast, gamma_t,
                        lg_t, gamma, lmbda, lg, O, T, I, advantage, value_target, stats, workspace, workspace_bytes, algo,
                        stats != nullptr, nullptr);
}

int trl_td_lambda_fwd(void *stream, int val_dtype, int coef_dtype, const void *reward, const void *next_state_value,
                      const uint8_t *done, const uint8_t *terminated, int mask_bcast, const void *gamma_t, const void *lmbda_t,
                      double gamma, double lmbda, int rolling_gamma, int64_t O, int64_t T, int64_t I,
                      void *returns, void *workspace, size_t workspace_bytes, int algo) {
    const bool saw_failure = observe_fused_failure();
    if (!dtype_ok(val_dtype) || !dtype_ok(coef_dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_td_lambda_fwd: bad dtype");
    if (int rc = check_dims("trl_td_lambda_fwd", O, T, I)) return rc;
    if (!reward || !next_state_value || !done || !terminated || !returns)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_td_lambda_fwd: null tensor pointer");
    if (!rolling_gamma && !gamma_t && !lmbda_t)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_td_lambda_fwd: rolling_gamma=0 needs a gamma or lmbda tensor");
    TdLambdaArgs a{reward, next_state_value, done, terminated, gamma_t, lmbda_t, gamma, lmbda, O, T, I, returns,
                   (mask_bcast && I > 1) ? 1 : 0};
    hipStream_t s = (hipStream_t)stream;
    LaunchResult res{TRL_ERR_UNSUPPORTED, 0, nullptr};
    if (algo != TRL_ALGO_AUTO && algo != TRL_ALGO_SCAN && algo != TRL_ALGO_SEQ && algo != TRL_ALGO_LOCKSTEP)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_td_lambda_fwd: unknown algo %d", algo);
    size_t lb_bytes = 0;
    void *lb = lb_region(workspace, workspace_bytes, O, T, I, &lb_bytes);
    if (rolling_gamma && algo == TRL_ALGO_AUTO && prefer_lockstep(val_dtype, coef_dtype, T, I) && !td_prefers_transposing(val_dtype, coef_dtype, T, I))
        res = launch_td_lambda_lockstep(s, val_dtype, coef_dtype, a);
    if (res.status == TRL_ERR_UNSUPPORTED && rolling_gamma && (algo == TRL_ALGO_AUTO || algo == TRL_ALGO_SCAN))
        res = launch_td_lambda_scan(s, val_dtype, coef_dtype, a, lb, lb_bytes);
    if (res.status == TRL_ERR_UNSUPPORTED && algo == TRL_ALGO_SCAN)
        return fail(TRL_ERR_UNSUPPORTED, "trl_td_lambda_fwd: SCAN family does not support this case (needs I == 1, rolling gamma)");
    if (res.status == TRL_ERR_UNSUPPORTED && rolling_gamma && ((algo == TRL_ALGO_AUTO && lockstep_over_seq(O, I)) || algo == TRL_ALGO_LOCKSTEP))
        res = launch_td_lambda_lockstep(s, val_dtype, coef_dtype, a);
    if (res.status == TRL_ERR_UNSUPPORTED && algo == TRL_ALGO_LOCKSTEP)
        return fail(TRL_ERR_UNSUPPORTED, "trl_td_lambda_fwd: LOCKSTEP family needs 2 <= I <= 31, rolling gamma");
    if (res.status == TRL_ERR_UNSUPPORTED) res = launch_td_lambda_seq(s, val_dtype, coef_dtype, a, rolling_gamma);
    if (res.status != 0) return fail(res.status, "trl_td_lambda_fwd: launch rejected (grid too large?)");
    if (int rc = check_launch("trl_td_lambda_fwd")) return rc;
    g_kernel = res.kernel;
    return done_ok("trl_td_lambda_fwd", saw_failure);
}

int trl_td1_fwd(void *stream, int val_dtype, int coef_dtype, const void *reward, const void *next_state_value,
                const uint8_t *done, const uint8_t *terminated, int mask_bcast, const void *gamma_t, double gamma,
                int rolling_gamma, int64_t O, int64_t T, int64_t I, void *returns, void *workspace, size_t workspace_bytes,
                int algo) {
    const bool saw_failure = observe_fused_failure();
    if (!dtype_ok(val_dtype) || !dtype_ok(coef_dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_td1_fwd: bad dtype");
    if (int rc = check_dims("trl_td1_fwd", O, T, I)) return rc;
    if (!reward || !next_state_value || !done || !terminated || !returns)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_td1_fwd: null tensor pointer");
    if (!rolling_gamma && !gamma_t) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_td1_fwd: rolling_gamma=0 needs a gamma tensor");
    TdLambdaArgs a{reward, next_state_value, done, terminated, gamma_t, nullptr, gamma, 1.0, O, T, I, returns,
                   (mask_bcast && I > 1) ? 1 : 0};
    hipStream_t s = (hipStream_t)stream;
    LaunchResult res{TRL_ERR_UNSUPPORTED, 0, nullptr};
    if (algo != TRL_ALGO_AUTO && algo != TRL_ALGO_SCAN && algo != TRL_ALGO_SEQ && algo != TRL_ALGO_LOCKSTEP)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_td1_fwd: unknown algo %d", algo);
    size_t lb_bytes = 0;
    void *lb = lb_region(workspace, workspace_bytes, O, T, I, &lb_bytes);
    if (rolling_gamma && algo == TRL_ALGO_AUTO && prefer_lockstep(val_dtype, coef_dtype, T, I) && !td_prefers_transposing(val_dtype, coef_dtype, T, I))
        res = launch_td1_lockstep(s, val_dtype, coef_dtype, a);
    if (res.status == TRL_ERR_UNSUPPORTED && rolling_gamma && (algo == TRL_ALGO_AUTO || algo == TRL_ALGO_SCAN))
        res = launch_td1_scan(s, val_dtype, coef_dtype, a, lb, lb_bytes);
    if (res.status == TRL_ERR_UNSUPPORTED && algo == TRL_ALGO_SCAN)
        return fail(TRL_ERR_UNSUPPORTED, "trl_td1_fwd: SCAN family does not support this case (needs I == 1, rolling gamma)");
    if (res.status == TRL_ERR_UNSUPPORTED && rolling_gamma && ((algo == TRL_ALGO_AUTO && lockstep_over_seq(O, I)) || algo == TRL_ALGO_LOCKSTEP))
        res = launch_td1_lockstep(s, val_dtype, coef_dtype, a);
    if (res.status == TRL_ERR_UNSUPPORTED && algo == TRL_ALGO_LOCKSTEP)
        return fail(TRL_ERR_UNSUPPORTED, "trl_td1_fwd: LOCKSTEP family needs 2 <= I <= 31, rolling gamma");
    if (res.status == TRL_ERR_UNSUPPORTED) res = launch_td1_seq(s, val_dtype, coef_dtype, a, rolling_gamma);
    if (res.status != 0) return fail(res.status, "trl_td1_fwd: launch rejected (grid too large?)");
    if (int rc = check_launch("trl_td1_fwd")) return rc;
    g_kernel = res.kernel;
    return done_ok("trl_td1_fwd", saw_failure);
}

int trl_vtrace_fwd(void *stream, int val_dtype, int coef_dtype, const void *log_pi, const void *log_mu,
                   const void *state_value, const void *next_state_value, const void *reward,
                   const uint8_t *done, const uint8_t *terminated, int mask_bcast, const void *gamma_t, double gamma,
                   double rho_thresh, double c_thresh, const void *rho_thresh_t, const void *c_thresh_t,
                   int64_t O, int64_t T, int64_t I,
                   void *advantage, void *vs, void *workspace, size_t workspace_bytes, int algo) {
    const bool saw_failure = observe_fused_failure();
    if (!dtype_ok(val_dtype) || !dtype_ok(coef_dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_vtrace_fwd: bad dtype");
    if (int rc = check_dims("trl_vtrace_fwd", O, T, I)) return rc;
    if (!log_pi || !log_mu || !state_value || !reward || !done || !terminated || !advantage || !vs)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_vtrace_fwd: null tensor pointer");
    if (int rc = check_shifted("trl_vtrace_fwd", next_state_value, I, algo)) return rc;
    VtraceArgs a{log_pi, log_mu, state_value, next_state_value, reward, done, terminated, gamma_t,
                 gamma, rho_thresh, c_thresh, O, T, I, advantage, vs, (mask_bcast && I > 1) ? 1 : 0, rho_thresh_t, c_thresh_t};
    hipStream_t s = (hipStream_t)stream;
    LaunchResult res{TRL_ERR_UNSUPPORTED, 0, nullptr};
    if (algo != TRL_ALGO_AUTO && algo != TRL_ALGO_SCAN && algo != TRL_ALGO_SEQ && algo != TRL_ALGO_LOCKSTEP)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_vtrace_fwd: unknown algo %d", algo);
    if ((rho_thresh_t || c_thresh_t) && algo != TRL_ALGO_AUTO && algo != TRL_ALGO_SEQ)
        return fail(TRL_ERR_UNSUPPORTED, "trl_vtrace_fwd: per-element clipping thresholds run on the SEQ family only");
    if (rho_thresh_t || c_thresh_t) algo = TRL_ALGO_SEQ;
    size_t lb_bytes = 0;
    void *lb = lb_region(workspace, workspace_bytes, O, T, I, &lb_bytes);
    if (algo == TRL_ALGO_AUTO && prefer_lockstep(val_dtype, coef_dtype, T, I) && !(next_state_value && vtrace_prefers_transposing(val_dtype, T, I)))
        res = launch_vtrace_lockstep(s, val_dtype, coef_dtype, a);
    if (res.status == TRL_ERR_UNSUPPORTED && (algo == TRL_ALGO_AUTO || algo == TRL_ALGO_SCAN))
        res = launch_vtrace_scan(s, val_dtype, coef_dtype, a, lb, lb_bytes);
    if (res.status == TRL_ERR_UNSUPPORTED && algo == TRL_ALGO_SCAN)
        return fail(TRL_ERR_UNSUPPORTED, "trl_vtrace_fwd: SCAN family does not support this shape (needs I == 1)");
    if (res.status == TRL_ERR_UNSUPPORTED && ((algo == TRL_ALGO_AUTO && lockstep_over_seq(O, I)) || algo == TRL_ALGO_LOCKSTEP))
        res = launch_vtrace_lockstep(s, val_dtype, coef_dtype, a);
    if (res.status == TRL_ERR_UNSUPPORTED && algo == TRL_ALGO_LOCKSTEP)
        return fail(TRL_ERR_UNSUPPORTED, "trl_vtrace_fwd: LOCKSTEP family needs 2 <= I <= 31");
    if (res.status == TRL_ERR_UNSUPPORTED) res = launch_vtrace_seq(s, val_dtype, coef_dtype, a);
    if (res.status != 0) return fail(res.status, "trl_vtrace_fwd: launch rejected (grid too large?)");
    if (int rc = check_launch("trl_vtrace_fwd")) return rc;
    g_kernel = res.kernel;
    return done_ok("trl_vtrace_fwd", saw_failure);
}

static int gae_normalize_impl(void *stream, int val_dtype, int coef_dtype, const void *reward, const void *state_value,
                              const void *next_state_value, const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                              const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                              int64_t O, int64_t T, int64_t I, void *advantage, void *value_target, double eps,
                              double *stats, void *workspace, size_t workspace_bytes, void *comm, int algo,
                              bool allocate_flag = true) {
    const bool saw_failure = observe_fused_failure();
    if (!dtype_ok(val_dtype) || !dtype_ok(coef_dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_normalize_fwd: bad dtype");
    if (int rc = check_dims("trl_gae_normalize_fwd", O, T, I)) return rc;
    if (!reward || !state_value || !done || !terminated || !advantage || !value_target)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_normalize_fwd: null tensor pointer");
    if (int rc = check_shifted("trl_gae_normalize_fwd", next_state_value, I, algo)) return rc;
    if (lg_t && !gamma_t) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_normalize_fwd: lg_t given without gamma_t");
    if (!workspace || workspace_bytes < trl_workspace_bytes(O, T, I) || ((uintptr_t)workspace & 255) != 0)
        return fail(TRL_ERR_WORKSPACE, "trl_gae_normalize_fwd: workspace is %zu bytes, need %zu (256-byte aligned, zero-filled once)",
                    workspace_bytes, trl_workspace_bytes(O, T, I));
    if (algo != TRL_ALGO_AUTO && algo != TRL_ALGO_SCAN && algo != TRL_ALGO_SEQ && algo != TRL_ALGO_FUSED)
        return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_normalize_fwd: unknown algo %d", algo);
    hipStream_t s = (hipStream_t)stream;
    if (saw_failure && algo == TRL_ALGO_FUSED)
        return fail(TRL_ERR_LAUNCH, "trl_gae_normalize_fwd: %s.  Run the fused step alone on the GPU or pass TRL_ALGO_AUTO / TRL_ALGO_SCAN.",
                    kGaveUpText);
    // AUTO on a degraded device (a fused launch gave up there): the two-launch route, which needs no co-residency
    if ((algo == TRL_ALGO_AUTO && !fused_degraded()) || algo == TRL_ALGO_FUSED) {
        GaeArgs a{reward, state_value, next_state_value, done, terminated, gamma_t, lg_t, gamma, lmbda, lg,
                  O, T, I, advantage, value_target, nullptr, 0};
        LaunchResult res = launch_gae_fused(s, val_dtype, coef_dtype, a, eps, workspace, stats, comm, fused_flag(allocate_flag));
        if (res.status == 0) {
            if (int rc = check_launch("trl_gae_normalize_fwd(fused)")) return rc;
            g_kernel = comm ? "gae_fused_xchg" : res.kernel;
            return done_ok("trl_gae_normalize_fwd", saw_failure);
        }
        if (res.status != TRL_ERR_UNSUPPORTED) return fail(res.status, "trl_gae_normalize_fwd: fused launch rejected");
        if (algo == TRL_ALGO_FUSED)
            return fail(TRL_ERR_UNSUPPORTED, "trl_gae_normalize_fwd: FUSED family needs I == 1, T <= 256*V and at most "
                                             "8 chunks per team (advantages are kept in registers; 4 for fp32 values with fp64 coefficients)");
    }
    double *st = stats ? stats : (double *)((char *)workspace + kWsStatsOffset);
    const int algo2 = algo == TRL_ALGO_FUSED ? TRL_ALGO_AUTO : algo;
    if (!comm) {
        // two launches: the scan writes its per-block moments, the scale pass sums them itself (every block, same fixed
        // order as the reduction kernel: identical statistics) -- unless there are too many (SEQ family on many rows)
        i64 np = 0;
        if (int rc = gae_fwd_impl(stream, val_dtype, coef_dtype, reward, state_value, next_state_value, done, terminated, mask_bcast,
                                  gamma_t, lg_t, gamma, lmbda, lg, O, T, I, advantage, value_target, nullptr, workspace,
                                  workspace_bytes, algo2, true, &np))
            return rc;
        const double *partials = (const double *)((char *)workspace + kWsPartialsOffset);
        if (np <= 4096) {
            launch_normalize_from_partials(s, val_dtype, advantage, O * T * I, partials, np, (double)O * (double)T * (double)I, eps,
                                           advantage, stats, 0);
            if (int rc = check_launch("trl_gae_normalize_fwd(normalize)")) return rc;
            g_kernel = "normalize";
            return done_ok("trl_gae_normalize_fwd", saw_failure);
        }
        launch_reduce_partials(s, partials, np, (double)O * (double)T * (double)I, st);
        if (int rc = check_launch("trl_gae_normalize_fwd(stats)")) return rc;
        if (int rc = trl_normalize(stream, val_dtype, advantage, O * T * I, st, eps, advantage)) return rc;
        return done_ok("trl_gae_normalize_fwd", saw_failure);
    }
    if (int rc = trl_gae_fwd(stream, val_dtype, coef_dtype, reward, state_value, next_state_value, done, terminated, mask_bcast,
                             gamma_t, lg_t, gamma, lmbda, lg, O, T, I, advantage, value_target, st, workspace, workspace_bytes, algo2))
        return rc;
    if (int rc = trl_xchg_allreduce_stats(stream, comm, st)) return rc;
    if (int rc = trl_normalize(stream, val_dtype, advantage, O * T * I, st, eps, advantage)) return rc;
    return done_ok("trl_gae_normalize_fwd", saw_failure);
}

int trl_gae_normalize_fwd(void *stream, int val_dtype, int coef_dtype, const void *reward, const void *state_value,
                          const void *next_state_value, const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                          const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                          int64_t O, int64_t T, int64_t I, void *advantage, void *value_target, double eps,
                          double *stats, void *workspace, size_t workspace_bytes, int algo) {
    return gae_normalize_impl(stream, val_dtype, coef_dtype, reward, state_value, next_state_value, done, terminated,
                              mask_bcast, gamma_t, lg_t, gamma, lmbda, lg, O, T, I, advantage, value_target, eps, stats, workspace,
                              workspace_bytes, nullptr, algo);
}

size_t trl_static_scratch_bytes(void) { return kStaticScratchBytes; }

int trl_gae_normalize_fwd_noalloc(void *stream, int val_dtype, int coef_dtype, const void *reward, const void *state_value,
                                  const void *next_state_value, const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                                  const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                                  int64_t O, int64_t T, int64_t I, void *advantage, void *value_target, double eps,
                                  double *stats, int algo) {
    if (int rc = check_dims("trl_gae_normalize_fwd_noalloc", O, T, I)) return rc;
    const size_t need = trl_workspace_bytes(O, T, I);
    if (need > kStaticScratchBytes)
        return fail(TRL_ERR_WORKSPACE, "trl_gae_normalize_fwd_noalloc: a problem of (O, T, I) = (%lld, %lld, %lld) needs %zu bytes of "
                    "scratch, the static scratch holds %zu; use trl_gae_normalize_fwd with a caller workspace",
                    (long long)O, (long long)T, (long long)I, need, kStaticScratchBytes);
    int err = 0;
    void *ws = static_scratch_ptr(&err);
    if (!ws) return fail(TRL_ERR_LAUNCH, err == 1 ? "trl_gae_normalize_fwd_noalloc: no current HIP device (or ordinal >= 64)"
                                                  : "trl_gae_normalize_fwd_noalloc: hipGetSymbolAddress failed for the static scratch");
    return gae_normalize_impl(stream, val_dtype, coef_dtype, reward, state_value, next_state_value, done, terminated,
                              mask_bcast, gamma_t, lg_t, gamma, lmbda, lg, O, T, I, advantage, value_target, eps, stats, ws,
                              kStaticScratchBytes, nullptr, algo, /*allocate_flag=*/false);
}

int trl_gae_normalize_xchg_fwd(void *stream, int val_dtype, int coef_dtype, const void *reward, const void *state_value,
                               const void *next_state_value, const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                               const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                               int64_t O, int64_t T, int64_t I, void *advantage, void *value_target, double eps,
                               double *stats, void *workspace, size_t workspace_bytes, void *comm, int algo) {
    if (!comm) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_gae_normalize_xchg_fwd: null communicator");
    return gae_normalize_impl(stream, val_dtype, coef_dtype, reward, state_value, next_state_value, done, terminated,
                              mask_bcast, gamma_t, lg_t, gamma, lmbda, lg, O, T, I, advantage, value_target, eps, stats, workspace,
                              workspace_bytes, comm, algo);
}

int trl_normalize(void *stream, int dtype, const void *x, int64_t n, const double *stats, double eps, void *out) {
    const bool saw_failure = observe_fused_failure();
    if (!dtype_ok(dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_normalize: bad dtype");
    if (n <= 0 || !x || !out || !stats) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_normalize: null pointer or n <= 0");
    launch_normalize((hipStream_t)stream, dtype, x, n, stats, eps, out);
    if (int rc = check_launch("trl_normalize")) return rc;
    g_kernel = "normalize";
    return done_ok("trl_normalize", saw_failure);
}

int trl_standardize(void *stream, int dtype, const void *x, int64_t n, double eps, void *out, double *stats,
                    void *workspace, size_t workspace_bytes) {
    const bool saw_failure = observe_fused_failure();
    if (!dtype_ok(dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_standardize: bad dtype");
    if (n <= 0 || !x || !out) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_standardize: null pointer or n <= 0");
    if (standardize_small_ok(n)) {
        launch_standardize_small((hipStream_t)stream, dtype, x, n, eps, out, stats);
        if (int rc = check_launch("trl_standardize")) return rc;
        g_kernel = "standardize_small";
        return done_ok("trl_standardize", saw_failure);
    }
    if (!workspace || workspace_bytes < trl_workspace_bytes(n, 1, 1) || ((uintptr_t)workspace & 7))
        return fail(TRL_ERR_WORKSPACE, "trl_standardize: workspace is %zu bytes, need %zu (8-byte aligned)", workspace_bytes,
                    trl_workspace_bytes(n, 1, 1));
    // one persistent launch (values kept in registers across an in-launch gather of the block moments; the workspace head
    // is the same {epoch, error, slots} header the single-launch GAE + normalise uses) unless the device is degraded
    if (!fused_degraded() && ((uintptr_t)workspace & 255) == 0) {
        const int rc = launch_standardize_persistent((hipStream_t)stream, dtype, x, n, 1, 1, eps, out, workspace, false, stats, fused_flag(true));
        if (rc == 0) {
            if (int rc2 = check_launch("trl_standardize(persistent)")) return rc2;
            g_kernel = "standardize_persistent";
            return done_ok("trl_standardize", saw_failure);
        }
    }
    // two launches: per-block moments, then a scale pass whose blocks sum the (<= 2048) partials themselves
    i64 np = 0;
    double *partials = (double *)((char *)workspace + kWsPartialsOffset);
    launch_moments((hipStream_t)stream, dtype, x, n, partials, &np);
    if (int rc = check_launch("trl_standardize(moments)")) return rc;
    launch_normalize_from_partials((hipStream_t)stream, dtype, x, n, partials, np, (double)n, eps, out, stats, 0);
    if (int rc = check_launch("trl_standardize(normalize)")) return rc;
    g_kernel = "normalize";
    return done_ok("trl_standardize", saw_failure);
}

// grouped workspace: [0, kPersistentNormWsBytes) {epoch, error, slots} of the one-launch route, then the per-block partials
// of the multi-launch route
int trl_standardize_grouped(void *stream, int dtype, const void *x, int64_t outer, int64_t G, int64_t inner, double eps, void *out,
                            double *stats, void *workspace, size_t workspace_bytes) {
    const bool saw_failure = observe_fused_failure();
    if (dtype_ok(dtype) && outer > 0 && G > 0 && inner > 0 && x && out && stats && workspace && ((uintptr_t)workspace & 255) == 0 &&
        workspace_bytes >= trl_grouped_workspace_bytes(outer, G, inner) && !fused_degraded() &&
        check_dims("trl_standardize_grouped", outer, G, inner) == TRL_OK) {
        const int rc = launch_standardize_persistent((hipStream_t)stream, dtype, x, outer, G, inner, eps, out, workspace, true, stats,
                                                     fused_flag(true));
        if (rc == 0) {
            if (int rc2 = check_launch("trl_standardize_grouped(persistent)")) return rc2;
            g_kernel = "standardize_grouped_persistent";
            return done_ok("trl_standardize_grouped", saw_failure);
        }
    }
    if (int rc = trl_moments_grouped(stream, dtype, x, outer, G, inner, stats, workspace, workspace_bytes)) return rc;
    if (int rc = trl_normalize_grouped(stream, dtype, x, outer, G, inner, stats, eps, out)) return rc;
    return done_ok("trl_standardize_grouped", saw_failure);
}

size_t trl_grouped_workspace_bytes(int64_t outer, int64_t G, int64_t inner) {
    if (outer <= 0 || G <= 0 || inner <= 0) return 0;
    return kPersistentNormWsBytes + (size_t)G * (size_t)grouped_blocks_per_group(outer, G, inner) * 2 * sizeof(double) + 256;
}

int trl_moments_grouped(void *stream, int dtype, const void *x, int64_t outer, int64_t G, int64_t inner, double *stats,
                        void *workspace, size_t workspace_bytes) {
    if (!dtype_ok(dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_moments_grouped: bad dtype");
    if (int rc = check_dims("trl_moments_grouped", outer, G, inner)) return rc;
    if (!x || !stats) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_moments_grouped: null pointer");
    if (G > 65535) return fail(TRL_ERR_UNSUPPORTED, "trl_moments_grouped: at most 65535 groups (got %lld)", (long long)G);
    if (!workspace || workspace_bytes < trl_grouped_workspace_bytes(outer, G, inner) || ((uintptr_t)workspace & 7))
        return fail(TRL_ERR_WORKSPACE, "trl_moments_grouped: workspace is %zu bytes, need %zu (8-byte aligned)", workspace_bytes,
                    trl_grouped_workspace_bytes(outer, G, inner));
    launch_moments_grouped((hipStream_t)stream, dtype, x, outer, G, inner, (double *)((char *)workspace + kPersistentNormWsBytes), stats);
    if (int rc = check_launch("trl_moments_grouped")) return rc;
    g_kernel = "moments_grouped";
    return TRL_OK;
}

int trl_normalize_grouped(void *stream, int dtype, const void *x, int64_t outer, int64_t G, int64_t inner, const double *stats,
                          double eps, void *out) {
    if (!dtype_ok(dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_normalize_grouped: bad dtype");
    if (int rc = check_dims("trl_normalize_grouped", outer, G, inner)) return rc;
    if (!x || !stats || !out) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_normalize_grouped: null pointer");
    if (G > 65535) return fail(TRL_ERR_UNSUPPORTED, "trl_normalize_grouped: at most 65535 groups (got %lld)", (long long)G);
    launch_normalize_grouped((hipStream_t)stream, dtype, x, outer, G, inner, stats, eps, out);
    if (int rc = check_launch("trl_normalize_grouped")) return rc;
    g_kernel = "normalize_grouped";
    return TRL_OK;
}

int trl_moments(void *stream, int dtype, const void *x, int64_t n, double *stats, void *workspace, size_t workspace_bytes) {
    if (!dtype_ok(dtype)) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_moments: bad dtype");
    if (n <= 0 || !x || !stats) return fail(TRL_ERR_INVALID_ARGUMENT, "trl_moments: null pointer or n <= 0");
    if (!workspace || workspace_bytes < trl_workspace_bytes(n, 1, 1) || ((uintptr_t)workspace & 7))
        return fail(TRL_ERR_WORKSPACE, "trl_moments: workspace is %zu bytes, need %zu (8-byte aligned)", workspace_bytes,
                    trl_workspace_bytes(n, 1, 1));
    i64 np = 0;
    double *partials = (double *)((char *)workspace + kWsPartialsOffset);
    launch_moments((hipStream_t)stream, dtype, x, n, partials, &np);
    if (int rc = check_launch("trl_moments")) return rc;
    launch_reduce_partials((hipStream_t)stream, partials, np, (double)n, stats);
    if (int rc = check_launch("trl_moments(reduce)")) return rc;
    g_kernel = "moments";
    return TRL_OK;
}

}  // extern "C"
