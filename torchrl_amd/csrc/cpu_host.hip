// Host entry points (trl_*_cpu, include/trl_value.h ABI 7): the estimators for tensors that live in HOST memory.
//
// The reference runs on whatever device its tensors are on (torchrl/objectives/value/functional.py:163-170 allocates on
// state_value.device; BASELINE config 1 is a CPU call), so a drop-in must take CPU tensors too.  This is the product's own
// host path, not the test oracle: it runs the very row functions of the SEQ kernel family (seq_rows.h, __host__ __device__)
// -- the reference's per-step expressions in the reference's order, compiled with -ffp-contract=off -- over the (o, i) rows,
// split across a few worker threads.  Results are bit-identical to the SEQ kernels (up to libm's exp in V-trace) and to the
// reference's sequential loop.  No HIP call is made here: these entry points work on a machine without a GPU.
//
// Moments {N, sum, sumsq} of the advantage: fp64, accumulated per block of 64 rows, blocks summed in index order -- the
// result does not depend on the number of threads.
#include "../../include/trl_value.h"

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "seq_rows.h"

int trl_internal_fail(int code, const char *fmt, ...);  // capi.hip: sets trl_last_error()

namespace {

std::atomic<int> g_threads{0};  // 0: decide from the environment on first use

int env_int(const char *name) {
    const char *v = getenv(name);
    if (!v || !*v) return 0;
    const int n = atoi(v);
    return n > 0 ? n : 0;
}

int worker_threads() {
    int n = g_threads.load(std::memory_order_relaxed);
    if (n > 0) return n;
    n = env_int("TRL_CPU_THREADS");
    if (n == 0) n = env_int("OMP_NUM_THREADS");
    if (n == 0) {
        n = (int)std::thread::hardware_concurrency();
        if (n > 16) n = 16;  // a shared machine reports every core; 16 is the GPU boxes' per-job share
    }
    if (n < 1) n = 1;
    if (n > 256) n = 256;
    g_threads.store(n, std::memory_order_relaxed);
    return n;
}

// fn(block) for block in [0, nblocks): blocks are handed out dynamically; small jobs run on the calling thread
template <class F>
void parallel_blocks(i64 nblocks, i64 work_per_block, F &&fn) {
    const i64 total = nblocks * work_per_block;
    int nt = worker_threads();
    if (total < (1 << 16)) nt = 1;
    if ((i64)nt > nblocks) nt = (int)nblocks;
    if (nt <= 1) {
        for (i64 b = 0; b < nblocks; ++b) fn(b);
        return;
    }
    std::atomic<i64> next{0};
    // a claim covers several blocks so that the shared counter is touched rarely
    const i64 grain = std::max<i64>(1, nblocks / ((i64)nt * 8));
    auto body = [&] {
        for (;;) {
            const i64 b0 = next.fetch_add(grain, std::memory_order_relaxed);
            if (b0 >= nblocks) return;
            const i64 b1 = std::min(nblocks, b0 + grain);
            for (i64 b = b0; b < b1; ++b) fn(b);
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    for (int t = 1; t < nt; ++t) pool.emplace_back(body);
    body();
    for (auto &th : pool) th.join();
}

constexpr i64 kRowsPerBlock = 64;

bool dtype_ok(int d) { return d == TRL_F32 || d == TRL_F64; }

int check_dims(const char *fn, i64 O, i64 T, i64 I) {
    if (O <= 0 || T <= 0 || I <= 0)
        return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: O, T, I must be positive (got %lld, %lld, %lld)", fn, (long long)O, (long long)T,
                                 (long long)I);
    const long double n = (long double)O * (long double)T * (long double)I;
    if (n > 4.0e18L) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: O*T*I too large", fn);
    return TRL_OK;
}

template <typename F>
int dispatch_types(const char *fn, int vdt, int cdt, F &&f) {
    if (!dtype_ok(vdt) || !dtype_ok(cdt)) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: bad dtype", fn);
    if (vdt == 0 && cdt == 0) f(float{}, float{});
    else if (vdt == 1 && cdt == 1) f(double{}, double{});
    else if (vdt == 1 && cdt == 0) f(double{}, float{});
    else f(float{}, double{});
    return TRL_OK;
}

// rows (o, i) in blocks of kRowsPerBlock: fn(row index)
template <class F>
void for_rows(i64 nrows, i64 T, F &&fn) {
    const i64 nb = (nrows + kRowsPerBlock - 1) / kRowsPerBlock;
    parallel_blocks(nb, kRowsPerBlock * T, [&](i64 b) {
        const i64 r1 = std::min(nrows, (b + 1) * kRowsPerBlock);
        for (i64 row = b * kRowsPerBlock; row < r1; ++row) fn(row);
    });
}

// GAE over all rows; stats (optional, host double[3]) <- {N, sum, sumsq} of the advantage
int gae_cpu(const char *fn, int vdt, int cdt, const GaeArgs &a, double *stats) {
    const i64 nrows = a.O * a.I;
    const i64 nb = (nrows + kRowsPerBlock - 1) / kRowsPerBlock;
    std::vector<double> partials(stats ? (size_t)(2 * nb) : 0);
    if (int rc = dispatch_types(fn, vdt, cdt, [&](auto ts, auto tg) {
            using TS = decltype(ts);
            using TG = decltype(tg);
            parallel_blocks(nb, kRowsPerBlock * a.T, [&](i64 b) {
                const i64 r1 = std::min(nrows, (b + 1) * kRowsPerBlock);
                double s = 0.0, ss = 0.0;
                for (i64 row = b * kRowsPerBlock; row < r1; ++row) {
                    const i64 o = row / a.I;
                    double rs = 0.0, rss = 0.0;
                    seqrow::gae_row<TS, TG>(a, o, row - o * a.I, rs, rss);
                    s += rs;
                    ss += rss;
                }
                if (stats) { partials[2 * b] = s; partials[2 * b + 1] = ss; }
            });
        }))
        return rc;
    if (stats) {
        double s = 0.0, ss = 0.0;
        for (i64 b = 0; b < nb; ++b) { s += partials[2 * b]; ss += partials[2 * b + 1]; }
        stats[0] = (double)a.O * (double)a.T * (double)a.I;
        stats[1] = s;
        stats[2] = ss;
    }
    return TRL_OK;
}

template <typename TS>
void normalize_t(const TS *x, i64 n, const double *stats, double eps, TS *out) {
    TS mean_s, std_s;
    mean_std<TS>(stats[0], stats[1], stats[2], eps, mean_s, std_s);
    constexpr i64 kBlock = 1 << 14;
    parallel_blocks((n + kBlock - 1) / kBlock, kBlock, [&](i64 b) {
        const i64 e1 = std::min(n, (b + 1) * kBlock);
        for (i64 e = b * kBlock; e < e1; ++e) out[e] = (x[e] - mean_s) / std_s;
    });
}

template <typename TS>
void moments_t(const TS *x, i64 n, double *stats) {
    constexpr i64 kBlock = 1 << 12;
    const i64 nb = (n + kBlock - 1) / kBlock;
    std::vector<double> partials((size_t)(2 * nb));
    parallel_blocks(nb, kBlock, [&](i64 b) {
        const i64 e1 = std::min(n, (b + 1) * kBlock);
        double s = 0.0, ss = 0.0;
        for (i64 e = b * kBlock; e < e1; ++e) {
            s += (double)x[e];
            ss += (double)x[e] * (double)x[e];
        }
        partials[2 * b] = s;
        partials[2 * b + 1] = ss;
    });
    double s = 0.0, ss = 0.0;
    for (i64 b = 0; b < nb; ++b) { s += partials[2 * b]; ss += partials[2 * b + 1]; }
    stats[0] = (double)n;
    stats[1] = s;
    stats[2] = ss;
}

}  // namespace

extern "C" {

int trl_cpu_threads(void) { return worker_threads(); }

int trl_set_cpu_threads(int n) {
    if (n < 0 || n > 256) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "trl_set_cpu_threads: 0 (from the environment) .. 256, got %d", n);
    g_threads.store(n, std::memory_order_relaxed);
    return TRL_OK;
}

int trl_gae_fwd_cpu(int val_dtype, int coef_dtype, const void *reward, const void *state_value, const void *next_state_value,
                    const uint8_t *done, const uint8_t *terminated, int mask_bcast, const void *gamma_t, const void *lg_t, double gamma,
                    double lmbda, double lg, int64_t O, int64_t T, int64_t I, void *advantage, void *value_target, double *stats) {
    const char *fn = "trl_gae_fwd_cpu";
    if (int rc = check_dims(fn, O, T, I)) return rc;
    if (!reward || !state_value || !done || !terminated || !advantage || !value_target)
        return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", fn);
    if (lg_t && !gamma_t) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: lg_t given without gamma_t", fn);
    GaeArgs a{reward, state_value, next_state_value, done, terminated, gamma_t, lg_t, gamma, lmbda, lg, O, T, I,
              advantage, value_target, nullptr, (mask_bcast && I > 1) ? 1 : 0};
    return gae_cpu(fn, val_dtype, coef_dtype, a, stats);
}

int trl_normalize_cpu(int dtype, const void *x, int64_t n, const double *stats, double eps, void *out) {
    if (!dtype_ok(dtype)) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "trl_normalize_cpu: bad dtype");
    if (n <= 0 || !x || !out || !stats) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "trl_normalize_cpu: null pointer or n <= 0");
    if (dtype == TRL_F32) normalize_t<float>((const float *)x, n, stats, eps, (float *)out);
    else normalize_t<double>((const double *)x, n, stats, eps, (double *)out);
    return TRL_OK;
}

int trl_moments_cpu(int dtype, const void *x, int64_t n, double *stats) {
    if (!dtype_ok(dtype)) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "trl_moments_cpu: bad dtype");
    if (n <= 0 || !x || !stats) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "trl_moments_cpu: null pointer or n <= 0");
    if (dtype == TRL_F32) moments_t<float>((const float *)x, n, stats);
    else moments_t<double>((const double *)x, n, stats);
    return TRL_OK;
}

int trl_gae_normalize_fwd_cpu(int val_dtype, int coef_dtype, const void *reward, const void *state_value, const void *next_state_value,
                              const uint8_t *done, const uint8_t *terminated, int mask_bcast, const void *gamma_t, const void *lg_t,
                              double gamma, double lmbda, double lg, int64_t O, int64_t T, int64_t I, void *advantage, void *value_target,
                              double eps, double *stats) {
    double local[3];
    double *st = stats ? stats : local;
    if (int rc = trl_gae_fwd_cpu(val_dtype, coef_dtype, reward, state_value, next_state_value, done, terminated, mask_bcast, gamma_t, lg_t,
                                 gamma, lmbda, lg, O, T, I, advantage, value_target, st))
        return rc;
    return trl_normalize_cpu(val_dtype, advantage, O * T * I, st, eps, advantage);
}

int trl_td_lambda_fwd_cpu(int val_dtype, int coef_dtype, const void *reward, const void *next_state_value, const uint8_t *done,
                          const uint8_t *terminated, int mask_bcast, const void *gamma_t, const void *lmbda_t, double gamma, double lmbda,
                          int rolling_gamma, int64_t O, int64_t T, int64_t I, void *returns) {
    const char *fn = "trl_td_lambda_fwd_cpu";
    if (int rc = check_dims(fn, O, T, I)) return rc;
    if (!reward || !next_state_value || !done || !terminated || !returns) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", fn);
    if (!rolling_gamma && !gamma_t && !lmbda_t)
        return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: rolling_gamma=0 needs a gamma or lmbda tensor", fn);
    TdLambdaArgs a{reward, next_state_value, done, terminated, gamma_t, lmbda_t, gamma, lmbda, O, T, I, returns, (mask_bcast && I > 1) ? 1 : 0};
    return dispatch_types(fn, val_dtype, coef_dtype, [&](auto ts, auto tg) {
        using TS = decltype(ts);
        using TG = decltype(tg);
        if (rolling_gamma) {
            for_rows(a.O * a.I, a.T, [&](i64 row) {
                const i64 o = row / a.I;
                seqrow::td_lambda_row<TS, TG>(a, o, row - o * a.I);
            });
        } else {  // O(T^2): every (o, i) row reruns its T tails
            for_rows(a.O * a.I, a.T * a.T / 2 + 1, [&](i64 row) {
                const i64 o = row / a.I;
                for (i64 k = 0; k < a.T; ++k) seqrow::td_lambda_fixed_elem<TS, TG>(a, o, k, row - o * a.I);
            });
        }
    });
}

int trl_td1_fwd_cpu(int val_dtype, int coef_dtype, const void *reward, const void *next_state_value, const uint8_t *done,
                    const uint8_t *terminated, int mask_bcast, const void *gamma_t, double gamma, int rolling_gamma, int64_t O, int64_t T,
                    int64_t I, void *returns) {
    const char *fn = "trl_td1_fwd_cpu";
    if (int rc = check_dims(fn, O, T, I)) return rc;
    if (!reward || !next_state_value || !done || !terminated || !returns) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", fn);
    if (!rolling_gamma && !gamma_t) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: rolling_gamma=0 needs a gamma tensor", fn);
    TdLambdaArgs a{reward, next_state_value, done, terminated, gamma_t, nullptr, gamma, 1.0, O, T, I, returns, (mask_bcast && I > 1) ? 1 : 0};
    return dispatch_types(fn, val_dtype, coef_dtype, [&](auto ts, auto tg) {
        using TS = decltype(ts);
        using TG = decltype(tg);
        if (rolling_gamma) {
            for_rows(a.O * a.I, a.T, [&](i64 row) {
                const i64 o = row / a.I;
                seqrow::td1_row<TS, TG, false>(a, o, 0, row - o * a.I);
            });
        } else {
            for_rows(a.O * a.I, a.T * a.T / 2 + 1, [&](i64 row) {
                const i64 o = row / a.I;
                for (i64 k = 0; k < a.T; ++k) seqrow::td1_row<TS, TG, true>(a, o, k, row - o * a.I);
            });
        }
    });
}

int trl_vtrace_fwd_cpu(int val_dtype, int coef_dtype, const void *log_pi, const void *log_mu, const void *state_value,
                       const void *next_state_value, const void *reward, const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                       const void *gamma_t, double gamma, double rho_thresh, double c_thresh, const void *rho_thresh_t,
                       const void *c_thresh_t, int64_t O, int64_t T, int64_t I, void *advantage, void *vs) {
    const char *fn = "trl_vtrace_fwd_cpu";
    if (int rc = check_dims(fn, O, T, I)) return rc;
    if (!log_pi || !log_mu || !state_value || !reward || !done || !terminated || !advantage || !vs)
        return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", fn);
    VtraceArgs a{log_pi, log_mu, state_value, next_state_value, reward, done, terminated, gamma_t, gamma, rho_thresh, c_thresh, O, T, I,
                 advantage, vs, (mask_bcast && I > 1) ? 1 : 0, rho_thresh_t, c_thresh_t};
    return dispatch_types(fn, val_dtype, coef_dtype, [&](auto ts, auto tg) {
        for_rows(a.O * a.I, a.T, [&](i64 row) {
            const i64 o = row / a.I;
            seqrow::vtrace_row<decltype(ts), decltype(tg)>(a, o, row - o * a.I);
        });
    });
}

// ---- backward ------------------------------------------------------------------------------------------------------------
int trl_gae_bwd_cpu(int val_dtype, int coef_dtype, const void *grad_advantage, const void *grad_value_target, const uint8_t *done,
                    const uint8_t *terminated, int mask_bcast, const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                    int64_t O, int64_t T, int64_t I, void *d_reward, void *d_state_value, void *d_next_state_value) {
    const char *fn = "trl_gae_bwd_cpu";
    if (int rc = check_dims(fn, O, T, I)) return rc;
    if (!done || !terminated || (!grad_advantage && !grad_value_target)) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", fn);
    if (!d_reward && !d_state_value && !d_next_state_value) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: no output requested", fn);
    if (lg_t && !gamma_t) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: lg_t given without gamma_t", fn);
    GaeBwdArgs a{grad_advantage, grad_value_target, done, terminated, gamma_t, lg_t, gamma, lmbda, lg, O, T, I,
                 d_reward, d_state_value, d_next_state_value, (mask_bcast && I > 1) ? 1 : 0};
    return dispatch_types(fn, val_dtype, coef_dtype, [&](auto ts, auto tg) {
        for_rows(a.O * a.I, a.T, [&](i64 row) {
            const i64 o = row / a.I;
            seqrow::gae_bwd_row<decltype(ts), decltype(tg)>(a, o, row - o * a.I);
        });
    });
}

static int td_bwd_cpu(const char *fn, int td1, int val_dtype, int coef_dtype, const void *grad_returns, const uint8_t *done,
                      const uint8_t *terminated, int mask_bcast, const void *gamma_t, const void *lmbda_t, double gamma, double lmbda,
                      int64_t O, int64_t T, int64_t I, void *d_reward, void *d_next_state_value) {
    if (int rc = check_dims(fn, O, T, I)) return rc;
    if (!done || !terminated || !grad_returns) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", fn);
    if (!d_reward && !d_next_state_value) return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: no output requested", fn);
    TdBwdArgs a{grad_returns, done, terminated, gamma_t, lmbda_t, gamma, lmbda, O, T, I, d_reward, d_next_state_value,
                (mask_bcast && I > 1) ? 1 : 0};
    return dispatch_types(fn, val_dtype, coef_dtype, [&](auto ts, auto tg) {
        using TS = decltype(ts);
        using TG = decltype(tg);
        for_rows(a.O * a.I, a.T, [&](i64 row) {
            const i64 o = row / a.I;
            if (td1) seqrow::td_bwd_row<TS, TG, true>(a, o, row - o * a.I);
            else seqrow::td_bwd_row<TS, TG, false>(a, o, row - o * a.I);
        });
    });
}

int trl_td_lambda_bwd_cpu(int val_dtype, int coef_dtype, const void *grad_returns, const uint8_t *done, const uint8_t *terminated,
                          int mask_bcast, const void *gamma_t, const void *lmbda_t, double gamma, double lmbda, int64_t O, int64_t T,
                          int64_t I, void *d_reward, void *d_next_state_value) {
    return td_bwd_cpu("trl_td_lambda_bwd_cpu", 0, val_dtype, coef_dtype, grad_returns, done, terminated, mask_bcast, gamma_t, lmbda_t, gamma,
                      lmbda, O, T, I, d_reward, d_next_state_value);
}

int trl_td1_bwd_cpu(int val_dtype, int coef_dtype, const void *grad_returns, const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                    const void *gamma_t, double gamma, int64_t O, int64_t T, int64_t I, void *d_reward, void *d_next_state_value) {
    return td_bwd_cpu("trl_td1_bwd_cpu", 1, val_dtype, coef_dtype, grad_returns, done, terminated, mask_bcast, gamma_t, nullptr, gamma, 1.0, O,
                      T, I, d_reward, d_next_state_value);
}

int trl_vtrace_bwd_cpu(int val_dtype, int coef_dtype, const void *grad_advantage, const void *grad_vs, const void *log_pi, const void *log_mu,
                       const void *state_value, const void *next_state_value, const void *reward, const void *vs, const uint8_t *done,
                       const uint8_t *terminated, int mask_bcast, const void *gamma_t, double gamma, double rho_thresh, double c_thresh,
                       const void *rho_thresh_t, const void *c_thresh_t, int64_t O, int64_t T, int64_t I, void *d_log_pi, void *d_log_mu,
                       void *d_state_value, void *d_next_state_value, void *d_reward, void *d_gamma_t) {
    const char *fn = "trl_vtrace_bwd_cpu";
    if (int rc = check_dims(fn, O, T, I)) return rc;
    if (!log_pi || !log_mu || !state_value || !next_state_value || !reward || !vs || !done || !terminated || (!grad_advantage && !grad_vs))
        return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: null tensor pointer", fn);
    if (!d_log_pi && !d_log_mu && !d_state_value && !d_next_state_value && !d_reward && !d_gamma_t)
        return trl_internal_fail(TRL_ERR_INVALID_ARGUMENT, "%s: no output requested", fn);
    VtraceBwdArgs a{grad_advantage, grad_vs, log_pi, log_mu, state_value, next_state_value, reward, vs, done, terminated, gamma_t,
                    gamma, rho_thresh, c_thresh, O, T, I, d_log_pi, d_log_mu, d_state_value, d_next_state_value, d_reward,
                    d_gamma_t, (mask_bcast && I > 1) ? 1 : 0, rho_thresh_t, c_thresh_t};
    return dispatch_types(fn, val_dtype, coef_dtype, [&](auto ts, auto tg) {
        for_rows(a.O * a.I, a.T, [&](i64 row) {
            const i64 o = row / a.I;
            seqrow::vtrace_bwd_row<decltype(ts), decltype(tg)>(a, o, row - o * a.I);
        });
    });
}

}  // extern "C"
