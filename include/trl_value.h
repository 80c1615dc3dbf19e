/*
 * trl_value.h -- C ABI of the MI355X (gfx950) value-estimation library.
 *
 * This is the drop-in boundary for TorchRL's value-estimation hot path.  The reference
 * (AsbjornLorenzen/torchrl v0.7.2) has no native code on this path: the functions below
 * replace everything *below* the Python signatures of
 *   torchrl/objectives/value/functional.py   (the ATen op soup of the loop / vec variants)
 * and are loaded the way the reference loads its only native module
 *   torchrl/_extension.py:17-31, setup.py:121-169  (`torchrl._torchrl` discovery convention).
 * INTEGRATION.md shows the reference-side ctypes stub a maintainer would add.
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types.  PyTorch (or any caller) owns every
 *    buffer: inputs are read, caller-allocated outputs are written, nothing is retained.
 *  - All pointers are DEVICE pointers (HIP).  Every tensor is the canonical contiguous
 *    (O, T, I) view of a `[*B, T, *F]` tensor with time at `time_dim`:
 *    O = prod(B), I = prod(F), element (o,t,i) at o*T*I + t*I + i.
 *  - Masks (`done`, `terminated`) are one byte per element, 0 or 1 (torch.bool storage).
 *  - `stream` is a hipStream_t (NULL = default stream).  Calls are asynchronous on that
 *    stream; the library never synchronises the host and never allocates device memory.
 *  - `val_dtype` is the dtype of reward/value tensors, `coef_dtype` the dtype the
 *    reference's coefficient tensors (`gamma * int_mask`) would have: python-float gammas
 *    are rounded to the torch default dtype even for float64 values (functional.py:172-174).
 *    Scalar coefficients are passed as doubles that already hold the rounded value.
 *    Arithmetic runs in promote(val_dtype, coef_dtype), op for op in the reference's order
 *    (no FMA contraction).
 *  - Return value: TRL_OK or an error code; `trl_last_error()` gives the message of the
 *    last failure on the calling thread.  The host wrapper maps them to the reference's
 *    Python exceptions.  One code is a WARNING, not a failure: TRL_WARN_FUSED_DEGRADED means the call did
 *    everything it was asked to (outputs are valid) and reports, once, that the single-launch route has been
 *    switched off on this GPU (see trl_fused_degraded below).
 *  - The `*_cpu` entry points (ABI 7) take HOST pointers, run on the calling thread plus a few worker threads and
 *    return when the outputs are written: the same estimators for tensors that live in host memory.
 */
#ifndef TRL_VALUE_H
#define TRL_VALUE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRL_ABI_VERSION 8

enum { TRL_F32 = 0, TRL_F64 = 1 };

enum {
    TRL_OK = 0,
    TRL_ERR_INVALID_ARGUMENT = 1, /* null pointer, non-positive size, bad dtype/algo */
    TRL_ERR_UNSUPPORTED = 2,      /* combination not implemented by the chosen algorithm */
    TRL_ERR_WORKSPACE = 3,        /* workspace missing or too small */
    TRL_ERR_LAUNCH = 4,           /* HIP reported a launch error */
    TRL_WARN_FUSED_DEGRADED = 100 /* NOT a failure: the call completed; reported once by the first call that notices that
                                     an earlier single-launch call (GAE + normalise, standardize) gave up (trl_fused_degraded) */
};

/* Kernel family.  AUTO picks by shape. */
enum {
    TRL_ALGO_AUTO = 0,
    TRL_ALGO_SEQ = 1, /* one lane per (o,i) row, sequential over t: bit-identical to the reference loop */
    TRL_ALGO_SCAN = 2, /* wavefront scan of affine maps over time: tolerance-level parity, bit-exact at done boundaries.
                          I == 1: any T.  I > 1: whole outer rows of at most 256*V elements (V = 4 fp32 / 2 fp64) through an
                          LDS transpose; GAE, TD(lambda), TD(1) with 64 < T <= 256 and I != 16: up to 1024*V elements */
    TRL_ALGO_FUSED = 3 /* trl_gae_normalize_fwd only: single persistent launch, advantages kept in
                          registers across an in-launch all-gather of the fp64 moments */,
    TRL_ALGO_LOCKSTEP = 4 /* 2 <= I <= 31 feature elements per step (groups of <= 8), any T: a lane owns one time step, the I
                          recurrences are scanned in lock step (no LDS transpose); same parity as SCAN */
};

int trl_abi_version(void);
const char *trl_last_error(void);
/* Name of the kernel family the last successful call on this thread dispatched to. */
const char *trl_last_kernel(void);

/* Bytes of device workspace the calls below need for a problem of this size.  The workspace must be
 * 256-byte aligned and ZERO-FILLED ONCE before its first use (it carries the epoch of the fused
 * kernel's in-launch exchange); afterwards only this library writes it.  Calls that share a
 * workspace must not overlap in time (one workspace per stream).
 * Besides the statistics, the workspace carries the chunk records of the decoupled look-back scan:
 * when there are too few rows to fill the chip (O < 7/8 #CUs, I == 1, T of 8 chunks or more) and a
 * workspace of this size is passed, trl_gae_fwd (stats == NULL), trl_td_lambda_fwd and
 * trl_vtrace_fwd scan the chunks of a row in parallel; without a workspace they scan each row's
 * chunks in turn inside one block (same results). */
size_t trl_workspace_bytes(int64_t O, int64_t T, int64_t I);
/* 1 if a single-launch (FUSED) call on this workspace ever gave up waiting for its other blocks (its
 * grid of #CUs persistent blocks must be co-resident: a GPU shared with other long-running kernels
 * can starve it; the wait is bounded at 1 s, the advantage is then written UN-NORMALISED -- finite, never
 * NaN -- and the process degrades by itself, see trl_fused_degraded), 0 if not, -1 on error.  Synchronises the device.
 * workspace == NULL (ABI 7): the static scratch of trl_gae_normalize_fwd_noalloc on the current device; its error word is
 * read AND cleared (nobody else can reach it), and a set word degrades the device as the host flag does.  Callers of the
 * no-alloc entry point have no host flag (allocating it is an allocation): they learn of a give-up from this call, or from
 * `stats` (NaN sum / sumsq on a give-up). */
int trl_workspace_error(const void *workspace);
/* Failure semantics of the single-launch routes (ABI 7; ABI 8: trl_standardize / trl_standardize_grouped too).  A persistent
 * launch that gives up raises a word in pinned host memory (allocated by the first such launch of trl_gae_normalize_fwd /
 * _xchg_fwd / trl_standardize / trl_standardize_grouped on a device).  The NEXT entry-point call on
 * that GPU -- any estimator, trl_normalize -- notices it without synchronising and marks the device DEGRADED: from then on
 * TRL_ALGO_AUTO runs GAE + normalise as two launches (scan + scale pass; no co-residency needed, ~2x the time at
 * [65536 x 32]) and the standardize calls as moments + scale pass.  That call does its own work and returns TRL_WARN_FUSED_DEGRADED once; later calls return TRL_OK.
 * Only an explicit TRL_ALGO_FUSED request fails (TRL_ERR_LAUNCH, on the call that notices the give-up).
 * trl_fused_degraded(): 1 if the current device is degraded, else 0.  trl_fused_reset(): re-arms the single launch on the
 * current device (the caller knows the GPU is its own again). */
int trl_fused_degraded(void);
int trl_fused_reset(void);

/*
 * Generalized advantage estimate.
 * Replaces: generalized_advantage_estimate / vec_generalized_advantage_estimate
 *           (torchrl/objectives/value/functional.py:121-182, 272-370).
 *   delta_t = r_t + (gamma_t * (1-terminated_t)) * v'_t - v_t
 *   A_t     = delta_t + A_{t+1} * (lg_t * (1-done_t)),  A_T = 0
 *   value_target = A + v
 * gamma_t / lg_t: optional per-element tensors (coef_dtype) of gamma and lmbda*gamma;
 *   when gamma_t != NULL and lg_t == NULL, lg_t is formed as (coef)lmbda * gamma_t in-kernel
 *   (what `lmbda * gamma` gives for a python-float lmbda).  When both are NULL the scalars
 *   `gamma` and `lg` apply.
 * mask_bcast: 0 = done / terminated are full (O, T, I) arrays; 1 = they are (O, T) arrays shared by
 *   the I feature elements of a step (what a stride-0 `expand` of a per-env mask over the agent
 *   dimension means, sota-implementations/multiagent/utils/utils.py:34-43) and are read in place.
 *   Ignored when I == 1.  Same meaning in every entry point below.
 * next_state_value == NULL (ABI 4): SHIFTED value layout.  `state_value` then points to a contiguous
 *   (O, T + 1, I) buffer -- the value network evaluated once on the T steps plus the last next-observation
 *   (what `shifted=True` produces, torchrl/objectives/value/advantages.py:440-490) -- and v_t / v'_t are its
 *   elements (o, t, i) / (o, t + 1, i), read in place: the value bytes are fetched once (fp32 GAE:
 *   22 -> 18 B per transition).  Outputs keep the (O, T, I) layout.  Needs I == 1 (any kernel family but
 *   LOCKSTEP) or TRL_ALGO_SEQ (any I); TRL_ERR_UNSUPPORTED otherwise.  Same meaning in
 *   trl_gae_normalize_fwd, trl_gae_normalize_xchg_fwd and trl_vtrace_fwd.
 * stats (optional, device double[3]): receives {N, sum(A), sum(A^2)} accumulated in fp64 in a
 *   fixed order (deterministic); needs `workspace` of trl_workspace_bytes().  It feeds
 *   trl_normalize() (after an all-reduce when the batch is sharded over GPUs).
 */
int trl_gae_fwd(void *stream, int val_dtype, int coef_dtype,
                const void *reward, const void *state_value, const void *next_state_value,
                const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                const void *gamma_t, const void *lg_t,
                double gamma, double lmbda, double lg,
                int64_t O, int64_t T, int64_t I,
                void *advantage, void *value_target,
                double *stats, void *workspace, size_t workspace_bytes, int algo);

/*
 * GAE followed by advantage normalisation over ALL elements -- the north-star step.
 * Replaces: GAE.forward with average_gae=True (torchrl/objectives/value/advantages.py:1436-1463):
 *   adv, value_target = gae(...);  adv = (adv - adv.mean()) / adv.std().clamp_min(eps)
 * Arguments as trl_gae_fwd; `advantage` receives the NORMALISED advantage.  `stats` (optional,
 * device double[3]) receives {N, sum(A), sum(A^2)} of the un-normalised advantage.
 * AUTO uses the single-launch FUSED kernel when the shape is eligible (I == 1, short rows, the
 * advantages fit in registers: up to 8 * #CUs * 1024 * V transitions), otherwise two launches
 * (the scan writing per-block moments, a scale pass that sums them itself).  Single-GPU only: a batch sharded over GPUs uses trl_gae_fwd
 * (stats) -> all-reduce -> trl_normalize.
 */
int trl_gae_normalize_fwd(void *stream, int val_dtype, int coef_dtype,
                          const void *reward, const void *state_value, const void *next_state_value,
                          const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                          const void *gamma_t, const void *lg_t,
                          double gamma, double lmbda, double lg,
                          int64_t O, int64_t T, int64_t I,
                          void *advantage, void *value_target, double eps,
                          double *stats, void *workspace, size_t workspace_bytes, int algo);

/*
 * trl_gae_normalize_fwd for callers that cannot allocate.  NO counterpart in the reference: it exists for callers that
 * cannot call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid allocation) and
 * therefore have no workspace to pass.  Same arguments and results as trl_gae_normalize_fwd minus the workspace: the
 * scratch is a file-scope device array inside the library, trl_static_scratch_bytes() (2 MiB) per GPU, held until the
 * process ends.  Calls no hipMalloc / hipFree or any allocating routine.  Problems whose trl_workspace_bytes(O, T, I)
 * exceeds the array are rejected with TRL_ERR_WORKSPACE.  Every call on a GPU shares that GPU's array (whatever the
 * stream): calls that overlap in time on one GPU race -- order them on one stream or with events.
 * The FIRST call on a device (and the first call that reaches a given kernel) queries the runtime -- hipGetSymbolAddress,
 * the occupancy of the persistent kernel -- and must therefore happen OUTSIDE graph capture; every later call only launches
 * (tests/test_zz_no_alloc.py: free device memory unchanged across calls on every route; captured and replayed as a graph).
 */
size_t trl_static_scratch_bytes(void);
int trl_gae_normalize_fwd_noalloc(void *stream, int val_dtype, int coef_dtype,
                                  const void *reward, const void *state_value, const void *next_state_value,
                                  const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                                  const void *gamma_t, const void *lg_t,
                                  double gamma, double lmbda, double lg,
                                  int64_t O, int64_t T, int64_t I,
                                  void *advantage, void *value_target, double eps,
                                  double *stats, int algo);

/*
 * TD(lambda) return.
 * Replaces: td_lambda_return_estimate / vec_td_lambda_return_estimate
 *           (functional.py:783-889, 1030-1183).
 *   v'_t <- v'_t * (1-terminated_t);  g = v'_{T-1}
 *   for t = T-1..0:  g <- g*(1-done_t) + v'_t*done_t;  g = R_t = r_t + gamma_t*((1-lmbda_t)*v'_t + lmbda_t*g)
 * gamma_t / lmbda_t: optional per-element tensors (coef_dtype), else scalars.
 * rolling_gamma == 0: the O(T^2) variant (functional.py:877-887): R_k is the tail run with
 *   gamma_k, lmbda_k held fixed; requires gamma_t or lmbda_t.
 * workspace: optional (NULL, 0), see trl_workspace_bytes.
 */
int trl_td_lambda_fwd(void *stream, int val_dtype, int coef_dtype,
                      const void *reward, const void *next_state_value,
                      const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                      const void *gamma_t, const void *lmbda_t,
                      double gamma, double lmbda, int rolling_gamma,
                      int64_t O, int64_t T, int64_t I,
                      void *returns, void *workspace, size_t workspace_bytes, int algo);

/*
 * TD(1) return.
 * Replaces: td1_return_estimate (functional.py:464-562); vec_td1_return_estimate is vec TD(lambda) at lmbda = 1
 *           (functional.py:691-700) and goes through trl_td_lambda_fwd.
 *   dnt_t = done_t & ~terminated_t;  g = v'_{T-1}  (the raw next value, not masked)
 *   for t = T-1..0:  g = R_t = r_t + (gamma_t*(1-terminated_t)) * ((1-dnt_t)*g + dnt_t*v'_t)
 * gamma_t: optional per-element tensor (coef_dtype), else the scalar `gamma` (already rounded to the value dtype:
 *   the reference builds torch.full_like(next_state_value, gamma)).
 * rolling_gamma == 0 (needs gamma_t): the O(T^2) variant (functional.py:550-561): R_k reruns the tail T-1..k with
 *   gamma_k held fixed and g starting from 0 -- unlike TD(lambda)'s, it does NOT bootstrap from v'_{T-1}.
 * workspace: optional (NULL, 0), see trl_workspace_bytes.
 */
int trl_td1_fwd(void *stream, int val_dtype, int coef_dtype,
                const void *reward, const void *next_state_value,
                const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                const void *gamma_t, double gamma, int rolling_gamma,
                int64_t O, int64_t T, int64_t I,
                void *returns, void *workspace, size_t workspace_bytes, int algo);

/*
 * V-trace advantage and targets.
 * Replaces: vtrace_advantage_estimate (functional.py:1270-1350).
 *   rho = exp(log_pi - log_mu); rb = min(rho, rho_thresh); cb = min(rho, c_thresh)
 *   x_t = rb_t*(r_t + gamma_t*(1-term_t)*v'_t - v_t) + (gamma_t*(1-done_t)*cb_t)*x_{t+1},  x_T = 0
 *   vs = x + v;  vs+_t = vs_{t+1} (t < T-1), v'_{T-1} (t = T-1; the shift ignores done)
 *   advantage = rb*(r + gamma*(1-term)*vs+ - v)
 * Outputs have dtype promote(val_dtype, coef_dtype) (the reference allocates nothing in val_dtype).
 * rho_thresh_t / c_thresh_t (ABI 5): optional per-element thresholds (val_dtype, (O, T, I)) for callers that pass a
 *   broadcastable threshold TENSOR (functional.py:1311-1330: rho.clamp_max(tensor)); they replace the scalars and run
 *   on the SEQ family (AUTO picks it; any other `algo` is TRL_ERR_UNSUPPORTED).  NULL: the scalars apply.
 * workspace: optional (NULL, 0), see trl_workspace_bytes.
 */
int trl_vtrace_fwd(void *stream, int val_dtype, int coef_dtype,
                   const void *log_pi, const void *log_mu,
                   const void *state_value, const void *next_state_value, const void *reward,
                   const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                   const void *gamma_t, double gamma, double rho_thresh, double c_thresh,
                   const void *rho_thresh_t, const void *c_thresh_t,
                   int64_t O, int64_t T, int64_t I,
                   void *advantage, void *vs, void *workspace, size_t workspace_bytes, int algo);

/*
 * Backward passes (adjoints) of the four estimators: what autograd through the reference's Python loops computes when
 * the estimators run with differentiable=True (torchrl/objectives/utils.py:74-83 builds every loss's estimator that
 * way; test/test_cost.py:15521-15584 asserts the gradient reaching `reward`).  The adjoint of the reverse scan
 * x_t = b_t + a_t x_{t+1} is the forward scan G_t = gbar_t + a_{t-1} G_{t-1}; the kernels read the incoming gradients
 * and the masks / coefficients once and write every requested gradient once (no flipped or shifted copies).
 * Conventions: incoming gradients may be NULL (= zeros) but not all of them; outputs may be NULL (not written) but not
 * all of them; masks, gamma_t / lg_t / lmbda_t, the scalars, mask_bcast, O / T / I and `algo` (AUTO, SCAN: I == 1, SEQ:
 * any shape) as in the forward call.  Gradients w.r.t. value tensors have val_dtype.  workspace: optional look-back
 * region as in the forward calls.
 *
 * trl_gae_bwd: gbar = grad_advantage + grad_value_target, a_t = lg_t (1 - done_t);
 *   d_reward = G,  d_state_value = grad_value_target - G,  d_next_state_value = G * gamma_t * (1 - terminated_t).
 *   (d_reward is the adjoint itself: the host forms gradients w.r.t. gamma / lmbda tensors from it.)
 */
int trl_gae_bwd(void *stream, int val_dtype, int coef_dtype,
                const void *grad_advantage, const void *grad_value_target,
                const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                int64_t O, int64_t T, int64_t I,
                void *d_reward, void *d_state_value, void *d_next_state_value,
                void *workspace, size_t workspace_bytes, int algo);
/* trl_td_lambda_bwd (rolling gamma only): a_t = gamma_t lmbda_t (1 - done_t);  d_reward = G,
 *   d_next_state_value_t = (G_t gamma_t ((1 - lmbda_t) + lmbda_t done_t) + [t = T-1] a_t G_t) (1 - terminated_t). */
int trl_td_lambda_bwd(void *stream, int val_dtype, int coef_dtype, const void *grad_returns,
                      const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                      const void *gamma_t, const void *lmbda_t, double gamma, double lmbda,
                      int64_t O, int64_t T, int64_t I,
                      void *d_reward, void *d_next_state_value,
                      void *workspace, size_t workspace_bytes, int algo);
/* trl_td1_bwd (rolling gamma only): gnt = gamma_t (1 - terminated_t), dnt = done & ~terminated, a_t = gnt_t (1 - dnt_t);
 *   d_reward = G,  d_next_state_value_t = G_t gnt_t dnt_t + [t = T-1] a_t G_t. */
int trl_td1_bwd(void *stream, int val_dtype, int coef_dtype, const void *grad_returns,
                const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                const void *gamma_t, double gamma,
                int64_t O, int64_t T, int64_t I,
                void *d_reward, void *d_next_state_value,
                void *workspace, size_t workspace_bytes, int algo);
/* trl_td_fixed_bwd: backward of the rolling_gamma=False forms (td_lambda_return_estimate, functional.py:877-887; td1_return_estimate,
 * :550-561; `td1` selects which).  R_k reruns the tail T-1..k with the coefficients of step k held fixed, so the gradients are column
 * sums of the O(T^2) matrix of tail products (d_reward, d_next_state_value: one wave per (o, i) row, deterministic wave reductions)
 * and forward-mode derivatives along each tail (d_gamma_t, d_lmbda_t: per element, coef_dtype-promoted compute dtype; they need
 * `reward` and `next_state_value`, which may be NULL otherwise).  Outputs that are NULL are not computed.  O(T^2) work per row, as
 * the forward call; no workspace, one family. */
int trl_td_fixed_bwd(void *stream, int val_dtype, int coef_dtype, int td1, const void *grad_returns,
                     const void *reward, const void *next_state_value,
                     const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                     const void *gamma_t, const void *lmbda_t, double gamma, double lmbda,
                     int64_t O, int64_t T, int64_t I,
                     void *d_reward, void *d_next_state_value, void *d_gamma_t, void *d_lmbda_t);
/* trl_vtrace_bwd: grad_advantage / grad_vs and `vs` (the forward call's second output, saved) have dtype
 *   promote(val_dtype, coef_dtype); gradients w.r.t. log_pi, log_mu (= -d_log_pi), state_value, next_state_value, reward
 *   have val_dtype; d_gamma_t (optional, per element) has the promoted dtype.  The clipped branches of min(rho, thresh)
 *   pass no gradient (torch.clamp_max). */
int trl_vtrace_bwd(void *stream, int val_dtype, int coef_dtype,
                   const void *grad_advantage, const void *grad_vs,
                   const void *log_pi, const void *log_mu,
                   const void *state_value, const void *next_state_value, const void *reward, const void *vs,
                   const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                   const void *gamma_t, double gamma, double rho_thresh, double c_thresh,
                   const void *rho_thresh_t, const void *c_thresh_t,
                   int64_t O, int64_t T, int64_t I,
                   void *d_log_pi, void *d_log_mu, void *d_state_value, void *d_next_state_value, void *d_reward,
                   void *d_gamma_t, void *workspace, size_t workspace_bytes, int algo);

/*
 * Advantage normalisation from precomputed statistics.
 * Replaces: `(adv - adv.mean()) / adv.std().clamp_min(eps)`
 *           (torchrl/objectives/value/advantages.py:1459-1463 GAE average_gae, eps 1e-4;
 *            :1828-1832 VTrace average_adv, eps 1e-5; torchrl/_utils.py:901-912 _standardize, eps 1e-6).
 * stats: device double[3] = {N, sum, sum of squares} over ALL elements (all ranks).
 * mean = sum/N; std = sqrt((sumsq - N*mean^2)/(N-1)) (unbiased); both rounded to `dtype`,
 * std clamped from below by eps, then out = (x - mean) / std in `dtype`.  out may alias x.
 */
int trl_normalize(void *stream, int dtype, const void *x, int64_t n,
                  const double *stats, double eps, void *out);

/*
 * {N, sum, sumsq} of an arbitrary tensor (fp64 accumulation, fixed order) into stats[3];
 * used for normalising V-trace advantages / TD(lambda) rewards and by tests.
 * Needs `workspace` of trl_workspace_bytes(n, 1, 1).
 */
int trl_moments(void *stream, int dtype, const void *x, int64_t n,
                double *stats, void *workspace, size_t workspace_bytes);

/*
 * trl_moments + trl_normalize in one call (out-of-place or in place): `(x - x.mean()) / x.std().clamp_min(eps)`,
 * _standardize(x) without exclude_dims (torchrl/_utils.py:901-912) -- what PPO applies to every minibatch of
 * advantages (torchrl/objectives/ppo.py:688, 975, 1282).  n <= 65536 (a minibatch): ONE single-block launch,
 * no workspace needed; larger: needs `workspace` of trl_workspace_bytes(n, 1, 1) and (ABI 8) runs as ONE persistent
 * launch too -- #CUs co-resident blocks keep their share of x in registers across an in-launch gather of the block
 * moments (the workspace header of the single-launch GAE + normalise: same rules, 256-byte aligned, zero-filled once;
 * same failure semantics, see trl_fused_degraded) -- up to 56 fp32 / 28 fp64 values per thread (14.7 M / 7.3 M elements
 * on 256 CUs); beyond that, on a degraded device or with a workspace that is only 8-byte aligned: moments -> scale pass
 * (two launches).  stats (optional, device double[3]) receives {N, sum, sumsq}.
 */
int trl_standardize(void *stream, int dtype, const void *x, int64_t n, double eps, void *out, double *stats,
                    void *workspace, size_t workspace_bytes);

/*
 * Per-group statistics and normalisation.
 * Replaces: _standardize(input, exclude_dims) (torchrl/_utils.py:875-936; PPO's advantage
 *           normalisation, torchrl/objectives/ppo.py:688, 975, 1282): mean / unbiased std over the
 *           dims NOT excluded, one pair per index of the excluded dims ("keep the agent dimension
 *           independent"), then (x - mean) / std.clamp_min(eps).
 * x is seen as a contiguous (outer, G, inner) array: element f = (o*G + g)*inner + j belongs to
 * group g (the host merges adjacent dims, or permutes, into this pattern); G <= 65535.
 * trl_moments_grouped: stats[3*g .. 3*g+2] = {outer*inner, sum, sumsq} of group g (fp64, fixed order);
 *   needs `workspace` of trl_grouped_workspace_bytes().  A batch sharded over GPUs all-reduces
 *   the G*3 doubles before normalising.
 * trl_normalize_grouped: out = (x - mean_g) / max(std_g, eps), mean / std rounded to `dtype`
 *   as in trl_normalize.  out may alias x.
 * trl_standardize_grouped: both in one call (single GPU); stats: device double[G*3], filled on the way.  ABI 8: ONE
 *   persistent launch when G <= 16 and G*inner <= 1024 (the multi-agent case: a handful of agents, inner == 1) -- as
 *   trl_standardize above, wave g of every block gathering group g -- else moments -> per-group reduce -> scale pass.
 * ABI 8: the workspace of trl_grouped_workspace_bytes() starts with the {epoch, error, slots} header of the one-launch
 *   route: it must be 256-byte aligned (8-byte alignment still works, on the multi-launch route) and ZERO-FILLED ONCE
 *   before its first use; afterwards only this library writes it; calls that share it must not overlap in time.
 */
size_t trl_grouped_workspace_bytes(int64_t outer, int64_t G, int64_t inner);
int trl_moments_grouped(void *stream, int dtype, const void *x, int64_t outer, int64_t G, int64_t inner,
                        double *stats, void *workspace, size_t workspace_bytes);
int trl_normalize_grouped(void *stream, int dtype, const void *x, int64_t outer, int64_t G, int64_t inner,
                          const double *stats, double eps, void *out);
int trl_standardize_grouped(void *stream, int dtype, const void *x, int64_t outer, int64_t G, int64_t inner,
                            double eps, void *out, double *stats, void *workspace, size_t workspace_bytes);

/*
 * Cross-GPU sum of the moments -- the one collective of the path when the env dimension is sharded
 * over GPUs (one process per GPU).  Replaces what `adv.mean()` / `adv.std()` over all elements
 * (advantages.py:1459-1463) become on a sharded batch: an all-reduce(sum) of 3 doubles, i.e.
 * torch.distributed.all_reduce(stats) over RCCL -- which stays the portable route and the
 * checker.  This one is the low-latency route inside a node: every rank stores its 24 bytes
 * directly into a mailbox in each peer's HBM over xGMI and polls its own mailbox (one 64-thread
 * block, a few microseconds; RCCL's smallest all-reduce costs several times the GAE kernel).
 *
 * Set-up, once per process (the host side exchanges the handles with any out-of-band channel,
 * e.g. torch.distributed.all_gather_object):
 *   trl_xchg_local_init(&mailbox, handle)      allocate + zero this rank's mailbox (fine-grained
 *                                              device memory on the current device), fill the
 *                                              trl_xchg_handle_bytes()-byte IPC handle
 *   trl_xchg_open(mailbox, handles, devices, rank, world, timeout_s, &comm)
 *                                              handles: world * handle_bytes, in rank order;
 *                                              devices: each rank's device ordinal (or NULL);
 *                                              world <= 64, all ranks on one node
 * Per step, on every rank, in the same order:
 *   trl_xchg_allreduce_stats(stream, comm, stats)   stats: device double[3] {N, sum, sumsq},
 *                                              summed over ranks IN PLACE, bit-identical on all
 *                                              ranks (fixed rank order), N exact.  Asynchronous;
 *                                              graph-capturable (the call counter lives on the device).
 * A rank that waits longer than timeout_s (default 20) for a peer sets the comm's error word and
 * trl_xchg_allreduce_stats leaves NaN statistics: trl_xchg_error(comm) (synchronises the device) returns 1.
 * In trl_gae_normalize_xchg_fwd's single launch a rank whose own gather gave up sends NaN moments, so every
 * rank sees the failure; each then writes the UN-NORMALISED advantage, sets its error words and raises the
 * host flag (the next estimator call degrades the device and returns TRL_WARN_FUSED_DEGRADED), as
 * trl_gae_normalize_fwd does on one GPU.
 * trl_gae_normalize_xchg_fwd = trl_gae_normalize_fwd on a batch sharded over the comm's ranks: the
 * moments are summed over all ranks INSIDE the launch (FUSED family: after the in-launch gather of
 * the block moments, block 0 stores this GPU's totals into the peers' mailboxes and every block
 * polls the local mailbox), so the whole sharded step stays one launch per GPU.  Shapes the FUSED
 * family does not take run trl_gae_fwd -> trl_xchg_allreduce_stats -> trl_normalize.  `stats`
 * receives the GLOBAL {N, sum, sumsq}.  Every rank must make the same sequence of calls on the comm.
 */
size_t trl_xchg_handle_bytes(void);
int trl_xchg_local_init(void **mailbox, void *ipc_handle_out);
int trl_xchg_open(void *mailbox, const void *ipc_handles, const int *peer_devices, int rank, int world,
                  double timeout_s, void **comm_out);
int trl_xchg_allreduce_stats(void *stream, void *comm, double *stats);
int trl_xchg_set_timeout(void *comm, double timeout_s);
int trl_xchg_error(void *comm);
int trl_xchg_destroy(void *comm, void *mailbox);
int trl_gae_normalize_xchg_fwd(void *stream, int val_dtype, int coef_dtype,
                               const void *reward, const void *state_value, const void *next_state_value,
                               const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                               const void *gamma_t, const void *lg_t,
                               double gamma, double lmbda, double lg,
                               int64_t O, int64_t T, int64_t I,
                               void *advantage, void *value_target, double eps,
                               double *stats, void *workspace, size_t workspace_bytes, void *comm, int algo);

/*
 * Host entry points (ABI 7): the same estimators for tensors in HOST memory -- every pointer below is a host pointer,
 * `stats` included; the call returns when the outputs are written.  The reference runs its estimators on whatever device
 * the tensors live on (functional.py:163-170 allocates on state_value.device; the CartPole-shaped BASELINE config is a CPU
 * call), so the drop-in takes CPU tensors too.  Arguments as in the device entry points of the same name, minus stream,
 * workspace and `algo`: there is one family, the row-sequential walk in the reference's operation order -- the very row
 * functions of the SEQ kernels (csrc/seq_rows.h) -- so results are bit-identical to TRL_ALGO_SEQ on the GPU (V-trace: up to
 * libm's exp) and to the reference's loop.  Rows are split over trl_cpu_threads() threads (TRL_CPU_THREADS, else
 * OMP_NUM_THREADS, else min(cores, 16); jobs under 65536 elements run on the calling thread); the moments are summed in a
 * fixed order that does not depend on the thread count.  No HIP call is made: they work without a GPU.
 * trl_td_fixed_bwd has no host counterpart (rolling_gamma=False is differentiable on the GPU only).
 */
int trl_cpu_threads(void);
int trl_set_cpu_threads(int n); /* 0: decide from the environment again */
int trl_gae_fwd_cpu(int val_dtype, int coef_dtype,
                    const void *reward, const void *state_value, const void *next_state_value,
                    const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                    const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                    int64_t O, int64_t T, int64_t I, void *advantage, void *value_target, double *stats);
int trl_gae_normalize_fwd_cpu(int val_dtype, int coef_dtype,
                              const void *reward, const void *state_value, const void *next_state_value,
                              const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                              const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                              int64_t O, int64_t T, int64_t I, void *advantage, void *value_target, double eps, double *stats);
int trl_td_lambda_fwd_cpu(int val_dtype, int coef_dtype, const void *reward, const void *next_state_value,
                          const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                          const void *gamma_t, const void *lmbda_t, double gamma, double lmbda, int rolling_gamma,
                          int64_t O, int64_t T, int64_t I, void *returns);
int trl_td1_fwd_cpu(int val_dtype, int coef_dtype, const void *reward, const void *next_state_value,
                    const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                    const void *gamma_t, double gamma, int rolling_gamma,
                    int64_t O, int64_t T, int64_t I, void *returns);
int trl_vtrace_fwd_cpu(int val_dtype, int coef_dtype, const void *log_pi, const void *log_mu,
                       const void *state_value, const void *next_state_value, const void *reward,
                       const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                       const void *gamma_t, double gamma, double rho_thresh, double c_thresh,
                       const void *rho_thresh_t, const void *c_thresh_t,
                       int64_t O, int64_t T, int64_t I, void *advantage, void *vs);
int trl_gae_bwd_cpu(int val_dtype, int coef_dtype, const void *grad_advantage, const void *grad_value_target,
                    const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                    const void *gamma_t, const void *lg_t, double gamma, double lmbda, double lg,
                    int64_t O, int64_t T, int64_t I, void *d_reward, void *d_state_value, void *d_next_state_value);
int trl_td_lambda_bwd_cpu(int val_dtype, int coef_dtype, const void *grad_returns,
                          const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                          const void *gamma_t, const void *lmbda_t, double gamma, double lmbda,
                          int64_t O, int64_t T, int64_t I, void *d_reward, void *d_next_state_value);
int trl_td1_bwd_cpu(int val_dtype, int coef_dtype, const void *grad_returns,
                    const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                    const void *gamma_t, double gamma,
                    int64_t O, int64_t T, int64_t I, void *d_reward, void *d_next_state_value);
int trl_vtrace_bwd_cpu(int val_dtype, int coef_dtype, const void *grad_advantage, const void *grad_vs,
                       const void *log_pi, const void *log_mu,
                       const void *state_value, const void *next_state_value, const void *reward, const void *vs,
                       const uint8_t *done, const uint8_t *terminated, int mask_bcast,
                       const void *gamma_t, double gamma, double rho_thresh, double c_thresh,
                       const void *rho_thresh_t, const void *c_thresh_t,
                       int64_t O, int64_t T, int64_t I,
                       void *d_log_pi, void *d_log_mu, void *d_state_value, void *d_next_state_value, void *d_reward, void *d_gamma_t);
/* {N, sum, sumsq} of x (host double[3]) / (x - mean) / max(std, eps) from such statistics, as trl_moments / trl_normalize */
int trl_moments_cpu(int dtype, const void *x, int64_t n, double *stats);
int trl_normalize_cpu(int dtype, const void *x, int64_t n, const double *stats, double eps, void *out);

#ifdef __cplusplus
}
#endif
#endif /* TRL_VALUE_H */
