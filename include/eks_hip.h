/* libeks_hip.so - C ABI of the MI355X (gfx950) ensemble-Kalman-smoother hot path.
 *
 * Drop-in boundary for the reference operator
 *     run_kalman_smoother(ys, m0s, S0s, As, Cs, Qs, ensemble_vars, s_frames, smooth_param,
 *                         blocks, lr, s_bounds_log, tol, safety_cap, h_fn) -> (s_finals, ms, Vs)
 * (/root/reference eks/core.py:159-302) and the stages it is built from.  Every entry point takes
 * plain DEVICE pointers + sizes + a hipStream_t, enqueues on that stream, never allocates, never
 * synchronises and returns an int status.  The caller owns all buffers; scratch comes from a
 * caller-provided workspace whose size the *_workspace_bytes functions report.
 *
 * Native data layout is FRAME-MAJOR (what the reference's drivers hold before they transpose for
 * JAX, eks/singlecam_smoother.py:166, eks/multicam_smoother.py:429-430):
 *     y, var   float32 [T][K][O]      observations / ensemble variances (R_t = diag(max(var,1e-12)),
 *                                     eks/utils.py:368-377 - never materialised as a matrix)
 *     ms       float32 [T][K][D]      smoothed means        (reference returns (K,T,D), core.py:296)
 *     Vs       float32 [T][K][D][D]   smoothed covariances  (reference returns (K,T,D,D), :297)
 * Model parameters are float64 device arrays with the reference's shapes:
 *     m0 [K][D], S0 [K][D][D], A [K][D][D], C [K][O][D], Q [K][D][D], s [K]  (eks/core.py:160-166)
 */
#ifndef EKS_HIP_H
#define EKS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* eks_stream_t; /* == hipStream_t */

/* status codes */
#define EKS_OK 0
#define EKS_ERR_NULL (-1)        /* a required pointer is NULL */
#define EKS_ERR_SHAPE (-2)       /* non-positive or inconsistent dimension */
#define EKS_ERR_UNSUPPORTED (-3) /* (D,O)/flag combination not built */
#define EKS_ERR_WORKSPACE (-4)   /* workspace missing or too small */
#define EKS_CSV_IO (-5)          /* eks_csv_read_numeric: the file cannot be opened / mapped */
#define EKS_CSV_FALLBACK 1       /* eks_csv_read_numeric: not an error - the file holds something other than numbers and
                                    missing values (quotes, text, ragged lines): let pandas read it */
#define EKS_ERR_HIP_BASE (-1000) /* -1000 - hipError_t */

/* flags */
#define EKS_FLAG_DIAG_MODEL 1u /* caller asserts A, C, Q, S0 are diagonal and D == O: the K*D
                                  coordinates are independent scalar chains (singlecam, reference
                                  eks/singlecam_smoother.py:246-284).  Off-diagonals are not read. */
#define EKS_FLAG_VS_DIAG 2u    /* Vs is written as [T][K][D] (diagonal of the covariance only,
                                  which is all the reference's drivers consume:
                                  singlecam_smoother.py:210-211, multicam_smoother.py:509-510,
                                  :540-542).  Default: full [T][K][D][D]. */
#define EKS_FLAG_UNIT_AC 4u    /* with DIAG_MODEL: caller asserts A = C = I (folds multiplies) */
#define EKS_FLAG_Q_PD 8u       /* caller asserts every Q[k] is positive definite.  eks_nll (one s per
                                  keypoint, with gradient) and eks_adam_run on the general (D, O) path may
                                  then take d nll / d log s from the smoothing distribution (Fisher's
                                  identity: exact, plain float64) instead of dual numbers through the scan.
                                  Without the flag the dual-number kernels run (any PSD Q). */

#define EKS_FLAG_ADAM_PREPARED 16u /* eks_adam_run: eks_adam_prepare has run on this (dims, y, A, workspace) and nothing
                                     has written the workspace since - the call skips its own pass over y */

typedef struct {
  int32_t n_keypoints; /* K */
  int32_t n_frames;    /* T */
  int32_t state_dim;   /* D */
  int32_t obs_dim;     /* O */
  uint32_t flags;
} eks_dims_t;

const char* eks_version(void);
const char* eks_status_string(int status);

/* ---- fixed-s smoothing: the final pass of run_kalman_smoother (eks/core.py:269-297), i.e.
 * dynamax extended_kalman_smoother with linear f,h vmapped over keypoints. ------------------ */
size_t eks_smooth_workspace_bytes(const eks_dims_t* dims);
int eks_smooth(const eks_dims_t* dims, const float* y, const float* var, const double* m0,
               const double* S0, const double* A, const double* C, const double* Q,
               const double* s, float* ms, float* Vs, void* workspace, size_t workspace_bytes,
               eks_stream_t stream);

/* ---- constant observation noise for the loss: eks/core.py:702-709
 * rconst[k][o] = max(nanmedian_t max(var[t][k][o], 1e-12), min_var)  (float64 out) ----------- */
size_t eks_const_r_workspace_bytes(const eks_dims_t* dims);
int eks_const_r(const eks_dims_t* dims, const float* var, double min_var, double* rconst,
                void* workspace, size_t workspace_bytes, eks_stream_t stream);

/* ---- filter negative log-likelihood with constant R on a set of smoothing-parameter values:
 * the loss of eks/core.py:640-650 (nll = -marginal_loglik of extended_kalman_filter, non-finite
 * -> 1e12).  s_cand is [n_cand] when per_keypoint == 0 (one grid shared by all keypoints: the
 * 64-candidate search of BASELINE.json config 3) or [K][n_cand] when per_keypoint == 1 (used with
 * n_cand == 1 by the Adam loop).  nll is [K][n_cand] float64; dnll (optional, may be NULL) is
 * d nll / d log s of the same shape (forward sensitivity replacing jax.value_and_grad, :652). -- */
size_t eks_nll_workspace_bytes(const eks_dims_t* dims, int32_t n_cand);
int eks_nll(const eks_dims_t* dims, const float* y, const double* rconst, const double* m0,
            const double* S0, const double* A, const double* C, const double* Q,
            const double* s_cand, int32_t n_cand, int32_t per_keypoint, double* nll, double* dnll,
            void* workspace, size_t workspace_bytes, eks_stream_t stream);

/* ---- order statistics for numpy.percentile's linear interpolation: center_predictions' per-keypoint
 * threshold on the worst ensemble variance (eks/utils.py:318-322, np.percentile(..., axis=0)) and the
 * v_quantile_threshold of the variance-inflation loop (eks/stats.py:109-112).  x [n_rows][n_cols] float32;
 * out[c] = {x_sorted[rank_lo], x_sorted[rank_hi]} of column c with NaNs sorted last as numpy sorts them
 * (rank_hi = rank_lo or rank_lo + 1), nan_count[c] = number of NaNs in the column (numpy returns NaN for
 * such a slice).  The interpolation itself is three float32 operations per column and stays with the
 * caller, which forms it with numpy's own expressions (eks_amd/utils.py percentile_from_order_stats). ---- */
int eks_order_stats(int32_t n_rows, int32_t n_cols, const float* x, int32_t rank_lo, int32_t rank_hi,
                    float* out, int32_t* nan_count, eks_stream_t stream);

/* ---- numpy.nanstd(x, axis=1) of a [n_rows][n_cols] float32 matrix, BIT FOR BIT: the optimiser's initial guess
 * (reference eks/core.py:104-133, compute_initial_guesses: round(nanstd(differences of the ensemble variances), 5))
 * seeds the Adam trajectory through a float32, so it has to be numpy's value, and numpy's value is a function of
 * its summation order (pairwise, eks_amd/csrc/eks_np_sum.hpp).  leaves: n_leaves pairs (start, length <= 128) of
 * that recursion's leaves left to right; ops: n_leaves - 1 triples (dst, a, b) in evaluation order over slots
 * 0 .. n_leaves - 1 = leaf sums, n_leaves .. = internal nodes, the last one the root (both tables depend on n_cols
 * only; eks_amd/hip_ops.py: np_sum_program builds them).  out[r] float32; a row without a finite value gives NaN.
 * (n_cols + 2 n_leaves) * 4 bytes must fit 64 KB of LDS: EKS_ERR_UNSUPPORTED otherwise. ---------------------- */
int eks_np_nanstd_rows(int32_t n_rows, int32_t n_cols, const float* x, const int32_t* leaves, int32_t n_leaves,
                       const int32_t* ops, int32_t n_ops, float* out, eks_stream_t stream);

/* ---- the same for the rows compute_initial_guesses actually reduces (eks/core.py:128-130), formed on the fly: row k,
 * element t * obs_dim + o = x[t + 1][k][o] - x[t][k][o] of a frame-major [n_frames][n_keypoints][obs_dim] float32
 * tensor (the first n_frames <= 2 000 frames of the ensemble variances) - one launch instead of a subtraction, a
 * transposing copy and the reduction.  leaves / ops: the tables for n_cols = (n_frames - 1) * obs_dim. ------------- */
int eks_np_nanstd_diff_rows(int32_t n_frames, int32_t n_keypoints, int32_t obs_dim, const float* x, const int32_t* leaves,
                            int32_t n_leaves, const int32_t* ops, int32_t n_ops, float* out, eks_stream_t stream);

/* ---- argmin over candidates + gather: s_out[k] = s_cand[argmin_c nll[k][c]] (first minimum,
 * like numpy.argmin).  idx_out (optional) receives the int32 indices. ---------------------- */
int eks_argmin_s(int32_t n_keypoints, int32_t n_cand, const double* nll, const double* s_cand,
                 double* s_out, int32_t* idx_out, eks_stream_t stream);

/* ---- the grid search in one call: eks_nll (value only) followed by eks_argmin_s, as ONE entry point so that the
 * scalar-chain grid kernels can take the argmin inside the assembly of the table (the block that finishes a tile of
 * keypoints last does it: no separate launch).  This is the smoothing-parameter search of BASELINE.json config 3
 * (the reference's own search, eks/core.py:562-699, is the Adam loop below).  Arguments as eks_nll (per_keypoint = 0)
 * and eks_argmin_s; nll [K][n_cand] is still written in full.  Shapes the fused kernels do not cover run the two
 * steps one after the other - the results are the same either way.  workspace: eks_nll_workspace_bytes. ------------ */
int eks_nll_argmin(const eks_dims_t* dims, const float* y, const double* rconst, const double* m0,
                   const double* S0, const double* A, const double* C, const double* Q,
                   const double* s_cand, int32_t n_cand, double* nll, double* s_out, int32_t* idx_out,
                   void* workspace, size_t workspace_bytes, eks_stream_t stream);

/* ---- one Adam iteration on u = log s for every block of keypoints, with the reference's
 * control flow (eks/core.py:652-681 singletons, :509-549 blocks): L_b = sum of member nll,
 * g_b = lr * sum of member dnll (zero where u is outside [lo, hi], as jnp.clip differentiates),
 * optax.adam(1.0) update (b1 .9, b2 .999, eps 1e-8, bias-corrected), then
 * done = isfinite(prev) && |L - prev| < tol * |log(max(prev, 1e-12))| + 1e-6; prev = L; ++iters.
 * Blocks that are done or at safety_cap are left untouched.  Blocks are given in CSR form
 * (block_offsets [n_blocks+1], block_members [K]).  state is [n_blocks][6] float64:
 * {u, mom, vel, prev_loss, iters, done}.  Writes s_keypoint[k] = exp(clip(u_block(k), lo, hi)) for
 * the next evaluation and *n_active = number of blocks still running after this step. -------- */
int eks_adam_step(int32_t n_blocks, const int32_t* block_offsets, const int32_t* block_members,
                  const double* nll, const double* dnll, double lr, double lo, double hi,
                  double tol, int32_t safety_cap, double* state, double* s_keypoint,
                  int32_t* n_active, eks_stream_t stream);

/* ---- n_iters iterations of { eks_nll (one s per keypoint, with dnll) -> eks_adam_step } enqueued
 * back to back: the body of the lax.while_loop of eks/core.py:654-681 (:520-549 for blocks) without
 * a host round trip per iteration.  Arguments as in eks_nll / eks_adam_step; s_keypoint [K] is
 * both the evaluation point and the step's output; nll, dnll [K] hold the last evaluation.
 * Iterations enqueued after a block has stopped leave it untouched, so the caller may issue
 * n_iters at a time and read *n_active in between.  workspace: eks_nll_workspace_bytes(dims, 1).
 * Scalar chains with one keypoint per block (the reference's default, blocks = []): all n_iters iterations are ONE
 * launch with a workgroup per keypoint and no exchange between workgroups.  Sessions of 1 024 frames and more (at most
 * four chains per keypoint) do not read y per iteration at all: one streaming pass leaves 256 lag sums of the inputs
 * u_t = y_t - a y_{t-1} per chain - they do not depend on s - and every iteration evaluates loss and gradient from those
 * plus the first / last 257 rows (eks_amd/csrc/eks_lag_adam.hip); a chain whose pole leaves the range the sums cover
 * (|rho| > 0.906) is evaluated exactly from a private copy of its frames instead.  eks_adam_prepare runs that pass ahead
 * of time (see EKS_FLAG_ADAM_PREPARED). */
int eks_adam_run(const eks_dims_t* dims, const float* y, const double* rconst, const double* m0,
                 const double* S0, const double* A, const double* C, const double* Q,
                 int32_t n_blocks, const int32_t* block_offsets, const int32_t* block_members,
                 double lr, double lo, double hi, double tol, int32_t safety_cap, int32_t n_iters,
                 double* state, double* s_keypoint, double* nll, double* dnll, int32_t* n_active,
                 void* workspace, size_t workspace_bytes, eks_stream_t stream);

/* ---- the pass over y of eks_adam_run's scalar-chain search, ahead of the optimiser's starting point: the reference
 * computes its initial guesses from the ensemble variances on the host (eks/core.py:233-236), and this pass needs only
 * y and A - a caller enqueues it, fetches the guesses while it runs, uploads the state and calls eks_adam_run with
 * EKS_FLAG_ADAM_PREPARED set in dims->flags.  Returns EKS_ERR_UNSUPPORTED (nothing enqueued) where eks_adam_run would
 * not use the sums (other model classes, blocks of several keypoints, short sessions): call eks_adam_run without the
 * flag then.  workspace: the one eks_adam_run will be given. */
int eks_adam_prepare(const eks_dims_t* dims, const float* y, const double* A, int32_t n_blocks, void* workspace,
                     size_t workspace_bytes, eks_stream_t stream);

/* ---- how many iterations one eks_adam_run call should ask for on this problem, device and library build: the calls
 * are what the caller's host round trips (reading *n_active) are spaced by, and what they cost differs by the form the
 * loop takes - 4 096 (i.e. the whole search in one call) where the search runs from cached lag sums, 64 for short sessions
 * (one launch, a workgroup per keypoint), 16 where every iteration is its own launch on scalar chains (an over-issued
 * iteration is a launch that returns at once), 4 on the general (D, O) path (an over-issued iteration is a full
 * evaluation).  No reference counterpart (the reference's loop is one XLA while_loop, eks/core.py:654-681). */
int32_t eks_adam_run_stride(const eks_dims_t* dims, int32_t n_blocks);

/* ---- IBL pupil smoother, eks/ibl_pupil_smoother.py:363-607.
 * Independent chains k < n_keypoints (one per session), AR(1) dynamics A_k = diag(a[k][:]),
 * process noise diag(q[k][:]), observation matrix C [K][O][D], TIME-VARYING R_t = diag(max(var,
 * 1e-12)) also in the loss (:514-518).  eks_ar1_nll: nll[k] = -marginal_loglik of the filter
 * (_nll_from_u, :540-552); with n_tan > 0, dnll[i][k] = derivative of nll[k] along the tangent
 * (da[i][k][:], dq[i][k][:]) - forward sensitivities replacing jax.value_and_grad (:570); with
 * EKS_FLAG_Q_PD (q > 0 in every coordinate) on the pupil's shape D = 3, O = 8 the same derivatives
 * come from the smoothing distribution inside the smoother's kernels.  The
 * final smoothing pass is eks_smooth with A = diag(a), Q = diag(q), s = 1 (:427-445). -------- */
size_t eks_ar1_nll_workspace_bytes(const eks_dims_t* dims, int32_t n_tan);
int eks_ar1_nll(const eks_dims_t* dims, const float* y, const float* var, const double* m0,
                const double* S0, const double* C, const double* a, const double* q,
                const double* da, const double* dq, int32_t n_tan, double* nll, double* dnll,
                void* workspace, size_t workspace_bytes, eks_stream_t stream);

/* ---- one iteration of the pupil optimiser per chain (:560-594): optax.adam(lr) (b1 .9, b2 .999,
 * eps 1e-8, bias-corrected) on u = (u_diam, u_com) with the gradient dnll [2][n] w.r.t. u, then
 * done = isfinite(prev) && |L - prev| < tol * |log(max(prev, 1e-12))| + 1e-6; prev = L; ++iters.
 * Chains that are done or at safety_cap are left untouched.  state is [n][9] float64
 * {u_d, u_c, mom_d, mom_c, vel_d, vel_c, prev_loss, iters, done}.  Always (re)writes the inputs
 * of the next eks_ar1_nll call from u: s = sigmoid(u) * (1 - 2e-3) + 1e-3 (:506-508),
 * a [n][3] = (s_d, s_c, s_c), q [n][3] = latent_var * (1 - a^2), and the two tangents
 * da, dq [2][n][3] = d(a, q)/du_d, d(a, q)/du_c.  nll == NULL: only that (initialisation).
 * *n_active = number of chains still running after this step. ------------------------------- */
int eks_pupil_adam_step(int32_t n_chains, const double* latent_var, const double* nll,
                        const double* dnll, double lr, double tol, int32_t safety_cap,
                        double* state, double* a, double* q, double* da, double* dq,
                        int32_t* n_active, eks_stream_t stream);

/* ---- n_iters iterations of { eks_ar1_nll (2 tangents) -> eks_pupil_adam_step } enqueued back to
 * back (the while_loop of :571-594).  a, q, da, dq must have been initialised by
 * eks_pupil_adam_step(nll = NULL).  nll [n], dnll [2][n] hold the last evaluation.
 * workspace: eks_ar1_nll_workspace_bytes(dims, 2). ------------------------------------------ */
int eks_pupil_adam_run(const eks_dims_t* dims, const float* y, const float* var, const double* m0,
                       const double* S0, const double* C, const double* latent_var, double lr,
                       double tol, int32_t safety_cap, int32_t n_iters, double* state, double* a,
                       double* q, double* da, double* dq, double* nll, double* dnll,
                       int32_t* n_active, void* workspace, size_t workspace_bytes,
                       eks_stream_t stream);

/* ---- extended Kalman filter / smoother with calibrated pinhole cameras (SURVEY.md section 8(f)
 * rank 3): run_kalman_smoother(h_fn = multi-camera projection), eks/core.py:188-190, :274-295, as
 * called from eks/multicam_smoother.py:369-407; the projection is make_jax_projection_fn (:814-868)
 * with its analytic Jacobian in place of jax.jacfwd.  State dim 3, obs dim 2 * n_cams (u, v per
 * camera).
 *   dims.n_keypoints = number of CHAINS K; chain k reads the observations of keypoint
 *   k % n_data_keypoints (several chains per keypoint = several values of s over the same data,
 *   which is how the optimiser takes central differences in log s).
 *   y [T][Kd][O] float32; exactly one of var [T][Kd][O] float32 (time-varying R_t = diag(max(var,
 *   1e-12)), final pass) and rconst [Kd][O] float64 (constant R, the loss of :640-650) is non-NULL.
 *   m0 [K][3], S0, A, Q [K][3][3], s [K] per chain.  cams [n_cams][32] float64: R row-major (9),
 *   t (3), fx, fy, cx, cy, skew, then the 14 OpenCV-ordered distortion coefficients k1 k2 p1 p2 k3
 *   k4 k5 k6 s1 s2 s3 s4 tx ty (radial POLYNOMIAL in r^2 as in the reference; tx, ty ignored), pad.
 *   xlin [K][T][3] float64, in/out: the linearisation points (predicted means).  On entry any
 *   finite guess (the prior mean, a triangulation, the previous call's result); on return the
 *   extended filter's predicted means.  The filter is solved as a fixed point over xlin by at
 *   most max_sweeps (<= 64) gated scan sweeps, stopping once no point moves by more than tol
 *   (relative to max(1, |x|)); info[0] = sweeps executed, info[1] = last such change (the result
 *   equals the sequential extended filter's when info[1] <= tol).
 *   ms [T][K][3], Vs [T][K][3][3] (or [T][K][3] with EKS_FLAG_VS_DIAG) float32 smoothed outputs;
 *   ms == NULL: filter only.  nll [K] = -marginal log-likelihood (may be NULL). -------------- */
size_t eks_ekf_smooth_workspace_bytes(const eks_dims_t* dims, int32_t want_smoother);
int eks_ekf_smooth(const eks_dims_t* dims, int32_t n_data_keypoints, const float* y,
                   const float* var, const double* rconst, const double* m0, const double* S0,
                   const double* A, const double* Q, const double* s, const double* cams,
                   int32_t n_cams, double* xlin, int32_t max_sweeps, double tol, float* ms,
                   float* Vs, double* nll, double* info, void* workspace, size_t workspace_bytes,
                   eks_stream_t stream);

/* ---- ensemble statistics, eks/core.py:25-101: markers float32 [M][V][T][K][3] (x,y,likelihood)
 * -> stats float32 [V][T][K][5] (x, y, var_x, var_y, likelihood).  avg_mode 0 median / 1 mean,
 * var_mode 0 confidence_weighted_var / 1 var. ------------------------------------------------ */
int eks_ensemble(int32_t n_models, int32_t n_cameras, int32_t n_frames, int32_t n_keypoints,
                 const float* markers, int32_t avg_mode, int32_t var_mode, float nan_replacement,
                 float* stats, eks_stream_t stream);

/* ---- variance inflation, one pass of the loop of
 * eks/multicam_smoother.py:695-708 for every ACTIVE keypoint: the per-frame factor-analysis
 * reconstruction residual and per-view 2x2 Mahalanobis distance of compute_mahalanobis
 * (eks/stats.py:119-151) given the loading matrix W and mean mu, then inflate_variance
 * (eks/multicam_smoother.py:724-764): v *= scalar for every (frame, view) whose distance exceeds
 * threshold, the whole frame when n_views == 2.  The factor-analysis fit between passes stays on
 * the host.  x [K][N][2C] float64 (centred predictions, stacked views c0x c0y c1x ...),
 * v [K][N][2C] float32 in/out, W [K][2C][L], mu [K][2C] float64, active [K] (NULL = all),
 * maha [K][N][C] float64 out (may be NULL), n_inflated [K] out = number of frames with a hit
 * (0 = this keypoint's loop is finished).  2 <= C <= 8, 1 <= L <= 6. ------------------------- */
int eks_maha_inflate(int32_t n_keypoints, int32_t n_frames, int32_t n_views, int32_t n_latent,
                     const double* x, float* v, const double* W, const double* mu,
                     const int32_t* active, double epsilon, double threshold, double scalar,
                     double* maha, int32_t* n_inflated, eks_stream_t stream);

/* ---- output epilogue of the linear multi-camera driver (eks/multicam_smoother.py:481-544):
 * stats [V][T][K][5] float32 (eks_ensemble's x, y, var_x, var_y, likelihood), ev [T][K][2V]
 * float32 (the variances the filter used, possibly inflated), ms [T][K][D], Vs [T][K][D][D]
 * float32 (eks_smooth), C [K][2V][D], mean [V][K][2] float64 ->
 * tables [V][T][K][9] float64: x, y = C m + mean | likelihood | x, y ensemble average |
 * x, y ensemble variance (= ev, :505-508) | x, y posterior variance = diag(C V C') + ev (:509-510);
 * latent [T][K][2D] float64 = (m, diag V) (:529-544; may be NULL).  D <= 6. ------------------ */
int eks_multicam_tables(int32_t n_views, int32_t n_frames, int32_t n_keypoints, int32_t state_dim,
                        const float* stats, const float* ev, const float* ms, const float* Vs,
                        const double* C, const double* mean, double* tables, double* latent,
                        eks_stream_t stream);

/* ---- optional per-kernel timing (used by bench.py's roofline object).  on = 1: each kernel launch
 * (stage) is bracketed by hipEvents on the caller's stream; on = 2: only the roofline kernels - the
 * smoother's replay kernels (HBM-bound) and the NLL grid kernel (VALU-bound); 0: off. eks_profile_drain waits for the
 * recorded events, writes up to max_n NUL-terminated kernel names back to back into `names` and
 * their durations in milliseconds into `ms`, clears the record and returns the count. -------- */
int eks_profile_enable(int on);
int eks_profile_drain(char* names, size_t names_bytes, float* ms, int32_t max_n);

/* ---- first-call latency.  The HIP runtime loads a translation unit's code object (1-3 MB each here) the first
 * time one of its kernels is launched, so the FIRST call of a process into each part of the library pays tens of
 * milliseconds of loading on top of its kernels - for the reference's own 2 000-frame recordings that is the whole
 * run time.  eks_warmup loads the units named in `units` now (no kernel runs, no device memory is touched; needs a
 * current device) so that a caller - e.g. on a background thread while its CSV files parse - takes the cost off its
 * first smoothing call.  ms_per_unit (optional, 9 floats, bit order) receives each unit's load time.  No reference
 * counterpart (XLA compiles on first call instead: eks/core.py:585-588 mentions "28 separate XLA compilations"). */
#define EKS_WARM_MISC 1u        /* eks_const_r, eks_ensemble, eks_order_stats, eks_argmin_s, eks_adam_step */
#define EKS_WARM_DIAG 2u        /* eks_smooth on scalar chains (EKS_FLAG_DIAG_MODEL) */
#define EKS_WARM_DIAG_NLL 4u    /* eks_nll / eks_adam_run on scalar chains */
#define EKS_WARM_DENSE 8u       /* general (D, O): scan, generic summarize / replay, eks_ekf_smooth */
#define EKS_WARM_DENSE_WAVE 16u /* general (D, O), narrow sessions (and their smoothing-distribution gradient) */
#define EKS_WARM_DENSE_WIDE 32u /* general (D, O), wide sessions */
#define EKS_WARM_LOSS 64u       /* eks_nll on the general path (dual numbers) */
#define EKS_WARM_LOSS_AR1 128u  /* eks_ar1_nll (pupil) */
#define EKS_WARM_MULTICAM 256u  /* eks_maha_inflate, eks_multicam_tables */
#define EKS_WARM_ALL 511u
int eks_warmup(uint32_t units, float* ms_per_unit);

/* ---- measurement hook.  The EKS_* tuning variables (alternative kernel
 * organisations kept for A/B runs; none is needed in use) are read ONCE, on the first call into the
 * library, never per call.  eks_knobs_reload re-reads them (a test that flips a variable between two
 * calls of one process); not to be called while another thread is inside the library.  Returns the
 * number of EKS_* variables found set.  No reference counterpart. ------------------------------ */
int eks_knobs_reload(void);

/* ---- HOST-side helpers of the boundary (no device code, no stream: they return when done) ------------------------
 * eks_csv_read_numeric: the numeric body of a prediction CSV - what `pd.read_csv(path, header=[0, 1, 2], index_col=0)`
 * (reference eks/utils.py:188; the 3 header rows are skip_lines) parses field by field - into out[n_rows][n_cols]
 * float64, EVERY column including the index column, blank lines skipped, empty fields and pandas' default NA strings
 * -> NaN, [+-]inf / infinity accepted.  The decimal -> double conversion is pandas' own default (its tokenizer's
 * precise_xstrtod: <= 17 significant digits, one multiplication / division by a power of ten), so the values are the
 * ones pandas returns, bit for bit.  out == NULL: size query (n_rows_out, n_cols_out only).  col_is_int (optional,
 * n_cols bytes): 1 where every field of the column was written as an integer (pandas gives such a column dtype int64).
 * n_threads: blocks of lines parsed concurrently (std::thread).  Returns EKS_OK, EKS_CSV_FALLBACK (see above; also for
 * integers beyond 2^53), EKS_CSV_IO, EKS_ERR_NULL, EKS_ERR_WORKSPACE (capacity < n_rows * n_cols). ------------------- */
int eks_csv_read_numeric(const char* path, int32_t skip_lines, double* out, int64_t capacity, int64_t* n_rows_out,
                         int32_t* n_cols_out, uint8_t* col_is_int, int32_t col_capacity, int32_t n_threads);

/* eks_csv_write_table: `DataFrame.to_csv(path)` of a float64 table with an int64 index, byte for byte (the result
 * tables of fit_eks_*: reference eks/singlecam_smoother.py:98-99, eks/multicam_smoother.py:151-152, :270-275): `header`
 * (header_bytes of text, newline included: the caller takes it from pandas - the empty slice's to_csv) followed by one
 * line per row, "index,v0,v1,...": every number as Python's repr (the shortest decimal string that reads back as the
 * same double; fixed notation for decimal exponents -4 .. 15, d.ddde+XX otherwise), a NaN as the empty field, inf /
 * -inf as such.  values [n_rows][n_cols] row-major.  Row blocks are formatted on n_threads threads and written in
 * order.  eks_format_repr (test hook): the text of n doubles concatenated into out, offsets [n + 1]. --------------- */
int eks_csv_write_table(const char* path, const char* header, int64_t header_bytes, const int64_t* index,
                        const double* values, int64_t n_rows, int32_t n_cols, int32_t n_threads);
int eks_format_repr(const double* values, int64_t n, char* out, int64_t capacity, int64_t* offsets);

/* eks_host_thread_speedup: how many times faster n_threads threads of this process finish n_threads units of work than
 * one thread finishes one (a ~1 ms measurement; a sandbox may expose several CPUs and still run a process's threads
 * one at a time).  The Python wrapper asks once and takes the threaded writer only where threads run side by side. -- */
double eks_host_thread_speedup(int32_t n_threads);

/* eks_host_model_flags: the EKS_FLAG_DIAG_MODEL / EKS_FLAG_UNIT_AC / EKS_FLAG_Q_PD bits that HOST copies of the parameters
 * S0, A, Q [K][D][D] and C [K][O][D] allow (what run_kalman_smoother decides before its first launch; the reference has
 * no counterpart - eks/core.py:159-177 hands whatever it is given to dynamax): DIAG_MODEL when D == O and every matrix
 * is diagonal, UNIT_AC when also A = C = I, Q_PD when every Q[k] is finite with lambda_min > min_eig_ratio x lambda_max.
 * Returns the flags (>= 0); EKS_ERR_UNSUPPORTED when Q is finite but NOT diagonal (its eigenvalues decide Q_PD: the
 * caller asks LAPACK). ---------------------------------------------------------------------------------------------- */
int eks_host_model_flags(int32_t n_keypoints, int32_t state_dim, int32_t obs_dim, const double* S0, const double* A,
                         const double* C, const double* Q, double min_eig_ratio);

/* eks_host_gather_cols: dst[r][0 .. width) = src[r][col_offset .. col_offset + width) (bytes) for n_rows rows of a
 * row-major HOST matrix with rows of src_row_bytes, on n_threads threads: a keypoint tile of the frame-major ensemble
 * variances (T, K, O) (reference eks/core.py:159-177 takes them in that layout) made contiguous for its upload. ---- */
int eks_host_gather_cols(const void* src, int64_t n_rows, int64_t src_row_bytes, int64_t col_offset_bytes,
                         int64_t width_bytes, void* dst, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif /* EKS_HIP_H */
