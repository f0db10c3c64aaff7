// extern "C" entry points of libeks_hip.so (declared in include/eks_hip.h): argument checks and
// dispatch between the scalar-chain path (EKS_FLAG_DIAG_MODEL) and the general small-matrix path.
#include <hip/hip_runtime.h>

#include "eks_adam.hpp"
#include "eks_diag_lane.hpp"
#include <chrono>

#include "eks_internal.hpp"

using namespace eks;

static int check_dims(const eks_dims_t* d) {
  if (!d) return EKS_ERR_NULL;
  if (d->n_keypoints <= 0 || d->n_frames <= 0 || d->state_dim <= 0 || d->obs_dim <= 0)
    return EKS_ERR_SHAPE;
  if ((d->flags & EKS_FLAG_DIAG_MODEL) && d->state_dim != d->obs_dim) return EKS_ERR_SHAPE;
  if ((d->flags & EKS_FLAG_UNIT_AC) && !(d->flags & EKS_FLAG_DIAG_MODEL)) return EKS_ERR_UNSUPPORTED;
  if ((long)d->n_keypoints * d->state_dim > (1L << 24)) return EKS_ERR_SHAPE;
  return EKS_OK;
}

extern "C" {

const char* eks_version(void) { return "eks_hip 0.1 (gfx950)"; }

const char* eks_status_string(int status) {
  switch (status) {
    case EKS_OK: return "ok";
    case EKS_ERR_NULL: return "null pointer";
    case EKS_ERR_SHAPE: return "bad shape";
    case EKS_ERR_UNSUPPORTED: return "unsupported (D, O) / flag combination";
    case EKS_ERR_WORKSPACE: return "workspace missing or too small";
    case EKS_CSV_IO: return "file cannot be opened or mapped";
    case EKS_CSV_FALLBACK: return "not a purely numeric table: read it with pandas";
    default: return status <= EKS_ERR_HIP_BASE ? hipGetErrorString((hipError_t)(EKS_ERR_HIP_BASE - status))
                                               : "unknown status";
  }
}

size_t eks_smooth_workspace_bytes(const eks_dims_t* d) {
  if (check_dims(d) != EKS_OK) return 0;
  if (d->flags & EKS_FLAG_DIAG_MODEL)
    return diag_smooth_workspace_bytes(d->n_frames, d->n_keypoints * d->state_dim);
  return dense_smooth_workspace_bytes(d->n_frames, d->n_keypoints, d->state_dim, d->obs_dim);
}

int eks_smooth(const eks_dims_t* d, const float* y, const float* var, const double* m0,
               const double* S0, const double* A, const double* C, const double* Q,
               const double* s, float* ms, float* Vs, void* workspace, size_t workspace_bytes,
               eks_stream_t stream) {
  const int rc = check_dims(d);
  if (rc != EKS_OK) return rc;
  if (!y || !var || !m0 || !S0 || !A || !C || !Q || !s || !ms || !Vs) return EKS_ERR_NULL;
  if (!workspace) return EKS_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->flags & EKS_FLAG_DIAG_MODEL) {
    const DiagModel M{m0, S0, A, C, Q, s, d->state_dim};
    return diag_smooth(*d, y, var, M, ms, Vs, workspace, workspace_bytes, st);
  }
  const DenseModel M{m0, S0, A, C, Q, s};
  return dense_smooth(*d, y, var, M, ms, Vs, workspace, workspace_bytes, st);
}

size_t eks_const_r_workspace_bytes(const eks_dims_t* d) {
  if (check_dims(d) != EKS_OK) return 0;
  return const_r_workspace_bytes(d->n_frames, d->n_keypoints * d->obs_dim);
}

int eks_const_r(const eks_dims_t* d, const float* var, double min_var, double* rconst,
                void* workspace, size_t workspace_bytes, eks_stream_t stream) {
  const int rc = check_dims(d);
  if (rc != EKS_OK) return rc;
  if (!var || !rconst) return EKS_ERR_NULL;
  if (!workspace) return EKS_ERR_WORKSPACE;
  return const_r(d->n_frames, d->n_keypoints * d->obs_dim, var, min_var, rconst, workspace,
                 workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

size_t eks_nll_workspace_bytes(const eks_dims_t* d, int32_t n_cand) {
  if (check_dims(d) != EKS_OK || n_cand <= 0) return 0;
  if (d->flags & EKS_FLAG_DIAG_MODEL)
    return diag_nll_workspace_bytes(d->n_frames, d->n_keypoints * d->state_dim, n_cand);
  return dense_nll_workspace_bytes(d->n_frames, d->n_keypoints, d->state_dim, d->obs_dim, n_cand);
}

int eks_nll(const eks_dims_t* d, const float* y, const double* rconst, const double* m0,
            const double* S0, const double* A, const double* C, const double* Q,
            const double* s_cand, int32_t n_cand, int32_t per_keypoint, double* nll, double* dnll,
            void* workspace, size_t workspace_bytes, eks_stream_t stream) {
  const int rc = check_dims(d);
  if (rc != EKS_OK) return rc;
  if (n_cand <= 0) return EKS_ERR_SHAPE;
  if (!y || !rconst || !m0 || !S0 || !A || !C || !Q || !s_cand || !nll) return EKS_ERR_NULL;
  if (!workspace) return EKS_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->flags & EKS_FLAG_DIAG_MODEL) {
    const DiagModel M{m0, S0, A, C, Q, nullptr, d->state_dim};
    return diag_nll(*d, y, rconst, M, s_cand, n_cand, per_keypoint, nll, dnll, workspace,
                    workspace_bytes, st);
  }
  const DenseModel M{m0, S0, A, C, Q, nullptr};
  return dense_nll(*d, y, rconst, M, s_cand, n_cand, per_keypoint, nll, dnll, workspace,
                   workspace_bytes, st);
}

int eks_nll_argmin(const eks_dims_t* d, const float* y, const double* rconst, const double* m0,
                   const double* S0, const double* A, const double* C, const double* Q,
                   const double* s_cand, int32_t n_cand, double* nll, double* s_out, int32_t* idx_out,
                   void* workspace, size_t workspace_bytes, eks_stream_t stream) {
  const int rc = check_dims(d);
  if (rc != EKS_OK) return rc;
  if (n_cand <= 0) return EKS_ERR_SHAPE;
  if (!y || !rconst || !m0 || !S0 || !A || !C || !Q || !s_cand || !nll || !s_out) return EKS_ERR_NULL;
  if (!workspace) return EKS_ERR_WORKSPACE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (d->flags & EKS_FLAG_DIAG_MODEL) {
    const DiagModel M{m0, S0, A, C, Q, nullptr, d->state_dim};
    return diag_nll(*d, y, rconst, M, s_cand, n_cand, 0, nll, nullptr, workspace, workspace_bytes, st, nullptr, s_out,
                    idx_out);
  }
  const DenseModel M{m0, S0, A, C, Q, nullptr};
  const int rc2 = dense_nll(*d, y, rconst, M, s_cand, n_cand, 0, nll, nullptr, workspace, workspace_bytes, st);
  if (rc2 != EKS_OK) return rc2;
  return argmin_s(d->n_keypoints, n_cand, nll, s_cand, s_out, idx_out, st);
}

int eks_order_stats(int32_t n_rows, int32_t n_cols, const float* x, int32_t rank_lo, int32_t rank_hi,
                    float* out, int32_t* nan_count, eks_stream_t stream) {
  if (n_rows <= 0 || n_cols <= 0 || rank_lo < 0 || rank_hi < rank_lo || rank_hi >= n_rows || rank_hi > rank_lo + 1)
    return EKS_ERR_SHAPE;
  if (!x || !out || !nan_count) return EKS_ERR_NULL;
  return order_stats(n_rows, n_cols, x, rank_lo, rank_hi, out, nan_count, reinterpret_cast<hipStream_t>(stream));
}

int eks_np_nanstd_rows(int32_t n_rows, int32_t n_cols, const float* x, const int32_t* leaves, int32_t n_leaves,
                       const int32_t* ops, int32_t n_ops, float* out, eks_stream_t stream) {
  if (n_rows <= 0 || n_cols <= 0 || n_leaves <= 0 || n_ops != n_leaves - 1) return EKS_ERR_SHAPE;
  if (!x || !leaves || !out || (n_ops > 0 && !ops)) return EKS_ERR_NULL;
  return np_nanstd_rows(n_rows, n_cols, x, leaves, n_leaves, ops, n_ops, out, reinterpret_cast<hipStream_t>(stream));
}

int eks_np_nanstd_diff_rows(int32_t n_frames, int32_t n_keypoints, int32_t obs_dim, const float* x, const int32_t* leaves,
                            int32_t n_leaves, const int32_t* ops, int32_t n_ops, float* out, eks_stream_t stream) {
  if (n_frames < 2 || n_keypoints <= 0 || obs_dim <= 0 || n_leaves <= 0 || n_ops != n_leaves - 1) return EKS_ERR_SHAPE;
  if (!x || !leaves || !out || (n_ops > 0 && !ops)) return EKS_ERR_NULL;
  return np_nanstd_diff_rows(n_frames, n_keypoints, obs_dim, x, leaves, n_leaves, ops, n_ops, out,
                             reinterpret_cast<hipStream_t>(stream));
}

int eks_argmin_s(int32_t n_keypoints, int32_t n_cand, const double* nll, const double* s_cand,
                 double* s_out, int32_t* idx_out, eks_stream_t stream) {
  if (n_keypoints <= 0 || n_cand <= 0) return EKS_ERR_SHAPE;
  if (!nll || !s_cand || !s_out) return EKS_ERR_NULL;
  return argmin_s(n_keypoints, n_cand, nll, s_cand, s_out, idx_out,
                  reinterpret_cast<hipStream_t>(stream));
}

int eks_adam_step(int32_t n_blocks, const int32_t* block_offsets, const int32_t* block_members,
                  const double* nll, const double* dnll, double lr, double lo, double hi,
                  double tol, int32_t safety_cap, double* state, double* s_keypoint,
                  int32_t* n_active, eks_stream_t stream) {
  if (n_blocks <= 0) return EKS_ERR_SHAPE;
  if (!block_offsets || !block_members || !nll || !dnll || !state || !s_keypoint || !n_active)
    return EKS_ERR_NULL;
  return adam_step(n_blocks, block_offsets, block_members, nll, dnll, lr, lo, hi, tol, safety_cap,
                   state, s_keypoint, n_active, reinterpret_cast<hipStream_t>(stream));
}

size_t eks_ar1_nll_workspace_bytes(const eks_dims_t* d, int32_t n_tan) {
  if (check_dims(d) != EKS_OK || n_tan < 0) return 0;
  return ar1_nll_workspace_bytes(d->n_frames, d->n_keypoints, d->state_dim, n_tan);
}

int eks_ar1_nll(const eks_dims_t* d, const float* y, const float* var, const double* m0,
                const double* S0, const double* C, const double* a, const double* q,
                const double* da, const double* dq, int32_t n_tan, double* nll, double* dnll,
                void* workspace, size_t workspace_bytes, eks_stream_t stream) {
  const int rc = check_dims(d);
  if (rc != EKS_OK) return rc;
  if (n_tan < 0) return EKS_ERR_SHAPE;
  if (!y || !var || !m0 || !S0 || !C || !a || !q || !nll) return EKS_ERR_NULL;
  if (n_tan > 0 && (!da || !dq || !dnll)) return EKS_ERR_NULL;
  if (!workspace) return EKS_ERR_WORKSPACE;
  return ar1_nll(*d, y, var, m0, S0, C, a, q, da, dq, n_tan, nll, dnll, workspace, workspace_bytes,
                 reinterpret_cast<hipStream_t>(stream));
}

int eks_pupil_adam_step(int32_t n_chains, const double* latent_var, const double* nll,
                        const double* dnll, double lr, double tol, int32_t safety_cap,
                        double* state, double* a, double* q, double* da, double* dq,
                        int32_t* n_active, eks_stream_t stream) {
  if (n_chains <= 0) return EKS_ERR_SHAPE;
  if (!latent_var || !state || !a || !q || !da || !dq || !n_active) return EKS_ERR_NULL;
  if (nll && !dnll) return EKS_ERR_NULL;
  return pupil_adam_step(n_chains, latent_var, nll, dnll, lr, tol, safety_cap, state, a, q, da, dq,
                         n_active, reinterpret_cast<hipStream_t>(stream));
}

int eks_adam_run(const eks_dims_t* d, const float* y, const double* rconst, const double* m0,
                 const double* S0, const double* A, const double* C, const double* Q,
                 int32_t n_blocks, const int32_t* block_offsets, const int32_t* block_members,
                 double lr, double lo, double hi, double tol, int32_t safety_cap, int32_t n_iters,
                 double* state, double* s_keypoint, double* nll, double* dnll, int32_t* n_active,
                 void* workspace, size_t workspace_bytes, eks_stream_t stream) {
  if (n_iters < 0) return EKS_ERR_SHAPE;
  if (!dnll) return EKS_ERR_NULL;
  {
    const int rc0 = check_dims(d);
    if (rc0 != EKS_OK) return rc0;
  }
  if (n_blocks <= 0) return EKS_ERR_SHAPE;
  if (!y || !rconst || !m0 || !S0 || !A || !C || !Q || !block_offsets || !block_members || !state ||
      !s_keypoint || !nll || !n_active)
    return EKS_ERR_NULL;
  if (!workspace) return EKS_ERR_WORKSPACE;
  if ((d->flags & EKS_FLAG_DIAG_MODEL) && n_iters > 0) {
    // Scalar chains: the loss kernels read the optimiser state and skip the keypoints whose block has
    // stopped (a wave none of whose 64 chains is still running returns at once; results of the others
    // are bit for bit those of the ungated loop), and when every block is one keypoint - the reference's
    // default, blocks = [] (eks/core.py:223-224) - the step is applied by the loss assembly itself:
    // two launches per iteration instead of three.
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int K = d->n_keypoints, N = K * d->state_dim;
    const size_t need = diag_nll_workspace_bytes(d->n_frames, N, 1);
    if (workspace_bytes < need) return EKS_ERR_WORKSPACE;
    char* tail = static_cast<char*>(workspace) + need - adam_extra_bytes(N);
    int32_t* kp_block = reinterpret_cast<int32_t*>(tail);
    int32_t* counter_b = reinterpret_cast<int32_t*>(tail + adam_extra_bytes(N) - 256);
    int rc = adam_prepare(n_blocks, K, block_offsets, block_members, kp_block, n_active, counter_b, st);
    if (rc != EKS_OK) return rc;
    {   // tile tickets of the single-launch loss kernel: zero once, every evaluation leaves them zero
      const hipError_t e = hipMemsetAsync(nll_ws_tickets(workspace, d->n_frames, N, 1), 0,
                                          (size_t)((N + 63) / 64) * sizeof(int32_t), st);
      if (e != hipSuccess) return EKS_ERR_HIP_BASE - (int)e;
    }
    const bool in_kernel = n_blocks == K && diag_nll_grad_tree(d->n_frames, K, d->state_dim);
    const DiagModel M{m0, S0, A, C, Q, nullptr, d->state_dim};
    if (diag_lag_adam_ok(d->n_frames, K, d->state_dim, n_blocks)) {
      // one keypoint per optimiser block, a session long enough for a head and 256 lags: one streaming pass for the lag
      // sums, then all n_iters iterations in ONE launch, a workgroup per keypoint, no pass over y per iteration
      // (eks_lag_adam.hip, round 6); n_active counts the keypoints still running
      const AdamFuse F{block_offsets, block_members, kp_block, lr, lo, hi, tol, safety_cap, 1, state, s_keypoint,
                       n_active, counter_b};
      rc = diag_lag_adam(*d, y, rconst, M, n_iters, nll, dnll, F, workspace, workspace_bytes, st);
      if (rc != EKS_ERR_UNSUPPORTED) return rc;
    }
    if (diag_nll_adam_persist_ok(d->n_frames, K, d->state_dim, n_blocks)) {
      // short sessions, one keypoint per optimiser block: all n_iters iterations in ONE launch, a workgroup per
      // keypoint (eks_diag_nll.hip: diag_nll_adam_persist_kernel); n_active counts the keypoints still running
      const AdamFuse F{block_offsets, block_members, kp_block, lr, lo, hi, tol, safety_cap, 1, state, s_keypoint,
                       n_active, counter_b};
      return diag_nll_adam_persist(*d, y, rconst, M, n_iters, nll, dnll, F, n_active, st);
    }
    for (int it = 0; it < n_iters; ++it) {
      // the LAST iteration of the call counts into n_active, the one before into the spare counter, ...
      const bool last_parity = ((n_iters - 1 - it) & 1) == 0;
      AdamFuse F{block_offsets, block_members, kp_block, lr, lo, hi, tol, safety_cap, in_kernel ? 1 : 0,
                 state, s_keypoint, last_parity ? n_active : counter_b, last_parity ? counter_b : n_active};
      rc = diag_nll(*d, y, rconst, M, s_keypoint, 1, 1, nll, dnll, workspace, workspace_bytes, st, &F);
      if (rc != EKS_OK) return rc;
      if (!in_kernel) {
        rc = adam_step(n_blocks, block_offsets, block_members, nll, dnll, lr, lo, hi, tol, safety_cap, state,
                       s_keypoint, n_active, st);
        if (rc != EKS_OK) return rc;
      }
    }
    return EKS_OK;
  }
  for (int it = 0; it < n_iters; ++it) {
    int rc = eks_nll(d, y, rconst, m0, S0, A, C, Q, s_keypoint, 1, 1, nll, dnll, workspace,
                     workspace_bytes, stream);
    if (rc != EKS_OK) return rc;
    rc = eks_adam_step(n_blocks, block_offsets, block_members, nll, dnll, lr, lo, hi, tol, safety_cap,
                       state, s_keypoint, n_active, stream);
    if (rc != EKS_OK) return rc;
  }
  return EKS_OK;
}

int eks_adam_prepare(const eks_dims_t* d, const float* y, const double* A, int32_t n_blocks, void* workspace,
                     size_t workspace_bytes, eks_stream_t stream) {
  const int rc = check_dims(d);
  if (rc != EKS_OK) return rc;
  if (n_blocks <= 0) return EKS_ERR_SHAPE;
  if (!y || !A) return EKS_ERR_NULL;
  if (!workspace) return EKS_ERR_WORKSPACE;
  if (!(d->flags & EKS_FLAG_DIAG_MODEL) || !diag_lag_adam_ok(d->n_frames, d->n_keypoints, d->state_dim, n_blocks))
    return EKS_ERR_UNSUPPORTED;
  if (workspace_bytes < diag_nll_workspace_bytes(d->n_frames, d->n_keypoints * d->state_dim, 1)) return EKS_ERR_WORKSPACE;
  return diag_lag_sums(*d, y, A, nullptr, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

int32_t eks_adam_run_stride(const eks_dims_t* d, int32_t n_blocks) {
  if (check_dims(d) != EKS_OK || n_blocks <= 0) return 4;
  if (!(d->flags & EKS_FLAG_DIAG_MODEL)) return 4;
  if (diag_lag_adam_ok(d->n_frames, d->n_keypoints, d->state_dim, n_blocks)) return 4096;
  if (diag_nll_adam_persist_ok(d->n_frames, d->n_keypoints, d->state_dim, n_blocks)) return 64;
  return 16;
}

int eks_pupil_adam_run(const eks_dims_t* d, const float* y, const float* var, const double* m0,
                       const double* S0, const double* C, const double* latent_var, double lr,
                       double tol, int32_t safety_cap, int32_t n_iters, double* state, double* a,
                       double* q, double* da, double* dq, double* nll, double* dnll,
                       int32_t* n_active, void* workspace, size_t workspace_bytes,
                       eks_stream_t stream) {
  if (n_iters < 0) return EKS_ERR_SHAPE;
  if (d && d->state_dim != 3) return EKS_ERR_SHAPE;   // (diameter, com_x, com_y)
  if (!nll || !dnll) return EKS_ERR_NULL;
  if (check_dims(d) != EKS_OK) return check_dims(d);
  if (!y || !var || !m0 || !S0 || !C || !latent_var || !state || !a || !q || !da || !dq || !n_active) return EKS_ERR_NULL;
  if (!workspace) return EKS_ERR_WORKSPACE;
  for (int it = 0; it < n_iters; ++it) {
    // loss + tangents + step in three launches where the smoothing-distribution form covers the shape (round 4)
    int rc = dense_wave_ar1_score_step(*d, y, var, m0, S0, C, latent_var, lr, tol, safety_cap, state, a, q, da, dq, nll,
                                       dnll, n_active, workspace, workspace_bytes,
                                       reinterpret_cast<hipStream_t>(stream));
    if (rc == EKS_OK) continue;
    if (rc != EKS_ERR_UNSUPPORTED) return rc;
    rc = eks_ar1_nll(d, y, var, m0, S0, C, a, q, da, dq, 2, nll, dnll, workspace, workspace_bytes,
                     stream);
    if (rc != EKS_OK) return rc;
    rc = eks_pupil_adam_step(d->n_keypoints, latent_var, nll, dnll, lr, tol, safety_cap, state, a, q,
                             da, dq, n_active, stream);
    if (rc != EKS_OK) return rc;
  }
  return EKS_OK;
}

size_t eks_ekf_smooth_workspace_bytes(const eks_dims_t* d, int32_t want_smoother) {
  if (check_dims(d) != EKS_OK) return 0;
  return ekf_smooth_workspace_bytes(d->n_frames, d->n_keypoints, want_smoother);
}

int eks_ekf_smooth(const eks_dims_t* d, int32_t n_data_keypoints, const float* y, const float* var,
                   const double* rconst, const double* m0, const double* S0, const double* A,
                   const double* Q, const double* s, const double* cams, int32_t n_cams,
                   double* xlin, int32_t max_sweeps, double tol, float* ms, float* Vs, double* nll,
                   double* info, void* workspace, size_t workspace_bytes, eks_stream_t stream) {
  const int rc = check_dims(d);
  if (rc != EKS_OK) return rc;
  if (!y || !m0 || !S0 || !A || !Q || !s || !cams || !xlin) return EKS_ERR_NULL;
  if (!var && !rconst) return EKS_ERR_NULL;
  if (ms && !Vs) return EKS_ERR_NULL;
  if (!workspace) return EKS_ERR_WORKSPACE;
  const DenseModel M{m0, S0, A, nullptr, Q, s};
  return ekf_smooth(*d, n_data_keypoints, y, var, rconst, M, cams, n_cams, xlin, max_sweeps, tol, ms,
                    Vs, nll, info, workspace, workspace_bytes, reinterpret_cast<hipStream_t>(stream));
}

int eks_ensemble(int32_t n_models, int32_t n_cameras, int32_t n_frames, int32_t n_keypoints,
                 const float* markers, int32_t avg_mode, int32_t var_mode, float nan_replacement,
                 float* stats, eks_stream_t stream) {
  if (n_models <= 0 || n_cameras <= 0 || n_frames <= 0 || n_keypoints <= 0) return EKS_ERR_SHAPE;
  if (avg_mode < 0 || avg_mode > 1 || var_mode < 0 || var_mode > 1) return EKS_ERR_UNSUPPORTED;
  if (!markers || !stats) return EKS_ERR_NULL;
  return ensemble_stats(n_models, n_cameras, n_frames, n_keypoints, markers, avg_mode, var_mode,
                        nan_replacement, stats, reinterpret_cast<hipStream_t>(stream));
}

int eks_maha_inflate(int32_t n_keypoints, int32_t n_frames, int32_t n_views, int32_t n_latent,
                     const double* x, float* v, const double* W, const double* mu,
                     const int32_t* active, double epsilon, double threshold, double scalar,
                     double* maha, int32_t* n_inflated, eks_stream_t stream) {
  if (n_keypoints <= 0 || n_frames <= 0 || n_views <= 0 || n_latent <= 0) return EKS_ERR_SHAPE;
  if (!x || !v || !W || !mu || !n_inflated) return EKS_ERR_NULL;
  return maha_inflate(n_keypoints, n_frames, n_views, n_latent, x, v, W, mu, active, epsilon, threshold,
                      scalar, maha, n_inflated, reinterpret_cast<hipStream_t>(stream));
}

int eks_multicam_tables(int32_t n_views, int32_t n_frames, int32_t n_keypoints, int32_t state_dim,
                        const float* stats, const float* ev, const float* ms, const float* Vs,
                        const double* C, const double* mean, double* tables, double* latent,
                        eks_stream_t stream) {
  if (n_views <= 0 || n_frames <= 0 || n_keypoints <= 0 || state_dim <= 0) return EKS_ERR_SHAPE;
  if (!stats || !ev || !ms || !Vs || !C || !mean || !tables) return EKS_ERR_NULL;
  return multicam_tables(n_views, n_frames, n_keypoints, state_dim, stats, ev, ms, Vs, C, mean, tables, latent,
                         reinterpret_cast<hipStream_t>(stream));
}

int eks_warmup(uint32_t units, float* ms_per_unit) {
  // order = the bits of EKS_WARM_* (include/eks_hip.h)
  static void (*const touch[])() = {eks::touch_misc,       eks::touch_diag,       eks::touch_diag_nll,
                                    eks::touch_dense,      eks::touch_dense_wave, eks::touch_dense_wide,
                                    eks::touch_loss,       eks::touch_loss_ar1,   eks::touch_multicam};
  for (unsigned i = 0; i < sizeof(touch) / sizeof(touch[0]); ++i) {
    if (ms_per_unit) ms_per_unit[i] = 0.f;
    if (!(units & (1u << i))) continue;
    const auto t0 = std::chrono::steady_clock::now();
    touch[i]();
    if (ms_per_unit)
      ms_per_unit[i] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  if (units & EKS_WARM_DIAG_NLL) eks::touch_lag_adam();        // (the search's own unit rides on the NLL bit)
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? EKS_OK : EKS_ERR_HIP_BASE - (int)e;
}

}  // extern "C"
