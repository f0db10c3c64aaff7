// eks_nll on the general (D, O) path: constant R, process noise s Q, d nll / d log s.
#include "eks_loss_kernels.hpp"

namespace eks {

size_t dense_nll_workspace_bytes(int T, int K, int D, int O, int n_cand) {
  size_t need = loss_workspace_bytes(T, K, D, n_cand);
  if (n_cand == 1 && dense_score_covers(T, K, D, O)) {
    const size_t w = dense_score_workspace_bytes(T, K, D, O);
    if (w > need) need = w;
  }
  return need;
}

int dense_nll(const eks_dims_t& d, const float* y, const double* rconst, const DenseModel& Mm,
              const double* s_cand, int n_cand, int per_keypoint, double* nll, double* dnll,
              void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim;
  if (D < 1 || D > 6 || O < 1 || O > 64) return EKS_ERR_UNSUPPORTED;
  if ((long)K * n_cand > 65535) return EKS_ERR_UNSUPPORTED;
  if (ws_bytes < dense_nll_workspace_bytes(T, K, D, O, n_cand)) return EKS_ERR_WORKSPACE;
  // One s per keypoint, value + gradient, Q positive definite (the caller's word: EKS_FLAG_Q_PD): the smoother's
  // own kernels in their SCORE form - the loss from the exact filter, its derivative from the smoothing
  // distribution (Fisher's identity; eks_dense_wave.hip) - instead of dual-number elements.
  if (dnll && n_cand == 1 && per_keypoint && (d.flags & EKS_FLAG_Q_PD) && dense_score_covers(T, K, D, O)) {
    const DenseModel Ms{Mm.m0, Mm.S0, Mm.A, Mm.C, Mm.Q, s_cand};
    return dense_score(d, y, rconst, Ms, nll, dnll, ws, ws_bytes, st);
  }
  LossGeom G{K, T, O, loss_chunk(T, K * n_cand), 0, n_cand};
  G.nc = loss_chunks(T, G.B);
  const DenseModelPtrs M{Mm.m0, Mm.S0, Mm.A, Mm.C, Mm.Q};
  const LossSpec P{nullptr, nullptr, nullptr, nullptr, s_cand, per_keypoint, ObsNoise{nullptr, rconst}};
  ProfScope ps("dense_nll", st);
  if (dnll) {
    EKS_DISPATCH_D(D, (loss_launch<DualD, DD, 1>(G, M, P, y, nll, dnll, static_cast<char*>(ws), st)))
  } else {
    EKS_DISPATCH_D(D, (loss_launch<double, DD, 1>(G, M, P, y, nll, nullptr, static_cast<char*>(ws), st)))
  }
  return hip_status(hipGetLastError());
}

}  // namespace eks

EKS_DEFINE_TOUCH(loss)
