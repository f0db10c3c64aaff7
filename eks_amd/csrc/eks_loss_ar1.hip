// eks_ar1_nll: the pupil loss (AR(1) dynamics with explicit tangents, time-varying R_t).
#include "eks_loss_kernels.hpp"

namespace eks {

size_t ar1_nll_workspace_bytes(int T, int K, int D, int n_tan) {
  size_t need = loss_workspace_bytes(T, K, D, n_tan > 0 ? n_tan : 1);
  if (n_tan > 0 && dense_wave_ar1_covers(T, K, D, 8)) {
    const size_t w = dense_wave_workspace_bytes(T, K, D);
    if (w > need) need = w;
  }
  return need;
}

int ar1_nll(const eks_dims_t& d, const float* y, const float* var, const double* m0,
            const double* S0, const double* C, const double* a, const double* q, const double* da,
            const double* dq, int n_tan, double* nll, double* dnll, void* ws, size_t ws_bytes,
            hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim;
  if (D < 1 || D > 6 || O < 1 || O > 64) return EKS_ERR_UNSUPPORTED;
  if (ws_bytes < ar1_nll_workspace_bytes(T, K, D, n_tan)) return EKS_ERR_WORKSPACE;
  // Process noise positive in every coordinate (the caller's word: EKS_FLAG_Q_PD) on the pupil's shape: loss from
  // the exact filter inside the smoother's wave kernels, the tangents' derivatives from the smoothing distribution
  // (eks_dense_wave.hip, MODE 2) instead of dual-number elements.
  if (n_tan > 0 && (d.flags & EKS_FLAG_Q_PD) && dense_wave_ar1_covers(T, K, D, O))
    return dense_wave_ar1_score(d, y, var, m0, S0, C, a, q, da, dq, n_tan, nll, dnll, ws, ws_bytes, st);
  const int ns = n_tan > 0 ? n_tan : 1;
  if ((long)K * ns > 65535) return EKS_ERR_UNSUPPORTED;
  LossGeom G{K, T, O, loss_chunk(T, K * ns), 0, ns};
  G.nc = loss_chunks(T, G.B);
  const DenseModelPtrs M{m0, S0, nullptr, C, nullptr};
  ProfScope ps("ar1_nll", st);
  if (n_tan > 0) {
    const LossSpec P{a, q, da, dq, nullptr, 0, ObsNoise{var, nullptr}};
    EKS_DISPATCH_D(D, (loss_launch<DualD, DD, 0>(G, M, P, y, nll, dnll, static_cast<char*>(ws), st)))
  } else {
    const LossSpec P{a, q, nullptr, nullptr, nullptr, 0, ObsNoise{var, nullptr}};
    EKS_DISPATCH_D(D, (loss_launch<double, DD, 0>(G, M, P, y, nll, nullptr, static_cast<char*>(ws), st)))
  }
  return hip_status(hipGetLastError());
}

}  // namespace eks

EKS_DEFINE_TOUCH(loss_ar1)
