// gfx950 kernels for the general (D, O) smoother and loss: multicam linear path,
// D = n_latent (3..6), O = 2 * n_cameras (reference eks/multicam_smoother.py:409-443).  Same
// three-phase chunked scan as the scalar-chain path, with float64 small matrices in registers:
//   D1 dense_summarize : lane = (keypoint, chunk [, candidate]) -> element (A, b, C, eta, J [, ell])
//   D2 dense_scan      : lane = keypoint, forward / backward over the chunk elements
//   D3 dense_replay    : lane = (keypoint, chunk): exact filter, fuse, RTS; filtered beliefs go
//                        through a per-lane scratch record stream (these problems are tiny:
//                        BASELINE config 4 is 4 keypoints x 50k frames, latency- not HBM-bound)
//   D4 dense_nll_assemble : lane = (keypoint, candidate): marginal log-likelihood (and d/dlog s)
#include <hip/hip_runtime.h>

#include "eks_dense_lane.hpp"
#include "eks_internal.hpp"

namespace eks {

struct DenseGeom {
  int K, T, O, B, nc, n_cand, per_keypoint;
};

static int dense_chunk(int T) {
  int b = 16;
  while (b < 512 && (long)b * b < T) b <<= 1;
  return b;
}

template <int D>
__global__ __launch_bounds__(64) void dense_summarize_kernel(DenseGeom G, DenseModelPtrs M,
                                                            const double* __restrict__ s,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ var,
                                                            double* __restrict__ elems) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G.K * G.nc) return;
  const int k = idx % G.K, j = idx / G.K;
  Mat<double, D> F, sQ;
  bool fid;
  load_dynamics<double, D>(M, k, s[k], F, sQ, fid);
  const int t0 = j * G.B, len = min(G.B, G.T - t0);
  const DElem<double, D> e =
      dense_summarize_chunk<double, D, false>(y, var, nullptr, G.K, G.O, k, t0, len, M, F, sQ, fid);
  store_delem<double, D>(elems + (size_t)idx * delem_doubles<D>(), e);
}

template <int D>
__global__ __launch_bounds__(64) void dense_scan_kernel(DenseGeom G, DenseModelPtrs M,
                                                       const double* __restrict__ elems,
                                                       double* __restrict__ prior,
                                                       double* __restrict__ suffix) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * G.K) return;
  constexpr int REC = D + D * D;
  const int k = idx % G.K;
  if (idx < G.K) {
    Vec<double, D> m;
    Mat<double, D> P;
    load_prior<D>(M, k, m, P);
    for (int j = 0; j < G.nc; ++j) {
      double* r = prior + ((size_t)j * G.K + k) * REC;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        r[a] = m.a[a];
#pragma unroll
        for (int b = 0; b < D; ++b) r[D + a * D + b] = P.a[a][b];
      }
      const DElem<double, D> e = load_delem<double, D>(elems + ((size_t)j * G.K + k) * delem_doubles<D>());
      delem_apply(e, m, P);
    }
  } else {
    Vec<double, D> eta = vec_zero<double, D>();
    Mat<double, D> J = mat_zero<double, D>();
    for (int j = G.nc - 1; j >= 0; --j) {
      double* r = suffix + ((size_t)j * G.K + k) * REC;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        r[a] = eta.a[a];
#pragma unroll
        for (int b = 0; b < D; ++b) r[D + a * D + b] = J.a[a][b];
      }
      const DElem<double, D> e = load_delem<double, D>(elems + ((size_t)j * G.K + k) * delem_doubles<D>());
      delem_back(e, eta, J);
    }
  }
}

template <int D>
__global__ __launch_bounds__(64) void dense_replay_kernel(DenseGeom G, DenseModelPtrs M,
                                                         const double* __restrict__ s,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ var,
                                                         const double* __restrict__ prior,
                                                         const double* __restrict__ suffix,
                                                         double* __restrict__ filt,
                                                         float* __restrict__ ms,
                                                         float* __restrict__ Vs, int vs_diag) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G.K * G.nc) return;
  constexpr int REC = D + D * D;
  const int k = idx % G.K, j = idx / G.K;
  Mat<double, D> F, sQ;
  bool fid;
  load_dynamics<double, D>(M, k, s[k], F, sQ, fid);
  Vec<double, D> m, eta;
  Mat<double, D> P, J;
  const double* rp = prior + (size_t)idx * REC;
  const double* rs = suffix + (size_t)idx * REC;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    m.a[a] = rp[a];
    eta.a[a] = rs[a];
#pragma unroll
    for (int b = 0; b < D; ++b) {
      P.a[a][b] = rp[D + a * D + b];
      J.a[a][b] = rs[D + a * D + b];
    }
  }
  const int t0 = j * G.B, len = min(G.B, G.T - t0);
  dense_replay_chunk<D>(y, var, G.K, G.O, k, t0, len, M, F, sQ, fid, m, P, eta, J,
                        filt + ((size_t)k * G.T + t0) * REC, ms, Vs, vs_diag != 0);
}

template <typename S, int D>
__global__ __launch_bounds__(64) void dense_nll_summarize_kernel(DenseGeom G, DenseModelPtrs M,
                                                                const double* __restrict__ s_cand,
                                                                const float* __restrict__ y,
                                                                const double* __restrict__ rconst,
                                                                double* __restrict__ elems) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G.K * G.nc * G.n_cand) return;
  const int k = idx % G.K, rest = idx / G.K, c = rest % G.n_cand, j = rest / G.n_cand;
  const double sv = G.per_keypoint ? s_cand[(size_t)k * G.n_cand + c] : s_cand[c];
  Mat<S, D> F, sQ;
  bool fid;
  load_dynamics<S, D>(M, k, make_real(S(), sv, sv), F, sQ, fid);
  const int t0 = j * G.B, len = min(G.B, G.T - t0);
  const DElem<S, D> e =
      dense_summarize_chunk<S, D, true>(y, nullptr, rconst, G.K, G.O, k, t0, len, M, F, sQ, fid);
  constexpr int NREC = delem_doubles<D>() * (sizeof(S) > sizeof(double) ? 2 : 1);
  store_delem<S, D>(elems + (size_t)idx * NREC, e);
}

template <typename S, int D>
__global__ __launch_bounds__(64) void dense_nll_assemble_kernel(DenseGeom G, DenseModelPtrs M,
                                                               const double* __restrict__ elems,
                                                               double* __restrict__ nll,
                                                               double* __restrict__ dnll) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G.K * G.n_cand) return;
  const int k = idx % G.K, c = idx / G.K;
  constexpr int NREC = delem_doubles<D>() * (sizeof(S) > sizeof(double) ? 2 : 1);
  Vec<double, D> m0;
  Mat<double, D> P0;
  load_prior<D>(M, k, m0, P0);
  Vec<S, D> m;
  Mat<S, D> P;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    m.a[a] = S(m0.a[a]);
#pragma unroll
    for (int b = 0; b < D; ++b) P.a[a][b] = S(P0.a[a][b]);
  }
  S ll = S(0.0);
  for (int j = 0; j < G.nc; ++j) {
    const size_t rec = ((size_t)j * G.n_cand + c) * G.K + k;
    const DElem<S, D> e = load_delem<S, D>(elems + rec * NREC);
    ll = ll + delem_apply(e, m, P);
  }
  const double v = -val(ll);
  const bool fin = isfinite(v);
  nll[(size_t)k * G.n_cand + c] = fin ? v : 1e12;  // eks/core.py:650
  if (dnll) dnll[(size_t)k * G.n_cand + c] = fin ? -der(ll) : 0.0;
}

// ------------------------------------------------------------------------------------------
size_t dense_smooth_workspace_bytes(int T, int K, int D, int O) {
  (void)O;
  const int B = dense_chunk(T), nc = (T + B - 1) / B;
  const size_t nv = 3 * D * D + 2 * D + 1, rec = D + D * D;
  return align_up((size_t)nc * K * nv * 8, 256) + 2 * align_up((size_t)nc * K * rec * 8, 256) +
         align_up((size_t)T * K * rec * 8, 256);
}

size_t dense_nll_workspace_bytes(int T, int K, int D, int O, int n_cand) {
  (void)O;
  const int B = dense_chunk(T), nc = (T + B - 1) / B;
  const size_t nv = 3 * D * D + 2 * D + 1;
  return align_up((size_t)nc * K * n_cand * nv * 2 * 8, 256);
}

#define EKS_DISPATCH_D(D_, BODY) \
  switch (D_) {                  \
    case 1: { constexpr int DD = 1; BODY; } break; \
    case 2: { constexpr int DD = 2; BODY; } break; \
    case 3: { constexpr int DD = 3; BODY; } break; \
    case 4: { constexpr int DD = 4; BODY; } break; \
    case 5: { constexpr int DD = 5; BODY; } break; \
    case 6: { constexpr int DD = 6; BODY; } break; \
    default: return EKS_ERR_UNSUPPORTED;           \
  }

int dense_smooth(const eks_dims_t& d, const float* y, const float* var, const DenseModel& Mm,
                 float* ms, float* Vs, void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim;
  if (D < 1 || D > 6 || O < 1 || O > 64) return EKS_ERR_UNSUPPORTED;
  if (ws_bytes < dense_smooth_workspace_bytes(T, K, D, O)) return EKS_ERR_WORKSPACE;
  DenseGeom G{K, T, O, dense_chunk(T), 0, 1, 0};
  G.nc = (T + G.B - 1) / G.B;
  const DenseModelPtrs M{Mm.m0, Mm.S0, Mm.A, Mm.C, Mm.Q};
  const size_t nv = 3 * D * D + 2 * D + 1, rec = D + D * D;
  char* p = static_cast<char*>(ws);
  double* elems = reinterpret_cast<double*>(p);
  p += align_up((size_t)G.nc * K * nv * 8, 256);
  double* prior = reinterpret_cast<double*>(p);
  p += align_up((size_t)G.nc * K * rec * 8, 256);
  double* suffix = reinterpret_cast<double*>(p);
  p += align_up((size_t)G.nc * K * rec * 8, 256);
  double* filt = reinterpret_cast<double*>(p);
  const int lanes = K * G.nc;
  const int vs_diag = (d.flags & EKS_FLAG_VS_DIAG) ? 1 : 0;
  EKS_DISPATCH_D(D, {
    {
      ProfScope ps("dense_summarize", st);
      hipLaunchKernelGGL(dense_summarize_kernel<DD>, dim3((lanes + 63) / 64), dim3(64), 0, st, G, M,
                         Mm.s, y, var, elems);
    }
    {
      ProfScope ps("dense_scan", st);
      hipLaunchKernelGGL(dense_scan_kernel<DD>, dim3((2 * K + 63) / 64), dim3(64), 0, st, G, M, elems,
                         prior, suffix);
    }
    {
      ProfScope ps("dense_replay", st);
      hipLaunchKernelGGL(dense_replay_kernel<DD>, dim3((lanes + 63) / 64), dim3(64), 0, st, G, M,
                         Mm.s, y, var, prior, suffix, filt, ms, Vs, vs_diag);
    }
  })
  return hip_status(hipGetLastError());
}

int dense_nll(const eks_dims_t& d, const float* y, const double* rconst, const DenseModel& Mm,
              const double* s_cand, int n_cand, int per_keypoint, double* nll, double* dnll,
              void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim;
  if (D < 1 || D > 6 || O < 1 || O > 64) return EKS_ERR_UNSUPPORTED;
  if (ws_bytes < dense_nll_workspace_bytes(T, K, D, O, n_cand)) return EKS_ERR_WORKSPACE;
  DenseGeom G{K, T, O, dense_chunk(T), 0, n_cand, per_keypoint};
  G.nc = (T + G.B - 1) / G.B;
  const DenseModelPtrs M{Mm.m0, Mm.S0, Mm.A, Mm.C, Mm.Q};
  double* elems = static_cast<double*>(ws);
  const int lanes = K * G.nc * n_cand, lanes2 = K * n_cand;
  if (dnll) {
    EKS_DISPATCH_D(D, {
      hipLaunchKernelGGL((dense_nll_summarize_kernel<DualD, DD>), dim3((lanes + 63) / 64), dim3(64), 0,
                         st, G, M, s_cand, y, rconst, elems);
      hipLaunchKernelGGL((dense_nll_assemble_kernel<DualD, DD>), dim3((lanes2 + 63) / 64), dim3(64), 0,
                         st, G, M, elems, nll, dnll);
    })
  } else {
    EKS_DISPATCH_D(D, {
      hipLaunchKernelGGL((dense_nll_summarize_kernel<double, DD>), dim3((lanes + 63) / 64), dim3(64),
                         0, st, G, M, s_cand, y, rconst, elems);
      hipLaunchKernelGGL((dense_nll_assemble_kernel<double, DD>), dim3((lanes2 + 63) / 64), dim3(64),
                         0, st, G, M, elems, nll, dnll);
    })
  }
  return hip_status(hipGetLastError());
}

}  // namespace eks
