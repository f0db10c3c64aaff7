// gfx950 kernels for the general (D, O) smoother: multicam linear path, D = n_latent (3..6),
// O = 2 * n_cameras (reference eks/multicam_smoother.py:409-443), and the pupil smoother's final
// pass (D = 3, O = 8).  Same three-phase chunked scan as the scalar-chain path, with float64 small
// matrices in registers; chunk elements are built predict-first (eks_dense_lane.hpp):
//   D1 dense_summarize : lane = (keypoint, chunk) -> element (A, b, C, eta, J); chunk 0's lane also
//                        updates the prior with frame 0 (the belief the scan starts from)
//   D2 dense_scan_*    : block-parallel scan of the chunk elements (general element composition
//                        through Cholesky / Woodbury forms, eks_dense_math.hpp delem_combine)
//   D3 dense_replay    : lane = (keypoint, chunk): exact filter, fuse, RTS; filtered beliefs go
//                        through a per-lane scratch record stream (these problems are tiny:
//                        BASELINE config 4 is 4 keypoints x 50k frames, latency- not HBM-bound)
// The losses of this path (eks_nll, eks_ar1_nll) live in eks_loss_kernels.hpp.
#include <hip/hip_runtime.h>

#include "eks_dense_lane.hpp"
#include "eks_internal.hpp"

namespace eks {

struct DenseGeom {
  int K, T, O, B, nc;
};

template <int D>
__global__ __launch_bounds__(64) void dense_summarize_kernel(DenseGeom G, DenseModelPtrs M,
                                                            const double* __restrict__ s,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ var,
                                                            double* __restrict__ elems,
                                                            double* __restrict__ first) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G.K * G.nc) return;
  const int k = idx % G.K, j = idx / G.K;
  Mat<double, D> F, sQ;
  bool fid;
  load_dynamics<double, D>(M, k, s[k], F, sQ, fid);
  const int t0 = j * G.B, len = min(G.B, G.T - t0);
  const DElem<double, D> e = dense_smooth_element<D>(y, var, G.K, G.O, k, t0, len, M, F, sQ, fid);
  store_delem<double, D>(elems + (size_t)idx * delem_doubles<D>(), e);
  if (j == 0) {   // the belief the scan starts from: the prior updated with frame 0
    Vec<double, D> m;
    Mat<double, D> P;
    load_prior<D>(M, k, m, P);
    belief_update_frame<D>(y, var, G.K, G.O, k, 0, M, m, P);
    double* r = first + (size_t)k * (D + D * D);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      r[a] = m.a[a];
#pragma unroll
      for (int b = 0; b < D; ++b) r[D + a * D + b] = P.a[a][b];
    }
  }
}

// ---- scan of the chunk elements, block-parallel like the scalar path (eks_diag.hip K2):
//   DS1 reduce : block = (keypoint, 64 consecutive chunks); ordered tree reduction in LDS
//   DS2 blocks : per keypoint, forward / backward walk over the few block aggregates
//   DS3 local  : Hillis-Steele inclusive scans (forward and reverse) of the block's elements in
//                LDS; the exclusive prefix is applied to the block's incoming belief, the
//                exclusive suffix pulls the block's outgoing information back.
constexpr int kDenseCB = 64;

template <int D>
__global__ __launch_bounds__(kDenseCB) void dense_scan_reduce_kernel(DenseGeom G,
                                                                    const double* __restrict__ elems,
                                                                    double* __restrict__ agg) {
  constexpr int NV = delem_doubles<D>();
  __shared__ double lds[kDenseCB * NV];
  const int k = blockIdx.x, blk = blockIdx.y, i = threadIdx.x;
  const int j = blk * kDenseCB + i;
  DElem<double, D> e = j < G.nc ? load_delem<double, D>(elems + ((size_t)j * G.K + k) * NV)
                                : delem_identity<double, D>();
  store_delem<double, D>(lds + i * NV, e);
  __syncthreads();
  for (int off = 1; off < kDenseCB; off <<= 1) {
    const bool act = (i & (2 * off - 1)) == 0;
    if (act) e = delem_combine(e, load_delem<double, D>(lds + (i + off) * NV));
    __syncthreads();
    if (act) store_delem<double, D>(lds + i * NV, e);
    __syncthreads();
  }
  if (i == 0) store_delem<double, D>(agg + ((size_t)blk * G.K + k) * NV, e);
}

template <int D>
__global__ __launch_bounds__(64) void dense_scan_blocks_kernel(DenseGeom G, int nblk,
                                                              const double* __restrict__ first,
                                                              const double* __restrict__ agg,
                                                              double* __restrict__ bprior,
                                                              double* __restrict__ bsuffix) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * G.K) return;
  constexpr int REC = D + D * D;
  constexpr int NV = delem_doubles<D>();
  const int k = idx % G.K;
  if (idx < G.K) {
    Vec<double, D> m;
    Mat<double, D> P;
    const double* f0 = first + (size_t)k * REC;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      m.a[a] = f0[a];
#pragma unroll
      for (int b = 0; b < D; ++b) P.a[a][b] = f0[D + a * D + b];
    }
    for (int q = 0; q < nblk; ++q) {
      double* r = bprior + ((size_t)q * G.K + k) * REC;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        r[a] = m.a[a];
#pragma unroll
        for (int b = 0; b < D; ++b) r[D + a * D + b] = P.a[a][b];
      }
      delem_apply(load_delem<double, D>(agg + ((size_t)q * G.K + k) * NV), m, P);
    }
  } else {
    Vec<double, D> eta = vec_zero<double, D>();
    Mat<double, D> J = mat_zero<double, D>();
    for (int q = nblk - 1; q >= 0; --q) {
      double* r = bsuffix + ((size_t)q * G.K + k) * REC;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        r[a] = eta.a[a];
#pragma unroll
        for (int b = 0; b < D; ++b) r[D + a * D + b] = J.a[a][b];
      }
      delem_back(load_delem<double, D>(agg + ((size_t)q * G.K + k) * NV), eta, J);
    }
  }
}

template <int D>
__global__ __launch_bounds__(kDenseCB) void dense_scan_local_kernel(DenseGeom G,
                                                                   const double* __restrict__ elems,
                                                                   const double* __restrict__ bprior,
                                                                   const double* __restrict__ bsuffix,
                                                                   double* __restrict__ prior,
                                                                   double* __restrict__ suffix) {
  constexpr int NV = delem_doubles<D>();
  constexpr int REC = D + D * D;
  __shared__ double lds[kDenseCB * NV];
  const int k = blockIdx.x, blk = blockIdx.y, i = threadIdx.x;
  const int j = blk * kDenseCB + i;
  const bool live = j < G.nc;
  const double* own = elems + ((size_t)j * G.K + k) * NV;
  // forward inclusive scan; the exclusive prefix of chunk i is what sits in slot i-1 afterwards
  DElem<double, D> e = live ? load_delem<double, D>(own) : delem_identity<double, D>();
  store_delem<double, D>(lds + i * NV, e);
  __syncthreads();
  for (int off = 1; off < kDenseCB; off <<= 1) {
    const bool has = i >= off;
    DElem<double, D> other;
    if (has) other = load_delem<double, D>(lds + (i - off) * NV);
    __syncthreads();
    if (has) e = delem_combine(other, e);
    store_delem<double, D>(lds + i * NV, e);
    __syncthreads();
  }
  {
    Vec<double, D> m;
    Mat<double, D> P;
    const double* r = bprior + ((size_t)blk * G.K + k) * REC;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      m.a[a] = r[a];
#pragma unroll
      for (int b = 0; b < D; ++b) P.a[a][b] = r[D + a * D + b];
    }
    if (i > 0) delem_apply(load_delem<double, D>(lds + (i - 1) * NV), m, P);
    if (live) {
      double* w = prior + ((size_t)j * G.K + k) * REC;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        w[a] = m.a[a];
#pragma unroll
        for (int b = 0; b < D; ++b) w[D + a * D + b] = P.a[a][b];
      }
    }
  }
  __syncthreads();
  // reverse inclusive scan; the exclusive suffix of chunk i sits in slot i+1 afterwards
  e = live ? load_delem<double, D>(own) : delem_identity<double, D>();
  store_delem<double, D>(lds + i * NV, e);
  __syncthreads();
  for (int off = 1; off < kDenseCB; off <<= 1) {
    const bool has = i + off < kDenseCB;
    DElem<double, D> other;
    if (has) other = load_delem<double, D>(lds + (i + off) * NV);
    __syncthreads();
    if (has) e = delem_combine(e, other);
    store_delem<double, D>(lds + i * NV, e);
    __syncthreads();
  }
  if (!live) return;
  Vec<double, D> eta;
  Mat<double, D> J;
  const double* r = bsuffix + ((size_t)blk * G.K + k) * REC;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    eta.a[a] = r[a];
#pragma unroll
    for (int b = 0; b < D; ++b) J.a[a][b] = r[D + a * D + b];
  }
  if (i + 1 < kDenseCB) delem_back(load_delem<double, D>(lds + (i + 1) * NV), eta, J);
  double* w = suffix + ((size_t)j * G.K + k) * REC;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    w[a] = eta.a[a];
#pragma unroll
    for (int b = 0; b < D; ++b) w[D + a * D + b] = J.a[a][b];
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
  if (j == 0) load_prior<D>(M, k, m, P);   // chunk 0 replays frame 0's update of the prior itself
  const int t0 = j * G.B, len = min(G.B, G.T - t0);
  dense_replay_chunk<D>(y, var, G.K, G.O, k, t0, len, M, F, sQ, fid, m, P, eta, J,
                        filt + ((size_t)k * G.T + t0) * REC, ms, Vs, vs_diag != 0);
}

// ------------------------------------------------------------------------------------------
constexpr int kDenseSmoothChunk = 32;   // frames per lane in the smoother (the scan is parallel)

size_t dense_smooth_workspace_bytes(int T, int K, int D, int O) {
  (void)O;
  const int B = kDenseSmoothChunk, nc = (T + B - 1) / B, nblk = (nc + kDenseCB - 1) / kDenseCB;
  const size_t nv = 3 * D * D + 2 * D + 1, rec = D + D * D;
  return align_up((size_t)nc * K * nv * 8, 256) + 2 * align_up((size_t)nc * K * rec * 8, 256) +
         align_up((size_t)nblk * K * nv * 8, 256) + 2 * align_up((size_t)nblk * K * rec * 8, 256) +
         align_up((size_t)T * K * rec * 8, 256) + align_up((size_t)K * rec * 8, 256);
}

int dense_smooth(const eks_dims_t& d, const float* y, const float* var, const DenseModel& Mm,
                 float* ms, float* Vs, void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim;
  if (D < 1 || D > 6 || O < 1 || O > 64) return EKS_ERR_UNSUPPORTED;
  if (ws_bytes < dense_smooth_workspace_bytes(T, K, D, O)) return EKS_ERR_WORKSPACE;
  DenseGeom G{K, T, O, kDenseSmoothChunk, 0};
  G.nc = (T + G.B - 1) / G.B;
  const int nblk = (G.nc + kDenseCB - 1) / kDenseCB;
  const DenseModelPtrs M{Mm.m0, Mm.S0, Mm.A, Mm.C, Mm.Q};
  const size_t nv = 3 * D * D + 2 * D + 1, rec = D + D * D;
  char* p = static_cast<char*>(ws);
  double* elems = reinterpret_cast<double*>(p);
  p += align_up((size_t)G.nc * K * nv * 8, 256);
  double* prior = reinterpret_cast<double*>(p);
  p += align_up((size_t)G.nc * K * rec * 8, 256);
  double* suffix = reinterpret_cast<double*>(p);
  p += align_up((size_t)G.nc * K * rec * 8, 256);
  double* agg = reinterpret_cast<double*>(p);
  p += align_up((size_t)nblk * K * nv * 8, 256);
  double* bprior = reinterpret_cast<double*>(p);
  p += align_up((size_t)nblk * K * rec * 8, 256);
  double* bsuffix = reinterpret_cast<double*>(p);
  p += align_up((size_t)nblk * K * rec * 8, 256);
  double* filt = reinterpret_cast<double*>(p);
  p += align_up((size_t)T * K * rec * 8, 256);
  double* first = reinterpret_cast<double*>(p);
  const int lanes = K * G.nc;
  const int vs_diag = (d.flags & EKS_FLAG_VS_DIAG) ? 1 : 0;
  EKS_DISPATCH_D(D, {
    {
      ProfScope ps("dense_summarize", st);
      hipLaunchKernelGGL(dense_summarize_kernel<DD>, dim3((lanes + 63) / 64), dim3(64), 0, st, G, M,
                         Mm.s, y, var, elems, first);
    }
    {
      ProfScope ps("dense_scan", st);
      const dim3 sgrid(K, nblk);
      hipLaunchKernelGGL(dense_scan_reduce_kernel<DD>, sgrid, dim3(kDenseCB), 0, st, G, elems, agg);
      hipLaunchKernelGGL(dense_scan_blocks_kernel<DD>, dim3((2 * K + 63) / 64), dim3(64), 0, st, G, nblk,
                         first, agg, bprior, bsuffix);
      hipLaunchKernelGGL(dense_scan_local_kernel<DD>, sgrid, dim3(kDenseCB), 0, st, G, elems, bprior,
                         bsuffix, prior, suffix);
    }
    {
      ProfScope ps("dense_replay", st);
      hipLaunchKernelGGL(dense_replay_kernel<DD>, dim3((lanes + 63) / 64), dim3(64), 0, st, G, M,
                         Mm.s, y, var, prior, suffix, filt, ms, Vs, vs_diag);
    }
  })
  return hip_status(hipGetLastError());
}

}  // namespace eks
