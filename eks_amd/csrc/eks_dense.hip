// gfx950 kernels for the general (D, O) smoother and loss: multicam linear path,
// D = n_latent (3..6), O = 2 * n_cameras (reference eks/multicam_smoother.py:409-443).  Same
// three-phase chunked scan as the scalar-chain path, with float64 small matrices in registers:
//   D1 dense_summarize : lane = (keypoint, chunk [, candidate]) -> element (A, b, C, eta, J [, ell])
//   D2 dense_scan_*    : block-parallel scan of the chunk elements (general element composition
//                        through Cholesky / Woodbury forms, eks_dense_math.hpp delem_combine)
//   D3 dense_replay    : lane = (keypoint, chunk): exact filter, fuse, RTS; filtered beliefs go
//                        through a per-lane scratch record stream (these problems are tiny:
//                        BASELINE config 4 is 4 keypoints x 50k frames, latency- not HBM-bound)
//   losses                : eks_nll / eks_ar1_nll, tree-composed chunk elements (below)
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

// ---- filter losses: eks_nll on the general (D, O) path (constant R, candidates s, d/dlog s) and
// eks_ar1_nll (pupil: AR(1) dynamics with explicit tangents, time-varying R_t).  Few chains,
// evaluated hundreds to thousands of times per session: everything is arranged for depth.  A
// "stream" is one (chain, candidate-or-tangent) pair.
//   L1 loss_chunks : workgroup = 64 consecutive chunks of one stream (frames 1..T-1 as predict-
//                    then-observe pairs, eks_dense_lane.hpp); each lane summarises its chunk
//                    (loss_summarize_chunk), then the 64 elements are composed in time order
//                    by a 6-level tree through LDS (delem_combine carries the log-likelihood)
//   L2 loss_reduce : the same tree over the previous level's aggregates, repeated until one
//                    element per stream remains; that launch updates the prior belief with frame
//                    0, applies the element and writes nll / dnll.
// Sensitivities are dual numbers: MODE 0 (AR(1)) stream c differentiates along (da[c], dq[c]);
// MODE 1 (scaled process noise s Q) differentiates with respect to log s.
constexpr int kLossCB = 64;

struct LossGeom {
  int K, T, O, B, nc, ns;   // ns = streams per chain
};

struct LossSpec {
  const double *a, *q, *da, *dq;   // MODE 0: [K][D] and tangents [ns][K][D]
  const double* s_cand;            // MODE 1: [ns] shared or [K][ns] per keypoint
  int per_keypoint;
  ObsNoise R;
};

template <typename S, int D, int MODE>
struct LossDyn;
template <typename S, int D>
struct LossDyn<S, D, 0> {
  using type = DynDiag<S, D>;
  static __device__ type load(const LossGeom& G, const DenseModelPtrs&, const LossSpec& P, int k, int c) {
    type dyn;
    const size_t toff = (size_t)c * G.K * D;
    load_ar1_dynamics<S, D>(P.a, P.q, P.da ? P.da + toff : nullptr, P.dq ? P.dq + toff : nullptr, k,
                            dyn.a, dyn.q);
    return dyn;
  }
};
template <typename S, int D>
struct LossDyn<S, D, 1> {
  using type = DynFull<S, D>;
  static __device__ type load(const LossGeom& G, const DenseModelPtrs& M, const LossSpec& P, int k, int c) {
    type dyn;
    const double sv = P.per_keypoint ? P.s_cand[(size_t)k * G.ns + c] : P.s_cand[c];
    load_dynamics<S, D>(M, k, make_real(S(), sv, sv), dyn.F, dyn.sQ, dyn.f_identity);
    return dyn;
  }
};

// Ordered tree reduction of the workgroup's elements (lane i holds element i of n); the result
// is in lane 0.  lds: kLossCB * NREC doubles, field-major.
template <typename S, int D>
__device__ void loss_tree_reduce(DElem<S, D>& e, int i, int n, double* lds) {
  for (int half = 1; half < n; half <<= 1) {
    const int span = half << 1;
    const bool send = (i & (span - 1)) == half, recv = (i & (span - 1)) == 0 && i + half < n;
    if (send && i < n) store_delem<S, D>(lds + i, e, kLossCB);
    __syncthreads();
    if (recv) e = delem_combine(e, load_delem<S, D>(lds + i + half, kLossCB));
  }
}

template <typename S, int D, int MODE>
__device__ void loss_finish(const LossGeom& G, const DenseModelPtrs& M, const LossSpec& P,
                            const float* __restrict__ y, const DElem<S, D>& e, int k, int c,
                            double* __restrict__ nll, double* __restrict__ dnll) {
  Vec<double, D> m0;
  Mat<double, D> P0;
  load_prior<D>(M, k, m0, P0);
  Vec<S, D> m;
  Mat<S, D> Pm;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    m.a[a] = S(m0.a[a]);
#pragma unroll
    for (int b = 0; b < D; ++b) Pm.a[a][b] = S(P0.a[a][b]);
  }
  S ll = loss_first_frame<S, D>(y, P.R, G.K, G.O, k, M, m, Pm);
  if (G.T > 1) ll = ll + delem_apply(e, m, Pm);
  if (MODE == 0) {
    if (c == 0) nll[k] = -val(ll);   // no 1e12 substitution in the pupil loss (:551-552)
    if (dnll) dnll[(size_t)c * G.K + k] = -der(ll);
  } else {
    const double v = -val(ll);
    const bool fin = isfinite(v);
    nll[(size_t)k * G.ns + c] = fin ? v : 1e12;  // eks/core.py:650
    if (dnll) dnll[(size_t)k * G.ns + c] = fin ? -der(ll) : 0.0;
  }
}

template <typename S, int D, int MODE>
__global__ __launch_bounds__(kLossCB) void loss_chunks_kernel(LossGeom G, DenseModelPtrs M, LossSpec P,
                                                             const float* __restrict__ y,
                                                             double* __restrict__ out,
                                                             double* __restrict__ nll,
                                                             double* __restrict__ dnll) {
  constexpr int NREC = delem_doubles<D>() * (sizeof(S) > sizeof(double) ? 2 : 1);
  __shared__ double lds[kLossCB * NREC];
  const int i = threadIdx.x, stream = blockIdx.y, k = stream % G.K, c = stream / G.K;
  const int j0 = blockIdx.x * kLossCB, n = min(kLossCB, G.nc - j0), j = j0 + i;
  DElem<S, D> e;
  if (i < n) {
    const typename LossDyn<S, D, MODE>::type dyn = LossDyn<S, D, MODE>::load(G, M, P, k, c);
    const int t0 = 1 + j * G.B;
    e = loss_summarize_chunk<S, D>(y, P.R, G.K, G.O, k, t0, min(G.B, G.T - t0), M, dyn);
  }
  loss_tree_reduce<S, D>(e, i, n, lds);
  if (i != 0) return;
  if (gridDim.x == 1)
    loss_finish<S, D, MODE>(G, M, P, y, e, k, c, nll, dnll);
  else
    store_delem<S, D>(out + ((size_t)stream * gridDim.x + blockIdx.x) * NREC, e);
}

template <typename S, int D, int MODE>
__global__ __launch_bounds__(kLossCB) void loss_reduce_kernel(LossGeom G, DenseModelPtrs M, LossSpec P,
                                                             const float* __restrict__ y, int n_in,
                                                             const double* __restrict__ in,
                                                             double* __restrict__ out,
                                                             double* __restrict__ nll,
                                                             double* __restrict__ dnll) {
  constexpr int NREC = delem_doubles<D>() * (sizeof(S) > sizeof(double) ? 2 : 1);
  __shared__ double lds[kLossCB * NREC];
  const int i = threadIdx.x, stream = blockIdx.y, k = stream % G.K, c = stream / G.K;
  const int j0 = blockIdx.x * kLossCB, n = min(kLossCB, n_in - j0);
  DElem<S, D> e;
  if (i < n) e = load_delem<S, D>(in + ((size_t)stream * n_in + j0 + i) * NREC);
  loss_tree_reduce<S, D>(e, i, n, lds);
  if (i != 0) return;
  if (gridDim.x == 1)
    loss_finish<S, D, MODE>(G, M, P, y, e, k, c, nll, dnll);
  else
    store_delem<S, D>(out + ((size_t)stream * gridDim.x + blockIdx.x) * NREC, e);
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

// frames per lane: short chunks keep the per-lane recursion short (a frame costs about a fifth
// of an element composition, and the tree adds one composition per doubling of the chunk
// count); they grow only when the launch would exceed a few waves per SIMD
static int loss_chunk(int T, int streams) {
  int b = 8;
  while ((long)((T + b - 1) / b) * streams > (1L << 18)) b <<= 1;
  return b;
}

// chunks cover frames 1..T-1 (frame 0 updates the prior in the finishing launch); at least one
// (possibly empty) chunk so that a launch exists to finish
static int loss_chunks(int T, int B) { return T > 1 ? (T - 1 + B - 1) / B : 1; }

static size_t loss_workspace_bytes(int T, int K, int D, int ns) {
  const int B = loss_chunk(T, K * ns), nc = loss_chunks(T, B);
  const size_t nv = 3 * D * D + 2 * D + 1;
  size_t total = 0;
  for (int n = (nc + kLossCB - 1) / kLossCB; n > 1; n = (n + kLossCB - 1) / kLossCB) {
    total += align_up((size_t)n * K * ns * nv * 2 * 8, 256);
    if (n <= kLossCB) break;
  }
  return total + 256;
}

template <typename S, int DD, int MODE>
static void loss_launch(const LossGeom& G, const DenseModelPtrs& M, const LossSpec& P, const float* y,
                        double* nll, double* dnll, char* ws, hipStream_t st) {
  constexpr size_t rec_bytes = (3 * DD * DD + 2 * DD + 1) * (sizeof(S) > sizeof(double) ? 2 : 1) * 8;
  const int streams = G.K * G.ns;
  int n = (G.nc + kLossCB - 1) / kLossCB;
  double* out = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL((loss_chunks_kernel<S, DD, MODE>), dim3(n, streams), dim3(kLossCB), 0, st, G, M, P, y,
                     out, nll, dnll);
  while (n > 1) {
    const int n_out = (n + kLossCB - 1) / kLossCB;
    double* in = out;
    out = reinterpret_cast<double*>(reinterpret_cast<char*>(in) +
                                    align_up((size_t)n * streams * rec_bytes, 256));
    hipLaunchKernelGGL((loss_reduce_kernel<S, DD, MODE>), dim3(n_out, streams), dim3(kLossCB), 0, st, G, M,
                       P, y, n, in, out, nll, dnll);
    n = n_out;
  }
}

size_t dense_nll_workspace_bytes(int T, int K, int D, int O, int n_cand) {
  (void)O;
  return loss_workspace_bytes(T, K, D, n_cand);
}

int dense_nll(const eks_dims_t& d, const float* y, const double* rconst, const DenseModel& Mm,
              const double* s_cand, int n_cand, int per_keypoint, double* nll, double* dnll,
              void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim;
  if (D < 1 || D > 6 || O < 1 || O > 64) return EKS_ERR_UNSUPPORTED;
  if ((long)K * n_cand > 65535) return EKS_ERR_UNSUPPORTED;
  if (ws_bytes < dense_nll_workspace_bytes(T, K, D, O, n_cand)) return EKS_ERR_WORKSPACE;
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

size_t ar1_nll_workspace_bytes(int T, int K, int D, int n_tan) {
  return loss_workspace_bytes(T, K, D, n_tan > 0 ? n_tan : 1);
}

int ar1_nll(const eks_dims_t& d, const float* y, const float* var, const double* m0,
            const double* S0, const double* C, const double* a, const double* q, const double* da,
            const double* dq, int n_tan, double* nll, double* dnll, void* ws, size_t ws_bytes,
            hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim;
  if (D < 1 || D > 6 || O < 1 || O > 64) return EKS_ERR_UNSUPPORTED;
  if (ws_bytes < ar1_nll_workspace_bytes(T, K, D, n_tan)) return EKS_ERR_WORKSPACE;
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
