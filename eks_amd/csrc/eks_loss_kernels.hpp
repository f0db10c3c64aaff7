// Kernels and launch helper of the filter losses on the general (D, O) path, shared by eks_loss.hip
// (eks_nll: constant R, s Q, d/dlog s) and eks_loss_ar1.hip (eks_ar1_nll: AR(1) tangents,
// time-varying R) - two translation units so that the dual-number instantiations build in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "eks_dense_lane.hpp"
#include "eks_internal.hpp"

namespace eks {

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

}  // namespace eks
