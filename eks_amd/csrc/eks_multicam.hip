// gfx950 kernels either side of the Kalman path of the linear multi-camera driver:
//
//   maha_inflate_kernel : one pass of the variance-inflation loop (reference
//                         eks/multicam_smoother.py:695-708): per frame, the factor-analysis
//                         reconstruction residual and each view's 2x2 Mahalanobis distance
//                         (eks/stats.py:119-151, Python loops over frames upstream), then
//                         inflate_variance (:724-764): variances of the (frame, view) pairs beyond
//                         the threshold x scalar, the whole frame when there are exactly two views.
//                         The factor-analysis fit between passes stays on the host (sklearn).
//   multicam_tables_kernel : the output epilogue (eks/multicam_smoother.py:481-544): reprojection
//                         C m + mean, posterior variance diag(C V C') + ensemble variance and the
//                         pass-through ensemble columns, written straight in the drivers' (T, K, 9)
//                         per-camera layout, plus the latent table.
//
// Both are one thread per frame (x keypoint): a handful of small-matrix operations in float64
// registers, HBM-bound.  No MFMA (n_latent <= 6, 2V <= 16).
#include <hip/hip_runtime.h>

#include "eks_internal.hpp"
#include "eks_math.hpp"

namespace eks {

constexpr int kMaxViews = 8;

// Symmetric positive definite L x L inverse in place (Cholesky; L <= 6).  Returns false if a pivot
// is not positive (numpy.linalg.inv would return garbage or raise there).
template <int L>
__device__ __forceinline__ bool spd_inverse(double (&A)[L][L]) {
  double G[L][L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= G[i][k] * G[j][k];
      if (i == j) {
        if (!(s > 0.0)) return false;
        G[i][i] = sqrt(s);
      } else {
        G[i][j] = s / G[j][j];
      }
    }
  }
  // inverse of the lower factor, then A^-1 = G^-T G^-1
  double Gi[L][L];
#pragma unroll
  for (int i = 0; i < L; ++i) {
#pragma unroll
    for (int j = 0; j < L; ++j) Gi[i][j] = 0.0;
    Gi[i][i] = 1.0 / G[i][i];
#pragma unroll
    for (int j = 0; j < i; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = j; k < i; ++k) s -= G[i][k] * Gi[k][j];
      Gi[i][j] = s / G[i][i];
    }
  }
#pragma unroll
  for (int i = 0; i < L; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = i; k < L; ++k) s += Gi[k][i] * Gi[k][j];
      A[i][j] = s;
      A[j][i] = s;
    }
  return true;
}

// x [K][N][2C] float64 (centred predictions), v [K][N][2C] float32 in/out, W [K][2C][L], mu [K][2C]
// float64.  1 / (v + epsilon) is formed in float32, as NumPy does for the reference's float32
// variance arrays (eks/stats.py:121 with the dtype flow of SURVEY.md A.4); everything else float64.
template <int L>
__global__ __launch_bounds__(256) void maha_inflate_kernel(int K, int N, int C,
                                                          const double* __restrict__ x,
                                                          float* __restrict__ v,
                                                          const double* __restrict__ W,
                                                          const double* __restrict__ mu,
                                                          const int32_t* __restrict__ active,
                                                          float eps, double threshold, float scalar,
                                                          double* __restrict__ maha,
                                                          int32_t* __restrict__ n_inflated) {
  const int k = blockIdx.y;
  if (active && !active[k]) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int O = 2 * C;
  bool any = false;
  if (i < N) {
    const double* Wk = W + (size_t)k * O * L;
    const double* muk = mu + (size_t)k * O;
    const size_t row = ((size_t)k * N + i) * O;
    double A[L][L], b[L];
#pragma unroll
    for (int a = 0; a < L; ++a) {
      b[a] = 0.0;
#pragma unroll
      for (int c = 0; c < L; ++c) A[a][c] = 0.0;
    }
    for (int o = 0; o < O; ++o) {
      const double p = (double)(1.0f / (v[row + o] + eps));
      const double r = x[row + o] - muk[o];
#pragma unroll
      for (int a = 0; a < L; ++a) {
        const double wp = Wk[o * L + a] * p;
        b[a] += wp * r;
#pragma unroll
        for (int c = 0; c <= a; ++c) A[a][c] += wp * Wk[o * L + c];
      }
    }
#pragma unroll
    for (int a = 0; a < L; ++a)
#pragma unroll
      for (int c = a + 1; c < L; ++c) A[a][c] = A[c][a];
    const bool ok = spd_inverse<L>(A);                      // A = B = (W' P W)^-1
    double z[L];
#pragma unroll
    for (int a = 0; a < L; ++a) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < L; ++c) s += A[a][c] * b[c];
      z[a] = s;
    }
    bool hit[kMaxViews];
    bool frame_hit = false;
    for (int c = 0; c < C; ++c) {
      double d2[2], WB[2][L];
      for (int q = 0; q < 2; ++q) {
        const int o = 2 * c + q;
        double xh = muk[o];
#pragma unroll
        for (int a = 0; a < L; ++a) xh += Wk[o * L + a] * z[a];
        d2[q] = x[row + o] - xh;
#pragma unroll
        for (int a = 0; a < L; ++a) {
          double s = 0.0;
#pragma unroll
          for (int e = 0; e < L; ++e) s += Wk[o * L + e] * A[e][a];
          WB[q][a] = s;
        }
      }
      double q00 = (double)v[row + 2 * c], q01 = 0.0, q11 = (double)v[row + 2 * c + 1];
#pragma unroll
      for (int a = 0; a < L; ++a) {
        q00 += WB[0][a] * Wk[(2 * c) * L + a];
        q01 += WB[0][a] * Wk[(2 * c + 1) * L + a];
        q11 += WB[1][a] * Wk[(2 * c + 1) * L + a];
      }
      const double det = q00 * q11 - q01 * q01;
      const double m = ok ? (d2[0] * (q11 * d2[0] - q01 * d2[1]) + d2[1] * (q00 * d2[1] - q01 * d2[0])) / det
                          : nan("");
      if (maha) maha[((size_t)k * N + i) * C + c] = m;
      hit[c] = m > threshold;
      frame_hit = frame_hit || hit[c];
    }
    for (int c = 0; c < C; ++c) {
      // with exactly two views a hit in either inflates the whole frame (reference :757-759)
      if (hit[c] || (C == 2 && frame_hit)) {
        v[row + 2 * c] *= scalar;
        v[row + 2 * c + 1] *= scalar;
      }
    }
    any = frame_hit;
  }
  const unsigned long long ballot = __ballot(any);
  if ((threadIdx.x & 63) == 0 && ballot) atomicAdd(&n_inflated[k], (int)__popcll(ballot));
}

int maha_inflate(int K, int N, int C, int L, const double* x, float* v, const double* W, const double* mu,
                 const int32_t* active, double epsilon, double threshold, double scalar, double* maha,
                 int32_t* n_inflated, hipStream_t st) {
  if (C < 2 || C > kMaxViews) return EKS_ERR_UNSUPPORTED;
  hipError_t e = hipMemsetAsync(n_inflated, 0, sizeof(int32_t) * K, st);
  if (e != hipSuccess) return hip_status(e);
  const dim3 grid((N + 255) / 256, K), block(256);
#define EKS_MAHA(LL)                                                                                   \
  case LL:                                                                                             \
    hipLaunchKernelGGL(maha_inflate_kernel<LL>, grid, block, 0, st, K, N, C, x, v, W, mu, active,      \
                       (float)epsilon, threshold, (float)scalar, maha, n_inflated);                   \
    break;
  switch (L) {
    EKS_MAHA(1)
    EKS_MAHA(2)
    EKS_MAHA(3)
    EKS_MAHA(4)
    EKS_MAHA(5)
    EKS_MAHA(6)
    default: return EKS_ERR_UNSUPPORTED;
  }
#undef EKS_MAHA
  return hip_status(hipGetLastError());
}

// stats [V][T][K][5] float32 (x, y, var_x, var_y, likelihood); ev [T][K][2V] float32; ms [T][K][D],
// Vs [T][K][D][D] float32; Cm [K][2V][D], mean [V][K][2] float64 -> tables [V][T][K][9] float64 and
// (optional) latent [T][K][2D] float64 = (m, diag V).
template <int D>
__global__ __launch_bounds__(256) void multicam_tables_kernel(int V, int T, int K,
                                                             const float* __restrict__ stats,
                                                             const float* __restrict__ ev,
                                                             const float* __restrict__ ms,
                                                             const float* __restrict__ Vs,
                                                             const double* __restrict__ Cm,
                                                             const double* __restrict__ mean,
                                                             double* __restrict__ tables,
                                                             double* __restrict__ latent) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)V * T * K) return;
  const int k = (int)(idx % K);
  const long vt = idx / K;
  const int t = (int)(vt % T), c = (int)(vt / T);
  const size_t tk = (size_t)t * K + k;
  double m[D], S[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    m[a] = (double)ms[tk * D + a];
#pragma unroll
    for (int b = 0; b < D; ++b) S[a][b] = (double)Vs[(tk * D + a) * D + b];
  }
  double out[9];
  const int O = 2 * V;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int o = 2 * c + q;
    const double* Cr = Cm + ((size_t)k * O + o) * D;
    double ym = mean[((size_t)c * K + k) * 2 + q], yv = (double)ev[tk * O + o];
#pragma unroll
    for (int a = 0; a < D; ++a) {
      ym += Cr[a] * m[a];
      double s = 0.0;
#pragma unroll
      for (int b = 0; b < D; ++b) s += S[a][b] * Cr[b];
      yv += Cr[a] * s;
    }
    out[q] = ym;
    out[7 + q] = yv;
    out[5 + q] = (double)ev[tk * O + o];
  }
  const float* st = stats + ((size_t)c * T * K + tk) * 5;
  out[2] = (double)st[4];
  out[3] = (double)st[0];
  out[4] = (double)st[1];
  double* dst = tables + (size_t)idx * 9;
#pragma unroll
  for (int q = 0; q < 9; ++q) EKS_STREAM_STORE(dst + q, out[q]);
  if (latent && c == 0) {
    double* lt = latent + tk * 2 * D;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      EKS_STREAM_STORE(lt + a, m[a]);
      EKS_STREAM_STORE(lt + D + a, S[a][a]);
    }
  }
}

int multicam_tables(int V, int T, int K, int D, const float* stats, const float* ev, const float* ms,
                    const float* Vs, const double* Cm, const double* mean, double* tables, double* latent,
                    hipStream_t st) {
  const long n = (long)V * T * K;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  EKS_DISPATCH_D(D, hipLaunchKernelGGL(multicam_tables_kernel<DD>, grid, block, 0, st, V, T, K, stats, ev, ms,
                                       Vs, Cm, mean, tables, latent));
  return hip_status(hipGetLastError());
}

}  // namespace eks

EKS_DEFINE_TOUCH(multicam)
