// Lane-level bodies of the scalar-chain ("diagonal model") kernels.  One lane owns one chain
// (keypoint coordinate) over one chunk of B consecutive frames.  Shared, unchanged, between
// eks_diag.hip (one GPU lane per call) and tests/host_sim (plain loops) so the float32 numerics can
// be checked on a CPU-only box against the float64 oracle.
//
// Layout (frame-major, the layout the reference's drivers hold before they transpose for JAX,
// eks/singlecam_smoother.py:166): y, var: float [T][N], N = K*D chains, chain n = k*D + d.
#pragma once
#include <cstddef>
#include <cstdint>

#include "eks_math.hpp"

namespace eks {

constexpr float kVarFloor = 1e-12f;  // eks/utils.py:373 clip(var, 1e-12, inf)

struct DiagModel {
  // device (or host, in the simulator) pointers to the reference's per-keypoint parameters
  const double* m0;  // [K][D]
  const double* S0;  // [K][D][D]
  const double* A;   // [K][D][D]
  const double* C;   // [K][D][D]   (O == D on this path)
  const double* Q;   // [K][D][D]
  const double* s;   // [K]
  int D;
};

EKS_HD ChainParams<float> load_chain_params(const DiagModel& M, int n) {
  const int k = n / M.D, d = n - k * M.D;
  const size_t dd = (size_t)k * M.D * M.D + (size_t)d * (M.D + 1);
  ChainParams<float> p;
  p.a = (float)M.A[dd];
  p.c = (float)M.C[dd];
  p.q_s = (float)(M.s[k] * M.Q[dd]);
  return p;
}

EKS_HD void load_chain_prior(const DiagModel& M, int n, float& m, float& P) {
  const int k = n / M.D, d = n - k * M.D;
  m = (float)M.m0[(size_t)k * M.D + d];
  P = (float)M.S0[(size_t)k * M.D * M.D + (size_t)d * (M.D + 1)];
}

// K1: element of chunk j of chain n.  `len` frames starting at t0 (len <= B).
template <int B, bool UNIT>
EKS_HD Elem<float> summarize_chunk(const float* __restrict__ y, const float* __restrict__ var,
                                   int N, int n, int t0, int len, const ChainParams<float>& p) {
  float yy[B], rr[B];
  const size_t base = (size_t)t0 * N + n;
#pragma unroll
  for (int i = 0; i < B; ++i) {
    if (i < len) {
      yy[i] = y[base + (size_t)i * N];
      rr[i] = var[base + (size_t)i * N];
    }
  }
  Elem<float> e = elem_identity<float>();
#pragma unroll
  for (int i = 0; i < B; ++i) {
    if (i < len) {
      const float r = rr[i] > kVarFloor ? rr[i] : kVarFloor;
      elem_append<float, UNIT>(e, yy[i], r, p);
    }
  }
  return e;
}

// K3: exact replay of chunk j of chain n from its incoming predicted belief (m, P) and the
// information (etaS, JS) about the state at the first frame AFTER the chunk.  Writes smoothed
// means ms[t][n] and covariances: VS_ROW == 0 -> Vs[t][n] (diagonal only);
// VS_ROW == D -> row d of the keypoint's DxD matrix, Vs[t][n*D + e] (zeros off the diagonal,
// the full `Vs (K,T,D,D)` contract of eks/core.py:297).
// the chunk's observations and variances into registers (2 B loads in flight per lane)
template <int B>
EKS_HD void load_chunk(const float* __restrict__ y, const float* __restrict__ var, int N, int n, int t0,
                       int len, float (&v0)[B], float (&v1)[B]) {
  const size_t base = (size_t)t0 * N + n;
#pragma unroll
  for (int i = 0; i < B; ++i) {
    if (i < len) {
      v0[i] = y[base + (size_t)i * N];
      v1[i] = var[base + (size_t)i * N];
    }
  }
}

// forward filter over the loaded chunk: (v0, v1) = (y, var) become the filtered (mean, variance);
// (m, P) enters as the predicted belief on the chunk's first frame and leaves as the predicted
// belief on the frame after the chunk
template <int B, bool UNIT>
EKS_HD void filter_loaded(float (&v0)[B], float (&v1)[B], int len, const ChainParams<float>& p, float& m,
                          float& P) {
#pragma unroll
  for (int i = 0; i < B; ++i) {
    if (i < len) {
      const float r = v1[i] > kVarFloor ? v1[i] : kVarFloor;
      float mf, Pf;
      filter_step<float, UNIT>(m, P, v0[i], r, p, mf, Pf);
      v0[i] = mf;
      v1[i] = Pf;
    }
  }
}

// RTS pass backwards over the filtered chunk from the smoothed belief (m, P) on the frame after
// it; streams ms / Vs out
template <int B, bool UNIT, int VS_ROW>
EKS_HD void smooth_store(const float (&v0)[B], const float (&v1)[B], float* __restrict__ ms_out,
                         float* __restrict__ Vs_out, int N, int n, int d, int t0, int len,
                         const ChainParams<float>& p, float m, float P) {
  const size_t base = (size_t)t0 * N + n;
#pragma unroll
  for (int i = B - 1; i >= 0; --i) {
    if (i < len) {
      rts_step<float, UNIT>(m, P, v0[i], v1[i], p);
      const size_t o = base + (size_t)i * N;
      EKS_STREAM_STORE(ms_out + o, m);
      if constexpr (VS_ROW == 0) {
        EKS_STREAM_STORE(Vs_out + o, P);
      } else if constexpr (VS_ROW == 1) {
        EKS_STREAM_STORE(Vs_out + o, P);
      } else if constexpr (VS_ROW == 2) {
        // the two chains of a keypoint write adjacent halves of its 2x2 row pair
        EKS_STREAM_STORE(Vs_out + o * 2, d == 0 ? P : 0.0f);
        EKS_STREAM_STORE(Vs_out + o * 2 + 1, d == 1 ? P : 0.0f);
      } else {
#pragma unroll
        for (int e = 0; e < VS_ROW; ++e) EKS_STREAM_STORE(Vs_out + o * VS_ROW + e, (e == d) ? P : 0.0f);
      }
    }
  }
}

template <int B, bool UNIT, int VS_ROW>
EKS_HD void replay_loaded(float (&v0)[B], float (&v1)[B], float* __restrict__ ms_out,
                          float* __restrict__ Vs_out, int N, int n, int d, int t0, int len,
                          const ChainParams<float>& p, float m, float P, float etaS, float JS) {
  filter_loaded<B, UNIT>(v0, v1, len, p, m, P);
  fuse_info(m, P, etaS, JS);  // (m, P) is now the smoothed belief on the frame after the chunk
  smooth_store<B, UNIT, VS_ROW>(v0, v1, ms_out, Vs_out, N, n, d, t0, len, p, m, P);
}

template <int B, bool UNIT, int VS_ROW>
EKS_HD void replay_chunk(const float* __restrict__ y, const float* __restrict__ var,
                         float* __restrict__ ms_out, float* __restrict__ Vs_out, int N, int n,
                         int d, int t0, int len, const ChainParams<float>& p, float m, float P,
                         float etaS, float JS) {
  float v0[B], v1[B];  // y -> mf -> (consumed);  r -> Pf
  load_chunk<B>(y, var, N, n, t0, len, v0, v1);
  replay_loaded<B, UNIT, VS_ROW>(v0, v1, ms_out, Vs_out, N, n, d, t0, len, p, m, P, etaS, JS);
}

}  // namespace eks
