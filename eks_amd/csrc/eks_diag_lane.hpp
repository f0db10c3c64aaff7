// Lane-level bodies of the scalar-chain ("diagonal model") kernels.  One lane owns one chain
// (keypoint coordinate) over one chunk of B consecutive frames.  Shared, unchanged, between
// eks_diag.hip (one GPU lane per call) and tests/host_sim (plain loops) so the float32 numerics can
// be checked on a CPU-only box against the float64 oracle.
//
// Layout (frame-major, the layout the reference's drivers hold before they transpose for JAX,
// eks/singlecam_smoother.py:166): y, var: float [T][N], N = K*D chains, chain n = k*D + d.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "eks_math.hpp"

namespace eks {

constexpr float kVarFloor = 1e-12f;  // eks/utils.py:373 clip(var, 1e-12, inf)
// Upper clamp, not in the reference: an infinite (or > 1e30) ensemble variance means "no observation" - the gain
// is 0 - but r g = inf * 0 is NaN in every form of the update (the reference's own P - K S K' included), and one
// NaN poisons the chain for good.  At 1e30 the frame's weight is 1e-30 of anything a pixel variance can be, r g
// rounds to exactly 1 and nothing overflows in float32 or float64.  NaN variances keep mapping to the floor.
constexpr float kVarCeil = 1e30f;
EKS_HD float clip_var(float v) { return fminf(fmaxf(v, kVarFloor), kVarCeil); }

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
  p.oma = (float)(1.0 - M.A[dd]);
  p.oma2 = (float)(1.0 - M.A[dd] * M.A[dd]);
  p.c = (float)M.C[dd];
  p.q_s = (float)(M.s[k] * M.Q[dd]);
  return p;
}

EKS_HD void load_chain_prior(const DiagModel& M, int n, float& m, float& P) {
  const int k = n / M.D, d = n - k * M.D;
  m = (float)M.m0[(size_t)k * M.D + d];
  P = (float)M.S0[(size_t)k * M.D * M.D + (size_t)d * (M.D + 1)];
}

// one 8-byte store of a 2-float row (offsets o * 2 floats are 8-byte aligned; Vs comes from the
// allocator, 256-byte aligned)
EKS_HD void store_row2(float* dst, float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float row2 __attribute__((ext_vector_type(2)));
  row2 v = {a, b};
  EKS_STREAM_STORE(reinterpret_cast<row2*>(dst), v);
#else
  dst[0] = a;
  dst[1] = b;
#endif
}

// Row access of one lane's chunk.  PointerRows: (wave-uniform row pointer) + (32-bit lane offset) -
// with t0 a scalar the compiler keeps the row offset in SGPRs, one 64-bit VALU add per access is
// left.  The fused gfx950 kernels use buffer resources instead (eks_diag.hip: BufferRows, no VALU
// address arithmetic at all); the first version spent 5 VALU instructions per load on addresses.
struct PointerRows {
  const float* y;
  const float* var;
  int N, n, t0;
  EKS_HD float load_y(int i) const { return (y + (size_t)(t0 + i) * (size_t)N)[(unsigned)n]; }
  EKS_HD float load_var(int i) const { return (var + (size_t)(t0 + i) * (size_t)N)[(unsigned)n]; }
};

// The chunk's observations and variances into registers (2 B loads in flight per lane).
template <int B, bool FULL = false, typename ROWS>
EKS_HD void load_rows(const ROWS& rows, int len, float (&v0)[B], float (&v1)[B]) {
#pragma unroll
  for (int i = 0; i < B; ++i) {
    if (FULL || i < len) {
      v0[i] = rows.load_y(i);
      v1[i] = rows.load_var(i);
    }
  }
}

template <int B, bool FULL = false>
EKS_HD void load_chunk(const float* __restrict__ y, const float* __restrict__ var, int N, int n, int t0,
                       int len, float (&v0)[B], float (&v1)[B]) {
  load_rows<B, FULL>(PointerRows{y, var, N, n, t0}, len, v0, v1);
}

// element of a loaded chunk
template <int B, bool UNIT, bool FULL = false>
EKS_HD Elem<float> summarize_loaded(const float (&yy)[B], const float (&rr)[B], int len,
                                    const ChainParams<float>& p) {
  Elem<float> e = elem_identity<float>();
#pragma unroll
  for (int i = 0; i < B; ++i) {
    if (FULL || i < len) {
      const float r = clip_var(rr[i]);
      elem_append<float, UNIT>(e, yy[i], r, p);
    }
  }
  return e;
}

// K1: element of chunk j of chain n.  `len` frames starting at t0 (len <= B).  FULL: the caller
// knows len == B (every chunk but a sequence's last): no per-frame predicate is compiled in.
template <int B, bool UNIT, bool FULL = false>
EKS_HD Elem<float> summarize_chunk(const float* __restrict__ y, const float* __restrict__ var,
                                   int N, int n, int t0, int len, const ChainParams<float>& p) {
  float yy[B], rr[B];
  load_chunk<B, FULL>(y, var, N, n, t0, len, yy, rr);
  return summarize_loaded<B, UNIT, FULL>(yy, rr, len, p);
}

// K3: exact replay of chunk j of chain n from its incoming predicted belief (m, P) and the
// information (etaS, JS) about the state at the first frame AFTER the chunk: filter_loaded, the
// fusion with the information, then smooth_rows.
// forward filter over the loaded chunk: (v0, v1) = (y, var) become the filtered (mean, variance);
// (m, P) enters as the predicted belief on the chunk's first frame and leaves as the predicted
// belief on the frame after the chunk
template <int B, bool UNIT, bool FULL = false>
EKS_HD void filter_loaded(float (&v0)[B], float (&v1)[B], int len, const ChainParams<float>& p, float& m,
                          float& P) {
#pragma unroll
  for (int i = 0; i < B; ++i) {
    if (FULL || i < len) {
      const float r = clip_var(v1[i]);
      float mf, Pf;
      filter_step<float, UNIT>(m, P, v0[i], r, p, mf, Pf);
      v0[i] = mf;
      v1[i] = Pf;
    }
  }
}

// Output rows of one lane's chunk: ms[t][n] and VS_ROW == 0 -> Vs[t][n] (diagonal only);
// VS_ROW == D -> row d of the keypoint's DxD matrix, Vs[t][n*D + e] (zeros off the diagonal, the
// full `Vs (K,T,D,D)` contract of eks/core.py:297).
template <int VS_ROW>
struct PointerStore {
  float* ms;
  float* Vs;
  int N, n, d, t0;
  EKS_HD void operator()(int i, float m, float P) const {
    constexpr int W = VS_ROW == 0 ? 1 : VS_ROW;            // floats of Vs per (frame, chain)
    const size_t row = (size_t)(t0 + i) * (size_t)N;      // wave-uniform, as in PointerRows
    EKS_STREAM_STORE(ms + row + (unsigned)n, m);
    float* vrow = Vs + row * W + (unsigned)n * (unsigned)W;
    if constexpr (VS_ROW <= 1) {
      EKS_STREAM_STORE(vrow, P);
    } else if constexpr (VS_ROW == 2) {
      // the two chains of a keypoint write adjacent rows of its 2x2 matrix: 8 bytes per lane, one store
      store_row2(vrow, d == 0 ? P : 0.0f, d == 1 ? P : 0.0f);
    } else {
#pragma unroll
      for (int e = 0; e < VS_ROW; ++e) EKS_STREAM_STORE(vrow + e, (e == d) ? P : 0.0f);
    }
  }
};

// RTS pass backwards over the filtered chunk from the smoothed belief (m, P) on the frame after
// it; streams ms / Vs out through `st(i, mean, variance)`
template <int B, bool UNIT, bool FULL = false, typename ST>
EKS_HD void smooth_rows(const float (&v0)[B], const float (&v1)[B], int len, const ChainParams<float>& p,
                        float m, float P, const ST& st) {
#pragma unroll
  for (int i = B - 1; i >= 0; --i) {
    if (FULL || i < len) {
      rts_step<float, UNIT>(m, P, v0[i], v1[i], p);
      st(i, m, P);
    }
  }
}

template <int B, bool UNIT, int VS_ROW, bool FULL = false>
EKS_HD void smooth_store(const float (&v0)[B], const float (&v1)[B], float* __restrict__ ms_out,
                         float* __restrict__ Vs_out, int N, int n, int d, int t0, int len,
                         const ChainParams<float>& p, float m, float P) {
  smooth_rows<B, UNIT, FULL>(v0, v1, len, p, m, P, PointerStore<VS_ROW>{ms_out, Vs_out, N, n, d, t0});
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
