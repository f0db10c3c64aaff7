// Lane-level bodies of the general (D, O) path; shared by eks_dense.hip and tests/host_sim.
// One lane owns one keypoint over one chunk of frames.  y, var: float [T][K][O].
#pragma once
#include "eks_dense_math.hpp"

namespace eks {

struct DenseModelPtrs {
  const double *m0, *S0, *A, *C, *Q;  // reference shapes, eks/core.py:160-166
};

template <typename S, int D>
EKS_HD void load_dynamics(const DenseModelPtrs& M, int k, S s, Mat<S, D>& F, Mat<S, D>& sQ,
                          bool& f_identity) {
  f_identity = true;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const double a = M.A[((size_t)k * D + i) * D + j];
      F.a[i][j] = S(a);
      f_identity = f_identity && (a == (i == j ? 1.0 : 0.0));
      sQ.a[i][j] = s * S(M.Q[((size_t)k * D + i) * D + j]);
    }
}

template <typename S, int D>
EKS_HD Vec<S, D> load_obs_row(const DenseModelPtrs& M, int k, int O, int o) {
  Vec<S, D> h;
#pragma unroll
  for (int i = 0; i < D; ++i) h.a[i] = S(M.C[((size_t)k * O + o) * D + i]);
  return h;
}

// K1: element of frames [t0, t0+len) of keypoint k.  CONST_R: constant observation variances
// rconst[k][o] (the loss, eks/core.py:602) and ell is accumulated; otherwise R_t from var.
template <typename S, int D, bool CONST_R>
EKS_HD DElem<S, D> dense_summarize_chunk(const float* __restrict__ y, const float* __restrict__ var,
                                         const double* __restrict__ rconst, int K, int O, int k,
                                         int t0, int len, const DenseModelPtrs& M,
                                         const Mat<S, D>& F, const Mat<S, D>& sQ, bool f_identity) {
  DElem<S, D> e = delem_identity<S, D>();
  for (int t = t0; t < t0 + len; ++t) {
    const size_t row = ((size_t)t * K + k) * O;
    for (int o = 0; o < O; ++o) {
      double r;
      if (CONST_R) {
        r = rconst[(size_t)k * O + o];
      } else {
        const float v = var[row + o];
        r = v > kVarFloor ? (double)v : (double)kVarFloor;
      }
      delem_observe(e, load_obs_row<S, D>(M, k, O, o), S((double)y[row + o]), S(r), CONST_R);
    }
    delem_predict(e, F, sQ, f_identity);
  }
  return e;
}

// K3: exact replay.  (m, P): predicted belief entering the chunk; (eta_s, J_s): information about
// the state at the first frame after the chunk.  `filt` is this lane's scratch: len records of
// D + D*D doubles (filtered mean and covariance), written forwards and read backwards.
template <int D>
EKS_HD void dense_replay_chunk(const float* __restrict__ y, const float* __restrict__ var, int K,
                               int O, int k, int t0, int len, const DenseModelPtrs& M,
                               const Mat<double, D>& F, const Mat<double, D>& sQ, bool f_identity,
                               Vec<double, D> m, Mat<double, D> P, const Vec<double, D>& eta_s,
                               const Mat<double, D>& J_s, double* __restrict__ filt,
                               float* __restrict__ ms, float* __restrict__ Vs, bool vs_diag) {
  constexpr int REC = D + D * D;
  for (int i = 0; i < len; ++i) {
    const int t = t0 + i;
    const size_t row = ((size_t)t * K + k) * O;
    for (int o = 0; o < O; ++o) {
      const Vec<double, D> h = load_obs_row<double, D>(M, k, O, o);
      const float v = var[row + o];
      const double r = v > kVarFloor ? (double)v : (double)kVarFloor;
      const Vec<double, D> u = mat_vec(P, h);
      const double g = 1.0 / (r + dot(h, u));
      const double gd = g * ((double)y[row + o] - dot(h, m));
#pragma unroll
      for (int a = 0; a < D; ++a) {
        m.a[a] += u.a[a] * gd;
#pragma unroll
        for (int b = 0; b < D; ++b) P.a[a][b] -= u.a[a] * u.a[b] * g;
      }
    }
    double* rec = filt + (size_t)i * REC;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      rec[a] = m.a[a];
#pragma unroll
      for (int b = 0; b < D; ++b) rec[D + a * D + b] = P.a[a][b];
    }
    if (!f_identity) {
      m = mat_vec(F, m);
      P = mat_mul_nt(mat_mul(F, P), F);
    }
    P = mat_add(P, sQ);
  }
  Vec<double, D> m_s;
  Mat<double, D> P_s;
  double logdet;
  condition_on_info(m, P, eta_s, J_s, m_s, P_s, logdet);
  for (int i = len - 1; i >= 0; --i) {
    const double* rec = filt + (size_t)i * REC;
    Vec<double, D> mf;
    Mat<double, D> Pf;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      mf.a[a] = rec[a];
#pragma unroll
      for (int b = 0; b < D; ++b) Pf.a[a][b] = rec[D + a * D + b];
    }
    const Mat<double, D> FP = f_identity ? Pf : mat_mul(F, Pf);                 // F Pf
    const Mat<double, D> Pp = mat_symmetrize(
        mat_add(f_identity ? Pf : mat_mul_nt(FP, F), sQ));                      // F Pf F^T + sQ
    const Mat<double, D> Z = chol_solve_mat(chol_psd(Pp), FP);                  // Pp^-1 F Pf = G^T
    const Vec<double, D> mp = f_identity ? mf : mat_vec(F, mf);
    Vec<double, D> dm;
#pragma unroll
    for (int a = 0; a < D; ++a) dm.a[a] = m_s.a[a] - mp.a[a];
    const Vec<double, D> Gdm = mat_t_vec(Z, dm);
#pragma unroll
    for (int a = 0; a < D; ++a) m_s.a[a] = mf.a[a] + Gdm.a[a];
    // P_s = Pf + G (P_s - Pp) G^T,  G = Z^T
    const Mat<double, D> dP = mat_sub(P_s, Pp);
    P_s = mat_symmetrize(mat_add(Pf, mat_mul(mat_mul_tn(Z, dP), Z)));
    const size_t ko = (size_t)(t0 + i) * K + k;
#pragma unroll
    for (int a = 0; a < D; ++a) ms[ko * D + a] = (float)m_s.a[a];
    if (vs_diag) {
#pragma unroll
      for (int a = 0; a < D; ++a) Vs[ko * D + a] = (float)P_s.a[a][a];
    } else {
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) Vs[(ko * D + a) * D + b] = (float)P_s.a[a][b];
    }
  }
}

// ---- element records in the workspace: doubles, value parts then (dual only) derivative parts
template <int D>
constexpr int delem_doubles() { return 3 * D * D + 2 * D + 1; }

template <typename S, int D>
EKS_HD void store_delem(double* __restrict__ rec, const DElem<S, D>& e) {
  constexpr int NV = delem_doubles<D>();
  int p = 0;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      rec[p] = val(e.A.a[i][j]);
      rec[p + D * D] = val(e.C.a[i][j]);
      rec[p + 2 * D * D] = val(e.J.a[i][j]);
      if constexpr (sizeof(S) > sizeof(double)) {
        rec[NV + p] = der(e.A.a[i][j]);
        rec[NV + p + D * D] = der(e.C.a[i][j]);
        rec[NV + p + 2 * D * D] = der(e.J.a[i][j]);
      }
      ++p;
    }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    rec[3 * D * D + i] = val(e.b.a[i]);
    rec[3 * D * D + D + i] = val(e.eta.a[i]);
    if constexpr (sizeof(S) > sizeof(double)) {
      rec[NV + 3 * D * D + i] = der(e.b.a[i]);
      rec[NV + 3 * D * D + D + i] = der(e.eta.a[i]);
    }
  }
  rec[NV - 1] = val(e.ell);
  if constexpr (sizeof(S) > sizeof(double)) rec[2 * NV - 1] = der(e.ell);
}

template <typename S, int D>
EKS_HD DElem<S, D> load_delem(const double* __restrict__ rec) {
  constexpr int NV = delem_doubles<D>();
  DElem<S, D> e;
  auto get = [&](int p) -> S {
    if constexpr (sizeof(S) > sizeof(double))
      return make_real(S(), rec[p], rec[NV + p]);
    else
      return S(rec[p]);
  };
  int p = 0;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      e.A.a[i][j] = get(p);
      e.C.a[i][j] = get(p + D * D);
      e.J.a[i][j] = get(p + 2 * D * D);
      ++p;
    }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    e.b.a[i] = get(3 * D * D + i);
    e.eta.a[i] = get(3 * D * D + D + i);
  }
  e.ell = get(NV - 1);
  return e;
}

template <int D>
EKS_HD void load_prior(const DenseModelPtrs& M, int k, Vec<double, D>& m, Mat<double, D>& P) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    m.a[i] = M.m0[(size_t)k * D + i];
#pragma unroll
    for (int j = 0; j < D; ++j) P.a[i][j] = M.S0[((size_t)k * D + i) * D + j];
  }
}

}  // namespace eks
