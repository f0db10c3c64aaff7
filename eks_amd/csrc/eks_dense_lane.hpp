// Lane-level bodies of the general (D, O) path; shared by eks_dense.hip and tests/host_sim.
// One lane owns one keypoint over one chunk of frames.  y, var: float [T][K][O].
#pragma once
#include "eks_dense_math.hpp"
#include "eks_pinhole.hpp"

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

// AR(1) dynamics given explicitly per chain: F = diag(a[k]), process noise diag(q[k]); with a
// dual scalar the tangents (da, dq) seed the derivative parts (eks/ibl_pupil_smoother.py:542-548)
template <typename S, int D>
EKS_HD void load_ar1_dynamics(const double* __restrict__ a, const double* __restrict__ q,
                              const double* __restrict__ da, const double* __restrict__ dq, int k,
                              Vec<S, D>& av, Vec<S, D>& qv) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const size_t p = (size_t)k * D + i;
    av.a[i] = make_real(S(), a[p], da ? da[p] : 0.0);
    qv.a[i] = make_real(S(), q[p], dq ? dq[p] : 0.0);
  }
}

template <typename S, int D>
EKS_HD Vec<S, D> load_obs_row(const DenseModelPtrs& M, int k, int O, int o) {
  Vec<S, D> h;
#pragma unroll
  for (int i = 0; i < D; ++i) h.a[i] = S(M.C[((size_t)k * O + o) * D + i]);
  return h;
}

// ---- filter losses (marginal log-likelihood and its forward sensitivities) --------------------
// Two users: the smoothing-parameter loss of eks/core.py:640-650 on the general (D, O) path
// (dynamics A, s Q with d/dlog s; CONSTANT R) and the pupil loss of eks/ibl_pupil_smoother.py:
// 540-552 (AR(1) dynamics diag(a), diag(q) with explicit tangents; TIME-VARYING R_t, :514-518).
//
// Frame 0 updates the prior belief directly (loss_first_frame); every later frame t enters a chunk
// element as the pair (predict into t, observe t).  With the predict FIRST the element's
// information about its entry state is bounded by the process noise, so its (eta, J, ell) stay
// moderate even when a variance sits at the 1e-12 clip - an element that opened with an
// observation would carry y^2 / r ~ 1e14 terms that only cancel in the final assembly.
//
// Per frame the O observations are folded into information form in plain doubles (they do not
// depend on the parameters) and absorbed as D pseudo-observations (delem_observe_info).  A frame
// whose variances span more than 8 decades is absorbed observation by observation instead (the
// information matrix would lose the large-variance rows to rounding).
struct ObsNoise {
  const float* var;       // [T][K][O] time-varying ensemble variances, or
  const double* rconst;   // [K][O] constant variances (exactly one of the two is non-null)
  EKS_HD double at(size_t row, int k, int O, int o) const {
    if (rconst) return rconst[(size_t)k * O + o];
    const float v = var[row + o];
    return (double)clip_var(v);
  }
};

template <typename S, int D>
struct DynDiag {            // x' = diag(a) x + N(0, diag(q))
  Vec<S, D> a, q;
  EKS_HD void predict(DElem<S, D>& e) const { delem_predict_diag(e, a, q); }
};
template <typename S, int D>
struct DynFull {            // x' = F x + N(0, sQ)
  Mat<S, D> F, sQ;
  bool f_identity;
  EKS_HD void predict(DElem<S, D>& e) const { delem_predict(e, F, sQ, f_identity); }
};

template <typename S, int D, typename Dyn>
EKS_HD DElem<S, D> loss_summarize_chunk(const float* __restrict__ y, const ObsNoise& R, int K, int O,
                                        int k, int t0, int len, const DenseModelPtrs& M,
                                        const Dyn& dyn) {
  DElem<S, D> e = delem_identity<S, D>();
  for (int t = t0; t < t0 + len; ++t) {
    dyn.predict(e);
    const size_t row = ((size_t)t * K + k) * O;
    Mat<double, D> Lam = mat_zero<double, D>();
    Vec<double, D> nu = vec_zero<double, D>();
    double c = O * kLog2Pi, rmin = 1e300, rmax = 0.0;
    for (int o = 0; o < O; ++o) {
      const double r = R.at(row, k, O, o);
      const double yo = (double)y[row + o], w = rcp(r);
      rmin = fmin(rmin, r);
      rmax = fmax(rmax, r);
      const Vec<double, D> h = load_obs_row<double, D>(M, k, O, o);
      c += log(r) + yo * yo * w;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        nu.a[i] += h.a[i] * (w * yo);
#pragma unroll
        for (int j = 0; j < D; ++j) Lam.a[i][j] += h.a[i] * (w * h.a[j]);
      }
    }
    if (rmax <= 1e8 * rmin) {
      delem_observe_info(e, Lam, nu, c);
    } else {
      for (int o = 0; o < O; ++o)
        delem_observe(e, load_obs_row<S, D>(M, k, O, o), S((double)y[row + o]), S(R.at(row, k, O, o)),
                      true);
    }
  }
  return e;
}

// Measurement update of the belief N(m, P) with frame 0 (scalar observation at a time); returns
// the frame's log-likelihood.
template <typename S, int D>
EKS_HD S loss_first_frame(const float* __restrict__ y, const ObsNoise& R, int K, int O, int k,
                          const DenseModelPtrs& M, Vec<S, D>& m, Mat<S, D>& P) {
  S ll = S(0.0);
  const size_t row = (size_t)k * O;
  for (int o = 0; o < O; ++o) {
    const Vec<S, D> h = load_obs_row<S, D>(M, k, O, o);
    const S r = S(R.at(row, k, O, o));
    const Vec<S, D> u = mat_vec(P, h);
    const S sigma = r + dot(h, u), g = rcp(sigma), d = S((double)y[row + o]) - dot(h, m), gd = g * d;
    ll = ll - S(0.5) * (S(kLog2Pi) + log_s(sigma) + d * gd);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      m.a[i] = m.a[i] + u.a[i] * gd;
#pragma unroll
      for (int j = 0; j < D; ++j) P.a[i][j] = P.a[i][j] - u.a[i] * u.a[j] * g;
    }
  }
  return ll;
}

// ---- observation sources of the smoother ---------------------------------------------------------
// visit(t, k, xl, fn) calls fn(h, y_eff, r) for each scalar observation of frame t of chain k:
// observation row h, effective observation and its variance.
//   LinearObs  : y = C x + v (reference eks/core.py:182-186): h = C[o], y_eff = y.
//   PinholeObs : y = h(x) + v with h the calibrated multi-camera projection (reference
//                eks/core.py:188-190, eks/multicam_smoother.py:871-898), linearised at xl (or, when
//                xl is null, at the stored linearisation point of the frame):
//                h = J_o(xl), y_eff = y - h_o(xl) + J_o(xl) xl - what the extended Kalman filter's
//                update does when xl is the predicted mean.
template <int D>
struct LinearObs {
  const float *y, *var;     // [T][K][O]
  int K, O;
  DenseModelPtrs M;
  template <typename Fn>
  EKS_HD void visit(int t, int k, const double* /*xl*/, Fn&& fn) const {
    const size_t row = ((size_t)t * K + k) * O;
    for (int o = 0; o < O; ++o) {
      const float v = var[row + o];
      fn(load_obs_row<double, D>(M, k, O, o), (double)y[row + o],
         (double)clip_var(v));
    }
  }
};

template <int D>
EKS_HD LinearObs<D> make_linear_obs(const float* y, const float* var, int K, int O,
                                    const DenseModelPtrs& M) {
  return LinearObs<D>{y, var, K, O, M};
}

// the same with the keypoint's CONSTANT variances (the optimiser's loss, eks/core.py:602, :702-709)
template <int D>
struct ConstLinearObs {
  const float* y;           // [T][K][O]
  const double* rconst;     // [K][O]
  int K, O;
  DenseModelPtrs M;
  template <typename Fn>
  EKS_HD void visit(int t, int k, const double* /*xl*/, Fn&& fn) const {
    const size_t row = ((size_t)t * K + k) * O;
    for (int o = 0; o < O; ++o) fn(load_obs_row<double, D>(M, k, O, o), (double)y[row + o], rconst[(size_t)k * O + o]);
  }
};
template <int D>
EKS_HD ConstLinearObs<D> make_const_linear_obs(const float* y, const double* rconst, int K, int O,
                                               const DenseModelPtrs& M) {
  return ConstLinearObs<D>{y, rconst, K, O, M};
}

struct PinholeObs {
  const float* y;           // [T][Kd][O], O = 2 * n_cams: (u, v) per camera
  ObsNoise R;               // [T][Kd][O] or constant [Kd][O]
  int Kd, O, T;             // chain k reads the data of keypoint k % Kd
  const double* cams;       // [n_cams][kCamDoubles]
  double* xlin;             // [K][T][3] linearisation points (predicted means of the last sweep)
  template <typename Fn>
  EKS_HD void visit(int t, int k, const double* xl, Fn&& fn) const {
    const int kd = k % Kd;
    const size_t row = ((size_t)t * Kd + kd) * O;
    double X[3];
    const double* src = xl ? xl : xlin + ((size_t)k * T + t) * 3;
    X[0] = src[0]; X[1] = src[1]; X[2] = src[2];
    for (int c = 0; 2 * c < O; ++c) {
      double uv[2], J[2][3];
      pinhole_project_jac(cams + (size_t)c * kCamDoubles, X, uv, J);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        Vec<double, 3> h;
        h.a[0] = J[a][0]; h.a[1] = J[a][1]; h.a[2] = J[a][2];
        const double lin = h.a[0] * X[0] + h.a[1] * X[1] + h.a[2] * X[2];
        fn(h, (double)y[row + 2 * c + a] - uv[a] + lin, R.at(row, kd, O, 2 * c + a));
      }
    }
  }
};

// Measurement update of the belief N(m, P) with frame t of keypoint k, one scalar observation at
// a time (R_t diagonal).  Returns the frame's log-likelihood when LL is set.
template <int D, bool LL = false, typename Obs>
EKS_HD double belief_update_obs(const Obs& obs, int k, int t, const double* xl, Vec<double, D>& m,
                                Mat<double, D>& P) {
  double ll = 0.0;
  obs.visit(t, k, xl, [&](const Vec<double, D>& h, double yv, double r) {
    const Vec<double, D> u = mat_vec(P, h);
    const double sigma = r + dot(h, u);
    const double g = rcp(sigma);
    const double d = yv - dot(h, m), gd = g * d;
    if (LL) ll -= 0.5 * (kLog2Pi + log(sigma) + d * gd);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      m.a[a] += u.a[a] * gd;
      const double ug = u.a[a] * g;                  // (gain folded in; symmetric pairs once: P stays symmetric)
#pragma unroll
      for (int b = a; b < D; ++b) {
        const double pv = P.a[a][b] - ug * u.a[b];
        P.a[a][b] = pv;
        P.a[b][a] = pv;
      }
    }
  });
  return ll;
}

template <int D>
EKS_HD void belief_update_frame(const float* __restrict__ y, const float* __restrict__ var, int K,
                                int O, int k, int t, const DenseModelPtrs& M, Vec<double, D>& m,
                                Mat<double, D>& P) {
  belief_update_obs<D>(LinearObs<D>{y, var, K, O, M}, k, t, nullptr, m, P);
}

// Smoother element of frames [t0, t0+len) of keypoint k: each frame enters as the pair (predict
// into t, observe t); frame 0 of the sequence is left out (it updates the prior directly, see
// dense_replay_chunk).  The boundary state between two chunks is therefore the FILTERED state of
// the earlier chunk's last frame, and what an element knows about its entry state is bounded by
// the process noise: (eta, J) stay moderate even when an ensemble variance sits at the 1e-12
// clip.  (An element that opens with such an observation carries J ~ 1e12 and the boundary
// algebra cancels catastrophically - measured: smoothed means off by 1e7.)
template <int D, typename Obs>
EKS_HD DElem<double, D> dense_smooth_element_obs(const Obs& obs, int k, int t0, int len,
                                                 const Mat<double, D>& F, const Mat<double, D>& sQ,
                                                 bool f_identity) {
  DElem<double, D> e = delem_identity<double, D>();
  for (int t = t0 > 0 ? t0 : 1; t < t0 + len; ++t) {
    delem_predict(e, F, sQ, f_identity);
    obs.visit(t, k, nullptr, [&](const Vec<double, D>& h, double yv, double r) {
      delem_observe(e, h, yv, r, false);
    });
  }
  return e;
}

template <int D>
EKS_HD DElem<double, D> dense_smooth_element(const float* __restrict__ y, const float* __restrict__ var,
                                             int K, int O, int k, int t0, int len,
                                             const DenseModelPtrs& M, const Mat<double, D>& F,
                                             const Mat<double, D>& sQ, bool f_identity) {
  return dense_smooth_element_obs<D>(LinearObs<D>{y, var, K, O, M}, k, t0, len, F, sQ, f_identity);
}

// K3: exact replay of frames [t0, t0+len).  (m, P): the filtered belief of frame t0-1 (the prior
// itself when t0 == 0); (eta_s, J_s): what all later frames say about the state at the chunk's
// last frame.  `filt` is this lane's scratch: len records of D + D*D doubles (filtered mean and
// covariance), written forwards and read backwards.
//
// EKF (extended filter, PinholeObs): every frame is linearised at the lane's own predicted mean
// (what the reference's dynamax filter does), the predicted mean replaces the
// stored linearisation point and the largest change is returned through `resid`; `ll` receives
// the chunk's log-likelihood.  ms == nullptr: filter only (no records, no backward pass).
// SCORE (linear observations): no outputs; `ll_out` receives the chunk's log-likelihood and `resid_out` its
// share of d loglik / d log s by Fisher's identity (eks_dense_wave.hip has the formula), including the
// transition into the chunk's first frame.
template <int D, bool EKF, typename Obs, bool SCORE = false>
EKS_HD void dense_replay_chunk_obs(const Obs& obs, int K, int k, int t0, int len,
                                   const Mat<double, D>& F, const Mat<double, D>& sQ,
                                   bool f_identity, Vec<double, D> m, Mat<double, D> P,
                                   const Vec<double, D>& eta_s, const Mat<double, D>& J_s,
                                   double* __restrict__ filt, float* __restrict__ ms,
                                   float* __restrict__ Vs, bool vs_diag, double* __restrict__ xlin,
                                   double* ll_out, double* resid_out, size_t fs = 1) {
  const Vec<double, D> m_in = m;
  const Mat<double, D> P_in = P;
  // fs: distance in doubles between consecutive fields of the scratch records (1: a lane's records are
  // contiguous; K: the records of the K keypoints are interleaved field by field, so the lanes of a wave -
  // consecutive keypoints - read and write whole segments.  Per-lane contiguous records cost the wide
  // multicam shape 1.28 ms in the replay: every 8-byte access of a wave touched 64 different lines.)
  constexpr int REC = D + D * D;
  double ll = 0.0, resid = 0.0;
  for (int i = 0; i < len; ++i) {
    const int t = t0 + i;
    if (t > 0) {
      if (!f_identity) {
        m = mat_vec(F, m);
        P = mat_mul_nt(mat_mul(F, P), F);
      }
      P = mat_add(P, sQ);
    }
    if constexpr (EKF) {
      // A non-finite predicted mean (a camera-plane crossing or garbage from a poor linearisation
      // in an EARLY sweep - the sequential filter never sees it) must not be stored: the next
      // sweep's elements would be built from NaN and every later prefix poisoned for good.  Such a
      // frame keeps its old linearisation point and reports an unconverged sweep.
      double xl[D];
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double old = xlin[a + (size_t)i * D];
        const bool fin = fabs(m.a[a]) <= 1.7e308;          // false for NaN and +-inf
        xl[a] = fin ? m.a[a] : old;
        const double scale = fabs(old) > 1.0 ? fabs(old) : 1.0;
        const double ch = fin ? fabs(xl[a] - old) / scale : 1e300;
        resid = ch > resid ? ch : resid;
        xlin[a + (size_t)i * D] = xl[a];
      }
      ll += belief_update_obs<D, true>(obs, k, t, xl, m, P);
    } else if constexpr (SCORE) {
      ll += belief_update_obs<D, true>(obs, k, t, nullptr, m, P);
    } else {
      belief_update_obs<D>(obs, k, t, nullptr, m, P);
    }
    if (!SCORE && ms == nullptr) continue;
    double* rec = filt + (size_t)i * REC * fs;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      rec[a * fs] = m.a[a];
#pragma unroll
      for (int b = 0; b < D; ++b) rec[(D + a * D + b) * fs] = P.a[a][b];
    }
  }
  if constexpr (EKF) {
    *ll_out = ll;
    *resid_out = resid;
  }
  if (!SCORE && ms == nullptr) return;
  auto emit = [&](int i, const Vec<double, D>& mo, const Mat<double, D>& Po) {
    if constexpr (SCORE) return;
    const size_t ko = (size_t)(t0 + i) * K + k;
#pragma unroll
    for (int a = 0; a < D; ++a) EKS_STREAM_STORE(ms + ko * D + a, (float)mo.a[a]);
    if (vs_diag) {
#pragma unroll
      for (int a = 0; a < D; ++a) EKS_STREAM_STORE(Vs + ko * D + a, (float)Po.a[a][a]);
    } else {
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) EKS_STREAM_STORE(Vs + (ko * D + a) * D + b, (float)Po.a[a][b]);
    }
  };
  Vec<double, D> m_s;
  Mat<double, D> P_s;
  double logdet;
  condition_on_info(m, P, eta_s, J_s, m_s, P_s, logdet);      // smoothed last frame of the chunk
  emit(len - 1, m_s, P_s);
  Mat<double, D> Qi;                                          // SCORE: (sQ)^-1
  double score = 0.0;
  if constexpr (SCORE) {
    Mat<double, D> eye = mat_zero<double, D>();
#pragma unroll
    for (int a = 0; a < D; ++a) eye.a[a][a] = 1.0;
    Qi = chol_solve_mat(chol_factor(sQ), eye);
  }
  for (int i = len - 2; i >= (SCORE ? -1 : 0); --i) {
    Vec<double, D> mf;
    Mat<double, D> Pf;
    if (i >= 0) {
      const double* rec = filt + (size_t)i * REC * fs;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        mf.a[a] = rec[a * fs];
#pragma unroll
        for (int b = 0; b < D; ++b) Pf.a[a][b] = rec[(D + a * D + b) * fs];
      }
    } else {                                                  // SCORE: back to the belief that entered the chunk
      if (t0 == 0 || len == 0) break;
      mf = m_in;
      Pf = mat_symmetrize(P_in);
    }
    const Vec<double, D> m_next = m_s;
    const Mat<double, D> P_next = P_s;
    const Mat<double, D> FP = f_identity ? Pf : mat_mul(F, Pf);                 // F Pf
    const Mat<double, D> Pp = mat_symmetrize(
        mat_add(f_identity ? Pf : mat_mul_nt(FP, F), sQ));                      // F Pf F^T + sQ
    const Mat<double, D> Z = chol_solve_mat(chol_factor(Pp), FP);                  // Pp^-1 F Pf = G^T
    const Vec<double, D> mp = f_identity ? mf : mat_vec(F, mf);
    Vec<double, D> dm;
#pragma unroll
    for (int a = 0; a < D; ++a) dm.a[a] = m_s.a[a] - mp.a[a];
    const Vec<double, D> Gdm = mat_t_vec(Z, dm);
#pragma unroll
    for (int a = 0; a < D; ++a) m_s.a[a] = mf.a[a] + Gdm.a[a];
    // P_s = Pf + G (P_s - Pp) G^T,  G = Z^T
    const Mat<double, D> dP = mat_sub(P_s, Pp);
    P_s = mat_sandwich_tn_plus(Z, dP, Pf);
    if constexpr (SCORE) {
      const Vec<double, D> Fm = f_identity ? m_s : mat_vec(F, m_s);
      Vec<double, D> dw;
#pragma unroll
      for (int a = 0; a < D; ++a) dw.a[a] = m_next.a[a] - Fm.a[a];
      const Mat<double, D> Cx = mat_mul_tn(Z, P_next);        // Cov(x_i, x_{i+1} | y)
      const Mat<double, D> FC = f_identity ? Cx : mat_mul(F, Cx);
      const Mat<double, D> FVF = f_identity ? P_s : mat_mul_nt(mat_mul(F, P_s), F);
      double tr = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b)
          tr += Qi.a[a][b] * (dw.a[a] * dw.a[b] + P_next.a[a][b] + FVF.a[a][b] - FC.a[a][b] - FC.a[b][a]);
      score += 0.5 * (tr - (double)D);
    } else {
      emit(i, m_s, P_s);
    }
  }
  if constexpr (SCORE) {
    *ll_out = ll;
    *resid_out = score;
  }
}

template <int D>
EKS_HD void dense_replay_chunk(const float* __restrict__ y, const float* __restrict__ var, int K,
                               int O, int k, int t0, int len, const DenseModelPtrs& M,
                               const Mat<double, D>& F, const Mat<double, D>& sQ, bool f_identity,
                               Vec<double, D> m, Mat<double, D> P, const Vec<double, D>& eta_s,
                               const Mat<double, D>& J_s, double* __restrict__ filt,
                               float* __restrict__ ms, float* __restrict__ Vs, bool vs_diag) {
  dense_replay_chunk_obs<D, false>(LinearObs<D>{y, var, K, O, M}, K, k, t0, len, F, sQ, f_identity,
                                   m, P, eta_s, J_s, filt, ms, Vs, vs_diag, nullptr, nullptr,
                                   nullptr);
}

// ---- element records in the workspace: doubles, value parts then (dual only) derivative parts
template <int D>
constexpr int delem_doubles() { return 3 * D * D + 2 * D + 1; }

// `stride` = distance in doubles between consecutive fields of one record (1: a contiguous
// record in global memory; CB: field-major records of a workgroup in LDS, conflict-free)
template <typename S, int D>
EKS_HD void store_delem(double* __restrict__ rec_, const DElem<S, D>& e, int stride = 1) {
  constexpr int NV = delem_doubles<D>();
  struct Strided {
    double* p;
    int s;
    EKS_HD double& operator[](int i) const { return p[(size_t)i * s]; }
  } rec{rec_, stride};
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
EKS_HD DElem<S, D> load_delem(const double* __restrict__ rec_, int stride = 1) {
  constexpr int NV = delem_doubles<D>();
  struct Strided {
    const double* p;
    int s;
    EKS_HD double operator[](int i) const { return p[(size_t)i * s]; }
  } rec{rec_, stride};
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
