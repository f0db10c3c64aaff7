// Small dense Kalman algebra in registers for the general (D, O) path: multicam linear,
// D = n_latent, O = 2 * n_cameras (reference eks/multicam_smoother.py:412-443, :554-597).
// Everything is float64 (inputs/outputs stay float32): the covariance-form updates subtract, and
// float64 keeps that harmless.  Scalar type S is double or a dual number, so
// the same code yields d/dlog s.  All loops have compile-time bounds: matrices live in VGPRs.
//
// R_t is diagonal on this path (eks/utils.py:368-377), so a frame's O observations are absorbed
// one scalar at a time (rank-1 updates) - or, in the losses, as D pseudo-observations after
// folding the frame into information form (delem_observe_info): exact, and no O x O inverse is
// ever formed in the per-frame code.
#pragma once
#include "eks_nll_lane.hpp"

namespace eks {

EKS_HD double sqrt_s(double x) { return sqrt(x); }
EKS_HD DualD sqrt_s(DualD x) {
  const double r = sqrt(x.v);
  return DualD(r, x.d / (2.0 * r));
}
EKS_HD double log_s(double x) { return log(x); }
EKS_HD DualD log_s(DualD x) { return DualD(log(x.v), x.d / x.v); }

template <typename S, int D>
struct Mat {
  S a[D][D];
};
template <typename S, int D>
struct Vec {
  S a[D];
};

template <typename S, int D>
EKS_HD Mat<S, D> mat_zero() {
  Mat<S, D> m;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) m.a[i][j] = S(0.0);
  return m;
}
template <typename S, int D>
EKS_HD Mat<S, D> mat_eye() {
  Mat<S, D> m = mat_zero<S, D>();
#pragma unroll
  for (int i = 0; i < D; ++i) m.a[i][i] = S(1.0);
  return m;
}
template <typename S, int D>
EKS_HD Vec<S, D> vec_zero() {
  Vec<S, D> v;
#pragma unroll
  for (int i = 0; i < D; ++i) v.a[i] = S(0.0);
  return v;
}
template <typename S, int D>
EKS_HD Mat<S, D> mat_mul(const Mat<S, D>& x, const Mat<S, D>& y) {
  Mat<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      S acc = S(0.0);
#pragma unroll
      for (int k = 0; k < D; ++k) acc = acc + x.a[i][k] * y.a[k][j];
      o.a[i][j] = acc;
    }
  return o;
}
// x * y^T
template <typename S, int D>
EKS_HD Mat<S, D> mat_mul_nt(const Mat<S, D>& x, const Mat<S, D>& y) {
  Mat<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      S acc = S(0.0);
#pragma unroll
      for (int k = 0; k < D; ++k) acc = acc + x.a[i][k] * y.a[j][k];
      o.a[i][j] = acc;
    }
  return o;
}
// x^T * y
template <typename S, int D>
EKS_HD Mat<S, D> mat_mul_tn(const Mat<S, D>& x, const Mat<S, D>& y) {
  Mat<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      S acc = S(0.0);
#pragma unroll
      for (int k = 0; k < D; ++k) acc = acc + x.a[k][i] * y.a[k][j];
      o.a[i][j] = acc;
    }
  return o;
}
template <typename S, int D>
EKS_HD Vec<S, D> mat_vec(const Mat<S, D>& x, const Vec<S, D>& v) {
  Vec<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    S acc = S(0.0);
#pragma unroll
    for (int k = 0; k < D; ++k) acc = acc + x.a[i][k] * v.a[k];
    o.a[i] = acc;
  }
  return o;
}
template <typename S, int D>
EKS_HD Vec<S, D> mat_t_vec(const Mat<S, D>& x, const Vec<S, D>& v) {
  Vec<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    S acc = S(0.0);
#pragma unroll
    for (int k = 0; k < D; ++k) acc = acc + x.a[k][i] * v.a[k];
    o.a[i] = acc;
  }
  return o;
}
template <typename S, int D>
EKS_HD S dot(const Vec<S, D>& x, const Vec<S, D>& y) {
  S acc = S(0.0);
#pragma unroll
  for (int k = 0; k < D; ++k) acc = acc + x.a[k] * y.a[k];
  return acc;
}
template <typename S, int D>
EKS_HD Mat<S, D> mat_add(const Mat<S, D>& x, const Mat<S, D>& y) {
  Mat<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) o.a[i][j] = x.a[i][j] + y.a[i][j];
  return o;
}
template <typename S, int D>
EKS_HD Mat<S, D> mat_sub(const Mat<S, D>& x, const Mat<S, D>& y) {
  Mat<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) o.a[i][j] = x.a[i][j] - y.a[i][j];
  return o;
}
template <typename S, int D>
EKS_HD Mat<S, D> mat_symmetrize(const Mat<S, D>& x) {
  Mat<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) o.a[i][j] = S(0.5) * (x.a[i][j] + x.a[j][i]);
  return o;
}

// Pivot of the Cholesky factorisation: l = sqrt(sum) and 1 / l.  In float64 on the device both come from ONE
// v_rsq_f64 seed refined by two Newton steps (y <- y (1.5 - 0.5 x y^2): full double accuracy after the second,
// l = x y) instead of an IEEE square root followed by an IEEE division - about 8 dependent instructions instead
// of 27, on the critical path of every composition of the scans and of every RTS step (round 3: the narrow-
// session kernels are bounded by exactly these chains).  The host keeps sqrt and 1 / x.
template <typename S>
EKS_HD void chol_pivot(const S& sum, S& l, S& inv) {
  l = sqrt_s(sum);
  inv = rcp(l);
}
#if defined(__HIP_DEVICE_COMPILE__)
template <>
EKS_HD void chol_pivot<double>(const double& sum, double& l, double& inv) {
  double y = __builtin_amdgcn_rsq(sum);
  const double h = 0.5 * sum;
  y = y * (1.5 - h * y * y);
  y = y * (1.5 - h * y * y);
  inv = y;
  l = sum * y;
}
// dual numbers: the value part the same way, d l = d x / (2 l), d (1 / l) = - d l / l^2
template <>
EKS_HD void chol_pivot<DualD>(const DualD& sum, DualD& l, DualD& inv) {
  double lv, iv;
  chol_pivot<double>(sum.v, lv, iv);
  const double dl = 0.5 * sum.d * iv;
  l = DualD(lv, dl);
  inv = DualD(iv, -iv * iv * dl);
}
#endif

// Lower Cholesky factor of a symmetric PSD matrix with the inverse of its diagonal (the solves multiply by
// it); a non-positive pivot zeroes its column (semi-definite factor) so singular covariances do not poison
// the recursion.
template <typename S, int D>
struct CholF {
  Mat<S, D> L;
  Vec<S, D> invd;
};

template <typename S, int D>
EKS_HD CholF<S, D> chol_factor(const Mat<S, D>& P) {
  CholF<S, D> F;
  F.L = mat_zero<S, D>();
#pragma unroll
  for (int j = 0; j < D; ++j) {
    S sum = P.a[j][j];
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k < j) sum = sum - F.L.a[j][k] * F.L.a[j][k];
    const bool ok = val(sum) > 0.0;
    S ljj = S(0.0), inv = S(0.0);
    if (ok) chol_pivot(sum, ljj, inv);
    F.L.a[j][j] = ljj;
    F.invd.a[j] = inv;
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (i > j) {
        S t = P.a[i][j];
#pragma unroll
        for (int k = 0; k < D; ++k)
          if (k < j) t = t - F.L.a[i][k] * F.L.a[j][k];
        F.L.a[i][j] = t * inv;
      }
  }
  return F;
}
template <typename S, int D>
EKS_HD Mat<S, D> chol_psd(const Mat<S, D>& P) {
  return chol_factor(P).L;
}

// Solve (Lg Lg^T) x = z for a vector (Lg lower, positive diagonal).
template <typename S, int D>
EKS_HD Vec<S, D> chol_solve(const CholF<S, D>& F, const Vec<S, D>& z) {
  Vec<S, D> w;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    S t = z.a[i];
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k < i) t = t - F.L.a[i][k] * w.a[k];
    w.a[i] = t * F.invd.a[i];
  }
  Vec<S, D> x;
#pragma unroll
  for (int ii = 0; ii < D; ++ii) {
    const int i = D - 1 - ii;
    S t = w.a[i];
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k > i) t = t - F.L.a[k][i] * x.a[k];
    x.a[i] = t * F.invd.a[i];
  }
  return x;
}
// Solve (Lg Lg^T) X = Z column by column.
template <typename S, int D>
EKS_HD Mat<S, D> chol_solve_mat(const CholF<S, D>& F, const Mat<S, D>& Z) {
  Mat<S, D> X;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    Vec<S, D> z;
#pragma unroll
    for (int i = 0; i < D; ++i) z.a[i] = Z.a[i][c];
    const Vec<S, D> x = chol_solve(F, z);
#pragma unroll
    for (int i = 0; i < D; ++i) X.a[i][c] = x.a[i];
  }
  return X;
}

// ---- triangular and symmetric forms (round 4).  A composition of two scan elements is ~650 dependent-ish float64
// instructions in the general forms above, and the narrow-session kernels run ONE wave per SIMD through ten of them
// per launch: their time IS this instruction count (in-kernel stamps: ~2 us per composition).  With C = L L^T,
// G = I + L^T J L = Lg Lg^T and the two D x D matrices
//     W = L Lg^-T,   Z = (J L) Lg^-T          (rows solved against the lower factor: right_solve_lt)
// every term of the composition is a product with W or Z:
//     M = (I + C J)^-1 = I - W Z^T,   M C = W W^T,   M^T J = J - Z Z^T
// - no D x D solve against G at all, symmetric results by construction (computed once per pair), and products with
// L skip its zeros.  ~370 instructions for D = 3.
// X L for lower-triangular L (entries above the diagonal are not read)
template <typename S, int D>
EKS_HD Mat<S, D> mat_mul_lower(const Mat<S, D>& x, const Mat<S, D>& L) {
  Mat<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      S acc = x.a[i][j] * L.a[j][j];
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (k > j) acc = acc + x.a[i][k] * L.a[k][j];
      o.a[i][j] = acc;
    }
  return o;
}
// lower triangle of I + L^T Y (the rest is zero: what chol_factor reads)
template <typename S, int D>
EKS_HD Mat<S, D> eye_plus_lt_y_lower(const Mat<S, D>& L, const Mat<S, D>& y) {
  Mat<S, D> o = mat_zero<S, D>();
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j)
      if (j <= i) {
        S acc = L.a[i][i] * y.a[i][j];
#pragma unroll
        for (int k = 0; k < D; ++k)
          if (k > i) acc = acc + L.a[k][i] * y.a[k][j];
        o.a[i][j] = i == j ? acc + S(1.0) : acc;
      }
  return o;
}
// X = B Lg^-T: row r of X solves Lg x = (row r of B)^T by forward substitution.  lower: B is lower-triangular.
template <typename S, int D, bool LOWER = false>
EKS_HD Mat<S, D> right_solve_lt(const CholF<S, D>& F, const Mat<S, D>& B) {
  Mat<S, D> X;
#pragma unroll
  for (int r = 0; r < D; ++r)
#pragma unroll
    for (int i = 0; i < D; ++i) {
      S t = (LOWER && i > r) ? S(0.0) : B.a[r][i];
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (k < i) t = t - F.L.a[i][k] * X.a[r][k];
      X.a[r][i] = t * F.invd.a[i];
    }
  return X;
}
// W W^T: every pair computed once, exactly symmetric
template <typename S, int D>
EKS_HD Mat<S, D> mat_aat(const Mat<S, D>& w) {
  Mat<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j)
      if (j >= i) {
        S acc = w.a[i][0] * w.a[j][0];
#pragma unroll
        for (int k = 1; k < D; ++k) acc = acc + w.a[i][k] * w.a[j][k];
        o.a[i][j] = acc;
        o.a[j][i] = acc;
      }
  return o;
}
// A^T X A + Y for symmetric X, Y (every pair computed once)
template <typename S, int D>
EKS_HD Mat<S, D> mat_sandwich_tn_plus(const Mat<S, D>& a, const Mat<S, D>& x, const Mat<S, D>& y) {
  const Mat<S, D> xa = mat_mul(x, a);
  Mat<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j)
      if (j >= i) {
        S acc = S(0.5) * (y.a[i][j] + y.a[j][i]);
#pragma unroll
        for (int k = 0; k < D; ++k) acc = acc + a.a[k][i] * xa.a[k][j];
        o.a[i][j] = acc;
        o.a[j][i] = acc;
      }
  return o;
}
// Lg^-1 (L^T x): L lower (zeros skipped), then forward substitution against the factor
template <typename S, int D>
EKS_HD Vec<S, D> lg_inv_lt_vec(const CholF<S, D>& F, const Mat<S, D>& L, const Vec<S, D>& x) {
  Vec<S, D> u;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    S t = L.a[i][i] * x.a[i];
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k > i) t = t + L.a[k][i] * x.a[k];
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k < i) t = t - F.L.a[i][k] * u.a[k];
    u.a[i] = t * F.invd.a[i];
  }
  return u;
}
// x - W (Z^T x)
template <typename S, int D>
EKS_HD Vec<S, D> vec_minus_w_zt(const Vec<S, D>& x, const Mat<S, D>& w, const Mat<S, D>& z) {
  const Vec<S, D> t = mat_t_vec(z, x);
  const Vec<S, D> wt = mat_vec(w, t);
  Vec<S, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i) o.a[i] = x.a[i] - wt.a[i];
  return o;
}

template <typename S, int D>
struct DElem {
  Mat<S, D> A, C, J;
  Vec<S, D> b, eta;
  S ell;
};

template <typename S, int D>
EKS_HD DElem<S, D> delem_identity() {
  DElem<S, D> e;
  e.A = mat_eye<S, D>();
  e.C = mat_zero<S, D>();
  e.J = mat_zero<S, D>();
  e.b = vec_zero<S, D>();
  e.eta = vec_zero<S, D>();
  e.ell = S(0.0);
  return e;
}

// Absorb one scalar observation y = h.x + N(0, r) into a running element (rank-1 forms; the
// matrix generalisation of eks_math.hpp's elem_append without the predict half).
template <typename S, int D>
EKS_HD void delem_observe(DElem<S, D>& e, const Vec<S, D>& h, S y, S r, bool want_ell) {
  const Vec<S, D> u = mat_vec(e.C, h);     // C h
  const Vec<S, D> w = mat_t_vec(e.A, h);   // A^T h
  const S sigma = r + dot(h, u);
  const S g = rcp(sigma);
  const S d = y - dot(h, e.b);
  const S gd = g * d;
  if (want_ell) e.ell = e.ell - S(0.5) * (S(kLog2Pi) + log_s(sigma) + d * gd);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    e.eta.a[i] = e.eta.a[i] + w.a[i] * gd;
    e.b.a[i] = e.b.a[i] + u.a[i] * gd;
  }
  // rank-1 updates with the gain folded into one factor (one FMA per entry) and the symmetric pairs of J and C
  // computed once: 21 + 2 D instructions for D = 3 instead of 81 - this runs once per scalar observation and frame
  Vec<S, D> wg, ug;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    wg.a[i] = w.a[i] * g;
    ug.a[i] = u.a[i] * g;
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) {
      e.A.a[i][j] = e.A.a[i][j] - ug.a[i] * w.a[j];
      if (j >= i) {
        const S jv = e.J.a[i][j] + wg.a[i] * w.a[j];        // (J, C are symmetric up to the rounding of a
        const S cv = e.C.a[i][j] - ug.a[i] * u.a[j];        //  predict step: the upper triangle speaks for both)
        e.J.a[i][j] = jv;
        e.J.a[j][i] = jv;
        e.C.a[i][j] = cv;
        e.C.a[j][i] = cv;
      }
    }
}

// Predict half of a frame: x' = F x + N(0, sQ).
template <typename S, int D>
EKS_HD void delem_predict(DElem<S, D>& e, const Mat<S, D>& F, const Mat<S, D>& sQ, bool f_identity) {
  if (!f_identity) {
    e.A = mat_mul(F, e.A);
    e.b = mat_vec(F, e.b);
    e.C = mat_mul_nt(mat_mul(F, e.C), F);
  }
  e.C = mat_add(e.C, sQ);
}

// Predict half of a frame for diagonal dynamics: x' = diag(a) x + N(0, diag(q)).
template <typename S, int D>
EKS_HD void delem_predict_diag(DElem<S, D>& e, const Vec<S, D>& a, const Vec<S, D>& q) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    e.b.a[i] = a.a[i] * e.b.a[i];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      e.A.a[i][j] = a.a[i] * e.A.a[i][j];
      e.C.a[i][j] = a.a[i] * a.a[j] * e.C.a[i][j];
    }
    e.C.a[i][i] = e.C.a[i][i] + q.a[i];
  }
}

// Absorb a whole frame given in information form, Lambda = H^T R^-1 H (symmetric D x D),
// nu = H^T R^-1 y and c = O log 2pi + sum log r + sum y^2 / r - all three independent of the
// model parameters, so plain doubles.  Lambda = Ll Ll^T (Cholesky, semi-definite safe) turns
// the frame into D unit-variance scalar pseudo-observations z = Ll^-1 nu with rows Ll^T:
// H'^T H' = Lambda, H'^T z = nu, and the log-likelihood differs from the true one by the
// constant -(c - D log 2pi - z.z)/2.  D rank-1 updates instead of O.
template <typename S, int D>
EKS_HD void delem_observe_info(DElem<S, D>& e, const Mat<double, D>& Lam, const Vec<double, D>& nu,
                               double c) {
  const Mat<double, D> Ll = chol_psd(Lam);
  Vec<double, D> z;
  double zz = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double t = nu.a[i];
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k < i) t -= Ll.a[i][k] * z.a[k];
    z.a[i] = Ll.a[i][i] > 0.0 ? t / Ll.a[i][i] : 0.0;
    zz += z.a[i] * z.a[i];
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    Vec<S, D> h;
#pragma unroll
    for (int j = 0; j < D; ++j) h.a[j] = S(j >= i ? Ll.a[j][i] : 0.0);
    delem_observe(e, h, S(z.a[i]), S(1.0), true);
  }
  e.ell = e.ell - S(0.5 * (c - D * kLog2Pi - zz));
}

// Posterior of the belief N(m, P) on x_in given the element's information (eta, J):
//   P_in = (P^-1 + J)^-1 = L (I + L^T J L)^-1 L^T,  P = L L^T   (stable: the inner matrix is >= I)
//   m_in = w - L (I + L^T J L)^-1 L^T J w,          w = m + P eta
// Also returns log|I + P J| (for the likelihood).
template <typename S, int D>
EKS_HD void condition_on_info(const Vec<S, D>& m, const Mat<S, D>& P, const Vec<S, D>& eta,
                              const Mat<S, D>& J, Vec<S, D>& m_in, Mat<S, D>& P_in, S& logdet) {
  const Mat<S, D> L = chol_psd(P);
  const CholF<S, D> Lg = chol_factor(eye_plus_lt_y_lower(L, mat_mul_lower(J, L)));
  logdet = S(0.0);
#pragma unroll
  for (int i = 0; i < D; ++i) logdet = logdet + S(2.0) * log_s(Lg.L.a[i][i]);
  Vec<S, D> w = mat_vec(P, eta);
#pragma unroll
  for (int i = 0; i < D; ++i) w.a[i] = w.a[i] + m.a[i];
  // W = L Lg^-T:  P_in = W W^T,  m_in = w - W Lg^-1 L^T J w
  const Mat<S, D> W = right_solve_lt<S, D, true>(Lg, L);
  const Vec<S, D> Wu = mat_vec(W, lg_inv_lt_vec(Lg, L, mat_vec(J, w)));
#pragma unroll
  for (int i = 0; i < D; ++i) m_in.a[i] = w.a[i] - Wu.a[i];
  P_in = mat_aat(W);
}

// Push N(m, P) through an element; returns the element's log marginal likelihood under it.
template <typename S, int D>
EKS_HD S delem_apply(const DElem<S, D>& e, Vec<S, D>& m, Mat<S, D>& P) {
  Vec<S, D> m_in;
  Mat<S, D> P_in;
  S logdet;
  condition_on_info(m, P, e.eta, e.J, m_in, P_in, logdet);
  const Vec<S, D> Jm = mat_vec(e.J, m);
  Vec<S, D> v;
#pragma unroll
  for (int i = 0; i < D; ++i) v.a[i] = e.eta.a[i] - Jm.a[i];
  const S ll = e.ell - S(0.5) * logdet + dot(m, e.eta) - S(0.5) * dot(m, Jm) +
               S(0.5) * dot(v, mat_vec(P_in, v));
  const Vec<S, D> Am = mat_vec(e.A, m_in);
#pragma unroll
  for (int i = 0; i < D; ++i) m.a[i] = Am.a[i] + e.b.a[i];
  P = mat_symmetrize(mat_add(mat_mul_nt(mat_mul(e.A, P_in), e.A), e.C));
  return ll;
}

// Pull information (eta, J) about x_out back through an element:
//   M^T = (I + J C)^-1 = I - J L (I + L^T J L)^-1 L^T,  C = L L^T
template <typename S, int D>
EKS_HD void delem_back(const DElem<S, D>& e, Vec<S, D>& eta, Mat<S, D>& J) {
  const Mat<S, D> L = chol_psd(e.C);
  const Mat<S, D> JL = mat_mul_lower(J, L);
  const CholF<S, D> Lg = chol_factor(eye_plus_lt_y_lower(L, JL));
  const Mat<S, D> Z = right_solve_lt<S, D>(Lg, JL);           // J L Lg^-T
  const Vec<S, D> Jb = mat_vec(J, e.b);
  Vec<S, D> v;
#pragma unroll
  for (int i = 0; i < D; ++i) v.a[i] = eta.a[i] - Jb.a[i];
  // M^T v = v - Z Lg^-1 L^T v
  const Vec<S, D> Zu = mat_vec(Z, lg_inv_lt_vec(Lg, L, v));
#pragma unroll
  for (int i = 0; i < D; ++i) v.a[i] = v.a[i] - Zu.a[i];
  const Vec<S, D> Atv = mat_t_vec(e.A, v);
#pragma unroll
  for (int i = 0; i < D; ++i) eta.a[i] = Atv.a[i] + e.eta.a[i];
  // J' = J - Z Z^T, then A^T J' A + J_e
  J = mat_sandwich_tn_plus(e.A, mat_sub(mat_symmetrize(J), mat_aat(Z)), e.J);
}

// Compose two elements, `i` (earlier frames) then `j` (later frames):
//   M = (I + C_i J_j)^-1 = I - L G^-1 L^T J_j,   C_i = L L^T,  G = I + L^T J_j L  (>= I: no pivoting)
//   A = A_j M A_i           b = A_j M (b_i + C_i eta_j) + b_j        C = A_j M C_i A_j^T + C_j
//   eta = A_i^T M^T (eta_j - J_j b_i) + eta_i                        J = A_i^T M^T J_j A_i + J_i
//   ell = ell_i + ell_j - log|G|/2 + b_i.eta_j - b_i^T J_j b_i / 2 + v^T (M C_i) v / 2,  v = eta_j - J_j b_i
//   (ELL = false: the smoother's scans do not need the log-likelihood term - three logs and two
//    quadratic forms less per composition)
template <typename S, int D, bool ELL = true>
EKS_HD DElem<S, D> delem_combine(const DElem<S, D>& ei, const DElem<S, D>& ej) {
  const Mat<S, D> L = chol_psd(ei.C);
  const Mat<S, D> JL = mat_mul_lower(ej.J, L);
  const CholF<S, D> Lg = chol_factor(eye_plus_lt_y_lower(L, JL));
  const Mat<S, D> W = right_solve_lt<S, D, true>(Lg, L);      // L Lg^-T
  const Mat<S, D> Z = right_solve_lt<S, D>(Lg, JL);           // J L Lg^-T
  DElem<S, D> o;
  // A = A_j (A_i - W Z^T A_i)
  o.A = mat_mul(ej.A, mat_sub(ei.A, mat_mul(W, mat_mul_tn(Z, ei.A))));
  // b = A_j M (b_i + C_i eta_j) + b_j
  Vec<S, D> w = mat_vec(ei.C, ej.eta);
#pragma unroll
  for (int i = 0; i < D; ++i) w.a[i] = w.a[i] + ei.b.a[i];
  const Vec<S, D> AMw = mat_vec(ej.A, vec_minus_w_zt(w, W, Z));
#pragma unroll
  for (int i = 0; i < D; ++i) o.b.a[i] = AMw.a[i] + ej.b.a[i];
  // C = A_j (W W^T) A_j^T + C_j = (A_j W)(A_j W)^T + C_j
  {
    const Mat<S, D> AW = mat_aat(mat_mul(ej.A, W));
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int k = 0; k < D; ++k)
        if (k >= i) {
          const S c = AW.a[i][k] + S(0.5) * (ej.C.a[i][k] + ej.C.a[k][i]);
          o.C.a[i][k] = c;
          o.C.a[k][i] = c;
        }
  }
  // eta = A_i^T M^T (eta_j - J_j b_i) + eta_i,   M^T = I - Z W^T
  const Vec<S, D> Jb = mat_vec(ej.J, ei.b);
  Vec<S, D> v;
#pragma unroll
  for (int i = 0; i < D; ++i) v.a[i] = ej.eta.a[i] - Jb.a[i];
  const Vec<S, D> Wtv = mat_t_vec(W, v);
  const Vec<S, D> ZWtv = mat_vec(Z, Wtv);
  Vec<S, D> Mtv;
#pragma unroll
  for (int i = 0; i < D; ++i) Mtv.a[i] = v.a[i] - ZWtv.a[i];
  const Vec<S, D> AtMtv = mat_t_vec(ei.A, Mtv);
#pragma unroll
  for (int i = 0; i < D; ++i) o.eta.a[i] = AtMtv.a[i] + ei.eta.a[i];
  // J = A_i^T (J_j - Z Z^T) A_i + J_i
  o.J = mat_sandwich_tn_plus(ei.A, mat_sub(mat_symmetrize(ej.J), mat_aat(Z)), ei.J);
  if constexpr (ELL) {
    S logdet = S(0.0);
#pragma unroll
    for (int i = 0; i < D; ++i) logdet = logdet + S(2.0) * log_s(Lg.L.a[i][i]);
    // v^T (M C_i) v = |W^T v|^2
    o.ell = ei.ell + ej.ell - S(0.5) * logdet + dot(ei.b, ej.eta) - S(0.5) * dot(ei.b, Jb) + S(0.5) * dot(Wtv, Wtv);
  } else {
    o.ell = S(0.0);
  }
  return o;
}

}  // namespace eks
