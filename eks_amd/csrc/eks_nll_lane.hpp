// Lane-level bodies of the constant-R filter log-likelihood on scalar chains: the loss the
// reference minimises over log s (eks/core.py:640-650: nll = -marginal_loglik of dynamax's
// extended_kalman_filter with R = diag(max(nanmedian_t var, 1e-4)), :602, :702-709).
//
// Time-parallel form: a lane owns (chain, time chunk, group of NCL candidate s values) and builds,
// per candidate, the chunk's element (A, b, C, eta, J) plus `ell`, the chunk's log-likelihood under
// x_in = 0 ("run-local filter").  A second, tiny kernel walks the chunks of a chain in order and
// assembles the exact marginal log-likelihood (nll_assemble below).
//
// With R constant the run-local variance C converges geometrically to the Riccati fixed point, so
// the chunk is processed in three wave-uniform regimes of decreasing cost:
//   0  full recursion (rcp + log per frame) until C is within 1e-6 of the closed-form fixed point,
//   1  C frozen: gains are constants; A (memory of x_in) still decays, eta/J still accumulate,
//   2  A < 1e-12: only the local mean b and the sum of squared innovations advance (3 flops/frame).
// d nll / d log s comes from running the same code on dual numbers (forward-mode AD; replaces
// jax.value_and_grad at eks/core.py:652).
#pragma once
#include <cmath>

#include "eks_diag_lane.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define EKS_WAVE_ALL(x) (__all(x) != 0)
#else
#define EKS_WAVE_ALL(x) (x)
#endif

namespace eks {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;

// ---- dual numbers: value and derivative w.r.t. u = log s ---------------------------------
template <typename F>
struct DualT {
  F v, d;
  EKS_HD DualT() : v(0), d(0) {}
  EKS_HD DualT(F x) : v(x), d(0) {}
  EKS_HD DualT(F x, F dx) : v(x), d(dx) {}
};
template <typename F>
EKS_HD DualT<F> operator+(DualT<F> a, DualT<F> b) { return {a.v + b.v, a.d + b.d}; }
template <typename F>
EKS_HD DualT<F> operator-(DualT<F> a, DualT<F> b) { return {a.v - b.v, a.d - b.d}; }
template <typename F>
EKS_HD DualT<F> operator*(DualT<F> a, DualT<F> b) { return {a.v * b.v, a.v * b.d + a.d * b.v}; }
template <typename F>
EKS_HD DualT<F> rcp(DualT<F> x) {
  const F r = rcp(x.v);
  return {r, -r * r * x.d};
}
using Dual = DualT<float>;
using DualD = DualT<double>;

EKS_HD float val(float x) { return x; }
EKS_HD float der(float) { return 0.f; }
EKS_HD double val(double x) { return x; }
EKS_HD double der(double) { return 0.0; }
template <typename F>
EKS_HD F val(DualT<F> x) { return x.v; }
template <typename F>
EKS_HD F der(DualT<F> x) { return x.d; }

EKS_HD float make_real(float, float v, float) { return v; }
EKS_HD Dual make_real(Dual, float v, float d) { return Dual(v, d); }
EKS_HD double make_real(double, double v, double) { return v; }
EKS_HD DualD make_real(DualD, double v, double d) { return DualD(v, d); }

EKS_HD float fast_log(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __logf(x);
#else
  return logf(x);
#endif
}
// log(S) given g = 1/S
EKS_HD float log_with_rcp(float S, float) { return fast_log(S); }
EKS_HD Dual log_with_rcp(Dual S, Dual g) { return Dual(fast_log(S.v), S.d * g.v); }
EKS_HD double log_with_rcp(double S, double) { return log(S); }
EKS_HD DualD log_with_rcp(DualD S, DualD g) { return DualD(log(S.v), S.d * g.v); }

// double-precision accumulator of a (possibly dual) float quantity
template <typename R>
struct Acc64;
template <>
struct Acc64<float> {
  double v = 0.0;
  EKS_HD void add(float x) { v += (double)x; }
};
template <>
struct Acc64<Dual> {
  double v = 0.0, d = 0.0;
  EKS_HD void add(Dual x) {
    v += (double)x.v;
    d += (double)x.d;
  }
};

// chunk summary for one (chain, candidate): element + run-local log-likelihood (and derivatives)
template <typename R>
struct NllElem {
  Elem<R> e;
  double ell, dell;
};

// Riccati fixed point of C' = a^2 C r / (r + c^2 C) + sq and its derivative w.r.t. log s.
EKS_HD void riccati_fixed_point(double a, double c, double r, double sq, double& Cinf, double& dCinf) {
  const double c2 = c * c;
  const double beta = r * (1.0 - a * a) - sq * c2;
  const double disc = sqrt(beta * beta + 4.0 * c2 * sq * r);
  // cancellation-free root: for beta > 0 use 2 sq r / (beta + disc)
  Cinf = beta > 0.0 ? (2.0 * sq * r) / (beta + disc) : (disc - beta) / (2.0 * c2);
  dCinf = sq * (r + c2 * Cinf) / (2.0 * c2 * Cinf + beta);
}

// One lane: chunk [t0, t0+len) of chain n for NCL candidates.  y: [T][N] float.
// sq[c] = s_c * q of the chain (value; its derivative w.r.t. log s is itself).
template <typename R, int NCL, bool UNIT>
EKS_HD void nll_summarize_chunk(const float* __restrict__ y, int N, int n, int t0, int len,
                                double r_d, double a_d, double c_d, const double* sq_d,
                                NllElem<R>* out) {
  const float r = (float)r_d;
  const R rR = R(r);
  ChainParams<R> pc[NCL];
  R CinfR[NCL], gI[NCL], rgI[NCL], tI[NCL], cgI[NCL], logSinf[NCL];
  Elem<R> e[NCL];
  Acc64<R> quad[NCL], logacc[NCL], acc2[NCL];
  const float af = (float)a_d, cf = (float)c_d;
#pragma unroll
  for (int k = 0; k < NCL; ++k) {
    pc[k].a = R(af);
    pc[k].c = R(cf);
    pc[k].q_s = make_real(R(), (float)sq_d[k], (float)sq_d[k]);
    double Ci, dCi;
    riccati_fixed_point(UNIT ? 1.0 : a_d, UNIT ? 1.0 : c_d, r_d, sq_d[k], Ci, dCi);
    CinfR[k] = make_real(R(), (float)Ci, (float)dCi);
    const R Sinf = UNIT ? (rR + CinfR[k]) : (rR + CinfR[k] * pc[k].c * pc[k].c);
    gI[k] = rcp(Sinf);
    rgI[k] = rR * gI[k];
    cgI[k] = UNIT ? gI[k] : pc[k].c * gI[k];
    tI[k] = CinfR[k] * cgI[k];
    logSinf[k] = log_with_rcp(Sinf, gI[k]);
    e[k] = elem_identity<R>();
  }
  int n_post = 0, i = 0;
  // ---- regime 0: full recursion until every candidate of every lane sits on its fixed point
  while (i < len) {
    const int nb = (len - i) < 8 ? (len - i) : 8;
    float yb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < nb) yb[q] = y[(size_t)(t0 + i + q) * N + n];
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NCL; ++k) {
      R qs = R(0.f), ls = R(0.f);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < nb) {
          R S, g, d;
          elem_append<R, UNIT>(e[k], R(yb[q]), rR, pc[k], S, g, d);
          qs = qs + d * d * g;
          ls = ls + log_with_rcp(S, g);
        }
      quad[k].add(qs);
      logacc[k].add(ls);
      const float tol = 1e-6f;
      ok = ok && fabsf(val(e[k].C) - val(CinfR[k])) <= tol * val(CinfR[k]) &&
           fabsf(der(e[k].C) - der(CinfR[k])) <= tol * fabsf(der(CinfR[k])) + 1e-30f;
    }
    i += nb;
    if (EKS_WAVE_ALL(ok)) {
#pragma unroll
      for (int k = 0; k < NCL; ++k) e[k].C = CinfR[k];
      break;
    }
  }
  // ---- regime 1: C frozen at the fixed point; A still decays, eta / J still accumulate
  while (i < len) {
    const int nb = (len - i) < 8 ? (len - i) : 8;
    float yb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < nb) yb[q] = y[(size_t)(t0 + i + q) * N + n];
    bool dead = true;
#pragma unroll
    for (int k = 0; k < NCL; ++k) {
      R s2 = R(0.f);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < nb) {
          const R d = UNIT ? (R(yb[q]) - e[k].b) : (R(yb[q]) - pc[k].c * e[k].b);
          s2 = s2 + d * d;
          const R Acg = e[k].A * cgI[k];
          e[k].eta = e[k].eta + Acg * d;
          e[k].J = e[k].J + (UNIT ? Acg * e[k].A : Acg * e[k].A * pc[k].c);
          e[k].b = UNIT ? (e[k].b + tI[k] * d) : pc[k].a * (e[k].b + tI[k] * d);
          e[k].A = UNIT ? e[k].A * rgI[k] : pc[k].a * e[k].A * rgI[k];
        }
      acc2[k].add(s2);
      dead = dead && fabsf(val(e[k].A)) < 1e-12f && fabsf(der(e[k].A)) < 1e-12f;
    }
    n_post += nb;
    i += nb;
    if (EKS_WAVE_ALL(dead)) {
#pragma unroll
      for (int k = 0; k < NCL; ++k) e[k].A = R(0.f);
      break;
    }
  }
  // ---- regime 2: only the run-local mean and the squared innovations advance
  while (i < len) {
    const int nb = (len - i) < 8 ? (len - i) : 8;
    float yb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q < nb) yb[q] = y[(size_t)(t0 + i + q) * N + n];
#pragma unroll
    for (int k = 0; k < NCL; ++k) {
      R s2 = R(0.f);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < nb) {
          const R d = UNIT ? (R(yb[q]) - e[k].b) : (R(yb[q]) - pc[k].c * e[k].b);
          s2 = s2 + d * d;
          e[k].b = UNIT ? (e[k].b + tI[k] * d) : pc[k].a * (e[k].b + tI[k] * d);
        }
      acc2[k].add(s2);
    }
    n_post += nb;
    i += nb;
  }
#pragma unroll
  for (int k = 0; k < NCL; ++k) {
    out[k].e = e[k];
    // ell = -0.5 * (len log 2pi + sum log S + sum d^2 / S)
    double q_v = quad[k].v + (double)val(gI[k]) * acc2[k].v;
    double l_v = logacc[k].v + (double)n_post * (double)val(logSinf[k]);
    out[k].ell = -0.5 * ((double)len * kLog2Pi + l_v + q_v);
    double q_d = 0.0, l_d = 0.0;
    if constexpr (sizeof(R) == sizeof(Dual)) {
      q_d = quad[k].d + (double)der(gI[k]) * acc2[k].v + (double)val(gI[k]) * acc2[k].d;
      l_d = logacc[k].d + (double)n_post * (double)der(logSinf[k]);
    }
    out[k].dell = -0.5 * (l_d + q_d);
  }
}

// Assemble the marginal log-likelihood of one chain for one candidate from its chunk summaries:
//   ll = sum_j [ ell_j - 0.5 log(1 + J_j P) + (eta_j m + 0.5 eta_j^2 P - 0.5 J_j m^2)/(1 + J_j P) ]
// with (m, P) the predicted belief entering chunk j (pushed through the elements in order).
// RD is double or DualD; `get(j)` returns the chunk's element as Elem<RD> and its (ell, dell).
template <typename RD, typename Getter>
EKS_HD RD nll_assemble(int nchunks, double m0, double S0, Getter get) {
  RD m = RD(m0), P = RD(S0);
  RD ll = RD(0.0);
  for (int j = 0; j < nchunks; ++j) {
    Elem<RD> e;
    RD ell;
    get(j, e, ell);
    const RD den = RD(1.0) + e.J * P;
    const RD inv = rcp(den);
    ll = ll + ell - RD(0.5) * log_with_rcp(den, inv) +
         (e.eta * m + RD(0.5) * e.eta * e.eta * P - RD(0.5) * e.J * m * m) * inv;
    const RD AI = e.A * inv;
    const RD m_n = AI * (m + P * e.eta) + e.b;
    P = AI * e.A * P + e.C;
    m = m_n;
  }
  return ll;
}

}  // namespace eks
