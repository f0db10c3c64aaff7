// Lane-level bodies of the constant-R filter log-likelihood on scalar chains: the loss the
// reference minimises over log s (eks/core.py:640-650: nll = -marginal_loglik of dynamax's
// extended_kalman_filter with R = diag(max(nanmedian_t var, 1e-4)), :602, :702-709).
//
// Time-parallel form: a lane owns (chain, time chunk, group of NCL candidate s values) and builds,
// per candidate, the chunk's summary: element (A, b, C, eta, J) plus `ell`, the chunk's
// log-likelihood for a reference entry state ("run-local filter").  A second, tiny kernel walks
// the chunks of a chain in order and assembles the exact marginal log-likelihood (nll_assemble).
//
// With R constant the run-local variance C converges geometrically to the Riccati fixed point, so
// an exact-entry chunk (known x_in; always chunk 0) is processed in three wave-uniform regimes of
// decreasing cost:
//   0  full recursion (rcp + log per frame) until C is within ~1e-6 (plus the float32 stall
//      distance ulp / (1 - rho)) of the closed-form fixed point, then C is snapped onto it,
//   1  C frozen: gains are constants; A (memory of x_in) still decays, eta/J still accumulate,
//   2  A dead: only the innovation d' = rho d + (y' - a y) and the sum of its squares advance:
//      two FMAs per frame and candidate, fused over the lane's candidates.
// A chunk deep enough into the sequence skips the regimes altogether (converged entry, see
// nll_summarize_chunk).  All regimes advance the innovation, never the run-local mean, and every
// summary is relative to the chunk's reference state y_0 / c (float32 would otherwise lose the
// NLL's low digits to terms that grow like y^2 and cancel in the assembly).
// d nll / d log s comes from running the same code on dual numbers (forward-mode AD; replaces
// jax.value_and_grad at eks/core.py:652).
#pragma once
#include <cmath>

#include "eks_diag_lane.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define EKS_WAVE_ALL(x) (__all(x) != 0)
// no instruction moves across this point: keeps the machine scheduler from interleaving the frames of an unrolled
// block (it otherwise holds 16 frames' worth of innovations live to cluster the FMAs, and spills)
#define EKS_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// ties values together at a point of the program (no instruction is emitted; they are opaque to the optimiser
// from here on).  Without it the optimiser takes the frame loop of the alive phase apart: the data-independent
// recurrence rho^t is run a whole unrolled block ahead, the accumulations are sunk to the block that flushes
// them, and sixteen frames' worth of intermediates go through scratch memory
#define EKS_OPAQUE4(x, y, z, u) asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(u))
#define EKS_OPAQUE2(x, y) asm volatile("" : "+v"(x), "+v"(y))
#else
#define EKS_OPAQUE4(x, y, z, u) do { } while (0)
#define EKS_OPAQUE2(x, y) do { } while (0)
#define EKS_WAVE_ALL(x) (x)
#define EKS_SCHED_FENCE() do { } while (0)
#endif

namespace eks {

// two float32 lanes in one 64-bit register pair: arithmetic on it compiles to gfx950's packed
// VALU (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32), i.e. two candidate filters per instruction.
// Round 3: the steady loops keep the candidates PAIRED in such registers from start to end (no shuffles in
// the loop).  A v_pk_fma_f32 occupies the SIMD for the cycles of two v_fma_f32, but ONE wave can issue it
// back to back where it can issue a plain VALU instruction only every ~4 cycles: at the kernel's two waves
// per SIMD the loop's dependency structure runs 2.60 instead of 3.23 cycles per candidate-FMA
// (tools/micro/pk_fma_rate.hip; round 1's attempt let the SLP vectoriser pack and paid for its v_mov
// shuffles).  Component-wise the arithmetic is the same fused multiply-add: results are bit-identical.
#ifndef EKS_NLL_PACKED
#if defined(__HIP_DEVICE_COMPILE__)
#define EKS_NLL_PACKED 1
#else
#define EKS_NLL_PACKED 0
#endif
#endif
#if defined(__clang__)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#else
typedef float f32x2 __attribute__((vector_size(8)));
#endif

template <int N>
struct IntTag {
  static constexpr int value = N;
};

constexpr double kLog2Pi = 1.8378770664093454835606594728112;
// poles above this run the recursion in the complement form d' = d + (u - (1 - rho) d): see LeanConst below
constexpr float kKappaRho = 0.98f;

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

EKS_HD DualT<float> make_dual(float x) { return DualT<float>(x, 0.f); }
EKS_HD DualT<float> make_dual(DualT<float> x) { return x; }
EKS_HD float from_dual(float, DualT<float> x) { return x.v; }
EKS_HD DualT<float> from_dual(DualT<float>, DualT<float> x) { return x; }
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
  EKS_HD float rounded() const { return (float)v; }
};
template <>
struct Acc64<Dual> {
  double v = 0.0, d = 0.0;
  EKS_HD void add(Dual x) {
    v += (double)x.v;
    d += (double)x.d;
  }
  EKS_HD Dual rounded() const { return Dual((float)v, (float)d); }
};

// chunk summary for one (chain, candidate): element + run-local log-likelihood (and derivatives)
template <typename R>
struct NllElem {
  Elem<R> e;
  double ell, dell;
  float xref;   // reference state of the chunk: the summary is a function of (x_in - xref)
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

// State of one lane: NCL candidate filters of one chain over one chunk.
template <typename R, int NCL, bool UNIT>
struct NllLane {
  ChainParams<R> pc[NCL];
  R CinfR[NCL], gI[NCL], rgI[NCL], cgI[NCL], logSinf[NCL];
  R rhoI[NCL];        // steady-state pole a r g = a (1 - c t) of the innovation recursion when a != 1
                      // (with a = c = 1 it IS r g: pole() reads rgI and this array is never live)
  R kapI[NCL];        // 1 - pole, rounded once from float64; used where slow_pole (kKappaRho)
  bool slow_pole;     // some candidate of the WAVE has a pole above kKappaRho: complement form of the recursion
  EKS_HD R pole(int k) const { return UNIT ? rgI[k] : rhoI[k]; }
  // d' = pole d + u, in the form that keeps the pole's distance from one to float32's RELATIVE precision when it is small
  EKS_HD R advance(int k, R d, R u) const { return slow_pole ? d + (u - kapI[k] * d) : pole(k) * d + u; }
  Elem<R> e[NCL];
  R dl[NCL];          // innovation of the last consumed frame
  R rg_last[NCL];     // r g = 1 - c K of the last consumed frame
  R kap_last[NCL];    // 1 - a (1 - c K) of the last consumed frame, formed without cancellation (regime 0 only)
  float oma;          // 1 - a, from float64
  float y_last;       // last consumed observation
  bool any_frame;
  // y' - a y is formed with a in float64: rounding a to float32 (6e-8) shifts every prediction
  // by 6e-8 |y|, a perturbation of the MODEL that slow candidates integrate (measured 1e-5 on the
  // NLL at |y| ~ 1000); one float64 FMA per frame and lane, shared by the lane's candidates
  double a_dbl;
  Acc64<R> quad[NCL], logacc[NCL], acc2[NCL];
  // eta and J of the summary: float32 partial sums of one 8-frame block, added up in float64.  The summary describes
  // the chunk for an entering state known EXACTLY (x_in = xref): with a slow candidate on a short, fast-moving
  // sequence its ell is far below the sequence's log-likelihood and eta^2 / 2J, applied with the prior, brings it
  // back - each of them 10x the result (T = 64, s = 5e-4: 5e4 against 8e3), and a running float32 sum of eta put
  // 1.3e-5 on that NLL (fuzz seed 911, case 64; round 5).
  Acc64<R> eta64[NCL], J64[NCL];
  int phase[NCL], n_post[NCL];
  float tolC[NCL];
  R rR;
  // |A| (= rho^t) below this no longer matters: A only ever multiplies x_in - xref, a few pixels
  // now that summaries are relative to the chunk's reference state, so what is dropped is
  // ~1e-5 px in the mean and ~1e-9 of the NLL (with the reference at the origin it had to be 1e-8)
  static constexpr float kDeadA = sizeof(R) == sizeof(float) ? 1e-5f : 1e-9f;

  // Transient code: consume NB (<= 8) frames with per-candidate regimes (wave-uniform).
  //
  // All regimes advance the INNOVATION, d' = (y' - a y) + a (1 - c K) d, never the run-local mean:
  // with observations hundreds of pixels from the origin and innovations of order one, forming
  // y - c b from a float32 b loses the innovation's low bits at every frame (measured 1.6e-5 on
  // the NLL of slow candidates at |y| ~ 600), whereas y' - a y is exact or nearly so.  The mean
  // is recovered once, when the chunk ends: c b = y - d  ->  b_next = a ((y - d) / c + K d).
  template <int NB>
  EKS_HD void consume(const float (&yb)[8]) {
#pragma unroll
    for (int k = 0; k < NCL; ++k) {
      float yp = y_last;
      if (phase[k] == 2) {
        // ---- regime 2: only the squared innovations advance
        R s2 = R(0.f), d = dl[k];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const float dy = UNIT ? (yb[q] - yp) : (float)((double)yb[q] - a_dbl * (double)yp);
          yp = yb[q];
          d = advance(k, d, R(dy));
          s2 = s2 + d * d;
        }
        dl[k] = d;
        acc2[k].add(s2);
        n_post[k] += NB;
      } else if (phase[k] == 1 && EKS_NLL_PACKED && sizeof(R) == sizeof(Dual) && NCL == 1) {
        // ---- regime 1 of the gradient path on packed (value, derivative) pairs (round 5: the dual-number operators
        //      below compile to ~35 issue slots per frame, this to 13; chunk 0 of the Adam loss kernel spends ~130
        //      frames here mid-search and was the wave every iteration waited for)
        const Dual pl = make_dual(slow_pole ? R(0.f) - kapI[k] : pole(k)), cgd = make_dual(cgI[k]), rgd = make_dual(rgI[k]);
        const Dual d0 = make_dual(dl[k]), A0 = make_dual(e[k].A);
        const f32x2 P2 = f32x2{pl.v, pl.v}, CG = f32x2{cgd.v, cgd.d}, RG = f32x2{rgd.v, rgd.d};
        const float cN = UNIT ? 1.f : val(pc[k].c), aN = UNIT ? 1.f : val(pc[k].a);
        f32x2 X = f32x2{d0.v, d0.d}, Ap = f32x2{A0.v, A0.d};
        f32x2 S2 = f32x2{0.f, 0.f}, ES = f32x2{0.f, 0.f}, JS = f32x2{0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const float dy = UNIT ? (yb[q] - yp) : (float)((double)yb[q] - a_dbl * (double)yp);
          yp = yb[q];
          const f32x2 step = P2 * X + f32x2{dy, pl.d * X[0]};
          X = slow_pole ? X + step : step;
          S2 = f32x2{X[0], X[0]} * X + S2;
          const f32x2 Acg = f32x2{Ap[0], Ap[0]} * CG + f32x2{0.f, Ap[1] * cgd.v};
          ES = f32x2{Acg[0], Acg[0]} * X + ES;
          ES[1] += Acg[1] * X[0];
          f32x2 AA = f32x2{Acg[0], Acg[0]} * Ap + f32x2{0.f, Acg[1] * Ap[0]};
          if (!UNIT) AA = AA * f32x2{cN, cN};
          JS = JS + AA;
          Ap = f32x2{Ap[0], Ap[0]} * RG + f32x2{0.f, Ap[1] * rgd.v};
          if (!UNIT) Ap = Ap * f32x2{aN, aN};
        }
        dl[k] = from_dual(R(), Dual(X[0], X[1]));
        e[k].A = from_dual(R(), Dual(Ap[0], Ap[1]));
        acc2[k].add(from_dual(R(), Dual(S2[0], 2.f * S2[1])));
        eta64[k].add(from_dual(R(), Dual(ES[0], ES[1])));
        J64[k].add(from_dual(R(), Dual(JS[0], JS[1])));
        n_post[k] += NB;
        const bool dead = fabsf(val(e[k].A)) < kDeadA && fabsf(der(e[k].A)) < kDeadA;
        if (EKS_WAVE_ALL(dead)) {
          phase[k] = 2;
          e[k].A = R(0.f);
        }
      } else if (phase[k] == 1) {
        // ---- regime 1: C frozen; A still decays, eta / J still accumulate
        R s2 = R(0.f), d = dl[k], es = R(0.f), js = R(0.f);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const float dy = UNIT ? (yb[q] - yp) : (float)((double)yb[q] - a_dbl * (double)yp);
          yp = yb[q];
          d = advance(k, d, R(dy));
          s2 = s2 + d * d;
          const R Acg = e[k].A * cgI[k];
          es = es + Acg * d;
          js = js + (UNIT ? Acg * e[k].A : Acg * e[k].A * pc[k].c);
          e[k].A = UNIT ? e[k].A * rgI[k] : pc[k].a * e[k].A * rgI[k];
        }
        dl[k] = d;
        acc2[k].add(s2);
        eta64[k].add(es);
        J64[k].add(js);
        n_post[k] += NB;
        const bool dead = fabsf(val(e[k].A)) < kDeadA && fabsf(der(e[k].A)) < kDeadA;
        if (EKS_WAVE_ALL(dead)) {
          phase[k] = 2;
          e[k].A = R(0.f);
        }
      } else {
        // ---- regime 0: full recursion until C sits on its fixed point
        const R c = UNIT ? R(1.f) : pc[k].c, a = UNIT ? R(1.f) : pc[k].a;
        // The variance advances as its DEVIATION from the fixed point: C' - C_inf = (C - C_inf) a^2 (r g) (r g_inf)
        // exactly (the Riccati map minus itself at C_inf), a product of factors below one - it decays with float32's
        // RELATIVE precision, where C' = a^2 C r g + s q stalls ~ulp / (1 - rho^2) away from C_inf: for poles at 0.999
        // that was 1e-4 of C_inf, the snap below had to be that coarse, and its decaying transient cost 3e-6 to 5e-6 on
        // the NLL of the slowest candidates at variances in the hundreds (round 5).
        // The innovation is carried over in the complement form d' = d + (u - kappa d), kappa = 1 - a (1 - c K) =
        // (1 - a) + a c^2 C g of the PREVIOUS frame, a sum of positive terms: as the gain settles, a pole a r g rounded
        // to float32 is a BIASED pole (up to 3e-8 / (1 - rho) of the gain), and with the variance now tracked to 1e-6
        // this regime lasts ~7 / (1 - rho) frames.
        R qs = R(0.f), ls = R(0.f), d = dl[k], rg = rg_last[k], kp = kap_last[k], es = R(0.f), js = R(0.f);
        R dC = e[k].C - CinfR[k];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const float dy = UNIT ? (yb[q] - yp) : (float)((double)yb[q] - a_dbl * (double)yp);
          yp = yb[q];
          d = d + (R(dy) - kp * d);
          const R Cq = CinfR[k] + dC;
          const R S = UNIT ? (rR + Cq) : (rR + Cq * c * c);
          const R g = rcp(S);
          rg = rR * g;                                 // 1 - c K of this frame
          const R Acg = UNIT ? e[k].A * g : e[k].A * c * g;
          es = es + Acg * d;
          js = js + (UNIT ? Acg * e[k].A : Acg * e[k].A * c);
          e[k].A = UNIT ? e[k].A * rg : a * e[k].A * rg;
          dC = UNIT ? (dC * rg) * rgI[k] : (a * a * dC * rg) * rgI[k];
          kp = UNIT ? Cq * g : R(oma) + a * c * c * Cq * g;
          qs = qs + d * d * g;
          ls = ls + log_with_rcp(S, g);
        }
        e[k].C = CinfR[k] + dC;
        dl[k] = d;
        rg_last[k] = rg;
        kap_last[k] = kp;
        quad[k].add(qs);
        logacc[k].add(ls);
        eta64[k].add(es);
        J64[k].add(js);
        const bool ok = fabsf(val(dC)) <= tolC[k] * val(CinfR[k]) &&
                        fabsf(der(dC)) <= 4.f * tolC[k] * fabsf(der(CinfR[k])) + 1e-30f;
        if (EKS_WAVE_ALL(ok)) {
          phase[k] = 1;
          e[k].C = CinfR[k];
          rg_last[k] = rgI[k];
        }
      }
    }
    y_last = yb[NB - 1];
    any_frame = true;
  }

  // the run-local mean after the last consumed frame (see consume): b = a ((y - d) / c + K d),
  // K = C c g = (1 - r g) / c of that frame
  EKS_HD void recover_mean() {
#pragma unroll
    for (int k = 0; k < NCL; ++k) {
      e[k].eta = eta64[k].rounded();
      e[k].J = J64[k].rounded();
    }
    if (!any_frame) return;
#pragma unroll
    for (int k = 0; k < NCL; ++k) {
      if (UNIT) {
        e[k].b = R(y_last) - rg_last[k] * dl[k];        // y - d + (1 - rg) d
      } else {
        const R ic = rcp(pc[k].c);
        e[k].b = pc[k].a * ((R(y_last) - dl[k]) * ic + (R(1.f) - rg_last[k]) * ic * dl[k]);
      }
    }
  }

  EKS_HD bool all_steady() const {
    bool all2 = true;
#pragma unroll
    for (int k = 0; k < NCL; ++k) all2 = all2 && phase[k] == 2;
    return all2;
  }
};

template <typename R, int NCL, bool UNIT>
EKS_HD void nll_lane_init(NllLane<R, NCL, UNIT>& L, double r_d, double a_d, double c_d,
                          const double* sq_d) {
  const float r = (float)r_d;
  L.rR = R(r);
  L.y_last = 0.f;
  L.any_frame = false;
  L.a_dbl = a_d;
  L.oma = (float)(1.0 - a_d);
  const float af = (float)a_d, cf = (float)c_d;
  bool slow = false;
#pragma unroll
  for (int k = 0; k < NCL; ++k) {
    L.pc[k].a = R(af);
    L.pc[k].oma = R((float)(1.0 - a_d));
    L.pc[k].oma2 = R((float)(1.0 - a_d * a_d));
    L.pc[k].c = R(cf);
    L.pc[k].q_s = make_real(R(), (float)sq_d[k], (float)sq_d[k]);
    double Ci, dCi;
    riccati_fixed_point(UNIT ? 1.0 : a_d, UNIT ? 1.0 : c_d, r_d, sq_d[k], Ci, dCi);
    L.CinfR[k] = make_real(R(), (float)Ci, (float)dCi);
    // The steady-state constants in float64 (values and d / d log s), each rounded ONCE.  Formed in
    // float32 the pole rho = a (1 - c t) picked up 2-3 roundings at magnitude ~1, i.e. an absolute
    // error of ~1e-7 against a distance 1 - rho of ~1e-2 for the slowest candidates (s near exp(-8),
    // R of a few px^2): the filter then runs with a gain that is off by 1e-5 relative, and so is the
    // NLL (fuzz seed 77: 1.1e-5 on a general diagonal model).  One rounding leaves the 3e-8 / (1 - rho)
    // that a float32 pole cannot avoid.
    {
      const double a1 = UNIT ? 1.0 : a_d, c1 = UNIT ? 1.0 : c_d;
      const double S_d = r_d + Ci * c1 * c1, dS_d = dCi * c1 * c1;
      const double g_d = 1.0 / S_d, dg_d = -g_d * g_d * dS_d;
      const double cg_d = c1 * g_d, dcg_d = c1 * dg_d;
      const double t_d = Ci * cg_d, dt_d = dCi * cg_d + Ci * dcg_d;
      L.gI[k] = make_real(R(), (float)g_d, (float)dg_d);
      L.rgI[k] = make_real(R(), (float)(r_d * g_d), (float)(r_d * dg_d));
      L.cgI[k] = make_real(R(), (float)cg_d, (float)dcg_d);
      L.logSinf[k] = make_real(R(), (float)log(S_d), (float)(dS_d * g_d));
      L.rhoI[k] = make_real(R(), (float)(a1 * (1.0 - c1 * t_d)), (float)(-a1 * c1 * dt_d));
      // 1 - pole: with a = c = 1 it is C g = t (no cancellation), otherwise 1 - a + a c t; d / d log s = -d pole
      const double kap_d = UNIT ? t_d : 1.0 - a1 * (1.0 - c1 * t_d);
      L.kapI[k] = make_real(R(), (float)kap_d, (float)(a1 * c1 * dt_d));
      slow = slow || fabs(a1 * (1.0 - c1 * t_d)) > (double)kKappaRho;
    }
    L.e[k] = elem_identity<R>();
    L.dl[k] = R(0.f);
    L.rg_last[k] = R(0.f);
    L.kap_last[k] = R(1.f);            // (the first innovation is u itself: d starts at zero)
    L.phase[k] = 0;
    L.n_post[k] = 0;
    // the float32 recursion stalls within ~ulp / (1 - rho) of the true fixed point, rho = (a r g)^2
    const float rho = UNIT ? val(L.rgI[k]) * val(L.rgI[k]) : af * af * val(L.rgI[k]) * val(L.rgI[k]);
    // Snapping C onto the fixed point when it is `tol` away perturbs the NLL by a decaying
    // transient (measured against the oracle: 2e-6 relative in total at 1e-4, 1e-5 at 1e-3); the
    // gradient path keeps the strict threshold.
    // (round 5: the variance's deviation decays exactly - see consume - so no allowance for a stall distance any more)
    (void)rho;
    L.tolC[k] = 1e-6f;
  }
  L.slow_pole = !EKS_WAVE_ALL(!slow);                    // (wave-uniform)
}

template <typename R, int NCL, bool UNIT>
EKS_HD void nll_lane_finish(NllLane<R, NCL, UNIT>& L, int len, float xref, NllElem<R>* out) {
  L.recover_mean();
#pragma unroll
  for (int k = 0; k < NCL; ++k) {
    out[k].e = L.e[k];
    out[k].xref = xref;
    // ell = -0.5 * (len log 2pi + sum log S + sum d^2 / S)
    double q_v = L.quad[k].v + (double)val(L.gI[k]) * L.acc2[k].v;
    double l_v = L.logacc[k].v + (double)L.n_post[k] * (double)val(L.logSinf[k]);
    out[k].ell = -0.5 * ((double)len * kLog2Pi + l_v + q_v);
    double q_d = 0.0, l_d = 0.0;
    if constexpr (sizeof(R) == sizeof(Dual)) {
      q_d = L.quad[k].d + (double)der(L.gI[k]) * L.acc2[k].v + (double)val(L.gI[k]) * L.acc2[k].d;
      l_d = L.logacc[k].d + (double)L.n_post[k] * (double)der(L.logSinf[k]);
    }
    out[k].dell = -0.5 * (l_d + q_d);
  }
}

// One lane: chunk [t0, t0+len) of chain n for NCL candidates.  y: [T][N] float.
// sq[c] = s_c * q of the chain (value; its derivative w.r.t. log s is itself).
//
// allow_converged_entry: for a chunk that starts deep enough into the sequence (t0 frames in) the
// TRUE filter's predicted variance has long reached the Riccati fixed point P_inf - with constant
// R it does not depend on the data - so the chunk can be summarised for an entering belief
// N(m_in, P_inf) instead of a known x_in: the gains are the steady ones from the first frame, the
// innovations are d_t = d0_t - c m_in rho^t with d0 the zero-start sequence, and
//   sum d_t^2 = S0 - 2 c m_in S1 + c^2 m_in^2 S2,   S1 = sum d0_t rho^t,  S2 = sum rho^2t.
// No transient regimes at all: four FMAs per frame and candidate while rho^t is alive, two after.
// Such a summary is marked by C = -1 and carries eta = c S1 / S_inf, J = c^2 S2 / S_inf; it is
// only valid in a strictly sequential assembly (nll_assemble), where P is P_inf when it is used.
// Taken only when every lane of the wave has rho^(2 t0) < 1e-20 and rho^t dies inside the chunk.
// `ld(i)` returns the observation of the lane's chain at frame t0 + i (RowsByPointer below; the
// gfx950 kernels pass a buffer-load functor whose row address is scalar arithmetic).
struct RowsByPointer {
  const float* p;      // y + t0 * N + n
  size_t rs;           // N
  EKS_HD float operator()(int i) const { return p[(size_t)i * rs]; }
};

template <typename R, int NCL, bool UNIT, typename LD>
EKS_HD void nll_summarize_chunk(const LD& ld, int t0, int len, double r_d, double a_d, double c_d,
                                const double* sq_d, NllElem<R>* out,
                                bool allow_converged_entry = false, int sequence_len = -1) {
  NllLane<R, NCL, UNIT> L;
  nll_lane_init<R, NCL, UNIT>(L, r_d, a_d, c_d, sq_d);
  // Snapping C onto the fixed point costs a one-off ~1e-5 of a frame's terms: nothing against a
  // chunk of thousands of frames, the whole error budget of a sequence of three.  Short chunks
  // stay in the full recursion (a caller that cuts a long sequence into short chunks says how long it is).
  if ((sequence_len < 0 ? len : sequence_len) < 256) {
#pragma unroll
    for (int k = 0; k < NCL; ++k) L.tolC[k] = -1.f;
  }
  const float af = (float)a_d;
  // Reference state: the chunk is summarised as a function of x_in - xref with xref = y_0 / c,
  // the state its own first observation points at.  With xref = 0 the zero-start innovations
  // open at |y| (hundreds of pixels), the summary's ell / eta / J grow like y^2 and only cancel
  // in the assembly - in float32 that cost 1.6e-5 on the NLL at |y| ~ 600.  The innovation
  // recursion itself does not know about the reference: starting it from "previous observation
  // = y_0 / a" makes the first innovation y_0 - c xref = 0.
  const float y0 = len > 0 ? ld(0) : 0.f;
  const float xref = UNIT ? y0 : y0 / (float)c_d;
  L.y_last = UNIT ? y0 : y0 / af;
  const int nfull = len / 8;
  int blk = 0;
  bool steady = false;
  if constexpr (sizeof(R) == sizeof(float)) {
    bool ok = allow_converged_entry && t0 > 0;
    float rho[NCL];
#pragma unroll
    for (int k = 0; k < NCL; ++k) {
      rho[k] = L.pole(k);
      const float nl = -logf(fmaxf(fabsf(rho[k]), 1e-30f));         // -ln |rho| > 0
      // rho^(2 t0) < 1e-20, and rho^t < kDeadA (ln 1e5 = 11.5) well inside the chunk's whole blocks
      ok = ok && fabsf(rho[k]) < 1.f && 2.f * (float)t0 * nl > 46.f &&
           11.6f / nl + 48.f < (float)(nfull * 8);
    }
    if (EKS_WAVE_ALL(ok)) {
      constexpr bool PK = EKS_NLL_PACKED && (NCL % 2 == 0);
      constexpr int NP = PK ? NCL / 2 : 1;
      float dk[NCL], w[NCL];
      double s1acc[NCL];
#pragma unroll
      for (int k = 0; k < NCL; ++k) {
        dk[k] = 0.f;
        w[k] = 1.f;
        s1acc[k] = 0.0;
      }
      float yprev = L.y_last;       // reference start: the first innovation is y_0 - c xref = 0
      float s1[NCL], s2[NCL];
#pragma unroll
      for (int k = 0; k < NCL; ++k) s1[k] = s2[k] = 0.f;
      // packed form: candidate pairs (2p, 2p + 1) live in 64-bit registers for the whole loop
      f32x2 rho2[NP], dk2[NP], w2[NP], s12[NP], s22[NP];
      if constexpr (PK) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          rho2[p] = f32x2{rho[2 * p], rho[2 * p + 1]};
          dk2[p] = f32x2{0.f, 0.f};
          w2[p] = f32x2{1.f, 1.f};
          s12[p] = f32x2{0.f, 0.f};
          s22[p] = f32x2{0.f, 0.f};
        }
      }
      auto eat4 = [&](const float (&yy)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float dy = UNIT ? (yy[q] - yprev) : (float)((double)yy[q] - a_d * (double)yprev);
          yprev = yy[q];
          if constexpr (PK) {
            const f32x2 dy2 = f32x2{dy, dy};
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              dk2[p] = rho2[p] * dk2[p] + dy2;
              s22[p] = s22[p] + dk2[p] * dk2[p];
              s12[p] = s12[p] + dk2[p] * w2[p];
              w2[p] = w2[p] * rho2[p];
            }
          } else {
#pragma unroll
            for (int k = 0; k < NCL; ++k) {
              dk[k] = rho[k] * dk[k] + dy;
              s2[k] += dk[k] * dk[k];
              s1[k] += dk[k] * w[k];
              w[k] *= rho[k];
            }
          }
        }
      };
      // pairs of 8-frame blocks, the second in flight while the first is consumed; partial sums
      // go to float64 and the rho^t trackers are examined every 32 frames
      float ya[8], yb[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) ya[q] = ld(q);
      bool alive = true;
      while (alive && blk + 2 <= nfull) {
#pragma unroll
        for (int q = 0; q < 8; ++q) yb[q] = ld(((blk + 1) * 8 + q));
        eat4(ya);
        if (blk + 2 < nfull) {
#pragma unroll
          for (int q = 0; q < 8; ++q) ya[q] = ld(((blk + 2) * 8 + q));
        }
        eat4(yb);
        blk += 2;
        if ((blk & 2) == 0 || blk + 2 > nfull) {
          bool dead = true;
          if constexpr (PK) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              s2[2 * p] = s22[p].x; s2[2 * p + 1] = s22[p].y;
              s1[2 * p] = s12[p].x; s1[2 * p + 1] = s12[p].y;
              w[2 * p] = w2[p].x; w[2 * p + 1] = w2[p].y;
              s22[p] = f32x2{0.f, 0.f};
              s12[p] = f32x2{0.f, 0.f};
            }
          }
#pragma unroll
          for (int k = 0; k < NCL; ++k) {
            L.acc2[k].add(s2[k]);
            s1acc[k] += (double)s1[k];
            s1[k] = s2[k] = 0.f;
            dead = dead && fabsf(w[k]) < L.kDeadA;
          }
          alive = !EKS_WAVE_ALL(dead);
        }
      }
      if constexpr (PK) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          dk[2 * p] = dk2[p].x;
          dk[2 * p + 1] = dk2[p].y;
        }
      }
#pragma unroll
      for (int k = 0; k < NCL; ++k) {
        L.phase[k] = 2;
        L.n_post[k] = blk * 8;
        L.dl[k] = dk[k];
        L.rg_last[k] = L.rgI[k];
        L.e[k].A = 0.f;
        L.e[k].C = -1.f;                                           // converged-entry marker
        L.eta64[k].v = s1acc[k] * (double)L.cgI[k];
        const float c_cg = UNIT ? L.cgI[k] : L.pc[k].c * L.cgI[k];
        L.J64[k].v = (double)(c_cg / (1.f - rho[k] * rho[k]));
      }
      L.y_last = yprev;
      L.any_frame = true;
      steady = true;
    }
  }
  // ---- transient: 8-frame blocks with per-candidate regimes until every candidate is steady
  while (blk < nfull && !steady) {
    float yb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) yb[q] = ld((blk * 8 + q));
    L.template consume<8>(yb);
    ++blk;
    steady = L.all_steady();
  }
  // ---- steady state, fused over candidates.  With frozen gains the innovation obeys
  //      d' = rho d + (y' - a y),  rho = a (1 - c t):  two FMAs per frame and candidate, the
  //      candidates interleaved inside the frame loop (independent chains issue back to back).
  //      Loads are double-buffered: the next 8 frames are in flight while 8 are consumed.
  if (steady && blk < nfull) {
    constexpr bool PK = EKS_NLL_PACKED && sizeof(R) == sizeof(float) && (NCL % 2 == 0);
    constexpr int NP = PK ? NCL / 2 : 1;
    // (pole form, or - where the wave has a pole above kKappaRho - the complement form with rho standing for -(1 - pole))
    R rho[NCL], dk[NCL], s2[NCL];
#pragma unroll
    for (int k = 0; k < NCL; ++k) {
      rho[k] = L.slow_pole ? R(0.f) - L.kapI[k] : L.pole(k);
      dk[k] = L.dl[k];
      s2[k] = R(0.f);
    }
    f32x2 rho2[NP], dk2[NP], s22[NP];        // packed form: candidate pairs in 64-bit registers (see f32x2)
    if constexpr (PK) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        rho2[p] = f32x2{val(rho[2 * p]), val(rho[2 * p + 1])};
        dk2[p] = f32x2{val(dk[2 * p]), val(dk[2 * p + 1])};
        s22[p] = f32x2{0.f, 0.f};
      }
    }
    float yprev = L.y_last;
    float ya[8], yb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) ya[q] = ld((blk * 8 + q));
    const int first = blk;
    // one candidate with its derivative (the gradient path): (d, d') and (sum d^2, sum d d') as packed pairs - four
    // issue slots per frame where the dual-number operators compile to 15 (round 5; see nll_conv_chunk_dual)
    constexpr bool PKD = EKS_NLL_PACKED && sizeof(R) == sizeof(Dual) && NCL == 1;
    f32x2 Xd = f32x2{val(dk[0]), der(dk[0])}, S2d = f32x2{0.f, 0.f};
    const f32x2 Rd = f32x2{val(rho[0]), val(rho[0])};
    const float rhod = der(rho[0]);
    auto eat_as = [&](const float (&yy)[8], auto kap_tag) {
      constexpr bool KAP = decltype(kap_tag)::value != 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float dy = UNIT ? (yy[q] - yprev) : (float)((double)yy[q] - a_d * (double)yprev);
        yprev = yy[q];
        if constexpr (PKD) {
          const f32x2 step = Rd * Xd + f32x2{dy, rhod * Xd[0]};
          if constexpr (KAP) Xd = Xd + step;                                  // d + (u - kappa d)
          else Xd = step;
          S2d = f32x2{Xd[0], Xd[0]} * Xd + S2d;
        } else if constexpr (PK) {
          const f32x2 dy2 = f32x2{dy, dy};
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            if constexpr (KAP) dk2[p] = dk2[p] + (rho2[p] * dk2[p] + dy2);      // d + (u - kappa d)
            else dk2[p] = rho2[p] * dk2[p] + dy2;
            s22[p] = s22[p] + dk2[p] * dk2[p];
          }
        } else {
#pragma unroll
          for (int k = 0; k < NCL; ++k) {
            if constexpr (KAP) dk[k] = dk[k] + (rho[k] * dk[k] + R(dy));
            else dk[k] = rho[k] * dk[k] + R(dy);
            s2[k] = s2[k] + dk[k] * dk[k];
          }
        }
      }
    };
    const bool kap_form = L.slow_pole;
    auto eat = [&](const float (&yy)[8]) {
      if (kap_form) eat_as(yy, IntTag<1>());
      else eat_as(yy, IntTag<0>());
    };
    auto flush = [&]() {
      if constexpr (PKD) {
        L.acc2[0].add(make_real(R(), S2d[0], 2.f * S2d[1]));
        S2d = f32x2{0.f, 0.f};
      } else if constexpr (PK) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          L.acc2[2 * p].add(make_real(R(), s22[p].x, 0.f));
          L.acc2[2 * p + 1].add(make_real(R(), s22[p].y, 0.f));
          s22[p] = f32x2{0.f, 0.f};
        }
      } else {
#pragma unroll
        for (int k = 0; k < NCL; ++k) {
          L.acc2[k].add(s2[k]);
          s2[k] = R(0.f);
        }
      }
    };
    for (; blk + 2 <= nfull; blk += 2) {
#pragma unroll
      for (int q = 0; q < 8; ++q) yb[q] = ld(((blk + 1) * 8 + q));
      eat(ya);
      if (blk + 2 < nfull) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ya[q] = ld(((blk + 2) * 8 + q));
      }
      eat(yb);
      if ((blk & 2) != 0) flush();          // float32 partial sums span at most 32 frames
    }
    if (blk < nfull) {
      eat(ya);
      ++blk;
    }
    flush();
    // back to the (d, y) state of the transient code so a ragged tail can continue
    const int eaten = (blk - first) * 8;
    if constexpr (PKD) dk[0] = make_real(R(), Xd[0], Xd[1]);
    if constexpr (PK) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        dk[2 * p] = make_real(R(), dk2[p].x, 0.f);
        dk[2 * p + 1] = make_real(R(), dk2[p].y, 0.f);
      }
    }
#pragma unroll
    for (int k = 0; k < NCL; ++k) {
      L.n_post[k] += eaten;
      L.dl[k] = dk[k];
      L.rg_last[k] = L.rgI[k];
    }
    L.y_last = yprev;
    L.any_frame = true;
  }
  for (int i = blk * 8; i < len; ++i) {        // ragged tail (or a chunk shorter than 8 frames)
    float y1[8];
    y1[0] = ld(i);
    L.template consume<1>(y1);
  }
  nll_lane_finish<R, NCL, UNIT>(L, len, xref, out);
}

// ---- lean form of the converged-entry summary (round 4) -----------------------------------------------------
// The grid search evaluates 64 candidates per chain; every chunk past the first enters with the converged filter
// variance (see nll_summarize_chunk), i.e. needs none of the transient-regime state above.  nll_lean_chunk is that
// branch alone, written so that nothing but the loops' own state is live inside them: NC candidates per lane
// paired in 64-bit registers (rho, d, sum d^2; while rho^t is alive also rho^t and sum d rho^t), float64 running
// sums, 16 frames of y in flight - ~180 VGPRs at NC = 16 where the general lane body needs 256 at NC = 8.  The
// steady-state constants of a candidate are functions of (r, s q) alone: they are formed in float64 when needed
// (lean_const: before the loops for the pole, after them for the gains) instead of being carried through.
// Arithmetic and operation order per candidate are those of nll_summarize_chunk's converged-entry branch: for the
// same chunk boundaries the two agree to float32 rounding of the constants (tests/test_host_sim.py).
struct LeanConst {
  float rho, g, rg, cg, logS;     // pole of the innovation recursion, 1 / S_inf, r / S_inf, c / S_inf, log S_inf
  float kap;                      // 1 - rho, rounded ONCE from float64 (the pole's complement: see kKappaRho)
};
// A float32 pole is off by up to 3e-8, i.e. by 3e-8 / (1 - rho) of its distance from one - and that is what the filter's
// gain, and with it the NLL of a slow candidate, is off by (measured 4e-6 at 1 - rho ~ 2e-3: variances in the hundreds).
// Where some pole of a wave's candidates lies above kKappaRho the recursion runs in the complement form
//     d' = d + (u - kappa d),   kappa = 1 - rho rounded once from float64 (6e-8 of ITSELF),
// one more operation per frame and candidate; below it the pole form's error is at most 3e-8 / 0.02 = 1.5e-6.
// float64 reciprocal / square root for positive, well-scaled arguments: hardware seed + Newton steps on the device
// (the library routines' scaling and fix-up sequences cost more than the rest of lean_const together, and it runs
// once per candidate and chunk), the library routines on the host.  Within an ulp or two of them - rho is rounded
// to float32 afterwards, so the float32 value differs from the exactly rounded one in at most its last bit in
// rare cases (an absolute 6e-8 on a pole whose distance from 1 is >= 1e-2 on the grid's slowest candidates).
EKS_HD double lean_rcp(double x) { return rcp(x); }
EKS_HD double lean_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);                       // ~1 / sqrt(x)
  double h = 0.5 * y, g = x * y;                            // g ~ sqrt(x), h ~ 1 / (2 sqrt(x))
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  return __builtin_fma(__builtin_fma(-g, g, x), h, g);      // one residual correction
#else
  return sqrt(x);
#endif
}
template <bool UNIT>
EKS_HD LeanConst lean_const(double r_d, double a_d, double c_d, double sq) {
  // riccati_fixed_point's value branch (its derivative is not needed here)
  const double a1 = UNIT ? 1.0 : a_d, c1 = UNIT ? 1.0 : c_d;
  const double c2 = c1 * c1;
  const double beta = r_d * (1.0 - a1 * a1) - sq * c2;
  const double disc = lean_sqrt(beta * beta + 4.0 * c2 * sq * r_d);
  const double Ci = beta > 0.0 ? (2.0 * sq * r_d) * lean_rcp(beta + disc) : (disc - beta) * lean_rcp(2.0 * c2);
  const double S_d = r_d + Ci * c2;
  const double g_d = lean_rcp(S_d);
  const double cg_d = c1 * g_d;
  const double t_d = Ci * cg_d;
  LeanConst k;
  k.g = (float)g_d;
  k.rg = (float)(r_d * g_d);
  k.cg = (float)cg_d;
  k.logS = -fast_log(k.g);                                  // log S = -log g, float32 (|error| ~1e-7: 1e-9 of an NLL)
  k.rho = UNIT ? k.rg : (float)(a1 * (1.0 - c1 * t_d));
  k.kap = UNIT ? (float)(Ci * g_d) : (float)(1.0 - a1 * (1.0 - c1 * t_d));      // (unit model: 1 - r g = C g, no cancellation)
  return k;
}

// A lean summary leaves the lane through a SINK; nothing is carried in registers across the frame loops (eta, known
// after pass 1, waits in the stash):
//   sink.xref(x) | sink.aj(k, A, J) (only for a summary with A != 0) | at the end: sink.eta(k, v), sink.b(k, v), sink.ell(k, v)
template <int NC>
struct LeanOut {                      // a sink that just keeps the fields (host simulator, micro-benchmarks)
  float A[NC], B[NC], Eta[NC], J[NC];
  double Ell[NC];
  float xr;
  EKS_HD void xref(float v) { xr = v; }
  EKS_HD void eta(int k, float v) { Eta[k] = v; }
  EKS_HD void aj(int k, float a, float j) { A[k] = a; J[k] = j; }
  EKS_HD void b(int k, float v) { B[k] = v; }
  EKS_HD void ell(int k, double v) { Ell[k] = v; }
};

// One lane: chunk [t0, t0 + len) of one chain for NC candidates, converged entry.  `sq(k)` returns s_k q of the
// lane's chain (fetched on demand: nothing candidate-specific stays in registers across the loops but the loop
// state).  `stash` parks the candidates' constants (g, rg, cg, log S: 4 NC floats per lane, element i at
// stash[i * stride]) between the start of the chunk and its end - LDS on the device, a local array in the host
// simulator.  Returns, wave-uniformly,
//   0  the chunk does not qualify (the filter variance has not converged t0 frames in: rho^(2 t0) >= 1e-20 for
//      some candidate of the wave) - nothing has been consumed, the caller summarises it with the exact-entry code;
//   1  summary with A = 0: rho^t died inside the chunk for every candidate (the usual case) - the mean entering
//      the next chunk is b whatever came before, J is the (chain, candidate) constant c cg / (1 - rho^2);
//   2  summary with A = rho^len != 0 for some candidate (a pole so close to one that rho^t outlives the chunk):
//      sink.aj carries the chunk's own values and the assembly has to walk the chunks in order.
template <int NC, bool UNIT, typename LD, typename SQ, typename SINK>
EKS_HD int nll_lean_chunk(const LD& ld, int t0, int len, double r_d, double a_d, double c_d, const SQ& sq,
                          float* stash, int stride, SINK& out) {
  static_assert(NC % 2 == 0, "candidates are paired");
  constexpr int NP = NC / 2;
  constexpr float kDeadA = 1e-5f;                       // NllLane<float>::kDeadA
  const int nfull = len / 8;
  const float af = (float)a_d;
  f32x2 rho2[NP], nk2[NP];                             // pole; minus its complement
  bool ok = t0 > 0;
  bool slow_pole = false;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    float rr[2], kk[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 2 * p + h;
      const LeanConst c = lean_const<UNIT>(r_d, a_d, c_d, sq(k));
      kk[h] = -c.kap;
      slow_pole = slow_pole || fabsf(c.rho) > kKappaRho;
      stash[(4 * k + 0) * stride] = c.g;
      stash[(4 * k + 1) * stride] = c.rg;
      stash[(4 * k + 2) * stride] = c.cg;        // (row 4 k + 3 parks eta from the end of pass 1 to the end of the chunk;
      rr[h] = c.rho;                             //  log S_inf = -log g is formed again there)
      const float nl = -logf(fmaxf(fabsf(c.rho), 1e-30f));
      ok = ok && fabsf(c.rho) < 1.f && 2.f * (float)t0 * nl > 46.f;
    }
    rho2[p] = f32x2{rr[0], rr[1]};
    nk2[p] = f32x2{kk[0], kk[1]};
  }
  if (!EKS_WAVE_ALL(ok)) return 0;
  const bool use_kappa = !EKS_WAVE_ALL(!slow_pole);      // (wave-uniform)
  const float y0 = len > 0 ? ld(0) : 0.f;
  out.xref(UNIT ? y0 : y0 / (float)c_d);
  const float ystart = UNIT ? y0 : y0 / af;            // reference start: the first innovation is y_0 - c xref = 0
  // Rows travel through TWO sets of four 8-frame buffers: while one set (32 frames) is consumed, the other is in
  // flight - requested at the top of the half-iteration before the one that consumes it, i.e. 32 frames (~1 us of
  // the main loop) ahead.  (The two-buffer form of the general lane body requests 8 frames ahead, ~0.27 us - less
  // than a loaded HBM round trip - so its two waves per SIMD regularly both sat waiting; and a ring refilled
  // buffer by buffer ends, as the compiler counts outstanding loads across the back edge, in a wait for ALL of them
  // at the loop top, the request just made included.  With whole sets that wait is for loads a half-iteration old.)
  // Requests past the chunk's last whole block are redirected to it instead of being skipped: the loops stay
  // straight-line code, and a row read twice costs nothing.
  constexpr int kSet = 4;
  const int last_blk = nfull > 0 ? nfull - 1 : 0;
  float ring[2][kSet][8];                              // (indexed by compile-time constants only: registers)
  auto request = [&](auto set_tag, int first) {        // ring[S][r] <- block first + r (clamped to the last)
    constexpr int S = decltype(set_tag)::value;
#pragma unroll
    for (int r = 0; r < kSet; ++r) {
      const int b = first + r < last_blk ? first + r : last_blk;
#pragma unroll
      for (int q = 0; q < 8; ++q) ring[S][r][q] = nfull > 0 ? ld(b * 8 + q) : 0.f;
    }
  };
  const IntTag<0> setA;
  const IntTag<1> setB;
  int blk = 0;                                         // whole blocks consumed by the running pass
  // ---- pass 1, while rho^t is alive: the true innovations are d_t = d0_t - c m_in rho^t (d0: zero-start), so the
  // chunk's likelihood needs sum d0_t rho^t beside sum d0_t^2.  Its own short pass over the chunk's first frames:
  // only (rho, d0, rho^t, the sum) of the pairs still alive are touched - NA is halved as the pairs die (wave-uniform,
  // examined every 32 frames; the grid kernel deals candidates to its waves round-robin, slowest first, so after
  // the first examinations only the lane's slowest pair is left and every wave pays the same few per cent; any
  // order is correct).  The sum stays in float32: its terms decay geometrically and it multiplies m_in - xref, a few
  // pixels - what float32 loses there is 1e-7 of a term that is itself ~1e-5 of the chunk's log-likelihood.
  bool alive;
  {
    f32x2 d1[NP], w2[NP], s12[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      d1[p] = f32x2{0.f, 0.f};
      w2[p] = f32x2{1.f, 1.f};
      s12[p] = f32x2{0.f, 0.f};
    }
    float yprev = ystart;
    auto frame1 = [&](float yy, auto na_tag) {
      constexpr int NA = decltype(na_tag)::value;
      const float dy = UNIT ? (yy - yprev) : (float)((double)yy - a_d * (double)yprev);
      yprev = yy;
      const f32x2 dy2 = f32x2{dy, dy};
#pragma unroll
      for (int p = 0; p < NA; ++p) {
        d1[p] = rho2[p] * d1[p] + dy2;
        s12[p] = s12[p] + d1[p] * w2[p];
        w2[p] = w2[p] * rho2[p];
      }
      EKS_SCHED_FENCE();
    };
    using TagAll = IntTag<NP>;
    int na = NP;                                   // pairs 0 .. na - 1 may still be alive
    auto run32 = [&](auto set_tag) {
      constexpr int S = decltype(set_tag)::value;
      auto go = [&](auto na_tag) {
#pragma unroll
        for (int r = 0; r < kSet; ++r) {
#pragma unroll
          for (int q = 0; q < 8; ++q) frame1(ring[S][r][q], na_tag);
        }
      };
      if (NP >= 8 && na <= 1) go(IntTag<1>());
      else if (NP >= 8 && na <= 2) go(IntTag<(NP >= 8 ? 2 : NP)>());
      else if (NP >= 8 && na <= 4) go(IntTag<(NP >= 8 ? 4 : NP)>());
      else go(TagAll());
      blk += kSet;
      int top = 0;
#pragma unroll
      for (int p = NP - 1; p >= 0; --p) {
        const bool dead = fabsf(w2[p][0]) < kDeadA && fabsf(w2[p][1]) < kDeadA;
        if (top == 0 && !EKS_WAVE_ALL(dead)) top = p + 1;
      }
      na = top < na ? top : na;                   // (never grows: a dead pair's rho^t is no longer advanced)
    };
    request(setA, 0);
    while (na > 0 && blk + kSet <= nfull) {
      request(setB, blk + kSet);
      run32(setA);
      if (!(na > 0 && blk + kSet <= nfull)) break;
      request(setA, blk + kSet);
      run32(setB);
    }
    alive = na > 0;
    if (alive) {
      // rho^t outlives the chunk's whole 32-frame sets (or the chunk has none): the remaining frames one at a
      // time - the summary keeps A = rho^len
      for (int i = blk * 8; i < len; ++i) frame1(ld(i), TagAll());
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float cg = stash[(4 * k + 2) * stride];
      stash[(4 * k + 3) * stride] = s12[k / 2][k & 1] * cg;
      if (alive) {
        const float rho = rho2[k / 2][k & 1], w = w2[k / 2][k & 1];
        const bool live_k = !(fabsf(w) < kDeadA);      // (a pair that left the alive set keeps a stale, dead rho^t)
        const float c_cg = UNIT ? cg : (float)c_d * cg;
        // sum rho^2t over the frames seen = (1 - rho^2n) / (1 - rho^2); rho^2n vanishes in float32 once rho^n is dead
        out.aj(k, live_k ? w : 0.f, live_k ? c_cg * (1.f - w * w) / (1.f - rho * rho) : c_cg / (1.f - rho * rho));
      }
    }
  }
  // ---- pass 2, the whole chunk: d0' = rho d0 + (y' - a y) and the sum of its squares - two FMAs per frame and
  // candidate.  The float32 partial sums go to float64 once per 32-frame set.
  float yprev = ystart;
  f32x2 dk2[NP], s22[NP];
  double acc2[NC];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    dk2[p] = f32x2{0.f, 0.f};
    s22[p] = f32x2{0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) acc2[k] = 0.0;
  auto pass2 = [&](auto kap_tag) {
    constexpr bool KAP = decltype(kap_tag)::value != 0;
    auto step = [&](int p, f32x2 dy2) {
      if constexpr (KAP) dk2[p] = dk2[p] + (nk2[p] * dk2[p] + dy2);      // d + (u - kappa d)
      else dk2[p] = rho2[p] * dk2[p] + dy2;
    };
    auto eat1 = [&](float yy) {
      const float dy = UNIT ? (yy - yprev) : (float)((double)yy - a_d * (double)yprev);
      yprev = yy;
      const f32x2 dy2 = f32x2{dy, dy};
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        step(p, dy2);
        s22[p] = s22[p] + dk2[p] * dk2[p];
      }
      EKS_SCHED_FENCE();
    };
    auto flush = [&]() {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        acc2[2 * p] += (double)s22[p][0];
        acc2[2 * p + 1] += (double)s22[p][1];
        s22[p] = f32x2{0.f, 0.f};
      }
    };
    auto eat_set = [&](auto set_tag, int n) {     // the first n (<= kSet) blocks of a set
      constexpr int S = decltype(set_tag)::value;
#pragma unroll
      for (int r = 0; r < kSet; ++r) {
        if (r < n) {
#pragma unroll
          for (int q = 0; q < 8; ++q) eat1(ring[S][r][q]);
        }
      }
      flush();                                    // float32 partial sums span at most 32 frames
    };
    auto eat_full = [&](auto set_tag) {           // a whole set, no per-block predicate in the main loop
      constexpr int S = decltype(set_tag)::value;
#pragma unroll
      for (int r = 0; r < kSet; ++r) {
#pragma unroll
        for (int q = 0; q < 8; ++q) eat1(ring[S][r][q]);
      }
      flush();
    };
    blk = 0;
    request(setA, 0);
    for (; blk + 2 * kSet <= nfull; blk += 2 * kSet) {
      request(setB, blk + kSet);
      eat_full(setA);
      request(setA, blk + 2 * kSet);
      eat_full(setB);
    }
    {
      const int rem = nfull - blk;                 // 0 .. 2 kSet - 1 whole blocks left; set A holds the first kSet
      if (rem > kSet) request(setB, blk + kSet);
      eat_set(setA, rem < kSet ? rem : kSet);
      if (rem > kSet) eat_set(setB, rem - kSet);
      blk = nfull;
    }
    for (int i = blk * 8; i < len; ++i) {        // ragged tail of the sequence's last chunk: a frame at a time
      const float yy = ld(i);
      const float dy = UNIT ? (yy - yprev) : (float)((double)yy - a_d * (double)yprev);
      yprev = yy;
      const f32x2 dy2 = f32x2{dy, dy};
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        step(p, dy2);
        const f32x2 sq2 = dk2[p] * dk2[p];
        acc2[2 * p] += (double)sq2[0];
        acc2[2 * p + 1] += (double)sq2[1];
      }
    }
  };
  if (use_kappa) pass2(IntTag<1>());
  else pass2(IntTag<0>());
  // ---- finish (nll_lane_finish / recover_mean of the general lane body, phase 2 throughout)
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const float g = stash[(4 * k + 0) * stride], rg = stash[(4 * k + 1) * stride];
    const float logS = -fast_log(g);
    out.eta(k, stash[(4 * k + 3) * stride]);
    const float dl = dk2[k / 2][k & 1];
    if (UNIT) {
      out.b(k, yprev - rg * dl);
    } else {
      const float cf = (float)c_d;
      const float ic = rcp(cf);
      out.b(k, af * ((yprev - dl) * ic + (1.f - rg) * ic * dl));
    }
    const double q_v = (double)g * acc2[k];
    const double l_v = (double)len * (double)logS;
    out.ell(k, -0.5 * ((double)len * kLog2Pi + l_v + q_v));
  }
  return alive ? 2 : 1;
}

// Assemble the marginal log-likelihood of one chain for one candidate from its chunk summaries:
//   ll = sum_j [ ell_j - 0.5 log(1 + J_j P) + (eta_j m + 0.5 eta_j^2 P - 0.5 J_j m^2)/(1 + J_j P) ]
// with (m, P) the predicted belief entering chunk j (pushed through the elements in order).
// RD is double or DualD; `get(j)` returns the chunk's element as Elem<RD> and its (ell, dell).
// Every summary is a function of the entering state relative to its chunk's reference xr
// (nll_summarize_chunk); its b is the absolute outgoing mean of the reference trajectory.
template <typename RD, typename Getter>
EKS_HD RD nll_assemble(int nchunks, double m0, double S0, Getter get) {
  RD m = RD(m0), P = RD(S0);
  RD ll = RD(0.0);
  Elem<RD> e_next;
  RD ell_next;
  double xr_next;
  get(0, e_next, ell_next, xr_next);
  for (int j = 0; j < nchunks; ++j) {
    const Elem<RD> e = e_next;
    const RD ell = ell_next;
    const RD mr = m - RD(xr_next);                       // entering mean relative to the reference
    if (j + 1 < nchunks) get(j + 1, e_next, ell_next, xr_next);   // in flight while chunk j is applied
    if (val(e.C) < 0.0) {     // converged-entry summary (nll_summarize_chunk): P is P_inf here
      ll = ll + ell + e.eta * mr - RD(0.5) * e.J * mr * mr;
      m = e.A * mr + e.b;
      continue;
    }
    const RD den = RD(1.0) + e.J * P;
    const RD inv = rcp(den);
    ll = ll + ell - RD(0.5) * log_with_rcp(den, inv) +
         (e.eta * mr + RD(0.5) * e.eta * e.eta * P - RD(0.5) * e.J * mr * mr) * inv;
    const RD AI = e.A * inv;
    const RD m_n = AI * (mr + P * e.eta) + e.b;
    P = AI * e.A * P + e.C;
    m = m_n;
  }
  return ll;
}

}  // namespace eks
