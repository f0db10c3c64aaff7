// Scalar-chain Kalman algebra shared by the gfx950 kernels (eks_diag.hip) and the host-side
// numerics simulator used by the CPU tests (tests/host_sim).  Header only, no HIP types.
//
// Model of one chain (one keypoint coordinate when A, C, Q, S0 are diagonal; singlecam has
// a = c = q = 1, reference eks/singlecam_smoother.py:246-284):
//     x_{t+1} = a x_t + w,  w ~ N(0, s q)          y_t = c x_t + v_t,  v_t ~ N(0, r_t)
// Filter ordering follows the reference's dynamax call (eks/core.py:290): (m, P) entering frame t
// is the PREDICTED belief on x_t; the frame is "update with y_t, then predict".
//
// An "element" summarises a run of consecutive frames as a map on the incoming belief
// (Sarkka & Garcia-Fernandez 2021 temporal parallelisation, update-then-predict ordering):
//     x_out | x_in ~ N(A x_in + b, C)                      (run-local filter started at x_in)
//     p(y_run | x_in) = exp(ell) * exp(eta x_in - J x_in^2 / 2)
// Every quantity except b, eta, ell is non-negative and every update below is a sum of products of
// non-negative terms or an innovation (y - c b): there is no subtractive cancellation, which is
// why float32 holds 1e-6 relative against the float64 oracle.
#pragma once

#if defined(__HIPCC__)
#define EKS_HD __host__ __device__ __forceinline__
#else
#define EKS_HD inline
struct float2 {
  float x, y;
};
#endif

// Final outputs (ms, Vs, output tables) are written once and never read again by this library.  A
// plain store leaves them dirty in the L2s and the 256 MB Infinity Cache, and the NEXT kernels'
// reads then pay for the write-back (MI355X: K3 leaves 614 MB of ms / Vs behind on the C3 shape;
// the median pass that follows took 111 us behind plain stores and 94 us behind non-temporal ones,
// the whole step 0.600 -> 0.580 ms, the C5 share 2.185 -> 2.123 ms; same box, alternating runs).
// -DEKS_PLAIN_STORES restores plain stores for A/B builds (tools: EKS_HIP_LIB selects the library).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(EKS_PLAIN_STORES)
#define EKS_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define EKS_STREAM_STORE(ptr, val) (*(ptr) = (val))
#endif

namespace eks {

template <typename R>
struct Elem {
  R A, b, C, eta, J;
};

template <typename R>
struct ChainParams {
  R a, c, q_s;  // transition, emission, s*q
  // 1 - a and 1 - a^2, each rounded ONCE from float64.  A float32 `a` is off by up to 6e-8 of itself, and on a decaying
  // chain under heavy smoothing the variances' fixed points divide by (1 - a^2 + gain terms) ~ 1e-2: a times a variance
  // is formed as X - oma2 X, a times a mean as m - oma m (round 6: 4e-6 -> below 2e-6 on Vs at a = 0.988, s q / r = 4e-5)
  R oma = R(0), oma2 = R(0);
  EKS_HD R times_a(R x) const { return x - oma * x; }
  EKS_HD R times_a2(R x) const { return x - oma2 * x; }
};

template <typename R>
EKS_HD R rcp(R x) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (sizeof(R) == 4) {
    // v_rcp_f32 (1 ulp) + one Newton step: <= 0.5 ulp-ish, keeps the 1e-6 budget.  The step is NaN exactly
    // where the seed already is the IEEE answer (x = inf -> 0, x = 0 -> inf: x * r0 = inf * 0): keep the seed
    float r0 = __builtin_amdgcn_rcpf(x);
    const float r1 = r0 * (2.0f - x * r0);
    return r1 == r1 ? r1 : r0;
  } else {
    // v_rcp_f64 seed + two Newton steps (y <- y + y (1 - x y)): within an ulp of the IEEE quotient in 5
    // dependent instructions instead of the 12-deep v_div_scale / v_div_fmas / v_div_fixup sequence.  Callers
    // divide by an innovation variance or 1 + C J (>= the variance floor; variances are clamped to 1e30 at
    // load, eks_diag_lane.hpp: clip_var), so the special cases that sequence exists for do not arise in a
    // healthy chain; where they do (x = inf, 0 or denormal: the refinement is inf * 0 = NaN) the seed, which is
    // the IEEE answer there, is returned instead of a NaN that would poison the chain.
    const double y0 = __builtin_amdgcn_rcp(x);
    double y = __builtin_fma(y0, __builtin_fma(-x, y0, 1.0), y0);
    y = __builtin_fma(y, __builtin_fma(-x, y, 1.0), y);
    return y == y ? y : y0;
  }
#else
  return R(1) / x;
#endif
}

template <typename R>
EKS_HD Elem<R> elem_identity() {
  return Elem<R>{R(1), R(0), R(0), R(0), R(0)};
}

// Append one frame (y, r) to a running element.  UNIT: a = c = 1 folded at compile time.
// Also returns the frame's innovation statistics for likelihood accumulation:
//   S = r + C c^2 (innovation variance of the run-local filter), g = 1/S, d = y - c b, so that the
//   frame's x_in-independent log-likelihood part is -0.5 * (log(2 pi S) + d^2 g).
template <typename R, bool UNIT>
EKS_HD void elem_append(Elem<R>& e, R y, R r, const ChainParams<R>& p, R& S, R& g, R& d) {
  const R c = UNIT ? R(1) : p.c;
  const R a = UNIT ? R(1) : p.a;
  const R Cc = UNIT ? e.C : e.C * c;
  S = UNIT ? (r + e.C) : (r + Cc * c);
  g = rcp(S);
  d = UNIT ? (y - e.b) : (y - c * e.b);
  const R rg = r * g;
  const R Acg = UNIT ? e.A * g : e.A * c * g;
  e.eta = e.eta + Acg * d;
  e.J = e.J + (UNIT ? Acg * e.A : Acg * e.A * c);
  e.b = UNIT ? (e.b + Cc * g * d) : p.times_a(e.b + Cc * g * d);
  e.A = UNIT ? e.A * rg : p.times_a(e.A * rg);
  e.C = UNIT ? (e.C * rg + p.q_s) : (p.times_a2(e.C * rg) + p.q_s);
}

template <typename R, bool UNIT>
EKS_HD void elem_append(Elem<R>& e, R y, R r, const ChainParams<R>& p) {
  R S, g, d;
  elem_append<R, UNIT>(e, y, r, p, S, g, d);
}

// Compose: first `i` (earlier frames) then `j` (later frames).
template <typename R>
EKS_HD Elem<R> elem_combine(const Elem<R>& i, const Elem<R>& j) {
  const R inv = rcp(R(1) + i.C * j.J);
  Elem<R> o;
  const R AjI = j.A * inv;
  o.A = AjI * i.A;
  o.b = AjI * (i.b + i.C * j.eta) + j.b;
  o.C = AjI * j.A * i.C + j.C;
  const R AiI = i.A * inv;
  o.eta = AiI * (j.eta - j.J * i.b) + i.eta;
  o.J = AiI * i.A * j.J + i.J;
  return o;
}

// Push the belief N(m, P) on x_in through an element: belief on x_out.
template <typename R>
EKS_HD void elem_apply(const Elem<R>& e, R& m, R& P) {
  const R inv = rcp(R(1) + e.J * P);
  const R AI = e.A * inv;
  m = AI * (m + P * e.eta) + e.b;
  P = AI * e.A * P + e.C;
}

// Pull information (eta, J) about x_out back through an element: information about x_in from the
// element's own frames and everything after them.
template <typename R>
EKS_HD void elem_back(const Elem<R>& e, R& eta, R& J) {
  const R inv = rcp(R(1) + e.C * J);
  const R AI = e.A * inv;
  const R eta_n = AI * (eta - J * e.b) + e.eta;
  const R J_n = AI * e.A * J + e.J;
  eta = eta_n;
  J = J_n;
}

// One filter frame on the predicted belief (m, P): writes the filtered belief, returns the next
// predicted belief in (m, P).  Posterior variance in product form P r / (P c^2 + r).
template <typename R, bool UNIT>
EKS_HD void filter_step(R& m, R& P, R y, R r, const ChainParams<R>& p, R& mf, R& Pf) {
  const R c = UNIT ? R(1) : p.c;
  const R a = UNIT ? R(1) : p.a;
  const R Pc = UNIT ? P : P * c;
  const R g = rcp(UNIT ? (P + r) : (Pc * c + r));
  const R d = UNIT ? (y - m) : (y - c * m);
  mf = m + Pc * g * d;
  Pf = P * r * g;
  m = UNIT ? mf : p.times_a(mf);
  P = UNIT ? (Pf + p.q_s) : (p.times_a2(Pf) + p.q_s);
}

// Combine the predicted belief on x (from the past) with information (eta, J) from the future.
template <typename R>
EKS_HD void fuse_info(R& m, R& P, R eta, R J) {
  const R inv = rcp(R(1) + J * P);
  m = (m + P * eta) * inv;
  P = P * inv;
}

// One RTS frame: (ms, Ps) is the smoothed belief on x_{t+1} on entry, on x_t on exit.
// With Pp = a^2 Pf + s q and h = s q / Pp the smoother gain is G = a Pf / Pp = (1 - h) / a, i.e. 1 - G = (h - (1 - a)) / a
// =: g exactly.  Under heavy smoothing (s q << Pf: h ~ 1e-2, and 1 - a of that size on a decaying chain) a float32 G
// sits within 1e-2 of one: it carries 1 - G to only 6e-8 / g of itself, and the variance recursion's fixed point is
// Pf h / (1 - G^2) - 6.5e-6 on Vs in round 5's fuzz sweeps (s ~ 5e-4, profiles/r05_fuzz3.txt).  Round 6: where g is small
// the step is taken in the DEVIATION form
//     Ps_t = Pf h + G^2 Ps_{t+1} = Ps_{t+1} + ( Pf h - g (2 - g) Ps_{t+1} )
// in which the small quantity multiplies a difference instead of being formed as a complement of one (1 - a arrives
// rounded once from float64: ChainParams::oma); elsewhere (|g| >= 1/4: light smoothing, or a frame in front of an
// occluded one whose Ps_{t+1} dwarfs Pf, where the deviation form would cancel) the products of non-negative terms of
// rounds 1-5 stand.  Per lane, by select.
template <typename R, bool UNIT>
EKS_HD void rts_step(R& ms, R& Ps, R mf, R Pf, const ChainParams<R>& p) {
  const R Pp = UNIT ? (Pf + p.q_s) : (p.times_a2(Pf) + p.q_s);
  const R ig = rcp(Pp);
  const R h = p.q_s * ig;
  const R G = UNIT ? Pf * ig : p.a * Pf * ig;
  const R amf = UNIT ? mf : p.times_a(mf);
  const R g = UNIT ? h : (h - p.oma) * rcp(p.a);
  ms = mf + G * (ms - amf);                // (the mean is not divided by 1 - G^2: its product form holds 1e-6 everywhere)
  const R Ps_prod = Pf * h + G * G * Ps;
#ifdef EKS_RTS_PRODUCT_ONLY                 // (A/B builds: rounds 1-5's step)
  (void)g;
  Ps = Ps_prod;
#else
  const R Ps_dev = Ps + (Pf * h - g * (R(2) - g) * Ps);
  Ps = (g < R(0.25) && g > R(-0.25)) ? Ps_dev : Ps_prod;
#endif
}

}  // namespace eks
