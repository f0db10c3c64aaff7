// Shared-lag form of the grid search's converged-entry chunk summaries (round 5).
//
// The loss of eks/core.py:640-650 is evaluated for 64 candidate values of s per chain.  Past the entry transient a
// candidate's innovation is the zero-start sequence d0_t = sum_m rho^m u_{t-m} of the shared input u_t = y_t - a y_{t-1}
// (nll_lean_chunk: two FMAs per frame and candidate).  For a pole with rho^NLAG negligible the chunk's sum of squares
// is a fixed combination of NLAG lag sums that ALL such candidates of a chain share:
//     sum_{t<L} d0_t^2 = [ c_0 + 2 sum_{k>=1} rho^k c_k - rho^2 d0_{L-1}^2 ] / (1 - rho^2),   c_k = sum_t u_t u_{t-k}
// (exact when all L - 1 lags are kept; u_t = 0 before the chunk), and the other two quantities a converged-entry
// summary needs come from the chunk's first and last NLAG inputs:
//     d0_{L-1} = sum_{m<NLAG} rho^m u_{L-1-m},      sum_t d0_t rho^t = [ sum_i rho^i u_i - rho^(L+1) d0_{L-1} ] / (1 - rho^2).
// So a block spends NLAG FMAs per frame on the lag sums instead of two per frame on every fast candidate (36 of
// BASELINE's 64 on the C3 shape: 128 -> 72 FMAs per chain and frame), and the fast candidates' summaries are
// formed from the lag sums in float64 by the assembly (lag_summary below).  The slow candidates keep the recursion
// (nll_lag_chunk: the lean lane body with NP pairs, plus this wave's turn at the lag products).
#pragma once
#include "eks_nll_lane.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define EKS_WAVE_ANY(x) (__any(x) != 0)
#else
#define EKS_WAVE_ANY(x) (x)
#endif

namespace eks {

// a * b + {c.hi, c.hi} / {c.lo, c.lo}: the inputs u of two consecutive frames share a 64-bit register pair, and the
// recursion adds one of them to both candidates of a pair.  The compiler folds the low-half splat into the
// instruction's op_sel_hi field but copies the high half to another register first (one v_mov per odd frame, 4 % of the
// frame loop): the high-half form is spelled out.
EKS_HD f32x2 fma_splat_lo(f32x2 a, f32x2 b, f32x2 c) { return a * b + f32x2{c[0], c[0]}; }
EKS_HD f32x2 fma_splat_hi(f32x2 a, f32x2 b, f32x2 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
#else
  return a * b + f32x2{c[1], c[1]};
#endif
}
// One wait for a whole 32-frame set of rows: row loads return in order, so once the set's LAST row has arrived all of
// it has.  Touching that register first (an empty statement the optimiser cannot see through, and nothing may be
// scheduled across) makes the compiler wait once - for all but the 32 requests of the other set - instead of in front
// of every one of the 32 uses.
#if defined(__HIP_DEVICE_COMPILE__)
#define EKS_ROWS_ARRIVED(last)              \
  do {                                      \
    asm volatile("" : "+v"(last));          \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)
#else
#define EKS_ROWS_ARRIVED(last) do { } while (0)
#endif

constexpr int kLagND = 8;                 // lag PAIRS: the lag sums c_0 .. c_{2 kLagND - 1}
constexpr int kLagN = 2 * kLagND;
// a pole is "fast" when rho^NLAG <= 6e-8 (1 - rho): truncating the series above at NLAG lags then moves the chunk's
// sum of squares by less than 2 rho^NLAG / (1 + rho) of itself whatever the data (all lag sums bounded by c_0)
EKS_HD double lag_rho_max(int nlag) {
  double lo = 0.0, hi = 0.999;
  for (int it = 0; it < 60; ++it) {
    const double m = 0.5 * (lo + hi);
    if (pow(m, (double)nlag) <= 6e-8 * (1.0 - m)) lo = m; else hi = m;
  }
  return lo;
}
// the smallest s q whose steady-state pole is at most rho_max in magnitude (the pole decreases in s q):
//     rho = a r / (r + c^2 C_inf),  C_inf = a^2 C_inf r / (r + c^2 C_inf) + s q
//  => with p = rho / |a|:  c^2 C_inf = r (1 - p) / p,  s q = r (1 - p) (1 - a^2 p) / (c^2 p)
EKS_HD double lag_sq_threshold(double r, double a, double c, double rho_max) {
  const double aa = fabs(a);
  if (!(aa > rho_max)) return 0.0;                       // every pole is below |a| <= rho_max
  const double p = rho_max / aa;
  return r * (1.0 - p) * (1.0 - a * a * p) / (c * c * p);
}

// float64 steady-state constants of one (chain, candidate) - lean_const without the float32 roundings
struct LagConst {
  double rho, g, rg, cg, logS, Jc;
};
template <bool UNIT>
EKS_HD LagConst lag_const(double r_d, double a_d, double c_d, double sq) {
  const double a1 = UNIT ? 1.0 : a_d, c1 = UNIT ? 1.0 : c_d;
  const double c2 = c1 * c1;
  const double beta = r_d * (1.0 - a1 * a1) - sq * c2;
  const double disc = sqrt(beta * beta + 4.0 * c2 * sq * r_d);
  const double Ci = beta > 0.0 ? (2.0 * sq * r_d) / (beta + disc) : (disc - beta) / (2.0 * c2);
  const double S_d = r_d + Ci * c2;
  LagConst k;
  k.g = 1.0 / S_d;
  k.rg = r_d * k.g;
  k.cg = c1 * k.g;
  k.logS = log(S_d);
  k.rho = UNIT ? k.rg : a1 * (1.0 - c1 * (Ci * k.cg));
  k.Jc = c1 * k.cg / (1.0 - k.rho * k.rho);
  return k;
}

// the same constants with hardware-seeded float64 square root / reciprocals and log S = -log g in float32 (as
// lean_const): what the gfx950 kernel forms per (chain, fast candidate) and chunk - ~40 instructions instead of the
// library routines' ~250.  J is not needed there (the assembly has it from the head role's table).
template <bool UNIT>
EKS_HD LagConst lag_const_fast(double r_d, double a_d, double c_d, double sq) {
  const double a1 = UNIT ? 1.0 : a_d, c1 = UNIT ? 1.0 : c_d;
  const double c2 = c1 * c1;
  const double beta = r_d * (1.0 - a1 * a1) - sq * c2;
  const double disc = lean_sqrt(beta * beta + 4.0 * c2 * sq * r_d);
  const double Ci = beta > 0.0 ? (2.0 * sq * r_d) * lean_rcp(beta + disc) : (disc - beta) * lean_rcp(2.0 * c2);
  LagConst k;
  k.g = lean_rcp(r_d + Ci * c2);
  k.rg = r_d * k.g;
  k.cg = c1 * k.g;
  k.logS = (double)(-fast_log((float)k.g));
  k.rho = UNIT ? k.rg : a1 * (1.0 - c1 * (Ci * k.cg));
  k.Jc = 0.0;
  return k;
}

// One fast candidate's converged-entry summary of one chunk of `len` frames from the chunk's lag sums c[NLAG], its
// first NLAG inputs uh[] and last NLAG inputs ut[] (time order) and its last observation: the outgoing mean b of
// the reference trajectory, eta and the run-local log-likelihood ell (A = 0, J = k.Jc; see nll_lean_chunk).
template <int NLAG, bool UNIT>
EKS_HD void lag_summary(const LagConst& k, double a_d, double c_d, int len, const double* c, const float* uh,
                        const float* ut, float ylast, double& b, double& eta, double& ell) {
  const double rho = k.rho;
  double h = c[NLAG - 1];
#pragma unroll
  for (int i = NLAG - 2; i >= 1; --i) h = c[i] + rho * h;
  double dl = 0.0, z = 0.0;
#pragma unroll
  for (int i = 0; i < NLAG; ++i) dl = rho * dl + (double)ut[i];            // the recursion over the last NLAG frames
#pragma unroll
  for (int i = NLAG - 1; i >= 0; --i) z = (double)uh[i] + rho * z;         // sum_i rho^i u_i
  const double inv = lean_rcp(1.0 - rho * rho);
  const double s0 = (c[0] + 2.0 * rho * h - rho * rho * dl * dl) * inv;
  eta = k.cg * z * inv;
  if (UNIT) {
    b = (double)ylast - k.rg * dl;
  } else {
    const double ic = lean_rcp(c_d);
    b = a_d * (((double)ylast - dl) * ic + (1.0 - k.rg) * ic * dl);
  }
  ell = -0.5 * ((double)len * kLog2Pi + (double)len * k.logS + k.g * s0);
}

// a LAGS sink that keeps everything (host simulator, micro-benchmarks)
template <int NLAG>
struct LagKeep {
  double c[NLAG];
  float uh[NLAG], ut[NLAG], yl;
  EKS_HD void add(int k, float v) { c[k] += (double)v; }
  EKS_HD void head(int i, float v) { uh[i] = v; }
  EKS_HD void tail(int i, float v) { ut[i] = v; }
  EKS_HD void ylast(float v) { yl = v; }
};

// One lane: a chunk of len (a multiple of 32) frames of one chain, converged entry (the CALLER has checked that the
// filter variance has converged at the chunk's first frame for every candidate), NP pairs of slow candidates by the
// recursion, plus the lag products of the 32-frame sets s with bit s % period of turn_mask set (wave-uniform: the waves
// of a block share the chunk's lag work by time - their masks partition the period; every wave computes the inputs u
// of every frame anyway).  `sq(k)` returns
// s_k q of slow candidate k of this lane's wave, slowest first; `stash` as nll_lean_chunk with 3 floats per candidate.  The `lead` wave
// also hands the chunk's first / last NLAG inputs and its last observation to `lags`.
// Returns 1 (A = 0 for every candidate) or 2 (rho^t of some candidate outlives the chunk: sink.aj has its A, J).
template <int NP, int ND, bool UNIT, typename LD, typename SQ, typename SINK, typename LAGS>
EKS_HD int nll_lag_chunk(const LD& ld, int len, double r_d, double a_d, double c_d, const SQ& sq, unsigned turn_mask,
                         int period, bool lead, float* stash, int stride, SINK& out, LAGS& lags) {
  static_assert(ND >= 1 && ND <= 15, "the inputs of 32 frames are kept: lags up to 30");
  constexpr int NC = 2 * NP, NLAG = 2 * ND;
  const int nsets = len / 32;
  const float af = (float)a_d;
  f32x2 rho2[NP], nk2[NP];                   // pole; minus its complement (kKappaRho)
  bool slow_pole = false;
  int nset_alive[NP];                      // 32-frame sets while rho^t of the pair is alive (wave-uniform)
  bool end_alive[NP];                      // ... and it still is when the chunk ends
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    float rr[2], kk[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = 2 * p + h;
      const LeanConst c = lean_const<UNIT>(r_d, a_d, c_d, sq(k));
      kk[h] = -c.kap;
      slow_pole = slow_pole || fabsf(c.rho) > kKappaRho;
      stash[(3 * k + 0) * stride] = c.g;          // (log S_inf = -log g is formed again at the end)
      stash[(3 * k + 1) * stride] = c.rg;
      stash[(3 * k + 2) * stride] = c.cg;
      rr[h] = c.rho;
    }
    rho2[p] = f32x2{rr[0], rr[1]};
    nk2[p] = f32x2{kk[0], kk[1]};
    // frames until rho^t < 1e-5 (NllLane<float>::kDeadA; ln 1e5 = 11.52), the slower pole of the pair, the slowest lane
    const float rm = fmaxf(fabsf(rr[0]), fabsf(rr[1]));
    const float nl = -logf(fminf(fmaxf(rm, 1e-30f), 0.99999994f));
    const float nf = fminf(11.6f / nl + 1.f, 1e9f);
    // in 32-frame sets, the wave's maximum by a bitwise search on ballots (the result is scalar); nsets + 1 stands for
    // "beyond the chunk"
    const int mine = (int)fminf((nf + 31.f) * (1.f / 32.f), (float)(nsets + 1));
    int ns = 0;
#pragma unroll
    for (int bit = 11; bit >= 0; --bit) {
      if (EKS_WAVE_ANY(mine >= (ns | (1 << bit)))) ns |= 1 << bit;
    }
    end_alive[p] = ns > nsets;
    nset_alive[p] = ns < nsets ? ns : nsets;
  }
  const float y0 = ld(0);
  out.xref(UNIT ? y0 : y0 / (float)c_d);
  const float ystart = UNIT ? y0 : y0 / af;            // reference start: the first input is y_0 - c xref = 0
  auto input = [&](float yy, float yp) { return UNIT ? (yy - yp) : (float)((double)yy - a_d * (double)yp); };
  // rows: two sets of four 8-frame buffers, one consumed while the other is in flight (nll_lean_chunk)
  constexpr int kSet = 4;
  float ring[2][kSet][8];
  auto request = [&](auto set_tag, int s) {            // ring[S] <- set s (clamped to the chunk's last set)
    constexpr int S = decltype(set_tag)::value;
    const int ss = s < nsets ? s : nsets - 1;
#pragma unroll
    for (int r = 0; r < kSet; ++r) {
#pragma unroll
      for (int q = 0; q < 8; ++q) ring[S][r][q] = ld((ss * kSet + r) * 8 + q);
    }
  };
  const IntTag<0> setA;
  const IntTag<1> setB;
  f32x2 X[16];                                         // the inputs u of the last 32 frames: pair a = frames 2a, 2a + 1
  // ---- pass 1, while rho^t is alive: Z = sum_i rho^i u_i per slow candidate, a set at a time (Horner inside the
  // set, Z += rho^(32 s) Z_s across sets): one FMA per frame and candidate still alive.
  f32x2 Z2[NP], W2[NP];
  {
    f32x2 R32[NP];
    int smax = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      Z2[p] = f32x2{0.f, 0.f};
      W2[p] = f32x2{1.f, 1.f};
      f32x2 t = rho2[p];
#pragma unroll
      for (int i = 0; i < 5; ++i) t = t * t;
      R32[p] = t;
      smax = nset_alive[p] > smax ? nset_alive[p] : smax;
    }
    float yprev = ystart;
    auto run_set = [&](auto set_tag, int s) {
      constexpr int S = decltype(set_tag)::value;
      EKS_ROWS_ARRIVED(ring[S][kSet - 1][7]);
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const float ya = ring[S][a / 4][(2 * a) % 8], yb = ring[S][a / 4][(2 * a + 1) % 8];
        X[a] = f32x2{input(ya, yprev), input(yb, ya)};
        yprev = yb;
      }
      if (s == 0 && lead) {
#pragma unroll
        for (int i = 0; i < NLAG; ++i) lags.head(i, X[i / 2][i & 1]);
      }
      int na = 0;                                      // pairs 0 .. na - 1 may still be alive in this set
#pragma unroll
      for (int p = 0; p < NP; ++p) na = nset_alive[p] > s ? p + 1 : na;
      auto go = [&](auto na_tag) {
        constexpr int NA = decltype(na_tag)::value;
        f32x2 zs[NA];
#pragma unroll
        for (int p = 0; p < NA; ++p) zs[p] = f32x2{0.f, 0.f};
#pragma unroll
        for (int a = 15; a >= 0; --a) {
#pragma unroll
          for (int p = 0; p < NA; ++p) zs[p] = fma_splat_hi(rho2[p], zs[p], X[a]);
#pragma unroll
          for (int p = 0; p < NA; ++p) zs[p] = fma_splat_lo(rho2[p], zs[p], X[a]);
          EKS_SCHED_FENCE();
        }
#pragma unroll
        for (int p = 0; p < NA; ++p) {
          Z2[p] = Z2[p] + W2[p] * zs[p];
          W2[p] = W2[p] * R32[p];
        }
      };
      if (NP > 2 && na > 2) go(IntTag<NP>());
      else if (NP > 1 && na > 1) go(IntTag<(NP > 1 ? 2 : 1)>());
      else go(IntTag<1>());
    };
    request(setA, 0);
    for (int s = 0; s < smax; s += 2) {
      request(setB, s + 1);
      run_set(setA, s);
      if (s + 1 >= smax) break;
      request(setA, s + 2);
      run_set(setB, s + 1);
    }
  }
  // ---- pass 2, the whole chunk: the recursion d0' = rho d0 + u and sum d0^2 of the slow candidates, the lag
  // products on this wave's sets
  f32x2 dk2[NP], s22[NP];
  double acc2[NC];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    dk2[p] = f32x2{0.f, 0.f};
    s22[p] = f32x2{0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) acc2[k] = 0.0;
#pragma unroll
  for (int a = 0; a < 16; ++a) X[a] = f32x2{0.f, 0.f};
  float yprev = ystart;
  const bool use_kappa = EKS_WAVE_ANY(slow_pole);      // (wave-uniform: the recursion's complement form, kKappaRho)
  auto pass2 = [&](auto kap_tag) {
    constexpr bool KAP = decltype(kap_tag)::value != 0;
    auto eat_set = [&](auto set_tag, auto lag_tag) {
      constexpr int S = decltype(set_tag)::value;
      constexpr bool LT = decltype(lag_tag)::value != 0;
      f32x2 E[ND], O[ND];
      float o0 = 0.f;
      if constexpr (LT) {
#pragma unroll
        for (int i = 0; i < ND; ++i) E[i] = O[i] = f32x2{0.f, 0.f};
      }
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const float ya = ring[S][a / 4][(2 * a) % 8], yb = ring[S][a / 4][(2 * a + 1) % 8];
        X[a] = f32x2{input(ya, yprev), input(yb, ya)};
        yprev = yb;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if constexpr (KAP) dk2[p] = dk2[p] + fma_splat_lo(nk2[p], dk2[p], X[a]);      // d + (u - kappa d)
          else dk2[p] = fma_splat_lo(rho2[p], dk2[p], X[a]);
          s22[p] = s22[p] + dk2[p] * dk2[p];
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if constexpr (KAP) dk2[p] = dk2[p] + fma_splat_hi(nk2[p], dk2[p], X[a]);
          else dk2[p] = fma_splat_hi(rho2[p], dk2[p], X[a]);
          s22[p] = s22[p] + dk2[p] * dk2[p];
        }
        if constexpr (LT) {
          // pair a against pair a - dl: even lag 2 dl from (lo lo, hi hi), odd lags 2 dl - 1 / 2 dl + 1 from (lo hi, hi lo)
#pragma unroll
          for (int dl = 0; dl < ND; ++dl) E[dl] = E[dl] + X[a] * X[(a - dl) & 15];
#pragma unroll
          for (int dl = 1; dl <= ND; ++dl) {
            const f32x2 xb = X[(a - dl) & 15];
            O[dl - 1] = O[dl - 1] + X[a] * f32x2{xb[1], xb[0]};
          }
          o0 = o0 + X[a][1] * X[a][0];
        }
        EKS_SCHED_FENCE();
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {                   // float32 partial sums span 32 frames
        acc2[2 * p] += (double)s22[p][0];
        acc2[2 * p + 1] += (double)s22[p][1];
        s22[p] = f32x2{0.f, 0.f};
      }
      if constexpr (LT) {
        lags.add(0, E[0][0] + E[0][1]);
        lags.add(1, o0 + O[0][0]);
#pragma unroll
        for (int dl = 1; dl < ND; ++dl) {
          lags.add(2 * dl, E[dl][0] + E[dl][1]);
          lags.add(2 * dl + 1, O[dl - 1][1] + O[dl][0]);
        }
      }
    };
    const IntTag<0> plain;
    const IntTag<1> lagset;
    int ph = 0;                                        // s % period
    auto mine = [&]() {
      const bool m = ((turn_mask >> ph) & 1u) != 0;
      ph = ph + 1 == period ? 0 : ph + 1;
      return m;
    };
    request(setA, 0);
    for (int s = 0; s < nsets; s += 2) {
      request(setB, s + 1);
      EKS_ROWS_ARRIVED(ring[0][kSet - 1][7]);          // (ahead of the branch: what its arms share is hoisted to here)
      if (mine()) eat_set(setA, lagset); else eat_set(setA, plain);
      if (s + 1 >= nsets) break;
      request(setA, s + 2);
      EKS_ROWS_ARRIVED(ring[1][kSet - 1][7]);
      if (mine()) eat_set(setB, lagset); else eat_set(setB, plain);
    }
  };
  if (use_kappa) pass2(IntTag<1>());
  else pass2(IntTag<0>());
  if (lead) {
#pragma unroll
    for (int i = 0; i < NLAG; ++i) lags.tail(i, X[(32 - NLAG + i) / 2][(32 - NLAG + i) & 1]);
    lags.ylast(yprev);
  }
  // ---- finish (as nll_lean_chunk)
  bool any_alive = false;                              // (wave-uniform)
#pragma unroll
  for (int p = 0; p < NP; ++p) any_alive = any_alive || end_alive[p];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const float g = stash[(3 * k + 0) * stride], rg = stash[(3 * k + 1) * stride], cg = stash[(3 * k + 2) * stride];
    const float logS = -fast_log(g);
    const float dl = dk2[k / 2][k & 1], rho = rho2[k / 2][k & 1];
    const float c_cg = UNIT ? cg : (float)c_d * cg;
    const float i1 = 1.f / (1.f - rho * rho);
    const float A = end_alive[k / 2] ? W2[k / 2][k & 1] : 0.f;      // rho^len
    if (any_alive) out.aj(k, A, c_cg * (1.f - A * A) * i1);          // (the wave's summaries then carry their own A, J)
    // sum_t d0_t rho^t = [ Z - rho^(len+1) d0_last ] / (1 - rho^2)
    out.eta(k, cg * (Z2[k / 2][k & 1] - rho * A * dl) * i1);
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
  return any_alive ? 2 : 1;
}


// ---------------------------------------------------------------------------------------------------------------
// Converged-entry chunk summary WITH d / d log s (round 5: the Adam loop, one value of s per keypoint).
// nll_summarize_chunk<Dual, 1> starts every chunk from a known state and the chunks' elements then have to be
// COMPOSED in order (float64 dual-number compositions: 12.5 of an iteration's 41 us on C3).  A chunk that starts after
// the filter variance has converged (rho^(2 t0) < 1e-20: the true filter's predicted variance does not depend on the
// data) needs none of that: its summary has A = 0, so its term of the log-likelihood
//     ell + eta mr - J mr^2 / 2,     mr = (mean the previous chunk hands on) - xref
// and its derivative need only the previous chunk's b and db - every chunk's term can be formed at once.
// One candidate per lane; value and derivative ride through the frame loop as float32 pairs (as in the exact-entry
// code), the sums go to float64 every 32 frames.  conv_chunk_ok() says whether a chunk length qualifies.
// ---------------------------------------------------------------------------------------------------------------
struct ConvDual {                 // the summary: every field a (value, d / d log s) pair in float64
  double b, db, eta, deta, J, dJ, ell, dell;
  float xref;
};
struct ConvConst {                // steady-state constants of one (chain, s) and their d / d log s (nll_lane_init's formulas)
  double rho, drho, g, dg, cg, dcg, rg, drg, S, dS;
  float nl;                       // -log |rho|
};
template <bool UNIT>
EKS_HD ConvConst conv_const(double r_d, double a_d, double c_d, double sq) {
  const double a1 = UNIT ? 1.0 : a_d, c1 = UNIT ? 1.0 : c_d;
  double Ci, dCi;
  riccati_fixed_point(a1, c1, r_d, sq, Ci, dCi);
  ConvConst k;
  k.S = r_d + Ci * c1 * c1;
  k.dS = dCi * c1 * c1;
  k.g = 1.0 / k.S;
  k.dg = -k.g * k.g * k.dS;
  k.cg = c1 * k.g;
  k.dcg = c1 * k.dg;
  const double t = Ci * k.cg, dt = dCi * k.cg + Ci * k.dcg;
  k.rho = a1 * (1.0 - c1 * t);
  k.drho = -a1 * c1 * dt;
  k.rg = r_d * k.g;
  k.drg = r_d * k.dg;
  k.nl = -logf(fminf(fmaxf(fabsf((float)k.rho), 1e-30f), 0.99999994f));
  return k;
}
// may every chunk of `bn` frames past the first be summarised this way?  rho^bn < 1e-10: the variance has converged
// when chunk 1 starts (rho^(2 bn) < 1e-20) and no chunk's outgoing mean remembers the incoming one (A = rho^bn)
EKS_HD bool conv_chunk_ok(const ConvConst& k, int bn) { return (float)bn * k.nl > 23.1f; }

// NB: frames per row buffer; two buffers, so 2 NB rows are requested ahead of the arithmetic.
// The recursion on (value, derivative) pairs in packed float32 (v_pk_fma_f32: both halves in one issue slot):
//     X = (d, d')            X  <- (rho, rho) X + (u, rho' d)                        sub, mul, pk_fma
//     S2 = (sum d^2, sum d d')   S2 <- (d, d) X + S2          (d / d log s of sum d^2 is twice its second half)   pk_fma
// and while rho^t is alive  W = (rho^t, (rho^t)'),  S1 <- (d, d) W + S1,  S1' += d' rho^t,  W <- (w, w)(rho, rho') + (0, w' rho).
// Four issue slots per frame once rho^t has died; written with the dual-number operators the compiler spent 15
// (measured: the frame loop was VALU-bound at 25 us of a 39 us iteration on C3).
// `pre`: the chunk's first NB rows, requested by the caller ahead of time (they do not depend on s: the loop-mode kernel
// asks for them before it waits for the step); nullptr: requested here.
template <bool UNIT, int NB = 8, typename LD>
EKS_HD void nll_conv_chunk_dual(const LD& ld, int len, const ConvConst& K, double a_d, double c_d, ConvDual& out,
                                const float* pre = nullptr) {
  const float rho = (float)K.rho, rhod = (float)K.drho;
  const f32x2 R = f32x2{rho, rho}, RD = f32x2{rho, rhod};
  const int nfull = len / NB;
  float ya[NB], yb[NB];
  if (nfull > 0) {
#pragma unroll
    for (int q = 0; q < NB; ++q) ya[q] = pre ? pre[q] : ld(q);
  }
  const float y0 = nfull > 0 ? ya[0] : ld(0);
  out.xref = UNIT ? y0 : (float)((double)y0 / c_d);
  float yprev = UNIT ? y0 : (float)((double)y0 / a_d);
  f32x2 X = f32x2{0.f, 0.f}, W = f32x2{1.f, 0.f}, S1 = f32x2{0.f, 0.f}, S2 = f32x2{0.f, 0.f};
  double s1v = 0.0, s1d = 0.0, s2v = 0.0, s2d = 0.0;
  bool alive = true;
  auto frame = [&](float yy, auto alive_tag) {
    constexpr bool AL = decltype(alive_tag)::value != 0;
    const float u = UNIT ? (yy - yprev) : (float)((double)yy - a_d * (double)yprev);
    yprev = yy;
    X = R * X + f32x2{u, rhod * X[0]};
    S2 = f32x2{X[0], X[0]} * X + S2;
    if constexpr (AL) {
      S1 = f32x2{X[0], X[0]} * W + S1;
      S1[1] += X[1] * W[0];
      W = f32x2{W[0], W[0]} * RD + f32x2{0.f, W[1] * rho};
    }
  };
  auto eat = [&](const float (&yy)[NB], auto alive_tag) {
#pragma unroll
    for (int q = 0; q < NB; ++q) frame(yy[q], alive_tag);
  };
  auto flush = [&]() {
    s2v += (double)S2[0]; s2d += 2.0 * (double)S2[1];
    S2 = f32x2{0.f, 0.f};
    s1v += (double)S1[0]; s1d += (double)S1[1];
    S1 = f32x2{0.f, 0.f};
  };
  constexpr int kPairsPerFlush = NB >= 16 ? 1 : 16 / NB;      // float32 partial sums span at most 32 frames
  int blk = 0, since = 0;
  for (; blk + 2 <= nfull; blk += 2) {
#pragma unroll
    for (int q = 0; q < NB; ++q) yb[q] = ld((blk + 1) * NB + q);
    if (alive) eat(ya, IntTag<1>()); else eat(ya, IntTag<0>());
    if (blk + 2 < nfull) {
#pragma unroll
      for (int q = 0; q < NB; ++q) ya[q] = ld((blk + 2) * NB + q);
    }
    if (alive) eat(yb, IntTag<1>()); else eat(yb, IntTag<0>());
    if (++since == kPairsPerFlush || blk + 4 > nfull) {
      since = 0;
      flush();
      if (alive) {
        const bool dead = fabsf(W[0]) < 1e-9f && fabsf(W[1]) < 1e-9f;
        alive = !EKS_WAVE_ALL(dead);
      }
    }
  }
  if (blk < nfull) {
    if (alive) eat(ya, IntTag<1>()); else eat(ya, IntTag<0>());
    ++blk;
  }
  const int ntail = len - blk * NB;                  // ragged tail: its rows requested together
  if (ntail > 0) {
#pragma unroll
    for (int q = 0; q < NB - 1; ++q) ya[q] = ld(blk * NB + (q < ntail ? q : ntail - 1));
#pragma unroll
    for (int q = 0; q < NB - 1; ++q) {
      if (q < ntail) frame(ya[q], IntTag<1>());
    }
  }
  flush();
  const double dl = (double)X[0], ddl = (double)X[1];
  // finish (float64 duals): b = a ((y - d) / c + K d), K = (1 - r g) / c; eta = c g S1; J = c^2 g / (1 - rho^2)
  if (UNIT) {
    out.b = (double)yprev - K.rg * dl;
    out.db = -(K.drg * dl + K.rg * ddl);
  } else {
    const double ic = 1.0 / c_d;
    out.b = a_d * ic * ((double)yprev - K.rg * dl);
    out.db = -a_d * ic * (K.drg * dl + K.rg * ddl);
  }
  const double c1 = UNIT ? 1.0 : c_d;
  out.eta = s1v * K.cg;
  out.deta = s1d * K.cg + s1v * K.dcg;
  const double iom = 1.0 / (1.0 - K.rho * K.rho);
  const double ccg = c1 * K.cg, dccg = c1 * K.dcg;
  out.J = ccg * iom;
  out.dJ = dccg * iom + ccg * iom * iom * 2.0 * K.rho * K.drho;
  out.ell = -0.5 * ((double)len * (kLog2Pi + log(K.S)) + K.g * s2v);
  out.dell = -0.5 * ((double)len * K.dS * K.g + K.dg * s2v + K.g * s2d);
}

}  // namespace eks
