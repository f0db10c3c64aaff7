// Host-side simulator of the scalar-chain kernels' float32 arithmetic.  TEST INFRASTRUCTURE ONLY:
// it calls the same lane bodies the gfx950 kernels call (eks_amd/csrc/eks_diag_lane.hpp) from
// plain loops, so the chunked-scan numerics can be compared with the float64 oracle on a CPU-only
// box.  It is not a fallback: nothing under eks_amd/ loads it.
#include <algorithm>
#include <vector>

#include "eks_diag_lane.hpp"

using namespace eks;

template <int B>
static void run(int T, int N, int D, bool unit, const float* y, const float* var,
                const DiagModel& M, float* ms, float* Vs_diag) {
  const int nc = (T + B - 1) / B;
  std::vector<Elem<float>> el((size_t)nc * N);
  for (int j = 0; j < nc; ++j)
    for (int n = 0; n < N; ++n) {
      ChainParams<float> p = load_chain_params(M, n);
      const int t0 = j * B, len = std::min(B, T - t0);
      el[(size_t)j * N + n] = unit ? summarize_chunk<B, true>(y, var, N, n, t0, len, p)
                                   : summarize_chunk<B, false>(y, var, N, n, t0, len, p);
    }
  std::vector<float> pm((size_t)nc * N), pP((size_t)nc * N), se((size_t)nc * N), sJ((size_t)nc * N);
  for (int n = 0; n < N; ++n) {
    float m, P;
    load_chain_prior(M, n, m, P);
    for (int j = 0; j < nc; ++j) {
      pm[(size_t)j * N + n] = m;
      pP[(size_t)j * N + n] = P;
      elem_apply(el[(size_t)j * N + n], m, P);
    }
    float eta = 0.f, J = 0.f;
    for (int j = nc - 1; j >= 0; --j) {
      se[(size_t)j * N + n] = eta;
      sJ[(size_t)j * N + n] = J;
      elem_back(el[(size_t)j * N + n], eta, J);
    }
  }
  for (int j = 0; j < nc; ++j)
    for (int n = 0; n < N; ++n) {
      ChainParams<float> p = load_chain_params(M, n);
      const int t0 = j * B, len = std::min(B, T - t0);
      const size_t i = (size_t)j * N + n;
      if (unit)
        replay_chunk<B, true, 0>(y, var, ms, Vs_diag, N, n, n % D, t0, len, p, pm[i], pP[i], se[i], sJ[i]);
      else
        replay_chunk<B, false, 0>(y, var, ms, Vs_diag, N, n, n % D, t0, len, p, pm[i], pP[i], se[i], sJ[i]);
    }
}

extern "C" int sim_diag_smooth(int T, int N, int D, int B, int unit, const float* y,
                               const float* var, const double* m0, const double* S0,
                               const double* A, const double* C, const double* Q, const double* s,
                               float* ms, float* Vs_diag) {
  DiagModel M{m0, S0, A, C, Q, s, D};
  switch (B) {
    case 8: run<8>(T, N, D, unit, y, var, M, ms, Vs_diag); break;
    case 16: run<16>(T, N, D, unit, y, var, M, ms, Vs_diag); break;
    case 32: run<32>(T, N, D, unit, y, var, M, ms, Vs_diag); break;
    case 64: run<64>(T, N, D, unit, y, var, M, ms, Vs_diag); break;
    default: return -1;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// constant-R filter NLL (+ d/dlog s) through the same lane bodies as the GPU kernels
// ---------------------------------------------------------------------------------------------
#include "eks_nll_lane.hpp"

template <typename R, typename RD, int NCL>
static void run_nll(int T, int N, int D, int BN, bool unit, bool conv, const float* y, const double* rconst,
                    const DiagModel& M, const double* s_cand, int n_cand, int per_kp, double* nll,
                    double* dnll) {
  const int K = N / D;
  const int ncn = (T + BN - 1) / BN;
  const int ngrp = (n_cand + NCL - 1) / NCL;
  std::vector<NllElem<R>> el((size_t)ncn * N * ngrp * NCL);
  for (int j = 0; j < ncn; ++j)
    for (int n = 0; n < N; ++n)
      for (int g = 0; g < ngrp; ++g) {
        const int k = n / D, d = n % D;
        const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
        double sq[NCL];
        for (int c = 0; c < NCL; ++c) {
          int ci = std::min(g * NCL + c, n_cand - 1);
          double s = per_kp ? s_cand[(size_t)k * n_cand + ci] : s_cand[ci];
          sq[c] = s * M.Q[dd];
        }
        const int t0 = j * BN, len = std::min(BN, T - t0);
        NllElem<R>* o = &el[(((size_t)j * N + n) * ngrp + g) * NCL];
        if (unit)
          nll_summarize_chunk<R, NCL, true>(RowsByPointer{y + (size_t)t0 * N + n, (size_t)N}, t0, len, rconst[n], M.A[dd], M.C[dd], sq, o, conv);
        else
          nll_summarize_chunk<R, NCL, false>(RowsByPointer{y + (size_t)t0 * N + n, (size_t)N}, t0, len, rconst[n], M.A[dd], M.C[dd], sq, o, conv);
      }
  for (int k = 0; k < K; ++k)
    for (int ci = 0; ci < n_cand; ++ci) {
      RD tot = RD(0.0);
      for (int d = 0; d < D; ++d) {
        const int n = k * D + d;
        const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
        auto get = [&](int j, Elem<RD>& e, RD& ell, double& xr) {
          const NllElem<R>& s = el[(((size_t)j * N + n) * ngrp + ci / NCL) * NCL + ci % NCL];
          xr = (double)s.xref;
          e.A = make_real(RD(), (double)val(s.e.A), (double)der(s.e.A));
          e.b = make_real(RD(), (double)val(s.e.b), (double)der(s.e.b));
          e.C = make_real(RD(), (double)val(s.e.C), (double)der(s.e.C));
          e.eta = make_real(RD(), (double)val(s.e.eta), (double)der(s.e.eta));
          e.J = make_real(RD(), (double)val(s.e.J), (double)der(s.e.J));
          ell = make_real(RD(), s.ell, s.dell);
        };
        tot = tot + nll_assemble<RD>(ncn, M.m0[(size_t)k * D + d], M.S0[dd], get);
      }
      nll[(size_t)k * n_cand + ci] = -val(tot);
      if (dnll) dnll[(size_t)k * n_cand + ci] = -der(tot);
    }
}

// grad: 0 value only (float lanes; the converged-entry summaries of chunks j >= 1 allowed, as in
// diag_nll with the sequential assembly), 1 dual numbers, 2 value only with exact-entry summaries
extern "C" int sim_diag_nll(int T, int N, int D, int BN, int unit, int grad, const float* y,
                            const double* rconst, const double* m0, const double* S0,
                            const double* A, const double* C, const double* Q,
                            const double* s_cand, int n_cand, int per_kp, double* nll,
                            double* dnll) {
  DiagModel M{m0, S0, A, C, Q, nullptr, D};
  if (grad == 1) {
    if (n_cand >= 8) run_nll<Dual, DualD, 8>(T, N, D, BN, unit, false, y, rconst, M, s_cand, n_cand, per_kp, nll, dnll);
    else run_nll<Dual, DualD, 1>(T, N, D, BN, unit, false, y, rconst, M, s_cand, n_cand, per_kp, nll, dnll);
  } else {
    if (n_cand >= 8) run_nll<float, double, 8>(T, N, D, BN, unit, grad == 0, y, rconst, M, s_cand, n_cand, per_kp, nll, nullptr);
    else run_nll<float, double, 1>(T, N, D, BN, unit, grad == 0, y, rconst, M, s_cand, n_cand, per_kp, nll, nullptr);
  }
  return 0;
}


// ---------------------------------------------------------------------------------------------
// the grid kernel of round 4 (diag_nll_grid_kernel): chunk 0 through the general lane body at 4 candidates per
// lane, chunks j >= 1 through nll_lean_chunk at 16 (falling back to the exact-entry summary when a chunk does not
// qualify), the summaries assembled by the sequential walk over a getter that knows both forms.  `n_lean_out`
// receives the number of (chain, chunk, group) units that took the lean path.
// ---------------------------------------------------------------------------------------------
template <bool UNIT>
static void run_nll_lean(int T, int N, int D, int B0, int BN, const float* y, const double* rconst, const DiagModel& M,
                         const double* s_cand, int n_cand, double* nll, int* n_lean_out) {
  constexpr int NC = 16, NH = 4;
  const int K = N / D;
  const int ncn = T <= B0 ? 1 : 1 + (T - B0 + BN - 1) / BN;
  const int ncp = (n_cand + NC - 1) / NC * NC, ng16 = ncp / NC;    // candidates dealt round-robin to ng16 waves
  struct Sum { float A, b, C, eta, J, xr; double ell; };
  std::vector<Sum> el((size_t)ncn * N * ncp);
  std::vector<float> Jc((size_t)N * ncp);
  int n_lean = 0;
  for (int n = 0; n < N; ++n) {
    const int k = n / D, d = n % D;
    const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
    const double q = M.Q[dd], r = rconst[n], a = M.A[dd], c = M.C[dd];
    auto put_full = [&](int j, int ci, const NllElem<float>& o) {
      el[((size_t)j * N + n) * ncp + ci] = Sum{o.e.A, o.e.b, o.e.C, o.e.eta, o.e.J, o.xref, o.ell};
    };
    for (int g = 0; g * NH < n_cand; ++g) {                         // head: chunk 0
      double sq[NH];
      for (int cc = 0; cc < NH; ++cc) sq[cc] = s_cand[std::min(g * NH + cc, n_cand - 1)] * q;
      NllElem<float> o[NH];
      nll_summarize_chunk<float, NH, UNIT>(RowsByPointer{y + n, (size_t)N}, 0, std::min(B0, T), r, a, c, sq, o, false);
      for (int cc = 0; cc < NH; ++cc) {
        const int ci = g * NH + cc;
        if (ci >= n_cand) continue;
        put_full(0, ci, o[cc]);
        const LeanConst lc = lean_const<UNIT>(r, a, c, sq[cc]);
        const float c_cg = UNIT ? lc.cg : (float)c * lc.cg;
        Jc[(size_t)n * ncp + ci] = c_cg / (1.f - lc.rho * lc.rho);
      }
    }
    for (int j = 1; j < ncn; ++j)
      for (int g = 0; g * NC < n_cand; ++g) {
        const int t0 = B0 + (j - 1) * BN, len = std::min(BN, T - t0);
        const RowsByPointer ld{y + (size_t)t0 * N + n, (size_t)N};
        auto cand_of = [&](int cc) { return cc * ng16 + g; };
        auto sqf = [&](int cc) { return s_cand[std::min(cand_of(cc), n_cand - 1)] * q; };
        float stash[4 * NC];
        LeanOut<NC> out;
        const int res = nll_lean_chunk<NC, UNIT>(ld, t0, len, r, a, c, sqf, stash, 1, out);
        if (res) {
          ++n_lean;
          for (int cc = 0; cc < NC; ++cc) {
            const int ci = cand_of(cc);
            if (ci < n_cand)
              el[((size_t)j * N + n) * ncp + ci] =
                  res == 2 ? Sum{out.A[cc], out.B[cc], -1.f, out.Eta[cc], out.J[cc], out.xr, out.Ell[cc]}
                           : Sum{0.f, out.B[cc], -1.f, out.Eta[cc], Jc[(size_t)n * ncp + ci], out.xr, out.Ell[cc]};
          }
        } else {
          for (int h = 0; h < NC / NH; ++h) {
            double sq[NH];
            for (int cc = 0; cc < NH; ++cc) sq[cc] = sqf(h * NH + cc);
            NllElem<float> o[NH];
            nll_summarize_chunk<float, NH, UNIT>(ld, t0, len, r, a, c, sq, o, false);
            for (int cc = 0; cc < NH; ++cc)
              if (cand_of(h * NH + cc) < n_cand) put_full(j, cand_of(h * NH + cc), o[cc]);
          }
        }
      }
  }
  for (int k = 0; k < K; ++k)
    for (int ci = 0; ci < n_cand; ++ci) {
      double tot = 0.0;
      for (int d = 0; d < D; ++d) {
        const int n = k * D + d;
        const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
        auto get = [&](int j, Elem<double>& e, double& ell, double& xr) {
          const Sum& s = el[((size_t)j * N + n) * ncp + ci];
          xr = (double)s.xr;
          e.A = s.A; e.b = s.b; e.C = s.C; e.eta = s.eta; e.J = s.J;
          ell = s.ell;
        };
        tot += nll_assemble<double>(ncn, M.m0[(size_t)k * D + d], M.S0[dd], get);
      }
      nll[(size_t)k * n_cand + ci] = -tot;
    }
  if (n_lean_out) *n_lean_out = n_lean;
}

extern "C" int sim_diag_nll_lean(int T, int N, int D, int B0, int BN, int unit, const float* y, const double* rconst,
                                 const double* m0, const double* S0, const double* A, const double* C, const double* Q,
                                 const double* s_cand, int n_cand, double* nll, int* n_lean) {
  DiagModel M{m0, S0, A, C, Q, nullptr, D};
  if (unit) run_nll_lean<true>(T, N, D, B0, BN, y, rconst, M, s_cand, n_cand, nll, n_lean);
  else run_nll_lean<false>(T, N, D, B0, BN, y, rconst, M, s_cand, n_cand, nll, n_lean);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// round 5: the shared-lag form of the grid kernel (eks_nll_lag.hpp).  A "block" is one chain here: the candidates whose
// steady-state pole is below lag_rho_max are summarised from the chunk's lag sums (lag_summary, float64), the others
// are dealt to four "waves" that run nll_lag_chunk with NP pairs each and take turns at the lag products.  Chunks
// that do not qualify (too early in the sequence, a length that is not a multiple of 32, fewer than 16 fast
// candidates) take the round-4 path.  n_lag_out: (chain, chunk) units that took the lag path.
// ---------------------------------------------------------------------------------------------
#include "eks_nll_lag.hpp"

// one "wave" of the lag block: NP pairs, its turn mask over 16 sets
template <bool UNIT, int NP, typename SQW, typename PUT>
static int lag_wave(const RowsByPointer& ld, int len, double r, double a, double c, int w, unsigned mask, const SQW& sqw,
                    LagKeep<kLagN>& keep, const PUT& put) {
  float stash[6 * NP];
  LeanOut<2 * NP> out;
  for (int k = 0; k < 2 * NP; ++k) out.A[k] = 0.f, out.J[k] = 0.f;
  auto sq = [&](int k) { return sqw(w, k); };
  const int res = nll_lag_chunk<NP, kLagND, UNIT>(ld, len, r, a, c, sq, mask, 16, w == 0, stash, 1, out, keep);
  for (int k = 0; k < 2 * NP; ++k) put(w, k, res, out);
  return res;
}

template <bool UNIT>
static void run_nll_lag(int T, int N, int D, int B0, int BN, const float* y, const double* rconst, const DiagModel& M,
                        const double* s_cand, int n_cand, double* nll, int* n_lag_out) {
  constexpr int NC = 16, NH = 4, NLAG = kLagN;
  const int K = N / D;
  const int ncn = T <= B0 ? 1 : 1 + (T - B0 + BN - 1) / BN;
  const int ncp = (n_cand + NC - 1) / NC * NC, ng16 = ncp / NC;
  struct Sum { double A, b, C, eta, J, xr, ell; };
  std::vector<Sum> el((size_t)ncn * N * ncp);
  std::vector<float> Jc((size_t)N * ncp);
  const double rho_max = lag_rho_max(NLAG);
  int n_lag = 0;
  for (int n = 0; n < N; ++n) {
    const int k = n / D, d = n % D;
    const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
    const double q = M.Q[dd], r = rconst[n], a = M.A[dd], c = M.C[dd];
    auto put_full = [&](int j, int ci, const NllElem<float>& o) {
      el[((size_t)j * N + n) * ncp + ci] = Sum{o.e.A, o.e.b, o.e.C, o.e.eta, o.e.J, o.xref, o.ell};
    };
    for (int g = 0; g * NH < n_cand; ++g) {                         // head: chunk 0
      double sq[NH];
      for (int cc = 0; cc < NH; ++cc) sq[cc] = s_cand[std::min(g * NH + cc, n_cand - 1)] * q;
      NllElem<float> o[NH];
      nll_summarize_chunk<float, NH, UNIT>(RowsByPointer{y + n, (size_t)N}, 0, std::min(B0, T), r, a, c, sq, o, false);
      for (int cc = 0; cc < NH; ++cc) {
        const int ci = g * NH + cc;
        if (ci >= n_cand) continue;
        put_full(0, ci, o[cc]);
        const LeanConst lc = lean_const<UNIT>(r, a, c, sq[cc]);
        const float c_cg = UNIT ? lc.cg : (float)c * lc.cg;
        Jc[(size_t)n * ncp + ci] = c_cg / (1.f - lc.rho * lc.rho);
      }
    }
    // the block's fast set and slow list (the same for every chunk of the chain)
    const double thr = lag_sq_threshold(r, UNIT ? 1.0 : a, UNIT ? 1.0 : c, rho_max);
    std::vector<int> fast(n_cand), slow;
    int nfast = 0;
    double sq_min = 1e300;
    for (int ci = 0; ci < n_cand; ++ci) {
      fast[ci] = s_cand[ci] * q >= thr;
      nfast += fast[ci];
      sq_min = std::min(sq_min, s_cand[ci] * q);
      if (!fast[ci]) slow.push_back(ci);
    }
    // pairs of consecutive places of the slow list go to the four waves round-robin (diag_nll_grid_kernel): P pairs ->
    // P / 4 per wave and one more for the first P % 4; the list is padded with fast candidates (results unused)
    const int nslow = (int)slow.size();
    int npairs = (nslow + 1) / 2;
    if (npairs < 4) npairs = 4;
    for (int ci = 0; ci < n_cand && (int)slow.size() < 2 * npairs; ++ci)
      if (fast[ci]) slow.push_back(ci);
    while ((int)slow.size() < 2 * npairs) slow.push_back(slow.back());
    const int base = npairs / 4, rem = npairs % 4;
    for (int j = 1; j < ncn; ++j) {
      const int t0 = B0 + (j - 1) * BN, len = std::min(BN, T - t0);
      const RowsByPointer ld{y + (size_t)t0 * N + n, (size_t)N};
      // converged entry for EVERY candidate: rho^(2 t0) < 1e-20 <=> |rho| < exp(-23 / t0) <=> s q above its threshold
      const bool qual = sq_min >= lag_sq_threshold(r, UNIT ? 1.0 : a, UNIT ? 1.0 : c, exp(-23.0 / (double)t0));
      const bool lagmode = qual && len % 32 == 0 && len >= 64 && nfast >= 16 && nslow <= 48;
      if (lagmode) {
        ++n_lag;
        LagKeep<NLAG> keep[4];
        for (int w = 0; w < 4; ++w)
          for (int i = 0; i < NLAG; ++i) keep[w].c[i] = 0.0;
        auto place = [&](int w, int kk) { return 2 * ((kk / 2) * 4 + w) + (kk & 1); };
        auto sqw = [&](int w, int kk) { return s_cand[slow[place(w, kk)]] * q; };
        auto put = [&](int w, int kk, int res, const auto& out) {
          const int ci = slow[place(w, kk)];
          if (fast[ci]) return;
          el[((size_t)j * N + n) * ncp + ci] =
              res == 2 && out.A[kk] != 0.f ? Sum{out.A[kk], out.B[kk], -1., out.Eta[kk], out.J[kk], out.xr, out.Ell[kk]}
                                           : Sum{0., out.B[kk], -1., out.Eta[kk], Jc[(size_t)n * ncp + ci], out.xr, out.Ell[kk]};
        };
        for (int w = 0; w < 4; ++w) {
          const int np = base + (w < rem ? 1 : 0);
          const int turns = rem == 0 ? 4 : (w < rem ? rem : rem + 4);
          const int first = rem == 0 ? 4 * w : (w < rem ? rem * w : rem * rem + (rem + 4) * (w - rem));
          const unsigned mask = ((1u << turns) - 1u) << first;
          switch (np) {
            case 1: lag_wave<UNIT, 1>(ld, len, r, a, c, w, mask, sqw, keep[w], put); break;
            case 2: lag_wave<UNIT, 2>(ld, len, r, a, c, w, mask, sqw, keep[w], put); break;
            case 3: lag_wave<UNIT, 3>(ld, len, r, a, c, w, mask, sqw, keep[w], put); break;
            case 4: lag_wave<UNIT, 4>(ld, len, r, a, c, w, mask, sqw, keep[w], put); break;
            case 5: lag_wave<UNIT, 5>(ld, len, r, a, c, w, mask, sqw, keep[w], put); break;
            default: lag_wave<UNIT, 6>(ld, len, r, a, c, w, mask, sqw, keep[w], put); break;
          }
        }
        double cs[NLAG];
        for (int i = 0; i < NLAG; ++i) cs[i] = keep[0].c[i] + keep[1].c[i] + keep[2].c[i] + keep[3].c[i];
        const float xr = UNIT ? ld(0) : ld(0) / (float)c;
        for (int ci = 0; ci < n_cand; ++ci) {
          if (!fast[ci]) continue;
          const LagConst kc = lag_const<UNIT>(r, a, c, s_cand[ci] * q);
          double b, eta, ell;
          lag_summary<NLAG, UNIT>(kc, a, c, len, cs, keep[0].uh, keep[0].ut, keep[0].yl, b, eta, ell);
          el[((size_t)j * N + n) * ncp + ci] = Sum{0., b, -1., eta, kc.Jc, xr, ell};
        }
        continue;
      }
      for (int g = 0; g * NC < n_cand; ++g) {                        // the round-4 path
        auto cand_of = [&](int cc) { return cc * ng16 + g; };
        auto sqf = [&](int cc) { return s_cand[std::min(cand_of(cc), n_cand - 1)] * q; };
        float stash[4 * NC];
        LeanOut<NC> out;
        const int res = nll_lean_chunk<NC, UNIT>(ld, t0, len, r, a, c, sqf, stash, 1, out);
        if (res) {
          for (int cc = 0; cc < NC; ++cc) {
            const int ci = cand_of(cc);
            if (ci < n_cand)
              el[((size_t)j * N + n) * ncp + ci] =
                  res == 2 ? Sum{out.A[cc], out.B[cc], -1., out.Eta[cc], out.J[cc], out.xr, out.Ell[cc]}
                           : Sum{0., out.B[cc], -1., out.Eta[cc], Jc[(size_t)n * ncp + ci], out.xr, out.Ell[cc]};
          }
        } else {
          for (int h = 0; h < NC / NH; ++h) {
            double sq[NH];
            for (int cc = 0; cc < NH; ++cc) sq[cc] = sqf(h * NH + cc);
            NllElem<float> o[NH];
            nll_summarize_chunk<float, NH, UNIT>(ld, t0, len, r, a, c, sq, o, false);
            for (int cc = 0; cc < NH; ++cc)
              if (cand_of(h * NH + cc) < n_cand) put_full(j, cand_of(h * NH + cc), o[cc]);
          }
        }
      }
    }
  }
  for (int k = 0; k < K; ++k)
    for (int ci = 0; ci < n_cand; ++ci) {
      double tot = 0.0;
      for (int d = 0; d < D; ++d) {
        const int n = k * D + d;
        const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
        auto get = [&](int j, Elem<double>& e, double& ell, double& xr) {
          const Sum& s = el[((size_t)j * N + n) * ncp + ci];
          xr = s.xr;
          e.A = s.A; e.b = s.b; e.C = s.C; e.eta = s.eta; e.J = s.J;
          ell = s.ell;
        };
        tot += nll_assemble<double>(ncn, M.m0[(size_t)k * D + d], M.S0[dd], get);
      }
      nll[(size_t)k * n_cand + ci] = -tot;
    }
  if (n_lag_out) *n_lag_out = n_lag;
}

extern "C" int sim_diag_nll_lag(int T, int N, int D, int B0, int BN, int unit, const float* y, const double* rconst,
                                const double* m0, const double* S0, const double* A, const double* C, const double* Q,
                                const double* s_cand, int n_cand, double* nll, int* n_lag) {
  DiagModel M{m0, S0, A, C, Q, nullptr, D};
  if (unit) run_nll_lag<true>(T, N, D, B0, BN, y, rconst, M, s_cand, n_cand, nll, n_lag);
  else run_nll_lag<false>(T, N, D, B0, BN, y, rconst, M, s_cand, n_cand, nll, n_lag);
  return 0;
}


// ---------------------------------------------------------------------------------------------
// round 5: the gradient evaluation without compositions (gf_conv_body of diag_nll_grad_fused_kernel): chunk 0 from a
// known entry state applied to the prior, every later chunk by nll_conv_chunk_dual, chunk j's term from chunk j - 1's b.
// Returns the number of (chain) evaluations that qualified (the others are skipped: nll = dnll = NaN).
// ---------------------------------------------------------------------------------------------
#include "eks_nll_lag.hpp"

template <bool UNIT>
static int run_nll_conv_grad(int T, int N, int D, int BN, const float* y, const double* rconst, const DiagModel& M,
                             const double* s_kp, double* nll, double* dnll) {
  const int K = N / D, ncn = (T + BN - 1) / BN;
  int n_ok = 0;
  for (int k = 0; k < K; ++k) {
    DualD tot(0.0);
    bool ok = ncn > 1;
    for (int d = 0; d < D && ok; ++d) {
      const int n = k * D + d;
      const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
      const double r = rconst[n], a = M.A[dd], c = M.C[dd], sq = s_kp[k] * M.Q[dd];
      const ConvConst KC = conv_const<UNIT>(r, a, c, sq);
      if (!conv_chunk_ok(KC, BN)) { ok = false; break; }
      double sqv[1] = {sq};
      NllElem<Dual> o0[1];
      nll_summarize_chunk<Dual, 1, UNIT>(RowsByPointer{y + n, (size_t)N}, 0, std::min(BN, T), r, a, c, sqv, o0, false);
      const DualD A(o0[0].e.A.v, o0[0].e.A.d), b(o0[0].e.b.v, o0[0].e.b.d), e0(o0[0].e.eta.v, o0[0].e.eta.d),
          J0(o0[0].e.J.v, o0[0].e.J.d), ell(o0[0].ell, o0[0].dell);
      const DualD mr = DualD(M.m0[(size_t)k * D + d] - (double)o0[0].xref), P = DualD(M.S0[dd]);
      const DualD den = DualD(1.0) + J0 * P;
      const DualD inv = rcp(den);
      tot = tot + ell - DualD(0.5) * log_with_rcp(den, inv) + (e0 * mr + DualD(0.5) * e0 * e0 * P - DualD(0.5) * J0 * mr * mr) * inv;
      DualD bprev = A * inv * (mr + P * e0) + b;
      for (int j = 1; j < ncn; ++j) {
        const int t0 = j * BN, len = std::min(BN, T - t0);
        ConvDual o;
        nll_conv_chunk_dual<UNIT>(RowsByPointer{y + (size_t)t0 * N + n, (size_t)N}, len, KC, a, c, o);
        const DualD m2 = bprev - DualD((double)o.xref);
        tot = tot + DualD(o.ell, o.dell) + DualD(o.eta, o.deta) * m2 - DualD(0.5) * DualD(o.J, o.dJ) * m2 * m2;
        bprev = DualD(o.b, o.db);
      }
    }
    if (ok) {
      nll[k] = -tot.v;
      dnll[k] = -tot.d;
      ++n_ok;
    } else {
      nll[k] = dnll[k] = std::nan("");
    }
  }
  return n_ok;
}

extern "C" int sim_diag_nll_conv_grad(int T, int N, int D, int BN, int unit, const float* y, const double* rconst,
                                      const double* m0, const double* S0, const double* A, const double* C, const double* Q,
                                      const double* s_kp, double* nll, double* dnll) {
  DiagModel M{m0, S0, A, C, Q, nullptr, D};
  return unit ? run_nll_conv_grad<true>(T, N, D, BN, y, rconst, M, s_kp, nll, dnll)
              : run_nll_conv_grad<false>(T, N, D, BN, y, rconst, M, s_kp, nll, dnll);
}
