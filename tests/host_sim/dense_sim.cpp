// Host-side simulator of the general (D, O) lane bodies.  TEST INFRASTRUCTURE ONLY: it calls the
// same headers the gfx950 kernels include (eks_amd/csrc/eks_dense_lane.hpp) from plain loops, so
// the chunked element algebra and the dual-number sensitivities of the AR(1) loss can be compared
// with the float64 oracle on a CPU-only box.  It is not a fallback: nothing under eks_amd/ loads it.
#include <algorithm>
#include <vector>

#include "eks_dense_lane.hpp"

using namespace eks;

// nll and (n_tan > 0) dnll[i] along the tangents (da[i], dq[i]) of one chain, chunks of B frames:
// what loss_chunks_kernel + loss_reduce_kernel compute (eks_dense.hip); tree = 0 applies the chunk
// elements one after the other instead of composing them first
template <typename S, int D>
static S ar1_ll(int T, int O, int B, bool tree, const float* y, const float* var, const DenseModelPtrs& M,
                const double* a, const double* q, const double* da, const double* dq) {
  DynDiag<S, D> dyn;
  load_ar1_dynamics<S, D>(a, q, da, dq, 0, dyn.a, dyn.q);
  const ObsNoise R{var, nullptr};
  Vec<double, D> m0;
  Mat<double, D> P0;
  load_prior<D>(M, 0, m0, P0);
  Vec<S, D> m;
  Mat<S, D> P;
  for (int i = 0; i < D; ++i) {
    m.a[i] = S(m0.a[i]);
    for (int j = 0; j < D; ++j) P.a[i][j] = S(P0.a[i][j]);
  }
  S ll = loss_first_frame<S, D>(y, R, 1, O, 0, M, m, P);
  if (T == 1) return ll;
  std::vector<DElem<S, D>> el;
  for (int t0 = 1; t0 < T; t0 += B) {
    const DElem<S, D> e = loss_summarize_chunk<S, D>(y, R, 1, O, 0, t0, std::min(B, T - t0), M, dyn);
    if (tree)
      el.push_back(e);
    else
      ll = ll + delem_apply(e, m, P);
  }
  if (!tree) return ll;
  // the kernels' order of composition: groups of 64, pairwise doubling inside a group (A1/A2)
  while (el.size() > 1) {
    std::vector<DElem<S, D>> next;
    for (size_t j0 = 0; j0 < el.size(); j0 += 64) {
      const size_t n = std::min<size_t>(64, el.size() - j0);
      for (size_t half = 1; half < n; half <<= 1)
        for (size_t i = 0; i + half < n; i += 2 * half) el[j0 + i] = delem_combine(el[j0 + i], el[j0 + i + half]);
      next.push_back(el[j0]);
    }
    el.swap(next);
  }
  return ll + delem_apply(el[0], m, P);
}

extern "C" int sim_ar1_nll(int T, int D, int O, int B, int tree, const float* y, const float* var,
                           const double* m0, const double* S0, const double* C, const double* a,
                           const double* q, const double* da, const double* dq, int n_tan,
                           double* nll, double* dnll) {
  if (D != 3) return -3;
  const DenseModelPtrs M{m0, S0, nullptr, C, nullptr};
  if (n_tan == 0) {
    *nll = -ar1_ll<double, 3>(T, O, B, tree != 0, y, var, M, a, q, nullptr, nullptr);
    return 0;
  }
  for (int i = 0; i < n_tan; ++i) {
    const DualD ll = ar1_ll<DualD, 3>(T, O, B, tree != 0, y, var, M, a, q, da + i * D, dq + i * D);
    *nll = -ll.v;
    dnll[i] = -ll.d;
  }
  return 0;
}

// The general (D, O) smoother as eks_dense.hip arranges it, from plain loops: chunk elements
// (dense_smooth_element), the scan as sequential applies / pull-backs over the chunk elements,
// then the exact replay of every chunk (dense_replay_chunk).  ms [T][K][D], Vs [T][K][D][D].
template <int D>
static void dense_smooth_sim(int T, int K, int O, int B, const float* y, const float* var,
                             const DenseModelPtrs& M, const double* s, float* ms, float* Vs) {
  constexpr int REC = D + D * D;
  const int nc = (T + B - 1) / B;
  std::vector<double> filt((size_t)B * REC);
  for (int k = 0; k < K; ++k) {
    Mat<double, D> F, sQ;
    bool fid;
    load_dynamics<double, D>(M, k, s[k], F, sQ, fid);
    std::vector<DElem<double, D>> el(nc);
    for (int j = 0; j < nc; ++j)
      el[j] = dense_smooth_element<D>(y, var, K, O, k, j * B, std::min(B, T - j * B), M, F, sQ, fid);
    std::vector<Vec<double, D>> pm(nc), se(nc);
    std::vector<Mat<double, D>> pP(nc), sJ(nc);
    Vec<double, D> m;
    Mat<double, D> P;
    load_prior<D>(M, k, m, P);
    belief_update_frame<D>(y, var, K, O, k, 0, M, m, P);
    for (int j = 0; j < nc; ++j) {
      pm[j] = m;
      pP[j] = P;
      delem_apply(el[j], m, P);
    }
    Vec<double, D> eta = vec_zero<double, D>();
    Mat<double, D> J = mat_zero<double, D>();
    for (int j = nc - 1; j >= 0; --j) {
      se[j] = eta;
      sJ[j] = J;
      delem_back(el[j], eta, J);
    }
    for (int j = 0; j < nc; ++j) {
      if (j == 0) load_prior<D>(M, k, pm[0], pP[0]);
      dense_replay_chunk<D>(y, var, K, O, k, j * B, std::min(B, T - j * B), M, F, sQ, fid, pm[j], pP[j],
                            se[j], sJ[j], filt.data(), ms, Vs, false);
    }
  }
}

extern "C" int sim_dense_smooth(int T, int K, int D, int O, int B, const float* y, const float* var,
                                const double* m0, const double* S0, const double* A, const double* C,
                                const double* Q, const double* s, float* ms, float* Vs) {
  const DenseModelPtrs M{m0, S0, A, C, Q};
  switch (D) {
    case 2: dense_smooth_sim<2>(T, K, O, B, y, var, M, s, ms, Vs); return 0;
    case 3: dense_smooth_sim<3>(T, K, O, B, y, var, M, s, ms, Vs); return 0;
    case 4: dense_smooth_sim<4>(T, K, O, B, y, var, M, s, ms, Vs); return 0;
    default: return -3;
  }
}

// eks_nll on the general (D, O) path: constant R, process noise s Q, d nll / d log s by dual
// numbers; chunk elements composed in the kernels' tree order (loss_chunks / loss_reduce, MODE 1).
template <typename S, int D>
static S scaled_ll(int T, int K, int O, int B, int k, const float* y, const double* rconst,
                   const DenseModelPtrs& M, double s) {
  DynFull<S, D> dyn;
  load_dynamics<S, D>(M, k, make_real(S(), s, s), dyn.F, dyn.sQ, dyn.f_identity);
  const ObsNoise R{nullptr, rconst};
  Vec<double, D> m0;
  Mat<double, D> P0;
  load_prior<D>(M, k, m0, P0);
  Vec<S, D> m;
  Mat<S, D> P;
  for (int i = 0; i < D; ++i) {
    m.a[i] = S(m0.a[i]);
    for (int j = 0; j < D; ++j) P.a[i][j] = S(P0.a[i][j]);
  }
  S ll = loss_first_frame<S, D>(y, R, K, O, k, M, m, P);
  if (T == 1) return ll;
  std::vector<DElem<S, D>> el;
  for (int t0 = 1; t0 < T; t0 += B)
    el.push_back(loss_summarize_chunk<S, D>(y, R, K, O, k, t0, std::min(B, T - t0), M, dyn));
  while (el.size() > 1) {
    std::vector<DElem<S, D>> next;
    for (size_t j0 = 0; j0 < el.size(); j0 += 64) {
      const size_t n = std::min<size_t>(64, el.size() - j0);
      for (size_t half = 1; half < n; half <<= 1)
        for (size_t i = 0; i + half < n; i += 2 * half) el[j0 + i] = delem_combine(el[j0 + i], el[j0 + i + half]);
      next.push_back(el[j0]);
    }
    el.swap(next);
  }
  return ll + delem_apply(el[0], m, P);
}

extern "C" int sim_dense_nll(int T, int K, int D, int O, int B, const float* y, const double* rconst,
                             const double* m0, const double* S0, const double* A, const double* C,
                             const double* Q, const double* s, double* nll, double* dnll) {
  const DenseModelPtrs M{m0, S0, A, C, Q};
  for (int k = 0; k < K; ++k) {
    DualD ll;
    switch (D) {
      case 2: ll = scaled_ll<DualD, 2>(T, K, O, B, k, y, rconst, M, s[k]); break;
      case 3: ll = scaled_ll<DualD, 3>(T, K, O, B, k, y, rconst, M, s[k]); break;
      case 4: ll = scaled_ll<DualD, 4>(T, K, O, B, k, y, rconst, M, s[k]); break;
      default: return -3;
    }
    nll[k] = -ll.v;
    dnll[k] = -ll.d;
  }
  return 0;
}

// ---- extended filter with pinhole cameras (eks_ekf_smooth) from plain loops ------------------
extern "C" void sim_pinhole(const double* cam, const double* X, double* uv, double* J) {
  double Jm[2][3];
  pinhole_project_jac(cam, X, uv, Jm);
  for (int a = 0; a < 2; ++a)
    for (int i = 0; i < 3; ++i) J[a * 3 + i] = Jm[a][i];
}

// One chain per data keypoint.  Filter sweeps (scan over elements linearised at xlin, then the
// per-chunk extended replay that rewrites xlin) until no linearisation point moves by more than
// tol, then one sweep with the backward pass.  Returns the number of filter sweeps.
extern "C" int sim_ekf_smooth(int T, int K, int n_cams, int B, const float* y, const float* var,
                              const double* rconst, const double* m0, const double* S0,
                              const double* A, const double* Q, const double* s, const double* cams,
                              double* xlin, int max_sweeps, double tol, float* ms, float* Vs,
                              double* nll, double* last_resid) {
  constexpr int D = 3, REC = D + D * D;
  const int O = 2 * n_cams, nc = (T + B - 1) / B;
  const DenseModelPtrs M{m0, S0, A, nullptr, Q};
  const PinholeObs obs{y, ObsNoise{var, rconst}, K, O, T, cams, xlin};
  std::vector<double> filt((size_t)B * REC);
  int sweeps = 0;
  double worst = 0.0;
  for (int pass = 0; pass <= max_sweeps; ++pass) {
    const bool final_pass = pass == max_sweeps || (pass > 0 && worst <= tol);
    worst = 0.0;
    for (int k = 0; k < K; ++k) {
      Mat<double, D> F, sQ;
      bool fid;
      load_dynamics<double, D>(M, k, s[k], F, sQ, fid);
      std::vector<DElem<double, D>> el(nc);
      for (int j = 0; j < nc; ++j)
        el[j] = dense_smooth_element_obs<D>(obs, k, j * B, std::min(B, T - j * B), F, sQ, fid);
      std::vector<Vec<double, D>> pm(nc), se(nc);
      std::vector<Mat<double, D>> pP(nc), sJ(nc);
      Vec<double, D> m;
      Mat<double, D> P;
      load_prior<D>(M, k, m, P);
      double xl[D] = {m.a[0], m.a[1], m.a[2]};
      belief_update_obs<D>(obs, k, 0, xl, m, P);
      for (int j = 0; j < nc; ++j) {
        pm[j] = m;
        pP[j] = P;
        delem_apply(el[j], m, P);
      }
      Vec<double, D> eta = vec_zero<double, D>();
      Mat<double, D> J = mat_zero<double, D>();
      for (int j = nc - 1; j >= 0; --j) {
        se[j] = eta;
        sJ[j] = J;
        delem_back(el[j], eta, J);
      }
      double ll_k = 0.0;
      for (int j = 0; j < nc; ++j) {
        if (j == 0) load_prior<D>(M, k, pm[0], pP[0]);
        double ll = 0.0, ch = 0.0;
        dense_replay_chunk_obs<D, true>(obs, K, k, j * B, std::min(B, T - j * B), F, sQ, fid, pm[j],
                                        pP[j], se[j], sJ[j], filt.data(), final_pass ? ms : nullptr,
                                        Vs, false, xlin + ((size_t)k * T + (size_t)j * B) * D, &ll, &ch);
        ll_k += ll;
        worst = std::max(worst, ch);
      }
      nll[k] = -ll_k;
    }
    if (final_pass) break;
    ++sweeps;
  }
  *last_resid = worst;
  return sweeps;
}
