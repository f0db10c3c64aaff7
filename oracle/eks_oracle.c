/* C twin of oracle/eks_oracle.py (float64).  TEST INFRASTRUCTURE ONLY: second opinion for the
 * NumPy oracle and the timed CPU baseline of bench.py ("cpu_baseline.kind": "port").  Nothing
 * under eks_amd/ links or loads it.  PARITY UNPINNED w.r.t. upstream numbers for the same reason
 * as the NumPy oracle (dynamax/jax absent; see the header of eks_oracle.py).
 *
 * Follows the recursion the reference runs through dynamax, general
 * small matrices exactly like the reference does even for the diagonal singlecam model:
 *   filter  : update-then-predict, S = C P C' + R, gain by Cholesky solve, P - K S K', symmetrise
 *             (eks/core.py:290, :469, :648 call sites of extended_kalman_filter / _smoother)
 *   smoother: RTS with G = P_f A' (A P_f A' + sQ)^-1
 *   loss    : nll = -marginal_loglik, non-finite -> 1e12 (eks/core.py:640-650)
 * Keypoints are independent (vmap at eks/core.py:293): OpenMP over keypoints.
 *
 * Layout (keypoint-major like the reference's JAX arrays): y [K][T][O]; R diag [K][T][O] or
 * [K][O] (r_const); ms [K][T][D]; Vs [K][T][D][D].
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXD 8
#define MAXO 16
#define LOG2PI 1.8378770664093454835606594728112

/* lower Cholesky of n x n SPD a (row-major, leading dim ld) in place; returns 0 on success */
static int chol(double* a, int n, int ld) {
  for (int j = 0; j < n; ++j) {
    double s = a[j * ld + j];
    for (int k = 0; k < j; ++k) s -= a[j * ld + k] * a[j * ld + k];
    if (!(s > 0.0)) return 1;
    const double l = sqrt(s);
    a[j * ld + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double t = a[i * ld + j];
      for (int k = 0; k < j; ++k) t -= a[i * ld + k] * a[j * ld + k];
      a[i * ld + j] = t / l;
    }
  }
  return 0;
}
/* solve (L L') x = b in place */
static void chol_solve(const double* L, int n, int ld, double* b) {
  for (int i = 0; i < n; ++i) {
    double t = b[i];
    for (int k = 0; k < i; ++k) t -= L[i * ld + k] * b[k];
    b[i] = t / L[i * ld + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double t = b[i];
    for (int k = i + 1; k < n; ++k) t -= L[k * ld + i] * b[k];
    b[i] = t / L[i * ld + i];
  }
}

/* one keypoint: filter (+ optional RTS).  Returns log-likelihood. */
static double one_keypoint(int T, int D, int O, const double* y, const double* Rd, int r_const,
                           const double* m0, const double* S0, const double* A, const double* C,
                           const double* Q, double s, double* mf, double* Pf, double* ms,
                           double* Vs) {
  double m[MAXD], P[MAXD * MAXD], sQ[MAXD * MAXD];
  double S[MAXO * MAXO], PCt[MAXD * MAXO], Kg[MAXD * MAXO], e[MAXO], tmp[MAXO];
  double ll = 0.0;
  memcpy(m, m0, sizeof(double) * D);
  memcpy(P, S0, sizeof(double) * D * D);
  for (int i = 0; i < D * D; ++i) sQ[i] = s * Q[i];
  for (int t = 0; t < T; ++t) {
    const double* yt = y + (size_t)t * O;
    const double* rt = r_const ? Rd : Rd + (size_t)t * O;
    /* PCt = P C' (D x O); S = C PCt + R */
    for (int i = 0; i < D; ++i)
      for (int o = 0; o < O; ++o) {
        double a = 0.0;
        for (int k = 0; k < D; ++k) a += P[i * D + k] * C[o * D + k];
        PCt[i * O + o] = a;
      }
    for (int o = 0; o < O; ++o) {
      double pred = 0.0;
      for (int k = 0; k < D; ++k) pred += C[o * D + k] * m[k];
      e[o] = yt[o] - pred;
      for (int p = 0; p < O; ++p) {
        double a = 0.0;
        for (int k = 0; k < D; ++k) a += C[o * D + k] * PCt[k * O + p];
        S[o * O + p] = a + (o == p ? rt[o] : 0.0);
      }
    }
    double Sfull[MAXO * MAXO];
    memcpy(Sfull, S, sizeof(double) * O * O);
    if (chol(S, O, O)) return NAN;
    double logdet = 0.0;
    for (int o = 0; o < O; ++o) logdet += 2.0 * log(S[o * O + o]);
    memcpy(tmp, e, sizeof(double) * O);
    chol_solve(S, O, O, tmp);
    double quad = 0.0;
    for (int o = 0; o < O; ++o) quad += e[o] * tmp[o];
    ll += -0.5 * (O * LOG2PI + logdet + quad);
    /* K = PCt S^-1 : solve per row of PCt (S symmetric) */
    for (int i = 0; i < D; ++i) {
      memcpy(tmp, PCt + i * O, sizeof(double) * O);
      chol_solve(S, O, O, tmp);
      memcpy(Kg + i * O, tmp, sizeof(double) * O);
    }
    /* m += K e ; P -= K S K' */
    for (int i = 0; i < D; ++i) {
      double a = 0.0;
      for (int o = 0; o < O; ++o) a += Kg[i * O + o] * e[o];
      m[i] += a;
    }
    double KS[MAXD * MAXO];
    for (int i = 0; i < D; ++i)
      for (int o = 0; o < O; ++o) {
        double a = 0.0;
        for (int p = 0; p < O; ++p) a += Kg[i * O + p] * Sfull[p * O + o];
        KS[i * O + o] = a;
      }
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        double a = 0.0;
        for (int o = 0; o < O; ++o) a += KS[i * O + o] * Kg[j * O + o];
        P[i * D + j] -= a;
      }
    for (int i = 0; i < D; ++i)
      for (int j = i + 1; j < D; ++j) {
        const double v = 0.5 * (P[i * D + j] + P[j * D + i]);
        P[i * D + j] = v;
        P[j * D + i] = v;
      }
    if (mf) {
      memcpy(mf + (size_t)t * D, m, sizeof(double) * D);
      memcpy(Pf + (size_t)t * D * D, P, sizeof(double) * D * D);
    }
    /* predict */
    double mn[MAXD], AP[MAXD * MAXD];
    for (int i = 0; i < D; ++i) {
      double a = 0.0;
      for (int k = 0; k < D; ++k) a += A[i * D + k] * m[k];
      mn[i] = a;
      for (int j = 0; j < D; ++j) {
        double b = 0.0;
        for (int k = 0; k < D; ++k) b += A[i * D + k] * P[k * D + j];
        AP[i * D + j] = b;
      }
    }
    memcpy(m, mn, sizeof(double) * D);
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        double b = 0.0;
        for (int k = 0; k < D; ++k) b += AP[i * D + k] * A[j * D + k];
        P[i * D + j] = b + sQ[i * D + j];
      }
  }
  if (!ms) return ll;
  /* RTS */
  memcpy(ms + (size_t)(T - 1) * D, mf + (size_t)(T - 1) * D, sizeof(double) * D);
  memcpy(Vs + (size_t)(T - 1) * D * D, Pf + (size_t)(T - 1) * D * D, sizeof(double) * D * D);
  for (int t = T - 2; t >= 0; --t) {
    const double* mft = mf + (size_t)t * D;
    const double* Pft = Pf + (size_t)t * D * D;
    double AP[MAXD * MAXD], Sp[MAXD * MAXD], G[MAXD * MAXD], mp[MAXD], col[MAXD];
    for (int i = 0; i < D; ++i) {
      double a = 0.0;
      for (int k = 0; k < D; ++k) a += A[i * D + k] * mft[k];
      mp[i] = a;
      for (int j = 0; j < D; ++j) {
        double b = 0.0;
        for (int k = 0; k < D; ++k) b += A[i * D + k] * Pft[k * D + j];
        AP[i * D + j] = b; /* A Pf */
      }
    }
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        double b = 0.0;
        for (int k = 0; k < D; ++k) b += AP[i * D + k] * A[j * D + k];
        Sp[i * D + j] = b + sQ[i * D + j];
      }
    double Spf[MAXD * MAXD];
    memcpy(Spf, Sp, sizeof(double) * D * D);
    if (chol(Sp, D, D)) return NAN;
    /* G' = Sp^-1 (A Pf): column j of G' = solve(Sp, column j of A Pf) -> G[j][:] */
    for (int j = 0; j < D; ++j) {
      for (int i = 0; i < D; ++i) col[i] = AP[i * D + j];
      chol_solve(Sp, D, D, col);
      for (int i = 0; i < D; ++i) G[j * D + i] = col[i];
    }
    const double* msn = ms + (size_t)(t + 1) * D;
    const double* Vsn = Vs + (size_t)(t + 1) * D * D;
    double* mst = ms + (size_t)t * D;
    double* Vst = Vs + (size_t)t * D * D;
    for (int i = 0; i < D; ++i) {
      double a = 0.0;
      for (int k = 0; k < D; ++k) a += G[i * D + k] * (msn[k] - mp[k]);
      mst[i] = mft[i] + a;
    }
    double GD[MAXD * MAXD];
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        double b = 0.0;
        for (int k = 0; k < D; ++k) b += G[i * D + k] * (Vsn[k * D + j] - Spf[k * D + j]);
        GD[i * D + j] = b;
      }
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        double b = 0.0;
        for (int k = 0; k < D; ++k) b += GD[i * D + k] * G[j * D + k];
        Vst[i * D + j] = Pft[i * D + j] + b;
      }
  }
  return ll;
}

int eksc_smooth(int K, int T, int D, int O, const double* y, const double* Rd, int r_const,
                const double* m0, const double* S0, const double* A, const double* C,
                const double* Q, const double* s, double* ms, double* Vs, double* nll,
                int nthreads) {
  if (D > MAXD || O > MAXO) return -3;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int k = 0; k < K; ++k) {
    double* mf = (double*)malloc(sizeof(double) * (size_t)T * D);
    double* Pf = (double*)malloc(sizeof(double) * (size_t)T * D * D);
    const double* Rk = r_const ? Rd + (size_t)k * O : Rd + (size_t)k * T * O;
    const double ll = one_keypoint(T, D, O, y + (size_t)k * T * O, Rk, r_const, m0 + (size_t)k * D,
                                   S0 + (size_t)k * D * D, A + (size_t)k * D * D,
                                   C + (size_t)k * O * D, Q + (size_t)k * D * D, s[k], mf, Pf,
                                   ms + (size_t)k * T * D, Vs + (size_t)k * T * D * D);
    if (nll) nll[k] = isfinite(ll) ? -ll : 1e12;
    free(mf);
    free(Pf);
  }
  return 0;
}

/* nll[k][c] for candidates s_cand[c] (shared grid) with constant R [K][O] */
int eksc_nll_grid(int K, int T, int D, int O, const double* y, const double* Rc, const double* m0,
                  const double* S0, const double* A, const double* C, const double* Q,
                  const double* s_cand, int n_cand, double* nll, int nthreads) {
  if (D > MAXD || O > MAXO) return -3;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int k = 0; k < K; ++k)
    for (int c = 0; c < n_cand; ++c) {
      const double ll = one_keypoint(T, D, O, y + (size_t)k * T * O, Rc + (size_t)k * O, 1,
                                     m0 + (size_t)k * D, S0 + (size_t)k * D * D,
                                     A + (size_t)k * D * D, C + (size_t)k * O * D,
                                     Q + (size_t)k * D * D, s_cand[c], NULL, NULL, NULL, NULL);
      nll[(size_t)k * n_cand + c] = isfinite(ll) ? -ll : 1e12;
    }
  return 0;
}

/* ---- directional derivatives of the filter log-likelihood by complex-step differentiation:
 * the same update-then-predict recursion in complex arithmetic with A + i h dA, Q + i h dQ
 * (h = 1e-30, no conjugations anywhere), d ll = Im(ll) / h to machine precision.  Serves the
 * loss of the pupil smoother, whose two parameters enter A and Q (eks/ibl_pupil_smoother.py:
 * 540-552); an independent route to the number the NumPy oracle gets by forward sensitivities.
 * R diagonal, time-varying [T][O] (r_const = 0) or constant [O]. */
typedef double complex cplx;

static int cchol(cplx* a, int n) { /* lower factor in place, plain (non-Hermitian) transposes */
  for (int j = 0; j < n; ++j) {
    cplx d = a[j * n + j];
    for (int k = 0; k < j; ++k) d -= a[j * n + k] * a[j * n + k];
    if (!(creal(d) > 0.0)) return 1;
    d = csqrt(d);
    a[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      cplx v = a[i * n + j];
      for (int k = 0; k < j; ++k) v -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = v / d;
    }
  }
  return 0;
}

static void cchol_solve(const cplx* L, int n, cplx* b) {
  for (int i = 0; i < n; ++i) {
    cplx v = b[i];
    for (int k = 0; k < i; ++k) v -= L[i * n + k] * b[k];
    b[i] = v / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    cplx v = b[i];
    for (int k = i + 1; k < n; ++k) v -= L[k * n + i] * b[k];
    b[i] = v / L[i * n + i];
  }
}

static cplx filter_ll_cplx(int T, int D, int O, const double* y, const double* Rd, int r_const,
                           const double* m0, const double* S0, const cplx* A, const double* C,
                           const cplx* Q) {
  cplx m[MAXD], P[MAXD * MAXD], S[MAXO * MAXO], PCt[MAXD * MAXO], Kg[MAXD * MAXO], e[MAXO],
      tmp[MAXO], ll = 0.0;
  for (int i = 0; i < D; ++i) m[i] = m0[i];
  for (int i = 0; i < D * D; ++i) P[i] = S0[i];
  for (int t = 0; t < T; ++t) {
    const double* yt = y + (size_t)t * O;
    const double* rt = r_const ? Rd : Rd + (size_t)t * O;
    for (int i = 0; i < D; ++i)
      for (int o = 0; o < O; ++o) {
        cplx a = 0.0;
        for (int k = 0; k < D; ++k) a += P[i * D + k] * C[o * D + k];
        PCt[i * O + o] = a;
      }
    for (int o = 0; o < O; ++o) {
      cplx pred = 0.0;
      for (int k = 0; k < D; ++k) pred += C[o * D + k] * m[k];
      e[o] = yt[o] - pred;
      for (int p = 0; p < O; ++p) {
        cplx a = 0.0;
        for (int k = 0; k < D; ++k) a += C[o * D + k] * PCt[k * O + p];
        S[o * O + p] = a + (o == p ? rt[o] : 0.0);
      }
    }
    if (cchol(S, O)) return NAN;
    cplx logdet = 0.0, quad = 0.0;
    for (int o = 0; o < O; ++o) logdet += 2.0 * clog(S[o * O + o]);
    memcpy(tmp, e, sizeof(cplx) * O);
    cchol_solve(S, O, tmp);
    for (int o = 0; o < O; ++o) quad += e[o] * tmp[o];
    ll += -0.5 * (O * LOG2PI + logdet + quad);
    for (int i = 0; i < D; ++i) {   /* K = P C' S^-1, row by row */
      memcpy(tmp, PCt + i * O, sizeof(cplx) * O);
      cchol_solve(S, O, tmp);
      memcpy(Kg + i * O, tmp, sizeof(cplx) * O);
    }
    for (int i = 0; i < D; ++i) {
      cplx a = 0.0;
      for (int o = 0; o < O; ++o) a += Kg[i * O + o] * e[o];
      m[i] += a;
    }
    for (int i = 0; i < D; ++i)     /* P - K S K' = P - K (P C')' */
      for (int j = 0; j < D; ++j) {
        cplx a = 0.0;
        for (int o = 0; o < O; ++o) a += Kg[i * O + o] * PCt[j * O + o];
        P[i * D + j] -= a;
      }
    for (int i = 0; i < D; ++i)
      for (int j = i + 1; j < D; ++j) {
        const cplx v = 0.5 * (P[i * D + j] + P[j * D + i]);
        P[i * D + j] = v;
        P[j * D + i] = v;
      }
    cplx mn[MAXD], AP[MAXD * MAXD];
    for (int i = 0; i < D; ++i) {
      cplx a = 0.0;
      for (int k = 0; k < D; ++k) a += A[i * D + k] * m[k];
      mn[i] = a;
      for (int j = 0; j < D; ++j) {
        cplx b = 0.0;
        for (int k = 0; k < D; ++k) b += A[i * D + k] * P[k * D + j];
        AP[i * D + j] = b;
      }
    }
    for (int i = 0; i < D; ++i) {
      m[i] = mn[i];
      for (int j = 0; j < D; ++j) {
        cplx b = 0.0;
        for (int k = 0; k < D; ++k) b += AP[i * D + k] * A[j * D + k];
        P[i * D + j] = b + Q[i * D + j];
      }
    }
  }
  return ll;
}

/* nll and dnll[d] = directional derivative of nll along (dA[d], dQ[d]), d < n_dir */
int eksc_nll_directional(int T, int D, int O, const double* y, const double* Rd, int r_const,
                         const double* m0, const double* S0, const double* A, const double* C,
                         const double* Q, const double* dA, const double* dQ, int n_dir,
                         double* nll, double* dnll) {
  if (D > MAXD || O > MAXO) return -3;
  const double h = 1e-30;
  cplx Ac[MAXD * MAXD], Qc[MAXD * MAXD];
  for (int d = 0; d < (n_dir > 0 ? n_dir : 1); ++d) {
    for (int i = 0; i < D * D; ++i) {
      Ac[i] = A[i] + (n_dir > 0 ? I * (h * dA[(size_t)d * D * D + i]) : 0.0);
      Qc[i] = Q[i] + (n_dir > 0 ? I * (h * dQ[(size_t)d * D * D + i]) : 0.0);
    }
    const cplx ll = filter_ll_cplx(T, D, O, y, Rd, r_const, m0, S0, Ac, C, Qc);
    *nll = -creal(ll);
    if (n_dir > 0) dnll[d] = -cimag(ll) / h;
  }
  return 0;
}

/* ---- the DIAGONAL model as scalar chains (bench.py's like-for-like CPU baseline) --------------------------------
 * With A, C, Q, S0 diagonal and D == O (singlecam: eks/singlecam_smoother.py:246-284) a keypoint is D independent
 * scalar chains; the reference does not exploit this (it runs dynamax's general matrices), the GPU path does.  This
 * is the same scalar recursion on the host: filter in the product forms P r / (P + r), RTS backwards, float64, the
 * filtered beliefs kept in two T-long arrays per chain.  OpenMP over chains.  ms [K][T][D], Vd [K][T][D] (the
 * diagonal of the covariance; the caller expands it), nll [K] (sum over the keypoint's chains).
 * eksc_nll_grid_diag: the constant-R loss for n_cand candidates (the candidates of a chain advance side by side in the
 * frame loop so that y is read once, as in the GPU kernel). */
int eksc_smooth_diag(int K, int T, int D, const double* y, const double* Rd, int r_const, const double* m0,
                     const double* S0, const double* A, const double* C, const double* Q, const double* s,
                     double* ms, double* Vd, double* nll, int nthreads) {
  if (K <= 0 || T <= 0 || D <= 0 || D > MAXD) return 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  for (int k = 0; k < K; ++k) nll[k] = 0.0;
  int fail = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int n = 0; n < K * D; ++n) {
    const int k = n / D, d = n % D;
    const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
    const double a = A[dd], c = C[dd], sq = s[k] * Q[dd];
    double* mf = (double*)malloc(sizeof(double) * 2 * (size_t)T);
    if (!mf) { fail = 1; continue; }
    double* Pf = mf + T;
    double m = m0[(size_t)k * D + d], P = S0[dd], ll = 0.0;
    for (int t = 0; t < T; ++t) {
      const double yt = y[((size_t)k * T + t) * D + d];
      double r = r_const ? Rd[(size_t)k * D + d] : Rd[((size_t)k * T + t) * D + d];
      if (!(r > 1e-12)) r = 1e-12;
      const double S = c * c * P + r, g = 1.0 / S, e = yt - c * m;
      ll += -0.5 * (LOG2PI + log(S) + e * e * g);
      const double Kg = P * c * g;
      m += Kg * e;
      P = P * r * g;
      mf[t] = m;
      Pf[t] = P;
      m = a * m;
      P = a * a * P + sq;
    }
    double msn = mf[T - 1], Vsn = Pf[T - 1];
    ms[((size_t)k * T + (T - 1)) * D + d] = msn;
    Vd[((size_t)k * T + (T - 1)) * D + d] = Vsn;
    for (int t = T - 2; t >= 0; --t) {
      const double Pp = a * a * Pf[t] + sq, G = Pf[t] * a / Pp;
      msn = mf[t] + G * (msn - a * mf[t]);
      Vsn = Pf[t] * sq / Pp + G * G * Vsn;
      ms[((size_t)k * T + t) * D + d] = msn;
      Vd[((size_t)k * T + t) * D + d] = Vsn;
    }
    free(mf);
#pragma omp atomic
    nll[k] -= ll;
  }
  return fail;
}

int eksc_nll_grid_diag(int K, int T, int D, const double* y, const double* Rc, const double* m0, const double* S0,
                       const double* A, const double* C, const double* Q, const double* s_cand, int n_cand,
                       double* nll, int nthreads) {
  if (K <= 0 || T <= 0 || D <= 0 || D > MAXD || n_cand <= 0 || n_cand > 256) return 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  for (size_t i = 0; i < (size_t)K * n_cand; ++i) nll[i] = 0.0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int n = 0; n < K * D; ++n) {
    const int k = n / D, d = n % D;
    const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
    const double a = A[dd], c = C[dd], q = Q[dd];
    double r = Rc[(size_t)k * D + d];
    if (!(r > 1e-12)) r = 1e-12;
    double m[256], P[256], ll[256], sq[256];
    for (int i = 0; i < n_cand; ++i) {
      m[i] = m0[(size_t)k * D + d];
      P[i] = S0[dd];
      ll[i] = 0.0;
      sq[i] = s_cand[i] * q;
    }
    for (int t = 0; t < T; ++t) {
      const double yt = y[((size_t)k * T + t) * D + d];
      for (int i = 0; i < n_cand; ++i) {
        const double S = c * c * P[i] + r, g = 1.0 / S, e = yt - c * m[i];
        ll[i] += log(S) + e * e * g;
        m[i] = a * (m[i] + P[i] * c * g * e);
        P[i] = a * a * P[i] * r * g + sq[i];
      }
    }
    for (int i = 0; i < n_cand; ++i) {
      const double v = 0.5 * ((double)T * LOG2PI + ll[i]);
#pragma omp atomic
      nll[(size_t)k * n_cand + i] += v;
    }
  }
  return 0;
}

int eksc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
