// gfx950 kernels for the general (D, O) smoother on WIDE sessions (many keypoints: lanes = consecutive
// keypoints, so every row access of a wave is one contiguous segment) - the summarize and replay phases of
// eks_dense.hip's three-phase scan for linear observations with D <= 3 and O = 2, 4, 6, 8 (multicam linear
// path, reference eks/multicam_smoother.py:409-443; O = 2 x cameras), rebuilt in round 3 around what the
// profile of `bench.py --workload c4w` (50 000 frames x 256 keypoints, D = 3, O = 4) showed:
//   * both phases asked for a frame's rows when they needed them - one exposed memory latency per frame and
//     lane: now the rows of the NEXT group of four frames are in flight while a group is processed;
//   * the replay kept its filtered beliefs (12 doubles per frame and keypoint) in a float64 scratch stream -
//     2.4 GB written and read back for 1.0 GB of algorithmic traffic, 0.97 ms of the step's 1.39: now only a
//     CHECKPOINT of the belief entering every group of four frames is kept, in LDS ([group][field][lane],
//     36 KB per wave at 32-frame chunks), and on the way back each group is filtered again from its checkpoint
//     (+0.75 filter steps per frame, no scratch traffic) with its four filtered beliefs in registers for the
//     RTS steps.
//   * the scan in between was the narrow path's: per (keypoint, block of 64 chunks) a Hillis-Steele scan of
//     whole ELEMENTS in both directions (6 levels x 64 compositions x 2, twelve barriers) writing an inclusive
//     prefix and suffix element per chunk (2 x 34 doubles), 0.20 ms of the step.  Wide sessions have lanes to
//     spare across keypoints, so the scan is now work-efficient and sequential per lane (dwide_scan_*): runs of
//     R consecutive chunk elements are composed once (R - 1 compositions per run), runs of R run aggregates
//     again, a lane per keypoint and direction walks the ~nc / R^2 top aggregates carrying a BELIEF /
//     INFORMATION pair (apply / pull-back), and on the way down a lane per (keypoint, run, direction) pushes
//     the pair through its run's rows, leaving for every chunk the belief entering it and the information
//     leaving it (2 x 12 doubles, field-major: coalesced) - which is all the replay wants.  Elements are
//     field-major too ([chunk][field][keypoint]).  c4w: scan 0.20 -> 0.14 ms, and the replay loses its own
//     apply / pull-back and reads coalesced (0.51 -> 0.45 ms).
// Same arithmetic as eks_dense_lane.hpp (rank-1 updates per scalar observation, float64), outputs bit-compatible
// within float64 rounding.
#include <hip/hip_runtime.h>

#include <cmath>
#include <type_traits>

#include "eks_dense_lane.hpp"
#include "eks_dense_shfl.hpp"
#include "eks_internal.hpp"

namespace eks {

constexpr int kWideGroup = 4;          // frames per group (one checkpoint per group)
constexpr int kWideMaxB = 32;          // frames per lane at most (dense_chunk): 8 checkpoints

struct WideGeom {
  int K, T, O, B, nc;
};

template <int O>
struct FrameRows {
  float y[O], v[O];
};

// the O values of (frame t, keypoint k): contiguous 8- (O even) or 16-byte (O % 4 == 0) pieces, naturally aligned
// when y / var come from the allocator; the load type only promises 4 bytes (an oddly offset view works too)
template <int O>
__device__ __forceinline__ void wide_load(const float* __restrict__ y, const float* __restrict__ var,
                                          size_t row, bool ok, FrameRows<O>& f) {
  constexpr int W = O % 4 == 0 ? 4 : 2;
  typedef float fw __attribute__((ext_vector_type(W), aligned(4)));
#pragma unroll
  for (int o = 0; o < O; o += W) {
    fw a = fw(0.f), b = fw(1.f);
    if (ok) {
      a = *reinterpret_cast<const fw*>(y + row + o);
      if (var) b = *reinterpret_cast<const fw*>(var + row + o);   // (SCORE form: constant R, no rows of var)
    }
#pragma unroll
    for (int q = 0; q < W; ++q) {
      f.y[o + q] = a[q];
      f.v[o + q] = b[q];
    }
  }
}

template <int O>
__device__ __forceinline__ void wide_load_group(const float* __restrict__ y, const float* __restrict__ var,
                                                int K, int k, int t_first, int t_end, FrameRows<O> (&g)[kWideGroup]) {
#pragma unroll
  for (int f = 0; f < kWideGroup; ++f) {
    const int t = t_first + f;
    wide_load<O>(y, var, ((size_t)t * K + k) * O, t < t_end, g[f]);
  }
}

template <int D, int O>
struct WideRowsC {          // observation rows of the lane's keypoint
  double c[O][D];
  __device__ __forceinline__ Vec<double, D> row(int o) const {
    Vec<double, D> h;
#pragma unroll
    for (int i = 0; i < D; ++i) h.a[i] = c[o][i];
    return h;
  }
};
template <int D, int O>
__device__ __forceinline__ WideRowsC<D, O> wide_obs_rows(const DenseModelPtrs& M, int k) {
  WideRowsC<D, O> R;
#pragma unroll
  for (int o = 0; o < O; ++o)
#pragma unroll
    for (int i = 0; i < D; ++i) R.c[o][i] = M.C[((size_t)k * O + o) * D + i];
  return R;
}

// SCORE: r = rk[o] (the keypoint's constant variances); LL: *ll accumulates the innovation log-densities
template <int D, int O, bool SCORE = false, bool LL = false>
__device__ __forceinline__ void wide_filter_frame(const WideRowsC<D, O>& H, const FrameRows<O>& fr,
                                                  Vec<double, D>& m, Mat<double, D>& P,
                                                  const double* rk = nullptr, double* ll = nullptr) {
#pragma unroll
  for (int o = 0; o < O; ++o) {
    const Vec<double, D> h = H.row(o);
    const Vec<double, D> u = mat_vec(P, h);
    double r;
    if constexpr (SCORE)
      r = rk[o];
    else
      r = (double)clip_var(fr.v[o]);
    const double sigma = r + dot(h, u);
    const double g = rcp(sigma);
    const double dv = (double)fr.y[o] - dot(h, m);
    const double gd = g * dv;
    if constexpr (LL) *ll -= 0.5 * (kLog2Pi + log(sigma) + dv * gd);
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
  }
}

template <int D>
__device__ __forceinline__ void wide_predict(const Mat<double, D>& F, const Mat<double, D>& sQ, bool fid,
                                             Vec<double, D>& m, Mat<double, D>& P) {
  if (!fid) {
    m = mat_vec(F, m);
    P = mat_mul_nt(mat_mul(F, P), F);
  }
  P = mat_add(P, sQ);
}

// ------------------------------------------------------------------------------------------------------
template <int D, int O, bool SCORE>
__global__ __launch_bounds__(64) void dwide_summarize_kernel(WideGeom G, DenseModelPtrs M,
                                                            const double* __restrict__ s,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ var,
                                                            const double* __restrict__ rconst,
                                                            double* __restrict__ elems, int soa,
                                                            double* __restrict__ first) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G.K * G.nc) return;
  const int k = idx % G.K, j = idx / G.K;
  Mat<double, D> F, sQ;
  bool fid;
  load_dynamics<double, D>(M, k, s[k], F, sQ, fid);
  const WideRowsC<D, O> H = wide_obs_rows<D, O>(M, k);
  double rk[O];                                       // SCORE: the keypoint's constant variances
#pragma unroll
  for (int o = 0; o < O; ++o) rk[o] = SCORE ? rconst[(size_t)k * O + o] : 0.0;
  const int t0 = j * G.B, t1 = min(t0 + G.B, G.T);
  DElem<double, D> e = delem_identity<double, D>();
  FrameRows<O> cur[kWideGroup], nxt[kWideGroup];
  wide_load_group<O>(y, var, G.K, k, t0, t1, cur);
  for (int tg = t0; tg < t1; tg += kWideGroup) {
    if (tg + kWideGroup < t1) wide_load_group<O>(y, var, G.K, k, tg + kWideGroup, t1, nxt);
#pragma unroll
    for (int f = 0; f < kWideGroup; ++f) {
      const int t = tg + f;
      if (t < t1 && t > 0) {                           // frame 0 updates the prior itself (below, and in replay)
        delem_predict(e, F, sQ, fid);
#pragma unroll
        for (int o = 0; o < O; ++o)
          delem_observe(e, H.row(o), (double)cur[f].y[o],
                        SCORE ? rk[o] : (double)clip_var(cur[f].v[o]), false);
      }
    }
#pragma unroll
    for (int f = 0; f < kWideGroup; ++f) cur[f] = nxt[f];
  }
  if (soa)   // [chunk][field][keypoint] for dwide_scan_*; else records, for the scan kernels of eks_dense.hip
    store_delem<double, D>(elems + (size_t)j * delem_doubles<D>() * G.K + k, e, G.K);
  else
    store_delem<double, D>(elems + (size_t)idx * delem_doubles<D>(), e);
  if (j == 0) {   // the belief the scan starts from: the prior updated with frame 0
    Vec<double, D> m;
    Mat<double, D> P;
    load_prior<D>(M, k, m, P);
    FrameRows<O> f0;
    wide_load<O>(y, var, (size_t)k * O, true, f0);
    wide_filter_frame<D, O, SCORE>(H, f0, m, P, rk);
    double* r = first + (size_t)k * (D + D * D);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      r[a] = m.a[a];
#pragma unroll
      for (int b = 0; b < D; ++b) r[D + a * D + b] = P.a[a][b];
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The scan between summarize and replay (see the header).  Records are field-major: field f of (row r,
// keypoint k) at base[(r * NFIELDS + f) * K + k].
struct WideScan {
  int K, nc, R, nr;              // chunks, chunks per run, runs
  const double* elems;           // [nc][NV][K]
  const double* first;           // [K][REC] (record-major, from the summarize kernel)
  double *agg;                   // [nr][NV][K]
  double *run_in, *run_out;      // [nr][REC][K]  belief entering / information leaving each run
  double *chunk_in, *chunk_out;  // [nc][REC][K]  the same per chunk: what the replay reads
};

template <int D>
__device__ __forceinline__ void wide_put_pair(double* __restrict__ p, int K, const Vec<double, D>& v,
                                              const Mat<double, D>& M) {
#pragma unroll
  for (int a = 0; a < D; ++a) {
    p[(size_t)a * K] = v.a[a];
#pragma unroll
    for (int b = 0; b < D; ++b) p[(size_t)(D + a * D + b) * K] = M.a[a][b];
  }
}
template <int D>
__device__ __forceinline__ void wide_get_pair(const double* __restrict__ p, int K, Vec<double, D>& v,
                                              Mat<double, D>& M) {
#pragma unroll
  for (int a = 0; a < D; ++a) {
    v.a[a] = p[(size_t)a * K];
#pragma unroll
    for (int b = 0; b < D; ++b) M.a[a][b] = p[(size_t)(D + a * D + b) * K];
  }
}

// lane = (keypoint, run): the run's elements composed in time order
template <int D>
__global__ __launch_bounds__(64) void dwide_scan_runs_kernel(WideScan W) {
  constexpr int NV = delem_doubles<D>();
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= W.K * W.nr) return;
  const int k = idx % W.K, r = idx / W.K;
  const int j0 = r * W.R, j1 = min(W.nc, j0 + W.R);
  DElem<double, D> e = load_delem<double, D>(W.elems + (size_t)j0 * NV * W.K + k, W.K);
  for (int j = j0 + 1; j < j1; ++j)
    e = delem_combine<double, D, false>(e, load_delem<double, D>(W.elems + (size_t)j * NV * W.K + k, W.K));
  store_delem<double, D>(W.agg + (size_t)r * NV * W.K + k, e, W.K);
}

// The top of the scan: at most 64 rows per keypoint.  Block = keypoint, wave 0 forward, wave 1 reverse, lane =
// row: inclusive prefix / suffix of the rows by a Hillis-Steele scan in wave shuffles (6 levels), then every lane
// pushes the starting belief through the rows before it / pulls zero information back through the rows after it.
struct WideTop {
  int K, n;                      // rows (<= 64)
  const double* rows;            // [n][NV][K]
  const double* first;           // [K][REC]
  double *row_in, *row_out;      // [n][REC][K]
};

template <int D>
__global__ __launch_bounds__(128) void dwide_scan_top_kernel(WideTop W) {
  constexpr int NV = delem_doubles<D>(), REC = D + D * D;
  const int k = blockIdx.x, lane = threadIdx.x & 63;
  const bool rev = threadIdx.x >= 64;
  const bool live = lane < W.n;
  DElem<double, D> x = live ? load_delem<double, D>(W.rows + (size_t)lane * NV * W.K + k, W.K)
                            : delem_identity<double, D>();
  for (int off = 1; off < 64; off <<= 1) {
    if (!rev) {
      const DElem<double, D> other = delem_shfl_up<D>(x, off);
      if (lane >= off) x = delem_combine<double, D, false>(other, x);
    } else {
      const DElem<double, D> other = delem_shfl_down<D>(x, off);
      if (lane + off < 64) x = delem_combine<double, D, false>(x, other);
    }
  }
  Vec<double, D> v = vec_zero<double, D>();
  Mat<double, D> M = mat_zero<double, D>();
  if (!rev) {
    const DElem<double, D> ex = delem_shfl_up<D>(x, 1);         // the rows before this one
    const double* f0 = W.first + (size_t)k * REC;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      v.a[a] = f0[a];
#pragma unroll
      for (int b = 0; b < D; ++b) M.a[a][b] = f0[D + a * D + b];
    }
    if (lane > 0) delem_apply(ex, v, M);
    if (live) wide_put_pair<D>(W.row_in + (size_t)lane * REC * W.K + k, W.K, v, M);
  } else {
    const DElem<double, D> ex = delem_shfl_down<D>(x, 1);       // the rows after this one
    if (lane + 1 < W.n) delem_back(ex, v, M);
    if (live) wide_put_pair<D>(W.row_out + (size_t)lane * REC * W.K + k, W.K, v, M);
  }
}

// lane = (keypoint, run, direction): the same inside the run, per chunk
template <int D>
__global__ __launch_bounds__(64) void dwide_scan_chunks_kernel(WideScan W) {
  constexpr int NV = delem_doubles<D>(), REC = D + D * D;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * W.K * W.nr) return;
  const int k = idx % W.K, rr = idx / W.K;
  const bool rev = rr >= W.nr;
  const int r = rev ? rr - W.nr : rr;
  const int j0 = r * W.R, j1 = min(W.nc, j0 + W.R);
  Vec<double, D> v;
  Mat<double, D> M;
  if (!rev) {
    wide_get_pair<D>(W.run_in + (size_t)r * REC * W.K + k, W.K, v, M);
    for (int j = j0; j < j1; ++j) {
      wide_put_pair<D>(W.chunk_in + (size_t)j * REC * W.K + k, W.K, v, M);
      if (j + 1 < j1) delem_apply(load_delem<double, D>(W.elems + (size_t)j * NV * W.K + k, W.K), v, M);
    }
  } else {
    wide_get_pair<D>(W.run_out + (size_t)r * REC * W.K + k, W.K, v, M);
    for (int j = j1 - 1; j >= j0; --j) {
      wide_put_pair<D>(W.chunk_out + (size_t)j * REC * W.K + k, W.K, v, M);
      if (j > j0) delem_back(load_delem<double, D>(W.elems + (size_t)j * NV * W.K + k, W.K), v, M);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
constexpr int kWideCB = 64;            // elements per block of the narrow path's scan (eks_dense.hip: kDenseCB)

template <int D, int O, bool SCORE>
__global__ __launch_bounds__(64) void dwide_replay_kernel(WideGeom G, DenseModelPtrs M,
                                                         const double* __restrict__ s,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ var,
                                                         const double* __restrict__ rconst,
                                                         double* __restrict__ part_ll,
                                                         double* __restrict__ part_score,
                                                         const double* __restrict__ pre,
                                                         const double* __restrict__ suf,
                                                         const double* __restrict__ bprior,
                                                         const double* __restrict__ bsuffix,
                                                         const double* __restrict__ chunk_in,
                                                         const double* __restrict__ chunk_out,
                                                         float* __restrict__ ms, float* __restrict__ Vs,
                                                         int vs_diag) {
  constexpr int NF = D + D * (D + 1) / 2;            // mean + upper triangle of the covariance
  constexpr int REC = D + D * D;
  constexpr int NV = delem_doubles<D>();
  __shared__ double ck[(kWideMaxB / kWideGroup) * NF * 64];   // [group][field][lane]
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G.K * G.nc) return;
  const int lane = threadIdx.x;
  const int k = idx % G.K, j = idx / G.K;
  Mat<double, D> F, sQ;
  bool fid;
  load_dynamics<double, D>(M, k, s[k], F, sQ, fid);
  const WideRowsC<D, O> H = wide_obs_rows<D, O>(M, k);
  const int t0 = j * G.B, t1 = min(t0 + G.B, G.T), len = t1 - t0;
  FrameRows<O> cur[kWideGroup], nxt[kWideGroup];
  wide_load_group<O>(y, var, G.K, k, t0, t1, cur);    // in flight while the boundary operations run
  // belief entering the chunk / information leaving it, from the scan (as dense_replay_kernel)
  Vec<double, D> m, eta;
  Mat<double, D> P, J;
  if (chunk_in != nullptr) {                          // dwide_scan_*: ready per chunk
    wide_get_pair<D>(chunk_in + (size_t)j * REC * G.K + k, G.K, m, P);
    wide_get_pair<D>(chunk_out + (size_t)j * REC * G.K + k, G.K, eta, J);
  } else {                                            // the scan kernels of eks_dense.hip
    const int blk = j / kWideCB, ia = j % kWideCB;
    const double* rp = bprior + ((size_t)blk * G.K + k) * REC;
    const double* rs = bsuffix + ((size_t)blk * G.K + k) * REC;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      m.a[a] = rp[a];
      eta.a[a] = rs[a];
#pragma unroll
      for (int b = 0; b < D; ++b) {
        P.a[a][b] = rp[D + a * D + b];
        J.a[a][b] = rs[D + a * D + b];
      }
    }
    if (ia > 0) delem_apply(load_delem<double, D>(pre + ((size_t)(j - 1) * G.K + k) * NV), m, P);
    if (ia + 1 < kWideCB && j + 1 < G.nc)
      delem_back(load_delem<double, D>(suf + ((size_t)(j + 1) * G.K + k) * NV), eta, J);
  }
  if (j == 0) load_prior<D>(M, k, m, P);              // chunk 0 replays frame 0's update of the prior itself
  double rk[O];                                       // SCORE: the keypoint's constant variances
#pragma unroll
  for (int o = 0; o < O; ++o) rk[o] = SCORE ? rconst[(size_t)k * O + o] : 0.0;
  const Vec<double, D> m_in = m;                      // filtered belief of frame t0 - 1 (SCORE: the transition into
  const Mat<double, D> P_in = P;                      //  the chunk's first frame belongs to this lane)
  double ll = 0.0, score = 0.0;
  // ---- forward: exact filter, one checkpoint (the belief entering the group) per four frames
  double* mine = ck + lane;
  auto save = [&](int g) {
    double* c = mine + (size_t)g * NF * 64;
    int f = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) c[(f++) * 64] = m.a[a];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = a; b < D; ++b) c[(f++) * 64] = 0.5 * (P.a[a][b] + P.a[b][a]);
  };
  auto restore = [&](int g) {
    const double* c = mine + (size_t)g * NF * 64;
    int f = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) m.a[a] = c[(f++) * 64];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = a; b < D; ++b) P.a[a][b] = P.a[b][a] = c[(f++) * 64];
  };
  const int ng = (len + kWideGroup - 1) / kWideGroup;
  for (int g = 0; g < ng; ++g) {
    const int tg = t0 + g * kWideGroup;
    if (g + 1 < ng) wide_load_group<O>(y, var, G.K, k, tg + kWideGroup, t1, nxt);
    save(g);
#pragma unroll
    for (int f = 0; f < kWideGroup; ++f) {
      const int t = tg + f;
      if (t < t1) {
        if (t > 0) wide_predict<D>(F, sQ, fid, m, P);
        wide_filter_frame<D, O, SCORE, SCORE>(H, cur[f], m, P, rk, &ll);
      }
    }
    if (g + 1 < ng) {
#pragma unroll
      for (int f = 0; f < kWideGroup; ++f) cur[f] = nxt[f];
    }
  }
  // (cur now holds the rows of the LAST group)
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
  auto put = [&](float* __restrict__ dst, const float* v, auto n_tag) {
    constexpr int n = decltype(n_tag)::value;
    int q = 0;
#pragma unroll
    for (; q + 4 <= n; q += 4) EKS_STREAM_STORE(reinterpret_cast<f4u*>(dst + q), (f4u{v[q], v[q + 1], v[q + 2], v[q + 3]}));
    if constexpr (n % 4 >= 2) {
      EKS_STREAM_STORE(reinterpret_cast<f2u*>(dst + q), (f2u{v[q], v[q + 1]}));
      q += 2;
    }
    if constexpr (n % 2 == 1) EKS_STREAM_STORE(dst + q, v[q]);
  };
  auto emit = [&](int t, const Vec<double, D>& mo, const Mat<double, D>& Po) {
    const size_t ko = (size_t)t * G.K + k;
    float mv[D], pv[D * D];
#pragma unroll
    for (int a = 0; a < D; ++a) {
      mv[a] = (float)mo.a[a];
#pragma unroll
      for (int b = 0; b < D; ++b) pv[a * D + b] = (float)Po.a[a][b];
    }
    put(ms + ko * D, mv, std::integral_constant<int, D>{});
    if (vs_diag) {
      float dv[D];
#pragma unroll
      for (int a = 0; a < D; ++a) dv[a] = pv[a * D + a];
      put(Vs + ko * D, dv, std::integral_constant<int, D>{});
    } else {
      put(Vs + ko * D * D, pv, std::integral_constant<int, D * D>{});
    }
  };
  Vec<double, D> m_s;
  Mat<double, D> P_s;
  double logdet;
  condition_on_info(m, P, eta, J, m_s, P_s, logdet);  // smoothed last frame of the chunk
  if constexpr (!SCORE) emit(t1 - 1, m_s, P_s);
  Mat<double, D> Qi;                                  // SCORE: (sQ)^-1
  if constexpr (SCORE) {
    Mat<double, D> eye = mat_zero<double, D>();
#pragma unroll
    for (int a = 0; a < D; ++a) eye.a[a][a] = 1.0;
    Qi = chol_solve_mat(chol_factor(sQ), eye);
  }
  // one RTS step from the filtered belief (mf, Pf) of a frame to its smoothed belief, given the smoothed belief
  // (m_s, P_s) of the next frame; SCORE adds the transition's term of Fisher's identity (eks_dense_wave.hip)
  auto rts_step = [&](const Vec<double, D>& mf, const Mat<double, D>& Pf) {
    const Mat<double, D> FP = fid ? Pf : mat_mul(F, Pf);
    const Mat<double, D> Pp = mat_symmetrize(mat_add(fid ? Pf : mat_mul_nt(FP, F), sQ));
    const Mat<double, D> Z = chol_solve_mat(chol_factor(Pp), FP);   // Pp^-1 F Pf = G^T
    const Vec<double, D> mp = fid ? mf : mat_vec(F, mf);
    Vec<double, D> dm;
#pragma unroll
    for (int a = 0; a < D; ++a) dm.a[a] = m_s.a[a] - mp.a[a];
    const Vec<double, D> Gdm = mat_t_vec(Z, dm);
    const Vec<double, D> m_next = m_s;
    const Mat<double, D> P_next = P_s;
#pragma unroll
    for (int a = 0; a < D; ++a) m_s.a[a] = mf.a[a] + Gdm.a[a];
    P_s = mat_sandwich_tn_plus(Z, mat_sub(P_s, Pp), Pf);            // Pf + G (P_s - Pp) G^T, every pair once
    if constexpr (SCORE) {
      const Vec<double, D> Fm = fid ? m_s : mat_vec(F, m_s);
      Vec<double, D> dw;
#pragma unroll
      for (int a = 0; a < D; ++a) dw.a[a] = m_next.a[a] - Fm.a[a];
      const Mat<double, D> Cx = mat_mul_tn(Z, P_next);            // Cov(x_i, x_{i+1} | y)
      const Mat<double, D> FC = fid ? Cx : mat_mul(F, Cx);
      const Mat<double, D> FVF = fid ? P_s : mat_mul_nt(mat_mul(F, P_s), F);
      double tr = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b)
          tr += Qi.a[a][b] * (dw.a[a] * dw.a[b] + P_next.a[a][b] + FVF.a[a][b] - FC.a[a][b] - FC.a[b][a]);
      score += 0.5 * (tr - (double)D);
    }
  };
  // ---- backward: every group is filtered again from its checkpoint, RTS over its four beliefs in registers
  for (int g = ng - 1; g >= 0; --g) {
    const int tg = t0 + g * kWideGroup;
    if (g > 0) wide_load_group<O>(y, var, G.K, k, tg - kWideGroup, t1, nxt);   // the group before, in flight
    restore(g);
    Vec<double, D> mf[kWideGroup];
    Mat<double, D> Pf[kWideGroup];
#pragma unroll
    for (int f = 0; f < kWideGroup; ++f) {
      const int t = tg + f;
      if (t < t1) {
        if (t > 0) wide_predict<D>(F, sQ, fid, m, P);
        wide_filter_frame<D, O, SCORE>(H, cur[f], m, P, rk);
      }
      mf[f] = m;
      Pf[f] = mat_symmetrize(P);
    }
#pragma unroll
    for (int f = kWideGroup - 1; f >= 0; --f) {
      const int t = tg + f;
      if (t < t1 - 1) {                               // the chunk's last frame is smoothed already
        rts_step(mf[f], Pf[f]);
        if constexpr (!SCORE) emit(t, m_s, P_s);
      }
    }
    if (g > 0) {
#pragma unroll
      for (int f = 0; f < kWideGroup; ++f) cur[f] = nxt[f];
    }
  }
  if constexpr (SCORE) {
    if (t0 > 0) rts_step(m_in, mat_symmetrize(P_in));   // back to frame t0 - 1: the transition into the chunk
    part_ll[idx] = ll;                                  // [chunk][keypoint]: coalesced
    part_score[idx] = score;
  }
}

// SCORE: nll[k], d nll / d log s [k] = minus the sums of the per-row partials of keypoint k (rows = chunks here,
// blocks of 64 chunks in eks_dense_wave.hip), summed in a fixed order (the same bits on every run): block =
// keypoint, thread i takes rows i, i + 256, ..., then a tree through LDS.  eks/core.py:650: a non-finite loss
// becomes 1e12 with zero gradient.
__global__ __launch_bounds__(256) void dense_score_finish_kernel(int K, int rows, const double* __restrict__ part_ll,
                                                                const double* __restrict__ part_score,
                                                                double* __restrict__ nll, double* __restrict__ dnll) {
  __shared__ double a[256], b[256];
  const int k = blockIdx.x, i = threadIdx.x;
  double ll = 0.0, sc = 0.0;
  for (int r = i; r < rows; r += 256) {
    ll += part_ll[(size_t)r * K + k];
    sc += part_score[(size_t)r * K + k];
  }
  a[i] = ll;
  b[i] = sc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (i < off) {
      a[i] += a[i + off];
      b[i] += b[i + off];
    }
    __syncthreads();
  }
  if (i == 0) {
    const double v = -a[0];
    const bool fin = isfinite(v);
    nll[k] = fin ? v : 1e12;
    dnll[k] = fin ? -b[0] : 0.0;
  }
}
int dense_score_finish(int K, int rows, const double* part_ll, const double* part_score, double* nll, double* dnll,
                       hipStream_t st) {
  hipLaunchKernelGGL(dense_score_finish_kernel, dim3(K), dim3(256), 0, st, K, rows, part_ll, part_score, nll, dnll);
  return hip_status(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------
bool dense_wide_covers(int D, int O, int B) {
  return (D == 2 || D == 3) && (O == 2 || O == 4 || O == 6 || O == 8) && B <= kWideMaxB &&
         B % kWideGroup == 0 && !knob_int(KNOB_DENSE_LEGACY, 0);
}

// Levels of runs under the 64-row top: L levels of R rows per run bring nc chunk elements down to
// ceil(nc / R^L) <= 64 rows.  Sequential depth: L (R - 1) compositions going up (~2.4 us each at these
// occupancies: rocprofv3 per launch on c4w, tools/c4w_prof.sh), the top's 6 shuffle levels, L (R - 1) apply /
// pull-back steps going down (~1.7 us each); every level costs two launches (~4 us each) and two passes over
// its rows (~16 ns per row at c4w's width; level 0's rows are the chunk elements themselves, whatever the plan).
constexpr int kWideTopRows = 64;
constexpr int kWideMaxLevels = 6;
struct WidePlan {
  int L, R;
  int n[kWideMaxLevels + 1];     // rows per level: n[0] = nc ... n[L] <= 64
};
static WidePlan wide_scan_plan(int nc) {
  WidePlan best{};
  double best_cost = 1e300;
  for (int L = 0; L <= kWideMaxLevels; ++L)
    for (int R = 2; R <= 32; ++R) {
      WidePlan p{};
      p.L = L;
      p.R = R;
      p.n[0] = nc;
      for (int l = 0; l < L; ++l) p.n[l + 1] = (p.n[l] + R - 1) / R;
      if (p.n[L] > kWideTopRows) continue;
      double cost = L * ((R - 1) * 4.1 + 8.0);
      for (int l = 1; l <= L; ++l) cost += 0.032 * p.n[l];      // two passes over the rows of the upper levels
      if (cost < best_cost) {
        best_cost = cost;
        best = p;
      }
      if (L == 0) break;
    }
  return best;
}
// doubles of scratch the scan needs besides the two per-chunk arrays: rows and belief / information pairs of
// the levels above the chunks
size_t dense_wide_scan_scratch_doubles(int K, int D, int nc) {
  const WidePlan p = wide_scan_plan(nc);
  size_t rows = 0;
  for (int l = 1; l <= p.L; ++l) rows += p.n[l];
  return rows * K * (3 * D * D + 2 * D + 1 + 2 * (D + D * D));
}

template <int D>
static void wide_scan_launches(const WidePlan& p, const WideScan* lv, const WideTop& top, hipStream_t st) {
  for (int l = 0; l < p.L; ++l)     // up: rows of level l -> run aggregates = rows of level l + 1
    hipLaunchKernelGGL(dwide_scan_runs_kernel<D>, dim3((unsigned)(((long)lv[l].K * lv[l].nr + 63) / 64)), dim3(64), 0,
                       st, lv[l]);
  hipLaunchKernelGGL(dwide_scan_top_kernel<D>, dim3((unsigned)top.K), dim3(128), 0, st, top);
  for (int l = p.L - 1; l >= 0; --l)   // down: a level's per-row pairs are the per-run pairs of the level below
    hipLaunchKernelGGL(dwide_scan_chunks_kernel<D>, dim3((unsigned)((2L * lv[l].K * lv[l].nr + 63) / 64)), dim3(64),
                       0, st, lv[l]);
}

int dense_wide_scan(int K, int D, int nc, const double* elems, const double* first, double* scratch,
                    double* chunk_in, double* chunk_out, hipStream_t st) {
  const size_t nv = 3 * D * D + 2 * D + 1, rec = D + D * D;
  const WidePlan p = wide_scan_plan(nc);
  const double* rows[kWideMaxLevels + 1];
  double *in[kWideMaxLevels + 1], *out[kWideMaxLevels + 1];
  rows[0] = elems;
  in[0] = chunk_in;
  out[0] = chunk_out;
  double* q = scratch;
  for (int l = 1; l <= p.L; ++l) {
    rows[l] = q;
    q += (size_t)p.n[l] * K * nv;
    in[l] = q;
    q += (size_t)p.n[l] * K * rec;
    out[l] = q;
    q += (size_t)p.n[l] * K * rec;
  }
  WideScan lv[kWideMaxLevels];
  for (int l = 0; l < p.L; ++l)
    lv[l] = WideScan{K, p.n[l], p.R, p.n[l + 1], rows[l], first, const_cast<double*>(rows[l + 1]), in[l + 1], out[l + 1],
                     in[l], out[l]};
  const WideTop top{K, p.n[p.L], rows[p.L], first, in[p.L], out[p.L]};
  if (D == 2)
    wide_scan_launches<2>(p, lv, top, st);
  else if (D == 3)
    wide_scan_launches<3>(p, lv, top, st);
  else
    return EKS_ERR_UNSUPPORTED;
  return hip_status(hipGetLastError());
}

int dense_wide_summarize(int T, int K, int D, int O, int B, int nc, const DenseModelPtrs& M, const double* s,
                         const float* y, const float* var, const double* rconst, double* elems, int soa,
                         double* first, hipStream_t st) {
  const bool score = var == nullptr;
  const WideGeom G{K, T, O, B, nc};
  const int lanes = K * nc;
  const dim3 grid((lanes + 63) / 64), block(64);
#define EKS_WS(DD, OO)                                                                                         \
  if (score)                                                                                                   \
    hipLaunchKernelGGL((dwide_summarize_kernel<DD, OO, true>), grid, block, 0, st, G, M, s, y, var, rconst,     \
                       elems, soa, first);                                                                     \
  else                                                                                                         \
    hipLaunchKernelGGL((dwide_summarize_kernel<DD, OO, false>), grid, block, 0, st, G, M, s, y, var, rconst,    \
                       elems, soa, first)
#define EKS_WS_O(DD)                    \
  switch (O) {                          \
    case 2: EKS_WS(DD, 2); break;       \
    case 4: EKS_WS(DD, 4); break;       \
    case 6: EKS_WS(DD, 6); break;       \
    case 8: EKS_WS(DD, 8); break;       \
    default: return EKS_ERR_UNSUPPORTED; \
  }
  if (D == 2) {
    EKS_WS_O(2)
  } else if (D == 3) {
    EKS_WS_O(3)
  } else {
    return EKS_ERR_UNSUPPORTED;
  }
#undef EKS_WS_O
#undef EKS_WS
  return hip_status(hipGetLastError());
}

int dense_wide_replay(int T, int K, int D, int O, int B, int nc, const DenseModelPtrs& M, const double* s,
                      const float* y, const float* var, const double* rconst, double* part_ll,
                      double* part_score, const double* pre, const double* suf, const double* bprior,
                      const double* bsuffix, const double* chunk_in, const double* chunk_out, float* ms,
                      float* Vs, int vs_diag, hipStream_t st) {
  const bool score = var == nullptr;
  const WideGeom G{K, T, O, B, nc};
  const int lanes = K * nc;
  const dim3 grid((lanes + 63) / 64), block(64);
#define EKS_WR(DD, OO)                                                                                          \
  if (score)                                                                                                    \
    hipLaunchKernelGGL((dwide_replay_kernel<DD, OO, true>), grid, block, 0, st, G, M, s, y, var, rconst, part_ll, \
                       part_score, pre, suf, bprior, bsuffix, chunk_in, chunk_out, ms, Vs, vs_diag);             \
  else                                                                                                          \
    hipLaunchKernelGGL((dwide_replay_kernel<DD, OO, false>), grid, block, 0, st, G, M, s, y, var, rconst,        \
                       part_ll, part_score, pre, suf, bprior, bsuffix, chunk_in, chunk_out, ms, Vs, vs_diag)
#define EKS_WR_O(DD)                    \
  switch (O) {                          \
    case 2: EKS_WR(DD, 2); break;       \
    case 4: EKS_WR(DD, 4); break;       \
    case 6: EKS_WR(DD, 6); break;       \
    case 8: EKS_WR(DD, 8); break;       \
    default: return EKS_ERR_UNSUPPORTED; \
  }
  if (D == 2) {
    EKS_WR_O(2)
  } else if (D == 3) {
    EKS_WR_O(3)
  } else {
    return EKS_ERR_UNSUPPORTED;
  }
#undef EKS_WR_O
#undef EKS_WR
  return hip_status(hipGetLastError());
}

}  // namespace eks

EKS_DEFINE_TOUCH(dense_wide)
