// gfx950 kernels for the general (D, O) smoother on NARROW sessions - BASELINE configs[3]: mirrored
// multicam, 2 views x 4 paws x 50 000 frames, D = 3, O = 4 (reference eks/multicam_smoother.py:409-443,
// :481-511) - where a few keypoints offer no parallelism but time.  Such a problem is DEPTH-bound: what
// counts is the longest chain of dependent float64 operations and of dependent memory round trips, not
// bytes.  Round 1's three-phase form (eks_dense.hip) spent 149 us on configs[3]: per-frame loads one HBM
// latency apart, filtered beliefs through a float64 scratch stream, and the scan as two launches of
// barrier-separated LDS passes.  Here (round 3):
//
//   DW1 dw_summarize : block = (keypoint, 64 consecutive 8-frame chunks), lane = chunk.  A lane requests
//                      all rows of its chunk at once (one memory latency per chunk instead of one per
//                      frame) and builds the chunk element in registers; then the block's two waves - both
//                      hold the same 64 elements, no exchange - run the forward and the reverse
//                      Hillis-Steele scan of them by WAVE SHUFFLES (no LDS, no barrier): exclusive prefix /
//                      suffix per chunk and the block aggregate.  The first scan level of the old design,
//                      its element stores and loads and one kernel boundary are gone.
//   DW2 dw_replay    : same blocks.  Wave 0 reduces the aggregates of the earlier blocks (shuffle tree) and
//                      pushes the prior through them, wave 1 pulls the information back through the later
//                      ones; one LDS hand-over.  Then lane = chunk: exact filter from the chunk's entering
//                      belief with the filtered beliefs kept in LDS ([frame][field][lane], conflict-free),
//                      fuse with the future's information, RTS backwards, outputs.  The second scan level is
//                      part of this launch: two launches in all instead of four.
//
// SCORE form of the same two kernels (round 3; the optimiser's loss on this path, eks/core.py:640-650 with
// jax.value_and_grad at :652): constant R instead of the frames' variances, no outputs; the replay's exact
// filter sums the innovation log-densities (the marginal log-likelihood) and its RTS pass sums Fisher's
// identity for the score,
//     d loglik / d log s = sum_t  E[ d/d log s  log N(x_t; F x_{t-1}, s Q) | y ]
//                        = sum_t  ( tr((sQ)^-1 E[w_t w_t^T | y]) - D ) / 2,     w_t = x_t - F x_{t-1},
//     E[w w^T | y] = dm dm^T + V_t + F V_{t-1} F^T - F V_{t-1,t} - (F V_{t-1,t})^T,  V_{t-1,t} = G_{t-1} V_t,
// exact for the exact smoothing distribution and Q positive definite (the caller's EKS_FLAG_Q_PD) - plain
// float64 where the dual-number kernels (eks_loss.hip) carry 68 doubles per element through every
// composition and spill (0.26 ms per evaluation on configs[3] against 0.08 here).
//   MODE 1: dynamics (F, s Q), constant R, derivative with respect to log s (the formula above).
//   MODE 2: the pupil loss (eks/ibl_pupil_smoother.py:540-552): AR(1) dynamics x_t = a . x_{t-1} + N(0, diag q),
//           the frames' own variances; the same identity per coordinate gives the two sums
//               S_q[i] = sum_t ( E[w_i^2 | y] / q_i - 1 ) / (2 q_i),      S_a[i] = sum_t E[w_i x_{t-1,i} | y] / q_i,
//               E[w_i x_{t-1,i}] = dm_i m_{t-1,i} + Cov(x_{t-1}, x_t)_ii - a_i V_{t-1,ii},
//           and d loglik / d theta = sum_i S_q[i] dq_i/dtheta + S_a[i] da_i/dtheta for every tangent (da, dq) the
//           caller passes (q > 0 in every coordinate: the caller's EKS_FLAG_Q_PD).
//
// Matrices are float64 in registers; R_t is diagonal, so a frame's observations are absorbed one scalar at
// a time (rank-1 forms, no inverse); the scan's compositions use the Cholesky / Woodbury forms of
// eks_dense_math.hpp without the log-likelihood term the smoother does not need.  Wide sessions (more than
// ~1000 blocks) keep the keypoint-major kernels of eks_dense.hip, whose rows are coalesced across keypoints.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "eks_adam.hpp"
#include "eks_dense_lane.hpp"
#include "eks_dense_shfl.hpp"
#include "eks_internal.hpp"

namespace eks {

// Frames per lane, a template parameter of the kernels chosen per problem by dw_chunk_frames (round 4): the DEPTH of
// these launches is what a chunk costs (B element steps, B filter steps, B smoother steps) plus what its scan costs
// (six shuffle levels per 64 chunks and a walk over the units' aggregates), so short sessions want short chunks and
// long ones long chunks.  Measured (tools/pupil_time.py, the pupil optimiser on one chain; B = 2 / 4 / 8):
// T = 2 000: 160 / 174 / 224 ms, T = 20 000: 217 / 216 / 250 ms, T = 200 000: - / 833 / 495 ms; configs[3]
// (4 chains x 50 000 frames): - / 99.6 / 68.2 us per smooth (16 frames: 94 us, round 3).
constexpr int kDwBMax = 8;
// the smallest chunk whose (keypoint, 64-chunk) units still number at most kDwUnitsShort
constexpr int kDwUnitsShort = 160;
static inline long dw_units(int T, int K, int B) {
  const long nc = ((long)T + B - 1) / B;
  return (long)K * ((nc + 63) / 64);
}
// (O > 8 - five and six cameras - keeps 8 frames: only that form is instantiated for them.  EKS_DW_CHUNK = 2 / 4 / 8
// forces a choice for A/B runs; O = 0 asks for the choice that needs the largest workspace.)
static inline int dw_chunk_frames(int T, int K, int O) {
  if (O > 8) return kDwBMax;
  const int forced = knob_int(KNOB_DW_CHUNK, 0);
  if (forced == 2 || forced == 4 || forced == 8) return forced;
  if (dw_units(T, K, 2) <= kDwUnitsShort) return 2;
  if (dw_units(T, K, 4) <= kDwUnitsShort) return 4;
  return kDwBMax;
}

// Diagnostic build only (-DEKS_DW_STAMPS, tools/dw_stamps.py): wave 0's lane 0 of the first 64 blocks
// stamps the 100 MHz real-time counter at the phase boundaries; nothing reads the stamps but the tool.
#ifdef EKS_DW_STAMPS
__device__ unsigned long long g_dw_stamps[2][64][16];
#define DW_STAMP(kern, ph)                                                              \
  do {                                                                                  \
    if ((threadIdx.x & 127) == 0 && blockIdx.x < 32) g_dw_stamps[kern][blockIdx.x * 2 + (threadIdx.x >> 7)][ph] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define DW_STAMP(kern, ph) do { } while (0)
#endif

struct DwGeom {
  int K, T, nc, nwb;         // keypoints, frames, chunks (ceil(T / B)), blocks per keypoint (ceil(nc / 64))
};

// observation rows of ONE keypoint (the block's keypoint is uniform: these live in scalar registers)
template <int D, int O>
struct ObsRows {
  double c[O][D];
  __device__ __forceinline__ Vec<double, D> row(int o) const {
    Vec<double, D> h;
#pragma unroll
    for (int i = 0; i < D; ++i) h.a[i] = c[o][i];
    return h;
  }
};
template <int D, int O>
__device__ __forceinline__ ObsRows<D, O> load_obs_rows(const DenseModelPtrs& M, int k) {
  ObsRows<D, O> R;
#pragma unroll
  for (int o = 0; o < O; ++o)
#pragma unroll
    for (int i = 0; i < D; ++i) R.c[o][i] = M.C[((size_t)k * O + o) * D + i];
  return R;
}

// A lane's chunk: all B rows (O floats of y, O of var each) are requested at once - one memory latency
// per chunk - and parked in LDS as [frame][lane][O], so the frame loops below stay ROLLED: these kernels run
// their code once per wave, and straight-line code for 16 frames x O observations (38 KB at O = 4) was
// fetched cold from L2 at ~7 cycles per instruction (in-kernel stamps, profiles/r03_probes.txt).
template <int O, int B>
struct DwRows {
  static constexpr int W = O % 4 == 0 ? 4 : 2;
  typedef float fw __attribute__((ext_vector_type(W)));
  fw a[B][O / W], b[B][O / W];
};
// request: all loads of the chunk in flight (nothing waits here - the caller issues its model loads next);
// park: into LDS, [frame][lane][O]
template <int O, int B>
__device__ __forceinline__ void dw_request_rows(const float* __restrict__ y, const float* __restrict__ var,
                                                size_t row0, size_t row_stride, int nrows, DwRows<O, B>& R) {
  // a keypoint's O values of one frame are contiguous: 8- / 16-byte pieces.  The pieces are naturally aligned
  // when y / var come from the allocator; the load type only promises 4 bytes (global loads may be misaligned),
  // so an oddly offset view of a larger array works too
  constexpr int W = DwRows<O, B>::W;
  typedef typename DwRows<O, B>::fw fw;
  typedef float fwu __attribute__((ext_vector_type(W), aligned(4)));
#pragma unroll
  for (int i = 0; i < B; ++i) {
    const size_t r = row0 + (size_t)i * row_stride;
#pragma unroll
    for (int o = 0; o < O / W; ++o) {
      if (i < nrows) {
        R.a[i][o] = *reinterpret_cast<const fwu*>(y + r + o * W);
        R.b[i][o] = var ? fw(*reinterpret_cast<const fwu*>(var + r + o * W)) : fw(1.f);   // (SCORE: constant R)
      } else {
        R.a[i][o] = fw(0.f);
        R.b[i][o] = fw(1.f);
      }
    }
  }
}
template <int O, int B>
__device__ __forceinline__ void dw_park_rows(const DwRows<O, B>& R, float* __restrict__ ly, float* __restrict__ lv,
                                             int lane) {
  constexpr int W = DwRows<O, B>::W;
  typedef typename DwRows<O, B>::fw fw;
#pragma unroll
  for (int i = 0; i < B; ++i)
#pragma unroll
    for (int o = 0; o < O / W; ++o) {
      *reinterpret_cast<fw*>(ly + ((size_t)i * 64 + lane) * O + o * W) = R.a[i][o];
      if (lv) *reinterpret_cast<fw*>(lv + ((size_t)i * 64 + lane) * O + o * W) = R.b[i][o];
    }
}

// dynamics of keypoint k: (F, s Q) of the model, or (MODE 2) diag(a), diag(q) given per coordinate
template <int D, int MODE>
__device__ __forceinline__ void dw_load_dynamics(const DenseModelPtrs& M, const double* __restrict__ s,
                                                 const double* __restrict__ ar_a, const double* __restrict__ ar_q,
                                                 int k, Mat<double, D>& F, Mat<double, D>& sQ, bool& fid) {
  if constexpr (MODE == 2) {
    F = mat_zero<double, D>();
    sQ = mat_zero<double, D>();
#pragma unroll
    for (int i = 0; i < D; ++i) {
      F.a[i][i] = ar_a[(size_t)k * D + i];
      sQ.a[i][i] = ar_q[(size_t)k * D + i];
    }
    fid = false;
  } else {
    load_dynamics<double, D>(M, k, s[k], F, sQ, fid);
  }
}

// ------------------------------------------------------------------------------------------------------
// SUBS wave pairs per workgroup (each pair = one (keypoint, 64 chunks) unit): with more units than CUs, two
// 2-wave workgroups on one CU could land on the same SIMDs and halve each other's float64 rate (measured:
// 392 units of 2 waves 45 us, their own lifetime 28 us); a 4-wave workgroup spreads over the CU's four SIMDs.
template <int D, int O, int SUBS, int MODE, int B>
__global__ __launch_bounds__(128 * SUBS) void dw_summarize_kernel(DwGeom G, DenseModelPtrs M,
                                                          const double* __restrict__ s,
                                                          const float* __restrict__ y,
                                                          const float* __restrict__ var,
                                                          const double* __restrict__ rconst,
                                                          const double* __restrict__ ar_a,
                                                          const double* __restrict__ ar_q,
                                                          double* __restrict__ pre_ex,
                                                          double* __restrict__ suf_ex,
                                                          double* __restrict__ agg,
                                                          double* __restrict__ first) {
  constexpr bool SCORE = MODE == 1;                   // constant R instead of rows of var
  constexpr int NV = delem_doubles<D>();
  // each wave parks its own copy of its rows (SCORE reads no variances: constant R)
  __shared__ float ly[2 * SUBS][B * 64 * O], lv[SCORE ? 1 : 2 * SUBS][SCORE ? 1 : B * 64 * O];
  const int unit = blockIdx.x * SUBS + (threadIdx.x >> 7);
  if (unit >= G.K * G.nwb) return;                    // (no barrier in this kernel)
  const int k = unit % G.K, wb = unit / G.K;
  const int lane = threadIdx.x & 63;
  const bool rev = (threadIdx.x & 64) != 0;           // odd wave: the reverse scan
  const int j = wb * 64 + lane;
  const bool live = j < G.nc;
  DW_STAMP(0, 0);
  const int t0 = live ? j * B : 0, len = live ? min(B, G.T - t0) : 0;
  float* my_y = ly[threadIdx.x >> 6];
  float* my_v = SCORE ? nullptr : lv[threadIdx.x >> 6];
  DwRows<O, B> rows;
  dw_request_rows<O, B>(y, var, ((size_t)t0 * G.K + k) * O, (size_t)G.K * O, len, rows);
  Mat<double, D> F, sQ;                               // (the model's loads go out behind the rows': one round trip)
  bool fid;
  dw_load_dynamics<D, MODE>(M, s, ar_a, ar_q, k, F, sQ, fid);
  const ObsRows<D, O> H = load_obs_rows<D, O>(M, k);
  double rk[O];                                       // SCORE: the keypoint's constant variances
#pragma unroll
  for (int o = 0; o < O; ++o) rk[o] = SCORE ? rconst[(size_t)k * O + o] : 0.0;
  dw_park_rows<O, B>(rows, my_y, my_v, lane);
  DElem<double, D> e = delem_identity<double, D>();
  DW_STAMP(0, 1);
#pragma unroll 1
  for (int i = 0; i < len; ++i) {
    if (t0 + i == 0) continue;                        // frame 0 updates the prior itself (dw_replay)
    delem_predict(e, F, sQ, fid);
    const float* py = my_y + ((size_t)i * 64 + lane) * O;
#pragma unroll
    for (int o = 0; o < O; ++o) {
      double r = rk[o];
      if constexpr (!SCORE) {
        const float v = my_v[((size_t)i * 64 + lane) * O + o];
        r = (double)clip_var(v);
      }
      delem_observe(e, H.row(o), (double)py[o], r, false);
    }
  }
  DW_STAMP(0, 2);
  // inclusive scan over the block's 64 chunk elements, forward in wave 0, reverse in wave 1 (rolled: the
  // composition's code exists once)
  DElem<double, D> x = e;
#pragma unroll 1
  for (int off = 1; off < 64; off <<= 1) {
    if (!rev) {
      const DElem<double, D> other = delem_shfl_up<D>(x, off);
      if (lane >= off) x = delem_combine<double, D, false>(other, x);
    } else {
      const DElem<double, D> other = delem_shfl_down<D>(x, off);
      if (lane + off < 64) x = delem_combine<double, D, false>(x, other);
    }
  }
  DW_STAMP(0, 3);
  if (!rev) {
    if (lane == 63) store_delem<double, D>(agg + ((size_t)wb * G.K + k) * NV, x);
    DElem<double, D> ex = delem_shfl_up<D>(x, 1);     // exclusive prefix: the elements before this chunk
    if (lane == 0) ex = delem_identity<double, D>();
    if (live) store_delem<double, D>(pre_ex + ((size_t)j * G.K + k) * NV, ex);
    if (wb == 0 && lane == 0) {                       // the belief the scan starts from: prior + frame 0
      Vec<double, D> m;
      Mat<double, D> P;
      load_prior<D>(M, k, m, P);
      if constexpr (SCORE) {
#pragma unroll
        for (int o = 0; o < O; ++o) {
          const Vec<double, D> h = H.row(o);
          const Vec<double, D> u = mat_vec(P, h);
          const double g = rcp(rk[o] + dot(h, u));
          const double gd = g * ((double)y[(size_t)k * O + o] - dot(h, m));
#pragma unroll
          for (int a = 0; a < D; ++a) {
            m.a[a] += u.a[a] * gd;
#pragma unroll
            for (int b = 0; b < D; ++b) P.a[a][b] -= u.a[a] * u.a[b] * g;
          }
        }
      } else {
        belief_update_obs<D>(make_linear_obs<D>(y, var, G.K, O, M), k, 0, nullptr, m, P);
      }
      double* r = first + (size_t)k * (D + D * D);
#pragma unroll
      for (int a = 0; a < D; ++a) {
        r[a] = m.a[a];
#pragma unroll
        for (int b = 0; b < D; ++b) r[D + a * D + b] = P.a[a][b];
      }
    }
    DW_STAMP(0, 4);
  } else {
    DElem<double, D> ex = delem_shfl_down<D>(x, 1);   // exclusive suffix: the elements after this chunk
    if (lane == 63) ex = delem_identity<double, D>();
    if (live) store_delem<double, D>(suf_ex + ((size_t)j * G.K + k) * NV, ex);
  }
}

// Time-ordered composition of the block aggregates [q_lo, q_hi) of keypoint k, delivered in lane 0: every
// lane composes its run of ceil(n / 64) consecutive aggregates, then a shuffle tree over the lanes that hold
// one (depth: run length - 1 + ceil(log2(lanes)) compositions; rolled: one copy of the composition's code).
template <int D>
__device__ __forceinline__ DElem<double, D> dw_compose_range(const double* __restrict__ agg, int K, int k,
                                                             int q_lo, int q_hi, int lane) {
  constexpr int NV = delem_doubles<D>();
  const int n = q_hi - q_lo, per = (n + 63) / 64;
  const int nl = per > 0 ? (n + per - 1) / per : 0;   // lanes that hold a run
  DElem<double, D> x = delem_identity<double, D>();
  const int a0 = q_lo + lane * per, a1 = min(q_hi, a0 + per);
  if (a0 < a1) {
    x = load_delem<double, D>(agg + ((size_t)a0 * K + k) * NV);
#pragma unroll 1
    for (int q = a0 + 1; q < a1; ++q)
      x = delem_combine<double, D, false>(x, load_delem<double, D>(agg + ((size_t)q * K + k) * NV));
  }
#pragma unroll 1
  for (int off = 1; off < nl; off <<= 1) {
    const DElem<double, D> other = delem_shfl_down<D>(x, off);
    if ((lane & (2 * off - 1)) == 0 && lane + off < nl) x = delem_combine<double, D, false>(x, other);
  }
  return x;
}

template <int D, int O, int SUBS, int MODE, int B>
__global__ __launch_bounds__(128 * SUBS) void dw_replay_kernel(DwGeom G, DenseModelPtrs M,
                                                       const double* __restrict__ s,
                                                       const float* __restrict__ y,
                                                       const float* __restrict__ var,
                                                       const double* __restrict__ rconst,
                                                       const double* __restrict__ ar_a,
                                                       const double* __restrict__ ar_q,
                                                       double* __restrict__ part,
                                                       const double* __restrict__ pre_ex,
                                                       const double* __restrict__ suf_ex,
                                                       const double* __restrict__ agg,
                                                       const double* __restrict__ first,
                                                       float* __restrict__ ms, float* __restrict__ Vs,
                                                       int vs_diag) {
  constexpr bool SCORE = MODE == 1;                   // constant R instead of rows of var
  constexpr bool SUMS = MODE != 0;                    // loss and derivative sums instead of ms / Vs
  constexpr int NV = delem_doubles<D>();
  constexpr int NF = D + D * (D + 1) / 2;             // filtered mean + upper triangle of the covariance
  constexpr int REC = D + D * D;
  __shared__ double recs_all[SUBS][B * NF * 64];   // [frame][field][lane]
  __shared__ double xch_all[SUBS][REC];
  __shared__ float ly_all[SUBS][B * 64 * O], lv_all[SCORE ? 1 : SUBS][SCORE ? 1 : B * 64 * O];
  const int sub = threadIdx.x >> 7;
  double* recs = recs_all[sub];
  double* xch = xch_all[sub];
  float* ly = ly_all[sub];
  float* lv = SCORE ? nullptr : lv_all[sub];
  const int unit = blockIdx.x * SUBS + sub;
  if (unit >= G.K * G.nwb) {                          // a spare wave pair still meets the workgroup's barrier
    __syncthreads();
    return;
  }
  const int k = unit % G.K, wb = unit / G.K;
  const int lane = threadIdx.x & 63;
  const bool rev = (threadIdx.x & 64) != 0;
  const int j = wb * 64 + lane;
  const bool live = j < G.nc;
  if (rev) {
    // information about the state leaving this block, from the aggregates of all later blocks
    Vec<double, D> eta = vec_zero<double, D>();
    Mat<double, D> J = mat_zero<double, D>();
    if (wb + 1 < G.nwb) {
      const DElem<double, D> tot = dw_compose_range<D>(agg, G.K, k, wb + 1, G.nwb, lane);
      if (lane == 0) delem_back(tot, eta, J);
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < D; ++a) {
        xch[a] = eta.a[a];
#pragma unroll
        for (int b = 0; b < D; ++b) xch[D + a * D + b] = J.a[a][b];
      }
    }
    __syncthreads();
    return;
  }
  // ---- wave 0: everything this lane will need is requested before the reduction starts
  DW_STAMP(1, 0);
  DwRows<O, B> rows;
  {
    const int jj = wb * 64 + lane;
    const int tt0 = jj < G.nc ? jj * B : 0, ll = jj < G.nc ? min(B, G.T - tt0) : 0;
    dw_request_rows<O, B>(y, var, ((size_t)tt0 * G.K + k) * O, (size_t)G.K * O, ll, rows);
  }
  Mat<double, D> F, sQ;
  bool fid;
  dw_load_dynamics<D, MODE>(M, s, ar_a, ar_q, k, F, sQ, fid);
  const ObsRows<D, O> H = load_obs_rows<D, O>(M, k);
  const int t0 = live ? j * B : 0, len = live ? min(B, G.T - t0) : 0;
  DElem<double, D> pe = delem_identity<double, D>(), se = delem_identity<double, D>();
  if (live) {
    pe = load_delem<double, D>(pre_ex + ((size_t)j * G.K + k) * NV);
    se = load_delem<double, D>(suf_ex + ((size_t)j * G.K + k) * NV);
  }
  // belief entering this block: the prior (updated with frame 0) through the aggregates of all earlier blocks
  Vec<double, D> m;
  Mat<double, D> P;
  {
    const double* f0 = first + (size_t)k * REC;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      m.a[a] = f0[a];
#pragma unroll
      for (int b = 0; b < D; ++b) P.a[a][b] = f0[D + a * D + b];
    }
  }
  dw_park_rows<O, B>(rows, ly, lv, lane);
  DW_STAMP(1, 1);
  if (wb > 0) {
    const DElem<double, D> tot = dw_compose_range<D>(agg, G.K, k, 0, wb, lane);
    if (lane == 0) delem_apply(tot, m, P);
#pragma unroll
    for (int a = 0; a < D; ++a) {                     // lane 0's belief to every lane
      m.a[a] = __shfl(m.a[a], 0);
#pragma unroll
      for (int b = 0; b < D; ++b) P.a[a][b] = __shfl(P.a[a][b], 0);
    }
  }
  DW_STAMP(1, 2);
  __syncthreads();                                    // wave 1's information is in LDS
  DW_STAMP(1, 3);
  Vec<double, D> eta;
  Mat<double, D> J;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    eta.a[a] = xch[a];
#pragma unroll
    for (int b = 0; b < D; ++b) J.a[a][b] = xch[D + a * D + b];
  }
  if (!SUMS && !live) return;                         // (sums: every lane joins the wave's reduction at the end)
  if (lane > 0) delem_apply(pe, m, P);                // through the block's chunks before this one
  if (lane < 63 && j + 1 < G.nc) delem_back(se, eta, J);   // back through those after it
  if (j == 0) load_prior<D>(M, k, m, P);              // chunk 0 replays frame 0's update of the prior itself
  DW_STAMP(1, 4);
  double rk[O];                                       // SCORE: the keypoint's constant variances
#pragma unroll
  for (int o = 0; o < O; ++o) rk[o] = SCORE ? rconst[(size_t)k * O + o] : 0.0;
  const Vec<double, D> m_in = m;                      // filtered belief of frame t0 - 1 (SCORE: the transition
  const Mat<double, D> P_in = P;                      //  into this chunk's first frame is this lane's)
  double ll = 0.0, score = 0.0;
  double sq_sum[D], sa_sum[D];                        // MODE 2: S_q, S_a per coordinate
#pragma unroll
  for (int a = 0; a < D; ++a) sq_sum[a] = sa_sum[a] = 0.0;
  // ---- exact filter over the chunk; filtered beliefs to LDS
  double* mine = recs + lane;
#pragma unroll 1
  for (int i = 0; i < len; ++i) {
    if (t0 + i > 0) {
      if (!fid) {
        m = mat_vec(F, m);
        P = mat_mul_nt(mat_mul(F, P), F);
      }
      P = mat_add(P, sQ);
    }
    const float* py = ly + ((size_t)i * 64 + lane) * O;
#pragma unroll
    for (int o = 0; o < O; ++o) {
      const Vec<double, D> h = H.row(o);
      const Vec<double, D> u = mat_vec(P, h);
      double r = rk[o];
      if constexpr (!SCORE) {
        const float vf = lv[((size_t)i * 64 + lane) * O + o];
        r = (double)clip_var(vf);
      }
      const double sigma = r + dot(h, u);
      const double g = rcp(sigma);
      const double dv = (double)py[o] - dot(h, m);
      const double gd = g * dv;
      if constexpr (SUMS) ll -= 0.5 * (kLog2Pi + log(sigma) + dv * gd);    // log N(y_o; h.m, sigma)
      // (the gain folded into one factor, the symmetric pairs of P computed once: P stays exactly symmetric)
#pragma unroll
      for (int a = 0; a < D; ++a) {
        m.a[a] += u.a[a] * gd;
        const double ug = u.a[a] * g;
#pragma unroll
        for (int b = a; b < D; ++b) {
          const double pv = P.a[a][b] - ug * u.a[b];
          P.a[a][b] = pv;
          P.a[b][a] = pv;
        }
      }
    }
    double* rc = mine + (size_t)i * NF * 64;
    int f = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) rc[(f++) * 64] = m.a[a];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = a; b < D; ++b) rc[(f++) * 64] = 0.5 * (P.a[a][b] + P.a[b][a]);
  }
  // a frame's outputs of one keypoint are contiguous (D and D x D floats, 4-byte aligned): few wide stores
  // instead of D + D x D scalar ones - each store instruction of these lanes touches 64 different lines
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
  auto put = [&](float* __restrict__ dst, const float* v, auto n_tag) {
    constexpr int n = decltype(n_tag)::value;
    int q = 0;
#pragma unroll
    // PLAIN stores: a line of ms / Vs is shared by the K keypoints' workgroups, each writing its 12 / 36
    // bytes; the L2s merge such pieces, non-temporal stores send every piece on by itself (configs[3]:
    // replay 54 -> 42 us)
    for (; q + 4 <= n; q += 4) *reinterpret_cast<f4u*>(dst + q) = f4u{v[q], v[q + 1], v[q + 2], v[q + 3]};
    if constexpr (n % 4 >= 2) {
      *reinterpret_cast<f2u*>(dst + q) = f2u{v[q], v[q + 1]};
      q += 2;
    }
    if constexpr (n % 2 == 1) dst[q] = v[q];
  };
  auto emit = [&](int i, const Vec<double, D>& mo, const Mat<double, D>& Po) {
    const size_t ko = (size_t)(t0 + i) * G.K + k;
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
  DW_STAMP(1, 5);
  Vec<double, D> m_s;
  Mat<double, D> P_s;
  double logdet;
  condition_on_info(m, P, eta, J, m_s, P_s, logdet);  // smoothed last frame of the chunk
  if constexpr (!SUMS) emit(len - 1, m_s, P_s);
  DW_STAMP(1, 6);
  Mat<double, D> Qi;                                  // MODE 1: (sQ)^-1
  if constexpr (MODE == 1) {
    Mat<double, D> eye = mat_zero<double, D>();
#pragma unroll
    for (int a = 0; a < D; ++a) eye.a[a][a] = 1.0;
    Qi = chol_solve_mat(chol_factor(sQ), eye);
  }
  // RTS backwards over the LDS records; SCORE goes one step further, to the filtered belief that entered the
  // chunk (frame t0 - 1): the transition into the chunk's first frame
  for (int i = len - 2; i >= (SUMS ? -1 : 0); --i) {
    Vec<double, D> mf;
    Mat<double, D> Pf;
    if (i >= 0) {
      const double* rc = mine + (size_t)i * NF * 64;
      int f = 0;
#pragma unroll
      for (int a = 0; a < D; ++a) mf.a[a] = rc[(f++) * 64];
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a; b < D; ++b) Pf.a[a][b] = Pf.a[b][a] = rc[(f++) * 64];
    } else {
      if (t0 == 0 || len == 0) break;                 // frame 0 has no transition into it
      mf = m_in;
      Pf = mat_symmetrize(P_in);
    }
    const Mat<double, D> FP = fid ? Pf : mat_mul(F, Pf);
    const Mat<double, D> Pp = mat_symmetrize(mat_add(fid ? Pf : mat_mul_nt(FP, F), sQ));
    const Mat<double, D> Z = chol_solve_mat(chol_factor(Pp), FP);      // Pp^-1 F Pf = G^T
    const Vec<double, D> mp = fid ? mf : mat_vec(F, mf);
    Vec<double, D> dm;
#pragma unroll
    for (int a = 0; a < D; ++a) dm.a[a] = m_s.a[a] - mp.a[a];
    const Vec<double, D> Gdm = mat_t_vec(Z, dm);
    const Vec<double, D> m_next = m_s;                // smoothed frame i + 1
    const Mat<double, D> P_next = P_s;
#pragma unroll
    for (int a = 0; a < D; ++a) m_s.a[a] = mf.a[a] + Gdm.a[a];
    P_s = mat_sandwich_tn_plus(Z, mat_sub(P_s, Pp), Pf);            // Pf + G (P_s - Pp) G^T, every pair once
    if constexpr (MODE == 2) {
      // per coordinate (F = diag a, Q = diag q): E[w^2] and E[w x_i] from the same smoothed moments
      const Mat<double, D> Cx = mat_mul_tn(Z, P_next);            // Cov(x_i, x_{i+1} | y)
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double av = F.a[a][a], iq = rcp(sQ.a[a][a]);
        const double dwv = m_next.a[a] - av * m_s.a[a];
        const double ew2 = dwv * dwv + P_next.a[a][a] + av * av * P_s.a[a][a] - 2.0 * av * Cx.a[a][a];
        const double ewx = dwv * m_s.a[a] + Cx.a[a][a] - av * P_s.a[a][a];
        sq_sum[a] += 0.5 * iq * (ew2 * iq - 1.0);
        sa_sum[a] += ewx * iq;
      }
    } else if constexpr (MODE == 1) {
      // E[w w^T | y] for w = x_{i+1} - F x_i:  dm dm^T + V_{i+1} + F V_i F^T - F C - (F C)^T,  C = Cov(x_i, x_{i+1}) = Z^T V_{i+1}
      const Vec<double, D> Fm = fid ? m_s : mat_vec(F, m_s);
      Vec<double, D> dw;
#pragma unroll
      for (int a = 0; a < D; ++a) dw.a[a] = m_next.a[a] - Fm.a[a];
      const Mat<double, D> Cx = mat_mul_tn(Z, P_next);
      const Mat<double, D> FC = fid ? Cx : mat_mul(F, Cx);
      const Mat<double, D> FVF = fid ? P_s : mat_mul_nt(mat_mul(F, P_s), F);
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
  DW_STAMP(1, 7);
  if constexpr (SUMS) {
    // the unit's sums in lane order (fixed: the same bits on every run): planes [sum][block of 64 chunks][keypoint]
    // = log-likelihood, then the score (MODE 1) or S_q[0..D), S_a[0..D) (MODE 2)
    constexpr int NS = MODE == 1 ? 2 : 1 + 2 * D;
    double v[NS];
    v[0] = ll;
    if constexpr (MODE == 1) {
      v[1] = score;
    } else {
#pragma unroll
      for (int a = 0; a < D; ++a) {
        v[1 + a] = sq_sum[a];
        v[1 + D + a] = sa_sum[a];
      }
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (!live) v[q] = 0.0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_down(v[q], off);
      if (lane == 0) part[(size_t)q * G.K * G.nwb + unit] = v[q];
    }
  }
}

// ------------------------------------------------------------------------------------------------------
bool dense_wave_covers(int T, int K, int D, int O) {
  if (knob_int(KNOB_DENSE_LEGACY, 0)) return false;
  // D = 2, 3 with up to four cameras; five and six cameras (O = 10, 12: the reference's fly rig run without a
  // calibration) for D = 3, one wave pair per workgroup (their rows take 2 x 24 KB of LDS per pair)
  if (!(((D == 2 || D == 3) && (O == 2 || O == 4 || O == 6 || O == 8)) || (D == 3 && (O == 10 || O == 12))))
    return false;
  return dw_units(T, K, dw_chunk_frames(T, K, O)) <= 1024;                       // depth-bound problems: every block resident at once
}

size_t dense_wave_workspace_bytes(int T, int K, int D) {
  const size_t cb = (size_t)dw_chunk_frames(T, K, 0);   // (an upper bound for O > 8, which keeps 8 frames)
  const size_t nc = ((size_t)T + cb - 1) / cb, nwb = (nc + 63) / 64;
  const size_t nv = 3 * D * D + 2 * D + 1, rec = D + D * D;
  return 2 * align_up(nc * K * nv * 8, 256) + align_up(nwb * K * nv * 8, 256) + align_up((size_t)K * rec * 8, 256) +
         align_up(nwb * K * (1 + 2 * (size_t)D) * 8, 256);   // partial sums: up to 1 + 2 D planes (MODE 2)
}

// MODE 2: nll[k] = -loglik (no non-finite substitution in the pupil loss, eks/ibl_pupil_smoother.py:551-552),
// dnll[t][k] = -sum_i (S_q[i] dq[t][k][i] + S_a[i] da[t][k][i]); block = keypoint, the unit sums added in block
// order by lane 0 of each of the 1 + 2 D sums
template <int D>
__global__ __launch_bounds__(64) void dw_ar1_finish_kernel(int K, int nwb, int n_tan, const double* __restrict__ part,
                                                          const double* __restrict__ da,
                                                          const double* __restrict__ dq, double* __restrict__ nll,
                                                          double* __restrict__ dnll) {
  __shared__ double tot[1 + 2 * D];
  const int k = blockIdx.x, q = threadIdx.x;
  if (q < 1 + 2 * D) {
    double acc = 0.0;
    for (int wb = 0; wb < nwb; ++wb) acc += part[((size_t)q * nwb + wb) * K + k];
    tot[q] = acc;
  }
  __syncthreads();
  if (q == 0) nll[k] = -tot[0];
  if (q < n_tan) {
    double g = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const size_t p = ((size_t)q * K + k) * D + i;
      g += tot[1 + i] * dq[p] + tot[1 + D + i] * da[p];
    }
    dnll[(size_t)q * K + k] = -g;
  }
}

// The pupil optimiser's iteration in THREE launches instead of five stream operations (round 4): the unit sums, the
// loss / tangent derivatives AND the Adam step of eks/ibl_pupil_smoother.py:571-594 for all chains by one block -
// thread (k, q) sums plane q of chain k, thread (k, 0) steps chain k and rewrites a, q, da, dq for the next
// evaluation, the count of chains still running is a block reduction (no memset, no atomics).  K <= 64 chains.
struct PupilStep {
  const double* latent_var;
  double* state;
  double *a, *q, *da, *dq;       // (writable views of DwAr1's inputs)
  int32_t* n_active;
  double lr, tol;
  int cap;
};
template <int D>
__global__ __launch_bounds__(512) void dw_ar1_finish_step_kernel(int K, int nwb, const double* __restrict__ part,
                                                                double* __restrict__ nll, double* __restrict__ dnll,
                                                                PupilStep P) {
  constexpr int NQ = 1 + 2 * D;
  __shared__ double tot[64][NQ + 1];
  __shared__ int running;
  const int k = threadIdx.x >> 3, q = threadIdx.x & 7;
  if (threadIdx.x == 0) running = 0;
  if (k < K && q < NQ) {
    double acc = 0.0;
    for (int wb = 0; wb < nwb; ++wb) acc += part[((size_t)q * nwb + wb) * K + k];
    tot[k][q] = acc;
  }
  __syncthreads();
  if (k < K && q < 2) {                       // the two tangents of the pupil's parameters (read BEFORE the step rewrites them)
    double g = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const size_t p = ((size_t)q * K + k) * D + i;
      g += tot[k][1 + i] * P.dq[p] + tot[k][1 + D + i] * P.da[p];
    }
    dnll[(size_t)q * K + k] = -g;
    if (q == 0) nll[k] = -tot[k][0];
  }
  __syncthreads();
  __threadfence_block();
  if (k < K && q == 0) {
    if (pupil_adam_step_chain(k, K, P.latent_var, nll, dnll, P.lr, P.tol, P.cap, P.state, P.a, P.q, P.da, P.dq))
      atomicAdd(&running, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) *P.n_active = running;
}

// mode 0: the smoother (var, ms, Vs); 1: loss + d/d log s (rconst, nll, dnll); 2: pupil loss + tangents
struct DwAr1 {
  const double *a, *q, *da, *dq;
  int n_tan;
  const PupilStep* step = nullptr;       // mode 2: fold the finish and the optimiser step into one launch
};
static int dense_wave_run(const eks_dims_t& d, int mode, const float* y, const float* var, const double* rconst,
                          const DwAr1& ar, const DenseModel& Mm, float* ms, float* Vs, double* nll, double* dnll,
                          void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim;
  const int cb = dw_chunk_frames(T, K, O);            // frames per lane: 2 / 4 for short sessions, else 8
  DwGeom G{K, T, (T + cb - 1) / cb, 0};
  G.nwb = (G.nc + 63) / 64;
  const size_t nv = 3 * D * D + 2 * D + 1, rec = D + D * D;
  const size_t need = dense_wave_workspace_bytes(T, K, D);
  if (ws_bytes < need) return EKS_ERR_WORKSPACE;
  char* p = static_cast<char*>(ws);
  double* pre_ex = reinterpret_cast<double*>(p);
  p += align_up((size_t)G.nc * K * nv * 8, 256);
  double* suf_ex = reinterpret_cast<double*>(p);
  p += align_up((size_t)G.nc * K * nv * 8, 256);
  double* agg = reinterpret_cast<double*>(p);
  p += align_up((size_t)G.nwb * K * nv * 8, 256);
  double* first = reinterpret_cast<double*>(p);
  p += align_up((size_t)K * rec * 8, 256);
  double* part = reinterpret_cast<double*>(p);
  const DenseModelPtrs M{Mm.m0, Mm.S0, Mm.A, Mm.C, Mm.Q};
  const int vs_diag = (d.flags & EKS_FLAG_VS_DIAG) ? 1 : 0;
  const int units = K * G.nwb;
  const bool two = units > 256 && O <= 8;             // more (keypoint, 64-chunk) units than CUs: 4-wave workgroups
  const dim3 grid((unsigned)(two ? (units + 1) / 2 : units)), block(two ? 256 : 128);
#define EKS_DW_SB(DD, OO, SS, MD, BB)                                                                    \
  {                                                                                                      \
    {                                                                                                    \
      ProfScope ps(MD ? "dense_score_summarize" : "dense_summarize", st);                                \
      hipLaunchKernelGGL((dw_summarize_kernel<DD, OO, SS, MD, BB>), grid, block, 0, st, G, M, Mm.s, y,    \
                         var, rconst, ar.a, ar.q, pre_ex, suf_ex, agg, first);                           \
    }                                                                                                    \
    ProfScope ps(MD ? "dense_score_replay" : "dense_replay", st);                                        \
    hipLaunchKernelGGL((dw_replay_kernel<DD, OO, SS, MD, BB>), grid, block, 0, st, G, M, Mm.s, y, var,    \
                       rconst, ar.a, ar.q, part, pre_ex, suf_ex, agg, first, ms, Vs, vs_diag);           \
  }
  // short chunks exist for one wave pair per workgroup only (they are chosen while the units are few) and up to
  // four cameras
#define EKS_DW_S(DD, OO, SS, MD)                     \
  {                                                  \
    if (SS == 1 && OO <= 8 && cb == 2)               \
      EKS_DW_SB(DD, (OO <= 8 ? OO : 8), 1, MD, 2)    \
    else if (SS == 1 && OO <= 8 && cb == 4)          \
      EKS_DW_SB(DD, (OO <= 8 ? OO : 8), 1, MD, 4)    \
    else                                             \
      EKS_DW_SB(DD, OO, SS, MD, kDwBMax)             \
  }
#define EKS_DW(DD, OO)            \
  if (two && mode == 1)           \
    EKS_DW_S(DD, OO, 2, 1)        \
  else if (two)                   \
    EKS_DW_S(DD, OO, 2, 0)        \
  else if (mode == 1)             \
    EKS_DW_S(DD, OO, 1, 1)        \
  else                            \
    EKS_DW_S(DD, OO, 1, 0)
#define EKS_DW_O(DD)                    \
  switch (O) {                          \
    case 2: EKS_DW(DD, 2) break;        \
    case 4: EKS_DW(DD, 4) break;        \
    case 6: EKS_DW(DD, 6) break;        \
    case 8: EKS_DW(DD, 8) break;        \
    default: return EKS_ERR_UNSUPPORTED; \
  }
#define EKS_DW1(DD, OO)           \
  if (mode == 1)                  \
    EKS_DW_S(DD, OO, 1, 1)        \
  else                            \
    EKS_DW_S(DD, OO, 1, 0)
  if (mode == 2) {                                    // the pupil shape only (3 states, 4 markers x 2)
    if (D != 3 || O != 8) return EKS_ERR_UNSUPPORTED;
    if (two)
      EKS_DW_S(3, 8, 2, 2)
    else
      EKS_DW_S(3, 8, 1, 2)
    if (ar.step && K <= 64 && ar.n_tan == 2)
      hipLaunchKernelGGL(dw_ar1_finish_step_kernel<3>, dim3(1), dim3(512), 0, st, K, G.nwb, part, nll, dnll, *ar.step);
    else
      hipLaunchKernelGGL(dw_ar1_finish_kernel<3>, dim3(K), dim3(64), 0, st, K, G.nwb, ar.n_tan, part, ar.da, ar.dq, nll,
                         dnll);
    return hip_status(hipGetLastError());
  }
  if (D == 2) {
    EKS_DW_O(2)
  } else if (D == 3 && O == 10) {
    EKS_DW1(3, 10)
  } else if (D == 3 && O == 12) {
    EKS_DW1(3, 12)
  } else if (D == 3) {
    EKS_DW_O(3)
  } else {
    return EKS_ERR_UNSUPPORTED;
  }
#undef EKS_DW1
#undef EKS_DW_O
#undef EKS_DW
#undef EKS_DW_S
#undef EKS_DW_SB
  if (mode == 1) return dense_score_finish(K, G.nwb, part, part + (size_t)K * G.nwb, nll, dnll, st);
  return hip_status(hipGetLastError());
}

int dense_wave_smooth(const eks_dims_t& d, const float* y, const float* var, const DenseModel& Mm, float* ms,
                      float* Vs, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!var) return EKS_ERR_NULL;
  return dense_wave_run(d, 0, y, var, nullptr, DwAr1{}, Mm, ms, Vs, nullptr, nullptr, ws, ws_bytes, st);
}

// nll[k], d nll / d log s [k] of the constant-R filter loss at Mm.s[k] (caller: Q positive definite)
int dense_wave_score(const eks_dims_t& d, const float* y, const double* rconst, const DenseModel& Mm, double* nll,
                     double* dnll, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!rconst || !nll || !dnll) return EKS_ERR_NULL;
  return dense_wave_run(d, 1, y, nullptr, rconst, DwAr1{}, Mm, nullptr, nullptr, nll, dnll, ws, ws_bytes, st);
}

// the pupil loss and its n_tan directional derivatives (eks_ar1_nll; caller: q > 0 in every coordinate)
bool dense_wave_ar1_covers(int T, int K, int D, int O) {
  return T >= 2 && D == 3 && O == 8 && dense_wave_covers(T, K, D, O) && !knob_int(KNOB_DENSE_DUAL_GRAD, 0);
}
int dense_wave_ar1_score(const eks_dims_t& d, const float* y, const float* var, const double* m0, const double* S0,
                         const double* C, const double* a, const double* q, const double* da, const double* dq,
                         int n_tan, double* nll, double* dnll, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!var || !a || !q || !da || !dq || !nll || !dnll) return EKS_ERR_NULL;
  if (n_tan < 1 || n_tan > 64) return EKS_ERR_UNSUPPORTED;
  const DenseModel Mm{m0, S0, nullptr, C, nullptr, nullptr};
  return dense_wave_run(d, 2, y, var, nullptr, DwAr1{a, q, da, dq, n_tan}, Mm, nullptr, nullptr, nll, dnll, ws,
                        ws_bytes, st);
}
// one iteration of the pupil optimiser: loss + two tangents + Adam step in three launches; returns EKS_ERR_UNSUPPORTED
// when the shape is not covered (the caller then takes eks_ar1_nll + eks_pupil_adam_step)
int dense_wave_ar1_score_step(const eks_dims_t& d, const float* y, const float* var, const double* m0, const double* S0,
                              const double* C, const double* latent_var, double lr, double tol, int cap, double* state,
                              double* a, double* q, double* da, double* dq, double* nll, double* dnll,
                              int32_t* n_active, void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints;
  if (!(d.flags & EKS_FLAG_Q_PD) || K > 64 || !dense_wave_ar1_covers(T, K, d.state_dim, d.obs_dim))
    return EKS_ERR_UNSUPPORTED;
  const PupilStep P{latent_var, state, a, q, da, dq, n_active, lr, tol, cap};
  const DenseModel Mm{m0, S0, nullptr, C, nullptr, nullptr};
  DwAr1 ar{a, q, da, dq, 2};
  ar.step = &P;
  return dense_wave_run(d, 2, y, var, nullptr, ar, Mm, nullptr, nullptr, nll, dnll, ws, ws_bytes, st);
}

}  // namespace eks

#ifdef EKS_DW_STAMPS
extern "C" int eks_debug_dw_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eks::g_dw_stamps), sizeof(eks::g_dw_stamps));
}
#endif

EKS_DEFINE_TOUCH(dense_wave)
