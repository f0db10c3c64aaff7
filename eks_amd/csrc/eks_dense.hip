// gfx950 kernels for the general (D, O) smoother: multicam linear path, D = n_latent (3..6),
// O = 2 * n_cameras (reference eks/multicam_smoother.py:409-443), and the pupil smoother's final
// pass (D = 3, O = 8).  Same three-phase chunked scan as the scalar-chain path, with float64 small
// matrices in registers; chunk elements are built predict-first (eks_dense_lane.hpp):
//   D1 dense_summarize : lane = (keypoint, chunk) -> element (A, b, C, eta, J); chunk 0's lane also
//                        updates the prior with frame 0 (the belief the scan starts from)
//   D2 dense_scan_*    : block-parallel scan of the chunk elements (general element composition
//                        through Cholesky / Woodbury forms, eks_dense_math.hpp delem_combine);
//                        forward and reverse scans run side by side in one workgroup
//   D3 dense_replay    : lane = (keypoint, chunk): exact filter, fuse, RTS; filtered beliefs go
//                        through a per-lane scratch record stream (these problems are tiny:
//                        BASELINE config 4 is 4 keypoints x 50k frames, latency- not HBM-bound)
// The losses of this path (eks_nll, eks_ar1_nll) live in eks_loss_kernels.hpp.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "eks_dense_lane.hpp"
#include "eks_internal.hpp"

namespace eks {

struct DenseGeom {
  int K, T, O;
  int B, nc;      // frames per replay lane, number of replay chunks
  int Bs, ncs;    // frames per chunk ELEMENT (summarize / scan), number of elements; B % Bs == 0,
                  // (B / Bs) divides 64 so a replay chunk's elements never straddle a scan block
};

// Sweeps of the extended filter (eks_ekf_smooth) are enqueued without host round trips: a sweep's
// kernels return at once when the previous sweep already met the tolerance.
struct Gate {
  const double* resid;   // largest change of a linearisation point in the previous sweep, or null
  double tol;
  __device__ bool closed() const { return resid != nullptr && *resid <= tol; }
};

template <int D, typename Obs>
__global__ __launch_bounds__(64) void dense_summarize_kernel(DenseGeom G, DenseModelPtrs M,
                                                            const double* __restrict__ s, Obs obs,
                                                            double* __restrict__ elems,
                                                            double* __restrict__ first, Gate gate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G.K * G.ncs || gate.closed()) return;
  const int k = idx % G.K, j = idx / G.K;
  Mat<double, D> F, sQ;
  bool fid;
  load_dynamics<double, D>(M, k, s[k], F, sQ, fid);
  const int t0 = j * G.Bs, len = min(G.Bs, G.T - t0);
  const DElem<double, D> e = dense_smooth_element_obs<D>(obs, k, t0, len, F, sQ, fid);
  store_delem<double, D>(elems + (size_t)idx * delem_doubles<D>(), e);
  if (j == 0) {   // the belief the scan starts from: the prior updated with frame 0
    Vec<double, D> m;
    Mat<double, D> P;
    load_prior<D>(M, k, m, P);
    double xl[D];   // the extended filter linearises frame 0 at the prior mean
#pragma unroll
    for (int a = 0; a < D; ++a) xl[a] = m.a[a];
    belief_update_obs<D>(obs, k, 0, xl, m, P);
    double* r = first + (size_t)k * (D + D * D);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      r[a] = m.a[a];
#pragma unroll
      for (int b = 0; b < D; ++b) r[D + a * D + b] = P.a[a][b];
    }
  }
}

// ---- scan of the chunk elements, block-parallel like the scalar path (eks_diag.hip K2):
//   DS1 scan   : block = (keypoint, 64 consecutive chunks), 128 threads: threads 0..63 run a
//                Hillis-Steele inclusive FORWARD scan of the block's elements in LDS while threads
//                64..127 run the inclusive REVERSE scan in a second LDS array (the two are
//                independent, so the depth is 6 compositions, not 12); both are written out per
//                chunk, the forward total is the block aggregate
//   DS2 blocks : per keypoint, the same two-sided scan over the block aggregates
//   (replay)   : each replay lane applies its exclusive prefix (the inclusive prefix of chunk i-1)
//                to the block's incoming belief and pulls the block's outgoing information back
//                through its exclusive suffix (the inclusive suffix of chunk i+1) - one apply and
//                one pull-back per lane instead of a third scan kernel.
constexpr int kDenseCB = 64;

template <int D>
__global__ __launch_bounds__(2 * kDenseCB) void dense_scan_kernel(DenseGeom G,
                                                                  const double* __restrict__ elems,
                                                                  double* __restrict__ pre,
                                                                  double* __restrict__ suf,
                                                                  double* __restrict__ agg, Gate gate) {
  constexpr int NV = delem_doubles<D>();
  __shared__ double lds[2 * kDenseCB * NV];
  if (gate.closed()) return;
  const int k = blockIdx.x, blk = blockIdx.y;
  const bool rev = threadIdx.x >= kDenseCB;
  const int i = threadIdx.x - (rev ? kDenseCB : 0);
  const int j = blk * kDenseCB + i;
  const bool live = j < G.ncs;
  double* mine = lds + (rev ? kDenseCB * NV : 0);
  DElem<double, D> e = live ? load_delem<double, D>(elems + ((size_t)j * G.K + k) * NV)
                            : delem_identity<double, D>();
  // records field-major in LDS (stride kDenseCB between fields: conflict-free; record-major rows of 34
  // doubles put every 8th lane on the same banks); the smoother's scans need no log-likelihood term
  store_delem<double, D>(mine + i, e, kDenseCB);
  __syncthreads();
  for (int off = 1; off < kDenseCB; off <<= 1) {
    const bool has = rev ? (i + off < kDenseCB) : (i >= off);
    DElem<double, D> other;
    if (has) other = load_delem<double, D>(mine + (rev ? i + off : i - off), kDenseCB);
    __syncthreads();
    if (has) e = rev ? delem_combine<double, D, false>(e, other) : delem_combine<double, D, false>(other, e);
    store_delem<double, D>(mine + i, e, kDenseCB);
    __syncthreads();
  }
  if (live) store_delem<double, D>((rev ? suf : pre) + ((size_t)j * G.K + k) * NV, e);
  if (!rev && i == kDenseCB - 1) store_delem<double, D>(agg + ((size_t)blk * G.K + k) * NV, e);
}

// DS2: one workgroup per keypoint scans the block aggregates the same way (threads 0..63 forward,
// 64..127 reverse), 64 aggregates per round with the running belief / information carried from
// round to round (the forward half walks the rounds upwards, the reverse half downwards).
template <int D>
__global__ __launch_bounds__(2 * kDenseCB) void dense_scan_blocks_kernel(DenseGeom G, int nblk,
                                                                         const double* __restrict__ first,
                                                                         const double* __restrict__ agg,
                                                                         double* __restrict__ bprior,
                                                                         double* __restrict__ bsuffix,
                                                                         Gate gate) {
  constexpr int REC = D + D * D;
  constexpr int NV = delem_doubles<D>();
  __shared__ double lds[2 * kDenseCB * NV];
  if (gate.closed()) return;
  const int k = blockIdx.x;
  const bool rev = threadIdx.x >= kDenseCB;
  const int i = threadIdx.x - (rev ? kDenseCB : 0);
  double* mine = lds + (rev ? kDenseCB * NV : 0);
  // every thread of a half carries its own copy of the running state (same arithmetic, same value)
  Vec<double, D> cv = vec_zero<double, D>();
  Mat<double, D> cM = mat_zero<double, D>();
  if (!rev) {
    const double* f0 = first + (size_t)k * REC;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      cv.a[a] = f0[a];
#pragma unroll
      for (int b = 0; b < D; ++b) cM.a[a][b] = f0[D + a * D + b];
    }
  }
  const int rounds = (nblk + kDenseCB - 1) / kDenseCB;
  for (int r = 0; r < rounds; ++r) {
    const int q = (rev ? rounds - 1 - r : r) * kDenseCB + i;
    const bool live = q < nblk;
    DElem<double, D> e = live ? load_delem<double, D>(agg + ((size_t)q * G.K + k) * NV)
                              : delem_identity<double, D>();
    store_delem<double, D>(mine + i, e, kDenseCB);
    __syncthreads();
    for (int off = 1; off < kDenseCB; off <<= 1) {
      const bool has = rev ? (i + off < kDenseCB) : (i >= off);
      DElem<double, D> other;
      if (has) other = load_delem<double, D>(mine + (rev ? i + off : i - off), kDenseCB);
      __syncthreads();
      if (has) e = rev ? delem_combine<double, D, false>(e, other) : delem_combine<double, D, false>(other, e);
      store_delem<double, D>(mine + i, e, kDenseCB);
      __syncthreads();
    }
    Vec<double, D> v = cv;
    Mat<double, D> Mx = cM;
    if (!rev) {
      if (i > 0) delem_apply(load_delem<double, D>(mine + (i - 1), kDenseCB), v, Mx);
      delem_apply(load_delem<double, D>(mine + (kDenseCB - 1), kDenseCB), cv, cM);
    } else {
      if (i + 1 < kDenseCB) delem_back(load_delem<double, D>(mine + (i + 1), kDenseCB), v, Mx);
      delem_back(load_delem<double, D>(mine, kDenseCB), cv, cM);
    }
    if (live) {
      double* w = (rev ? bsuffix : bprior) + ((size_t)q * G.K + k) * REC;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        w[a] = v.a[a];
#pragma unroll
        for (int b = 0; b < D; ++b) w[D + a * D + b] = Mx.a[a][b];
      }
    }
    __syncthreads();
  }
}

template <int D, bool EKF, typename Obs, bool SCORE = false>
__global__ __launch_bounds__(64) void dense_replay_kernel(DenseGeom G, DenseModelPtrs M,
                                                         const double* __restrict__ s, Obs obs,
                                                         const double* __restrict__ pre,
                                                         const double* __restrict__ suf,
                                                         const double* __restrict__ bprior,
                                                         const double* __restrict__ bsuffix,
                                                         double* __restrict__ filt,
                                                         float* __restrict__ ms,
                                                         float* __restrict__ Vs, int vs_diag,
                                                         double* __restrict__ xlin,
                                                         double* __restrict__ ll_chunk,
                                                         double* __restrict__ resid, Gate gate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G.K * G.nc || gate.closed()) return;
  constexpr int REC = D + D * D;
  const int k = idx % G.K, j = idx / G.K;
  Mat<double, D> F, sQ;
  bool fid;
  load_dynamics<double, D>(M, k, s[k], F, sQ, fid);
  constexpr int NV = delem_doubles<D>();
  Vec<double, D> m, eta;
  Mat<double, D> P, J;
  // this lane's frames are covered by the elements ea .. eb (one element on the linear path)
  const int sub = G.B / G.Bs;
  const int ea = j * sub, eb = min(ea + sub - 1, G.ncs - 1);
  const int blk = ea / kDenseCB, ia = ea % kDenseCB, ib = eb % kDenseCB;
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
  // exclusive prefix of the first / exclusive suffix of the last element inside the block of 64
  if (ia > 0) delem_apply(load_delem<double, D>(pre + ((size_t)(ea - 1) * G.K + k) * NV), m, P);
  if (ib + 1 < kDenseCB && eb + 1 < G.ncs)
    delem_back(load_delem<double, D>(suf + ((size_t)(eb + 1) * G.K + k) * NV), eta, J);
  if (j == 0) load_prior<D>(M, k, m, P);   // chunk 0 replays frame 0's update of the prior itself
  const int t0 = j * G.B, len = min(G.B, G.T - t0);
  double ll = 0.0, ch = 0.0;
  dense_replay_chunk_obs<D, EKF, Obs, SCORE>(obs, G.K, k, t0, len, F, sQ, fid, m, P, eta, J,
                                             filt ? filt + (size_t)t0 * REC * G.K + k : nullptr, ms, Vs,
                                             vs_diag != 0, EKF ? xlin + ((size_t)k * G.T + t0) * D : nullptr, &ll,
                                             &ch, (size_t)G.K);
  if constexpr (SCORE) {                 // per-chunk partial sums, [chunk][keypoint]: ll_chunk, then resid as the score
    ll_chunk[idx] = ll;
    resid[idx] = ch;
  }
  if constexpr (EKF) {
    ll_chunk[idx] = ll;
    // non-negative doubles order like their bit patterns; a NaN (diverged linearisation) has the
    // largest pattern and keeps the following sweeps open
    atomicMax(reinterpret_cast<unsigned long long*>(resid),
              (unsigned long long)__double_as_longlong(ch != ch ? 1e300 : ch));
  }
}

// ------------------------------------------------------------------------------------------
constexpr int kDenseSmoothChunk = 32;   // frames per lane in the smoother (the scan is parallel)

// Frames per lane for a (T, K) problem.  Shorter chunks buy lanes and cut the sequential depth of
// summarize / replay; the two-level scan costs about the same up to 64 x 64 chunks per keypoint.
// Linear path: 16 frames while that does not oversubscribe the device (measured on BASELINE config
// 4, T = 50 000 x K = 4: 0.202 ms at 32, 0.148 at 16, 0.154 at 8), else 32.  Extended filter: 32
// (its sweeps converge chunk-wise: 16-frame chunks are shorter than the filter's memory and need
// 8+ sweeps where 32-frame chunks need 3).  EKS_DENSE_CHUNK overrides (measurement knob).
static int dense_chunk(int T, int K, bool ekf = false) {
  const int forced = knob_int(KNOB_DENSE_CHUNK, 0);
  if (forced >= 2 && forced <= 256) return forced;
  if (ekf) return kDenseSmoothChunk;
  return (long long)K * ((T + 15) / 16) <= (1 << 18) ? 16 : kDenseSmoothChunk;
}

// Which organisation a (T, K, D, O) smoothing problem takes, and its workspace:
//   wave : narrow sessions, eks_dense_wave.hip
//   runs : keypoint-major kernels with the per-lane scan, eks_dense_wide.hip - chunk elements, the belief /
//          information pair per chunk, the scan's upper levels, two partial sums per lane (SCORE form): no
//          per-frame stream (288 GB hold 50 000 frames x ~170 000 keypoints of D = 3 this way; the generic layout's
//          float64 filtered-belief stream alone would take 96 B per keypoint-frame)
//   else : the generic kernels (any D <= 6, O <= 64; also the keypoint-major kernels with the tree scan behind
//          EKS_DENSE_TREE_SCAN) with prefix / suffix elements per chunk and the filtered-belief stream
enum DensePath { kDenseWave, kDenseRuns, kDenseGeneric };
static DensePath dense_path(int T, int K, int D, int O) {
  if (dense_wave_covers(T, K, D, O)) return kDenseWave;
  if (dense_wide_covers(D, O, dense_chunk(T, K)) && !knob_int(KNOB_DENSE_TREE_SCAN, 0)) return kDenseRuns;
  return kDenseGeneric;
}
struct RunsLayout {
  double *elems, *chunk_in, *chunk_out, *scratch, *first, *part_ll, *part_score;
  size_t bytes;
};
static RunsLayout runs_layout(int K, int D, int nc, char* base) {
  const size_t nv = 3 * D * D + 2 * D + 1, rec = D + D * D;
  RunsLayout L;
  size_t off = 0;
  auto take = [&](size_t doubles) {
    double* p = reinterpret_cast<double*>(base + off);
    off += align_up(doubles * 8, 256);
    return p;
  };
  L.elems = take((size_t)nc * K * nv);
  L.chunk_in = take((size_t)nc * K * rec);
  L.chunk_out = take((size_t)nc * K * rec);
  L.scratch = take(dense_wide_scan_scratch_doubles(K, D, nc));
  L.first = take((size_t)K * rec);
  L.part_ll = take((size_t)nc * K);
  L.part_score = take((size_t)nc * K);
  L.bytes = off;
  return L;
}

struct GenericLayout {     // chunk elements, their inclusive prefix / suffix per chunk, block aggregates and
  double *elems, *pre, *suf, *agg, *bprior, *bsuffix, *filt, *first;   // boundaries, the filtered-belief stream
  size_t bytes;
};
static GenericLayout generic_layout(int T, int K, int D, int nc, char* base) {
  const size_t nv = 3 * D * D + 2 * D + 1, rec = D + D * D;
  const size_t nblk = (nc + kDenseCB - 1) / kDenseCB;
  GenericLayout L;
  size_t off = 0;
  auto take = [&](size_t doubles) {
    double* p = reinterpret_cast<double*>(base + off);
    off += align_up(doubles * 8, 256);
    return p;
  };
  L.elems = take((size_t)nc * K * nv);
  L.pre = take((size_t)nc * K * nv);
  L.suf = take((size_t)nc * K * nv);
  L.agg = take(nblk * K * nv);
  L.bprior = take(nblk * K * rec);
  L.bsuffix = take(nblk * K * rec);
  L.filt = take((size_t)T * K * rec);
  L.first = take((size_t)K * rec);
  L.bytes = off;
  return L;
}

size_t dense_smooth_workspace_bytes(int T, int K, int D, int O) {
  const int B = dense_chunk(T, K), nc = (T + B - 1) / B;
  switch (dense_path(T, K, D, O)) {
    case kDenseWave: return dense_wave_workspace_bytes(T, K, D);
    case kDenseRuns: return runs_layout(K, D, nc, nullptr).bytes;
    default: return generic_layout(T, K, D, nc, nullptr).bytes;
  }
}

int dense_smooth(const eks_dims_t& d, const float* y, const float* var, const DenseModel& Mm,
                 float* ms, float* Vs, void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim;
  if (D < 1 || D > 6 || O < 1 || O > 64) return EKS_ERR_UNSUPPORTED;
  if (ws_bytes < dense_smooth_workspace_bytes(T, K, D, O)) return EKS_ERR_WORKSPACE;
  // narrow sessions (configs[3]: 4 keypoints) are depth-bound: two launches with the scan in wave
  // shuffles and the filtered beliefs in LDS (eks_dense_wave.hip); wide ones stream keypoint-major
  const DensePath path = dense_path(T, K, D, O);
  if (path == kDenseWave) return dense_wave_smooth(d, y, var, Mm, ms, Vs, ws, ws_bytes, st);
  DenseGeom G{K, T, O, dense_chunk(T, K), 0, 0, 0};
  G.nc = (T + G.B - 1) / G.B;
  G.Bs = G.B;
  G.ncs = G.nc;
  const int nblk = (G.nc + kDenseCB - 1) / kDenseCB;
  const DenseModelPtrs M{Mm.m0, Mm.S0, Mm.A, Mm.C, Mm.Q};
  if (path == kDenseRuns) {
    const RunsLayout L = runs_layout(K, D, G.nc, static_cast<char*>(ws));
    const int vs_diag = (d.flags & EKS_FLAG_VS_DIAG) ? 1 : 0;
    int rc;
    {
      ProfScope ps("dense_summarize", st);
      rc = dense_wide_summarize(T, K, D, O, G.B, G.nc, M, Mm.s, y, var, nullptr, L.elems, 1, L.first, st);
      if (rc != EKS_OK) return rc;
    }
    {
      ProfScope ps("dense_scan", st);
      rc = dense_wide_scan(K, D, G.nc, L.elems, L.first, L.scratch, L.chunk_in, L.chunk_out, st);
      if (rc != EKS_OK) return rc;
    }
    ProfScope ps("dense_replay", st);
    return dense_wide_replay(T, K, D, O, G.B, G.nc, M, Mm.s, y, var, nullptr, nullptr, nullptr, nullptr, nullptr,
                             nullptr, nullptr, L.chunk_in, L.chunk_out, ms, Vs, vs_diag, st);
  }
  const GenericLayout L = generic_layout(T, K, D, G.nc, static_cast<char*>(ws));
  double *elems = L.elems, *pre = L.pre, *suf = L.suf, *agg = L.agg, *bprior = L.bprior, *bsuffix = L.bsuffix,
         *filt = L.filt, *first = L.first;
  const int lanes = K * G.nc;
  const int vs_diag = (d.flags & EKS_FLAG_VS_DIAG) ? 1 : 0;
  const Gate open{nullptr, 0.0};
  // (EKS_DENSE_TREE_SCAN: the keypoint-major summarize / replay around the tree scan of whole elements)
  const bool wide = dense_wide_covers(D, O, G.B);
  EKS_DISPATCH_D(D, {
    const LinearObs<DD> obs = make_linear_obs<DD>(y, var, K, O, M);
    {
      ProfScope ps("dense_summarize", st);
      if (wide) {
        const int rc = dense_wide_summarize(T, K, D, O, G.B, G.nc, M, Mm.s, y, var, nullptr, elems, 0, first, st);
        if (rc != EKS_OK) return rc;
      } else {
        hipLaunchKernelGGL((dense_summarize_kernel<DD, LinearObs<DD>>), dim3((lanes + 63) / 64),
                           dim3(64), 0, st, G, M, Mm.s, obs, elems, first, open);
      }
    }
    {
      ProfScope ps("dense_scan", st);
      const dim3 sgrid(K, nblk);
      hipLaunchKernelGGL(dense_scan_kernel<DD>, sgrid, dim3(2 * kDenseCB), 0, st, G, elems, pre, suf,
                         agg, open);
      hipLaunchKernelGGL(dense_scan_blocks_kernel<DD>, dim3(K), dim3(2 * kDenseCB), 0, st, G, nblk,
                         first, agg, bprior, bsuffix, open);
    }
    {
      ProfScope ps("dense_replay", st);
      if (wide) {
        const int rc = dense_wide_replay(T, K, D, O, G.B, G.nc, M, Mm.s, y, var, nullptr, nullptr, nullptr, pre, suf,
                                         bprior, bsuffix, nullptr, nullptr, ms, Vs, vs_diag, st);
        if (rc != EKS_OK) return rc;
      } else {
        hipLaunchKernelGGL((dense_replay_kernel<DD, false, LinearObs<DD>>), dim3((lanes + 63) / 64),
                           dim3(64), 0, st, G, M, Mm.s, obs, pre, suf, bprior, bsuffix, filt, ms, Vs,
                           vs_diag, nullptr, nullptr, nullptr, open);
      }
    }
  })
  return hip_status(hipGetLastError());
}

// ---- SCORE form: the optimiser's loss and its derivative from the smoother's own kernels ---------------
// nll[k], d nll / d log s [k] of the constant-R filter loss at Mm.s[k] (eks/core.py:640-652), for Q positive
// definite: the value from the exact filter inside the replay kernels, the derivative from the smoothing
// distribution (Fisher's identity; eks_dense_wave.hip has the formula).  Narrow sessions take the wave kernels,
// wider ones the keypoint-major kernels with the per-lane scan; other shapes have no SCORE form (the caller keeps
// the dual-number kernels of eks_loss.hip).
bool dense_score_covers(int T, int K, int D, int O) {
  (void)K;
  return T >= 2 && D >= 1 && D <= 6 && O >= 1 && O <= 64 && !knob_int(KNOB_DENSE_DUAL_GRAD, 0);
}
size_t dense_score_workspace_bytes(int T, int K, int D, int O) { return dense_smooth_workspace_bytes(T, K, D, O); }
int dense_score(const eks_dims_t& d, const float* y, const double* rconst, const DenseModel& Mm, double* nll,
                double* dnll, void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim;
  if (!dense_score_covers(T, K, D, O)) return EKS_ERR_UNSUPPORTED;
  if (ws_bytes < dense_score_workspace_bytes(T, K, D, O)) return EKS_ERR_WORKSPACE;
  const DensePath path = dense_path(T, K, D, O);
  if (path == kDenseWave) return dense_wave_score(d, y, rconst, Mm, nll, dnll, ws, ws_bytes, st);
  const int B = dense_chunk(T, K), nc = (T + B - 1) / B;
  const DenseModelPtrs M{Mm.m0, Mm.S0, Mm.A, Mm.C, Mm.Q};
  if (path == kDenseGeneric) {
    // any D <= 6, O <= 64: the generic kernels with constant variances (layout of dense_smooth: prefix / suffix
    // elements per chunk, the filtered-belief stream; per-chunk partial sums where the chunk elements were - they
    // are dead once the scan has run)
    DenseGeom G{K, T, O, B, nc, B, nc};
    const int nblk = (nc + kDenseCB - 1) / kDenseCB;
    const GenericLayout L = generic_layout(T, K, D, nc, static_cast<char*>(ws));
    double *elems = L.elems, *pre = L.pre, *suf = L.suf, *agg = L.agg, *bprior = L.bprior, *bsuffix = L.bsuffix,
           *filt = L.filt, *first = L.first;
    double* part_ll = elems;
    double* part_score = elems + (size_t)nc * K;
    const int lanes = K * nc;
    const Gate open{nullptr, 0.0};
    EKS_DISPATCH_D(D, {
      const ConstLinearObs<DD> obs = make_const_linear_obs<DD>(y, rconst, K, O, M);
      {
        ProfScope ps("dense_score_summarize", st);
        hipLaunchKernelGGL((dense_summarize_kernel<DD, ConstLinearObs<DD>>), dim3((lanes + 63) / 64), dim3(64), 0, st,
                           G, M, Mm.s, obs, elems, first, open);
      }
      {
        ProfScope ps("dense_score_scan", st);
        hipLaunchKernelGGL(dense_scan_kernel<DD>, dim3(K, nblk), dim3(2 * kDenseCB), 0, st, G, elems, pre, suf, agg,
                           open);
        hipLaunchKernelGGL(dense_scan_blocks_kernel<DD>, dim3(K), dim3(2 * kDenseCB), 0, st, G, nblk, first, agg,
                           bprior, bsuffix, open);
      }
      ProfScope ps("dense_score_replay", st);
      hipLaunchKernelGGL((dense_replay_kernel<DD, false, ConstLinearObs<DD>, true>), dim3((lanes + 63) / 64), dim3(64),
                         0, st, G, M, Mm.s, obs, pre, suf, bprior, bsuffix, filt, nullptr, nullptr, 0, nullptr,
                         part_ll, part_score, open);
    })
    const int rc0 = hip_status(hipGetLastError());
    if (rc0 != EKS_OK) return rc0;
    return dense_score_finish(K, nc, part_ll, part_score, nll, dnll, st);
  }
  const RunsLayout L = runs_layout(K, D, nc, static_cast<char*>(ws));
  int rc;
  {
    ProfScope ps("dense_score_summarize", st);
    rc = dense_wide_summarize(T, K, D, O, B, nc, M, Mm.s, y, nullptr, rconst, L.elems, 1, L.first, st);
    if (rc != EKS_OK) return rc;
  }
  {
    ProfScope ps("dense_score_scan", st);
    rc = dense_wide_scan(K, D, nc, L.elems, L.first, L.scratch, L.chunk_in, L.chunk_out, st);
    if (rc != EKS_OK) return rc;
  }
  {
    ProfScope ps("dense_score_replay", st);
    rc = dense_wide_replay(T, K, D, O, B, nc, M, Mm.s, y, nullptr, rconst, L.part_ll, L.part_score, nullptr, nullptr,
                           nullptr, nullptr, L.chunk_in, L.chunk_out, nullptr, nullptr, 0, st);
    if (rc != EKS_OK) return rc;
  }
  return dense_score_finish(K, nc, L.part_ll, L.part_score, nll, dnll, st);
}

// ---- extended Kalman filter / smoother with calibrated pinhole cameras ------------------------
// Reference: run_kalman_smoother(h_fn=...) (eks/core.py:188-190, :274-295) as driven by the
// calibrated branch of eks/multicam_smoother.py:369-407; D = 3, O = 2 * n_cams.
//
// The extended filter linearises frame t at its own predicted mean, which makes the recursion
// sequential in the reference.  Here it is solved as a fixed point instead: with linearisation
// points X fixed the model is a linear one with time-varying observation rows, which the chunked
// scan handles in parallel over (chain, chunk); the replay then runs the TRUE extended filter
// inside each chunk from the chunk's entry belief and writes its predicted means back as the new
// X.  At the fixed point (entry beliefs consistent with X) the result IS the sequential extended
// filter; every sweep makes at least one more chunk exact, in practice 2-4 sweeps reach 1e-10
// because the filter forgets its entry belief within a few frames.  Sweeps are gated on the
// device (Gate), so the whole solve is one enqueue without host round trips.
constexpr int kEkfMaxSweeps = 64;
// The replay lanes keep 32-frame chunks (the sweeps converge chunk-wise) but the chunk ELEMENTS are
// built over half chunks: twice the lanes and half the sequential depth in the summarize kernel.
constexpr int kEkfSub = 2;

__global__ __launch_bounds__(64) void ekf_finish_kernel(int K, int nc, int n_sweeps, double tol,
                                                       const double* __restrict__ ll_chunk,
                                                       const double* __restrict__ resid,
                                                       double* __restrict__ nll,
                                                       double* __restrict__ info) {
  const int k = blockIdx.x, i = threadIdx.x;
  double acc = 0.0;
  for (int j = i; j < nc; j += 64) acc += ll_chunk[(size_t)j * K + k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (i == 0) {
    if (nll) nll[k] = -acc;
    if (k == 0 && info) {
      int ran = 1;
      while (ran < n_sweeps && !(resid[ran - 1] <= tol)) ++ran;
      info[0] = (double)ran;          // filter sweeps executed
      info[1] = resid[ran - 1];       // largest relative change of a linearisation point in the last
    }
  }
}

// the residual of the last filter sweep that ran (the slots of the sweeps gated after it are still zero) -> resid[slot]
__global__ void ekf_last_resid_kernel(int n_sweeps, double tol, double* __restrict__ resid, int slot) {
  int ran = 1;
  while (ran < n_sweeps && !(resid[ran - 1] <= tol)) ++ran;
  resid[slot] = resid[ran - 1];
}

static size_t ekf_ws_layout(int T, int K, bool smooth, double** ptrs, char* base) {
  constexpr int D = 3;
  const int B = dense_chunk(T, K, true), nc = (T + B - 1) / B;
  const int Bs = B % kEkfSub == 0 ? B / kEkfSub : B, ncs = (T + Bs - 1) / Bs;
  const int nblk = (ncs + kDenseCB - 1) / kDenseCB;
  const size_t nv = 3 * D * D + 2 * D + 1, rec = D + D * D;
  const size_t sizes[10] = {(size_t)ncs * K * nv * 8,  (size_t)ncs * K * nv * 8,
                            (size_t)ncs * K * nv * 8,  (size_t)nblk * K * nv * 8,
                            (size_t)nblk * K * rec * 8, (size_t)nblk * K * rec * 8,
                            smooth ? (size_t)T * K * rec * 8 : 0, (size_t)K * rec * 8,
                            (size_t)nc * K * 8,         (size_t)(kEkfMaxSweeps + 2) * 8};
  size_t off = 0;
  for (int i = 0; i < 10; ++i) {
    if (ptrs) ptrs[i] = reinterpret_cast<double*>(base + off);
    off += align_up(sizes[i], 256);
  }
  return off;
}

size_t ekf_smooth_workspace_bytes(int T, int K, int smooth) {
  return ekf_ws_layout(T, K, smooth != 0, nullptr, nullptr);
}

int ekf_smooth(const eks_dims_t& d, int n_data_keypoints, const float* y, const float* var,
               const double* rconst, const DenseModel& Mm, const double* cams, int n_cams,
               double* xlin, int max_sweeps, double tol, float* ms, float* Vs, double* nll,
               double* info, void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, O = d.obs_dim, Kd = n_data_keypoints;
  if (D != 3 || n_cams < 1 || O != 2 * n_cams || Kd < 1 || K % Kd != 0) return EKS_ERR_UNSUPPORTED;
  if (max_sweeps < 1 || max_sweeps > kEkfMaxSweeps) return EKS_ERR_SHAPE;
  if ((var == nullptr) == (rconst == nullptr)) return EKS_ERR_SHAPE;
  const bool smooth = ms != nullptr;
  if (ws_bytes < ekf_smooth_workspace_bytes(T, K, smooth)) return EKS_ERR_WORKSPACE;
  double* w[10];
  ekf_ws_layout(T, K, smooth, w, static_cast<char*>(ws));
  double *elems = w[0], *pre = w[1], *suf = w[2], *agg = w[3], *bprior = w[4], *bsuffix = w[5],
         *filt = w[6], *first = w[7], *ll_chunk = w[8], *resid = w[9];
  DenseGeom G{K, T, O, dense_chunk(T, K, true), 0, 0, 0};
  G.nc = (T + G.B - 1) / G.B;
  G.Bs = G.B % kEkfSub == 0 ? G.B / kEkfSub : G.B;
  G.ncs = (T + G.Bs - 1) / G.Bs;
  const int nblk = (G.ncs + kDenseCB - 1) / kDenseCB, lanes = K * G.nc, lanes_k1 = K * G.ncs;
  const DenseModelPtrs M{Mm.m0, Mm.S0, Mm.A, nullptr, Mm.Q};
  const PinholeObs obs{y, ObsNoise{var, rconst}, Kd, O, T, cams, xlin};
  const int vs_diag = (d.flags & EKS_FLAG_VS_DIAG) ? 1 : 0;
  const dim3 sgrid(K, nblk);
  hipError_t e = hipMemsetAsync(resid, 0, (kEkfMaxSweeps + 2) * 8, st);
  if (e != hipSuccess) return hip_status(e);
  // (build_gate: the element / scan kernels of the sweep; replay_gate: its replay)
  auto sweep = [&](const Gate& build_gate, const Gate& replay_gate, double* resid_out, bool with_smoother) {
    hipLaunchKernelGGL((dense_summarize_kernel<3, PinholeObs>), dim3((lanes_k1 + 63) / 64), dim3(64), 0,
                       st, G, M, Mm.s, obs, elems, first, build_gate);
    hipLaunchKernelGGL(dense_scan_kernel<3>, sgrid, dim3(2 * kDenseCB), 0, st, G, elems, pre, suf, agg,
                       build_gate);
    hipLaunchKernelGGL(dense_scan_blocks_kernel<3>, dim3(K), dim3(2 * kDenseCB), 0, st, G, nblk,
                       first, agg, bprior, bsuffix, build_gate);
    hipLaunchKernelGGL((dense_replay_kernel<3, true, PinholeObs>), dim3((lanes + 63) / 64), dim3(64),
                       0, st, G, M, Mm.s, obs, pre, suf, bprior, bsuffix, with_smoother ? filt : nullptr,
                       with_smoother ? ms : nullptr, Vs, vs_diag, xlin, ll_chunk, resid_out, replay_gate);
  };
  {
    ProfScope ps("ekf_filter_sweeps", st);
    for (int i = 0; i < max_sweeps; ++i) {
      const Gate g{i > 0 ? resid + i - 1 : nullptr, tol};
      sweep(g, g, resid + i, false);
    }
  }
  if (smooth) {
    // Once the filter sweeps have met the tolerance (the last sweep that ran moved no linearisation point by more
    // than tol; the slots of the sweeps gated after it are still zero) the elements, prefixes and suffixes in the
    // workspace ARE the converged ones - rebuilt at the new points they would differ by ~tol (1e-10 against 1e-5 bars) -
    // so the smoothing sweep is its replay alone (round 5: three launches and a third of a sweep's time less); it
    // rebuilds them only when the sweeps ran out unconverged.  The shortcut is for tight tolerances only: with a loose
    // caller tol (1e-2, say) "converged" points may still be 1e-2 from the elements in the workspace, so the rebuild is
    // skipped only below 1e-8.
    ProfScope ps("ekf_smooth_sweep", st);
    hipLaunchKernelGGL(ekf_last_resid_kernel, dim3(1), dim3(1), 0, st, max_sweeps, tol, resid, kEkfMaxSweeps + 1);
    sweep(Gate{resid + kEkfMaxSweeps + 1, tol < 1e-8 ? tol : 1e-8}, Gate{nullptr, 0.0}, resid + kEkfMaxSweeps, true);
  }
  hipLaunchKernelGGL(ekf_finish_kernel, dim3(K), dim3(64), 0, st, K, G.nc, max_sweeps, tol, ll_chunk,
                     resid, nll, info);
  return hip_status(hipGetLastError());
}

}  // namespace eks

EKS_DEFINE_TOUCH(dense)
