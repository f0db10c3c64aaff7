// gfx950 kernels for the constant-R filter NLL on scalar chains (the loss of eks/core.py:640-650)
// evaluated for many smoothing-parameter values at once:
//   * grid mode  : n_cand values shared by all keypoints (BASELINE.json config 3: 64 candidates)
//   * Adam mode  : one value per keypoint, with d nll / d log s (forward-mode dual numbers)
//
//   N1 diag_nll_summarize : wave = (64-chain tile, time chunk, group of NCL candidates); the waves of
//                           a block are the candidate groups of ONE (tile, chunk).  Each lane keeps
//                           NCL candidate filters in registers.  Chunk 0 (short) starts from a
//                           known state and passes through the transient regimes; every later
//                           chunk enters with the converged filter variance (eks_nll_lane.hpp):
//                           two FMAs per frame and candidate (four while rho^t is alive), the
//                           candidates interleaved in the frame loop, rows of y loaded through a
//                           buffer resource (scalar row offsets) 8 frames ahead.  Summaries are
//                           relative to the chunk's reference state y_0 / c.  The chunk length
//                           is chosen so that the grid is a whole number of 256-CU rounds.
//                           (Tried and dropped, round 1: packed fp32, a transient/steady kernel
//                           split with LDS-staged tiles, deeper load rings - see DESIGN.md.)
//   N2 diag_nll_assemble  : thread = (chain, candidate): walks the chunk summaries in time order
//                           in float64, the chains of a keypoint are summed by wave shuffles
//                           -> nll[K][n_cand] (and dnll); with few (keypoint, candidate) pairs -
//                           the Adam loop - a tree variant composes the summaries in log depth.
// y is read once from HBM: 4 B per chain-frame regardless of the candidate count.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "eks_adam.hpp"
#include "eks_internal.hpp"
#include "eks_nll_lag.hpp"
#include "eks_nll_lane.hpp"

namespace eks {

constexpr int kNllChunk = 4096;      // frames per lane, grid mode (measured best with NCL = 8)
constexpr int kNllChunkGrad = 512;   // frames per lane, Adam mode (one candidate: needs more lanes)
constexpr int kNclGrid = 8;
constexpr int kNllChunkMin = 2048;
constexpr int kNllChunk0 = 1024;     // frames in chunk 0, grid mode (C3 at the time: 768 0.253 ms, 1024 0.267,
                                     // 1536 0.287, 3200 0.341; 512 0.342 - chunk 1 then starts before the
                                     // variance has converged and falls back to the exact-entry summary)

struct NllWs {
  // planes indexed [(j * ncp + c) * N + n]
  float *A, *b, *C, *eta, *J;            // values
  float *dA, *db, *dC, *deta, *dJ;       // derivatives (grad mode only)
  double *ell, *dell;
  float* xr;                             // chunk reference states [ncn][N] (candidate-independent)
  int ncp;                               // padded candidate count
};

struct NllGeom {
  int N, T, D, ncn, BN;
  int B0;                // frames in chunk 0 (chunk j >= 1 covers [B0 + (j-1) BN, B0 + j BN))
  int nt_log2, ntile, ngrp, n_cand, per_keypoint;
  int converged_entry;   // chunks j >= 1 may assume the filter variance has converged (float path)
};

// Row loads of y through a buffer resource (gfx950 `buffer_load_dword v, v_off, s[rsrc], s_row offen`):
// the row offset lives in an SGPR, so the 16 loads per 16 frames of the steady loop cost no VALU
// address arithmetic.  Offsets are 32-bit: used when a chunk spans < 2 GiB (launch code).
struct BufferRows {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff, row_bytes;
  __device__ __forceinline__ float operator()(int i) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (unsigned)i * row_bytes, 0));
  }
};

template <typename R, int NCL, bool UNIT, bool TILE64>
__global__ __launch_bounds__(512) void diag_nll_summarize_kernel(NllGeom G, DiagModel M, NllWs W,
                                          const float* __restrict__ y,
                                          const double* __restrict__ rconst,
                                          const double* __restrict__ s_cand, AdamFuse F) {
  // TILE64: a wave holds 64 chains of ONE chunk and one candidate group, so everything derived
  // from the wave index is scalar (readfirstlane tells the compiler): chunk bounds, loop trip
  // counts and the row addresses of y (SGPR base + per-lane offset, no VALU address arithmetic)
  int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (TILE64) wave = __builtin_amdgcn_readfirstlane(wave);
  const int lane = threadIdx.x & 63;
  const int g = wave % G.ngrp;
  const int rest = wave / G.ngrp;
  const int tile = rest % G.ntile;
  const int cg = rest / G.ntile;
  const int nt = 1 << G.nt_log2;
  const int n = tile * nt + (lane & (nt - 1));
  const int j = TILE64 ? cg : cg * (64 >> G.nt_log2) + (lane >> G.nt_log2);
  if (n >= G.N || j >= G.ncn) return;
  const int k = n / G.D, d = n - k * G.D;
  // Adam loop: keypoints whose optimiser block has stopped (or reached the cap) need no loss any more -
  // a wave none of whose chains is still being optimised returns at once (wave-uniform, so the arithmetic
  // of the chains that go on is untouched; reference eks/core.py:669-674 masks finished lanes the same way)
  if (F.state != nullptr) {
    const bool running = adam_block_running(F.state, F.kp_block[k], F.cap);
    if (!__any(running)) return;
  }
  const size_t dd = (size_t)k * G.D * G.D + (size_t)d * (G.D + 1);
  const double q = M.Q[dd];
  double sq[NCL];
#pragma unroll
  for (int c = 0; c < NCL; ++c) {
    // contiguous candidate groups: neighbours on the log-s grid converge at similar speed, so a
    // wave leaves the transient regimes as early as its candidates allow
    const int ci = min(g * NCL + c, G.n_cand - 1);
    const double s = G.per_keypoint ? s_cand[(size_t)k * G.n_cand + ci] : s_cand[ci];
    sq[c] = s * q;
  }
  const int t0 = j == 0 ? 0 : G.B0 + (j - 1) * G.BN;
  const int len = j == 0 ? min(G.B0, G.T) : min(G.BN, G.T - t0);
  NllElem<R> out[NCL];
  if constexpr (TILE64) {
    // buffer loads: resource based at the chunk's first row (scalar), per-lane byte offset 4 n,
    // row offset i * 4 N in an SGPR - a load costs no VALU address arithmetic
    const BufferRows ld{__builtin_amdgcn_make_buffer_rsrc(
                            const_cast<float*>(y + (size_t)t0 * G.N + tile * 64), 0, 0x7FFFFFFF, 0x00020000),
                        (unsigned)(lane * 4), (unsigned)(G.N * 4)};
    nll_summarize_chunk<R, NCL, UNIT>(ld, t0, len, rconst[n], M.A[dd], M.C[dd], sq, out,
                                      G.converged_entry != 0);
  } else {
    const RowsByPointer ld{y + (size_t)t0 * G.N + n, (size_t)G.N};
    nll_summarize_chunk<R, NCL, UNIT>(ld, t0, len, rconst[n], M.A[dd], M.C[dd], sq, out,
                                      G.converged_entry != 0);
  }
  if (g == 0) W.xr[(size_t)j * G.N + n] = out[0].xref;
#pragma unroll
  for (int c = 0; c < NCL; ++c) {
    const int ci = g * NCL + c;
    if (ci >= G.n_cand) continue;
    const size_t o = ((size_t)j * W.ncp + ci) * G.N + n;
    W.A[o] = val(out[c].e.A);
    W.b[o] = val(out[c].e.b);
    W.C[o] = val(out[c].e.C);
    W.eta[o] = val(out[c].e.eta);
    W.J[o] = val(out[c].e.J);
    W.ell[o] = out[c].ell;
    if constexpr (sizeof(R) == sizeof(Dual)) {
      W.dA[o] = der(out[c].e.A);
      W.db[o] = der(out[c].e.b);
      W.dC[o] = der(out[c].e.C);
      W.deta[o] = der(out[c].e.eta);
      W.dJ[o] = der(out[c].e.J);
      W.dell[o] = out[c].dell;
    }
  }
}


template <bool GRAD>
__global__ __launch_bounds__(256) void diag_nll_assemble_kernel(NllGeom G, DiagModel M, NllWs W,
                                                               int K, double* __restrict__ nll,
                                                               double* __restrict__ dnll) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= K * G.n_cand) return;
  const int k = idx % K, ci = idx / K;
  using RD = typename std::conditional<GRAD, DualD, double>::type;
  RD tot = RD(0.0);
  for (int d = 0; d < G.D; ++d) {
    const int n = k * G.D + d;
    const size_t dd = (size_t)k * G.D * G.D + (size_t)d * (G.D + 1);
    auto get = [&](int j, Elem<RD>& e, RD& ell, double& xr) {
      const size_t o = ((size_t)j * W.ncp + ci) * G.N + n;
      xr = (double)W.xr[(size_t)j * G.N + n];
      if constexpr (GRAD) {
        e.A = DualD(W.A[o], W.dA[o]);
        e.b = DualD(W.b[o], W.db[o]);
        e.C = DualD(W.C[o], W.dC[o]);
        e.eta = DualD(W.eta[o], W.deta[o]);
        e.J = DualD(W.J[o], W.dJ[o]);
        ell = DualD(W.ell[o], W.dell[o]);
      } else {
        e.A = W.A[o];
        e.b = W.b[o];
        e.C = W.C[o];
        e.eta = W.eta[o];
        e.J = W.J[o];
        ell = W.ell[o];
      }
    };
    tot = tot + nll_assemble<RD>(G.ncn, M.m0[(size_t)k * G.D + d], M.S0[dd], get);
  }
  // eks/core.py:650: non-finite -> 1e12 (gradient of the constant branch is 0)
  const double v = -val(tot);
  const bool fin = isfinite(v);
  nll[(size_t)k * G.n_cand + ci] = fin ? v : 1e12;
  if constexpr (GRAD) dnll[(size_t)k * G.n_cand + ci] = fin ? -der(tot) : 0.0;
}

// N2 for D a power of two (the usual D = 2): thread = (chain, candidate) with the chain index
// fastest - coalesced plane reads, half the sequential depth - and the D chain log-likelihoods of
// a keypoint, which sit in adjacent lanes, are summed with wave shuffles.
template <bool GRAD>
__global__ __launch_bounds__(256) void diag_nll_assemble_chain_kernel(NllGeom G, DiagModel M, NllWs W,
                                                                     double* __restrict__ nll,
                                                                     double* __restrict__ dnll) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= G.N * G.n_cand) return;         // whole groups of D lanes leave together
  const int n = idx % G.N, ci = idx / G.N;
  const int k = n / G.D, d = n - k * G.D;
  using RD = typename std::conditional<GRAD, DualD, double>::type;
  const size_t dd = (size_t)k * G.D * G.D + (size_t)d * (G.D + 1);
  auto get = [&](int j, Elem<RD>& e, RD& ell, double& xr) {
    const size_t o = ((size_t)j * W.ncp + ci) * G.N + n;
    xr = (double)W.xr[(size_t)j * G.N + n];
    if constexpr (GRAD) {
      e.A = DualD(W.A[o], W.dA[o]);
      e.b = DualD(W.b[o], W.db[o]);
      e.C = DualD(W.C[o], W.dC[o]);
      e.eta = DualD(W.eta[o], W.deta[o]);
      e.J = DualD(W.J[o], W.dJ[o]);
      ell = DualD(W.ell[o], W.dell[o]);
    } else {
      e.A = W.A[o];
      e.b = W.b[o];
      e.C = W.C[o];
      e.eta = W.eta[o];
      e.J = W.J[o];
      ell = W.ell[o];
    }
  };
  const RD tot = nll_assemble<RD>(G.ncn, M.m0[(size_t)k * G.D + d], M.S0[dd], get);
  double v = val(tot), g = der(tot);
  for (int off = 1; off < G.D; off <<= 1) {
    v += __shfl_xor(v, off);
    g += __shfl_xor(g, off);
  }
  if (d != 0) return;
  v = -v;
  const bool fin = isfinite(v);              // eks/core.py:650
  nll[(size_t)k * G.n_cand + ci] = fin ? v : 1e12;
  if constexpr (GRAD) dnll[(size_t)k * G.n_cand + ci] = fin ? -g : 0.0;
}

// N2' tree variant for few chain-streams (the Adam loop: one candidate per keypoint, ~200 chunk
// summaries per chain): block = (64 lanes, D chains of one (keypoint, candidate)).  Each lane
// composes a contiguous run of chunk elements, the 64 partial elements are composed in time order
// by a 6-level tree through LDS, lane 0 pushes the prior through the result; the D chain
// log-likelihoods are summed through LDS.  Depth ~ ncn/64 + 6 compositions instead of ncn.
template <typename RD>
struct NllAcc {
  Elem<RD> e;
  RD ell;
  double xr;   // reference of the run's FIRST chunk: the run is a function of (x_in - xr)
};

template <typename RD>
__device__ inline NllAcc<RD> nll_acc_combine(const NllAcc<RD>& i, const NllAcc<RD>& j) {
  // run i hands its (absolute) outgoing mean to run j, which wants it relative to ITS reference
  Elem<RD> ie = i.e;
  ie.b = i.e.b - RD(j.xr);
  const RD den = RD(1.0) + ie.C * j.e.J;
  const RD inv = rcp(den);
  NllAcc<RD> o;
  o.ell = i.ell + j.ell - RD(0.5) * log_with_rcp(den, inv) +
          (j.e.eta * ie.b + RD(0.5) * j.e.eta * j.e.eta * ie.C - RD(0.5) * j.e.J * ie.b * ie.b) * inv;
  o.e = elem_combine(ie, j.e);
  o.xr = i.xr;
  return o;
}

constexpr int kAsmLanes = 64;

template <bool GRAD>
__global__ void diag_nll_assemble_tree_kernel(NllGeom G, DiagModel M, NllWs W, int K,
                                              double* __restrict__ nll, double* __restrict__ dnll,
                                              AdamFuse F) {
  using RD = typename std::conditional<GRAD, DualD, double>::type;
  constexpr int NF = GRAD ? 13 : 7;               // 6 element fields (+ derivatives) + reference
  extern __shared__ double lds[];                 // [D][NF][64] + [D][2]
  const int i = threadIdx.x, d = threadIdx.y;
  const int k = blockIdx.x % K, ci = blockIdx.x / K;
  if (F.state != nullptr) {                       // Adam loop (see AdamFuse)
    if (blockIdx.x == 0 && i == 0 && d == 0) *F.n_active_next = 0;
    if (!adam_block_running(F.state, F.kp_block[k], F.cap)) return;   // block-uniform
  }
  const int n = k * G.D + d;
  double* my = lds + (size_t)d * NF * kAsmLanes;
  auto get = [&](int j) {
    const size_t o = ((size_t)j * W.ncp + ci) * G.N + n;
    NllAcc<RD> a;
    a.xr = (double)W.xr[(size_t)j * G.N + n];
    if constexpr (GRAD) {
      a.e.A = DualD(W.A[o], W.dA[o]);
      a.e.b = DualD(W.b[o], W.db[o]);
      a.e.C = DualD(W.C[o], W.dC[o]);
      a.e.eta = DualD(W.eta[o], W.deta[o]);
      a.e.J = DualD(W.J[o], W.dJ[o]);
      a.ell = DualD(W.ell[o], W.dell[o]);
    } else {
      a.e.A = W.A[o];
      a.e.b = W.b[o];
      a.e.C = W.C[o];
      a.e.eta = W.eta[o];
      a.e.J = W.J[o];
      a.ell = W.ell[o];
    }
    return a;
  };
  auto put = [&](int slot, const NllAcc<RD>& a) {
    const RD f[6] = {a.e.A, a.e.b, a.e.C, a.e.eta, a.e.J, a.ell};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      my[q * kAsmLanes + slot] = val(f[q]);
      if constexpr (GRAD) my[(6 + q) * kAsmLanes + slot] = der(f[q]);
    }
    my[(NF - 1) * kAsmLanes + slot] = a.xr;
  };
  auto take = [&](int slot) {
    RD f[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      if constexpr (GRAD)
        f[q] = DualD(my[q * kAsmLanes + slot], my[(6 + q) * kAsmLanes + slot]);
      else
        f[q] = my[q * kAsmLanes + slot];
    }
    NllAcc<RD> a;
    a.e.A = f[0]; a.e.b = f[1]; a.e.C = f[2]; a.e.eta = f[3]; a.e.J = f[4]; a.ell = f[5];
    a.xr = my[(NF - 1) * kAsmLanes + slot];
    return a;
  };
  const int per = (G.ncn + kAsmLanes - 1) / kAsmLanes;
  const int j0 = i * per, j1 = min(G.ncn, j0 + per);
  const int nlive = (G.ncn + per - 1) / per;      // lanes that own at least one chunk
  NllAcc<RD> acc;
  if (j0 < j1) {
    acc = get(j0);
    for (int j = j0 + 1; j < j1; ++j) acc = nll_acc_combine(acc, get(j));
  }
  for (int half = 1; half < nlive; half <<= 1) {
    const int span = half << 1;
    const bool send = (i & (span - 1)) == half && i < nlive;
    const bool recv = (i & (span - 1)) == 0 && i + half < nlive;
    if (send) put(i, acc);
    __syncthreads();
    if (recv) acc = nll_acc_combine(acc, take(i + half));
  }
  double* tot = lds + (size_t)G.D * NF * kAsmLanes;
  if (i == 0) {
    const size_t dd = (size_t)k * G.D * G.D + (size_t)d * (G.D + 1);
    const RD m = RD(M.m0[(size_t)k * G.D + d] - acc.xr), P = RD(M.S0[dd]);   // relative to the reference
    const RD den = RD(1.0) + acc.e.J * P;
    const RD inv = rcp(den);
    const RD ll = acc.ell - RD(0.5) * log_with_rcp(den, inv) +
                  (acc.e.eta * m + RD(0.5) * acc.e.eta * acc.e.eta * P - RD(0.5) * acc.e.J * m * m) * inv;
    tot[2 * d] = val(ll);
    tot[2 * d + 1] = der(ll);
  }
  __syncthreads();
  if (i == 0 && d == 0) {
    double v = 0.0, g = 0.0;
    for (int q = 0; q < G.D; ++q) {
      v += tot[2 * q];
      g += tot[2 * q + 1];
    }
    v = -v;
    const bool fin = isfinite(v);              // eks/core.py:650
    nll[(size_t)k * G.n_cand + ci] = fin ? v : 1e12;
    if constexpr (GRAD) dnll[(size_t)k * G.n_cand + ci] = fin ? -g : 0.0;
    if constexpr (GRAD) {
      // every optimiser block is this one keypoint: the step follows at once (this thread reads back
      // the two values it has just written), no separate launch
      if (F.state != nullptr && F.step_in_kernel) {
        if (adam_step_block(F.kp_block[k], F.offs, F.members, nll, dnll, F.lr, F.lo, F.hi, F.tol, F.cap,
                            F.state, F.s_keypoint))
          atomicAdd(F.n_active_cur, 1);
      }
    }
  }
}

// The WHOLE optimiser loop of a short session in one launch (round 4).  With one keypoint per optimiser block - the
// reference's default, blocks = [] (eks/core.py:223-224) - keypoints do not interact at all (the vmapped
// lax.while_loop of eks/core.py:654-681 masks finished lanes, nothing else), so a workgroup can own a keypoint for
// all iterations of an eks_adam_run call: thread (i, d) summarises chunk i of chain d (64 ... 256 chunks of 8+ frames,
// value + d/d log s in float32 duals as everywhere on this path), the summaries are composed by the LDS tree of
// diag_nll_assemble_tree_kernel in float64 duals, thread (0, 0) finishes, applies the Adam step and the stop
// rule, and the block goes round again with the new s - no launch, no global exchange, no other block.  Sessions
// of the reference's own size (2 000 frames, a handful of keypoints) spent 32 us per iteration in two launches
// whose lanes each walked 512 frames; here a lane walks 32 - 64.  Up to kPersistMaxT frames; longer sessions keep the
// per-iteration kernels, whose chunks spread over the chip.
constexpr int kPersistMaxT = 16384;

template <bool UNIT>
__global__ __launch_bounds__(512) void diag_nll_adam_persist_kernel(int T, int N, int D, int cl, int n_iters, DiagModel M,
                                             const float* __restrict__ y, const double* __restrict__ rconst,
                                             double* __restrict__ nll, double* __restrict__ dnll, AdamFuse F,
                                             int32_t* __restrict__ n_active) {
  constexpr int NF = 13;                           // 6 element fields + derivatives + reference
  extern __shared__ double lds[];                  // [D][NF][LN] | tot[D][2] | flag
  const int i = threadIdx.x, d = threadIdx.y;
  const int LN = blockDim.x;                       // lanes (= chunks) per chain: 64 ... 256
  const int k = blockIdx.x;
  const int kb = F.kp_block[k];
  const int n = k * D + d;
  double* my = lds + (size_t)d * NF * LN;
  double* tot = lds + (size_t)D * NF * LN;
  int* running = reinterpret_cast<int*>(tot + 2 * D);
  const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
  const double r_n = rconst[n], a_n = M.A[dd], c_n = M.C[dd], q_n = M.Q[dd];
  const int t0 = i * cl, len = max(0, min(cl, T - t0));
  const int nlive = (T + cl - 1) / cl;             // lanes that own a chunk
  const RowsByPointer ld{y + (size_t)t0 * N + n, (size_t)N};
  auto put = [&](int slot, const NllAcc<DualD>& a) {
    const DualD f[6] = {a.e.A, a.e.b, a.e.C, a.e.eta, a.e.J, a.ell};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      my[q * LN + slot] = f[q].v;
      my[(6 + q) * LN + slot] = f[q].d;
    }
    my[(NF - 1) * LN + slot] = a.xr;
  };
  auto take = [&](int slot) {
    DualD f[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) f[q] = DualD(my[q * LN + slot], my[(6 + q) * LN + slot]);
    NllAcc<DualD> a;
    a.e.A = f[0]; a.e.b = f[1]; a.e.C = f[2]; a.e.eta = f[3]; a.e.J = f[4]; a.ell = f[5];
    a.xr = my[(NF - 1) * LN + slot];
    return a;
  };
  bool alive = adam_block_running(F.state, kb, F.cap);               // block-uniform
  for (int it = 0; it < n_iters && alive; ++it) {
    NllAcc<DualD> acc;
    if (len > 0) {
      double sq[1] = {F.s_keypoint[k] * q_n};
      NllElem<Dual> out[1];
      nll_summarize_chunk<Dual, 1, UNIT>(ld, t0, len, r_n, a_n, c_n, sq, out, false);
      acc.e.A = DualD(out[0].e.A.v, out[0].e.A.d);
      acc.e.b = DualD(out[0].e.b.v, out[0].e.b.d);
      acc.e.C = DualD(out[0].e.C.v, out[0].e.C.d);
      acc.e.eta = DualD(out[0].e.eta.v, out[0].e.eta.d);
      acc.e.J = DualD(out[0].e.J.v, out[0].e.J.d);
      acc.ell = DualD(out[0].ell, out[0].dell);
      acc.xr = (double)out[0].xref;
    }
    for (int half = 1; half < nlive; half <<= 1) {
      const int span = half << 1;
      const bool send = (i & (span - 1)) == half && i < nlive;
      const bool recv = (i & (span - 1)) == 0 && i + half < nlive;
      if (send) put(i, acc);
      __syncthreads();
      if (recv) acc = nll_acc_combine(acc, take(i + half));
      __syncthreads();                            // (the slots are written again at the next level / iteration)
    }
    if (i == 0) {
      const DualD m = DualD(M.m0[(size_t)k * D + d] - acc.xr), P = DualD(M.S0[dd]);   // relative to the reference
      const DualD den = DualD(1.0) + acc.e.J * P;
      const DualD inv = rcp(den);
      const DualD ll = acc.ell - DualD(0.5) * log_with_rcp(den, inv) +
                       (acc.e.eta * m + DualD(0.5) * acc.e.eta * acc.e.eta * P - DualD(0.5) * acc.e.J * m * m) * inv;
      tot[2 * d] = ll.v;
      tot[2 * d + 1] = ll.d;
    }
    __syncthreads();
    if (i == 0 && d == 0) {
      double v = 0.0, g = 0.0;
      for (int q = 0; q < D; ++q) {
        v += tot[2 * q];
        g += tot[2 * q + 1];
      }
      v = -v;
      const bool fin = isfinite(v);                // eks/core.py:650
      nll[k] = fin ? v : 1e12;
      dnll[k] = fin ? -g : 0.0;
      *running = adam_step_block(kb, F.offs, F.members, nll, dnll, F.lr, F.lo, F.hi, F.tol, F.cap, F.state,
                                 F.s_keypoint)
                     ? 1
                     : 0;
    }
    __syncthreads();                               // the new s (global, written by this block) and the verdict
    alive = *running != 0;
  }
  if (i == 0 && d == 0 && alive) atomicAdd(n_active, 1);
}

// may eks_adam_run hand a whole call (n_iters iterations) to diag_nll_adam_persist_kernel?
bool diag_nll_adam_persist_ok(int T, int K, int D, int n_blocks) {
  // (64 lanes x D <= 512 threads; up to 512 keypoints every workgroup is resident and the keypoints advance side by
  //  side - wider sessions go round by round and the chip-wide single-launch kernel streams them faster)
  //  Measured (tools/small_session_time.py, tools/adam_time.py; per-iteration kernels -> whole loop): 2 000 frames x
  //  4 / 16 / 64 keypoints 3.3 / 3.7 / 5.1 -> 1.9 / 2.3 / 3.1 ms, 10 000 x 16 4.4 -> 4.1 ms, but 10 000 x 64 - which the
  //  chip-wide single-launch kernel serves - 2.8 -> 3.9 ms: beyond 4 096 frames only sessions of at most 32 chains.
  if (n_blocks != K || K > 512 || T < 2 || T > kPersistMaxT || D < 1 || D > 8 || knob_int(KNOB_ADAM_PER_ITERATION, 0))
    return false;
  return T <= 4096 || K * D <= 32;
}

int diag_nll_adam_persist(const eks_dims_t& d, const float* y, const double* rconst, const DiagModel& M, int n_iters,
                          double* nll, double* dnll, const AdamFuse& F, int32_t* n_active, hipStream_t st) {
  const int T = d.n_frames, K = d.n_keypoints, D = d.state_dim, N = K * D;
  // lanes per chain: 64, doubled while a lane's chunk would exceed 64 frames (measured on 2 000 x 4: 64 lanes of 32
  // frames 2.0 ms for the whole run_kalman_smoother call, 256 lanes of 8 frames 2.5 ms - every lane pays the set-up of
  // its chunk's recursion and every doubling a tree level with two workgroup barriers), up to what the workgroup
  // (512 threads) and its LDS (13 doubles per lane and chain) hold
  int LN = 64;
  // (512 threads: the lane body wants ~200 VGPRs - at 1 024 threads it spilled 400 bytes per lane)
  while (LN < 256 && LN * 64 < T && 2 * LN * D <= 512 && (size_t)2 * LN * D * 13 * sizeof(double) <= 140 * 1024) LN *= 2;
  int cl = ((T + LN - 1) / LN + 7) / 8 * 8;
  if (cl < 8) cl = 8;
  const size_t shm = ((size_t)D * 13 * LN + 2 * D + 2) * sizeof(double);
  if (shm > 64 * 1024) {
    static const bool raised = [] {
      const int lim = 150 * 1024;
      return hipFuncSetAttribute(reinterpret_cast<const void*>(diag_nll_adam_persist_kernel<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess &&
             hipFuncSetAttribute(reinterpret_cast<const void*>(diag_nll_adam_persist_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lim) == hipSuccess;
    }();
    if (!raised) return EKS_ERR_UNSUPPORTED;
  }
  if (d.flags & EKS_FLAG_UNIT_AC)
    hipLaunchKernelGGL(diag_nll_adam_persist_kernel<true>, dim3(K), dim3(LN, D), shm, st, T, N, D, cl, n_iters, M, y,
                       rconst, nll, dnll, F, n_active);
  else
    hipLaunchKernelGGL(diag_nll_adam_persist_kernel<false>, dim3(K), dim3(LN, D), shm, st, T, N, D, cl, n_iters, M, y,
                       rconst, nll, dnll, F, n_active);
  return hip_status(hipGetLastError());
}

// N1+N2 in ONE launch for the Adam loop (one value of s per keypoint, value + d/d log s; round 3).  The
// two-launch form above spent 36 us in the chunk summaries and 21 us + a launch gap composing them: the
// tree kernel is one block per keypoint whose lanes gather 13 planes x ~200 chunks with a stride of a whole
// plane row (1.3 M scattered loads per evaluation).  Here
//   * block = (64-chain tile, group of kGfWaves CONSECUTIVE chunks), wave = chunk, lane = chain: the
//     chunk summaries never leave registers as float32 - they are promoted to float64 duals and composed
//     in time order by a 3-level tree through LDS; wave 0 stores ONE group summary (13 doubles per chain,
//     [tile][group][field][lane]: coalesced);
//   * the block that takes the LAST ticket of its tile (agent-scope release / acquire around an atomic
//     counter - nobody waits for anybody inside one evaluation, so no co-residency assumption) reads the tile's group
//     summaries back
//     (coalesced, a contiguous run of groups per wave), composes them the same way, pushes the prior
//     through the result, sums the D chains of a keypoint with wave shuffles, writes nll / dnll and - when
//     every optimiser block is one keypoint - applies the Adam step of its 64 / D keypoints on the spot;
//   * all blocks of a tile share its chains, so the "every keypoint of the tile has stopped" exit is
//     block-uniform and the ticket count of a tile is all or nothing.
// The chunk length is chosen so that the grid is a whole number of 256-CU rounds (C3: 8 tiles x 32 groups).
// Round 5: where the tile's poles allow, the chunks past the first are summarised from a converged entry and their
// terms SUMMED (gf_evaluate's first branch: no compositions at all); the form described above is what remains for slow
// poles.  Round 6: the search of the reference's default mode no longer comes here (eks_lag_adam.hip: no pass over y per
// iteration) - this kernel serves eks_nll with a gradient and the optimiser's iterations on the shapes the lag form does
// not take (more than four chains per keypoint); round 5's in-launch loop over iterations (cooperative launch, tagged
// hand-off words, give-up path) went with it.
constexpr int kGfWaves = 8;
constexpr int kGfFields = 13;          // (A, b, C, eta, J, ell) x (value, derivative) + reference state
constexpr int kGfChunkMin = 256;      // (shorter chunks never leave the full recursion: nll_summarize_chunk)

struct GradFuseWs {
  double* grp;          // [ntile][ngroups][kGfFields][64]
  int32_t* tickets;     // [ntile], a multiple of ngroups between evaluations (every evaluation of a tile adds ngroups)
  int ngroups;
  int conv_allowed;     // converged-entry chunk terms where the poles allow (EKS_NLL_GRAD_TREE=1: always the tree)
};

__device__ __forceinline__ void gf_put(double* slot, const NllAcc<DualD>& a) {
  const DualD f[6] = {a.e.A, a.e.b, a.e.C, a.e.eta, a.e.J, a.ell};
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    slot[q * 64] = f[q].v;
    slot[(6 + q) * 64] = f[q].d;
  }
  slot[12 * 64] = a.xr;
}
__device__ __forceinline__ NllAcc<DualD> gf_take(const double* slot) {
  DualD f[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) f[q] = DualD(slot[q * 64], slot[(6 + q) * 64]);
  NllAcc<DualD> a;
  a.e.A = f[0]; a.e.b = f[1]; a.e.C = f[2]; a.e.eta = f[3]; a.e.J = f[4]; a.ell = f[5];
  a.xr = slot[12 * 64];
  return a;
}
// compose the accumulators of the block's first `nvalid` waves in wave order; the result is wave 0's.
// Every LDS slot is written once per call.
__device__ __forceinline__ void gf_block_tree(double* lds, int w, int lane, int nvalid, NllAcc<DualD>& acc) {
#pragma unroll
  for (int half = 1; half < kGfWaves; half <<= 1) {
    const int span = half << 1;
    if ((w & (span - 1)) == half && w < nvalid) gf_put(lds + (size_t)w * kGfFields * 64 + lane, acc);
    __syncthreads();
    if ((w & (span - 1)) == 0 && w + half < nvalid)
      acc = nll_acc_combine(acc, gf_take(lds + (size_t)(w + half) * kGfFields * 64 + lane));
  }
}

// Publication of a block's results to blocks on OTHER XCDs (each XCD has its own L2).  A release fence at
// agent scope is `buffer_wbl2 sc1` - it walks the whole L2 and cost 5 us per block here (in-kernel stamps,
// tools/gf_stamps.py).  Instead every published value is stored with an agent-scope atomic store (gfx950:
// `global_store ... sc1`, written through the L2), the wave waits for the stores to be acknowledged
// (s_waitcnt vmcnt(0): the same ordering the memory model's own release sequences rely on between two
// agent-scope atomics) and only then takes the ticket with an agent-scope atomic.  The
// readers use an agent-scope acquire fence (`buffer_inv sc1`) after observing the atomic.  No plain store
// is published this way, so no dirty L2 line has to be written back.
__device__ __forceinline__ void gf_publish(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gf_put_published(double* slot, const NllAcc<DualD>& a) {
  const DualD f[6] = {a.e.A, a.e.b, a.e.C, a.e.eta, a.e.J, a.ell};
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    gf_publish(slot + q * 64, f[q].v);
    gf_publish(slot + (6 + q) * 64, f[q].d);
  }
  gf_publish(slot + 12 * 64, a.xr);
}
// This ordering argument is about gfx950's caches, not about the HIP memory model (which promises nothing for a
// relaxed store followed by a relaxed read-modify-write): an sc1 store is acknowledged only once it has been
// written through the XCD's L2, so after vmcnt(0) every published value is in memory before the ticket moves.
// The library is built for gfx950 only; any other target must take the portable form (a release at agent
// scope on the ticket, i.e. the 5 us L2 write-back) - hence the guard.  tests/test_gpu_kernels.py repeats the
// fused evaluation against EKS_NLL_GRAD_UNFUSED bit for bit (a stale summary would show as a mismatch).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "gf_publish / gf_stores_acknowledged rely on gfx950's write-through sc1 stores: use a release at agent scope on the ticket for other targets"
#endif
__device__ __forceinline__ void gf_stores_acknowledged() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
}


// a chain's log-likelihood and derivative (wave 0 of the tile's last block) -> the keypoint's loss, gradient and step.
// `st`: the optimiser state of the lane's keypoint, requested beside the group slots.
__device__ __forceinline__ void gf_finish(const NllGeom& G, const AdamFuse& F, int lane, int k, int d, int kb, bool running,
                                          double v, double g, const AdamRegs& st, double* __restrict__ nll,
                                          double* __restrict__ dnll) {
  for (int off = 1; off < G.D; off <<= 1) {          // the D chains of a keypoint sit in adjacent lanes
    v += __shfl_xor(v, off);
    g += __shfl_xor(g, off);
  }
  bool still = false;
  if (running && d == 0) {
    v = -v;
    const bool fin = isfinite(v);                    // eks/core.py:650
    const double L = fin ? v : 1e12, dL = fin ? -g : 0.0;
    nll[k] = L;
    dnll[k] = dL;
    double s_new;
    if (F.state != nullptr && F.step_in_kernel)
      still = adam_step_single<false>(kb, k, st, L, dL, F.lr, F.lo, F.hi, F.tol, F.cap, F.state, F.s_keypoint, &s_new);
  }
  if (F.state != nullptr && F.step_in_kernel) {
    const int cnt = __popcll(__ballot(still));
    if (lane == 0 && cnt) atomicAdd(F.n_active_cur, cnt);
  }
}
// (wave 0 of the last block, behind the acquire fence) the state gf_finish will need
__device__ __forceinline__ AdamRegs gf_state_request(const AdamFuse& F, int w, int d, int kb, bool running) {
  if (w == 0 && d == 0 && running && F.state != nullptr && F.step_in_kernel) return adam_load(F.state, kb);
  return AdamRegs{0, 0, 0, 0, 0, 1.0};
}

// ---- round 5: the evaluation without compositions.  With every chunk past the first summarised from a converged
// entry (nll_conv_chunk_dual: A = 0), chunk j's term of the log-likelihood needs only chunk j - 1's outgoing mean:
//     ll = ll_0(prior) + sum_{j >= 1} [ ell_j + eta_j mr_j - J_j mr_j^2 / 2 ],   mr_j = b_{j-1} - xref_j
// (diag_nll_assemble_par_kernel's identity, here with d / d log s riding along in float64).  Inside a block the b's
// meet in LDS; the term of a block's FIRST chunk needs the previous block's last b, so each block publishes
// (sum of its finished terms, its last b, its first chunk's eta, J, xref: 9 doubles per lane instead of 13) and the
// tile's last block adds the deferred terms - independent loads and a sum where the tree had log-depth compositions.
// frames per row buffer of the converged-entry chunk body (two buffers: 2 ROWS rows requested ahead per wave).  8 where the
// launch fills the chip (C3: the streaming phase runs at the HBM peak, 16 measured slower); 16 for small launches - a
// single-tile session has ~50 workgroups, each limited by what it keeps in flight
constexpr int kGfRowsFull = 8, kGfRowsFew = 16;
#ifndef EKS_GF_FEW_BLOCKS
#define EKS_GF_FEW_BLOCKS 128
#endif
constexpr int kGfFewBlocks = EKS_GF_FEW_BLOCKS;     // launches of at most this many workgroups take kGfRowsFew (A/B builds: 0 / 100000)
constexpr int kGcSum = 0, kGcB = 2, kGcEta = 4, kGcJ = 6, kGcXr = 8;      // field rows of a group's slot ([field][64])

// everything one evaluation needs that does not change between iterations
template <typename LD>
struct GfCtx {
  const NllGeom& G;
  const DiagModel& M;
  const GradFuseWs& W;
  const AdamFuse& F;
  const LD& ld;
  double* lds;
  int* last_flag;
  double* mine;
  double* __restrict__ nll;
  double* __restrict__ dnll;
  int w, lane, tile, grp, j, nvalid, t0, len, k, d, kb;
  size_t dd;
  double r_n, a_n, c_n, q_n;
};

// one evaluation at s_now.  Returns whether this block was its tile's last (block-uniform); in that block wave 0 has
// written the keypoints' loss and gradient and applied the step.
template <bool UNIT, int ROWS, typename LD>
__device__ __forceinline__ bool gf_evaluate(const GfCtx<LD>& X, double s_now, bool running, const float (&pre)[ROWS]) {
  const NllGeom& G = X.G;
  const GradFuseWs& W = X.W;
  const int w = X.w, lane = X.lane;
  const double sq_n = s_now * X.q_n;
  // every chunk past the first by its converged-entry summary when the tile's poles allow it (the same answer in
  // every block of the tile: it depends on the chains' constants alone)
  bool conv = false;
  ConvConst KC;
  if (W.conv_allowed && G.ncn > 1) {
    KC = conv_const<UNIT>(X.r_n, X.a_n, X.c_n, sq_n);
    conv = __all(conv_chunk_ok(KC, min(G.B0, G.BN)));
  }
  if (conv) {
    double* bx = X.lds;                                // [kGfWaves][2][64]: the mean each chunk hands on
    double* part = X.lds + kGfWaves * 2 * 64;          // [kGfWaves][2][64]: the waves' terms
    DualD term(0.0), eta(0.0), Jc(0.0);
    double xr = 0.0;
    if (X.j == 0) {                                    // chunk 0: known entry state, applied to the prior here
      double sq[1] = {sq_n};
      NllElem<Dual> out[1];
      nll_summarize_chunk<Dual, 1, UNIT>(X.ld, X.t0, X.len, X.r_n, X.a_n, X.c_n, sq, out, false, G.T);
      const DualD A(out[0].e.A.v, out[0].e.A.d), b(out[0].e.b.v, out[0].e.b.d), e0(out[0].e.eta.v, out[0].e.eta.d),
          J0(out[0].e.J.v, out[0].e.J.d), ell(out[0].ell, out[0].dell);
      const DualD mr = DualD(X.M.m0[(size_t)X.k * G.D + X.d] - (double)out[0].xref), P = DualD(X.M.S0[X.dd]);
      const DualD den = DualD(1.0) + J0 * P;
      const DualD inv = rcp(den);
      term = ell - DualD(0.5) * log_with_rcp(den, inv) +
             (e0 * mr + DualD(0.5) * e0 * e0 * P - DualD(0.5) * J0 * mr * mr) * inv;
      const DualD bn = A * inv * (mr + P * e0) + b;     // (b is absolute, mr relative to xref)
      bx[(w * 2 + 0) * 64 + lane] = bn.v;
      bx[(w * 2 + 1) * 64 + lane] = bn.d;
    } else if (X.j < G.ncn) {
      ConvDual o;
      nll_conv_chunk_dual<UNIT, ROWS>(X.ld, X.len, KC, X.a_n, X.c_n, o, X.len >= ROWS ? pre : nullptr);
      term = DualD(o.ell, o.dell);
      eta = DualD(o.eta, o.deta);
      Jc = DualD(o.J, o.dJ);
      xr = (double)o.xref;
      bx[(w * 2 + 0) * 64 + lane] = o.b;
      bx[(w * 2 + 1) * 64 + lane] = o.db;
    }
    __syncthreads();
    if (w > 0 && X.j < G.ncn) {
      const DualD mr = DualD(bx[((w - 1) * 2 + 0) * 64 + lane], bx[((w - 1) * 2 + 1) * 64 + lane]) - DualD(xr);
      term = term + eta * mr - DualD(0.5) * Jc * mr * mr;
    }
    part[(w * 2 + 0) * 64 + lane] = term.v;
    part[(w * 2 + 1) * 64 + lane] = term.d;
    __syncthreads();
    if (w == 0) {
      double sv = 0.0, sd = 0.0;
#pragma unroll
      for (int q = 0; q < kGfWaves; ++q) {
        sv += part[(q * 2 + 0) * 64 + lane];
        sd += part[(q * 2 + 1) * 64 + lane];
      }
      double* mine = X.mine;
      gf_publish(mine + (kGcSum + 0) * 64, sv);
      gf_publish(mine + (kGcSum + 1) * 64, sd);
      gf_publish(mine + (kGcB + 0) * 64, bx[((X.nvalid - 1) * 2 + 0) * 64 + lane]);
      gf_publish(mine + (kGcB + 1) * 64, bx[((X.nvalid - 1) * 2 + 1) * 64 + lane]);
      gf_publish(mine + (kGcEta + 0) * 64, eta.v);
      gf_publish(mine + (kGcEta + 1) * 64, eta.d);
      gf_publish(mine + (kGcJ + 0) * 64, Jc.v);
      gf_publish(mine + (kGcJ + 1) * 64, Jc.d);
      gf_publish(mine + kGcXr * 64, xr);
      gf_stores_acknowledged();
      if (lane == 0) *X.last_flag = (atomicAdd(W.tickets + X.tile, 1) + 1) % W.ngroups == 0;
    }
    __syncthreads();
    if (!*X.last_flag) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other blocks' slots (and, loop mode, the optimiser state)
    // ---- the tile's last block: every group's sum and deferred first term at once, wave w taking groups w, w + 8, ...
    // (four groups' fields - and the keypoints' optimiser state - requested together: one trip to memory, not four)
    const AdamRegs st = gf_state_request(X.F, w, X.d, X.kb, running);
    const double* base = W.grp + (size_t)X.tile * W.ngroups * kGfFields * 64 + lane;
    DualD tot(0.0);
    for (int gb = w; gb < W.ngroups; gb += 4 * kGfWaves) {
      double f[4][9];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int g = min(gb + q * kGfWaves, W.ngroups - 1);
        const double* sl = base + (size_t)g * kGfFields * 64;
        const double* pv = g > 0 ? sl - (size_t)kGfFields * 64 : sl;
        f[q][0] = sl[(kGcSum + 0) * 64]; f[q][1] = sl[(kGcSum + 1) * 64];
        f[q][2] = pv[(kGcB + 0) * 64];   f[q][3] = pv[(kGcB + 1) * 64];
        f[q][4] = sl[(kGcEta + 0) * 64]; f[q][5] = sl[(kGcEta + 1) * 64];
        f[q][6] = sl[(kGcJ + 0) * 64];   f[q][7] = sl[(kGcJ + 1) * 64];
        f[q][8] = sl[kGcXr * 64];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int g = gb + q * kGfWaves;
        if (g < W.ngroups) {
          tot = tot + DualD(f[q][0], f[q][1]);
          if (g > 0) {
            const DualD mr = DualD(f[q][2], f[q][3]) - DualD(f[q][8]);
            tot = tot + DualD(f[q][4], f[q][5]) * mr - DualD(0.5) * DualD(f[q][6], f[q][7]) * mr * mr;
          }
        }
      }
    }
    part[(w * 2 + 0) * 64 + lane] = tot.v;             // (every wave is past its reads of `part`: the barrier above)
    part[(w * 2 + 1) * 64 + lane] = tot.d;
    __syncthreads();
    if (w == 0) {
      double v = 0.0, g = 0.0;
#pragma unroll
      for (int q = 0; q < kGfWaves; ++q) {
        v += part[(q * 2 + 0) * 64 + lane];
        g += part[(q * 2 + 1) * 64 + lane];
      }
      gf_finish(G, X.F, lane, X.k, X.d, X.kb, running, v, g, st, X.nll, X.dnll);
    }
    return true;
  }
  // ---- exact-entry summaries composed in order: block tree, group slots, the last block's walk and tree (round 3)
  NllAcc<DualD> acc;
  if (X.j < G.ncn) {
    double sq[1] = {sq_n};
    NllElem<Dual> out[1];
    nll_summarize_chunk<Dual, 1, UNIT>(X.ld, X.t0, X.len, X.r_n, X.a_n, X.c_n, sq, out, false, G.T);
    acc.e.A = DualD(out[0].e.A.v, out[0].e.A.d);
    acc.e.b = DualD(out[0].e.b.v, out[0].e.b.d);
    acc.e.C = DualD(out[0].e.C.v, out[0].e.C.d);
    acc.e.eta = DualD(out[0].e.eta.v, out[0].e.eta.d);
    acc.e.J = DualD(out[0].e.J.v, out[0].e.J.d);
    acc.ell = DualD(out[0].ell, out[0].dell);
    acc.xr = (double)out[0].xref;
  }
  gf_block_tree(X.lds, w, lane, X.nvalid, acc);
  if (w == 0) {
    gf_put_published(X.mine, acc);
    gf_stores_acknowledged();
    if (lane == 0) *X.last_flag = (atomicAdd(W.tickets + X.tile, 1) + 1) % W.ngroups == 0;
  }
  __syncthreads();
  if (!*X.last_flag) return false;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other blocks' group summaries
  // ---- the tile's last block: compose its groups (a contiguous run per wave), finish, step
  const AdamRegs st = gf_state_request(X.F, w, X.d, X.kb, running);
  const int per = (W.ngroups + kGfWaves - 1) / kGfWaves;
  const int g0 = w * per, g1 = min(W.ngroups, g0 + per);
  const int nlive = (W.ngroups + per - 1) / per;
  const double* base = W.grp + (size_t)X.tile * W.ngroups * kGfFields * 64 + lane;
  if (g0 < g1) {
    acc = gf_take(base + (size_t)g0 * kGfFields * 64);
    for (int g = g0 + 1; g < g1; ++g) acc = nll_acc_combine(acc, gf_take(base + (size_t)g * kGfFields * 64));
  }
  gf_block_tree(X.lds, w, lane, nlive, acc);
  if (w == 0) {
    const DualD m = DualD(X.M.m0[(size_t)X.k * G.D + X.d] - acc.xr), P = DualD(X.M.S0[X.dd]);   // relative to the reference
    const DualD den = DualD(1.0) + acc.e.J * P;
    const DualD inv = rcp(den);
    const DualD ll = acc.ell - DualD(0.5) * log_with_rcp(den, inv) +
                     (acc.e.eta * m + DualD(0.5) * acc.e.eta * acc.e.eta * P - DualD(0.5) * acc.e.J * m * m) * inv;
    gf_finish(G, X.F, lane, X.k, X.d, X.kb, running, ll.v, ll.d, st, X.nll, X.dnll);
  }
  return true;
}

template <bool UNIT, int ROWS>
__global__ __launch_bounds__(64 * kGfWaves) void diag_nll_grad_fused_kernel(NllGeom G, DiagModel M, GradFuseWs W,
                                                                           const float* __restrict__ y,
                                                                           const double* __restrict__ rconst,
                                                                           const double* s_kp,
                                                                           double* __restrict__ nll,
                                                                           double* __restrict__ dnll, AdamFuse F) {
  __shared__ double lds[kGfWaves * kGfFields * 64];
  __shared__ int last_flag;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = blockIdx.x % G.ntile, grp = blockIdx.x / G.ntile;
  // (a launch counts into alternating counters and zeroes the next one)
  if (F.state != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *F.n_active_next = 0;
  const int n_raw = tile * 64 + lane;
  const bool chain_ok = n_raw < G.N;
  const int n = chain_ok ? n_raw : G.N - 1;          // lanes past the last chain shadow it (results unused)
  const int k = n / G.D, d = n - k * G.D;
  const size_t dd = (size_t)k * G.D * G.D + (size_t)d * (G.D + 1);
  const int j = grp * kGfWaves + w;
  const int t0 = j == 0 ? 0 : G.B0 + (j - 1) * G.BN;     // chunk 0 is the short one (its transient is the expensive part)
  const BufferRows ld{__builtin_amdgcn_make_buffer_rsrc(
                          const_cast<float*>(y + (size_t)(j < G.ncn ? t0 : 0) * G.N + tile * 64), 0, 0x7FFFFFFF,
                          0x00020000),
                      (unsigned)((n - tile * 64) * 4), (unsigned)(G.N * 4)};
  const int kb = F.state != nullptr ? F.kp_block[k] : 0;
  const GfCtx<BufferRows> X{G, M, W, F, ld, lds, &last_flag,
                            W.grp + ((size_t)tile * W.ngroups + grp) * kGfFields * 64 + lane, nll, dnll,
                            w, lane, tile, grp, j, min(kGfWaves, G.ncn - grp * kGfWaves), t0,
                            j < G.ncn ? min(j == 0 ? G.B0 : G.BN, G.T - t0) : 0, k, d, kb, dd,
                            rconst[n], M.A[dd], M.C[dd], M.Q[dd]};
  bool running = chain_ok;
  if (F.state != nullptr) running = chain_ok && adam_block_running(F.state, kb, F.cap);
  if (F.state != nullptr && !__any(running)) return;     // the same answer in every wave of the tile's blocks
  // the first rows of a converged-entry chunk are requested before the constants of the evaluation are formed
  float pre[ROWS];
  if (j >= 1 && X.len >= ROWS) {
#pragma unroll
    for (int q = 0; q < ROWS; ++q) pre[q] = ld(q);
  } else {
#pragma unroll
    for (int q = 0; q < ROWS; ++q) pre[q] = 0.f;
  }
  gf_evaluate<UNIT, ROWS>(X, s_kp[k], running, pre);
}

// N2'' chunk-parallel assembly for the grid search (value only, converged-entry summaries): a
// converged-entry summary has A = 0 - the mean entering the NEXT chunk is its b, whatever came
// before - so the walk over the chunks has no sequential dependency beyond chunk 0:
//     ll = ll_0(prior) + sum_{j >= 1} [ ell_j + eta_j mr_j - J_j mr_j^2 / 2 ],  mr_j = m_j - xref_j,
//     m_1 from the exact summary of chunk 0 applied to the prior, m_j = b_{j-1} for j >= 2.
// Block = (64-chain tile, candidate): wave w takes chunks w, w + 16, ... with lanes = chains
// (coalesced plane rows), the b's meet in LDS, the terms are summed through LDS in float64.  A
// chain with an exact-entry summary past chunk 0 (a wave that could not assume convergence)
// falls back to the sequential walk, done by its lane of wave 0.  One round of loads instead of
// ncn dependent ones: 17.6 -> ~6 us on the C3 shape.
constexpr int kAsmWaves = 16;
constexpr int kAsmPer = 8;                        // chunks per wave of the grid kernel's assembly (registers): ncn <= 128

__global__ __launch_bounds__(64 * kAsmWaves) void diag_nll_assemble_par_kernel(NllGeom G, DiagModel M,
                                                                              NllWs W,
                                                                              double* __restrict__ nll) {
  extern __shared__ double dyn[];                  // b_next[ncn][64] | part[kAsmWaves][64]
  __shared__ int seq[64];                          // chain needs the sequential walk
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x % G.ntile, ci = blockIdx.x / G.ntile;
  const int n = tile * 64 + lane;
  const bool live = n < G.N;
  double* bnext = dyn;                             // mean entering chunk j + 1 (absolute)
  double* part = dyn + (size_t)G.ncn * 64;
  if (w == 0) seq[lane] = 0;
  __syncthreads();
  const int k = live ? n / G.D : 0, d = live ? n - k * G.D : 0;
  const size_t dd = (size_t)k * G.D * G.D + (size_t)d * (G.D + 1);
  // pass 1: every summary's outgoing mean; chunk 0 is exact and is applied to the prior here
  double ll0 = 0.0;
  for (int j = w; j < G.ncn; j += kAsmWaves) {
    if (!live) break;
    const size_t o = ((size_t)j * W.ncp + ci) * G.N + n;
    const double A = W.A[o], b = W.b[o], C = W.C[o];
    if (j == 0) {
      const double eta = W.eta[o], J = W.J[o], ell = W.ell[o];
      const double xr = (double)W.xr[(size_t)j * G.N + n];
      const double m = M.m0[(size_t)k * G.D + d], P = M.S0[dd];
      const double mr = m - xr, den = 1.0 + J * P, inv = 1.0 / den;
      ll0 = ell - 0.5 * log(den) + (eta * mr + 0.5 * eta * eta * P - 0.5 * J * mr * mr) * inv;
      bnext[lane] = A * inv * (mr + P * eta) + b;
    } else {
      if (!(C < 0.0)) seq[lane] = 1;                // exact entry past chunk 0: not parallel
      bnext[(size_t)j * 64 + lane] = b;             // A = 0 for a converged-entry summary
    }
  }
  __syncthreads();
  // pass 2: the chunks' terms
  double acc = (w == 0) ? ll0 : 0.0;
  for (int j = (w == 0 ? kAsmWaves : w); j < G.ncn; j += kAsmWaves) {
    if (!live) break;
    const size_t o = ((size_t)j * W.ncp + ci) * G.N + n;
    const double eta = W.eta[o], J = W.J[o], ell = W.ell[o];
    const double mr = bnext[(size_t)(j - 1) * 64 + lane] - (double)W.xr[(size_t)j * G.N + n];
    acc += ell + eta * mr - 0.5 * J * mr * mr;
  }
  part[w * 64 + lane] = acc;
  __syncthreads();
  if (w != 0) return;
  double tot = 0.0;
  if (live) {
    if (seq[lane]) {                               // rare: the generic sequential walk
      auto get = [&](int j, Elem<double>& e, double& ell, double& xr) {
        const size_t o = ((size_t)j * W.ncp + ci) * G.N + n;
        xr = (double)W.xr[(size_t)j * G.N + n];
        e.A = W.A[o]; e.b = W.b[o]; e.C = W.C[o]; e.eta = W.eta[o]; e.J = W.J[o];
        ell = W.ell[o];
      };
      tot = nll_assemble<double>(G.ncn, M.m0[(size_t)k * G.D + d], M.S0[dd], get);
    } else {
#pragma unroll
      for (int q = 0; q < kAsmWaves; ++q) tot += part[q * 64 + lane];
    }
  }
  for (int off = 1; off < G.D; off <<= 1) tot += __shfl_xor(tot, off);   // the keypoint's D chains
  if (!live || d != 0) return;
  const double v = -tot;
  nll[(size_t)k * G.n_cand + ci] = isfinite(v) ? v : 1e12;                // eks/core.py:650
}

// ---- grid search, round 4: head + lean roles in ONE launch ------------------------------------------------------
// The 64-candidate grid is 2 FMAs per frame, chain and candidate; the general kernel above carries every regime's
// state through the frame loop (256 VGPRs at 8 candidates per lane: 8 waves per (tile, chunk), 18 % of its VALU
// instructions are not those FMAs and its chunk-0 blocks set the launch's length).  Here
//   * blocks [0, nhead): HEAD role - chunk 0 of every chain (known entry state, transient regimes) through the
//     general lane body at 4 candidates per lane: ntile x n_cand / 4 short waves that start first and run beside
//     the rest;
//   * the other blocks: LEAN role - block = (64-chain tile, chunk j >= 1), wave = 16 candidates: nll_lean_chunk,
//     the converged-entry summary alone (~180 VGPRs: two waves per SIMD, half the row loads and shared dy per
//     candidate-FMA, the candidates' constants parked in LDS).  A wave whose chunk does not qualify (wave-uniform:
//     the filter variance has not converged that early in the sequence) summarises it with the exact-entry code, 4
//     candidates at a time, and raises the (chunk, tile, group) flag; a pole so close to one that rho^t outlives
//     the chunk keeps the lean form with A = rho^len (flag 2).  Flagged (tile, candidate)s are assembled in order.
// Lean summaries are three planes (b, eta float32, ell float64: 16 B per chunk, chain and candidate instead of 28):
// A = 0, C = -1 are implied and J is a constant of the (chain, candidate), written once by the head wave.
constexpr int kLeanNC = 16;
constexpr int kHeadNCL = 4;
constexpr int kLeanWaves = 4;                // waves per block, both roles

struct LeanGeom {
  int nhead_blocks;      // head role: ceil(ntile * ngrp4 / kLeanWaves)
  int ngrp4, ngrp16;     // candidate groups of the two roles
  int32_t* flags;        // [ncn][ntile][ncp], per candidate: 0 = lean summary with A = 0, 1 = exact-entry summary (full
                         // planes valid), 2 = lean summary with its own A, J
  // ---- round 5: shared-lag form (eks_nll_lag.hpp).  A block whose chunk qualifies (whole 32-frame sets, converged
  // entry for EVERY candidate, at least kLagMinFast fast candidates - decided identically by its four waves from the
  // chains' constants, no exchange) deals only the slow candidates to its waves, accumulates the chunk's lag sums beside
  // them and forms the fast candidates' summaries from those at the end - into the same planes.
  int lag_on;            // the launch may use the form at all
  double rho_max;        // a candidate is fast when its pole is at most this for every chain of the tile
  // ---- table of the assembly (diag_nll_assemble_kp_kernel), written by the head role
  double* tab;           // [N][kTabFields][ncp]: J of the candidate's converged-entry summaries, chunk 0's term of the
                         // log-likelihood given the prior, the mean entering chunk 1
};
constexpr int kTabFields = 3;
enum { TAB_J32, TAB_LL0, TAB_B0 };
// Layouts of the grid path (round 5: the assembly's lanes are CANDIDATES, so everything per (chain, candidate) has the
// candidate fastest):  summary planes [j][N][ncp];  chunk references xr [j][N];  flags [j][ntile][ncp].
constexpr int kLagMinFast = 16;        // fewer fast candidates: the lag products cost more than they save
constexpr int kLagMaxNP = 6;           // slow pairs per wave (nslow <= 48 when at least 16 of 64 are fast)
// LDS of a block, in doubles: the round-4 lean role parks 4 floats x 16 candidates per lane and wave (64 KB); the lag
// form [4 waves][kLagN][64] float64 lag accumulators (32 KB) + 3 floats x 2 NP candidates per lane and wave (<= 36 KB)
constexpr int kGridLdsDoubles = (kLeanWaves * kLagN * 64) + (kLeanWaves * 3 * 2 * kLagMaxNP * 64) / 2;
static_assert(kGridLdsDoubles * 8 >= kLeanWaves * 4 * kLeanNC * 64 * 4, "the lean role's stash must fit");

// ---- the lag form of one (tile, chunk) block ---------------------------------------------------------------------------
// LDS of the block (kGridLdsDoubles doubles), in time order:
//   main loop   [4 waves][kLagN][64] float64 lag accumulators | the waves' stashes of constants
//   after A     the stash region holds the block's totals: lag sums [kLagN][64] float64, first / last inputs
//               [2 kLagN][64] float32 and the last observation [64] float32 (the lead wave's)
//   after C     the whole array is the output tile: b, eta [64 chains][kTilePitch] float32, ell [64][kTilePitch] float64
//               (candidate fastest, odd pitch: lanes = chains write it, lanes = candidates read it, both without
//               bank conflicts) - the block's 64 x 64 results leave as whole rows of the [j][N][ncp] planes.
//               With lanes = chains storing straight to those planes every store instruction touched 64 different
//               lines: 48 such instructions per wave at the END of every wave's chunk, all at once - measured as 15 us
//               on the C3 launch.
constexpr int kTilePitch = 65;
constexpr int kLagTotDoubles = kLagN * 64;                                   // lag totals
constexpr int kLagLdsLagAcc = kLeanWaves * kLagN * 64;                       // (doubles) the stash region starts here
static_assert((2 * 64 * kTilePitch * 4 + 64 * kTilePitch * 8 + 7) / 8 <= kGridLdsDoubles, "the output tile must fit");
static_assert(kLagTotDoubles + (2 * kLagN + 1) * 64 / 2 + 64 <= kGridLdsDoubles - kLagLdsLagAcc, "totals behind the accumulators");

struct LagDevSink {
  double* acc;             // LDS: this wave's [kLagN][64] accumulators, at the lane
  float uh[kLagN], ut[kLagN], yl;   // (the lead wave's: the chunk's first / last inputs, its last observation)
  __device__ __forceinline__ void add(int k, float v) const { acc[k * 64] += (double)v; }
  __device__ __forceinline__ void head(int i, float v) { uh[i] = v; }
  __device__ __forceinline__ void tail(int i, float v) { ut[i] = v; }
  __device__ __forceinline__ void ylast(float v) { yl = v; }
};

// NP slow pairs in this wave.  rk: this LANE's candidate's place in the list "slow candidates in index order, then the
// fast ones" (lanes >= n_cand: none); the pairs of consecutive places go to the waves round-robin: slot k of wave w
// is place 2 (4 (k / 2) + w) + k % 2.  turn_mask: this wave's lag sets of every 16.  Returns the lane
// body's verdict (1 / 2).  Every wave of the block runs this (the barriers inside are the block's).
template <int NP, bool UNIT>
__device__ __forceinline__ int lag_block_body(const NllGeom& G, const LeanGeom& LG, const NllWs& W, const BufferRows& ld,
                                              int j, int tile, int n, bool chain_ok, int w, int lane, int len, double q,
                                              double r_n, double a_n, double c_n, const double* sc, unsigned long long fm,
                                              int rk, unsigned turn_mask, double* lds) {
  constexpr int NC = 2 * NP;
  int cand[NC];
  bool used[NC];           // slot k holds a slow candidate (a fast one that pads the list is computed but not kept)
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const unsigned long long hit = __ballot(lane < G.n_cand && rk == 2 * ((k >> 1) * kLeanWaves + w) + (k & 1));
    const int c = hit ? __builtin_ctzll(hit) : G.n_cand - 1;
    cand[k] = __builtin_amdgcn_readfirstlane(c);
    used[k] = hit != 0 && ((fm >> c) & 1ull) == 0;
  }
  double* acc = lds + ((size_t)w * kLagN) * 64 + lane;
#pragma unroll
  for (int k = 0; k < kLagN; ++k) acc[k * 64] = 0.0;
  float* stash = reinterpret_cast<float*>(lds + kLagLdsLagAcc) + ((size_t)w * 3 * 2 * kLagMaxNP) * 64 + lane;   // (NP differs between waves)
  LagDevSink lsink;
  lsink.acc = acc;
  LeanOut<NC> out;
#pragma unroll
  for (int k = 0; k < NC; ++k) out.A[k] = out.J[k] = 0.f;
  auto sqf = [&](int k) { return sc[cand[k]] * q; };
  const int res = nll_lag_chunk<NP, kLagND, UNIT>(ld, len, r_n, a_n, c_n, sqf, turn_mask, 16, w == 0, stash, 64, out, lsink);
  if (chain_ok && w == 0) W.xr[(size_t)j * G.N + n] = out.xr;
  if (res == 2 && chain_ok) {                          // (wave-uniform, rare: the summaries carry their own A, J)
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      if (!used[k]) continue;
      const size_t o = ((size_t)j * G.N + n) * W.ncp + cand[k];
      W.A[o] = out.A[k];
      W.J[o] = out.J[k];
    }
  }
  // ---- A: every wave's partial lag sums are complete and its stash is dead
  __syncthreads();
  double* tot = lds + kLagLdsLagAcc;                                   // [kLagN][64]
  float* uu = reinterpret_cast<float*>(tot + kLagTotDoubles);          // [2 kLagN + 1][64]
#pragma unroll
  for (int i = 0; i < kLagN / kLeanWaves; ++i) {
    const int kk = w * (kLagN / kLeanWaves) + i;
    double v = 0.0;
#pragma unroll
    for (int ww = 0; ww < kLeanWaves; ++ww) v += lds[((size_t)ww * kLagN + kk) * 64 + lane];
    tot[kk * 64 + lane] = v;
  }
  if (w == 0) {
#pragma unroll
    for (int i = 0; i < kLagN; ++i) {
      uu[i * 64 + lane] = lsink.uh[i];
      uu[(kLagN + i) * 64 + lane] = lsink.ut[i];
    }
    uu[2 * kLagN * 64 + lane] = lsink.yl;
  }
  // ---- B: totals in LDS; every wave takes its own copy
  __syncthreads();
  double cs[kLagN];
  float uh[kLagN], ut[kLagN];
#pragma unroll
  for (int i = 0; i < kLagN; ++i) {
    cs[i] = tot[i * 64 + lane];
    uh[i] = uu[i * 64 + lane];
    ut[i] = uu[(kLagN + i) * 64 + lane];
  }
  const float yl = uu[2 * kLagN * 64 + lane];
  // ---- C: LDS is free - the output tile
  __syncthreads();
  float* tb = reinterpret_cast<float*>(lds);                           // b   [64][kTilePitch]
  float* te = tb + 64 * kTilePitch;                                    // eta [64][kTilePitch]
  double* tl = lds + (2 * 64 * kTilePitch * 4 + 7) / 8;                // ell [64][kTilePitch]
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    if (!used[k]) continue;
    tb[lane * kTilePitch + cand[k]] = out.B[k];
    te[lane * kTilePitch + cand[k]] = out.Eta[k];
    tl[lane * kTilePitch + cand[k]] = out.Ell[k];
  }
  // the fast candidates, dealt to the waves round-robin: each from the chunk's lag sums, in float64
  {
    const int nfast = __popcll(fm);
    const unsigned long long below = (1ull << lane) - 1ull;
    const bool isfast = ((fm >> lane) & 1ull) != 0;
    const int frank = __popcll(fm & below);
#pragma unroll 1
    for (int r = w; r < nfast; r += kLeanWaves) {
      const unsigned long long hit = __ballot(isfast && frank == r);
      const int c = __builtin_amdgcn_readfirstlane(__builtin_ctzll(hit));
      const LagConst kc = lag_const_fast<UNIT>(r_n, a_n, c_n, sc[c] * q);
      double b, eta, ell;
      lag_summary<kLagN, UNIT>(kc, a_n, c_n, len, cs, uh, ut, yl, b, eta, ell);
      tb[lane * kTilePitch + c] = (float)b;
      te[lane * kTilePitch + c] = (float)eta;
      tl[lane * kTilePitch + c] = ell;
    }
  }
  // ---- D: the tile is complete: whole rows out (wave w: chains 16 w .. 16 w + 15; lane = candidate)
  __syncthreads();
  if (lane < W.ncp) {
#pragma unroll 4
    for (int i = 0; i < 64 / kLeanWaves; ++i) {
      const int ch = w * (64 / kLeanWaves) + i;
      const int nn = tile * 64 + ch;
      if (nn >= G.N) break;                            // (wave-uniform)
      const size_t o = ((size_t)j * G.N + nn) * W.ncp + lane;
      W.b[o] = tb[ch * kTilePitch + lane];
      W.eta[o] = te[ch * kTilePitch + lane];
      W.ell[o] = tl[ch * kTilePitch + lane];
    }
  }
  return res;
}

// Diagnostic build only (-DEKS_GRID_STAMPS, tools/grid_stamps.py): lane 0 of every wave stamps the 100 MHz real-time
// counter when it starts and when it ends, with its role (0 head, 1 lean, 2 lag form, 3 exact-entry fallback).
#ifdef EKS_GRID_STAMPS
__device__ unsigned long long g_grid_stamps[2048][kLeanWaves][4];
#define GRID_STAMP_BEGIN()                                                                                   \
  const unsigned long long grid_t0_ = __builtin_amdgcn_s_memrealtime()
#define GRID_STAMP_END(role)                                                                                 \
  do {                                                                                                       \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) {                                                      \
      unsigned long long* st_ = g_grid_stamps[blockIdx.x][threadIdx.x >> 6];                                 \
      unsigned xcc_;                                                                                         \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                    \
      st_[0] = grid_t0_; st_[1] = __builtin_amdgcn_s_memrealtime(); st_[2] = (role); st_[3] = xcc_;          \
    }                                                                                                        \
  } while (0)
#else
#define GRID_STAMP_BEGIN() do { } while (0)
#define GRID_STAMP_END(role) do { } while (0)
#endif

template <bool UNIT>
__global__ __launch_bounds__(64 * kLeanWaves, 2) void diag_nll_grid_kernel(NllGeom G, LeanGeom LG, DiagModel M, NllWs W,
                                                                         const float* __restrict__ y,
                                                                         const double* __restrict__ rconst,
                                                                         const double* __restrict__ s_cand) {
  __shared__ double lds[kGridLdsDoubles];
  float (*stash)[4 * kLeanNC][64] = reinterpret_cast<float (*)[4 * kLeanNC][64]>(lds);
  GRID_STAMP_BEGIN();
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  auto store_full = [&](int j, int ci, int n, const NllElem<float>& o) {
    const size_t off = ((size_t)j * G.N + n) * W.ncp + ci;
    W.A[off] = o.e.A;
    W.b[off] = o.e.b;
    W.C[off] = o.e.C;
    W.eta[off] = o.e.eta;
    W.J[off] = o.e.J;
    W.ell[off] = o.ell;
  };
  if ((int)blockIdx.x < LG.nhead_blocks) {
    // ---- head: chunk 0, kHeadNCL candidates per lane
    const int hw = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kLeanWaves + w);
    const int tile = hw / LG.ngrp4, g = hw - tile * LG.ngrp4;
    if (tile >= G.ntile) return;
    const int n = tile * 64 + lane;
    if (n >= G.N) return;
    const int k = n / G.D, d = n - k * G.D;
    const size_t dd = (size_t)k * G.D * G.D + (size_t)d * (G.D + 1);
    const double q = M.Q[dd], r_n = rconst[n], a_n = M.A[dd], c_n = M.C[dd];
    double sq[kHeadNCL];
    // candidate c of head wave g is c ngrp4 + g: every wave holds the same mix of slow and fast candidates (with
    // contiguous groups the wave of the four slowest ran 130 us on C3, twice the others - longer than the lag-form
    // blocks beside it)
    auto head_cand = [&](int c) { return c * LG.ngrp4 + g; };
#pragma unroll
    for (int c = 0; c < kHeadNCL; ++c) {
      const int ci = min(head_cand(c), G.n_cand - 1);
      sq[c] = (G.per_keypoint ? s_cand[(size_t)k * G.n_cand + ci] : s_cand[ci]) * q;
    }
    const int len = min(G.B0, G.T);
    const BufferRows ld{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + tile * 64), 0, 0x7FFFFFFF, 0x00020000),
                        (unsigned)(lane * 4), (unsigned)(G.N * 4)};
    NllElem<float> out[kHeadNCL];
    nll_summarize_chunk<float, kHeadNCL, UNIT>(ld, 0, len, r_n, a_n, c_n, sq, out, false);
    if (g == 0) W.xr[n] = out[0].xref;
#pragma unroll
    for (int c = 0; c < kHeadNCL; ++c) {
      const int ci = head_cand(c);
      if (ci >= G.n_cand) continue;
      store_full(0, ci, n, out[c]);
      // the assembly's table: J of the converged-entry summaries (c cg / (1 - rho^2) as nll_lean_chunk forms it),
      // chunk 0 applied to the prior (its term and the mean it hands on)
      const LeanConst lc = lean_const<UNIT>(r_n, a_n, c_n, sq[c]);
      const float c_cg = UNIT ? lc.cg : (float)c_n * lc.cg;
      double* tb = LG.tab + (size_t)n * kTabFields * W.ncp + ci;
      tb[TAB_J32 * W.ncp] = (double)(c_cg / (1.f - lc.rho * lc.rho));
      const double eA = out[c].e.A, eb = out[c].e.b, eeta = out[c].e.eta, eJ = out[c].e.J;
      const double m = M.m0[(size_t)k * G.D + d], P = M.S0[dd];
      const double mr = m - (double)out[c].xref, den = 1.0 + eJ * P, inv = 1.0 / den;
      tb[TAB_LL0 * W.ncp] = out[c].ell - 0.5 * log(den) + (eeta * mr + 0.5 * eeta * eeta * P - 0.5 * eJ * mr * mr) * inv;
      tb[TAB_B0 * W.ncp] = eA * inv * (mr + P * eeta) + eb;
    }
    GRID_STAMP_END(0);
    return;
  }
  // ---- lean: (tile, chunk j >= 1), wave = 16 candidates
  const int lb = (int)blockIdx.x - LG.nhead_blocks;
  const int tile = lb % G.ntile, j = 1 + lb / G.ntile;
  if (j >= G.ncn) return;
  const int n_raw = tile * 64 + lane;
  const bool chain_ok = n_raw < G.N;
  const int n = chain_ok ? n_raw : G.N - 1;              // lanes past the last chain shadow it (nothing stored)
  const int k = n / G.D, d = n - k * G.D;
  const size_t dd = (size_t)k * G.D * G.D + (size_t)d * (G.D + 1);
  const double q = M.Q[dd], r_n = rconst[n], a_n = M.A[dd], c_n = M.C[dd];
  const int t0 = G.B0 + (j - 1) * G.BN, len = min(G.BN, G.T - t0);
  const BufferRows ld{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + (size_t)t0 * G.N + tile * 64), 0,
                                                        0x7FFFFFFF, 0x00020000),
                      (unsigned)((n - tile * 64) * 4), (unsigned)(G.N * 4)};
  const double* sc = G.per_keypoint ? s_cand + (size_t)k * G.n_cand : s_cand;
  // ---- round 5: the shared-lag form, when the block's chunk qualifies.  Every wave decides from the same numbers.
  if (LG.lag_on && (len & 31) == 0 && len >= 64) {
    const double a1 = UNIT ? 1.0 : a_n, c1 = UNIT ? 1.0 : c_n;
    const double thr_fast = lag_sq_threshold(r_n, a1, c1, LG.rho_max);
    // converged entry for every candidate: rho^(2 t0) < 1e-20 <=> |rho| < exp(-23 / t0) <=> s q above its threshold
    const double thr_qual = lag_sq_threshold(r_n, a1, c1, (double)__expf(-23.f / (float)t0));
    unsigned long long fm = 0;
    bool qual = true;
    for (int c = 0; c < G.n_cand; ++c) {
      const double sqc = sc[c] * q;
      if (__all(sqc >= thr_fast)) fm |= 1ull << c;
      qual = qual && __all(sqc >= thr_qual);
    }
    const int nfast = __popcll(fm), nslow = G.n_cand - nfast;
    // (general diagonal models form every input u in float64: with 5 or 6 pairs per wave their frame loops spill)
    if (qual && nfast >= kLagMinFast && nslow <= 8 * (UNIT ? kLagMaxNP : 4)) {
      const unsigned long long below = (1ull << lane) - 1ull;
      const int rk = ((fm >> lane) & 1ull) ? nslow + __popcll(fm & below) : __popcll(~fm & below);
      // The slow candidates go to the waves in PAIRS, round-robin, slowest first: P pairs -> P / 4 per wave and one more
      // for the first P % 4 waves (padding every wave to the same count cost 32 recursions for C3's 28 slow candidates,
      // and the kernel is bound by the FMAs it issues).  The waves with fewer pairs take more of the lag sets: with a
      // lag set at ~4.25 pair-sets of FMAs, `rem` turns of 16 for the waves with the extra pair and rem + 4 for the
      // others level the work (rem = 0: four each).
      int npairs = (nslow + 1) / 2;
      if (npairs < kLeanWaves) npairs = kLeanWaves;
      const int base = npairs / kLeanWaves, rem = npairs % kLeanWaves;
      const int np = base + (w < rem ? 1 : 0);
      const int turns = rem == 0 ? 4 : (w < rem ? rem : rem + 4);
      const int first = rem == 0 ? 4 * w : (w < rem ? rem * w : rem * rem + (rem + 4) * (w - rem));
      const unsigned turn_mask = ((1u << turns) - 1u) << first;
      int res = 1;
#define EKS_LAG_BODY(NP_) \
  res = lag_block_body<NP_, UNIT>(G, LG, W, ld, j, tile, n, chain_ok, w, lane, len, q, r_n, a_n, c_n, sc, fm, rk, turn_mask, lds)
      switch (np) {
        case 1: EKS_LAG_BODY(1); break;
        case 2: EKS_LAG_BODY(2); break;
        case 3: EKS_LAG_BODY(3); break;
        case 4: EKS_LAG_BODY(4); break;
        case 5: if constexpr (UNIT) EKS_LAG_BODY(5); break;
        default: if constexpr (UNIT) EKS_LAG_BODY(6); break;
      }
#undef EKS_LAG_BODY
      // flags of this wave's slow candidates (the pair of place rk / 2 is wave (rk / 2) % 4's); the fast ones never flag
      if (lane < G.n_cand && !((fm >> lane) & 1ull) && ((rk >> 1) & 3) == w)
        LG.flags[((size_t)j * G.ntile + tile) * W.ncp + lane] = res == 2 ? 2 : 0;
      if (lane < G.n_cand && ((fm >> lane) & 1ull) && (lane & 3) == w)
        LG.flags[((size_t)j * G.ntile + tile) * W.ncp + lane] = 0;
      GRID_STAMP_END(2);
      return;
    }
  }
  // ---- the round-4 form: every candidate by the recursion, 16 per wave
  // the grid's candidates are dealt to the tile's waves round-robin: slot c of wave w is candidate c ngrp16 + w, so
  // every wave holds the same mix of slow and fast candidates, slowest first (the staged alive phase of
  // nll_lean_chunk then costs every wave the same, a few per cent; contiguous groups left the slowest group's
  // waves 25 % longer than the rest of a launch whose blocks all run in one round)
  __shared__ int tile_valid[kLeanWaves];
  const int ncand = G.n_cand, stride16 = LG.ngrp16;
  auto cand_of = [&](int c) { return c * stride16 + w; };
  auto sqf = [&](int c) { return sc[min(cand_of(c), ncand - 1)] * q; };
  // slot c is candidate c stride16 + w: valid while c stride16 + w < n_cand
  const int nvalid = min(kLeanNC, ncand > w ? (ncand - w + stride16 - 1) / stride16 : 0);
  // the summaries stay in registers from the end of the lane body to the block's output tile (as the lag form: whole
  // rows of the [j][N][ncp] planes instead of 48 scattered store instructions per wave); a summary with A != 0 stores
  // its A, J at once (rare)
  struct Keep {
    const NllWs& W;
    size_t base, cstride, xr_off;
    int nvalid;
    bool store, store_xr;
    float B[kLeanNC], Eta[kLeanNC];
    double Ell[kLeanNC];
    __device__ __forceinline__ void xref(float v) const { if (store_xr) W.xr[xr_off] = v; }
    __device__ __forceinline__ void eta(int k, float v) { Eta[k] = v; }
    __device__ __forceinline__ void aj(int k, float a, float jv) const {
      if (store && k < nvalid) {
        W.A[base + k * cstride] = a;
        W.J[base + k * cstride] = jv;
      }
    }
    __device__ __forceinline__ void b(int k, float v) { B[k] = v; }
    __device__ __forceinline__ void ell(int k, double v) { Ell[k] = v; }
  };
  Keep keep{W, ((size_t)j * G.N + n) * W.ncp + w, (size_t)stride16, (size_t)j * G.N + n, nvalid, chain_ok, chain_ok && w == 0,
            {}, {}, {}};
  int lean = 0;
  if (w < LG.ngrp16) {
    lean = nll_lean_chunk<kLeanNC, UNIT>(ld, t0, len, r_n, a_n, c_n, sqf, &stash[w][0][lane], 64, keep);
    // flag: 0 lean summary with A = 0 (the usual case) | 2 lean summary with A = rho^len != 0 (own A, J planes) |
    // 1 exact-entry summary (full planes) - anything but 0 sends the (tile, candidate)'s assembly down the sequential walk
    if (lane < kLeanNC && cand_of(lane) < ncand)
      LG.flags[((size_t)j * G.ntile + tile) * W.ncp + cand_of(lane)] = lean == 1 ? 0 : (lean == 2 ? 2 : 1);
    if (!lean) {
      // ---- the chunk does not qualify for the converged-entry summary: exact entry, kHeadNCL candidates at a time,
      // stored field by field (these waves put nothing into the tile)
      for (int h = 0; h < kLeanNC / kHeadNCL; ++h) {
        double sq[kHeadNCL];
#pragma unroll
        for (int c = 0; c < kHeadNCL; ++c) sq[c] = sqf(h * kHeadNCL + c);
        NllElem<float> o4[kHeadNCL];
        nll_summarize_chunk<float, kHeadNCL, UNIT>(ld, t0, len, r_n, a_n, c_n, sq, o4, false);
        if (!chain_ok) continue;
        if (w == 0 && h == 0) W.xr[(size_t)j * G.N + n] = o4[0].xref;
#pragma unroll
        for (int c = 0; c < kHeadNCL; ++c) {
          const int ci = cand_of(h * kHeadNCL + c);
          if (ci < G.n_cand) store_full(j, ci, n, o4[c]);
        }
      }
    }
  }
  // ---- the block's output tile (every wave of the block arrives here)
  __syncthreads();                                                     // the stashes are dead
  {
    float* tb = reinterpret_cast<float*>(lds);                         // b   [64][kTilePitch]
    float* te = tb + 64 * kTilePitch;                                  // eta [64][kTilePitch]
    double* tl = lds + (2 * 64 * kTilePitch * 4 + 7) / 8;              // ell [64][kTilePitch]
    if (lean) {
#pragma unroll
      for (int c = 0; c < kLeanNC; ++c) {
        if (c >= nvalid) continue;
        tb[lane * kTilePitch + cand_of(c)] = keep.B[c];
        te[lane * kTilePitch + cand_of(c)] = keep.Eta[c];
        tl[lane * kTilePitch + cand_of(c)] = keep.Ell[c];
      }
    }
    if (lane == 0) tile_valid[w] = lean != 0;
    __syncthreads();
    // whole rows out (wave w: chains 16 w .. 16 w + 15; lane = candidate, whose wave is candidate % ngrp16)
    if (lane < ncand && tile_valid[lane % stride16]) {
#pragma unroll 4
      for (int i = 0; i < 64 / kLeanWaves; ++i) {
        const int ch = w * (64 / kLeanWaves) + i;
        const int nn = tile * 64 + ch;
        if (nn >= G.N) break;                                          // (wave-uniform)
        const size_t o = ((size_t)j * G.N + nn) * W.ncp + lane;
        W.b[o] = tb[ch * kTilePitch + lane];
        W.eta[o] = te[ch * kTilePitch + lane];
        W.ell[o] = tl[ch * kTilePitch + lane];
      }
    }
  }
  GRID_STAMP_END(lean ? 1 : 3);
}

// ---- assembly of the grid kernel's summaries, argmin included (round 5) ---------------------------------------------
// A converged-entry summary has A = 0, so chunk j's term of the log-likelihood needs only the mean chunk j - 1 hands
// on:  ll = ll_0(prior) + sum_{j >= 1} [ ell_j + eta_j mr_j - J mr_j^2 / 2 ],  mr_j = b_{j-1} - xref_j.
// Block = KEYPOINT, wave = (chain d of the keypoint, group of consecutive chunks), LANE = CANDIDATE:
//   * a chunk's summaries of one chain are one coalesced row of the [j][N][ncp] planes (however they were formed:
//     recursion or lag sums), all rows of a batch requested before any is used;
//   * the wave walks its chunks in order, so the mean a chunk hands on stays in a register;
//   * the waves' sums meet in LDS: wave 0 adds them in a fixed order, writes nll[k][0 .. n_cand) as one row and takes
//     the argmin across its lanes (first minimum, numpy.argmin semantics) - no separate argmin launch, no exchange
//     between blocks;
//   * a candidate with a flagged summary anywhere in the sequence (exact-entry summary, or a pole whose rho^t outlives
//     its chunk) is walked in order, from the prior, by the first wave of each chain (nll_assemble).
constexpr int kAsmKpWaves = 16;
constexpr int kAsmBatch = 8;            // chunks whose rows a wave requests at a time

struct GridAsmOut {
  const double* s_cand;
  double* s_out;         // [K] s_cand at the argmin (may be null: no argmin)
  int32_t* idx_out;      // [K] (may be null)
};

template <bool UNIT>
__global__ __launch_bounds__(64 * kAsmKpWaves) void diag_nll_assemble_kp_kernel(NllGeom G, LeanGeom LG, DiagModel M, NllWs W,
                                                                               GridAsmOut O, int ncgw, int cpw,
                                                                               double* __restrict__ nll) {
  __shared__ double accs[kAsmKpWaves][64];
  __shared__ int flg[kAsmKpWaves][64];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int k = blockIdx.x;
  const int d = wv / ncgw, cg = wv - d * ncgw;                    // (blockDim = D * ncgw waves)
  const int n = k * G.D + d, tile = n >> 6;
  const size_t N = (size_t)G.N, ncp = (size_t)W.ncp;
  const size_t dd = (size_t)k * G.D * G.D + (size_t)d * (G.D + 1);
  const bool cvalid = lane < G.n_cand;
  const double* tb = LG.tab + (size_t)n * kTabFields * ncp + lane;
  const double J32 = tb[TAB_J32 * ncp];
  const int j0 = 1 + cg * cpw, j1 = min(G.ncn, j0 + cpw);
  double acc = 0.0;
  int flagor = 0;
  if (j0 < j1) {
    // the mean entering the wave's first chunk
    double m_in;
    if (j0 == 1) {
      m_in = tb[TAB_B0 * ncp];
      acc = tb[TAB_LL0 * ncp];
    } else {
      m_in = (double)W.b[((size_t)(j0 - 1) * N + n) * ncp + lane];
    }
    for (int jb = j0; jb < j1; jb += kAsmBatch) {
      // every row of the batch is requested before anything is evaluated
      float b_r[kAsmBatch], eta_r[kAsmBatch], xr_r[kAsmBatch];
      double ell_r[kAsmBatch];
      int fl_r[kAsmBatch];
#pragma unroll
      for (int q = 0; q < kAsmBatch; ++q) {
        const int j = jb + q < j1 ? jb + q : j1 - 1;              // (past the end: the last chunk again, unused)
        const size_t row = ((size_t)j * N + n) * ncp + lane;
        xr_r[q] = W.xr[(size_t)j * N + n];
        fl_r[q] = LG.flags[((size_t)j * G.ntile + tile) * ncp + lane];
        b_r[q] = W.b[row];
        eta_r[q] = W.eta[row];
        ell_r[q] = W.ell[row];
      }
#pragma unroll
      for (int q = 0; q < kAsmBatch; ++q) {
        if (jb + q >= j1) continue;                                // (wave-uniform)
        flagor |= fl_r[q];
        const double mr = m_in - (double)xr_r[q];
        acc += ell_r[q] + (double)eta_r[q] * mr - 0.5 * J32 * mr * mr;
        m_in = (double)b_r[q];
      }
    }
  }
  accs[wv][lane] = acc;
  flg[wv][lane] = cvalid ? flagor : 0;
  __syncthreads();
  // ---- flagged candidates: the first wave of each chain walks them in order, from the prior
  if (cg == 0) {
    int any = 0;
    for (int q = 0; q < ncgw; ++q) any |= flg[d * ncgw + q][lane];
    if (__any(any != 0)) {
      double tot = 0.0;
      if (any != 0) {
        auto get = [&](int j, Elem<double>& e, double& ell, double& xr) {
          const size_t o = ((size_t)j * N + n) * ncp + lane;
          xr = (double)W.xr[(size_t)j * N + n];
          e.b = W.b[o]; e.eta = W.eta[o]; ell = W.ell[o];
          const int fl = j == 0 ? 1 : LG.flags[((size_t)j * G.ntile + tile) * ncp + lane];
          if (fl == 1) {
            e.A = W.A[o]; e.C = W.C[o]; e.J = W.J[o];
          } else if (fl == 2) {
            e.A = W.A[o]; e.C = -1.0; e.J = W.J[o];
          } else {
            e.A = 0.0; e.C = -1.0; e.J = J32;
          }
        };
        tot = nll_assemble<double>(G.ncn, M.m0[(size_t)k * G.D + d], M.S0[dd], get);
      }
      // the walk's total replaces the chain's chunk-parallel sums for those candidates
      if (any != 0) {
        accs[wv][lane] = tot;
        for (int q = 1; q < ncgw; ++q) accs[d * ncgw + q][lane] = 0.0;
      }
    }
  }
  __syncthreads();
  if (wv != 0) return;
  double tot = 0.0;
  const int nw = G.D * ncgw;
  for (int q = 0; q < nw; ++q) tot += accs[q][lane];              // chains in order, each chain's chunk groups in order
  double v = -tot;
  v = isfinite(v) ? v : 1e12;                                     // eks/core.py:650
  if (cvalid) nll[(size_t)k * G.n_cand + lane] = v;
  if (O.s_out == nullptr) return;
  constexpr int kNone = 0x7FFFFFFF;
  int best = cvalid ? lane : kNone;
  double bv = v;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(best, off);
    if (oi != kNone && (best == kNone || ov < bv || (ov == bv && oi < best))) {
      bv = ov;
      best = oi;
    }
  }
  if (lane == 0) {
    O.s_out[k] = O.s_cand[best];
    if (O.idx_out) O.idx_out[k] = best;
  }
}

// may the gradient evaluation take the single-launch kernel (diag_nll_grad_fused_kernel)?  One value of s
// per keypoint, whole 64-chain tiles addressed through 32-bit buffer offsets, the chains of a keypoint in
// adjacent lanes of one wave.
static int grad_fused_chunk(int T, int N);
static bool grad_fused_ok(int T, int N, int D, int n_cand, int per_keypoint) {
  return n_cand == 1 && per_keypoint && N > 32 && (D & (D - 1)) == 0 && D <= 64 &&
         (long)grad_fused_chunk(T, N) * N * 4 < (1L << 31) && !knob_int(KNOB_NLL_GRAD_UNFUSED, 0);
}
// frames per chunk of that kernel: blocks = tiles x groups of kGfWaves chunks, a whole number of 256-CU
// rounds when the problem is large enough (C3: 8 tiles x 32 groups x 8 chunks of 392 frames)
static int grad_fused_chunk(int T, int N) {
  const long ntile = (N + 63) / 64;
  long rounds = ((long)T * ntile + 128L * kGfWaves * kNllChunkGrad) / (256L * kGfWaves * kNllChunkGrad);
  if (rounds < 1) rounds = 1;
  long groups = (256 * rounds + ntile - 1) / ntile;
  if (groups < 1) groups = 1;
  long bn = (T + groups * kGfWaves - 1) / (groups * kGfWaves);
  bn = (bn + 7) / 8 * 8;
  if (bn < kGfChunkMin) bn = kGfChunkMin;
  const int b = knob_int(KNOB_NLL_GRAD_CHUNK, (int)bn);
  return b < 8 ? 8 : b;
}

// may the grid search take diag_nll_grid_kernel?  Whole 64-chain tiles, the chains of a keypoint in adjacent lanes
// of one wave, at least one lean wave's worth of candidates, a sequence long enough for chunks past the first, and
// the assembly's chunk table within LDS.
constexpr int kLeanChunk = 1600;       // target frames per lean chunk (C3: 62 chunks + chunk 0: one round of 512 blocks)
constexpr int kLeanChunkMin = 1024;    // (shorter chunks: rho^t of the slow candidates outlives them - A != 0 summaries)
// chunk geometry of the grid kernel: chunk 0 of b0 frames, then chunks of bn; blocks = head + tiles x chunks fill a
// whole number of rounds of 2 blocks per CU.  Returns ncn (0: the sequence is too short for chunks past the first).
static int lean_geometry(int T, int N, int n_cand, int* b0_out, int* bn_out) {
  const int b0 = knob_int(KNOB_NLL_CHUNK0, kNllChunk0);
  if (T < b0 + kLeanChunkMin) return 0;
  const long ntile = (N + 63) / 64;
  const long nhead = (ntile * ((n_cand + kHeadNCL - 1) / kHeadNCL) + kLeanWaves - 1) / kLeanWaves;
  const long rest = T - b0;
  const int target = knob_int(KNOB_NLL_CHUNK, kLeanChunk);
  long rounds = (rest * ntile + 256L * target) / (512L * target);
  if (rounds < 1) rounds = 1;
  long nch = (512 * rounds - nhead) / ntile;
  if (nch < 1) nch = 1;
  long bn = (rest + nch - 1) / nch;
  bn = (bn + 31) / 32 * 32;            // whole 32-frame sets (the lag form's unit)
  if (bn < kLeanChunkMin) bn = kLeanChunkMin;
  if (knob_set(KNOB_NLL_CHUNK)) bn = target < 64 ? 64 : (target + 15) / 16 * 16;      // (A/B runs and tests)
  *b0_out = b0;
  *bn_out = (int)bn;
  return 1 + (int)((rest + bn - 1) / bn);
}
// may the grid search take diag_nll_grid_kernel?  Whole 64-chain tiles, the chains of a keypoint in adjacent lanes
// of one wave, at least one lean wave's worth of candidates, a sequence long enough for chunks past the first, a
// chunk within 32-bit buffer offsets, and the assembly's chunk table within LDS.
static bool lean_grid_ok(int T, int N, int D, int n_cand) {
  if (knob_int(KNOB_NLL_LEGACY, 0)) return false;
  // (a lean block is kLeanWaves waves of kLeanNC candidates: up to 64 candidates; longer grids keep the general kernel)
  if (n_cand < kLeanNC || n_cand > kLeanNC * kLeanWaves || N <= 32 || (D & (D - 1)) != 0 || D > kAsmKpWaves) return false;
  int b0, bn;
  const int ncn = lean_geometry(T, N, n_cand, &b0, &bn);
  if (ncn < 2 || (long)(bn > b0 ? bn : b0) * N * 4 >= (1L << 31)) return false;
  return true;
}

static NllGeom make_geom(int T, int N, int D, int n_cand, int per_keypoint, bool grad, int ncl) {
  NllGeom G;
  G.N = N;
  G.T = T;
  G.D = D;
  G.BN = grad ? kNllChunkGrad : kNllChunk;
  G.B0 = G.BN;
  if (grad && grad_fused_ok(T, N, D, n_cand, per_keypoint)) {
    G.BN = grad_fused_chunk(T, N);
    // Chunk 0 starts from a known state and pays the transient regimes (the dual-number recursion in full until the
    // variance has converged): at the others' length its wave was the one every iteration waited for (C3, mid-search:
    // 38 us against 24; with 128 frames the tiles whose poles are above 0.84 mid-search lose the converged-entry form,
    // 128 / 200 / 256 / 392 frames: 5.06 / 4.98 / 4.93 / 5.08 ms for the whole search).  Round 5: two thirds of BN.
    G.B0 = G.BN >= 192 ? (G.BN * 2 / 3) / 8 * 8 : G.BN;
    if (knob_set(KNOB_NLL_CHUNK0)) {
      const int b0 = knob_int(KNOB_NLL_CHUNK0, 128) / 8 * 8;
      G.B0 = b0 < 8 ? 8 : (b0 > G.BN ? G.BN : b0);
    }
  }
  if (!grad) {
    // one block per (64-chain tile, chunk): pick the chunk length so that the grid is a whole
    // number of 256-CU rounds (C3: 8 tiles x 32 chunks = 256 blocks)
    const int ntile64 = (N + 63) / 64;
    long rounds = ((long)T * ntile64 + 128L * kNllChunk) / (256L * kNllChunk);
    if (rounds < 1) rounds = 1;
    long chunks = (256 * rounds + ntile64 - 1) / ntile64;
    if (chunks < 1) chunks = 1;
    int bn = (int)((T + chunks - 1) / chunks);
    bn = (bn + 64 - 1) / 64 * 64;
    if (bn < kNllChunkMin) bn = kNllChunkMin;   // short chunks cost accuracy (one float32 element
                                                // per chunk) and transient work
    G.BN = knob_int(KNOB_NLL_CHUNK, bn);
    if (G.BN < kNllChunkGrad) G.BN = kNllChunkGrad;
    G.B0 = G.BN;
    // Chunk 0 is the only one that always pays the start-up transient (its element starts from a
    // known state; later chunks enter with the converged variance, nll_summarize_chunk), so it is
    // kept shorter than the others - its block must not be the one every other block waits for -
    // and the other chunks share the remaining frames so that the block count is unchanged.
    const long nch = (T + G.BN - 1) / G.BN;
    const int b0 = knob_int(KNOB_NLL_CHUNK0, kNllChunk0);
    if (nch >= 3 && b0 < G.BN && !knob_set(KNOB_NLL_CHUNK)) {
      G.B0 = b0;
      G.BN = (int)((T - b0 + nch - 2) / (nch - 1));
      G.BN = (G.BN + 63) / 64 * 64;
    }
  }
  G.ncn = T <= G.B0 ? 1 : 1 + (T - G.B0 + G.BN - 1) / G.BN;
  int nt_log2 = 0;
  while ((1 << nt_log2) < N && nt_log2 < 6) ++nt_log2;
  G.nt_log2 = nt_log2;
  G.ntile = (N + (1 << nt_log2) - 1) >> nt_log2;
  G.n_cand = n_cand;
  G.ngrp = (n_cand + ncl - 1) / ncl;
  G.per_keypoint = per_keypoint;
  G.converged_entry = 0;
  return G;
}

// Tuning knobs (read once): EKS_NLL_NCL in {1,2,4,8} candidates per lane, EKS_NLL_CHUNK >= 512
// frames per lane.  Defaults are what bench.py measured best on MI355X.
static inline int pick_ncl(int n_cand, bool grad) {
  if (grad || n_cand < 2) return 1;
  int ncl = knob_int(KNOB_NLL_NCL, kNclGrid);
  if (ncl != 1 && ncl != 2 && ncl != 4 && ncl != 8) ncl = kNclGrid;
  while (ncl > 1 && n_cand < ncl) ncl >>= 1;
  return ncl;
}

// few (keypoint, candidate) pairs and many chunks: the chunk summaries are composed by a tree
static bool nll_uses_tree(int K, int D, int n_cand, int ncn) {
  return (long)K * n_cand * D <= 8192 && ncn >= 8 && D <= 16;
}

// does the gradient evaluation of a (T, K, D) problem (one value of s per keypoint) end in the tree
// assembly - the kernel that can apply the optimiser step itself (AdamFuse::step_in_kernel)?
bool diag_nll_grad_tree(int T, int K, int D) {
  const NllGeom G = make_geom(T, K * D, D, 1, 1, true, 1);
  return grad_fused_ok(T, K * D, D, 1, 1) || nll_uses_tree(K, D, 1, G.ncn);
}

size_t diag_nll_workspace_bytes(int T, int N, int n_cand) {
  // sized for the larger of the two modes (grad planes + smaller chunks)
  const int ncn = (T + kNllChunkGrad - 1) / kNllChunkGrad + 1;
  const size_t ncp = align_up((size_t)n_cand, 16);      // (kLeanNC: the grid kernel pads to whole lean waves)
  const size_t fl = align_up((size_t)ncn * ncp * N * sizeof(float), 256);
  const size_t db = align_up((size_t)ncn * ncp * N * sizeof(double), 256);
  // 10 element planes + the chunk references (one candidate's worth is used) + the Adam loop's
  // keypoint -> block map and its second counter (eks_adam_run); the search from cached lag sums (eks_lag_adam.hip)
  // keeps its partial sums and chain-major copies in the same bytes
  size_t main = 11 * fl + 2 * db;
  if (n_cand == 1) {
    const size_t lag = diag_lag_adam_workspace_bytes(T, N);
    if (lag > main) main = lag;
  }
  return main + adam_extra_bytes(N);
}

// [keypoint -> block map : N ints][tile tickets : ceil(N / 64) ints][second counter : 256 B]
size_t adam_extra_bytes(int N) {
  return align_up((size_t)N * sizeof(int32_t), 256) + align_up((size_t)((N + 63) / 64) * sizeof(int32_t), 256) + 256;
}
int32_t* nll_ws_tickets(void* ws, int T, int N, int n_cand) {
  char* tail = static_cast<char*>(ws) + diag_nll_workspace_bytes(T, N, n_cand) - adam_extra_bytes(N);
  return reinterpret_cast<int32_t*>(tail + align_up((size_t)N * sizeof(int32_t), 256));
}

// the single-launch gradient evaluation (and, with F.step_in_kernel, the optimiser step of its keypoints)
static int grad_fused_launch(const eks_dims_t& d, const NllGeom& G, const float* y, const double* rconst, const DiagModel& M,
                             const double* s_kp, double* nll, double* dnll, void* ws, hipStream_t st, const AdamFuse& F,
                             bool tickets_zeroed) {
  const int T = d.n_frames, N = d.n_keypoints * d.state_dim;
  GradFuseWs FW;
  FW.ngroups = (G.ncn + kGfWaves - 1) / kGfWaves;
  FW.grp = static_cast<double*>(ws);
  FW.tickets = nll_ws_tickets(ws, T, N, 1);
  FW.conv_allowed = !knob_int(KNOB_NLL_GRAD_TREE, 0);
  const size_t grp_bytes = (size_t)G.ntile * FW.ngroups * kGfFields * 64 * sizeof(double);
  if (grp_bytes > diag_nll_workspace_bytes(T, N, 1) - adam_extra_bytes(N)) return EKS_ERR_WORKSPACE;
  const dim3 grid((unsigned)(G.ntile * FW.ngroups)), block(64 * kGfWaves);
  const bool unit = (d.flags & EKS_FLAG_UNIT_AC) != 0, few = (int)grid.x <= kGfFewBlocks;
  if (!tickets_zeroed) {     // (eks_adam_run zeroes the tickets once; every evaluation leaves them zero)
    const hipError_t e = hipMemsetAsync(FW.tickets, 0, (size_t)G.ntile * sizeof(int32_t), st);
    if (e != hipSuccess) return hip_status(e);
  }
  ProfScope ps("diag_nll_grad_fused", st);
#define EKS_GF_LAUNCH(UN, RW) \
  hipLaunchKernelGGL((diag_nll_grad_fused_kernel<UN, RW>), grid, block, 0, st, G, M, FW, y, rconst, s_kp, nll, dnll, F)
  if (few) {
    if (unit) EKS_GF_LAUNCH(true, kGfRowsFew); else EKS_GF_LAUNCH(false, kGfRowsFew);
  } else {
    if (unit) EKS_GF_LAUNCH(true, kGfRowsFull); else EKS_GF_LAUNCH(false, kGfRowsFull);
  }
#undef EKS_GF_LAUNCH
  return hip_status(hipGetLastError());
}

static int diag_nll_impl(const eks_dims_t& d, const float* y, const double* rconst, const DiagModel& M,
                         const double* s_cand, int n_cand, int per_keypoint, double* nll, double* dnll,
                         void* ws, size_t ws_bytes, hipStream_t st, const AdamFuse* fuse, double* s_out, int32_t* idx_out,
                         bool* argmin_done);

int diag_nll(const eks_dims_t& d, const float* y, const double* rconst, const DiagModel& M,
             const double* s_cand, int n_cand, int per_keypoint, double* nll, double* dnll,
             void* ws, size_t ws_bytes, hipStream_t st, const AdamFuse* fuse, double* s_out, int32_t* idx_out) {
  bool done = false;
  const int rc = diag_nll_impl(d, y, rconst, M, s_cand, n_cand, per_keypoint, nll, dnll, ws, ws_bytes, st, fuse, s_out,
                               idx_out, &done);
  if (rc != EKS_OK || !s_out || done) return rc;
  if (per_keypoint) return EKS_ERR_UNSUPPORTED;      // (the separate argmin gathers from one shared grid)
  return argmin_s(d.n_keypoints, n_cand, nll, s_cand, s_out, idx_out, st);
}

static int diag_nll_impl(const eks_dims_t& d, const float* y, const double* rconst, const DiagModel& M,
                         const double* s_cand, int n_cand, int per_keypoint, double* nll, double* dnll,
                         void* ws, size_t ws_bytes, hipStream_t st, const AdamFuse* fuse, double* s_out, int32_t* idx_out,
                         bool* argmin_done) {
  const int T = d.n_frames, D = d.state_dim, K = d.n_keypoints, N = K * D;
  if (ws_bytes < diag_nll_workspace_bytes(T, N, n_cand)) return EKS_ERR_WORKSPACE;
  const bool grad = dnll != nullptr;
  AdamFuse F{};                                   // all null: no gating, no fused step
  if (fuse) F = *fuse;
  const int ncl = pick_ncl(n_cand, grad);
  NllGeom G = make_geom(T, N, D, n_cand, per_keypoint, grad, ncl);
  if (grad && grad_fused_ok(T, N, D, n_cand, per_keypoint))
    return grad_fused_launch(d, G, y, rconst, M, s_cand, nll, dnll, ws, st, F, fuse != nullptr);
  // ---- grid search on whole 64-chain tiles: head + lean roles in one launch (round 4)
  if (!grad && !F.state && lean_grid_ok(T, N, D, n_cand)) {
    G.nt_log2 = 6;
    G.ntile = (N + 63) / 64;
    G.ngrp = 0;
    G.converged_entry = 1;
    G.ncn = lean_geometry(T, N, n_cand, &G.B0, &G.BN);
    LeanGeom LG;
    LG.ngrp4 = (n_cand + kHeadNCL - 1) / kHeadNCL;
    LG.ngrp16 = (n_cand + kLeanNC - 1) / kLeanNC;
    LG.nhead_blocks = (G.ntile * LG.ngrp4 + kLeanWaves - 1) / kLeanWaves;
    NllWs W;
    W.ncp = (int)align_up((size_t)n_cand, kLeanNC);
    const size_t fl = align_up((size_t)G.ncn * W.ncp * N * sizeof(float), 256);
    const size_t db = align_up((size_t)G.ncn * W.ncp * N * sizeof(double), 256);
    if (2 * db + 11 * fl + adam_extra_bytes(N) > ws_bytes) return EKS_ERR_WORKSPACE;
    char* p = static_cast<char*>(ws);
    W.ell = reinterpret_cast<double*>(p);
    W.dell = nullptr;
    LG.flags = reinterpret_cast<int32_t*>(p + db);          // (the gradient's plane: unused on this path)
    p += 2 * db;
    float* unused_plane = nullptr;
    float** planes[7] = {&W.A, &W.b, &W.C, &W.eta, &W.J, &unused_plane, &W.xr};
    for (int i = 0; i < 7; ++i) *planes[i] = reinterpret_cast<float*>(p + i * fl);
    W.dA = W.db = W.dC = W.deta = W.dJ = nullptr;
    // [flags : ncn x ntile x ncp ints] in the gradient's plane
    const size_t flag_bytes = align_up((size_t)G.ncn * G.ntile * W.ncp * sizeof(int32_t), 256);
    if (flag_bytes > db || (long)G.BN * N * 4 >= (1L << 31)) return EKS_ERR_WORKSPACE;
    // the shared-lag form (round 5)
    LG.lag_on = !per_keypoint && n_cand >= 2 * kLagMinFast && !knob_int(KNOB_NLL_NOLAG, 0);
    LG.rho_max = lag_rho_max(kLagN);
    // the assembly's table behind the float planes
    {
      const size_t tab_bytes = align_up((size_t)N * kTabFields * W.ncp * sizeof(double), 256);
      LG.tab = reinterpret_cast<double*>(p + 7 * fl);
      if (2 * db + 7 * fl + tab_bytes + adam_extra_bytes(N) > ws_bytes) return EKS_ERR_WORKSPACE;
    }
    {
      ProfScope ps("diag_nll_summarize", st);
      const dim3 grid((unsigned)(LG.nhead_blocks + G.ntile * (G.ncn - 1))), block(64 * kLeanWaves);
      if (d.flags & EKS_FLAG_UNIT_AC)
        hipLaunchKernelGGL(diag_nll_grid_kernel<true>, grid, block, 0, st, G, LG, M, W, y, rconst, s_cand);
      else
        hipLaunchKernelGGL(diag_nll_grid_kernel<false>, grid, block, 0, st, G, LG, M, W, y, rconst, s_cand);
    }
    ProfScope ps2("diag_nll_assemble", st);
    if (s_out != nullptr && per_keypoint) return EKS_ERR_UNSUPPORTED;
    const GridAsmOut AO{s_cand, s_out, idx_out};
    // waves of a keypoint's block: its D chains x groups of consecutive chunks (at most kAsmKpWaves in all)
    int ncgw = kAsmKpWaves / D;
    if (ncgw > G.ncn - 1) ncgw = G.ncn - 1;
    const int cpw = (G.ncn - 1 + ncgw - 1) / ncgw;
    ncgw = (G.ncn - 1 + cpw - 1) / cpw;
    const dim3 agrid((unsigned)K), ablock((unsigned)(64 * D * ncgw));
    if (d.flags & EKS_FLAG_UNIT_AC)
      hipLaunchKernelGGL(diag_nll_assemble_kp_kernel<true>, agrid, ablock, 0, st, G, LG, M, W, AO, ncgw, cpw, nll);
    else
      hipLaunchKernelGGL(diag_nll_assemble_kp_kernel<false>, agrid, ablock, 0, st, G, LG, M, W, AO, ncgw, cpw, nll);
    *argmin_done = s_out != nullptr;
    return hip_status(hipGetLastError());
  }
  // (the tree cannot take converged-entry summaries: they are only valid in sequential order)
  const bool tree = nll_uses_tree(K, D, n_cand, G.ncn);
  if (F.state && F.step_in_kernel && !(grad && tree)) return EKS_ERR_UNSUPPORTED;   // (caller asks diag_nll_grad_tree)
  G.converged_entry = !grad && !tree;
  NllWs W;
  W.ncp = G.ngrp * ncl;
  const size_t fl = align_up((size_t)G.ncn * W.ncp * N * sizeof(float), 256);
  const size_t db = align_up((size_t)G.ncn * W.ncp * N * sizeof(double), 256);
  char* p = static_cast<char*>(ws);
  W.ell = reinterpret_cast<double*>(p);
  W.dell = reinterpret_cast<double*>(p + db);
  p += 2 * db;
  float** planes[11] = {&W.A, &W.b, &W.C, &W.eta, &W.J, &W.dA, &W.db, &W.dC, &W.deta, &W.dJ, &W.xr};
  for (int i = 0; i < 11; ++i) *planes[i] = reinterpret_cast<float*>(p + i * fl);

  const int cpw = 64 >> G.nt_log2;
  const long waves = (long)G.ngrp * G.ntile * ((G.ncn + cpw - 1) / cpw);
  int wpb = 4;
  for (int w = 8; w >= 2; w >>= 1)
    if (G.ngrp % w == 0) {
      wpb = w;
      break;
    }
  {
    const int ew = knob_int(KNOB_NLL_WPB, 0);
    if (ew == 1 || ew == 2 || ew == 4 || ew == 8) wpb = ew;
  }
  const dim3 grid((unsigned)((waves + wpb - 1) / wpb)), block(64 * wpb);
  const bool unit = d.flags & EKS_FLAG_UNIT_AC;
  const bool tile64 = G.nt_log2 == 6 && (long)(G.BN > G.B0 ? G.BN : G.B0) * N * 4 < (1L << 31);
#define EKS_NLL_LAUNCH2(RT, NCL, UN, T64)                                                          \
  hipLaunchKernelGGL((diag_nll_summarize_kernel<RT, NCL, UN, T64>), grid, block, 0, st, G, M, W, y, \
                     rconst, s_cand, F)
#define EKS_NLL_LAUNCH(RT, NCL)                          \
  do {                                                   \
    if (unit && tile64)                                  \
      EKS_NLL_LAUNCH2(RT, NCL, true, true);              \
    else if (unit)                                       \
      EKS_NLL_LAUNCH2(RT, NCL, true, false);             \
    else if (tile64)                                     \
      EKS_NLL_LAUNCH2(RT, NCL, false, true);             \
    else                                                 \
      EKS_NLL_LAUNCH2(RT, NCL, false, false);            \
  } while (0)
  {
    ProfScope ps("diag_nll_summarize", st);
    if (grad) {
      EKS_NLL_LAUNCH(Dual, 1);
    } else if (ncl == 8) {
      EKS_NLL_LAUNCH(float, 8);
    } else if (ncl == 4) {
      EKS_NLL_LAUNCH(float, 4);
    } else if (ncl == 2) {
      EKS_NLL_LAUNCH(float, 2);
    } else {
      EKS_NLL_LAUNCH(float, 1);
    }
  }
#undef EKS_NLL_LAUNCH
#undef EKS_NLL_LAUNCH2
  const int total = K * n_cand;
  ProfScope ps2("diag_nll_assemble", st);
  if (tree) {
    const dim3 tb(kAsmLanes, D);
    const size_t shm = ((size_t)D * (grad ? 13 : 7) * kAsmLanes + 2 * D) * sizeof(double);
    if (grad)
      hipLaunchKernelGGL(diag_nll_assemble_tree_kernel<true>, dim3(total), tb, shm, st, G, M, W, K, nll,
                         dnll, F);
    else
      hipLaunchKernelGGL(diag_nll_assemble_tree_kernel<false>, dim3(total), tb, shm, st, G, M, W, K,
                         nll, dnll, F);
    return hip_status(hipGetLastError());
  }
  if (!grad && G.converged_entry && G.nt_log2 == 6 && (D & (D - 1)) == 0 && 64 % D == 0 && G.ncn >= 4) {
    const size_t shm = ((size_t)G.ncn + kAsmWaves) * 64 * sizeof(double);
    if (shm <= 60 * 1024) {
      hipLaunchKernelGGL(diag_nll_assemble_par_kernel, dim3((unsigned)(G.ntile * n_cand)), dim3(64 * kAsmWaves),
                         shm, st, G, M, W, nll);
      return hip_status(hipGetLastError());
    }
  }
  if ((D & (D - 1)) == 0 && D <= 64) {       // chains of a keypoint in adjacent lanes
    const long lanes = (long)N * n_cand;
    if (grad)
      hipLaunchKernelGGL(diag_nll_assemble_chain_kernel<true>, dim3((unsigned)((lanes + 255) / 256)),
                         dim3(256), 0, st, G, M, W, nll, dnll);
    else
      hipLaunchKernelGGL(diag_nll_assemble_chain_kernel<false>, dim3((unsigned)((lanes + 255) / 256)),
                         dim3(256), 0, st, G, M, W, nll, dnll);
    return hip_status(hipGetLastError());
  }
  if (grad)
    hipLaunchKernelGGL(diag_nll_assemble_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, st, G,
                       M, W, K, nll, dnll);
  else
    hipLaunchKernelGGL(diag_nll_assemble_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, st,
                       G, M, W, K, nll, dnll);
  return hip_status(hipGetLastError());
}

}  // namespace eks

#ifdef EKS_GRID_STAMPS
extern "C" int eks_debug_grid_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eks::g_grid_stamps), sizeof(eks::g_grid_stamps));
}
#endif

EKS_DEFINE_TOUCH(diag_nll)
