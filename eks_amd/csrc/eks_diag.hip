// gfx950 kernels for the scalar-chain ("diagonal model") smoother: A, C, Q, S0 diagonal and D == O,
// so every keypoint coordinate is an independent scalar Kalman chain (singlecam: reference
// eks/singlecam_smoother.py:246-284 builds A = C = Q = I).  Replaces dynamax's sequential
// lax.scan pair (extended_kalman_smoother called at eks/core.py:290) with a three-kernel chunked
// associative scan so the time axis is parallel:
//
//   K1 diag_summarize : lane = (chain, chunk of B frames).  Reads y, var once, composes the chunk's
//                       element (A, b, C, eta, J) in registers, writes 20 B per chunk.
//   K2 diag_scan_*    : per chain, scans the chunk elements: forward -> predicted belief entering
//                       every chunk; backward -> information about the state just after every
//                       chunk from all later frames.  Block-local LDS scans + a short pass
//                       over block aggregates (three small launches).
//   K3 diag_replay    : lane = (chain, chunk).  Reads y, var again, replays the exact filter from
//                       the chunk's incoming belief keeping the B filtered (m, P) pairs IN
//                       REGISTERS, fuses the outgoing belief with the future information and runs
//                       RTS backwards over the registers, streaming ms / Vs out.
//
// For N > 32 chains the scan is folded into its neighbours (round 2): a block of K1 / K3 is
// kFW (= 4) consecutive chunks of one 64-chain tile, K1's block composes its kFW elements into the
// block aggregate (LDS ticket: the last wave to arrive), one small launch scans the aggregates per
// chain, and every K3 wave composes the <= kFW - 1 elements before / after its own chunk inside the
// block itself - 3 launches
// instead of 5, the per-chunk scan results (25.6 MB written + read on the C3 shape) and the second
// read of the elements never exist.
//
// (Tried for small problems, round 2: ONE launch with two grid barriers and the chunk kept in
// registers in between.  Bit-identical results, but a counter barrier over 158-391 workgroups costs
// far more than the 1.5-1.9 us of a kernel boundary on this part - 10 000 x 64 keypoints: 52 us
// against 20 us for the three launches, 50 000 x 30: 187 against 29 - see
// profiles/r02_overlap_probes.txt.  Dropped.)
//
// HBM traffic: 2 x (y + var) in, 1 x (ms + Vs) out (+ ~4 % for the chunk elements); no filtered
// state ever touches memory.  No MFMA: the algebra is scalar.  No LDS in K1/K3: each datum is
// consumed by the lane that loads it; lanes of a wave are consecutive chains, so every row access
// is a contiguous 256 B (x4 for the four waves of a block: 1 KiB contiguous per frame).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "eks_diag_lane.hpp"
#include "eks_internal.hpp"

namespace eks {

// ------------------------------------------------------------------------------------------
// lane -> (chain n, chunk j) mapping shared by K1 and K3
// ------------------------------------------------------------------------------------------
struct LaneMap {
  int N, T, nc;       // chains, frames, chunks
  int nt_log2;        // log2 of chains per wave row (NT = min(64, pow2ceil(N)))
  int ntile;          // ceil(N / NT)
  int reverse;        // K3 only: walk the chunks backwards in time
};

__device__ __forceinline__ bool lane_coords(const LaneMap& L, int& n, int& j) {
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int tile = wave % L.ntile;
  const int cg = wave / L.ntile;
  const int nt = 1 << L.nt_log2;
  n = tile * nt + (lane & (nt - 1));
  j = cg * (64 >> L.nt_log2) + (lane >> L.nt_log2);
  return n < L.N && j < L.nc;
}

struct DiagWs {
  // chunk elements [nc][N]
  float *eA, *eb, *eC, *eEta, *eJ;
  // scan results [nc][N]: predicted belief entering chunk j; information after chunk j
  float *pm, *pP, *sEta, *sJ;
};

template <int B, bool UNIT>
__global__ __launch_bounds__(256) void diag_summarize_kernel(LaneMap L, DiagModel M, DiagWs W,
                                                            const float* __restrict__ y,
                                                            const float* __restrict__ var) {
  int n, j;
  if (!lane_coords(L, n, j)) return;
  if (L.reverse) j = L.nc - 1 - j;
  const ChainParams<float> p = load_chain_params(M, n);
  const int t0 = j * B;
  const int len = min(B, L.T - t0);
  const Elem<float> e = summarize_chunk<B, UNIT>(y, var, L.N, n, t0, len, p);
  const size_t o = (size_t)j * L.N + n;
  W.eA[o] = e.A;
  W.eb[o] = e.b;
  W.eC[o] = e.C;
  W.eEta[o] = e.eta;
  W.eJ[o] = e.J;
}

template <int B, bool UNIT, int VS_ROW>
__global__ __launch_bounds__(256) void diag_replay_kernel(LaneMap L, DiagModel M, DiagWs W,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ var,
                                                         float* __restrict__ ms,
                                                         float* __restrict__ Vs) {
  int n, j;
  if (!lane_coords(L, n, j)) return;
  // K1 streamed y, var forwards in time; walking the chunks backwards here lets the tail of that
  // stream be served from the 256 MiB Infinity Cache (EKS_REPLAY_FORWARD=1 disables, for A/B)
  if (L.reverse) j = L.nc - 1 - j;
  const ChainParams<float> p = load_chain_params(M, n);
  const int t0 = j * B;
  const int len = min(B, L.T - t0);
  const size_t o = (size_t)j * L.N + n;
  replay_chunk<B, UNIT, VS_ROW>(y, var, ms, Vs, L.N, n, n % M.D, t0, len, p, W.pm[o], W.pP[o],
                                W.sEta[o], W.sJ[o]);
}

// ------------------------------------------------------------------------------------------
// K2: scan of the chunk elements, three small launches.  Block = kScanCH chains x kScanCB
// consecutive chunks (one thread per element, so every load is issued at once):
//   S1 reduce : ordered tree reduction of the block's kScanCB elements -> block aggregate
//   S2 blocks : one thread per chain walks the (few) block aggregates: belief entering each
//               block (forward) and information leaving each block (backward)
//   S3 local  : inclusive scans of the block's elements, forward and reverse; exclusive prefixes
//               applied to the block's incoming belief / pulled back from the block's outgoing
//               information -> per-chunk (pm, pP, sEta, sJ)
// S1 and S3 load with lanes along chains (coalesced), transpose through LDS and compose with wave
// shuffles, one wave per chain: two barriers instead of the two per level of the first version
// (Hillis-Steele through LDS).  Same time on the 512-chain C3 shape (latency-bound either way),
// 14 % less on the 8192-chain C5 shape.
// ------------------------------------------------------------------------------------------
constexpr int kScanCH = 16;
constexpr int kScanCB = 64;

struct ScanWs {
  float *gA, *gb, *gC, *gEta, *gJ;   // block aggregates [nblk][N]
  float *bm, *bP, *bEta, *bJ;        // block-level scan results [nblk][N]
  int nblk;
};

__device__ __forceinline__ Elem<float> load_elem(const DiagWs& W, size_t o) {
  return Elem<float>{W.eA[o], W.eb[o], W.eC[o], W.eEta[o], W.eJ[o]};
}

// LDS tile of a block's elements, [chunk][chain] with the chain dimension padded to 17 so that
// both access patterns are conflict-free: filled with lanes along chains (the coalesced global
// order), read back with lanes along chunks (one wave = the 64 chunks of one chain).
struct ScanLds {
  float A[kScanCB][kScanCH + 1], b[kScanCB][kScanCH + 1], C[kScanCB][kScanCH + 1],
      eta[kScanCB][kScanCH + 1], J[kScanCB][kScanCH + 1];
  __device__ __forceinline__ void put(int i, int c, const Elem<float>& e) {
    A[i][c] = e.A; b[i][c] = e.b; C[i][c] = e.C; eta[i][c] = e.eta; J[i][c] = e.J;
  }
  __device__ __forceinline__ Elem<float> get(int i, int c) const {
    return Elem<float>{A[i][c], b[i][c], C[i][c], eta[i][c], J[i][c]};
  }
};

__device__ __forceinline__ Elem<float> shfl_up_elem(const Elem<float>& e, int off) {
  return Elem<float>{__shfl_up(e.A, off), __shfl_up(e.b, off), __shfl_up(e.C, off), __shfl_up(e.eta, off),
                     __shfl_up(e.J, off)};
}
__device__ __forceinline__ Elem<float> shfl_down_elem(const Elem<float>& e, int off) {
  return Elem<float>{__shfl_down(e.A, off), __shfl_down(e.b, off), __shfl_down(e.C, off),
                     __shfl_down(e.eta, off), __shfl_down(e.J, off)};
}

// Both scan kernels: thread t loads element (chain t % 16, chunk t / 16) - 64-byte runs along the
// chains - into the LDS tile; then wave w owns chain w with its lanes along the 64 chunks and
// composes through wave shuffles (no further barriers).
__global__ __launch_bounds__(kScanCH* kScanCB) void diag_scan_reduce_kernel(int N, int nc, DiagWs W,
                                                                           ScanWs S) {
  __shared__ ScanLds L;
  const int cl = threadIdx.x % kScanCH, i = threadIdx.x / kScanCH;
  const int n = blockIdx.x * kScanCH + cl, j = blockIdx.y * kScanCB + i;
  L.put(i, cl, (n < N && j < nc) ? load_elem(W, (size_t)j * N + n) : elem_identity<float>());
  __syncthreads();
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  Elem<float> e = L.get(lane, w);
#pragma unroll
  for (int off = 1; off < kScanCB; off <<= 1) {       // ordered tree: lane 0 ends with e_0 o ... o e_63
    const Elem<float> other = shfl_down_elem(e, off);
    if ((lane & (2 * off - 1)) == 0) e = elem_combine(e, other);
  }
  const int nw = blockIdx.x * kScanCH + w;
  if (lane == 0 && nw < N) {
    const size_t o = (size_t)blockIdx.y * N + nw;
    S.gA[o] = e.A; S.gb[o] = e.b; S.gC[o] = e.C; S.gEta[o] = e.eta; S.gJ[o] = e.J;
  }
}

// S2: one wave per chain, lanes along the block aggregates (64 per pass): inclusive forward /
// reverse compositions by shuffles; the exclusive prefix pushed through the chain's running belief
// gives the belief entering each block, the exclusive suffix pulled back from the running
// information gives what leaves it.  (The first version walked the aggregates one by one.)
__global__ __launch_bounds__(256) void diag_scan_blocks_kernel(int N, DiagModel M, ScanWs S) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;                                       // wave-uniform
  auto agg = [&](int q) {
    const size_t o = (size_t)q * N + n;
    return q < S.nblk ? Elem<float>{S.gA[o], S.gb[o], S.gC[o], S.gEta[o], S.gJ[o]} : elem_identity<float>();
  };
  float m, P;
  load_chain_prior(M, n, m, P);
  for (int q0 = 0; q0 < S.nblk; q0 += 64) {                 // forward: belief entering block q0 + lane
    const Elem<float> own = agg(q0 + lane);
    Elem<float> f = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const Elem<float> up = shfl_up_elem(f, off);
      if (lane >= off) f = elem_combine(up, f);
    }
    Elem<float> excl = shfl_up_elem(f, 1);
    if (lane == 0) excl = elem_identity<float>();
    float mq = m, Pq = P;
    elem_apply(excl, mq, Pq);
    if (q0 + lane < S.nblk) {
      const size_t o = (size_t)(q0 + lane) * N + n;
      S.bm[o] = mq;
      S.bP[o] = Pq;
    }
    // carry the belief past these 64 aggregates (lane 63 holds their whole composition)
    const Elem<float> all{__shfl(f.A, 63), __shfl(f.b, 63), __shfl(f.C, 63), __shfl(f.eta, 63), __shfl(f.J, 63)};
    elem_apply(all, m, P);
  }
  float eta = 0.f, J = 0.f;
  const int npass = (S.nblk + 63) / 64;
  for (int pass = npass - 1; pass >= 0; --pass) {           // backward: information leaving each block
    const int q0 = pass * 64;
    const Elem<float> own = agg(q0 + lane);
    Elem<float> r = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const Elem<float> dn = shfl_down_elem(r, off);
      if (lane + off < 64) r = elem_combine(r, dn);
    }
    Elem<float> after = shfl_down_elem(r, 1);
    if (lane == 63) after = elem_identity<float>();
    float eq = eta, Jq = J;
    elem_back(after, eq, Jq);
    if (q0 + lane < S.nblk) {
      const size_t o = (size_t)(q0 + lane) * N + n;
      S.bEta[o] = eq;
      S.bJ[o] = Jq;
    }
    const Elem<float> all{__shfl(r.A, 0), __shfl(r.b, 0), __shfl(r.C, 0), __shfl(r.eta, 0), __shfl(r.J, 0)};
    elem_back(all, eta, J);
  }
}

__global__ __launch_bounds__(kScanCH* kScanCB) void diag_scan_local_kernel(int N, int nc, DiagWs W,
                                                                          ScanWs S) {
  __shared__ ScanLds L;
  const int cl = threadIdx.x % kScanCH, i = threadIdx.x / kScanCH;
  const int n = blockIdx.x * kScanCH + cl, j = blockIdx.y * kScanCB + i;
  const bool live = n < N && j < nc;
  L.put(i, cl, live ? load_elem(W, (size_t)j * N + n) : elem_identity<float>());
  __syncthreads();
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const Elem<float> own = L.get(lane, w);
  // forward inclusive scan F[i] = e_0 o ... o e_i, reverse inclusive scan R[i] = e_i o ... o e_63
  Elem<float> f = own, r = own;
#pragma unroll
  for (int off = 1; off < kScanCB; off <<= 1) {
    const Elem<float> up = shfl_up_elem(f, off), dn = shfl_down_elem(r, off);
    if (lane >= off) f = elem_combine(up, f);
    if (lane + off < kScanCB) r = elem_combine(r, dn);
  }
  Elem<float> excl = shfl_up_elem(f, 1), after = shfl_down_elem(r, 1);
  if (lane == 0) excl = elem_identity<float>();
  if (lane == kScanCB - 1) after = elem_identity<float>();
  const int nw = blockIdx.x * kScanCH + w;
  float m = 0.f, P = 0.f, eta = 0.f, J = 0.f;
  if (nw < N) {
    const size_t ob = (size_t)blockIdx.y * N + nw;
    m = S.bm[ob];
    P = S.bP[ob];
    eta = S.bEta[ob];
    J = S.bJ[ob];
  }
  elem_apply(excl, m, P);
  elem_back(after, eta, J);
  __syncthreads();                       // every wave has read its chain: the tile is free
  L.A[lane][w] = m;
  L.b[lane][w] = P;
  L.C[lane][w] = eta;
  L.eta[lane][w] = J;
  __syncthreads();
  if (!live) return;
  const size_t o = (size_t)j * N + n;
  W.pm[o] = L.A[i][cl];
  W.pP[o] = L.b[i][cl];
  W.sEta[o] = L.C[i][cl];
  W.sJ[o] = L.eta[i][cl];
}

// ------------------------------------------------------------------------------------------
// Fused form for N > 32 (64-chain tiles): block = kFW waves = kFW consecutive chunks of one tile.
// ------------------------------------------------------------------------------------------
constexpr int kFW = 4;
static_assert(kFW == 4, "the fused kernels' LDS exchange, the group scan's slot sizes and the measured "
                        "thresholds in diag_smooth() assume four chunks per block");

struct BlockMap {
  int N, T, nc;
  int ntile;     // 64-chain tiles covered by THIS launch (all of ceil(N / 64), or one pass's share)
  int ngrp;      // chunk groups = ceil(nc / kFW)
  int reverse;   // walk the chunk groups backwards in time
  int tile0;     // first tile of this launch (keypoint-tiled passes, diag_smooth)
};

// Row access of the fused kernels through buffer resources based at the first row of the wave's
// chunk and tile (scalar): `buffer_load_dword v, v_lane, s[rsrc], s_row offen` - the row offset is
// an SGPR, the lane offset a constant VGPR, so loads and stores cost no VALU address arithmetic
// (K1 ran 27 VALU instructions per frame for 13 of arithmetic before; a chunk's rows span
// 32 x N x 4 W bytes, checked against the 2 GiB offset range by the launch code).
struct BufferRows {
  __amdgpu_buffer_rsrc_t ry, rv;
  unsigned voff, row_bytes;
  __device__ __forceinline__ float load_y(int i) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, voff, (unsigned)i * row_bytes, 0));
  }
  __device__ __forceinline__ float load_var(int i) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, voff, (unsigned)i * row_bytes, 0));
  }
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7FFFFFFF, 0x00020000);
}

constexpr int kAuxNt = 2;                  // gfx950 cache policy: non-temporal (see EKS_STREAM_STORE)

template <int VS_ROW>
struct BufferStore {
  __amdgpu_buffer_rsrc_t rm, rV;
  unsigned voff, row_bytes;                // of ms; Vs rows are W times as wide
  int d;
  __device__ __forceinline__ void operator()(int i, float m, float P) const {
    constexpr unsigned W = VS_ROW == 0 ? 1 : VS_ROW;
    const unsigned so = (unsigned)i * row_bytes;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m), rm, voff, so, kAuxNt);
    if constexpr (VS_ROW <= 1) {
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, P), rV, voff, so, kAuxNt);
    } else if constexpr (VS_ROW == 2) {
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      const u2 v = {__builtin_bit_cast(unsigned, d == 0 ? P : 0.0f), __builtin_bit_cast(unsigned, d == 1 ? P : 0.0f)};
      __builtin_amdgcn_raw_buffer_store_b64(v, rV, voff * 2, so * 2, kAuxNt);
    } else {
#pragma unroll
      for (unsigned e = 0; e < W; ++e)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (int)e == d ? P : 0.0f), rV,
                                              voff * W + 4 * e, so * W, kAuxNt);
    }
  }
};

// K1f: chunk elements + the block's aggregate (time-ordered composition of its kFW elements by
// the last wave to arrive, lanes = chains; rows past the end of the sequence are identities).
template <int B, bool UNIT, bool RC>
__global__ __launch_bounds__(64 * kFW) void diag_summarize_blk_kernel(BlockMap L, DiagModel M, DiagWs W,
                                                                     ScanWs S,
                                                                     const float* __restrict__ y,
                                                                     const float* __restrict__ var) {
  __shared__ float sh[5][kFW][64];
  __shared__ int arrived;
  // the wave index through readfirstlane: chunk, first frame and length are then scalars to the
  // compiler (row addresses in SGPRs, the tail test a scalar branch)
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (threadIdx.x == 0) arrived = 0;
  __syncthreads();                         // (at the very start: costs nothing, the waves launch together)
  const int tile = L.tile0 + blockIdx.x % L.ntile;
  int grp = blockIdx.x / L.ntile;
  if (L.reverse) grp = L.ngrp - 1 - grp;
  const int n = tile * 64 + lane, j = grp * kFW + w;
  Elem<float> e = elem_identity<float>();
  if (n < L.N && j < L.nc) {
    const ChainParams<float> p = load_chain_params(M, n);
    const int t0 = j * B, len = min(B, L.T - t0);
    const size_t first = (size_t)t0 * L.N + (size_t)tile * 64;
    const BufferRows rows{rows_rsrc(y + first), rows_rsrc(var + first), (unsigned)lane * 4, (unsigned)L.N * 4};
    float yy[B], rr[B];
    if (len == B) {                        // wave-uniform: all chunks but a sequence's last
      load_rows<B, true>(rows, B, yy, rr);
      e = summarize_loaded<B, UNIT, true>(yy, rr, B, p);
    } else {
      load_rows<B, false>(rows, len, yy, rr);
      e = summarize_loaded<B, UNIT, false>(yy, rr, len, p);
    }
    if constexpr (!RC) {                   // RC: K3 summarises its chunks again, nothing to keep
      const size_t o = (size_t)j * L.N + n;
      W.eA[o] = e.A;
      W.eb[o] = e.b;
      W.eC[o] = e.C;
      W.eEta[o] = e.eta;
      W.eJ[o] = e.J;
    }
  }
  sh[0][w][lane] = e.A;
  sh[1][w][lane] = e.b;
  sh[2][w][lane] = e.C;
  sh[3][w][lane] = e.eta;
  sh[4][w][lane] = e.J;
  // no workgroup barrier: the wave that arrives LAST composes the block aggregate (LDS operations
  // of a wave complete in order, so its ticket is taken after its element is in LDS), the others
  // retire at once and free their slots
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  int ticket = 0;
  if (lane == 0) ticket = atomicAdd(&arrived, 1);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  if (ticket != kFW - 1 || n >= L.N) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  Elem<float> a{sh[0][0][lane], sh[1][0][lane], sh[2][0][lane], sh[3][0][lane], sh[4][0][lane]};
#pragma unroll
  for (int q = 1; q < kFW; ++q)
    a = elem_combine(a, Elem<float>{sh[0][q][lane], sh[1][q][lane], sh[2][q][lane], sh[3][q][lane],
                                    sh[4][q][lane]});
  const size_t o = (size_t)grp * L.N + n;
  S.gA[o] = a.A;
  S.gb[o] = a.b;
  S.gC[o] = a.C;
  S.gEta[o] = a.eta;
  S.gJ[o] = a.J;
}

// K3f: replay of chunk j = grp * kFW + w.  The belief entering the block and the information
// leaving it come from the scan of the block aggregates; the wave pushes the former through the
// elements of the chunks before its own and pulls the latter back through those after it (<= 7
// compositions each, element rows shared by the block's waves through L2) before it streams its
// own 64 rows of y, var (holding both sets of elements in registers beside the chunk cost 180 VGPRs
// and two thirds of the occupancy).
template <int B, bool UNIT, int VS_ROW, bool RC>
__global__ __launch_bounds__(64 * kFW) void diag_replay_blk_kernel(BlockMap L, DiagModel M, DiagWs W,
                                                                  ScanWs S,
                                                                  const float* __restrict__ y,
                                                                  const float* __restrict__ var,
                                                                  float* __restrict__ ms,
                                                                  float* __restrict__ Vs) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = L.tile0 + blockIdx.x % L.ntile;
  int grp = blockIdx.x / L.ntile;
  if (L.reverse) grp = L.ngrp - 1 - grp;
  const int n = tile * 64 + lane, j = grp * kFW + w;
  const int j0 = grp * kFW, j1 = min(j0 + kFW, L.nc);
  constexpr int VW = VS_ROW == 0 ? 1 : VS_ROW;
  if constexpr (!RC) {
    if (n >= L.N || j >= L.nc) return;
    // everything the forward pass needs is requested at once: the chunk's own 64 rows, then the
    // elements of the block's earlier chunks (wave-uniform predicate)
    const ChainParams<float> p = load_chain_params(M, n);
    const int t0 = j * B;
    const int len = min(B, L.T - t0);
    const bool full = len == B;            // wave-uniform: all chunks but a sequence's last
    const size_t first = (size_t)t0 * L.N + (size_t)tile * 64;
    const BufferRows rows{rows_rsrc(y + first), rows_rsrc(var + first), (unsigned)lane * 4, (unsigned)L.N * 4};
    float v0[B], v1[B];
    if (full) load_rows<B, true>(rows, B, v0, v1);
    else load_rows<B, false>(rows, len, v0, v1);
    Elem<float> ef[kFW - 1];
#pragma unroll
    for (int q = 0; q < kFW - 1; ++q)
      if (j0 + q < j) ef[q] = load_elem(W, (size_t)(j0 + q) * L.N + n);
    const size_t ob = (size_t)grp * L.N + n;
    float m = S.bm[ob], P = S.bP[ob], eta = S.bEta[ob], J = S.bJ[ob];
#pragma unroll
    for (int q = 0; q < kFW - 1; ++q)
      if (j0 + q < j) elem_apply(ef[q], m, P);
    // the later chunks' elements arrive while the filter runs over the registers
    Elem<float> eb[kFW - 1];
#pragma unroll
    for (int q = 0; q < kFW - 1; ++q)
      if (j + 1 + q < j1) eb[q] = load_elem(W, (size_t)(j + 1 + q) * L.N + n);
    if (full) filter_loaded<B, UNIT, true>(v0, v1, B, p, m, P);
    else filter_loaded<B, UNIT, false>(v0, v1, len, p, m, P);
#pragma unroll
    for (int q = kFW - 2; q >= 0; --q)
      if (j + 1 + q < j1) elem_back(eb[q], eta, J);
    fuse_info(m, P, eta, J);
    const BufferStore<VS_ROW> st{rows_rsrc(ms + first), rows_rsrc(Vs + first * VW), (unsigned)lane * 4,
                                 (unsigned)L.N * 4, n % M.D};
    if (full) smooth_rows<B, UNIT, true>(v0, v1, B, p, m, P, st);
    else smooth_rows<B, UNIT, false>(v0, v1, len, p, m, P, st);
  } else {
    // K1 kept no chunk elements: every wave summarises its own chunk again from the rows it has just
    // loaded (15 VALU instructions per frame) and the block's waves exchange the elements through
    // LDS, read where they are used - no neighbour-element registers (124 VGPRs, 4 waves per SIMD).
    __shared__ float sh[5][kFW][64];
    const bool live = n < L.N && j < L.nc;   // every wave reaches the block's barrier
    ChainParams<float> p{1.f, 1.f, 0.f};
    const int t0 = j < L.nc ? j * B : 0;
    const int len = j < L.nc ? min(B, L.T - t0) : 0;
    const bool full = len == B;
    const size_t first = (size_t)t0 * L.N + (size_t)tile * 64;
    const BufferRows rows{rows_rsrc(y + first), rows_rsrc(var + first), (unsigned)lane * 4, (unsigned)L.N * 4};
    float v0[B], v1[B];
    Elem<float> own = elem_identity<float>();
    if (live) {
      p = load_chain_params(M, n);
      if (full) {
        load_rows<B, true>(rows, B, v0, v1);
        own = summarize_loaded<B, UNIT, true>(v0, v1, B, p);
      } else {
        load_rows<B, false>(rows, len, v0, v1);
        own = summarize_loaded<B, UNIT, false>(v0, v1, len, p);
      }
    }
    sh[0][w][lane] = own.A; sh[1][w][lane] = own.b; sh[2][w][lane] = own.C; sh[3][w][lane] = own.eta;
    sh[4][w][lane] = own.J;
    __syncthreads();
    if (!live) return;
    auto lds_elem = [&](int q) {
      return Elem<float>{sh[0][q][lane], sh[1][q][lane], sh[2][q][lane], sh[3][q][lane], sh[4][q][lane]};
    };
    const size_t ob = (size_t)grp * L.N + n;
    float m = S.bm[ob], P = S.bP[ob], eta = S.bEta[ob], J = S.bJ[ob];
#pragma unroll
    for (int q = 0; q < kFW - 1; ++q)
      if (q < w) elem_apply(lds_elem(q), m, P);
    Elem<float> eb[kFW - 1];
#pragma unroll
    for (int q = 0; q < kFW - 1; ++q)
      if (j + 1 + q < j1) eb[q] = lds_elem(w + 1 + q);
    if (full) filter_loaded<B, UNIT, true>(v0, v1, B, p, m, P);
    else filter_loaded<B, UNIT, false>(v0, v1, len, p, m, P);
#pragma unroll
    for (int q = kFW - 2; q >= 0; --q)
      if (j + 1 + q < j1) elem_back(eb[q], eta, J);
    fuse_info(m, P, eta, J);
    const BufferStore<VS_ROW> st{rows_rsrc(ms + first), rows_rsrc(Vs + first * VW), (unsigned)lane * 4,
                                 (unsigned)L.N * 4, n % M.D};
    if (full) smooth_rows<B, UNIT, true>(v0, v1, B, p, m, P, st);
    else smooth_rows<B, UNIT, false>(v0, v1, len, p, m, P, st);
  }
}

// S2f: scan of the block aggregates, ngrp per chain (hundreds to thousands).  Block = CH chains x
// 64 slots (thread t: chain t % CH, slot t / CH; CH = 16: 1024 threads, every load instruction
// reads whole 64-byte segments of the [group][chain] planes - with 4 chains per block the 16-byte
// pieces moved 77 MB for 20 MB of aggregates); a slot owns `per` consecutive aggregates, all
// requested at once (registers, PER at a time): it composes them, the slots of a chain are scanned
// by stride-CH shuffles inside each wave (64 / CH slots) and through LDS across the waves, and each
// slot walks its aggregates again from the belief entering its first one (forward) / the
// information leaving its last one (backward), writing the per-block results.
template <int PER, int CH>
__global__ __launch_bounds__(64 * CH) void diag_scan_groups_kernel(int N, int n0, int n1, DiagModel M, ScanWs S) {
  constexpr int NW = CH;                   // waves per block (64 slots x CH chains / 64 lanes)
  __shared__ float tot[2][5][NW][CH];      // [direction][field][wave][chain]
  const int c = threadIdx.x % CH, slot = threadIdx.x / CH, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // Workgroups are dealt to the 8 XCDs round-robin and each XCD has its own L2: with CH = 4 the 16-byte pieces
  // of one 64-byte line belong to four consecutive blocks, i.e. to four different L2s, and the launch fetched
  // 64 MB for 8 MB of aggregates on the C3 shape (rocprofv3 FETCH_SIZE, profiles/r04_a_pmc_summary.txt).  Blocks
  // of one XCD therefore take CONSECUTIVE chain groups (XCD x: the x-th eighth of the chains).
  const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int n = n0 + lb * CH + c;           // n0: first chain of this launch (keypoint-tiled passes)
  const int per = (S.nblk + 63) / 64;      // <= PER, or the slot re-reads in batches of PER
  const int q0 = min(slot * per, S.nblk), q1 = min(q0 + per, S.nblk);
  const bool live = n < n1;                // chains [n0, n1) of the N whose planes these are
  auto agg = [&](int q) {
    const size_t o = (size_t)q * N + n;
    return Elem<float>{S.gA[o], S.gb[o], S.gC[o], S.gEta[o], S.gJ[o]};
  };
  Elem<float> own = elem_identity<float>();
  Elem<float> e[PER];
  if (live) {
    for (int b0 = q0; b0 < q1; b0 += PER) {
#pragma unroll
      for (int i = 0; i < PER; ++i)
        if (b0 + i < q1) e[i] = agg(b0 + i);
#pragma unroll
      for (int i = 0; i < PER; ++i)
        if (b0 + i < q1) own = elem_combine(own, e[i]);
    }
  }
  // inclusive scans over the wave's slots of each chain (lanes CH apart), both directions
  Elem<float> f = own, r = own;
#pragma unroll
  for (int off = CH; off < 64; off <<= 1) {
    const Elem<float> up = shfl_up_elem(f, off), dn = shfl_down_elem(r, off);
    if (lane >= off) f = elem_combine(up, f);
    if (lane + off < 64) r = elem_combine(r, dn);
  }
  if (lane >= 64 - CH) {                   // the wave's whole composition, per chain
    tot[0][0][w][c] = f.A; tot[0][1][w][c] = f.b; tot[0][2][w][c] = f.C; tot[0][3][w][c] = f.eta; tot[0][4][w][c] = f.J;
  }
  if (lane < CH) {
    tot[1][0][w][c] = r.A; tot[1][1][w][c] = r.b; tot[1][2][w][c] = r.C; tot[1][3][w][c] = r.eta; tot[1][4][w][c] = r.J;
  }
  __syncthreads();
  if (!live) return;
  // exclusive prefix of this slot: earlier waves' totals, then the wave's earlier slots
  Elem<float> pre = elem_identity<float>(), post = elem_identity<float>();
  for (int v = 0; v < w; ++v)
    pre = elem_combine(pre, Elem<float>{tot[0][0][v][c], tot[0][1][v][c], tot[0][2][v][c], tot[0][3][v][c], tot[0][4][v][c]});
  for (int v = NW - 1; v > w; --v)
    post = elem_combine(Elem<float>{tot[1][0][v][c], tot[1][1][v][c], tot[1][2][v][c], tot[1][3][v][c], tot[1][4][v][c]}, post);
  Elem<float> fe = shfl_up_elem(f, CH), re = shfl_down_elem(r, CH);
  if (lane < CH) fe = elem_identity<float>();
  if (lane >= 64 - CH) re = elem_identity<float>();
  pre = elem_combine(pre, fe);
  post = elem_combine(re, post);
  float m, P;
  load_chain_prior(M, n, m, P);
  elem_apply(pre, m, P);                   // belief entering aggregate q0
  float eta = 0.f, J = 0.f;
  elem_back(post, eta, J);                 // information leaving aggregate q1 - 1
  const bool held = q1 - q0 <= PER;        // the single batch is still in registers
  for (int b0 = q0; b0 < q1; b0 += PER) {
    if (!held) {
#pragma unroll
      for (int i = 0; i < PER; ++i)
        if (b0 + i < q1) e[i] = agg(b0 + i);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i)
      if (b0 + i < q1) {
        const size_t o = (size_t)(b0 + i) * N + n;
        S.bm[o] = m;
        S.bP[o] = P;
        elem_apply(e[i], m, P);
      }
  }
  for (int b1 = q1; b1 > q0; b1 -= PER) {  // batches [b1 - PER, b1) from the end
    const int lo = max(q0, b1 - PER);
    if (!held) {
#pragma unroll
      for (int i = 0; i < PER; ++i)
        if (lo + i < b1) e[i] = agg(lo + i);
    }
#pragma unroll
    for (int i = PER - 1; i >= 0; --i)
      if (lo + i < b1) {
        const size_t o = (size_t)(lo + i) * N + n;
        S.bEta[o] = eta;
        S.bJ[o] = J;
        elem_back(e[i], eta, J);
      }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
#ifndef EKS_DIAG_CHUNK
#define EKS_DIAG_CHUNK 32
#endif
constexpr int kChunk = EKS_DIAG_CHUNK;  // frames per lane; 2*B VGPRs hold the chunk in K3

static inline size_t plane_bytes(int nc, int N) { return align_up((size_t)nc * N * sizeof(float), 256); }

size_t diag_smooth_workspace_bytes(int T, int N) {
  const int nc = (T + kChunk - 1) / kChunk;
  const int nblk = (nc + kFW - 1) / kFW;       // fused grouping (>= the legacy nc / kScanCB blocks)
  return 9 * plane_bytes(nc, N) + 9 * plane_bytes(nblk, N);
}

static LaneMap make_lane_map(int T, int N, int B) {
  LaneMap L;
  L.N = N;
  L.T = T;
  L.nc = (T + B - 1) / B;
  int nt_log2 = 0;
  while ((1 << nt_log2) < N && nt_log2 < 6) ++nt_log2;
  L.nt_log2 = nt_log2;
  L.ntile = (N + (1 << nt_log2) - 1) >> nt_log2;
  L.reverse = 0;
  return L;
}

template <bool UNIT>
static void launch_replay(int vs_row, dim3 grid, hipStream_t st, const LaneMap& L, const DiagModel& M,
                          const DiagWs& W, const float* y, const float* var, float* ms, float* Vs) {
#define EKS_REPLAY(R)                                                                             \
  case R:                                                                                         \
    hipLaunchKernelGGL((diag_replay_kernel<kChunk, UNIT, R>), grid, dim3(256), 0, st, L, M, W, y, \
                       var, ms, Vs);                                                              \
    break;
  switch (vs_row) {
    EKS_REPLAY(0)
    EKS_REPLAY(1)
    EKS_REPLAY(2)
    EKS_REPLAY(3)
    EKS_REPLAY(4)
    EKS_REPLAY(5)
    EKS_REPLAY(6)
    EKS_REPLAY(7)
    EKS_REPLAY(8)
  }
#undef EKS_REPLAY
}

int diag_smooth(const eks_dims_t& d, const float* y, const float* var, const DiagModel& M,
                float* ms, float* Vs, void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, D = d.state_dim, N = d.n_keypoints * D;
  const bool vs_diag = d.flags & EKS_FLAG_VS_DIAG;
  if (!vs_diag && D > 8) return EKS_ERR_UNSUPPORTED;
  if (ws_bytes < diag_smooth_workspace_bytes(T, N)) return EKS_ERR_WORKSPACE;
  const LaneMap Lf = make_lane_map(T, N, kChunk);
  const LaneMap& L = Lf;
  const size_t pb = plane_bytes(L.nc, N);
  char* base = static_cast<char*>(ws);
  DiagWs W;
  float** planes[9] = {&W.eA, &W.eb, &W.eC, &W.eEta, &W.eJ, &W.pm, &W.pP, &W.sEta, &W.sJ};
  for (int i = 0; i < 9; ++i) *planes[i] = reinterpret_cast<float*>(base + i * pb);
  // Fused scan (measured, MI355X, fixed s, full Vs; three-kernel scan -> fused): 100 000 x 256
  // keypoints 0.309 -> 0.293 ms, 10 000 x 64 31 -> 22 us, 50 000 x 1024 0.632 -> 0.602 ms,
  // 50 000 x 4096 2.45 -> 2.41 ms (there the fused summarize / replay run 17 % / 7 % slower than
  // the plain ones - barrier, neighbour compositions - which eats most of what the scan saves).
  // Not for sequences whose group scan would leave its registers (more than 16 aggregates per
  // slot, T > 131 072: 400 000 x 64 0.345 -> 0.391 ms).
  const int ngrp_f = (L.nc + kFW - 1) / kFW;
  // (the fused kernels address a chunk's rows with 32-bit buffer offsets: kChunk rows of N x D floats)
  const bool in_range = (size_t)kChunk * N * (vs_diag ? 1 : D) * sizeof(float) < ((size_t)1 << 31);
  const bool fused = L.nt_log2 == 6 && in_range &&
                     (knob_set(KNOB_SMOOTH_UNFUSED) ? knob_int(KNOB_SMOOTH_UNFUSED, 0) == 0 : ngrp_f <= 1024);
  ScanWs S;
  S.nblk = fused ? (L.nc + kFW - 1) / kFW : (L.nc + kScanCB - 1) / kScanCB;
  const size_t sb = plane_bytes(S.nblk, N);
  float** splanes[9] = {&S.gA, &S.gb, &S.gC, &S.gEta, &S.gJ, &S.bm, &S.bP, &S.bEta, &S.bJ};
  for (int i = 0; i < 9; ++i) *splanes[i] = reinterpret_cast<float*>(base + 9 * pb + i * sb);
  const int vs_row = vs_diag ? 0 : D;
  const bool unit = d.flags & EKS_FLAG_UNIT_AC;
  // K1 and K3 walk the chunks in opposite directions: the tail of one stream of y, var is the
  // head of the next and can be served from the 256 MiB Infinity Cache.  EKS_SUMMARIZE_REVERSE
  // picks which of the two runs backwards (A/B knob).
  const int k1_reverse = knob_int(KNOB_SUMMARIZE_REVERSE, 0) == 1 ? 1 : 0;
  const int k3_reverse = knob_set(KNOB_REPLAY_FORWARD) ? (knob_int(KNOB_REPLAY_FORWARD, 0) == 1 ? 0 : 1) : !k1_reverse;
  if (fused) {
    // (keypoint-tiled passes - the three launches per group of 64-chain tiles, so that a pass's rows are still in the
    //  Infinity Cache when its K3 asks for them - were built and measured in round 3: 264 -> 290-425 us on C3, K3 is no
    //  faster from the cache than from HBM and the group scan is paid per pass; the knob that selected them is gone, the
    //  loop below makes one pass)
    const bool rc_all = knob_set(KNOB_REPLAY_RECOMPUTE) ? knob_int(KNOB_REPLAY_RECOMPUTE, 0) == 1
                                                        : 5 * pb >= ((size_t)16 << 20);
    const int tp = L.ntile;
    for (int tile0 = 0; tile0 < L.ntile; tile0 += tp) {
    const int ntl = min(tp, L.ntile - tile0);
    const int n0 = tile0 * 64, Np = min(N - n0, ntl * 64);
    BlockMap Bm{N, T, L.nc, ntl, S.nblk, k1_reverse, tile0};
    const dim3 bgrid((unsigned)((long)Bm.ntile * Bm.ngrp)), bblock(64 * kFW);
    // Problems whose chunk elements do not stay on chip do not keep them between K1 and K3: K3
    // summarises its chunks again (diag_replay_blk_kernel, RC).  Same box, alternating runs
    // (EKS_REPLAY_RECOMPUTE = 0 / 1): C5's share (8192 chains x 1563 chunks, 256 MB of elements) K1
    // 0.64 -> 0.56 ms, K3 1.37 -> 1.33 ms, step 2.06 -> 1.94 ms; C3 (512 chains, 32 MB) K1 83 -> 75 us,
    // K3 168 -> 165 us, step 0.597 -> 0.591 ms; C2 (128 chains, 0.8 MB) K3 8.9 -> 13.3 us - one more
    // dependent pass in a latency-bound launch.  Hence the threshold on the element bytes.
    const bool rc = rc_all;
#define EKS_K1_BLK(UN)                                                                                  \
  do {                                                                                                  \
    if (rc)                                                                                             \
      hipLaunchKernelGGL((diag_summarize_blk_kernel<kChunk, UN, true>), bgrid, bblock, 0, st, Bm, M, W, \
                         S, y, var);                                                                    \
    else                                                                                                \
      hipLaunchKernelGGL((diag_summarize_blk_kernel<kChunk, UN, false>), bgrid, bblock, 0, st, Bm, M,   \
                         W, S, y, var);                                                                 \
  } while (0)
#define EKS_K3_BLK(UN, R)                                                                               \
  do {                                                                                                  \
    if (rc)                                                                                             \
      hipLaunchKernelGGL((diag_replay_blk_kernel<kChunk, UN, R, true>), bgrid, bblock, 0, st, Bm, M, W, \
                         S, y, var, ms, Vs);                                                            \
    else                                                                                                \
      hipLaunchKernelGGL((diag_replay_blk_kernel<kChunk, UN, R, false>), bgrid, bblock, 0, st, Bm, M,   \
                         W, S, y, var, ms, Vs);                                                         \
  } while (0)
    {
      ProfScope ps("diag_summarize", st);
      if (unit)
        EKS_K1_BLK(true);
      else
        EKS_K1_BLK(false);
    }
    {
      ProfScope ps("diag_scan", st);
      // a slot's aggregates stay in registers when there are at most 16 of them (T <= 131 072)
      // 4 chains per block for narrow problems (more blocks: 10 000 x 64 keypoints 7.1 vs 9.4 us),
      // 16 for wide ones (whole 64-byte segments: 50 000 x 4096 keypoints 119 -> 46 us)
      const bool ch4 = knob_set(KNOB_SCAN_CH) ? knob_int(KNOB_SCAN_CH, 4) == 4 : Np < 2048;
      const bool per8 = (S.nblk + 63) / 64 <= 8;
      const int Ne = n0 + Np;                 // chains [n0, Ne) are scanned by this launch
      if (ch4) {
        if (per8)
          hipLaunchKernelGGL((diag_scan_groups_kernel<8, 4>), dim3((Np + 3) / 4), dim3(256), 0, st, N, n0, Ne, M, S);
        else
          hipLaunchKernelGGL((diag_scan_groups_kernel<16, 4>), dim3((Np + 3) / 4), dim3(256), 0, st, N, n0, Ne, M, S);
      } else {
        if (per8)
          hipLaunchKernelGGL((diag_scan_groups_kernel<8, 16>), dim3((Np + 15) / 16), dim3(1024), 0, st, N, n0, Ne, M, S);
        else
          hipLaunchKernelGGL((diag_scan_groups_kernel<16, 16>), dim3((Np + 15) / 16), dim3(1024), 0, st, N, n0, Ne, M, S);
      }
    }
    ProfScope ps("diag_replay", st);
    Bm.reverse = k3_reverse;
#define EKS_REPLAY_BLK(R)                                                                              \
  case R:                                                                                              \
    if (unit)                                                                                          \
      EKS_K3_BLK(true, R);                                                                             \
    else                                                                                               \
      EKS_K3_BLK(false, R);                                                                            \
    break;
    switch (vs_row) {
      EKS_REPLAY_BLK(0)
      EKS_REPLAY_BLK(1)
      EKS_REPLAY_BLK(2)
      EKS_REPLAY_BLK(3)
      EKS_REPLAY_BLK(4)
      EKS_REPLAY_BLK(5)
      EKS_REPLAY_BLK(6)
      EKS_REPLAY_BLK(7)
      EKS_REPLAY_BLK(8)
    }
    }  // passes
#undef EKS_REPLAY_BLK
#undef EKS_K3_BLK
#undef EKS_K1_BLK
    return hip_status(hipGetLastError());
  }

  const int cpw = 64 >> L.nt_log2;
  const long waves = (long)L.ntile * ((L.nc + cpw - 1) / cpw);
  const dim3 grid((unsigned)((waves + 3) / 4));
  {
    ProfScope ps("diag_summarize", st);
    LaneMap L = Lf;
    L.reverse = k1_reverse;
    if (unit)
      hipLaunchKernelGGL((diag_summarize_kernel<kChunk, true>), grid, dim3(256), 0, st, L, M, W, y, var);
    else
      hipLaunchKernelGGL((diag_summarize_kernel<kChunk, false>), grid, dim3(256), 0, st, L, M, W, y, var);
  }
  {
    ProfScope ps("diag_scan", st);
    const dim3 sgrid((N + kScanCH - 1) / kScanCH, S.nblk), sblock(kScanCH * kScanCB);
    hipLaunchKernelGGL(diag_scan_reduce_kernel, sgrid, sblock, 0, st, N, L.nc, W, S);
    hipLaunchKernelGGL(diag_scan_blocks_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, M, S);
    hipLaunchKernelGGL(diag_scan_local_kernel, sgrid, sblock, 0, st, N, L.nc, W, S);
  }
  {
    ProfScope ps("diag_replay", st);
    LaneMap L = Lf;
    L.reverse = k3_reverse;
    if (unit)
      launch_replay<true>(vs_row, grid, st, L, M, W, y, var, ms, Vs);
    else
      launch_replay<false>(vs_row, grid, st, L, M, W, y, var, ms, Vs);
  }
  return hip_status(hipGetLastError());
}

}  // namespace eks

EKS_DEFINE_TOUCH(diag)
