// gfx950 kernels for the scalar-chain ("diagonal model") smoother: A, C, Q, S0 diagonal and D == O,
// so every keypoint coordinate is an independent scalar Kalman chain (singlecam: reference
// eks/singlecam_smoother.py:246-284 builds A = C = Q = I).  Replaces dynamax's sequential
// lax.scan pair (extended_kalman_smoother called at eks/core.py:290) with a three-kernel chunked
// associative scan so the time axis is parallel:
//
//   K1 diag_summarize : lane = (chain, chunk of B frames).  Reads y, var once, composes the chunk's
//                       element (A, b, C, eta, J) in registers, writes 20 B per chunk.
//   K2 diag_scan      : per chain, scans the chunk elements: forward -> predicted belief entering
//                       every chunk; backward -> information about the state just after every
//                       chunk from all later frames.  Two-level (segments through LDS).
//   K3 diag_replay    : lane = (chain, chunk).  Reads y, var again, replays the exact filter from
//                       the chunk's incoming belief keeping the B filtered (m, P) pairs IN
//                       REGISTERS, fuses the outgoing belief with the future information and runs
//                       RTS backwards over the registers, streaming ms / Vs out.
//
// HBM traffic: 2 x (y + var) in, 1 x (ms + Vs) out (+ ~7 % for the chunk elements); no filtered
// state ever touches memory.  No MFMA: the algebra is scalar.  No LDS in K1/K3: each datum is
// consumed by the lane that loads it; lanes of a wave are consecutive chains, so every row access
// is a contiguous 256 B (x4 for the four waves of a block: 1 KiB contiguous per frame).
#include <hip/hip_runtime.h>

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
  const ChainParams<float> p = load_chain_params(M, n);
  const int t0 = j * B;
  const int len = min(B, L.T - t0);
  const size_t o = (size_t)j * L.N + n;
  replay_chunk<B, UNIT, VS_ROW>(y, var, ms, Vs, L.N, n, n % M.D, t0, len, p, W.pm[o], W.pP[o],
                                W.sEta[o], W.sJ[o]);
}

// ------------------------------------------------------------------------------------------
// K2: scan of the chunk elements.  Block = CH chains x NSEG time segments (CH * NSEG threads).
// ------------------------------------------------------------------------------------------
constexpr int kScanCH = 16;
constexpr int kScanNSEG = 64;

__device__ __forceinline__ Elem<float> load_elem(const DiagWs& W, size_t o) {
  return Elem<float>{W.eA[o], W.eb[o], W.eC[o], W.eEta[o], W.eJ[o]};
}

__global__ __launch_bounds__(kScanCH* kScanNSEG) void diag_scan_kernel(int N, int nc, DiagModel M,
                                                                      DiagWs W) {
  __shared__ float sA[kScanNSEG][kScanCH], sb[kScanNSEG][kScanCH], sC[kScanNSEG][kScanCH],
      sEta[kScanNSEG][kScanCH], sJ[kScanNSEG][kScanCH];
  const int cl = threadIdx.x % kScanCH;
  const int seg = threadIdx.x / kScanCH;
  const int n = blockIdx.x * kScanCH + cl;
  const bool live = n < N;
  const int seglen = (nc + kScanNSEG - 1) / kScanNSEG;
  const int j0 = min(nc, seg * seglen);
  const int j1 = min(nc, j0 + seglen);

  // up-sweep: composite element of this segment
  Elem<float> acc = elem_identity<float>();
  if (live)
    for (int j = j0; j < j1; ++j) acc = elem_combine(acc, load_elem(W, (size_t)j * N + n));
  sA[seg][cl] = acc.A;
  sb[seg][cl] = acc.b;
  sC[seg][cl] = acc.C;
  sEta[seg][cl] = acc.eta;
  sJ[seg][cl] = acc.J;
  __syncthreads();
  if (!live) return;

  // belief entering this segment / information leaving it
  float m, P;
  load_chain_prior(M, n, m, P);
  for (int q = 0; q < seg; ++q) {
    const Elem<float> e{sA[q][cl], sb[q][cl], sC[q][cl], sEta[q][cl], sJ[q][cl]};
    elem_apply(e, m, P);
  }
  float eta = 0.f, J = 0.f;
  for (int q = kScanNSEG - 1; q > seg; --q) {
    const Elem<float> e{sA[q][cl], sb[q][cl], sC[q][cl], sEta[q][cl], sJ[q][cl]};
    elem_back(e, eta, J);
  }
  // down-sweeps
  for (int j = j0; j < j1; ++j) {
    const size_t o = (size_t)j * N + n;
    W.pm[o] = m;
    W.pP[o] = P;
    elem_apply(load_elem(W, o), m, P);
  }
  for (int j = j1 - 1; j >= j0; --j) {
    const size_t o = (size_t)j * N + n;
    W.sEta[o] = eta;
    W.sJ[o] = J;
    elem_back(load_elem(W, o), eta, J);
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
constexpr int kChunk = 32;  // frames per lane; 2*B VGPRs hold the chunk in K3

static inline size_t plane_bytes(int nc, int N) { return align_up((size_t)nc * N * sizeof(float), 256); }

size_t diag_smooth_workspace_bytes(int T, int N) {
  const int nc = (T + kChunk - 1) / kChunk;
  return 9 * plane_bytes(nc, N);
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
  const LaneMap L = make_lane_map(T, N, kChunk);
  const size_t pb = plane_bytes(L.nc, N);
  char* base = static_cast<char*>(ws);
  DiagWs W;
  float** planes[9] = {&W.eA, &W.eb, &W.eC, &W.eEta, &W.eJ, &W.pm, &W.pP, &W.sEta, &W.sJ};
  for (int i = 0; i < 9; ++i) *planes[i] = reinterpret_cast<float*>(base + i * pb);

  const int cpw = 64 >> L.nt_log2;
  const long waves = (long)L.ntile * ((L.nc + cpw - 1) / cpw);
  const dim3 grid((unsigned)((waves + 3) / 4));
  const bool unit = d.flags & EKS_FLAG_UNIT_AC;
  {
    ProfScope ps("diag_summarize", st);
    if (unit)
      hipLaunchKernelGGL((diag_summarize_kernel<kChunk, true>), grid, dim3(256), 0, st, L, M, W, y, var);
    else
      hipLaunchKernelGGL((diag_summarize_kernel<kChunk, false>), grid, dim3(256), 0, st, L, M, W, y, var);
  }
  {
    ProfScope ps("diag_scan", st);
    hipLaunchKernelGGL(diag_scan_kernel, dim3((N + kScanCH - 1) / kScanCH), dim3(kScanCH * kScanNSEG),
                       0, st, N, L.nc, M, W);
  }
  const int vs_row = vs_diag ? 0 : D;
  {
    ProfScope ps("diag_replay", st);
    if (unit)
      launch_replay<true>(vs_row, grid, st, L, M, W, y, var, ms, Vs);
    else
      launch_replay<false>(vs_row, grid, st, L, M, W, y, var, ms, Vs);
  }
  return hip_status(hipGetLastError());
}

}  // namespace eks
