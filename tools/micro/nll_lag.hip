// Micro-benchmark of the LAG form of diag_nll_grid_kernel on the C3 shape (100 000 frames x 512 chains): the real lane
// body (eks_nll_lag.hpp: nll_lag_chunk) with the kernel's row loads, LDS stash and lag accumulators, without head role,
// classification, stores or assembly.  Variants: slow pairs per wave (NP), waves per block (NW), how the lag sets are
// dealt (masks), lag pairs (ND), chunk count.  Per-wave stamps give the wave durations and the shader clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -I ../../eks_amd/csrc nll_lag.hip -o bin/nll_lag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
#include "eks_nll_lag.hpp"
using namespace eks;

struct BufferRows {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff, row_bytes;
  __device__ __forceinline__ float operator()(int i) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (unsigned)i * row_bytes, 0));
  }
};
template <int ND>
struct LagLds {
  double* acc;
  float* sinkf;
  __device__ __forceinline__ void add(int k, float v) const { acc[k * 64] += (double)v; }
  __device__ __forceinline__ void head(int i, float v) const { sinkf[i] = v; }
  __device__ __forceinline__ void tail(int i, float v) const { sinkf[32 + i] = v; }
  __device__ __forceinline__ void ylast(float v) const { sinkf[64] = v; }
};

// NPA: pairs of waves 0, 1; NPB: pairs of waves 2, 3 (uneven deal); masks over `period` sets
template <int NPA, int NPB, int ND, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void k(const float* __restrict__ y, int N, int T, int B0, int BN, int ncn,
                                                     const double* __restrict__ rc, const double* __restrict__ sc,
                                                     float* __restrict__ ob, double* __restrict__ oell,
                                                     float* __restrict__ olag, unsigned long long* __restrict__ stamps,
                                                     int shareA, int shareB) {
  constexpr int NPM = NPA > NPB ? NPA : NPB;
  __shared__ double lacc[NW][2 * ND][64];
  __shared__ float stash[NW][3 * 2 * NPM][64];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ntile = N / 64;
  const int tile = blockIdx.x % ntile, j = 1 + blockIdx.x / ntile;
  if (j >= ncn) return;
  const int n = tile * 64 + lane;
  const int t0 = B0 + (j - 1) * BN, len = min(BN, T - t0) / 32 * 32;
  const BufferRows ld{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + (size_t)t0 * N + tile * 64), 0, 0x7FFFFFFF, 0x00020000),
                      (unsigned)(lane * 4), (unsigned)(N * 4)};
  const double r = rc[n];
#pragma unroll
  for (int i = 0; i < 2 * ND; ++i) lacc[w][i][lane] = 0.0;
  LagLds<ND> lags{&lacc[w][0][lane], olag + ((size_t)blockIdx.x * NW + w) * 80 * 64 + lane * 80};
  const int period = (NW / 2) * (shareA + shareB);
  auto body = [&](auto np_tag, unsigned mask) {
    constexpr int NP = decltype(np_tag)::value;
    auto sqf = [&](int c) { return sc[min(c * NW + w, 63)]; };          // slow candidates dealt round-robin, slowest first
    LeanOut<2 * NP> out;
    const int res = nll_lag_chunk<NP, ND, true>(ld, len, r, 1.0, 1.0, sqf, mask, period, w == 0, &stash[w][0][lane], 64, out, lags);
#pragma unroll
    for (int c = 0; c < 2 * NP; ++c) {
      const size_t off = ((size_t)j * 64 + w * 2 * NP + c) * N + n;
      ob[off] = out.B[c];
      oell[off] = out.Ell[c] + out.Eta[c] + res;
    }
  };
  // the block's waves partition a period of NW / 2 (shareA + shareB) sets: shareA consecutive sets to each wave of the
  // first half, shareB to each of the second
  const unsigned mask = w < NW / 2 ? ((1u << shareA) - 1u) << (w * shareA)
                                   : ((1u << shareB) - 1u) << ((NW / 2) * shareA + (w - NW / 2) * shareB);
  if (w < NW / 2 || NPA == NPB) body(IntTag<NPA>(), mask);
  else body(IntTag<NPB>(), mask);
  __syncthreads();
  if (w == 0) {
    double v = 0;
    for (int ww = 0; ww < NW; ++ww)
      for (int i = 0; i < 2 * ND; ++i) v += lacc[ww][i][lane];
    oell[(size_t)j * 64 * N + n] += v;
  }
  if (lane == 0) {
    unsigned long long* st = stamps + (size_t)(blockIdx.x * NW + w) * 4;
    st[0] = r0; st[1] = __builtin_amdgcn_s_memrealtime(); st[2] = __builtin_amdgcn_s_memtime() - c0;
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); st[3] = xcc;
  }
}

struct Bufs { float* y; double *rc, *sc, *oell; float *ob, *olag; unsigned long long* stamps; };

// shareA / shareB: lag sets per period for each wave of the first / second half of the block
template <int NPA, int NPB, int ND, int NW>
void run(const Bufs& b, int T, int N, int nch, int shareA, int shareB, const char* what) {
  const int B0 = 1024;
  int BN = ((T - B0 + nch - 1) / nch + 31) / 32 * 32;
  const int ncn = 1 + (T - B0 + BN - 1) / BN;
  const int blocks = (N / 64) * (ncn - 1);
  hipEvent_t a, e; (void)hipEventCreate(&a); (void)hipEventCreate(&e);
  auto launch = [&]() {
    hipLaunchKernelGGL((k<NPA, NPB, ND, NW>), dim3(blocks), dim3(64 * NW), 0, 0, b.y, N, T, B0, BN, ncn, b.rc, b.sc, b.ob, b.oell,
                       b.olag, b.stamps, shareA, shareB);
  };
  for (int r = 0; r < 3; ++r) launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e); (void)hipEventSynchronize(e);
  float ms; (void)hipEventElapsedTime(&ms, a, e);
  std::vector<unsigned long long> hs((size_t)blocks * NW * 4);
  (void)hipMemcpy(hs.data(), b.stamps, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost);
  std::vector<double> dur; double clk = 0;
  for (int i = 0; i < blocks * NW; ++i) { dur.push_back((hs[i * 4 + 1] - hs[i * 4]) * 0.01); clk += (double)hs[i * 4 + 2] / ((hs[i * 4 + 1] - hs[i * 4]) * 10.0); }
  std::sort(dur.begin(), dur.end());
  printf("%-58s BN=%5d %4d blocks x %d waves: %7.1f us   wave durations min %.1f / median %.1f / p90 %.1f / max %.1f us, clock %.2f GHz\n",
         what, BN, blocks, NW, ms * 1e3 / reps, dur.front(), dur[dur.size() / 2], dur[dur.size() * 9 / 10], dur.back(), clk / (blocks * NW));
}

int main() {
  const int T = 100000, N = 512;
  Bufs b;
  (void)hipMalloc(&b.y, sizeof(float) * (size_t)T * N);
  (void)hipMalloc(&b.rc, sizeof(double) * N);
  (void)hipMalloc(&b.sc, sizeof(double) * 64);
  (void)hipMalloc(&b.ob, sizeof(float) * (size_t)260 * 64 * N);
  (void)hipMalloc(&b.oell, sizeof(double) * (size_t)260 * 64 * N);
  (void)hipMalloc(&b.olag, sizeof(float) * (size_t)8 * 260 * 8 * 80 * 64);
  (void)hipMalloc(&b.stamps, sizeof(unsigned long long) * 4 * 8 * 8 * 260);
  std::vector<float> h((size_t)T * N);
  unsigned s = 12345;
  std::vector<float> x(N, 200.f);
  for (int t = 0; t < T; ++t)
    for (int n = 0; n < N; ++n) {
      s = s * 1664525u + 1013904223u; const float u1 = (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f;
      s = s * 1664525u + 1013904223u; const float u2 = (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f;
      x[n] += 1.5f * u1;
      h[(size_t)t * N + n] = x[n] + 2.2f * u2;
    }
  (void)hipMemcpy(b.y, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
  std::vector<double> hr(N, 0.42), hs(64);
  for (int c = 0; c < 64; ++c) hs[c] = exp(-8.0 + 16.0 * c / 63.0);
  (void)hipMemcpy(b.rc, hr.data(), sizeof(double) * N, hipMemcpyHostToDevice);
  (void)hipMemcpy(b.sc, hs.data(), sizeof(double) * 64, hipMemcpyHostToDevice);
  run<4, 4, 8, 4>(b, T, N, 60, 1, 1, "4 waves x NP 4, 16 lags, turns 1:1:1:1 (shipped)");
  run<4, 3, 8, 4>(b, T, N, 60, 1, 2, "4 waves, NP 4,4,3,3, 16 lags, turns 1:1:2:2");
  run<3, 3, 8, 4>(b, T, N, 60, 1, 1, "4 waves x NP 3 (24 slow), 16 lags");
  run<3, 3, 12, 4>(b, T, N, 60, 1, 1, "4 waves x NP 3 (24 slow), 24 lags");
  run<4, 4, 12, 4>(b, T, N, 60, 1, 1, "4 waves x NP 4, 24 lags");
  run<2, 2, 8, 4>(b, T, N, 60, 1, 1, "4 waves x NP 2, 16 lags");
  run<6, 6, 8, 4>(b, T, N, 60, 1, 1, "4 waves x NP 6, 16 lags");
  run<7, 7, 8, 2>(b, T, N, 60, 1, 1, "2 waves x NP 7 (28 slow), 16 lags, 60 chunks (1 wave / SIMD)");
  run<7, 7, 8, 2>(b, T, N, 124, 1, 1, "2 waves x NP 7 (28 slow), 16 lags, 124 chunks (2 waves / SIMD)");
  run<4, 4, 8, 4>(b, T, N, 124, 1, 1, "4 waves x NP 4, 16 lags, 124 chunks (two rounds)");
  return 0;
}
