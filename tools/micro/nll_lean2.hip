// Micro-benchmark of the LEAN role of diag_nll_grid_kernel on the C3 shape (100 000 frames x 512 chains x 64
// candidates): the real lane body (eks_nll_lane.hpp: nll_lean_chunk) with the kernel's row loads and LDS stash,
// without head role, stores or assembly.  Variants: candidates dealt contiguously or round-robin to the waves,
// chunk count (one round of 2 blocks per CU, or fewer / more), and whatever -D switches the header understands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -I ../../eks_amd/csrc nll_lean2.hip -o bin/nll_lean2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
#include "eks_nll_lane.hpp"
using namespace eks;

struct BufferRows {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff, row_bytes;
  __device__ __forceinline__ float operator()(int i) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (unsigned)i * row_bytes, 0));
  }
};

template <int NC, bool RR>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ y, int N, int T, int B0, int BN, int ncn,
                                            const double* __restrict__ rc, const double* __restrict__ sc,
                                            float* __restrict__ ob, double* __restrict__ oell, int* __restrict__ res,
                                            unsigned long long* __restrict__ stamps) {
  __shared__ float stash[4][4 * NC][64];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ntile = N / 64;
  const int tile = blockIdx.x % ntile, j = 1 + blockIdx.x / ntile;
  if (j >= ncn) return;
  const int n = tile * 64 + lane;
  const int t0 = B0 + (j - 1) * BN, len = min(BN, T - t0);
  const BufferRows ld{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + (size_t)t0 * N + tile * 64), 0, 0x7FFFFFFF, 0x00020000),
                      (unsigned)(lane * 4), (unsigned)(N * 4)};
  const double r = rc[n];
  const int ng = 64 / NC;
  auto sqf = [&](int c) { return sc[RR ? c * ng + w : w * NC + c]; };
  LeanOut<NC> out;
  const int ok = nll_lean_chunk<NC, true>(ld, t0, len, r, 1.0, 1.0, sqf, &stash[w][0][lane], 64, out);
  if (lane == 0) {
    res[blockIdx.x * 4 + w] = ok;
    unsigned long long* st = stamps + (size_t)(blockIdx.x * 4 + w) * 4;
    st[0] = r0; st[1] = __builtin_amdgcn_s_memrealtime(); st[2] = __builtin_amdgcn_s_memtime() - c0;
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); st[3] = xcc;
  }
  if (!ok) return;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const size_t off = ((size_t)j * 64 + w * NC + c) * N + n;
    ob[off] = out.B[c];
    oell[off] = out.Ell[c] + out.Eta[c];
  }
}

template <int NC, bool RR>
void run(const float* y, int T, int N, int nch, const double* rc, const double* sc, float* ob, double* oell, int* res,
         const char* what) {
  static unsigned long long* stamps = nullptr;
  if (!stamps) (void)hipMalloc(&stamps, sizeof(unsigned long long) * 4 * 4 * 8 * 140);
  const int B0 = 1024;
  int BN = ((T - B0 + nch - 1) / nch + 15) / 16 * 16;
  const int ncn = 1 + (T - B0 + BN - 1) / BN;
  const int blocks = (N / 64) * (ncn - 1);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NC, RR>), dim3(blocks), dim3(256), 0, 0, y, N, T, B0, BN, ncn, rc, sc, ob, oell, res, stamps);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<NC, RR>), dim3(blocks), dim3(256), 0, 0, y, N, T, B0, BN, ncn, rc, sc, ob, oell, res, stamps);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  std::vector<int> h(blocks * 4);
  (void)hipMemcpy(h.data(), res, sizeof(int) * h.size(), hipMemcpyDeviceToHost);
  int c1 = 0, c2 = 0, c0 = 0;
  for (int v : h) { c0 += v == 0; c1 += v == 1; c2 += v == 2; }
  std::vector<unsigned long long> hs((size_t)blocks * 16);
  (void)hipMemcpy(hs.data(), stamps, sizeof(unsigned long long) * hs.size(), hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull, tmax = 0; std::vector<double> dur; double clk = 0;
  for (int i = 0; i < blocks * 4; ++i) { tmin = std::min(tmin, hs[i * 4]); tmax = std::max(tmax, hs[i * 4 + 1]); dur.push_back((hs[i * 4 + 1] - hs[i * 4]) * 0.01); clk += (double)hs[i * 4 + 2] / ((hs[i * 4 + 1] - hs[i * 4]) * 10.0); }
  std::sort(dur.begin(), dur.end());
  double start_spread = 0; for (int i = 0; i < blocks * 4; ++i) start_spread = std::max(start_spread, (hs[i * 4] - tmin) * 0.01);
  const double us = ms * 1e3 / reps, flops = 2.0 * 2.0 * (double)(T - B0) * N * 64;
  printf("%-44s BN=%5d %4d blocks: %7.1f us = %5.1f TFLOP/s useful  (waves: %d lean, %d lean A!=0, %d not qualified)\n", what, BN,
         blocks, us, flops / (us * 1e-6) / 1e12, c1, c2, c0);
  {
    double sx[8] = {0}, sw[4] = {0}; int nx[8] = {0}, nw[4] = {0};
    for (int i = 0; i < blocks * 4; ++i) {
      const double d = (hs[i * 4 + 1] - hs[i * 4]) * 0.01;
      const int x = (int)(hs[i * 4 + 3] & 7);
      sx[x] += d; nx[x]++; sw[i & 3] += d; nw[i & 3]++;
    }
    printf("    mean wave duration per XCD:");
    for (int x = 0; x < 8; ++x) printf(" %.0f", nx[x] ? sx[x] / nx[x] : 0.0);
    printf(" us; per wave slot of the block:");
    for (int x = 0; x < 4; ++x) printf(" %.0f", sw[x] / nw[x]);
    // first half of the grid (first block on each CU) vs second half
    double h0 = 0, h1 = 0; int n0 = 0, n1 = 0;
    for (int i = 0; i < blocks * 4; ++i) { const double d = (hs[i * 4 + 1] - hs[i * 4]) * 0.01; if (i < blocks * 2) { h0 += d; n0++; } else { h1 += d; n1++; } }
    printf(" us; first / second half of the grid: %.0f / %.0f us\n", h0 / n0, h1 / n1);
  }
  printf("    last launch: first start -> last end %.1f us; wave durations min %.1f / median %.1f / p90 %.1f / max %.1f us; latest start +%.1f us; shader clock %.2f GHz\n",
         (tmax - tmin) * 0.01, dur.front(), dur[dur.size() / 2], dur[dur.size() * 9 / 10], dur.back(), start_spread, clk / (blocks * 4));
}

int main() {
  const int T = 100000, N = 512;
  float* y; double *rc, *sc, *oell; float* ob; int* res;
  (void)hipMalloc(&y, sizeof(float) * (size_t)T * N);
  (void)hipMalloc(&rc, sizeof(double) * N);
  (void)hipMalloc(&sc, sizeof(double) * 64);
  (void)hipMalloc(&ob, sizeof(float) * (size_t)140 * 64 * N);
  (void)hipMalloc(&oell, sizeof(double) * (size_t)140 * 64 * N);
  (void)hipMalloc(&res, sizeof(int) * 8 * 140 * 4);
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
  (void)hipMemcpy(y, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
  std::vector<double> hr(N, 0.42), hs(64);
  for (int c = 0; c < 64; ++c) hs[c] = exp(-8.0 + 16.0 * c / 63.0);
  (void)hipMemcpy(rc, hr.data(), sizeof(double) * N, hipMemcpyHostToDevice);
  (void)hipMemcpy(sc, hs.data(), sizeof(double) * 64, hipMemcpyHostToDevice);
  run<16, false>(y, T, N, 60, rc, sc, ob, oell, res, "NC=16 contiguous groups, 60 chunks");
  run<16, true>(y, T, N, 60, rc, sc, ob, oell, res, "NC=16 round-robin, 60 chunks");
  run<16, false>(y, T, N, 64, rc, sc, ob, oell, res, "NC=16 contiguous groups, 64 chunks");
  run<16, true>(y, T, N, 64, rc, sc, ob, oell, res, "NC=16 round-robin, 64 chunks");
  run<16, true>(y, T, N, 32, rc, sc, ob, oell, res, "NC=16 round-robin, 32 chunks");
  run<16, true>(y, T, N, 128, rc, sc, ob, oell, res, "NC=16 round-robin, 128 chunks");
  return 0;
}
