// Micro-benchmark of the LEAN role of diag_nll_grid_kernel on the C3 shape (100 000 frames x 512 chains x 64
// candidates): the real lane body (eks_nll_lane.hpp: nll_lean_chunk) with the kernel's row loads and LDS stash,
// without head role, stores or assembly.  Variants: candidates dealt contiguously or round-robin to the waves,
// chunk count (one round of 2 blocks per CU, or fewer / more), and whatever -D switches the header understands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -I ../../eks_amd/csrc nll_lean2.hip -o bin/nll_lean2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
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
                                            float* __restrict__ ob, double* __restrict__ oell, int* __restrict__ res) {
  __shared__ float stash[4][4 * NC][64];
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
  if (lane == 0) res[blockIdx.x * 4 + w] = ok;
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
  const int B0 = 1024;
  int BN = ((T - B0 + nch - 1) / nch + 15) / 16 * 16;
  const int ncn = 1 + (T - B0 + BN - 1) / BN;
  const int blocks = (N / 64) * (ncn - 1);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NC, RR>), dim3(blocks), dim3(256), 0, 0, y, N, T, B0, BN, ncn, rc, sc, ob, oell, res);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<NC, RR>), dim3(blocks), dim3(256), 0, 0, y, N, T, B0, BN, ncn, rc, sc, ob, oell, res);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  std::vector<int> h(blocks * 4);
  (void)hipMemcpy(h.data(), res, sizeof(int) * h.size(), hipMemcpyDeviceToHost);
  int c1 = 0, c2 = 0, c0 = 0;
  for (int v : h) { c0 += v == 0; c1 += v == 1; c2 += v == 2; }
  const double us = ms * 1e3 / reps, flops = 2.0 * 2.0 * (double)(T - B0) * N * 64;
  printf("%-44s BN=%5d %4d blocks: %7.1f us = %5.1f TFLOP/s useful  (waves: %d lean, %d lean A!=0, %d not qualified)\n", what, BN,
         blocks, us, flops / (us * 1e-6) / 1e12, c1, c2, c0);
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
