// Micro-benchmark: what v_fma_f32 issue rate does the NLL steady loop's dependency structure
// (NC recursions d = rho d + dy; s2 += d d) sustain at W waves per SIMD?  Prints cycles per
// VALU wave-instruction per SIMD.   hipcc --offload-arch=gfx950 -O3 fma_rate.hip -o fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NC>
__global__ void k(float* out, int iters, float seed) {
  float rho[NC], d[NC], s2[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) { rho[c] = 0.5f + 0.01f * c + seed; d[c] = seed * c; s2[c] = 0.f; }
  float dy = seed + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      dy = dy * 1.0001f + 0.5f;          // stands in for the load + subtract
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        d[c] = rho[c] * d[c] + dy;
        s2[c] = s2[c] + d[c] * d[c];
      }
    }
  }
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) t += s2[c] + d[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int NC>
void run(int waves_per_simd) {
  const int iters = 4000;
  const int threads = 256;                        // 4 waves per block = 1 per SIMD
  const int blocks = 256 * waves_per_simd;        // blocks per CU = waves per SIMD
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<NC>, dim3(blocks), dim3(threads), 0, 0, out, 10, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<NC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.001f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double instr_per_wave = (double)iters * 16 * (2.0 * NC + 1);
  const double per_simd = instr_per_wave * waves_per_simd;
  printf("NC=%d waves/SIMD=%d: %.3f ms -> %.2f ns per VALU instr per SIMD (= %.2f cycles at 2.4 GHz), %.1f TFLOP/s\n",
         NC, waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4,
         2.0 * 64 * instr_per_wave * blocks * 4 / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) run<8>(w);
  for (int w : {1, 2, 4}) run<4>(w);
  for (int w : {1, 2}) run<16>(w);
  return 0;
}
