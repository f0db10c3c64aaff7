// Micro-benchmark: what v_fma_f32 issue rate does the NLL steady loop's dependency structure
// (NC recursions d = rho d + dy; s2 += d d) sustain at W waves per SIMD?  Prints cycles per
// VALU wave-instruction per SIMD.   hipcc --offload-arch=gfx950 -O3 fma_rate.hip -o fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NC>
__global__ void k(float* out, int iters, float seed, unsigned long long* stamps) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
  if (stamps && threadIdx.x == 0) {   // in-kernel clock: d memtime / d memrealtime x 100 MHz (MI355X_MICROARCH.md)
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int NC>
void run(int waves_per_simd) {
  const int iters = 4000;
  const int threads = 256;                        // 4 waves per block = 1 per SIMD
  const int blocks = 256 * waves_per_simd;        // blocks per CU = waves per SIMD
  float* out;
  (void)hipMalloc(&out, sizeof(float) * blocks * threads);
  unsigned long long* st;
  (void)hipMalloc(&st, 16 * blocks);
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL(k<NC>, dim3(blocks), dim3(threads), 0, 0, out, 10, 0.001f, nullptr);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<NC>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.001f, st);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(2 * blocks);
  (void)hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
  const double ghz = 0.1 * (double)h[2 * (blocks / 2)] / (double)h[2 * (blocks / 2) + 1];
  const double instr_per_wave = (double)iters * 16 * (2.0 * NC + 1);
  const double per_simd = instr_per_wave * waves_per_simd;
  printf("NC=%d waves/SIMD=%d: %.3f ms -> %.2f ns per VALU instr per SIMD (= %.2f cycles at 2.4 GHz), %.1f TFLOP/s; in-kernel clock %.2f GHz -> %.2f cycles\n",
         NC, waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4,
         2.0 * 64 * instr_per_wave * blocks * 4 / (ms * 1e-3) / 1e12, ghz, ms * 1e6 / per_simd * ghz);
  (void)hipFree(st);
  (void)hipFree(out);
}

int main() {
  for (int w : {1, 2, 3, 4, 6, 8}) run<8>(w);
  for (int w : {1, 2, 4, 8}) run<4>(w);
  for (int w : {1, 2, 4}) run<16>(w);
  return 0;
}
