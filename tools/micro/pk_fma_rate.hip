// Micro-benchmark: the NLL steady loop's dependency structure (NC recursions d = rho d + dy; s2 += d d per
// frame) with the candidates PAIRED in 64-bit registers so that each pair costs two v_pk_fma_f32 per frame
// instead of four v_fma_f32.  Question: how many cycles does a v_pk_fma_f32 wave-instruction occupy the SIMD,
// i.e. does a hand-packed loop (no shuffles: the pairs never leave their registers) beat the scalar one at the
// kernel's two waves per SIMD?   hipcc --offload-arch=gfx950 -O3 pk_fma_rate.hip -o pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int NC, bool PACKED>
__global__ void k(float* out, int iters, float seed, unsigned long long* stamps) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float t = 0.f;
  float dy = seed + threadIdx.x;
  if constexpr (PACKED) {
    f2 rho[NC / 2], d[NC / 2], s2[NC / 2];
#pragma unroll
    for (int c = 0; c < NC / 2; ++c) {
      rho[c] = f2{0.5f + 0.02f * c + seed, 0.51f + 0.02f * c + seed};
      d[c] = f2{seed * c, seed};
      s2[c] = f2{0.f, 0.f};
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        dy = dy * 1.0001f + 0.5f;
        const f2 dy2 = f2{dy, dy};
#pragma unroll
        for (int c = 0; c < NC / 2; ++c) {
          d[c] = __builtin_elementwise_fma(rho[c], d[c], dy2);
          s2[c] = __builtin_elementwise_fma(d[c], d[c], s2[c]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NC / 2; ++c) t += s2[c].x + s2[c].y + d[c].x + d[c].y;
  } else {
    float rho[NC], d[NC], s2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { rho[c] = 0.5f + 0.01f * c + seed; d[c] = seed * c; s2[c] = 0.f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        dy = dy * 1.0001f + 0.5f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          d[c] = rho[c] * d[c] + dy;
          s2[c] = s2[c] + d[c] * d[c];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) t += s2[c] + d[c];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
  if (stamps && threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int NC, bool PACKED>
void run(int waves_per_simd) {
  const int iters = 4000, threads = 256, blocks = 256 * waves_per_simd;
  float* out; (void)hipMalloc(&out, sizeof(float) * blocks * threads);
  unsigned long long* st; (void)hipMalloc(&st, 16 * blocks);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k<NC, PACKED>), dim3(blocks), dim3(threads), 0, 0, out, 10, 0.001f, nullptr);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<NC, PACKED>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.001f, st);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(2 * blocks);
  (void)hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
  const double ghz = 0.1 * (double)h[2 * (blocks / 2)] / (double)h[2 * (blocks / 2) + 1];
  const double fma_per_wave = (double)iters * 16 * 2.0 * NC;          // candidate FMAs (lane-wise), per wave
  const double ns_per_fma_per_simd = ms * 1e6 / (fma_per_wave * waves_per_simd);
  printf("%s NC=%d waves/SIMD=%d: %.3f ms, %.2f cycles per candidate-FMA per SIMD at the in-kernel clock %.2f GHz, %.1f TFLOP/s\n",
         PACKED ? "packed" : "scalar", NC, waves_per_simd, ms, ns_per_fma_per_simd * ghz, ghz,
         2.0 * 64 * fma_per_wave * blocks * 4 / (ms * 1e-3) / 1e12);
  (void)hipFree(st); (void)hipFree(out);
}

int main() {
  for (int w : {1, 2, 3, 4}) { run<8, false>(w); run<8, true>(w); }
  for (int w : {1, 2}) { run<16, false>(w); run<16, true>(w); }
  return 0;
}
