// Micro-benchmark for a leaner NLL grid kernel: the steady two-FMA loop with row loads (buffer loads 16 frames
// ahead), the shared dy, candidates paired in 64-bit registers (v_pk_fma_f32) and the float64 flush every 32
// frames - NC candidates per lane, on the C3 shape (100 000 frames x 512 chains x 64 candidates):
//   NC = 8  : 8 waves per (tile, chunk), 32 chunks  -> 2 waves per SIMD   (the shipped kernel's geometry)
//   NC = 16 : 4 waves per (tile, chunk), 32 chunks  -> 1 wave per SIMD
//   NC = 16 : 4 waves per (tile, chunk), 64 chunks  -> 2 waves per SIMD
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize nll_lean.hip -o nll_lean
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NC>
__global__ __launch_bounds__(512) void k(const float* __restrict__ y, int T, int N, int nchunk, double* __restrict__ out) {
  const int wpb = blockDim.x >> 6;                       // waves per block = candidate groups
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int ntile = N / 64;
  const int tile = blockIdx.x % ntile, j = blockIdx.x / ntile;
  const int len = (T + nchunk - 1) / nchunk, t0 = j * len, t1 = min(T, t0 + len);
  const int n = tile * 64 + lane;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + (size_t)t0 * N + tile * 64), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned voff = lane * 4, rb = N * 4;
  auto ld = [&](int i) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (unsigned)i * rb, 0)); };
  f2 rho[NC / 2], d[NC / 2], s2[NC / 2];
  double acc[NC];
#pragma unroll
  for (int c = 0; c < NC / 2; ++c) {
    rho[c] = f2{0.30f + 0.004f * (w * NC + 2 * c), 0.302f + 0.004f * (w * NC + 2 * c)};
    d[c] = f2{0.f, 0.f};
    s2[c] = f2{0.f, 0.f};
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0.0;
  const int nfull = (t1 - t0) / 8;
  float ya[8], yb[8], yprev = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) ya[q] = ld(q);
  auto eat = [&](const float (&yy)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float dy = yy[q] - yprev;
      yprev = yy[q];
      const f2 dy2 = f2{dy, dy};
#pragma unroll
      for (int c = 0; c < NC / 2; ++c) {
        d[c] = rho[c] * d[c] + dy2;
        s2[c] = s2[c] + d[c] * d[c];
      }
    }
  };
  for (int blk = 0; blk + 2 <= nfull; blk += 2) {
#pragma unroll
    for (int q = 0; q < 8; ++q) yb[q] = ld((blk + 1) * 8 + q);
    eat(ya);
    if (blk + 2 < nfull) {
#pragma unroll
      for (int q = 0; q < 8; ++q) ya[q] = ld((blk + 2) * 8 + q);
    }
    eat(yb);
    if (blk & 2) {
#pragma unroll
      for (int c = 0; c < NC / 2; ++c) {
        acc[2 * c] += (double)s2[c].x;
        acc[2 * c + 1] += (double)s2[c].y;
        s2[c] = f2{0.f, 0.f};
      }
    }
  }
  double t = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) t += acc[c];
#pragma unroll
  for (int c = 0; c < NC / 2; ++c) t += s2[c].x + s2[c].y + d[c].x;
  out[((size_t)blockIdx.x * wpb + w) * 64 + lane] = t;
}

template <int NC>
void run(const float* y, int T, int N, int nchunk, double* out, const char* what) {
  const int wpb = 64 / NC, blocks = (N / 64) * nchunk;
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<NC>, dim3(blocks), dim3(64 * wpb), 0, 0, y, T, N, nchunk, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<NC>, dim3(blocks), dim3(64 * wpb), 0, 0, y, T, N, nchunk, out);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / reps, flops = 2.0 * 2.0 * (double)T * N * 64;
  printf("%-58s %4d blocks x %d waves: %7.1f us  = %.1f TFLOP/s useful\n", what, blocks, wpb, us, flops / (us * 1e-6) / 1e12);
}

int main() {
  const int T = 100000, N = 512;
  float* y; double* out;
  (void)hipMalloc(&y, sizeof(float) * (size_t)T * N);
  (void)hipMalloc(&out, 8 * 64 * 8 * 4096);
  std::vector<float> h((size_t)T * N);
  unsigned s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f); }
  (void)hipMemcpy(y, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
  run<8>(y, T, N, 32, out, "NC=8,  32 chunks (2 waves/SIMD, shipped geometry)");
  run<16>(y, T, N, 32, out, "NC=16, 32 chunks (1 wave/SIMD)");
  run<16>(y, T, N, 64, out, "NC=16, 64 chunks (2 waves/SIMD)");
  run<32>(y, T, N, 64, out, "NC=32, 64 chunks (2 waves/block: 1 wave/SIMD)");
  run<8>(y, T, N, 64, out, "NC=8,  64 chunks (4 waves/SIMD if registers allow)");
  return 0;
}
