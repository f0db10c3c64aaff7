// Micro-benchmark: HBM read rate of the chain kernels' access pattern against wider loads, and the
// FETCH_SIZE calibration MI355X_MICROARCH.md asks for ("other access widths are uncalibrated").
//   rows4   : the fused K1/K3 pattern - a wave reads 32 rows x 256 B (64 chains x 4 B), rows N x 4 B
//             apart, a block = 4 consecutive 32-row chunks of one 64-chain tile (tile-major grid)
//   rows16  : the same bytes per wave as 4 rows x 16 lanes x 16 B per load instruction (8 loads)
//   stream4 / stream16 : plain contiguous streams, 4 B and 16 B per lane
// Each reads the whole [T][N] float array once (T x N x 4 B, default 1 GiB so that the 256 MB
// Infinity Cache cannot serve it) and prints GB/s from HIP events.  Under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./load_width
// the per-kernel FETCH_SIZE against the known byte count calibrates the counter for each width.
//   hipcc --offload-arch=gfx950 -O3 load_width.hip -o load_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void rows4(const float* __restrict__ y, int N, int ntile, float* out) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = blockIdx.x % ntile, grp = blockIdx.x / ntile;
  const size_t first = (size_t)(grp * 4 + w) * 32 * N + (size_t)tile * 64;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + first), 0, 0x7FFFFFFF, 0x00020000);
  float v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i)
    v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, (unsigned)i * N * 4, 0));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += v[i];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void rows16(const float* __restrict__ y, int N, int ntile, float* out) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = blockIdx.x % ntile, grp = blockIdx.x / ntile;
  const size_t first = (size_t)(grp * 4 + w) * 32 * N + (size_t)tile * 64;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + first), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned voff = (unsigned)(lane >> 4) * N * 4 + (lane & 15) * 16;   // 4 rows per instruction
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  u4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, (unsigned)i * 4 * N * 4, 0);
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  if (s == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

// two arrays (y and var, as K1 reads them): ALT = loads alternate between the arrays, else all of
// the first array's rows, then the second's; W16 = 16 B per lane, 4 rows per instruction
template <bool ALT, bool W16>
__global__ __launch_bounds__(256) void rows_two(const float* __restrict__ y, const float* __restrict__ var, int N,
                                                int ntile, float* out) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = blockIdx.x % ntile, grp = blockIdx.x / ntile;
  const size_t first = (size_t)(grp * 4 + w) * 32 * N + (size_t)tile * 64;
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(y + first), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(var + first), 0, 0x7FFFFFFF, 0x00020000);
  unsigned s = 0;
  if constexpr (W16) {
    const unsigned voff = (unsigned)(lane >> 4) * N * 4 + (lane & 15) * 16;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 a[8], b[8];
    if constexpr (ALT) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a[i] = __builtin_amdgcn_raw_buffer_load_b128(r0, voff, (unsigned)i * 4 * N * 4, 0);
        b[i] = __builtin_amdgcn_raw_buffer_load_b128(r1, voff, (unsigned)i * 4 * N * 4, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_raw_buffer_load_b128(r0, voff, (unsigned)i * 4 * N * 4, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) b[i] = __builtin_amdgcn_raw_buffer_load_b128(r1, voff, (unsigned)i * 4 * N * 4, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x ^ a[i].y ^ a[i].z ^ a[i].w ^ b[i].x ^ b[i].y ^ b[i].z ^ b[i].w;
  } else {
    unsigned a[32], b[32];
    if constexpr (ALT) {
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        a[i] = __builtin_amdgcn_raw_buffer_load_b32(r0, lane * 4, (unsigned)i * N * 4, 0);
        b[i] = __builtin_amdgcn_raw_buffer_load_b32(r1, lane * 4, (unsigned)i * N * 4, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) a[i] = __builtin_amdgcn_raw_buffer_load_b32(r0, lane * 4, (unsigned)i * N * 4, 0);
#pragma unroll
      for (int i = 0; i < 32; ++i) b[i] = __builtin_amdgcn_raw_buffer_load_b32(r1, lane * 4, (unsigned)i * N * 4, 0);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) s += a[i] ^ b[i];
  }
  if (s == 0x12345678u) out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

__global__ __launch_bounds__(256) void stream4(const float* __restrict__ y, size_t n, float* out) {
  float s = 0.f;
  const size_t base = (size_t)blockIdx.x * 256 * 32 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += y[base + (size_t)i * 256];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void stream16(const f4* __restrict__ y, size_t n, float* out) {
  float s = 0.f;
  const size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f4 v = y[base + (size_t)i * 256];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static void timed(const char* name, double bytes, F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) launch();
  hipEventRecord(a, 0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("%-10s %8.1f us per pass  %7.0f GB/s\n", name, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 512;            // chains (row = N x 4 B)
  const int T = argc > 2 ? atoi(argv[2]) : 524288;         // frames (multiple of 128)
  const size_t n = (size_t)N * T;
  float *y, *out;
  hipMalloc(&y, n * 4);
  hipMalloc(&out, 1 << 20);
  hipMemset(y, 0, n * 4);
  const int ntile = N / 64, blocks = ntile * (T / 128);
  printf("[T=%d][N=%d] float, %.1f MB per pass, %d blocks of 256\n", T, N, n * 4 / 1e6, blocks);
  timed("rows4", n * 4.0, [&] { hipLaunchKernelGGL(rows4, dim3(blocks), dim3(256), 0, 0, y, N, ntile, out); });
  timed("rows16", n * 4.0, [&] { hipLaunchKernelGGL(rows16, dim3(blocks), dim3(256), 0, 0, y, N, ntile, out); });
  timed("stream4", n * 4.0, [&] { hipLaunchKernelGGL(stream4, dim3(n / (256 * 32)), dim3(256), 0, 0, y, n, out); });
  timed("stream16", n * 4.0, [&] { hipLaunchKernelGGL(stream16, dim3(n / (256 * 32)), dim3(256), 0, 0, (const f4*)y, n, out); });
  // two arrays of half the frames each: the same bytes per pass
  const int T2 = T / 2, blocks2 = ntile * (T2 / 128);
  const float* var = y + (size_t)N * T2;
  timed("2arr alt4", n * 4.0, [&] { hipLaunchKernelGGL((rows_two<true, false>), dim3(blocks2), dim3(256), 0, 0, y, var, N, ntile, out); });
  timed("2arr seq4", n * 4.0, [&] { hipLaunchKernelGGL((rows_two<false, false>), dim3(blocks2), dim3(256), 0, 0, y, var, N, ntile, out); });
  timed("2arr alt16", n * 4.0, [&] { hipLaunchKernelGGL((rows_two<true, true>), dim3(blocks2), dim3(256), 0, 0, y, var, N, ntile, out); });
  timed("2arr seq16", n * 4.0, [&] { hipLaunchKernelGGL((rows_two<false, true>), dim3(blocks2), dim3(256), 0, 0, y, var, N, ntile, out); });
  hipDeviceSynchronize();
  return 0;
}
