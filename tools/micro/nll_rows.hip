// Micro-benchmark: does the NLL grid's steady loop wait for its rows when y is not on chip, and would a deeper,
// block-shared prefetch help?  Emulates diag_nll_summarize_kernel's steady state: a block of 8 waves owns one
// (64-chain tile, chunk of LEN rows); every wave runs NC = 8 recursions d = rho d + dy, s2 += d d over the SAME rows
// of y [T][N] (the 8 waves stand for the 8 candidate groups), 2 waves per SIMD, one block per CU.
//   regs8  : each wave loads its own rows, 8 in flight while 8 are consumed (what ships)
//   regs16 : the same with three 8-row buffers (rows requested 16 ahead)
//   dma    : every wave keeps its own LDS ring filled by direct-to-LDS loads, RING - 8 rows ahead, no barrier
//   ring   : the block's waves share an LDS ring of 2 x HALF rows: wave w requests the rows r = w (mod 8) of the next
//            half before consuming the current one from LDS, stores them when they arrive, one barrier per half
// Each variant is timed back to back (y stays in the 256 MB Infinity Cache) and with a 1 GB streaming kernel between
// launches (y comes from HBM).    hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize nll_rows.hip -o bin/nll_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int NC = 8;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ float row(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned i, unsigned row_bytes) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, i * row_bytes, 0));
}

struct State {
  float rho[NC], d[NC], s2[NC], yprev;
  __device__ __forceinline__ void init(int w, int lane) {
#pragma unroll
    for (int c = 0; c < NC; ++c) { rho[c] = 0.3f + 0.05f * c + 0.01f * w; d[c] = 0.f; s2[c] = 0.f; }
    yprev = 0.f;
  }
  __device__ __forceinline__ void eat(float y) {
    const float dy = y - yprev;
    yprev = y;
#pragma unroll
    for (int c = 0; c < NC; ++c) { d[c] = rho[c] * d[c] + dy; s2[c] += d[c] * d[c]; }
  }
  __device__ __forceinline__ float total() const {
    float t = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) t += s2[c];
    return t;
  }
};

template <int BUFS>   // 2: rows 8 ahead; 3: rows 16 ahead
__global__ __launch_bounds__(512) void regs(const float* __restrict__ y, int N, int ntile, int len, float* out) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = blockIdx.x % ntile, chunk = blockIdx.x / ntile;
  const __amdgpu_buffer_rsrc_t r = rsrc(y + (size_t)chunk * len * N + (size_t)tile * 64);
  const unsigned voff = lane * 4, rb = N * 4;
  State S; S.init(w, lane);
  float buf[BUFS][8];
  const int nb = len / 8;
#pragma unroll
  for (int b = 0; b < BUFS - 1; ++b)
#pragma unroll
    for (int q = 0; q < 8; ++q) buf[b][q] = row(r, voff, b * 8 + q, rb);
  for (int blk = 0; blk < nb; blk += BUFS) {
#pragma unroll
    for (int b = 0; b < BUFS; ++b) {
      const int nxt = blk + b + BUFS - 1;                    // the block requested now
      if (nxt < nb) {
#pragma unroll
        for (int q = 0; q < 8; ++q) buf[(b + BUFS - 1) % BUFS][q] = row(r, voff, nxt * 8 + q, rb);
      }
      if (blk + b < nb) {
#pragma unroll
        for (int q = 0; q < 8; ++q) S.eat(buf[b][q]);
      }
    }
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = S.total();
}

template <int HALF>
__global__ __launch_bounds__(512) void ring(const float* __restrict__ y, int N, int ntile, int len, float* out) {
  __shared__ float lds[2][HALF][64];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = blockIdx.x % ntile, chunk = blockIdx.x / ntile;
  const __amdgpu_buffer_rsrc_t r = rsrc(y + (size_t)chunk * len * N + (size_t)tile * 64);
  const unsigned voff = lane * 4, rb = N * 4;
  constexpr int PER = HALF / 8;                              // rows of a half per wave
  State S; S.init(w, lane);
  float in[PER];
  const int nh = len / HALF;
#pragma unroll
  for (int q = 0; q < PER; ++q) lds[0][w + 8 * q][lane] = row(r, voff, w + 8 * q, rb);
  __syncthreads();
  for (int h = 0; h < nh; ++h) {
    if (h + 1 < nh) {
#pragma unroll
      for (int q = 0; q < PER; ++q) in[q] = row(r, voff, (h + 1) * HALF + w + 8 * q, rb);
    }
#pragma unroll 4
    for (int i = 0; i < HALF; i += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = lds[h & 1][i + q][lane];
#pragma unroll
      for (int q = 0; q < 8; ++q) S.eat(v[q]);
    }
    if (h + 1 < nh) {
#pragma unroll
      for (int q = 0; q < PER; ++q) lds[(h + 1) & 1][w + 8 * q][lane] = in[q];
    }
    __syncthreads();
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = S.total();
}

// per-wave ring filled by direct-to-LDS loads (buffer_load_dword ... lds: no registers, no barrier): the rows are
// requested RING - 8 frames ahead; the wave waits with vmcnt(RING - 8) for the block it is about to read
template <int RING>
__global__ __launch_bounds__(512) void dma(const float* __restrict__ y, int N, int ntile, int len, float* out) {
  __shared__ float lds[8][RING][64];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = blockIdx.x % ntile, chunk = blockIdx.x / ntile;
  const __amdgpu_buffer_rsrc_t r = rsrc(y + (size_t)chunk * len * N + (size_t)tile * 64);
  const unsigned voff = lane * 4, rb = N * 4;
  State S; S.init(w, lane);
  const int nb = len / 8;
  auto issue = [&](int blk) {              // rows of block blk -> their ring slots
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = blk * 8 + q;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, &lds[w][i % RING][0], 4, voff, (unsigned)i * rb, 0, 0);
    }
  };
  constexpr int AHEAD = RING / 8 - 1;       // blocks in flight beyond the one being read
  for (int b = 0; b < AHEAD && b < nb; ++b) issue(b);
  for (int blk = 0; blk < nb; ++blk) {
    if (blk + AHEAD < nb) issue(blk + AHEAD);
    __builtin_amdgcn_s_waitcnt(0x0F70 | ((AHEAD * 8) & 0xF) | ((((AHEAD * 8) >> 4) & 0x3) << 14));   // vmcnt(AHEAD * 8)
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = lds[w][(blk * 8 + q) % RING][lane];
#pragma unroll
    for (int q = 0; q < 8; ++q) S.eat(v[q]);
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = S.total();
}

__global__ void stream(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.f;
}

template <typename F>
static void timed(const char* name, F launch, float* big, size_t nbig) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float t[2];
  for (int evict = 0; evict < 2; ++evict) {
    float sum = 0.f;
    for (int i = 0; i < 12; ++i) {
      if (evict) hipLaunchKernelGGL(stream, dim3(4096), dim3(256), 0, 0, big, nbig);
      (void)hipEventRecord(a, 0);
      launch();
      (void)hipEventRecord(b, 0);
      (void)hipEventSynchronize(b);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, a, b);
      if (i >= 2) sum += ms;
    }
    t[evict] = sum / 10;
  }
  printf("%-10s back to back %7.1f us   y from HBM %7.1f us\n", name, 1e3 * t[0], 1e3 * t[1]);
}

int main(int argc, char** argv) {
  const int N = 512, ntile = N / 64, nchunk = 32, len = 3072, T = nchunk * len;
  float *y, *out, *big;
  const size_t nbig = (size_t)256 << 20;
  (void)hipMalloc(&y, (size_t)T * N * 4);
  (void)hipMalloc(&out, (size_t)ntile * nchunk * 512 * 4);
  (void)hipMalloc(&big, nbig * 4);
  (void)hipMemset(y, 0, (size_t)T * N * 4);
  (void)hipMemset(big, 0, nbig * 4);
  printf("y [T=%d][N=%d] = %.1f MB, %d blocks of 8 waves, %d rows per wave\n", T, N, (double)T * N * 4 / 1e6,
         ntile * nchunk, len);
  const dim3 grid(ntile * nchunk), blk(512);
  timed("regs8", [&] { hipLaunchKernelGGL(regs<2>, grid, blk, 0, 0, y, N, ntile, len, out); }, big, nbig);
  timed("regs16", [&] { hipLaunchKernelGGL(regs<3>, grid, blk, 0, 0, y, N, ntile, len, out); }, big, nbig);
  timed("ring32", [&] { hipLaunchKernelGGL(ring<32>, grid, blk, 0, 0, y, N, ntile, len, out); }, big, nbig);
  timed("ring64", [&] { hipLaunchKernelGGL(ring<64>, grid, blk, 0, 0, y, N, ntile, len, out); }, big, nbig);
  timed("ring128", [&] { hipLaunchKernelGGL(ring<128>, grid, blk, 0, 0, y, N, ntile, len, out); }, big, nbig);
  timed("dma32", [&] { hipLaunchKernelGGL(dma<32>, grid, blk, 0, 0, y, N, ntile, len, out); }, big, nbig);
  timed("dma64", [&] { hipLaunchKernelGGL(dma<64>, grid, blk, 0, 0, y, N, ntile, len, out); }, big, nbig);
  (void)hipDeviceSynchronize();
  return 0;
}
