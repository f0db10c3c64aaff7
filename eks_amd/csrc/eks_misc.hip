// gfx950 kernels around the filter: constant-R by exact median over time (eks/core.py:702-709),
// argmin over the candidate grid, and the ensemble statistics stage (eks/core.py:25-101).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "eks_internal.hpp"

namespace eks {

// ==========================================================================================
// constant R: rconst[n] = max(nanmedian_t max(var[t][n], 1e-12), min_var)     (eks/core.py:702-709)
// Exact selection on the float bit patterns (positive floats order like their bits).
// ==========================================================================================
constexpr int kMedWaves = 16;  // waves per block of the two full passes

__device__ __forceinline__ uint32_t var_key(float v, bool& valid) {
  valid = !(v != v);
  const float c = v > 1e-12f ? v : 1e-12f;  // clip(var, 1e-12, inf), eks/utils.py:373
  return __float_as_uint(c);
}

// ------------------------------------------------------------------------------------------
// Fast path (T > kMedCap): two full passes instead of five.
//   B0 sample   : 256 evenly spaced rows per chain; the sample's order statistics 4 sigma either
//                 side of its median bracket the true median: [lo, hi] holds ~25 % of the frames.
//   B1 hist     : full pass; frames below lo are counted, frames inside [lo, hi] go to 256 linear
//                 bins of the KEY range (LDS histograms as above).
//   B2 narrow   : per chain, the bin(s) holding the two middle ranks -> [lo2, hi2] (~100 frames).
//   B3 collect  : full pass; frames inside [lo2, hi2] are appended to a per-chain list.
//   B4 finish   : exact selection of the middle ranks inside the list (rank by counting in LDS).
// A chain whose bracket misses the median, or whose bin holds more than kMedCap frames (heavy
// duplicates), is flagged and served by median_column_kernel: one block per flagged chain, an
// MSB-first radix select (4 x 8 bits + one sweep for the upper middle of even counts) over the
// chain's column - five strided sweeps, slow but exact, and a single (normally empty) launch.
// Short sequences (T <= kMedCap) are selected directly from the whole column.
// ------------------------------------------------------------------------------------------
constexpr int kMedCap = 1024;
constexpr int kMedFlight = 16;   // rows in flight per lane in the two full passes (measured best with 256 blocks)
constexpr int kMedSamples = 256;

struct BracketWs {
  uint32_t *lo, *hi, *less, *valid;     // [N]
  uint32_t* hist;                        // [256][N]
  uint32_t *lo2, *hi2, *less2, *cnt2;    // [N]
  uint32_t* list;                        // [N][kMedCap]
  uint32_t* fallback;                    // [N] 1 -> use the radix path for this chain
  uint32_t* any_fallback;                // [1]
};

// exact middle-rank selection inside `vals[0..L)` (LDS), ranks a <= b, by counting
__device__ __forceinline__ void select_two(const uint32_t* vals, int L, uint32_t a, uint32_t b,
                                           uint32_t* out_lo, uint32_t* out_hi) {
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    const uint32_t v = vals[i];
    uint32_t less = 0, eq = 0;
    for (int jj = 0; jj < L; ++jj) {
      const uint32_t u = vals[jj];
      less += u < v;
      eq += u == v;
    }
    if (a >= less && a < less + eq) *out_lo = v;
    if (b >= less && b < less + eq) *out_hi = v;
  }
}

__global__ __launch_bounds__(256) void median_small_kernel(int T, int N, const float* __restrict__ var,
                                                          double min_var, double* __restrict__ rconst) {
  __shared__ uint32_t vals[kMedCap];
  __shared__ uint32_t cnt, v_lo, v_hi;
  const int n = blockIdx.x;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    bool valid;
    const uint32_t key = var_key(var[(size_t)t * N + n], valid);
    if (valid) vals[atomicAdd(&cnt, 1u)] = key;
  }
  __syncthreads();
  const uint32_t c = cnt;
  if (c) select_two(vals, (int)c, (c - 1) / 2, c / 2, &v_lo, &v_hi);
  __syncthreads();
  if (threadIdx.x == 0) {
    double med = c ? 0.5 * (double)__uint_as_float(v_lo) + 0.5 * (double)__uint_as_float(v_hi) : nan("");
    rconst[n] = (med != med) ? med : (med > min_var ? med : min_var);
  }
}

__global__ __launch_bounds__(kMedSamples) void bracket_sample_kernel(int T, int N,
                                                                    const float* __restrict__ var,
                                                                    BracketWs B) {
  __shared__ uint32_t smp[kMedSamples];
  __shared__ uint32_t nvalid;
  const int n = blockIdx.x, i = threadIdx.x;
  if (i == 0) nvalid = 0;
  __syncthreads();
  bool valid;
  const uint32_t key = var_key(var[(size_t)(((long)i * T) / kMedSamples) * N + n], valid);
  smp[i] = valid ? key : 0xFFFFFFFFu;                      // NaNs sort last
  if (valid) atomicAdd(&nvalid, 1u);
  __syncthreads();
  const uint32_t nv = nvalid;
  if (i == 0) {
    B.less[n] = 0; B.valid[n] = 0; B.cnt2[n] = 0;
    B.fallback[n] = nv < 64 ? 1u : 0u;
    if (nv < 64) atomicOr(B.any_fallback, 1u);
  }
  if (nv < 64) return;
  // rank of my sample (stable for ties), then the bracket ranks 4 sigma (sigma = sqrt(nv)/2) out
  uint32_t rank = 0;
  for (int jj = 0; jj < kMedSamples; ++jj) rank += (smp[jj] < smp[i]) || (smp[jj] == smp[i] && jj < i);
  const int delta = (int)(2.0f * sqrtf((float)nv)) + 1;
  const int mid = ((int)nv - 1) / 2;
  const int r_lo = max(0, mid - delta), r_hi = min((int)nv - 1, mid + 1 + delta);
  if ((int)rank == r_lo) B.lo[n] = smp[i];
  if ((int)rank == r_hi) B.hi[n] = smp[i];
}

__global__ __launch_bounds__(64 * kMedWaves) void bracket_hist_kernel(int T, int N, int rows_per_block,
                                                                     const float* __restrict__ var,
                                                                     BracketWs B) {
  __shared__ uint32_t h[256][64];
  for (int i = threadIdx.x; i < 256 * 64; i += 64 * kMedWaves) (&h[0][0])[i] = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntile = (N + 63) / 64;
  const int tile = blockIdx.x % ntile, slab = blockIdx.x / ntile;
  const int n = tile * 64 + lane;
  const int t_begin = slab * rows_per_block;
  const int t_end = min(T, t_begin + rows_per_block);
  uint32_t less = 0, nvalid = 0;
  if (n < N && !B.fallback[n]) {
    const uint32_t lo = B.lo[n], hi = B.hi[n];
    const unsigned long long width = (unsigned long long)(hi - lo) + 1ull;
    for (int t = t_begin + wave; t < t_end; t += kMedFlight * kMedWaves) {
      float v[kMedFlight];
#pragma unroll
      for (int u = 0; u < kMedFlight; ++u) {
        const int tt = t + kMedWaves * u;
        v[u] = tt < t_end ? var[(size_t)tt * N + n] : __uint_as_float(0x7FC00000u);
      }
#pragma unroll
      for (int u = 0; u < kMedFlight; ++u) {
        bool valid;
        const uint32_t key = var_key(v[u], valid);
        if (!valid) continue;
        ++nvalid;
        if (key < lo) {
          ++less;
        } else if (key <= hi) {
          const uint32_t bin = (uint32_t)(((unsigned long long)(key - lo) * 256ull) / width);
          atomicAdd(&h[bin][lane], 1u);
        }
      }
    }
  }
  __syncthreads();
  if (n < N && !B.fallback[n]) {
    for (int b = wave; b < 256; b += kMedWaves) {
      const uint32_t c = h[b][lane];
      if (c) atomicAdd(&B.hist[(size_t)b * N + n], c);
    }
    if (less) atomicAdd(&B.less[n], less);
    if (nvalid) atomicAdd(&B.valid[n], nvalid);
  }
}

// one wave per chain: lane l owns bins 4l .. 4l+3, wave-level exclusive scan of the lane sums
__global__ __launch_bounds__(256) void bracket_narrow_kernel(int N, BracketWs B) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N || B.fallback[n]) return;                  // wave-uniform
  const uint32_t cnt = B.valid[n], less = B.less[n];
  const uint32_t lo = B.lo[n], hi = B.hi[n];
  const unsigned long long width = (unsigned long long)(hi - lo) + 1ull;
  const uint32_t r_lo = cnt ? (cnt - 1) / 2 : 0, r_hi = cnt / 2;
  uint32_t c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = B.hist[(size_t)(4 * lane + i) * N + n];
  const uint32_t mine = c[0] + c[1] + c[2] + c[3];
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off);
    if (lane >= off) incl += up;
  }
  uint32_t cum = less + incl - mine;                    // frames below my first bin
  int b_lo = -1, b_hi = -1;
  uint32_t less2 = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (r_lo >= cum && r_lo < cum + c[i]) {
      b_lo = 4 * lane + i;
      less2 = cum;
    }
    if (r_hi >= cum && r_hi < cum + c[i]) b_hi = 4 * lane + i;
    cum += c[i];
  }
  // at most one lane holds each of b_lo / b_hi: broadcast them
  const unsigned long long m_lo = __ballot(b_lo >= 0), m_hi = __ballot(b_hi >= 0);
  const bool ok = cnt != 0 && m_lo != 0 && m_hi != 0;
  int g_lo = 0, g_hi = 0;
  uint32_t g_less2 = 0;
  if (ok) {
    const int l_lo = __ffsll((long long)m_lo) - 1, l_hi = __ffsll((long long)m_hi) - 1;
    g_lo = __shfl(b_lo, l_lo);
    g_less2 = __shfl(less2, l_lo);
    g_hi = __shfl(b_hi, l_hi);
  }
  uint32_t inside = 0;                                  // frames inside bins g_lo .. g_hi
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int bb = 4 * lane + i;
    if (ok && bb >= g_lo && bb <= g_hi) inside += c[i];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) inside += __shfl_xor(inside, off);
  if (lane != 0) return;
  if (!ok || inside > (uint32_t)kMedCap) {
    B.fallback[n] = 1u;                                 // bracket missed the median / heavy duplicates
    atomicOr(B.any_fallback, 1u);
    return;
  }
  // key range of bins g_lo .. g_hi: bin b starts at lo + ceil(b * width / 256)
  B.lo2[n] = lo + (uint32_t)(((unsigned long long)g_lo * width + 255ull) / 256ull);
  B.hi2[n] = lo + (uint32_t)(((unsigned long long)(g_hi + 1) * width + 255ull) / 256ull) - 1u;
  B.less2[n] = g_less2;
}

__global__ __launch_bounds__(64 * kMedWaves) void bracket_collect_kernel(int T, int N, int rows_per_block,
                                                                        const float* __restrict__ var,
                                                                        BracketWs B) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntile = (N + 63) / 64;
  const int tile = blockIdx.x % ntile, slab = blockIdx.x / ntile;
  const int n = tile * 64 + lane;
  if (n >= N || B.fallback[n]) return;
  const int t_begin = slab * rows_per_block;
  const int t_end = min(T, t_begin + rows_per_block);
  const uint32_t lo2 = B.lo2[n], hi2 = B.hi2[n];
  uint32_t* list = B.list + (size_t)n * kMedCap;
  for (int t = t_begin + wave; t < t_end; t += kMedFlight * kMedWaves) {
    float v[kMedFlight];
#pragma unroll
    for (int u = 0; u < kMedFlight; ++u) {
      const int tt = t + kMedWaves * u;
      v[u] = tt < t_end ? var[(size_t)tt * N + n] : __uint_as_float(0x7FC00000u);
    }
#pragma unroll
    for (int u = 0; u < kMedFlight; ++u) {
      bool valid;
      const uint32_t key = var_key(v[u], valid);
      if (valid && key >= lo2 && key <= hi2) {
        const uint32_t pos = atomicAdd(&B.cnt2[n], 1u);
        if (pos < (uint32_t)kMedCap) list[pos] = key;
      }
    }
  }
}

__global__ __launch_bounds__(256) void bracket_finish_kernel(int N, double min_var, BracketWs B,
                                                            double* __restrict__ rconst) {
  __shared__ uint32_t vals[kMedCap];
  __shared__ uint32_t v_lo, v_hi;
  const int n = blockIdx.x;
  if (B.fallback[n]) return;                            // the radix path writes this chain
  const uint32_t L = min(B.cnt2[n], (uint32_t)kMedCap);
  for (int i = threadIdx.x; i < (int)L; i += blockDim.x) vals[i] = B.list[(size_t)n * kMedCap + i];
  __syncthreads();
  const uint32_t cnt = B.valid[n];
  select_two(vals, (int)L, (cnt - 1) / 2 - B.less2[n], cnt / 2 - B.less2[n], &v_lo, &v_hi);
  __syncthreads();
  if (threadIdx.x == 0) {
    const double med = 0.5 * (double)__uint_as_float(v_lo) + 0.5 * (double)__uint_as_float(v_hi);
    rconst[n] = med > min_var ? med : min_var;
  }
}

// exact median of one chain's column by radix select inside one block (fallback path)
__global__ __launch_bounds__(256) void median_column_kernel(int T, int N, const float* __restrict__ var,
                                                           double min_var, BracketWs B,
                                                           double* __restrict__ rconst) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh_prefix, sh_rank, sh_cnt, sh_less, sh_eq, sh_next;
  const int n = blockIdx.x;
  if (*B.any_fallback == 0u || B.fallback[n] == 0u) return;
  uint32_t prefix = 0, rank = 0, less = 0;
  for (int pass = 0; pass < 4; ++pass) {
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
      bool valid;
      const uint32_t key = var_key(var[(size_t)t * N + n], valid);
      if (!valid) continue;
      if (pass > 0 && (key >> (shift + 8)) != prefix) continue;
      atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (pass == 0) {
        uint32_t cnt = 0;
        for (int b = 0; b < 256; ++b) cnt += hist[b];
        sh_cnt = cnt;
        rank = cnt ? (cnt - 1) / 2 : 0;
      }
      uint32_t cum = 0;
      int bin = 255;
      for (int b = 0; b < 256; ++b) {
        if (cum + hist[b] > rank) {
          bin = b;
          break;
        }
        cum += hist[b];
      }
      less += cum;
      rank -= cum;
      prefix = (prefix << 8) | (uint32_t)bin;
      sh_prefix = prefix;
      sh_rank = rank;
      sh_less = less;
      sh_eq = hist[bin];
      sh_next = 0xFFFFFFFFu;
    }
    __syncthreads();
    prefix = sh_prefix;
    rank = sh_rank;
    less = sh_less;
  }
  const uint32_t cnt = sh_cnt, eq = sh_eq, key_lo = prefix;
  uint32_t best = 0xFFFFFFFFu;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    bool valid;
    const uint32_t key = var_key(var[(size_t)t * N + n], valid);
    if (valid && key > key_lo && key < best) best = key;
  }
  if (best != 0xFFFFFFFFu) atomicMin(&sh_next, best);
  __syncthreads();
  if (threadIdx.x == 0) {
    double med;
    if (cnt == 0) {
      med = nan("");                                    // np.nanmedian of an all-NaN slice
    } else {
      const double lo = (double)__uint_as_float(key_lo);
      const double hi = (cnt / 2 < less + eq) ? lo : (double)__uint_as_float(sh_next);
      med = 0.5 * lo + 0.5 * hi;
    }
    rconst[n] = (med != med) ? med : (med > min_var ? med : min_var);   // clip(nan) stays nan
  }
}

static inline size_t arr_bytes(size_t n) { return align_up(n * 4, 256); }

size_t const_r_workspace_bytes(int N) {
  return 10 * arr_bytes(N) + arr_bytes((size_t)256 * N) + arr_bytes((size_t)N * kMedCap) + 256;
}

int const_r(int T, int N, const float* var, double min_var, double* rconst, void* ws,
            size_t ws_bytes, hipStream_t st) {
  if (ws_bytes < const_r_workspace_bytes(N)) return EKS_ERR_WORKSPACE;
  ProfScope ps("const_r_select", st);
  if (T <= kMedCap) {
    hipLaunchKernelGGL(median_small_kernel, dim3(N), dim3(256), 0, st, T, N, var, min_var, rconst);
    return hip_status(hipGetLastError());
  }
  char* p = static_cast<char*>(ws);
  BracketWs B;
  uint32_t** arrs[10] = {&B.lo, &B.hi, &B.less, &B.valid, &B.lo2, &B.hi2, &B.less2, &B.cnt2,
                         &B.fallback, &B.any_fallback};
  for (int i = 0; i < 10; ++i) {
    *arrs[i] = reinterpret_cast<uint32_t*>(p);
    p += arr_bytes(N);
  }
  B.hist = reinterpret_cast<uint32_t*>(p);
  p += arr_bytes((size_t)256 * N);
  B.list = reinterpret_cast<uint32_t*>(p);

  hipError_t e = hipMemsetAsync(B.hist, 0, arr_bytes((size_t)256 * N), st);
  if (e == hipSuccess) e = hipMemsetAsync(B.any_fallback, 0, 4, st);
  if (e != hipSuccess) return hip_status(e);
  const int ntile = (N + 63) / 64;
  // one 16-wave block per CU, each long enough to amortise its LDS flush
  int rows = (int)(((long)T * ntile + 255) / 256);
  if (rows < kMedFlight * kMedWaves) rows = kMedFlight * kMedWaves;
  rows = (rows + kMedWaves - 1) / kMedWaves * kMedWaves;
  const int nslab = (T + rows - 1) / rows;
  const dim3 grid(ntile * nslab), big(64 * kMedWaves);
  hipLaunchKernelGGL(bracket_sample_kernel, dim3(N), dim3(kMedSamples), 0, st, T, N, var, B);
  hipLaunchKernelGGL(bracket_hist_kernel, grid, big, 0, st, T, N, rows, var, B);
  hipLaunchKernelGGL(bracket_narrow_kernel, dim3((N + 3) / 4), dim3(256), 0, st, N, B);
  hipLaunchKernelGGL(bracket_collect_kernel, grid, big, 0, st, T, N, rows, var, B);
  hipLaunchKernelGGL(bracket_finish_kernel, dim3(N), dim3(256), 0, st, N, min_var, B, rconst);
  hipLaunchKernelGGL(median_column_kernel, dim3(N), dim3(256), 0, st, T, N, var, min_var, B, rconst);
  return hip_status(hipGetLastError());
}

// ==========================================================================================
// argmin over candidates (first minimum, numpy.argmin semantics) + gather of s
// ==========================================================================================
// one wave per keypoint: lanes stride over the candidates, (value, index) min-reduction that
// keeps the FIRST minimum (numpy.argmin semantics)
__global__ __launch_bounds__(256) void argmin_kernel(int K, int n_cand, const double* __restrict__ nll,
                                                    const double* __restrict__ s_cand,
                                                    double* __restrict__ s_out,
                                                    int32_t* __restrict__ idx_out) {
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= K) return;
  const double* row = nll + (size_t)k * n_cand;
  constexpr int kNone = 0x7FFFFFFF;
  double bv = 0.0;
  int best = kNone;
  for (int c = lane; c < n_cand; c += 64) {
    const double v = row[c];
    if (best == kNone || v < bv) {
      bv = v;
      best = c;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(bv, off);
    const int oi = __shfl_xor(best, off);
    if (oi != kNone && (best == kNone || ov < bv || (ov == bv && oi < best))) {
      bv = ov;
      best = oi;
    }
  }
  if (lane == 0) {
    s_out[k] = s_cand[best];
    if (idx_out) idx_out[k] = best;
  }
}

int argmin_s(int K, int n_cand, const double* nll, const double* s_cand, double* s_out,
             int32_t* idx_out, hipStream_t st) {
  hipLaunchKernelGGL(argmin_kernel, dim3((K + 3) / 4), dim3(256), 0, st, K, n_cand, nll, s_cand,
                     s_out, idx_out);
  return hip_status(hipGetLastError());
}

// ==========================================================================================
// Adam on log s with the reference's stop rule (eks/core.py:652-681, :509-549)
// ==========================================================================================
__global__ void adam_step_kernel(int nb, const int32_t* __restrict__ offs,
                                 const int32_t* __restrict__ members, const double* __restrict__ nll,
                                 const double* __restrict__ dnll, double lr, double lo, double hi,
                                 double tol, int cap, double* __restrict__ state,
                                 double* __restrict__ s_keypoint, int32_t* __restrict__ n_active) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  double* st = state + (size_t)b * 6;
  double u = st[0], mom = st[1], vel = st[2], prev = st[3], iters = st[4], done = st[5];
  if (done == 0.0 && iters < (double)cap) {
    double L = 0.0, g = 0.0;
    for (int i = offs[b]; i < offs[b + 1]; ++i) {
      L += nll[members[i]];
      g += dnll[members[i]];
    }
    if (u < lo || u > hi) g = 0.0;
    g *= lr;
    const double cnt = iters + 1.0;
    mom = 0.9 * mom + 0.1 * g;
    vel = 0.999 * vel + 0.001 * g * g;
    const double mhat = mom / (1.0 - pow(0.9, cnt));
    const double vhat = vel / (1.0 - pow(0.999, cnt));
    u = u - mhat / (sqrt(vhat) + 1e-8);
    const bool stop = isfinite(prev) &&
                      fabs(L - prev) < tol * fabs(log(fmax(prev, 1e-12))) + 1e-6;
    prev = L;
    iters = cnt;
    done = stop ? 1.0 : 0.0;
    st[0] = u; st[1] = mom; st[2] = vel; st[3] = prev; st[4] = iters; st[5] = done;
  }
  const double s = exp(fmin(fmax(u, lo), hi));
  for (int i = offs[b]; i < offs[b + 1]; ++i) s_keypoint[members[i]] = s;
  if (done == 0.0 && iters < (double)cap) atomicAdd(n_active, 1);
}

int adam_step(int n_blocks, const int32_t* offs, const int32_t* members, const double* nll,
              const double* dnll, double lr, double lo, double hi, double tol, int cap,
              double* state, double* s_keypoint, int32_t* n_active, hipStream_t st) {
  hipError_t e = hipMemsetAsync(n_active, 0, sizeof(int32_t), st);
  if (e != hipSuccess) return hip_status(e);
  hipLaunchKernelGGL(adam_step_kernel, dim3((n_blocks + 127) / 128), dim3(128), 0, st, n_blocks, offs,
                     members, nll, dnll, lr, lo, hi, tol, cap, state, s_keypoint, n_active);
  return hip_status(hipGetLastError());
}

// ---- two-parameter optimiser of the pupil smoother (eks/ibl_pupil_smoother.py:560-604): one
// thread per chain.  state: {u_d, u_c, mom_d, mom_c, vel_d, vel_c, prev_loss, iters, done}.
// Emits the AR(1) dynamics of the next evaluation and the tangents d/du_d, d/du_c.
__global__ void pupil_adam_step_kernel(int n, const double* __restrict__ latent_var,
                                       const double* __restrict__ nll,
                                       const double* __restrict__ dnll, double lr, double tol, int cap,
                                       double* __restrict__ state, double* __restrict__ a,
                                       double* __restrict__ q, double* __restrict__ da,
                                       double* __restrict__ dq, int32_t* __restrict__ n_active) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double* st = state + (size_t)k * 9;
  double u[2] = {st[0], st[1]};
  double prev = st[6], iters = st[7], done = st[8];
  if (nll && done == 0.0 && iters < (double)cap) {
    const double L = nll[k], cnt = iters + 1.0;
    const double c1 = 1.0 - pow(0.9, cnt), c2 = 1.0 - pow(0.999, cnt);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const double g = dnll[(size_t)i * n + k];
      const double mom = 0.9 * st[2 + i] + 0.1 * g;
      const double vel = 0.999 * st[4 + i] + 0.001 * g * g;
      u[i] -= lr * (mom / c1) / (sqrt(vel / c2) + 1e-8);
      st[i] = u[i];
      st[2 + i] = mom;
      st[4 + i] = vel;
    }
    const bool stop = isfinite(prev) &&
                      fabs(L - prev) < tol * fabs(log(fmax(prev, 1e-12))) + 1e-6;
    prev = L;
    iters = cnt;
    done = stop ? 1.0 : 0.0;
    st[6] = prev; st[7] = iters; st[8] = done;
  }
  // s = sigmoid(u) (1 - 2 eps) + eps, eps = 1e-3 (:506-508); A = diag(s_d, s_c, s_c),
  // Q = diag(var (1 - s^2)) (:542-548)
  constexpr double eps = 1e-3;
  double s[2], ds[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double sig = 1.0 / (1.0 + exp(-u[i]));
    s[i] = sig * (1.0 - 2.0 * eps) + eps;
    ds[i] = sig * (1.0 - sig) * (1.0 - 2.0 * eps);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int i = j == 0 ? 0 : 1;
    const double lv = latent_var[(size_t)k * 3 + j];
    const size_t p = (size_t)k * 3 + j;
    a[p] = s[i];
    q[p] = lv * (1.0 - s[i] * s[i]);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      da[(size_t)t * n * 3 + p] = t == i ? ds[i] : 0.0;
      dq[(size_t)t * n * 3 + p] = t == i ? -2.0 * s[i] * ds[i] * lv : 0.0;
    }
  }
  if (done == 0.0 && iters < (double)cap) atomicAdd(n_active, 1);
}

int pupil_adam_step(int n, const double* latent_var, const double* nll, const double* dnll, double lr,
                    double tol, int cap, double* state, double* a, double* q, double* da, double* dq,
                    int32_t* n_active, hipStream_t st) {
  hipError_t e = hipMemsetAsync(n_active, 0, sizeof(int32_t), st);
  if (e != hipSuccess) return hip_status(e);
  hipLaunchKernelGGL(pupil_adam_step_kernel, dim3((n + 63) / 64), dim3(64), 0, st, n, latent_var, nll,
                     dnll, lr, tol, cap, state, a, q, da, dq, n_active);
  return hip_status(hipGetLastError());
}

// ==========================================================================================
// ensemble statistics, eks/core.py:58-85.  One lane per (camera, frame, keypoint); the M member
// values live in registers; median by a small sort with NaNs pushed to the end.
// ==========================================================================================
constexpr int kMaxModels = 16;

__device__ __forceinline__ float nan_to_num(float v, float nan_rep) {
  if (v != v) return nan_rep;
  if (v > FLT_MAX) return FLT_MAX;   // jnp.nan_to_num maps +-inf to the largest finite float
  if (v < -FLT_MAX) return -FLT_MAX;
  return v;
}

template <int MM>
__device__ __forceinline__ void member_stats(const float (&a)[MM], int M, bool median, float& avg,
                                             double& var) {
  // nan-aware mean / variance (ddof 0) in double; median via insertion sort of the valid values
  float s[MM];
  int cnt = 0;
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < MM; ++i) {
    const bool ok = i < M && !(a[i] != a[i]);
    s[i] = ok ? a[i] : INFINITY;
    if (ok) {
      ++cnt;
      sum += (double)a[i];
    }
  }
  if (cnt == 0) {
    avg = NAN;
    var = NAN;
    return;
  }
  const double mean = sum / cnt;
  double ss = 0.0;
#pragma unroll
  for (int i = 0; i < MM; ++i) {
    const bool ok = i < M && !(a[i] != a[i]);
    if (ok) {
      const double d = (double)a[i] - mean;
      ss += d * d;
    }
  }
  var = ss / cnt;
  if (!median) {
    avg = (float)mean;
    return;
  }
#pragma unroll
  for (int i = 1; i < MM; ++i) {
#pragma unroll
    for (int j = MM - 1; j >= 1; --j) {
      if (j <= i) {  // one bubble sweep of the prefix; static indices keep s[] in registers
        const float lo = fminf(s[j - 1], s[j]), hi = fmaxf(s[j - 1], s[j]);
        s[j - 1] = lo;
        s[j] = hi;
      }
    }
  }
  const int i_hi = cnt / 2, i_lo = (cnt - 1) / 2;
  float v_lo = 0.f, v_hi = 0.f;
#pragma unroll
  for (int i = 0; i < MM; ++i) {
    if (i == i_lo) v_lo = s[i];
    if (i == i_hi) v_hi = s[i];
  }
  avg = 0.5f * (v_lo + v_hi);
}

template <int MM>
__global__ __launch_bounds__(256) void ensemble_kernel(int M, long VTK, const float* __restrict__ mk,
                                                      int avg_mode, int var_mode, float nan_rep,
                                                      float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= VTK) return;
  float x[MM], y[MM];
  double lsum = 0.0;
#pragma unroll
  for (int m = 0; m < MM; ++m) {
    if (m < M) {
      const float* p = mk + ((size_t)m * VTK + i) * 3;
      x[m] = p[0];
      y[m] = p[1];
      lsum += (double)p[2];
    } else {
      x[m] = NAN;
      y[m] = NAN;
    }
  }
  const double conf = lsum / M;
  float ax, ay;
  double vx, vy;
  member_stats<MM>(x, M, avg_mode == 0, ax, vx);
  member_stats<MM>(y, M, avg_mode == 0, ay, vy);
  float fvx, fvy;
  if (M == 1) {
    fvx = fvy = (float)(1.0 / (conf > 1e-5 ? conf : 1e-5));
    if (conf != conf) fvx = fvy = NAN;
  } else if (var_mode == 0) {
    fvx = (float)(vx / conf);
    fvy = (float)(vy / conf);
  } else {
    fvx = (float)vx;
    fvy = (float)vy;
  }
  float* o = out + (size_t)i * 5;
  o[0] = ax;
  o[1] = ay;
  o[2] = nan_to_num(fvx, nan_rep);
  o[3] = nan_to_num(fvy, nan_rep);
  o[4] = (float)conf;
}

int ensemble_stats(int M, int V, int T, int K, const float* markers, int avg_mode, int var_mode,
                   float nan_replacement, float* stats, hipStream_t st) {
  if (M > kMaxModels) return EKS_ERR_UNSUPPORTED;
  const long VTK = (long)V * T * K;
  const dim3 grid((unsigned)((VTK + 255) / 256));
  if (M <= 8)
    hipLaunchKernelGGL(ensemble_kernel<8>, grid, dim3(256), 0, st, M, VTK, markers, avg_mode, var_mode,
                       nan_replacement, stats);
  else
    hipLaunchKernelGGL(ensemble_kernel<kMaxModels>, grid, dim3(256), 0, st, M, VTK, markers, avg_mode,
                       var_mode, nan_replacement, stats);
  return hip_status(hipGetLastError());
}

}  // namespace eks
