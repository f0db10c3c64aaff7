// gfx950 kernels around the filter: constant-R by exact median over time (eks/core.py:702-709),
// argmin over the candidate grid, and the ensemble statistics stage (eks/core.py:25-101).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdlib>

#include "eks_adam.hpp"
#include "eks_internal.hpp"
#include "eks_np_sum.hpp"

namespace eks {

// ==========================================================================================
// constant R: rconst[n] = max(nanmedian_t max(var[t][n], 1e-12), min_var)     (eks/core.py:702-709)
// Exact selection on the float bit patterns (positive floats order like their bits).
// ==========================================================================================

__device__ __forceinline__ uint32_t var_key(float v, bool& valid) {
  valid = !(v != v);
  const float c = v > 1e-12f ? v : 1e-12f;  // clip(var, 1e-12, inf), eks/utils.py:373
  return __float_as_uint(c);
}

// ------------------------------------------------------------------------------------------
// Fast path (T > kMedSmall): ONE full pass over var.
//   S1 sample  : ~4096 evenly spaced rows, transposed through LDS into per-chain sample columns
//                (coalesced both ways; 4 % of the data).
//   S2 bracket : per chain, the sample's order statistics 4.5 sigma either side of its median
//                (two histogram passes in LDS) bracket the true median: [lo, hi] holds ~6 % of the frames.
//   S3 collect : the full pass.  Lanes = chains, 16 or 32 rows per wave in flights of 16 loads per
//                lane; frames below lo and valid frames are counted in registers, frames inside
//                [lo, hi] are staged per chain and wave in LDS (branch-free, register slot
//                counters) and appended to the chain's list as one contiguous run per block
//                (one global atomic per chain and block).
//   S4 finish  : per chain, exact selection of the two middle ranks inside the list (radix select
//                in LDS).
// A chain whose bracket misses the median, or whose bracket holds more than kMedList frames
// (heavy duplicates), is served by its S4 block from the whole column instead (median_column:
// MSB-first radix select, 4 x 8 bits + one sweep for the upper middle of even counts - five
// strided sweeps, slow but exact and rare).
// Short sequences (T <= kMedSmall) are selected directly from the whole column.
// (Round-1 history: 5 radix passes 1.40 ms -> sample / histogram / collect, two full passes,
//  0.20 ms -> this.)
// ------------------------------------------------------------------------------------------
constexpr int kMedSmall = 1024;      // T up to here: one block per chain selects from the column
constexpr int kMedSamples = 4096;    // sample rows per chain (fewer for short sequences)
constexpr int kMedList = 16384;      // capacity of a chain's in-bracket list
#ifndef EKS_COL_FLIGHT
#define EKS_COL_FLIGHT 16
#endif
#ifndef EKS_COL_SLOTS
#define EKS_COL_SLOTS 16
#endif
constexpr int kColFlight = EKS_COL_FLIGHT;       // loads in flight per lane; rows per wave (a launch parameter): a multiple of it
constexpr int kColWaves = 8;         // waves per block of the full pass (same 64 chains)

struct BracketWs {
  uint32_t *lo, *hi, *less, *valid, *cnt;   // [N]
  uint32_t* smp;                             // [N][S]
  uint32_t* list;                            // [N][cap]
  uint32_t cap;                              // list capacity per chain (list_capacity(T) >= kMedList)
  uint32_t* fallback;                        // [N] 1 -> too few valid samples: select from the column
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
  __shared__ uint32_t vals[kMedSmall];
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


// k-th smallest (0-based) of vals[0..L) in LDS by MSB-first radix select; the whole block (256
// threads) takes part.  Keys are first mapped to (key - base) << lsh so that the 8 bits of the
// first pass spread the keys over all 256 bins (the keys of one chain share their top bits, and
// same-address LDS atomics serialise); 0xFFFFFFFF (invalid) stays last.  hist: 256 counters,
// sh: 2 words of LDS scratch.
__device__ __forceinline__ uint32_t radix_norm(uint32_t key, uint32_t base, int lsh) {
  return key == 0xFFFFFFFFu ? key : (key - base) << lsh;
}

__device__ uint32_t lds_radix_select(const uint32_t* vals, int L, uint32_t rank, uint32_t base, int lsh,
                                     uint32_t* hist, uint32_t* sh) {
  uint32_t prefix = 0;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[threadIdx.x] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += 256) {
      const uint32_t key = radix_norm(vals[i], base, lsh);
      if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {       // wave 0: lane l owns bins 4l..4l+3, exclusive scan of lane sums
      const int lane = threadIdx.x;
      uint32_t c[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) c[q] = hist[4 * lane + q];
      const uint32_t mine = c[0] + c[1] + c[2] + c[3];
      uint32_t incl = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
      }
      uint32_t cum = incl - mine;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (rank >= cum && rank < cum + c[q]) {
          sh[0] = (prefix << 8) | (uint32_t)(4 * lane + q);
          sh[1] = rank - cum;
        }
        cum += c[q];
      }
    }
    __syncthreads();
    prefix = sh[0];
    rank = sh[1];
    __syncthreads();
  }
  return (prefix >> lsh) + base;
}

// Same selection, but after the first histogram pass the keys of the rank's bin (about L / 256 of
// them once the keys are range-normalised) are compacted into `small` and the rank is resolved
// there by counting - one more sweep over the list instead of three.  Also returns, when it lies
// in the same bin, the next order statistic (rank + 1) through *next (has_next says whether it
// did).  Bins with more than kSelSmall keys (heavy duplicates) take the remaining radix passes.
// hist: 256 counters, sh: 6 words, small: kSelSmall words of LDS.
constexpr int kSelSmall = 256;

template <int NT>
__device__ uint32_t lds_select_compact(const uint32_t* vals, int L, uint32_t rank, uint32_t base, int lsh,
                                       uint32_t* hist, uint32_t* sh, uint32_t* small, uint32_t* next,
                                       bool* has_next) {
  if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
  if (threadIdx.x == 0) sh[3] = 0u;
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += NT) atomicAdd(&hist[radix_norm(vals[i], base, lsh) >> 24], 1u);
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    uint32_t c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = hist[4 * lane + q];
    const uint32_t mine = c[0] + c[1] + c[2] + c[3];
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    uint32_t cum = incl - mine;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (rank >= cum && rank < cum + c[q]) {
        sh[0] = (uint32_t)(4 * lane + q);     // the bin
        sh[1] = rank - cum;                   // rank inside the bin
        sh[2] = c[q];                         // keys in the bin
      }
      cum += c[q];
    }
  }
  __syncthreads();
  const uint32_t bin = sh[0], q_in = sh[1], m = sh[2];
  __syncthreads();
  if (m > (uint32_t)kSelSmall) {              // heavy duplicates: finish with the radix passes
    *has_next = false;
    uint32_t prefix = bin;
    uint32_t r = q_in;
    for (int pass = 1; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
      __syncthreads();
      for (int i = threadIdx.x; i < L; i += NT) {
        const uint32_t key = radix_norm(vals[i], base, lsh);
        if ((key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        uint32_t c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = hist[4 * lane + q];
        const uint32_t mine = c[0] + c[1] + c[2] + c[3];
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const uint32_t up = __shfl_up(incl, off);
          if (lane >= off) incl += up;
        }
        uint32_t cum = incl - mine;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (r >= cum && r < cum + c[q]) {
            sh[0] = (prefix << 8) | (uint32_t)(4 * lane + q);
            sh[1] = r - cum;
          }
          cum += c[q];
        }
      }
      __syncthreads();
      prefix = sh[0];
      r = sh[1];
      __syncthreads();
    }
    return (prefix >> lsh) + base;
  }
  for (int i = threadIdx.x; i < L; i += NT) {
    const uint32_t key = radix_norm(vals[i], base, lsh);
    if ((key >> 24) == bin) small[atomicAdd(&sh[3], 1u)] = key;
  }
  __syncthreads();
  const bool want_next = q_in + 1u < m;
  if (threadIdx.x < (int)m) {                 // m <= 256: one key per thread, counted against all
    const uint32_t v = small[threadIdx.x];
    uint32_t less = 0, eq = 0;
    for (uint32_t j = 0; j < m; ++j) {
      const uint32_t o = small[j];
      less += o < v;
      eq += o == v;
    }
    if (less <= q_in && q_in < less + eq) sh[4] = v;                       // equal keys write the same value
    if (want_next && less <= q_in + 1u && q_in + 1u < less + eq) sh[5] = v;
  }
  __syncthreads();
  *has_next = want_next;
  if (want_next) *next = (sh[5] >> lsh) + base;
  return (sh[4] >> lsh) + base;
}

// Bracket two ranks r_lo <= r_hi of vals[0..L) at 16-bit resolution of the normalised keys: two
// histogram passes (the second one refines the two first-pass bins at once) instead of two exact
// radix selects.  Returns the low edge of r_lo's cell and the high edge of r_hi's: every key of
// rank r_lo..r_hi lies inside.  hist: 512 counters, sh: 4 words.
template <int NT>
__device__ void lds_bracket(const uint32_t* vals, int L, uint32_t r_lo, uint32_t r_hi, uint32_t base,
                            int lsh, uint32_t* hist, uint32_t* sh, uint32_t& lo, uint32_t& hi) {
  // wave 0 / wave 1 locate rank r in hist[off .. off + 256): bin -> sh[slot], rank inside -> sh[slot + 1]
  auto find = [&](uint32_t rank, int off, int slot) {
    const int lane = threadIdx.x & 63;
    uint32_t c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = hist[off + 4 * lane + q];
    const uint32_t mine = c[0] + c[1] + c[2] + c[3];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    uint32_t cum = incl - mine;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (rank >= cum && rank < cum + c[q]) {
        sh[slot] = (uint32_t)(4 * lane + q);
        sh[slot + 1] = rank - cum;
      }
      cum += c[q];
    }
  };
  if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
  if (threadIdx.x < 256) hist[256 + threadIdx.x] = 0u;
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += NT) atomicAdd(&hist[radix_norm(vals[i], base, lsh) >> 24], 1u);
  __syncthreads();
  if (threadIdx.x < 64) find(r_lo, 0, 0);
  else if (threadIdx.x < 128) find(r_hi, 0, 2);
  __syncthreads();
  const uint32_t b_lo = sh[0], q_lo = sh[1], b_hi = sh[2], q_hi = sh[3];
  __syncthreads();
  if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
  if (threadIdx.x < 256) hist[256 + threadIdx.x] = 0u;
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += NT) {
    const uint32_t k = radix_norm(vals[i], base, lsh);
    if ((k >> 24) == b_lo) atomicAdd(&hist[(k >> 16) & 255u], 1u);
    if ((k >> 24) == b_hi) atomicAdd(&hist[256 + ((k >> 16) & 255u)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 64) find(q_lo, 0, 0);
  else if (threadIdx.x < 128) find(q_hi, 256, 2);
  __syncthreads();
  const uint32_t c_lo = (b_lo << 8) | sh[0], c_hi = (b_hi << 8) | sh[2];   // 16-bit cells
  lo = ((c_lo << 16) >> lsh) + base;
  hi = (((c_hi << 16) | 0xFFFFu) >> lsh) + base;
  __syncthreads();
}

// min / max over the valid keys of vals[0..L) (block-wide, through LDS scratch mm[2])
template <int NT>
__device__ void lds_key_range(const uint32_t* vals, int L, uint32_t* mm, uint32_t& lo, uint32_t& hi) {
  if (threadIdx.x == 0) {
    mm[0] = 0xFFFFFFFFu;
    mm[1] = 0u;
  }
  __syncthreads();
  uint32_t mn = 0xFFFFFFFFu, mx = 0u;
  for (int i = threadIdx.x; i < L; i += NT) {
    const uint32_t k = vals[i];
    if (k != 0xFFFFFFFFu) {
      mn = min(mn, k);
      mx = max(mx, k);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
    mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&mm[0], mn);
    atomicMax(&mm[1], mx);
  }
  __syncthreads();
  lo = mm[0];
  hi = mm[1];
  __syncthreads();
}

static inline int sample_count(int T) { return T / 2 < kMedSamples ? T / 2 : kMedSamples; }

// S1: sample rows i -> floor(i T / S), 64 chains x 64 samples per block, transposed through LDS
__global__ __launch_bounds__(256) void sample_transpose_kernel(int T, int N, int S,
                                                              const float* __restrict__ var,
                                                              BracketWs B) {
  __shared__ uint32_t tile[64][65];
  const int n0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // all 16 rows of the wave requested before the first is used (the rows are ~T / S frames apart:
  // one dependent HBM latency each if the loop is left rolled - 14.7 us for 17 MB of traffic)
  float v[16];
  const int n = n0 + lane;
  // sample i is row floor(i T / S); S is kMedSamples = 2^12 for every T >= 8 192: a shift instead of sixteen emulated
  // 64-bit divisions per thread (a third of this launch)
  const int sshift = (S & (S - 1)) == 0 ? __builtin_ctz((unsigned)S) : -1;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + w + 4 * q;
    const long row = sshift >= 0 ? ((long)i * T) >> sshift : ((long)i * T) / S;
    v[q] = (i < S && n < N) ? var[(size_t)row * N + n] : __uint_as_float(0x7FC00000u);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    bool valid;
    const uint32_t k = var_key(v[q], valid);
    tile[w + 4 * q][lane] = valid ? k : 0xFFFFFFFFu;         // NaNs (and padding) sort last
  }
  __syncthreads();
  for (int c = w; c < 64; c += 4) {
    const int n = n0 + c, i = i0 + lane;
    if (n < N && i < S) B.smp[(size_t)n * S + i] = tile[lane][c];
  }
}

// S2: one block per chain
template <int NT>
__global__ __launch_bounds__(NT) void sample_bracket_kernel(int N, int S, BracketWs B) {
  __shared__ uint32_t vals[kMedSamples];
  __shared__ uint32_t hist[512];
  __shared__ uint32_t sh[4], nvalid;
  const int n = blockIdx.x;
  if (threadIdx.x == 0) nvalid = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (int i = threadIdx.x; i < S; i += NT) {
    const uint32_t k = B.smp[(size_t)n * S + i];
    vals[i] = k;
    mine += k != 0xFFFFFFFFu;
  }
  if (mine) atomicAdd(&nvalid, mine);
  __syncthreads();
  const uint32_t nv = nvalid;
  if (threadIdx.x == 0) {
    B.less[n] = 0; B.valid[n] = 0; B.cnt[n] = 0;
    B.fallback[n] = nv < 64 ? 1u : 0u;
  }
  if (nv < 64) return;
  // bracket ranks 4.5 sigma out (sigma of the sample median's rank = sqrt(nv) / 2)
  const int delta = (int)(2.25f * sqrtf((float)nv)) + 1;
  const int mid = ((int)nv - 1) / 2;
  const int r_lo = max(0, mid - delta), r_hi = min((int)nv - 1, mid + 1 + delta);
  uint32_t kmin, kmax;
  lds_key_range<NT>(vals, S, sh, kmin, kmax);
  const int lsh = kmax > kmin ? __clz((int)(kmax - kmin)) : 0;
  // the bracket only has to CONTAIN the sample's order statistics r_lo .. r_hi: 1/65536 of the
  // sample's key range is resolution enough (the exact selection happens in S4)
  uint32_t lo, hi;
  lds_bracket<NT>(vals, S, (uint32_t)r_lo, (uint32_t)r_hi, kmin, lsh, hist, sh, lo, hi);
  if (threadIdx.x == 0) {
    B.lo[n] = lo;
    B.hi[n] = min(hi, kmax);
  }
}

// S3: the full pass.  Block = 8 waves over the SAME 64 chains (8 consecutive R-row slabs, flights
// of 16 loads per lane; R: collect_rows_per_wave).  Every wave stages its in-bracket keys per chain in its OWN LDS
// slots with the slot counter in a register - no atomics and no branches on the per-row path.
// After the barrier the runs of the 8 waves are written out back to back, one global atomic per
// chain and block.
constexpr int kColSlots = EKS_COL_SLOTS;   // expected 32 rows x ~7 % = 2.3 in-bracket keys per lane and wave
static_assert(kColWaves == 8, "the run write-out maps lanes to (8 source waves) x (8 slots)");

__global__ __launch_bounds__(64 * kColWaves, 8) void bracket_collect_kernel(int T, int N, int R,
                                                                        const float* __restrict__ var,
                                                                        BracketWs B) {
  __shared__ uint32_t stage[kColWaves][kColSlots][65];   // [wave][slot][chain], padded
  __shared__ uint32_t counts[kColWaves][64], lessv[kColWaves][64], validv[kColWaves][64], base[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ntile = (N + 63) / 64;
  const int tile = blockIdx.x % ntile, slab = blockIdx.x / ntile;
  const int n = tile * 64 + lane;
  const bool live = n < N && !B.fallback[n];
  const int t_begin = (slab * kColWaves + w) * R;    // R rows per wave (a multiple of kColFlight, chosen by the launch)
  const int t_end = min(T, t_begin + R);
  uint32_t less = 0, nvalid = 0, mine = 0;
  if (live && t_begin < T) {
    const uint32_t lo = B.lo[n], span = B.hi[n] - lo, cap = (uint32_t)kColSlots - 1u;
    // rows through a buffer resource based at this wave's first row and tile: the row offset is
    // scalar arithmetic
    const int t_beg_u = __builtin_amdgcn_readfirstlane(t_begin);
    const int rows = __builtin_amdgcn_readfirstlane(t_end - t_begin);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(var + (size_t)t_beg_u * N + tile * 64), 0, 0x7FFFFFFF, 0x00020000);
    const unsigned voff = (unsigned)lane * 4u, row_bytes = (unsigned)N * 4u;
    // branch-free per key: it is written to the lane's next slot and the slot only advances for
    // an in-bracket key (LDS write bandwidth is nowhere near a limit; per-element branches and the
    // mask bookkeeping they forced on the compiler were).  Slot kColSlots-1 is scratch; a lane
    // that finds more in-bracket keys than slots (heavy duplicates) keeps counting and appends the
    // surplus directly afterwards (rescan below).
    auto eat = [&](const float (&v)[kColFlight]) {
#pragma unroll
      for (int u = 0; u < kColFlight; ++u) {
        bool valid;
        const uint32_t key = var_key(v[u], valid);
        nvalid += valid;
        less += valid && key < lo;
        const bool in = valid && (key - lo) <= span;
        stage[w][mine < cap ? mine : cap][lane] = key;
        mine += in;
      }
    };
    // The same for a flight WITHOUT a NaN in any lane (the rule): the clip is an integer maximum on the bit
    // patterns (negative values and -0 are negative integers, everything else orders like its bits), validity
    // needs no per-key work, and one v_cmp_u_f32 tests two keys for NaN at once - 8.5 vector instructions per
    // key instead of 12 (the pass issues 13.5 M of them on the C3 shape: a quarter of its time at 4 cycles each).
    auto eat_clean = [&](const float (&v)[kColFlight]) -> bool {
      bool nan = false;
#pragma unroll
      for (int u = 0; u + 1 < kColFlight; u += 2) nan |= __builtin_isunordered(v[u], v[u + 1]);
      if (kColFlight & 1) nan |= v[kColFlight - 1] != v[kColFlight - 1];
      if (__any(nan)) return false;
      constexpr int kClipBits = 0x2b8cbccc;                    // 1e-12f
      static_assert(kColFlight > 0, "");
#pragma unroll
      for (int u = 0; u < kColFlight; ++u) {
        const int bits = __builtin_bit_cast(int, v[u]);
        const uint32_t key = (uint32_t)(bits > kClipBits ? bits : kClipBits);
        less += key < lo;
        const bool in = (key - lo) <= span;
        stage[w][mine < cap ? mine : cap][lane] = key;
        mine += in;
      }
      nvalid += kColFlight;
      return true;
    };
    if (rows == R) {                 // whole slab: unconditional loads, kColFlight in flight
      // (requesting both flights up front - 64 loads per lane - changes nothing: 0.116 vs 0.114 ms
      // for the whole median on the C3 shape; the pass is not bound by a wave's own latency)
      for (int t = 0; t < R; t += kColFlight) {
        float v[kColFlight];
#pragma unroll
        for (int u = 0; u < kColFlight; ++u)
          v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
              rsrc, voff, (unsigned)(t + u) * row_bytes, 0));
        if (!eat_clean(v)) eat(v);
      }
    } else {                         // the ragged last slab: rows past the end count as NaN
      for (int t = 0; t < rows; t += kColFlight) {
        float v[kColFlight];
#pragma unroll
        for (int u = 0; u < kColFlight; ++u)
          v[u] = t + u < rows ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                    rsrc, voff, (unsigned)(t + u) * row_bytes, 0))
                              : __uint_as_float(0x7FC00000u);
        eat(v);
      }
    }
    if (mine > cap) {       // rare: rescan this lane's rows (L2-hot) and append the keys past the slots
      uint32_t seen = 0;
      for (int t = 0; t < rows; ++t) {
        bool valid;
        const uint32_t key = var_key(var[(size_t)(t_begin + t) * N + n], valid);
        if (valid && (key - lo) <= span && seen++ >= cap) {
          const uint32_t g = atomicAdd(&B.cnt[n], 1u);
          if (g < B.cap) B.list[(size_t)n * B.cap + g] = key;
        }
      }
      mine = cap;
    }
  }
  // the 8 waves' counters meet in LDS and wave 0 issues ONE atomic per counter and chain: every
  // wave adding its own `less` / `valid` (1 568 waves per tile onto the same 256-byte row) cost
  // ~10 us of the pass on the C3 shape
  counts[w][lane] = mine;
  lessv[w][lane] = less;
  validv[w][lane] = nvalid;
  __syncthreads();
  if (w == 0) {
    uint32_t tot = 0, tl = 0, tv = 0;
#pragma unroll
    for (int q = 0; q < kColWaves; ++q) {
      tot += counts[q][lane];
      tl += lessv[q][lane];
      tv += validv[q][lane];
    }
    base[lane] = tot ? atomicAdd(&B.cnt[n], tot) : 0u;     // tot > 0 implies a live chain
    if (tl) atomicAdd(&B.less[n], tl);                      // (non-zero only for live chains)
    if (tv) atomicAdd(&B.valid[n], tv);
  }
  __syncthreads();
  // wave w writes the runs of chains w, w + 8, ...; lane = (source wave, slot within a group of 8)
  const int sw = lane >> 3, sq = lane & 7;
  for (int c = w; c < 64; c += kColWaves) {
    const int nn = tile * 64 + c;
    if (nn >= N) break;
    uint32_t off = base[c];
#pragma unroll
    for (int q = 0; q < kColWaves; ++q) off += q < sw ? counts[q][c] : 0u;
    const uint32_t cnt = counts[sw][c];
#pragma unroll
    for (int q0 = 0; q0 < kColSlots; q0 += 8) {
      const uint32_t q = (uint32_t)(q0 + sq);
      if (q < cnt && off + q < B.cap)
        B.list[(size_t)nn * B.cap + off + q] = stage[sw][q][c];
    }
  }
}

// exact median of one chain's column by radix select inside one block (fallback path)
// (called by the whole 256-thread block of bracket_finish_kernel; hist: 256 counters, sc: 6 words)
template <int NT>
__device__ void median_column(int T, int N, const float* __restrict__ var, double min_var, int n,
                              double* __restrict__ rconst, uint32_t* hist, uint32_t* sc) {
  uint32_t &sh_prefix = sc[0], &sh_rank = sc[1], &sh_cnt = sc[2], &sh_less = sc[3], &sh_eq = sc[4],
           &sh_next = sc[5];
  // one strided sweep over the column, 16 loads in flight per thread (a lone block is otherwise
  // bound by T / 256 dependent memory latencies per sweep)
  auto sweep = [&](auto&& f) {
    constexpr int kU = 16;
    for (int t0 = threadIdx.x; t0 < T; t0 += NT * kU) {
      float v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int t = t0 + NT * u;
        v[u] = t < T ? var[(size_t)t * N + n] : __uint_as_float(0x7FC00000u);
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        bool valid;
        const uint32_t key = var_key(v[u], valid);
        if (valid) f(key);
      }
    }
  };
  uint32_t prefix = 0, rank = 0, less = 0;
  for (int pass = 0; pass < 4; ++pass) {
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    sweep([&](uint32_t key) {
      if (pass > 0 && (key >> (shift + 8)) != prefix) return;
      atomicAdd(&hist[(key >> shift) & 255u], 1u);
    });
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
  sweep([&](uint32_t key) {
    if (key > key_lo && key < best) best = key;
  });
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

// Ranks a <= b <= a + 1 (0-based) of the L keys of one chain's in-bracket list IN GLOBAL MEMORY (contiguous, so the
// sweeps are coalesced and L2-resident): sequences so long that the ~7 % of their frames inside the bracket exceed
// the LDS list of bracket_finish_kernel (T > ~230 000) - before round 4 such chains fell back to median_column's
// five strided sweeps of the whole column (400 000 frames x 16 keypoints: 1.55 ms for the call instead of 0.1).
// Same MSB-first radix select, 4 x 8 bits + one sweep for the successor.  Whole 256-thread block; hist: 256
// counters, sc: 6 words.
template <int NT>
__device__ void select_two_from_list(const uint32_t* __restrict__ keys, uint32_t L, uint32_t a, uint32_t b,
                                     uint32_t* hist, uint32_t* sc, uint32_t& v_lo, uint32_t& v_hi) {
  uint32_t &sh_prefix = sc[0], &sh_rank = sc[1], &sh_less = sc[3], &sh_eq = sc[4], &sh_next = sc[5];
  uint32_t prefix = 0, rank = a, less = 0;
  for (int pass = 0; pass < 4; ++pass) {
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    for (uint32_t i = threadIdx.x; i < L; i += NT) {
      const uint32_t key = keys[i];
      if (pass > 0 && (key >> (shift + 8)) != prefix) continue;
      atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t cum = 0;
      int bin = 255;
      for (int q = 0; q < 256; ++q) {
        if (cum + hist[q] > rank) {
          bin = q;
          break;
        }
        cum += hist[q];
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
  const uint32_t eq = sh_eq, key_lo = prefix;
  uint32_t best = 0xFFFFFFFFu;
  for (uint32_t i = threadIdx.x; i < L; i += NT) {
    const uint32_t key = keys[i];
    if (key > key_lo && key < best) best = key;
  }
  if (best != 0xFFFFFFFFu) atomicMin(&sh_next, best);
  __syncthreads();
  v_lo = key_lo;
  v_hi = (b < less + eq) ? key_lo : sh_next;
  __syncthreads();
}

// S4: one block per chain
template <int NT>
__global__ __launch_bounds__(NT) void bracket_finish_kernel(int T, int N, const float* __restrict__ var,
                                                            double min_var, BracketWs B,
                                                            double* __restrict__ rconst) {
  __shared__ uint32_t vals[kMedList];
  __shared__ uint32_t hist[256], small[kSelSmall];
  __shared__ uint32_t sh[2], sc[6];
  const int n = blockIdx.x;
  const uint32_t cnt = B.valid[n], less = B.less[n], inside = B.cnt[n];
  const uint32_t r_lo = cnt ? (cnt - 1) / 2 : 0, r_hi = cnt / 2;
  // too few valid samples, the bracket missed the median, or heavy duplicates overflowed the
  // list: this block selects from the chain's whole column instead (block-uniform branch)
  if (B.fallback[n] || cnt == 0 || inside > B.cap || r_lo < less || r_hi >= less + inside) {
    median_column<NT>(T, N, var, min_var, n, rconst, hist, sc);
    return;
  }
  if (inside > (uint32_t)kMedList) {    // a long sequence: the list does not fit LDS, select in place (block-uniform)
    uint32_t v_lo, v_hi;
    select_two_from_list<NT>(B.list + (size_t)n * B.cap, inside, r_lo - less, r_hi - less, hist, sc, v_lo, v_hi);
    if (threadIdx.x == 0) {
      const double med = 0.5 * (double)__uint_as_float(v_lo) + 0.5 * (double)__uint_as_float(v_hi);
      rconst[n] = med > min_var ? med : min_var;
    }
    return;
  }
  for (int i = threadIdx.x; i < (int)inside; i += NT) vals[i] = B.list[(size_t)n * B.cap + i];
  __syncthreads();
  const uint32_t blo = B.lo[n], bhi = B.hi[n];          // every listed key lies in [blo, bhi]
  const int lsh = bhi > blo ? __clz((int)(bhi - blo)) : 0;
  uint32_t nxt_in_bin = 0;
  bool has_next = false;
  const uint32_t v_lo = lds_select_compact<NT>(vals, (int)inside, r_lo - less, blo, lsh, hist, sc, small,
                                           &nxt_in_bin, &has_next);
  uint32_t v_hi = v_lo;
  if (r_hi != r_lo && has_next) {
    v_hi = nxt_in_bin;                       // the upper middle rank sits in the same first-pass bin
  } else if (r_hi != r_lo) {
    // the upper middle rank is v_lo again if enough keys are <= v_lo, else the next larger key
    if (threadIdx.x == 0) {
      sh[0] = 0u;
      sh[1] = 0xFFFFFFFFu;
    }
    __syncthreads();
    uint32_t le = 0, nxt = 0xFFFFFFFFu;
    for (int i = threadIdx.x; i < (int)inside; i += NT) {
      const uint32_t k = vals[i];
      le += k <= v_lo;
      if (k > v_lo) nxt = min(nxt, k);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      le += __shfl_xor((int)le, off);
      nxt = min(nxt, (uint32_t)__shfl_xor((int)nxt, off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&sh[0], le);
      atomicMin(&sh[1], nxt);
    }
    __syncthreads();
    v_hi = (r_hi - less) < sh[0] ? v_lo : sh[1];
  }
  if (threadIdx.x == 0) {
    const double med = 0.5 * (double)__uint_as_float(v_lo) + 0.5 * (double)__uint_as_float(v_hi);
    rconst[n] = med > min_var ? med : min_var;
  }
}

static inline size_t arr_bytes(size_t n) { return align_up(n * 4, 256); }

// Rows per wave of the full pass (a multiple of kColFlight).  Measured on MI355X, whole eks_const_r call
// (tools/med_rows_sweep.py; 16 / 32 / 48 / 64 rows): 100 000 x 256 keypoints 81.7 / 71.6 / 70.6 / 80.2 us (the
// pass itself 42.9 us at 32 rows against 51.5 at the 64 of rounds 2-3, and 58 / 66 at 80 / 96: a block holds its
// LDS and wave slots until its slowest wave has finished, and long waves leave a long, thin tail), 50 000 x 4 096:
// 675 / 642 / 646 / 648, 10 000 x 64: 29.4 / 29.5 / 30.0 / 30.8, 3 000 x 30: 24.5 / 25.2 / 26.3 / 36.4.
// Hence 32 rows, or 16 while that leaves CUs without a block.  EKS_MED_ROWS overrides (A/B runs).
static int collect_rows_per_wave(int T, int ntile) {
  static const int cus = [] {
    int dev = 0, n = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  const int forced = knob_int(KNOB_MED_ROWS, 0);
  if (forced >= kColFlight && forced <= 128) return forced / kColFlight * kColFlight;
  const long blocks32 = (long)((T + 32 * kColWaves - 1) / (32 * kColWaves)) * ntile;
  return blocks32 >= cus ? 32 : 16;
}

// capacity of a chain's in-bracket list: the LDS list of the finish kernel, or - for sequences whose bracket
// (4.5 sigma of the 4096-row sample's median rank either side: ~7.1 % of the frames) outgrows it - 9 % of T
static inline uint32_t list_capacity(int T) {
  const size_t want = (size_t)T * 9 / 100 + 1024;
  return want <= (size_t)kMedList ? (uint32_t)kMedList : (uint32_t)align_up(want, 64);
}

size_t const_r_workspace_bytes(int T, int N) {
  return 8 * arr_bytes(N) + arr_bytes((size_t)N * kMedSamples) + arr_bytes((size_t)N * list_capacity(T)) + 256;
}

int const_r(int T, int N, const float* var, double min_var, double* rconst, void* ws,
            size_t ws_bytes, hipStream_t st) {
  if (ws_bytes < const_r_workspace_bytes(T, N)) return EKS_ERR_WORKSPACE;
  ProfScope ps("const_r_select", st);
  if (T <= kMedSmall) {
    hipLaunchKernelGGL(median_small_kernel, dim3(N), dim3(256), 0, st, T, N, var, min_var, rconst);
    return hip_status(hipGetLastError());
  }
  char* p = static_cast<char*>(ws);
  BracketWs B;
  uint32_t** arrs[6] = {&B.lo, &B.hi, &B.less, &B.valid, &B.cnt, &B.fallback};
  for (int i = 0; i < 6; ++i) {
    *arrs[i] = reinterpret_cast<uint32_t*>(p);
    p += arr_bytes(N);
  }
  B.smp = reinterpret_cast<uint32_t*>(p);
  p += arr_bytes((size_t)N * kMedSamples);
  B.list = reinterpret_cast<uint32_t*>(p);
  B.cap = list_capacity(T);

  const int S = sample_count(T);
  const int ntile = (N + 63) / 64;
  hipLaunchKernelGGL(sample_transpose_kernel, dim3(ntile, (S + 63) / 64), dim3(256), 0, st, T, N, S,
                     var, B);
  if (knob_int(KNOB_MED_BRACKET_THREADS, 256) == 512)
    hipLaunchKernelGGL(sample_bracket_kernel<512>, dim3(N), dim3(512), 0, st, N, S, B);
  else
    hipLaunchKernelGGL(sample_bracket_kernel<256>, dim3(N), dim3(256), 0, st, N, S, B);
  const int R = collect_rows_per_wave(T, ntile);
  const int nslab = (T + R * kColWaves - 1) / (R * kColWaves);
  hipLaunchKernelGGL(bracket_collect_kernel, dim3((unsigned)(ntile * nslab)), dim3(64 * kColWaves), 0, st,
                     T, N, R, var, B);
  // (the block's helpers stride by blockDim.x: any multiple of 256 threads; EKS_MED_FINISH_THREADS for A/B runs)
  // measured on C3 (512 chains, ~7 000 listed keys each): 256 threads 16.1 us, 512: 13.6, 1 024: 16.9
  int ft = knob_int(KNOB_MED_FINISH_THREADS, 512);
  if (ft != 256 && ft != 512) ft = 512;
  if (ft == 512)
    hipLaunchKernelGGL(bracket_finish_kernel<512>, dim3(N), dim3(512), 0, st, T, N, var, min_var, B, rconst);
  else
    hipLaunchKernelGGL(bracket_finish_kernel<256>, dim3(N), dim3(256), 0, st, T, N, var, min_var, B, rconst);
  return hip_status(hipGetLastError());
}

// ==========================================================================================
// numpy.nanstd of every row of a [K][n] float32 matrix, bit for bit (eks_np_sum.hpp).  Block = row: the row sits
// in LDS, thread i sums leaf i of numpy's pairwise recursion, thread 0 walks the combine program - both tables are
// a function of n alone and come from the caller (hip_ops.np_nanstd_rows builds them once per n).
// ==========================================================================================
// DIFF: row k is not read but formed - element t O + o = x[t + 1][k][o] - x[t][k][o] of a frame-major [.][K][O] tensor
// (the frame-to-frame differences of the ensemble variances, reference eks/core.py:128-129; the float32 subtraction is
// the same IEEE operation as NumPy's, so the row is the host's bit for bit without the subtraction and the transposing
// copy as launches of their own)
template <bool DIFF>
__global__ __launch_bounds__(256) void np_nanstd_rows_kernel(int K, int n, const float* __restrict__ d, int O,
                                                            const int32_t* __restrict__ leaves, int n_leaves,
                                                            const int32_t* __restrict__ ops, int n_ops,
                                                            float* __restrict__ out) {
  extern __shared__ float npl[];             // row[n] | slots[n_leaves + n_ops]
  __shared__ int nan_count;
  __shared__ float avg_sh;
  float* row = npl;
  float* slot = npl + n;
  const int k = blockIdx.x;
  if (threadIdx.x == 0) nan_count = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    float v;
    if constexpr (DIFF) {
      const int t = i / O, o = i - t * O;
      const size_t at = ((size_t)t * K + k) * O + o;
      v = np_sub(d[at + (size_t)K * O], d[at]);
    } else {
      v = d[(size_t)k * n + i];
    }
    row[i] = v;
    mine += v != v;
  }
  if (mine) atomicAdd(&nan_count, mine);
  __syncthreads();
  const int cnt = n - nan_count;
  if (cnt == 0) {                            // numpy: nan (and a RuntimeWarning)
    if (threadIdx.x == 0) out[k] = __uint_as_float(0x7FC00000u);
    return;
  }
  auto reduce = [&](auto&& value) -> float {  // the pairwise sum of value(0 .. n - 1); result valid in thread 0
    for (int l = threadIdx.x; l < n_leaves; l += 256) {
      const int s0 = leaves[2 * l], len = leaves[2 * l + 1];
      slot[l] = np_leaf_sum(len, [&](int i) { return value(s0 + i); });
    }
    __syncthreads();
    float res = 0.f;
    if (threadIdx.x == 0) {
      for (int q = 0; q < n_ops; ++q) slot[ops[3 * q]] = np_add(slot[ops[3 * q + 1]], slot[ops[3 * q + 2]]);
      res = np_add(0.f, slot[n_leaves + n_ops - 1]);
    }
    __syncthreads();
    return res;
  };
  const float s = reduce([&](int i) {
    const float v = row[i];
    return v != v ? 0.f : v;
  });
  if (threadIdx.x == 0) avg_sh = np_divide_by_count(s, cnt);
  __syncthreads();
  const float avg = avg_sh;
  const float v2 = reduce([&](int i) {
    const float v = row[i];
    const float x = v != v ? 0.f : np_sub(v, avg);
    return np_mul(x, x);
  });
  if (threadIdx.x == 0) out[k] = (float)sqrt((double)np_divide_by_count(v2, cnt));
}

int np_nanstd_rows(int K, int n, const float* d, const int32_t* leaves, int n_leaves, const int32_t* ops, int n_ops,
                   float* out, hipStream_t st) {
  const size_t shm = ((size_t)n + n_leaves + n_ops) * sizeof(float);
  if (shm > 64 * 1024) return EKS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(np_nanstd_rows_kernel<false>, dim3(K), dim3(256), shm, st, K, n, d, 1, leaves, n_leaves, ops, n_ops,
                     out);
  return hip_status(hipGetLastError());
}

// x [n_frames][K][O] frame-major: rows of (n_frames - 1) O differences per keypoint
int np_nanstd_diff_rows(int n_frames, int K, int O, const float* x, const int32_t* leaves, int n_leaves,
                        const int32_t* ops, int n_ops, float* out, hipStream_t st) {
  const int n = (n_frames - 1) * O;
  const size_t shm = ((size_t)n + n_leaves + n_ops) * sizeof(float);
  if (shm > 64 * 1024) return EKS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(np_nanstd_rows_kernel<true>, dim3(K), dim3(256), shm, st, K, n, x, O, leaves, n_leaves, ops, n_ops,
                     out);
  return hip_status(hipGetLastError());
}

// ==========================================================================================
// Order statistics of the columns of a [T][N] float32 matrix: the two neighbours numpy.percentile
// interpolates between (reference eks/utils.py:318-322 center_predictions, eks/stats.py:109-112 the
// variance-inflation loop; numpy sorts NaNs to the end and the drivers need to know whether a column has any).
// One block per column, exact MSB-first radix select on the order-preserving bit pattern of the floats
// (four strided sweeps, 16 loads in flight per thread) plus one sweep for the successor of the selected key.
// These matrices are small (frames x keypoints of one session), the sweeps are not a hot path.
// ==========================================================================================
__device__ __forceinline__ uint32_t float_order_key(float v) {
  if (v != v) return 0xFFFFFFFFu;                       // NaN: after everything else, +inf included
  const uint32_t u = __float_as_uint(v);
  const uint32_t k = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return k == 0xFFFFFFFFu ? 0xFFFFFFFEu : k;            // (no finite value or infinity maps there anyway)
}
__device__ __forceinline__ float float_from_order_key(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}

__global__ __launch_bounds__(256) void order_stats_kernel(int T, int N, const float* __restrict__ x,
                                                         int r_lo, int r_hi, float* __restrict__ out,
                                                         int32_t* __restrict__ nan_count) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh_prefix, sh_rank, sh_less, sh_eq, sh_next, sh_nan;
  const int n = blockIdx.x;
  auto sweep = [&](auto&& f) {
    constexpr int kU = 16;
    for (int t0 = threadIdx.x; t0 < T; t0 += 256 * kU) {
      float v[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int t = t0 + 256 * u;
        v[u] = t < T ? x[(size_t)t * N + n] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kU; ++u)
        if (t0 + 256 * u < T) f(float_order_key(v[u]));
    }
  };
  uint32_t prefix = 0, rank = (uint32_t)r_lo, less = 0;
  if (threadIdx.x == 0) sh_nan = 0u;
  for (int pass = 0; pass < 4; ++pass) {
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    uint32_t nans = 0;
    sweep([&](uint32_t key) {
      if (pass == 0) nans += key == 0xFFFFFFFFu;
      if (pass > 0 && (key >> (shift + 8)) != prefix) return;
      atomicAdd(&hist[(key >> shift) & 255u], 1u);
    });
    if (pass == 0 && nans) atomicAdd(&sh_nan, nans);
    __syncthreads();
    if (threadIdx.x == 0) {
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
  const uint32_t key_lo = prefix, eq = sh_eq;
  uint32_t best = 0xFFFFFFFFu;
  sweep([&](uint32_t key) {
    if (key > key_lo && key < best) best = key;
  });
  if (best != 0xFFFFFFFFu) atomicMin(&sh_next, best);
  __syncthreads();
  if (threadIdx.x == 0) {
    // rank r_hi holds key_lo again while it lies inside the run of equal keys, else the next larger key
    const uint32_t key_hi = ((uint32_t)r_hi < less + eq) ? key_lo : sh_next;
    out[2 * n] = float_from_order_key(key_lo);
    out[2 * n + 1] = key_hi == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : float_from_order_key(key_hi);
    if (key_lo == 0xFFFFFFFFu) out[2 * n] = __uint_as_float(0x7FC00000u);
    nan_count[n] = (int32_t)sh_nan;
  }
}

int order_stats(int T, int N, const float* x, int r_lo, int r_hi, float* out, int32_t* nan_count, hipStream_t st) {
  hipLaunchKernelGGL(order_stats_kernel, dim3(N), dim3(256), 0, st, T, N, x, r_lo, r_hi, out, nan_count);
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
  if (adam_step_block(b, offs, members, nll, dnll, lr, lo, hi, tol, cap, state, s_keypoint))
    atomicAdd(n_active, 1);
}

// up to kAdamOneBlock optimiser blocks: ONE workgroup walks them and writes the count of those still
// running itself - no memset and no atomics in front of every iteration's launch (the memset was a
// 4.8 us launch of its own, 5 % of an iteration on the C3 shape)
constexpr int kAdamOneBlock = 4096;
__global__ __launch_bounds__(256) void adam_step_one_block_kernel(
    int nb, const int32_t* __restrict__ offs, const int32_t* __restrict__ members,
    const double* __restrict__ nll, const double* __restrict__ dnll, double lr, double lo, double hi,
    double tol, int cap, double* __restrict__ state, double* __restrict__ s_keypoint,
    int32_t* __restrict__ n_active) {
  __shared__ int running;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  int mine = 0;
  for (int b = threadIdx.x; b < nb; b += 256)
    mine += adam_step_block(b, offs, members, nll, dnll, lr, lo, hi, tol, cap, state, s_keypoint) ? 1 : 0;
  if (mine) atomicAdd(&running, mine);
  __syncthreads();
  if (threadIdx.x == 0) *n_active = running;
}

int adam_step(int n_blocks, const int32_t* offs, const int32_t* members, const double* nll,
              const double* dnll, double lr, double lo, double hi, double tol, int cap,
              double* state, double* s_keypoint, int32_t* n_active, hipStream_t st) {
  if (n_blocks <= kAdamOneBlock) {
    hipLaunchKernelGGL(adam_step_one_block_kernel, dim3(1), dim3(256), 0, st, n_blocks, offs, members, nll,
                       dnll, lr, lo, hi, tol, cap, state, s_keypoint, n_active);
    return hip_status(hipGetLastError());
  }
  hipError_t e = hipMemsetAsync(n_active, 0, sizeof(int32_t), st);
  if (e != hipSuccess) return hip_status(e);
  hipLaunchKernelGGL(adam_step_kernel, dim3((n_blocks + 127) / 128), dim3(128), 0, st, n_blocks, offs,
                     members, nll, dnll, lr, lo, hi, tol, cap, state, s_keypoint, n_active);
  return hip_status(hipGetLastError());
}

// keypoint -> optimiser block map of the CSR block list and two zeroed counters, once per eks_adam_run
__global__ void adam_prepare_kernel(int nb, const int32_t* __restrict__ offs, const int32_t* __restrict__ members,
                                    int32_t* __restrict__ kp_block, int32_t* __restrict__ ca,
                                    int32_t* __restrict__ cb) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0) {
    *ca = 0;
    *cb = 0;
  }
  if (b >= nb) return;
  for (int i = offs[b]; i < offs[b + 1]; ++i) kp_block[members[i]] = b;
}

int adam_prepare(int n_blocks, int K, const int32_t* offs, const int32_t* members, int32_t* kp_block,
                 int32_t* counter_a, int32_t* counter_b, hipStream_t st) {
  (void)K;
  hipLaunchKernelGGL(adam_prepare_kernel, dim3((n_blocks + 255) / 256), dim3(256), 0, st, n_blocks, offs, members,
                     kp_block, counter_a, counter_b);
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
  if (pupil_adam_step_chain(k, n, latent_var, nll, dnll, lr, tol, cap, state, a, q, da, dq)) atomicAdd(n_active, 1);
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
  // the member count is a template parameter: the sort network and the sums are sized for M
  // itself (M = 5 is the usual ensemble: 10 compare-exchanges instead of the 28 of an 8-wide sort)
#define EKS_ENS(MM_)                                                                               \
  hipLaunchKernelGGL(ensemble_kernel<MM_>, grid, dim3(256), 0, st, M, VTK, markers, avg_mode, var_mode, \
                     nan_replacement, stats)
  switch (M) {
    case 1: EKS_ENS(1); break;
    case 2: EKS_ENS(2); break;
    case 3: EKS_ENS(3); break;
    case 4: EKS_ENS(4); break;
    case 5: EKS_ENS(5); break;
    case 6: EKS_ENS(6); break;
    case 7: EKS_ENS(7); break;
    case 8: EKS_ENS(8); break;
    default: EKS_ENS(kMaxModels); break;
  }
#undef EKS_ENS
  return hip_status(hipGetLastError());
}

}  // namespace eks

EKS_DEFINE_TOUCH(misc)
