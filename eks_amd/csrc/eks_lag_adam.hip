// The Adam search on log s (reference eks/core.py:562-699: one optimiser per keypoint, loss = constant-R filter NLL of
// eks/core.py:640-650) WITHOUT a pass over the observations per iteration (round 6).
//
// On a scalar chain (x' = a x + N(0, s q), y = c x + N(0, r), r constant) the innovation obeys
//     e_{t+1} = rho_t e_t + u_{t+1},     rho_t = a r / S_t,     u_t = y_t - a y_{t-1},
// and the inputs u do NOT depend on s or r.  The predicted variance is a Moebius iteration with fixed points
// P_inf > 0 > P_-:  (P_t - P_inf) / (P_t - P_-) = kappa^t w_0,  kappa = S_- / S_inf = rho^2 - so S_t is known in closed
// form for every t at once - and past the head the variance has converged, d_t = Dz_t + rho^(t - B0) e_B0 with Dz the
// zero-start recursion on the inputs from frame F = B0 + 1 on, whose sum of squares is a polynomial in the pole:
//     sum_t Dz_t^2 = [ c_0 + 2 sum_{k>=1} rho^k c_k - rho^2 Dz_{T-1}^2 ] / (1 - rho^2),    c_k = sum_{t >= F + k} u_t u_{t-k}.
//
//   L1 lag_sums_kernel    one streaming pass per search: block = (64-chain tile, time chunk), wave w = lags 16 w .. 16 w + 15
//                         as packed float32 products of the current inputs with the inputs 16 w frames back (17
//                         v_pk_fma_f32 per frame pair and wave), the inputs through one ring in LDS, float64 sums in
//                         registers every 64 frames.
//   L2 lag_reduce_kernel  the chunks' partial sums -> c[chain][256] (float64, lags >= 1 doubled).
//   L3 lag_adam_kernel    block = KEYPOINT, wave = chain, the whole search in one launch with no exchange between
//                         blocks: per iteration the head [0, B0) exactly and time-parallel (closed-form variances, a
//                         64-lane DPP scan of the affine maps e -> rho e + u, four frames per lane), the rest from the
//                         256 lag sums and the first / last 256 inputs, everything in float64 dual numbers
//                         (d / d log s), then the optimiser step in registers.  A chain whose pole leaves the range the
//                         lag sums cover (|rho|^256 <= 1e-10 (1 - |rho|): |rho| <= 0.906) is evaluated EXACTLY instead, by
//                         streaming its own frames from a private chain-major copy (lane = time chunk, same scan):
//                         slower (~30 us per iteration at T = 100 000), never wrong.
// tools/lag_adam_proto.py is the NumPy statement of the same identities (CPU tests: tests/test_lag_identity.py).
#include <hip/hip_runtime.h>

#include "eks_adam.hpp"
#include "eks_internal.hpp"
#include "eks_nll_lane.hpp"

namespace eks {

constexpr int kLaB0 = 256;             // frames of the head: exact, time-parallel (64 lanes x 4 frames)
constexpr int kLaF = kLaB0 + 1;        // first frame whose input enters the lag sums
constexpr int kLaL = 256;              // lag sums c_0 .. c_255
constexpr int kLaMinT = 1024;          // (shorter sessions: diag_nll_adam_persist_kernel)
constexpr int kLaMaxD = 4;
static_assert(kLaB0 == kLaL, "the search kernel deals four head frames and four lags to every lane");

// |rho| up to which 256 lag sums give the polynomial to 2e-10 (1 - |rho|)^-1 of itself WHATEVER the data (all lag sums
// are bounded by c_0): |rho|^256 <= 1e-10 (1 - |rho|)
static double lag_adam_rho_max() {
  double lo = 0.0, hi = 0.999;
  for (int it = 0; it < 60; ++it) {
    const double m = 0.5 * (lo + hi);
    if (pow(m, (double)kLaL) <= 1e-10 * (1.0 - m)) lo = m; else hi = m;
  }
  return lo;
}

// ---- rows through a buffer resource based `base_row` rows into the array (scalar row offsets: no VALU address
// arithmetic).  Frames in front of B0 read row B0 and frames past the end the last row: with a = 1 the inputs
// u_t = y_t - y_{t-1} of all frames outside [F, T) are then exactly zero - what the lag sums want - without a mask
// (other models multiply by one)
struct LagRows {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff, row_bytes;
  int base_row, last_row;
  __device__ __forceinline__ float operator()(int t) const {       // t: frame index (wave-uniform)
    int tc = t < last_row ? t : last_row;
    tc = tc > kLaB0 ? tc : kLaB0;
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (unsigned)(tc - base_row) * row_bytes, 0));
  }
};

struct LagPre {
  int T, N, D, ntile, nch, CL;         // chunk j: frames [F + j CL, F + (j + 1) CL), CL a multiple of 64
  const float* y;
  const double* A;                     // [K][D][D]
  double* part;                        // [nch][N][kLaL]
  const double* state;                 // optimiser state: a tile none of whose keypoints still runs is skipped
  const int32_t* kp_block;
  int cap;
};

constexpr int kRgHist = 288;             // frames in front of the chunk that the prologue fills (a multiple of 32 >= 272)

// The pre-pass, on the matrix cores (third form of round 6).  The 256 lag sums of a chain are ONE 16 x 16 matrix product
// with the frames as its inner dimension, no entry computed twice:
//     C[i][j] = sum_t u_{t+i} u_{t-16 j} = c_{i + 16 j},     A[i][t] = u_{t+i},  B[t][j] = u_{t-16 j}
// (every pair (tau, tau - k) appears once: tau = t + i for the one i = k mod 16), so v_mfma_f32_16x16x4_f32 - exact
// float32 FMA chains at the vector rate - takes four frames of one chain per instruction.  A group of 16 frames from T0
// is four instructions p = 0 .. 3 whose inner index kq stands for frame T0 + 4 kq + p: lane (kq, x) feeds
// A = u_{T0+4kq+p+x} and B = u_{T0+4kq+p-16x}; the accumulator's register r of lane l is lag 16 (l & 15) + 4 (l >> 4) + r.
// Block = (64-chain tile, time chunk), 8 waves, wave w = chains 8 w .. 8 w + 7 of the tile (two waves per SIMD: the eight
// chains of a wave share a lane's ring addresses, which come from a small table in LDS).  The chunk's inputs go through one
// ring of 384 frames in LDS (every 32 frames each
// wave loads five rows and stores four frames' inputs, two sets ahead of their use; lane = chain), chain-major in time
// order with four words of padding per 64 frames: a lane's B operands of a group are four consecutive words on a 16-byte
// boundary - ONE ds_read_b128, the sixteen lanes of a quarter wave (frames 16 apart) in sixteen different bank quads -
// its A operands four ds_read_b32 of nineteen consecutive words per quarter wave.  The operands of the next pair of chains
// are requested before the current pair's eight instructions issue.  Float32 accumulators span 64 frames and are added
// into float64 registers.
// Measured on BASELINE configs[2] (profiles/r06_probes.txt section 8; us per pass on the same box): the packed-FMA forms
// of this round (8 waves x 32 lags, every wave loading and subtracting both streams: 388; 16 waves x 16 lags behind one
// ring of inputs, 17 v_pk_fma_f32 per frame pair: 313-330, 0.54 of the vector rate under its power limit); matrix
// cores with (t mod 16, t / 16) planes in LDS and eight ds_read_b32 per four instructions: 279 (half of the LDS pipe's
// cycles bank conflicts); this layout with 16 waves x 4 chains: 299; 8 waves x 8 chains: 280; the lanes' addresses from a
// table instead of twenty vector instructions per group: 272.  The matrix instructions
// alone (no operand reads, no producers, no barrier) take 206, with producers and barrier 224: the instruction holds its
// SIMD's vector issue while it runs, so every address, conversion and float64 addition beside it is added time.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kMfRing = 384;                  // frames in the ring: 240 of delay + the set in use + two in flight + 18 ahead
constexpr int kMfGroups = kMfRing / 16;
constexpr int kMfChP = kMfRing + 4 * (kMfRing / 64) + 4;     // floats per chain: 4 words of padding per 64 frames (412)
static_assert(kMfChP % 4 == 0 && kMfRing % 64 == 0, "a chain's ring starts on a 16-byte boundary");

constexpr int kMfWaves = 8;                   // two per SIMD: 8 chains per wave share a lane's ring addresses
constexpr int kMfCh = 64 / kMfWaves;          // chains per wave
constexpr int kMfFr = 32 / kMfWaves;          // frames of a 32-frame set a wave produces

template <bool UNIT>
__global__ __launch_bounds__(64 * kMfWaves) void lag_sums_kernel(LagPre P) {
  __shared__ __attribute__((aligned(16))) float ring[64 * kMfChP];
  // a lane's ring offsets (B, then A of the four instructions) for a group that starts in ring column g: the wrap of the
  // ring and its padding make them lane-dependent functions of g - twenty vector instructions per group if computed,
  // five LDS reads from this table (an f32 MFMA holds its SIMD's vector issue: vector instructions beside it are added time)
  __shared__ int tab[kMfGroups][5][64];
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int tile = blockIdx.x % P.ntile, j = blockIdx.x / P.ntile;
  const int n_raw = tile * 64 + lane;
  const int n = n_raw < P.N ? n_raw : P.N - 1;
  const int k = n / P.D, d = n - k * P.D;
  if (P.state != nullptr) {
    const bool running = n_raw < P.N && adam_block_running(P.state, P.kp_block[k], P.cap);
    if (!__any(running)) return;                      // (the same answer in every wave of the tile's blocks)
  }
  const double a_d = P.A[(size_t)k * P.D * P.D + (size_t)d * (P.D + 1)];
  const int T = P.T;
  const int ts0 = kLaF + j * P.CL;                    // (ts0 - 1 is a multiple of 64)
  const int len = min(P.CL, T - ts0);
  const int nsets = (len + 31) / 32;
  const int base_row = max(ts0 - kRgHist - 1, 0);
  const LagRows ld{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.y + (size_t)base_row * P.N + (size_t)tile * 64), 0,
                                                     0x7FFFFFFF, 0x00020000),
                   (unsigned)((n - tile * 64) * 4), (unsigned)(P.N * 4), base_row, T - 1};
  auto input = [&](float yy, float yp) { return UNIT ? (yy - yp) : (float)((double)yy - a_d * (double)yp); };
  auto ringpos = [](int f) {                          // (f - 1: groups start on multiples of 16; f may be negative)
    const int r = (f - 1) % kMfRing;
    return r < 0 ? r + kMfRing : r;
  };
  auto slot = [&](int f) {                            // frame f (wave-uniform) in a chain's ring
    const int r = ringpos(f);
    return r + 4 * (r >> 6);
  };
  // ---- producers: of every 32 frames from t, wave w owns frames t + 4 w .. t + 4 w + 3 (five rows); lane = chain
  auto rows_of = [&](int t, float (&r)[kMfFr + 1]) {
#pragma unroll
    for (int q = 0; q <= kMfFr; ++q) r[q] = ld(t + kMfFr * w - 1 + q);
  };
  float* mine = ring + lane * kMfChP;
  auto store_u = [&](int t, const float (&r)[kMfFr + 1]) {
#pragma unroll
    for (int q = 0; q < kMfFr; ++q) {
      const int f = t + kMfFr * w + q;
      float u = input(r[q + 1], r[q]);
      if (!UNIT) u = (f >= kLaF && f < T) ? u : 0.f;  // (with a = 1 the clamped rows give zero by themselves: LagRows)
      mine[slot(f)] = u;
    }
  };
  {
    constexpr int NP = (kRgHist + 64) / 32;
    float r[NP][kMfFr + 1];
#pragma unroll
    for (int i = 0; i < NP; ++i) rows_of(ts0 - kRgHist + 32 * i, r[i]);
#pragma unroll
    for (int i = 0; i < NP; ++i) store_u(ts0 - kRgHist + 32 * i, r[i]);
  }
  float nxt[kMfFr + 1];
  rows_of(ts0 + 64, nxt);
  const int kq = lane >> 4, x = lane & 15;
  for (int e = w; e < kMfGroups * 5; e += kMfWaves) {  // (wave-uniform e)
    const int g = e / 5, a = e - 5 * g;
    int r = 16 * g + 4 * kq + (a == 0 ? -16 * x : x + a - 1);
    r = r < 0 ? r + kMfRing : r;
    r = r >= kMfRing ? r - kMfRing : r;
    tab[g][a][lane] = r + 4 * (r >> 6);
  }
  __syncthreads();
  // ---- consumers: lane = (kq, x) of the operands; chains kMfCh w + cc at + cc kMfChP words (an immediate of the read)
  const float* mych = ring + kMfCh * w * kMfChP;
  struct Ops {
    float a[2][4];
    f32x4 b[2];
  };
  struct Addr {
    int ib, ia[4];
  };
  auto address = [&](int T0) {                         // a lane's five ring offsets for the group of 16 frames from T0
    Addr q;
    const int g = ringpos(T0) >> 4;                   // (wave-uniform; ringpos(T0) is a multiple of 16)
    q.ib = tab[g][0][lane];
#pragma unroll
    for (int p = 0; p < 4; ++p) q.ia[p] = tab[g][1 + p][lane];
    return q;
  };
  auto request = [&](const Addr& q, int h, Ops& o) {   // operands of the chains 2 h, 2 h + 1
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float* ch = mych + (2 * h + c) * kMfChP;
      o.b[c] = *reinterpret_cast<const f32x4*>(ch + q.ib);
#pragma unroll
      for (int p = 0; p < 4; ++p) o.a[c][p] = ch[q.ia[p]];
    }
  };
  f32x4 acc[kMfCh];
  double sum[kMfCh][4];
#pragma unroll
  for (int cc = 0; cc < kMfCh; ++cc) {
    acc[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) sum[cc][r] = 0.0;
  }
  auto multiply = [&](int h, const Ops& o) {          // eight instructions alternating between the pair's accumulators
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
        acc[2 * h + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[c][p], o.b[c][p], acc[2 * h + c], 0, 0, 0);
    }
  };
  // the pairs of a group in turn; the operands of the pair kMfAhead steps on (into the next group, and from a set's second
  // group into the next set's first, whose frames are visible since the last barrier) are requested before the current
  // pair's instructions issue
  constexpr int NH = kMfCh / 2;
  constexpr int kMfAhead = 1;                         // (three pairs ahead, four operand buffers: 272 us against 266)
  static_assert(kMfAhead >= 1 && kMfAhead <= NH && (2 * NH) % (kMfAhead + 1) == 0,
                "requests stay within the next group; the operand buffers take turns with the period of a set");
  Ops ops[kMfAhead + 1];
  Addr qa = address(ts0);
#pragma unroll
  for (int h = 0; h < kMfAhead; ++h) request(qa, h, ops[h]);
  for (int s = 0; s < nsets; ++s) {
    const int t = ts0 + 32 * s;
    store_u(t + 64, nxt);
    rows_of(t + 96, nxt);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const Addr qn = address(t + 16 * (g + 1));
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int step = g * NH + h, ahead = h + kMfAhead;
        if (ahead < NH) request(qa, ahead, ops[(step + kMfAhead) % (kMfAhead + 1)]);
        else request(qn, ahead - NH, ops[(step + kMfAhead) % (kMfAhead + 1)]);
        multiply(h, ops[step % (kMfAhead + 1)]);
      }
      qa = qn;
    }
    if ((s & 1) || s + 1 == nsets) {
#pragma unroll
      for (int cc = 0; cc < kMfCh; ++cc) {              // float32 partial sums span 64 frames
#pragma unroll
        for (int r = 0; r < 4; ++r) sum[cc][r] += (double)acc[cc][r];
        acc[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();
  }
  // lane (kq, x) holds lags 16 x + 4 kq + r of its chains: 32 consecutive bytes each
#pragma unroll
  for (int cc = 0; cc < kMfCh; ++cc) {
    const int nn = tile * 64 + kMfCh * w + cc;
    if (nn >= P.N) break;                              // (wave-uniform)
    double* o = P.part + ((size_t)j * P.N + nn) * kLaL + 16 * x + 4 * kq;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = sum[cc][r];
  }
}

__global__ void lag_reduce_kernel(int N, int nch, const double* __restrict__ part, double* __restrict__ ck) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;           // = n * 256 + lag
  if (idx >= (long)N * kLaL) return;
  const int lag = (int)(idx & (kLaL - 1));
  double s = 0.0;
  for (int j = 0; j < nch; ++j) s += part[(size_t)j * N * kLaL + idx];
  ck[idx] = lag ? 2.0 * s : s;
}

// ---- cross-lane pieces of the search kernel (DPP: row shifts inside rows of 16, row broadcasts across them) ----------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double la_dpp(double old, double x) {
  return __builtin_amdgcn_update_dpp(old, x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ double la_readlane(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l), hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}
// the same with zeros where a lane has no source (bound_ctrl): every lane is written, so no register is set up to
// hold the `old` value (two v_mov per 64-bit value and step otherwise)
template <int CTRL>
__device__ __forceinline__ double la_dpp_z(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, CTRL, 0xf, 0xf, true);
}
// sum / product over the 64 lanes, the same value in every lane.  The sum only has to be right in lane 63: the row
// broadcasts go to every row (row 1 takes row 0's total, row 2 row 1's, row 3 row 2's; then rows 2, 3 take lane 31's
// R0 + R1), which leaves R0 + R1 + R2 + R3 in the last row and needs no row mask, hence no `old`
__device__ __forceinline__ double la_wave_sum(double x) {
  x += la_dpp_z<0x111>(x);
  x += la_dpp_z<0x112>(x);
  x += la_dpp_z<0x114>(x);
  x += la_dpp_z<0x118>(x);
  x += la_dpp_z<0x142>(x);                // row_bcast:15
  x += la_dpp_z<0x143>(x);                // row_bcast:31
  return la_readlane(x, 63);
}
__device__ __forceinline__ double la_wave_prod(double x) {
  x *= la_dpp<0x111>(1.0, x);
  x *= la_dpp<0x112>(1.0, x);
  x *= la_dpp<0x114>(1.0, x);
  x *= la_dpp<0x118>(1.0, x);
  x *= la_dpp<0x142, 0xa>(1.0, x);
  x *= la_dpp<0x143, 0xc>(1.0, x);
  return la_readlane(x, 63);
}
// e_out = A e_in + b with d / d log s riding along
struct LaAff {
  DualD A, b;
};
__device__ __forceinline__ LaAff la_then(const LaAff& first, const LaAff& second) {
  return LaAff{second.A * first.A, second.A * first.b + second.b};
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ void la_scan_step(LaAff& x) {
  LaAff e;
  e.A.v = la_dpp<CTRL, ROW_MASK>(1.0, x.A.v);
  e.A.d = la_dpp<CTRL, ROW_MASK>(0.0, x.A.d);
  e.b.v = la_dpp<CTRL, ROW_MASK>(0.0, x.b.v);
  e.b.d = la_dpp<CTRL, ROW_MASK>(0.0, x.b.d);
  x = la_then(e, x);
}
// a shift inside the rows of 16: lanes without a source read zeros, which IS the identity map but for the high word
// of A = 1.0 (one select instead of eight moves)
template <int SH>
__device__ __forceinline__ void la_scan_row_step(LaAff& x, int lane) {
  LaAff e;
  const double av = la_dpp_z<0x110 + SH>(x.A.v);
  const int hi = (lane & 15) < SH ? 0x3FF00000 : __double2hiint(av);
  e.A.v = __hiloint2double(hi, __double2loint(av));
  e.A.d = la_dpp_z<0x110 + SH>(x.A.d);
  e.b.v = la_dpp_z<0x110 + SH>(x.b.v);
  e.b.d = la_dpp_z<0x110 + SH>(x.b.d);
  x = la_then(e, x);
}
// inclusive scan in lane order: afterwards lane i maps the innovation entering lane 0 to the one leaving lane i
__device__ __forceinline__ void la_scan(LaAff& x, int lane) {
  la_scan_row_step<1>(x, lane);
  la_scan_row_step<2>(x, lane);
  la_scan_row_step<4>(x, lane);
  la_scan_row_step<8>(x, lane);
  la_scan_step<0x142, 0xa>(x);
  la_scan_step<0x143, 0xc>(x);
}
// x^e for a per-lane whole exponent below 2^NBITS
template <int NBITS>
__device__ __forceinline__ double la_pow_bits(double x, unsigned e) {
  double r = 1.0, b = x;
#pragma unroll
  for (int i = 0; i < NBITS; ++i) {
    r = (e >> i) & 1u ? r * b : r;
    b *= b;
  }
  return r;
}

// steady-state constants of one chain at one s, with d / d log s
struct LaConst {
  double a, c, r, c2, P0;
  DualD Sinf, g, rho, dS;      // dS = S_inf - S_-
  double dlr;                  // d log rho / d log s = -g dS_inf
  double kap, w0, dw0;         // kappa = S_- / S_inf, d log kappa = 2 dlr
  double om0;                  // 1 - w_0 = (P_inf - P_-) / (P_0 - P_-), formed without the cancellation of 1 - w_0
};
__device__ __forceinline__ LaConst la_const(double a, double c, double q, double r, double P0, double s) {
  LaConst K;
  K.a = a; K.c = c; K.r = r; K.c2 = c * c; K.P0 = P0;
  const double sq = s * q;
  double Pinf, dPinf;
  riccati_fixed_point(a, c, r, sq, Pinf, dPinf);
  K.Sinf = DualD(r + K.c2 * Pinf, K.c2 * dPinf);
  K.g = rcp(K.Sinf);
  K.dlr = -K.g.v * K.Sinf.d;
  K.rho = DualD(a * r * K.g.v, a * r * K.g.d);
  const DualD Sm = DualD(a * r * K.rho.v, a * r * K.rho.d);
  K.dS = K.Sinf - Sm;
  K.kap = Sm.v * K.g.v;
  const double ic2 = rcp(K.c2);
  const double Pm = (Sm.v - r) * ic2, dPm = Sm.d * ic2;
  const double den = P0 - Pm, iden = rcp(den);
  K.w0 = (P0 - Pinf) * iden;
  K.om0 = (Pinf - Pm) * iden;
  K.dw0 = (-dPinf * den + (P0 - Pinf) * dPm) * iden * iden;
  return K;
}
// innovation variance of frame t, its reciprocal and the frame's pole (w_t = w_0 kappa^t given as kt = kappa^t)
__device__ __forceinline__ void la_frame(const LaConst& K, int t, double kt, DualD& St, DualD& gt, DualD& rt) {
  const double dkt = (double)t * kt * 2.0 * K.dlr;
  const DualD w(K.w0 * kt, K.dw0 * kt + K.w0 * dkt);
  const double iom = rcp(1.0 - w.v);
  const DualD ratio(w.v * iom, w.d * iom * iom);
  St = K.Sinf + K.dS * ratio;
  if (t == 0) St = DualD(K.c2 * K.P0 + K.r, 0.0);
  gt = rcp(St);
  rt = DualD(K.a * K.r * gt.v, K.a * K.r * gt.d);
}

struct LagAdam {
  int T, N, D;
  const float* y;
  const double* rconst;
  const double *m0, *S0, *A, *C, *Q;
  const double* ck;            // [N][kLaL]
  float* yT;                   // [N][pitch] chain-major copies (pitch = T rounded up to 16 floats: a lane's 8-frame
                               // blocks are 32-byte aligned), made by a chain's wave the first time it needs one
  const int32_t* kp_block;
  double lr, lo, hi, tol;
  int cap, n_iters;
  double rho_max;
  int head_frames;             // 0: by the pole | 64, 128, 256 (tests, A/B)
  double *state, *s_keypoint, *nll, *dnll;
  int32_t* n_active;
};

// what a chain's two waves keep for the whole search (lane j: lags 4 j .. 4 j + 3).  The HEAD - the frames evaluated one by
// one from the prior, before the lag sums take over - is 64, 128 or 256 frames long (1, 2 or 4 per lane), chosen per
// evaluation from the pole: the variance's transient w_t = w_0 kappa^t must be dead at its end, and the searches spend
// most of their iterations at poles below 0.7 where 64 frames do.  What depends on the head's length H - the inputs that
// follow the head's frames, the lag sums over t >= H + 1 + k and the first inputs of that region - is kept in registers
// for H = 256 and in LDS rows (`sets`) for the two shorter heads.
struct LaLane {
  double un[4];                // HEAD wave: u_{t+1} for the lane's frames t = 4 lane + f of the 256-frame head
  double c2k[4], uF[4];        // LAG wave: lag sums (c_0, 2 c_k) of the region from F = 257 and its first inputs u_{F+k}
  double ut[4];                // LAG wave: last inputs u_{T-1-k}
  double e0;                   // innovation of frame 0: y_0 - c m_0
};
constexpr int kLaStage = 576;               // u_0 .. u_575 of a chain while its sets are formed (lags up to 255 past 256)
constexpr int kLaSetRows = 19;              // H = 64: un[1] c2k[4] uF[4] | H = 128: un[2] c2k[4] uF[4]; [row][lane]
constexpr int kLaDynPerChain = kLaStage + kLaSetRows * 64;

// ---- the evaluation, on the TWO waves of a chain (on different SIMDs: float64 instructions issue at half the float32
// rate, the evaluation is ~350 of them and one wave per chain left half of the chip's SIMDs idle).  The HEAD wave walks
// the head's frames and takes log S_inf; the LAG wave forms the lag polynomial, Z, the tail sum and the head's
// log-determinant.  What each hands over is a few numbers; the loss is
//     v  = [ Hv + Bv + E (Zv + ca E) ] / 2,   dv = [ Hd + Bd + E (Zd + ca' E) + E' (Zv + 2 ca E) ] / 2
// with (E, E') the innovation of frame H and its derivative, (Hv, Hd) the head's quadratic term + T log(2 pi S_inf),
// (Bv, Bd) = ca x polynomial - ca rho^2 Dl^2 + log-determinant of the head, (Zv, Zd) the coefficients of E from
// 2 ca rho E Z, and ca = g / (1 - rho^2).
struct LaHeadOut {
  double Hv, Hd, Ev, Ed;
};
struct LaLagOut {
  double Bv, Bd, Zv, Zd, cav, cad;
};
__device__ __forceinline__ DualD la_combine(const LaHeadOut& h, const LaLagOut& l) {
  const double v = 0.5 * (h.Hv + l.Bv + h.Ev * (l.Zv + l.cav * h.Ev));
  const double dv = 0.5 * (h.Hd + l.Bd + h.Ev * (l.Zd + l.cad * h.Ev) + h.Ed * (l.Zv + 2.0 * l.cav * h.Ev));
  return DualD(v, dv);
}

// the head's frames NF lane .. NF lane + NF - 1 (un: the inputs that follow them); H = 64 NF
template <int NF>
__device__ __forceinline__ LaHeadOut la_head(const LaConst& K, const double (&un)[4], double e0, int T, int lane) {
  // rho^(NF lane) by squaring; kappa^(NF lane) is its square (kappa = rho^2)
  const double rho = K.rho.v, rho2 = rho * rho;
  const double hb = la_pow_bits<6>(NF == 1 ? rho : NF == 2 ? rho2 : rho2 * rho2, (unsigned)lane);
  // S_t = S_inf (1 - w_{t+1}) / (1 - w_t): the frame's 1 / S_t and pole are g and rho times m_t = (1 - w_t) / (1 - w_{t+1}),
  // d log m_t = h_{t+1} - h_t with h_t = dw_t / (1 - w_t) - one reciprocal per frame (and one more per lane)
  double om[NF + 1], h[NF + 1], inv[NF + 1];
  {
    const double c2w = 2.0 * K.w0 * K.dlr;
    const double base = K.dw0 + (double)(NF * lane) * c2w;
    double kf = hb * hb;
#pragma unroll
    for (int j = 0; j <= NF; ++j) {
      om[j] = 1.0 - K.w0 * kf;
      if (j == 0) om[0] = lane == 0 ? K.om0 : om[0];
      inv[j] = rcp(om[j]);
      h[j] = kf * (base + (double)j * c2w) * inv[j];
      kf *= K.kap;
    }
  }
  const double ar = K.a * K.r;
  DualD gt[NF], rt[NF];
  LaAff el{DualD(1.0), DualD(0.0)};
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const double m = om[f] * inv[f + 1];
    double dd = K.dlr + (h[f + 1] - h[f]);                    // d log g_t = d log rho_t
    if (f == 0) dd = lane == 0 ? 0.0 : dd;                    // frame 0: S_0 = c^2 P_0 + r does not depend on s
    const double gv = K.g.v * m;
    gt[f] = DualD(gv, gv * dd);
    rt[f] = DualD(ar * gt[f].v, ar * gt[f].d);
    el = la_then(el, LaAff{rt[f], DualD(un[f])});
  }
  la_scan(el, lane);
  const DualD e_first(e0);
  const DualD e_out = el.A * e_first + el.b;
  // the innovation entering the lane's first frame: the previous lane's e_out (wave_shr:1; lane 0 keeps `old`)
  DualD e(la_dpp<0x138>(e_first.v, e_out.v), la_dpp<0x138>(0.0, e_out.d));
  DualD quad(0.0);
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    quad = quad + gt[f] * e * e;
    if (f + 1 < NF) e = rt[f] * e + DualD(un[f]);
  }
  LaHeadOut o;
  o.Ev = la_readlane(e_out.v, 63);                            // innovation of frame H
  o.Ed = la_readlane(e_out.d, 63);
  o.Hv = la_wave_sum(quad.v) + (double)T * (kLog2Pi + log(K.Sinf.v));
  o.Hd = la_wave_sum(quad.d) + (double)T * K.Sinf.d * K.g.v;
  return o;
}

// lags 4 lane .. 4 lane + 3 of the region behind a head of H = 64 NF frames (c2k, uF: that region's sums and first
// inputs; ut: the last inputs).  k64 = kappa^64.
template <int NF>
__device__ __forceinline__ LaLagOut la_lags(const LaConst& K, const double (&c2k)[4], const double (&uF)[4],
                                            const double (&ut)[4], double k64, int lane) {
  constexpr int H = 64 * NF;
  const double rho = K.rho.v, rho2 = rho * rho;
  const double pbase = la_pow_bits<6>(rho2 * rho2, (unsigned)lane);
  const double pf[4] = {pbase, pbase * rho, pbase * rho2, pbase * rho2 * rho};
  double sp = 0.0, spk = 0.0, sz = 0.0, szk = 0.0, sd = 0.0, sdk = 0.0;
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const double kk = (double)(4 * lane + f) * pf[f];
    sp += pf[f] * c2k[f];  spk += kk * c2k[f];
    sz += pf[f] * uF[f];   szk += kk * uF[f];
    sd += pf[f] * ut[f];   sdk += kk * ut[f];
  }
  // the loss is linear in the lag polynomial and in Z with coefficients known before any sum is taken: those two - and
  // their derivatives - are combined in the lane; only the last inputs' sum Dl enters squared and travels alone
  const DualD one(1.0);
  const DualD iom = rcp(one - K.rho * K.rho);
  const DualD ca = K.g * iom;                                  // coefficient of the lag polynomial
  const DualD M = DualD(2.0) * ca * K.rho;                     // ... of E Z  (2 g E X1, X1 = rho Z / (1 - rho^2))
  const double bv = la_wave_sum(ca.v * sp), bd = la_wave_sum(ca.d * sp + ca.v * K.dlr * spk);
  const double zv = la_wave_sum(M.v * sz), zd = la_wave_sum(M.d * sz + M.v * K.dlr * szk);
  const DualD Dl(la_wave_sum(sd), la_wave_sum(sdk) * K.dlr);
  const DualD Nn = ca * K.rho * K.rho * Dl * Dl;               // (rho^(2 (T - H)) < 1e-60 here)
  // the head's log-determinant in closed form: prod_{t=1}^{H-1} S_t / S_inf telescopes to (1 - w_H) / (1 - w_1);
  // frame 0 is S_0 itself
  double kH = k64;
  if (NF >= 2) kH *= kH;
  if (NF == 4) kH *= kH;
  const double w1 = K.w0 * K.kap, dw1 = K.kap * (K.dw0 + K.w0 * 2.0 * K.dlr);
  const double wH = K.w0 * kH, dwH = kH * (K.dw0 + K.w0 * (double)H * 2.0 * K.dlr);
  const double i1 = rcp(1.0 - w1), iH = rcp(1.0 - wH);
  const double S0 = K.c2 * K.P0 + K.r;
  const double logdet_head = log(S0 * K.g.v * (1.0 - wH) * i1);          // sum_{t < H} log(S_t / S_inf)
  const double dlogdet_head = dw1 * i1 - dwH * iH - K.Sinf.d * K.g.v;    // its derivative (frame 0: -d log S_inf)
  LaLagOut o;
  o.Bv = bv - Nn.v + logdet_head;
  o.Bd = bd - Nn.d + dlogdet_head;
  o.Zv = zv;
  o.Zd = zd;
  o.cav = ca.v;
  o.cad = ca.d;
  return o;
}

// the chain's NLL and d / d log s by streaming its own frames (any pole): lane = time chunk of the chain-major copy
__device__ __forceinline__ DualD la_nll_stream(const LaConst& K, const float* __restrict__ yc, int T, double e0, int lane) {
  const int cl = ((T + 63) / 64 + 7) / 8 * 8;            // (whole 8-frame blocks: aligned 16-byte loads)
  const int t0 = min(lane * cl, T), t1 = min(T, t0 + cl);
  // kappa^t0 by squaring
  double kt = 1.0;
  {
    double b = K.kap;
    for (unsigned e = (unsigned)t0; e; e >>= 1) {
      if (e & 1u) kt *= b;
      b *= b;
    }
  }
  // the innovations of the lane's frames as alpha e_in + beta (e_in: the innovation of its first frame): sums of
  // g_t (alpha e_in + beta)^2 as three dual numbers.  Once w_t = w_0 kappa^t is below 1e-18 for every lane the variance
  // IS the fixed point: constants instead of two reciprocals per frame, and the weights g come out of the sums.
  DualD al(1.0), be(0.0), Qaa(0.0), Qab(0.0), Qbb(0.0), Saa(0.0), Sab(0.0), Sbb(0.0);
  double pprod = 1.0, dlog = 0.0;
  int n_steady = 0;
  constexpr int NB = 8;
  float buf[NB + 1], nbuf[NB + 1];
  // rows t .. t + NB: two aligned 16-byte loads (the copy's pitch is padded: a block past the end reads the pad, whose
  // contents are never used - every use below is selected by t + q + 1 < T) and the row after them
  const int tpad = (T + 15) / 16 * 16;
  auto fetch = [&](int t, float (&b)[NB + 1]) {
    const int tb = min(t, tpad - NB);
    const f32x4 lo = *reinterpret_cast<const f32x4*>(yc + tb), hi = *reinterpret_cast<const f32x4*>(yc + tb + 4);
    b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3];
    b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = hi[3];
    b[8] = yc[min(tb + NB, tpad - 1)];
  };
  const double w0a = fabs(K.w0);
  if (t0 < t1) fetch(t0, nbuf);
  for (int t = t0; t < t1; t += NB) {
#pragma unroll
    for (int q = 0; q <= NB; ++q) buf[q] = nbuf[q];
    fetch(t + NB, nbuf);                                       // the next block's rows travel beside this block's arithmetic
    const bool steady = __all(w0a * kt < 1e-18) != 0 && t > 0;
    const bool al_dead = __all(al.v == 0.0 && al.d == 0.0) != 0;
    if (steady && al_dead && __all(t + NB <= t1 && t + NB < T || t >= t1) != 0) {
      // whole blocks in the steady state with the lane's first innovation forgotten: beta and sum beta^2 alone, written
      // out (five float64 FMAs and a conversion per frame; the operators above spent 20 instructions)
      if (t < t1) {
        double bv = be.v, bd = be.d, sv = 0.0, sd2 = 0.0, yq = (double)buf[0];
        const double rv = K.rho.v, rd = K.rho.d;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          sv = __builtin_fma(bv, bv, sv);
          sd2 = __builtin_fma(bv, bd, sd2);
          const double yn = (double)buf[q + 1];
          const double u = __builtin_fma(-K.a, yq, yn);
          yq = yn;
          const double nbd = __builtin_fma(rv, bd, rd * bv);
          bv = __builtin_fma(rv, bv, u);
          bd = nbd;
        }
        be = DualD(bv, bd);
        Sbb = Sbb + DualD(sv, 2.0 * sd2);
        n_steady += NB;
      }
    } else if (steady && al_dead) {
      // ... ragged blocks of the same regime
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        if (t + q < t1) {
          Sbb = Sbb + be * be;
          const double u = t + q + 1 < T ? (double)buf[q + 1] - K.a * (double)buf[q] : 0.0;
          be = K.rho * be + DualD(u);
          ++n_steady;
        }
      }
    } else if (steady) {
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        if (t + q < t1) {
          Saa = Saa + al * al;
          Sab = Sab + al * be;
          Sbb = Sbb + be * be;
          const double u = t + q + 1 < T ? (double)buf[q + 1] - K.a * (double)buf[q] : 0.0;
          al = K.rho * al;
          be = K.rho * be + DualD(u);
          if (fabs(al.v) < 1e-20) al = DualD(0.0);   // (its terms are below 1e-20 of the sums)
          ++n_steady;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int tt = t + q;
        if (tt < t1) {
          DualD St, gt, rt;
          la_frame(K, tt, kt, St, gt, rt);
          kt *= K.kap;
          pprod *= St.v * K.g.v;
          dlog += St.d * gt.v;
          Qaa = Qaa + gt * al * al;
          Qab = Qab + gt * al * be;
          Qbb = Qbb + gt * be * be;
          const double u = tt + 1 < T ? (double)buf[q + 1] - K.a * (double)buf[q] : 0.0;
          al = rt * al;
          be = rt * be + DualD(u);
          if (fabs(al.v) < 1e-20) al = DualD(0.0);   // (its terms are below 1e-20 of the sums)
        }
      }
    }
  }
  Qaa = Qaa + K.g * Saa;
  Qab = Qab + K.g * Sab;
  Qbb = Qbb + K.g * Sbb;
  dlog += (double)n_steady * K.Sinf.d * K.g.v;
  LaAff el{al, be};
  la_scan(el, lane);
  const DualD e_first(e0);
  const DualD e_out = el.A * e_first + el.b;
  DualD e(__shfl_up(e_out.v, 1), __shfl_up(e_out.d, 1));
  if (lane == 0) e = e_first;
  DualD contrib = Qaa * e * e + DualD(2.0) * Qab * e + Qbb;
  if (t0 >= T) contrib = DualD(0.0);
  const double q_v = la_wave_sum(contrib.v), q_d = la_wave_sum(contrib.d);
  const double dl_sum = la_wave_sum(dlog), pp = la_wave_prod(pprod);
  const double v = 0.5 * ((double)T * (kLog2Pi + log(K.Sinf.v)) + log(pp) + q_v);
  return DualD(v, 0.5 * (dl_sum + q_d));
}

// Diagnostic build only (-DEKS_LAG_STAMPS, tools/lag_stamps.py): lane 0 of every chain's wave adds up the shader-clock
// cycles of an iteration's sections (constants | evaluation | exchange + barrier | step) and its iterations.
#ifdef EKS_LAG_STAMPS
__device__ unsigned long long g_lag_stamps[1024][8];
#define LAG_STAMP(i)                                                     \
  do {                                                                   \
    const unsigned long long now_ = __builtin_readcyclecounter();        \
    lag_acc_[i] += now_ - lag_t_;                                        \
    lag_t_ = now_;                                                       \
  } while (0)
#else
#define LAG_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(128 * kLaMaxD) void lag_adam_kernel(LagAdam P) {
  // block = keypoint; wave 2 d + h: chain d, h = 0 the HEAD wave, h = 1 the LAG wave (la_head / la_lags)
  __shared__ double xch[2][kLaMaxD][10];
  const int lane = threadIdx.x, D = P.D;
  const int d = __builtin_amdgcn_readfirstlane((int)threadIdx.y >> 1), half = __builtin_amdgcn_readfirstlane((int)threadIdx.y & 1);
  const int k = blockIdx.x;
  const int kb = P.kp_block[k];
  const int T = P.T, N = P.N;
  double* st = P.state + (size_t)kb * kAdamState;
  double u = st[0], mom = st[1], vel = st[2], prev = st[3], iters = st[4], done = st[5];
  if (!(done == 0.0 && iters < (double)P.cap)) return;          // block-uniform
  const int n = k * D + d;
  const size_t dd = (size_t)k * D * D + (size_t)d * (D + 1);
  const double a = P.A[dd], c = P.C[dd], q = P.Q[dd], r = P.rconst[n], P0 = P.S0[dd], m0 = P.m0[(size_t)k * D + d];
  // ---- what does not change between iterations
  extern __shared__ double la_dyn[];                                  // [D][kLaStage] inputs, then [D][19][64] sets
  double* ush = la_dyn + (size_t)d * kLaStage;
  double* sets = la_dyn + (size_t)D * kLaStage + (size_t)d * kLaSetRows * 64;
  LaLane L;
  {
    const float* yn = P.y + n;
    // the chain's first inputs u_t = y_t - a y_{t-1}, t < 576, through LDS (u_0 is never used); its two waves share them
    for (int i = half; i < kLaStage / 64; i += 2) {
      const int t = 64 * i + lane;
      const float y1 = yn[(size_t)t * N], y0 = yn[(size_t)max(t - 1, 0) * N];
      ush[t] = (double)y1 - a * (double)y0;
    }
    L.e0 = (double)yn[0] - c * m0;
#pragma unroll
    for (int f = 0; f < 4; ++f) L.c2k[f] = L.uF[f] = L.ut[f] = L.un[f] = 0.0;
    if (half == 1) {
      const double* ckn = P.ck + (size_t)n * kLaL + 4 * lane;
#pragma unroll
      for (int f = 0; f < 4; ++f) L.c2k[f] = ckn[f];
      float ht[5];
#pragma unroll
      for (int f = 0; f < 5; ++f) ht[f] = yn[(size_t)max(T - 1 - 4 * lane - f, 0) * N];   // rows T - 1 - 4 lane, downwards
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int tt = T - 1 - 4 * lane - f;
        L.ut[f] = tt >= kLaF ? (double)ht[f] - a * (double)ht[f + 1] : 0.0;
      }
    }
  }
  __syncthreads();
  if (half == 0) {
    // the inputs that follow the frames of a head of 256 / 64 / 128 frames
#pragma unroll
    for (int f = 0; f < 4; ++f) L.un[f] = ush[4 * lane + f + 1];
    sets[0 * 64 + lane] = ush[lane + 1];                              // H = 64: un
    sets[9 * 64 + lane] = ush[2 * lane + 1];                          // H = 128: un[0], un[1]
    sets[10 * 64 + lane] = ush[2 * lane + 2];
  } else {
    // the lag sums of a region that starts at F = H + 1 are the cached ones (F = 257) plus the products of the frames in
    // between,  c_k(H) = c_k(256) + sum_{t = H + 1 + k}^{256 + k} u_t u_{t-k}   (u_{t-k} = u_{H+1+i}: the same for every
    // lane): 128 + 64 products per lag, once per search
#pragma unroll
    for (int f = 0; f < 4; ++f) L.uF[f] = ush[kLaF + 4 * lane + f];
    double acc128[4] = {0.0, 0.0, 0.0, 0.0}, acc64[4] = {0.0, 0.0, 0.0, 0.0};
    {
      const double* ua = ush + 129 + 4 * lane;                       // u_{129 + k + i}, k = 4 lane + f
      double w0 = ua[0], w1 = ua[1], w2 = ua[2];
#pragma unroll 8
      for (int i = 0; i < 128; ++i) {
        const double w3 = ua[i + 3], ub = ush[129 + i];
        acc128[0] += w0 * ub; acc128[1] += w1 * ub; acc128[2] += w2 * ub; acc128[3] += w3 * ub;
        w0 = w1; w1 = w2; w2 = w3;
      }
    }
    {
      const double* ua = ush + 65 + 4 * lane;
      double w0 = ua[0], w1 = ua[1], w2 = ua[2];
#pragma unroll 8
      for (int i = 0; i < 64; ++i) {
        const double w3 = ua[i + 3], ub = ush[65 + i];
        acc64[0] += w0 * ub; acc64[1] += w1 * ub; acc64[2] += w2 * ub; acc64[3] += w3 * ub;
        w0 = w1; w1 = w2; w2 = w3;
      }
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const double mult = 4 * lane + f == 0 ? 1.0 : 2.0;             // (c_0, 2 c_k)
      const double c128 = L.c2k[f] + mult * acc128[f];
      sets[(11 + f) * 64 + lane] = c128;
      sets[(1 + f) * 64 + lane] = c128 + mult * acc64[f];
      sets[(15 + f) * 64 + lane] = ush[129 + 4 * lane + f];
      sets[(5 + f) * 64 + lane] = ush[65 + 4 * lane + f];
    }
  }
  // (each wave reads back only what it wrote itself - the head wave the un rows, the lag wave the others - so no barrier)
  bool have_copy = false;
  float* yc = P.yT + (size_t)n * (((size_t)T + 15) / 16 * 16);
  double b1t = adam_pow_count(0.9, iters), b2t = adam_pow_count(0.999, iters);
#ifdef EKS_LAG_STAMPS
  unsigned long long lag_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lag_t_ = __builtin_readcyclecounter();
  const unsigned long long lag_start_ = lag_t_;
#endif
  for (int it = 0; it < P.n_iters; ++it) {
    LAG_STAMP(4);
    const double s = exp(fmin(fmax(u, P.lo), P.hi));
    // (both waves of a chain run the SAME instructions up to the branch on `half`: the same constants, the same head)
    const LaConst K = la_const(a, c, q, r, P0, s);
    LAG_STAMP(0);
    double* mine = xch[it & 1][d];
    if (fabs(K.rho.v) <= P.rho_max) {                            // (wave-uniform: one chain per pair of waves)
      // the head: as short as the variance's transient allows (|w_H| H = |w_0| kappa^H H below 1e-17 at its end)
      const double kap2 = K.kap * K.kap, k4 = kap2 * kap2, k16 = (k4 * k4) * (k4 * k4), k64 = (k16 * k16) * (k16 * k16);
      const double aw = fabs(K.w0);
      int nf = 4;
      if (aw * k64 * k64 * 128.0 <= 1e-17) nf = 2;
      if (aw * k64 * 64.0 <= 1e-17) nf = 1;
      if (P.head_frames) nf = P.head_frames / 64;                // (tests)
      nf = __builtin_amdgcn_readfirstlane(nf);
      if (half == 0) {
        LaHeadOut o;
        if (nf == 1) {
          const double un1[4] = {sets[0 * 64 + lane], 0.0, 0.0, 0.0};
          o = la_head<1>(K, un1, L.e0, T, lane);
        } else if (nf == 2) {
          const double un2[4] = {sets[9 * 64 + lane], sets[10 * 64 + lane], 0.0, 0.0};
          o = la_head<2>(K, un2, L.e0, T, lane);
        } else {
          o = la_head<4>(K, L.un, L.e0, T, lane);
        }
        if (lane == 0) {
          mine[0] = o.Hv; mine[1] = o.Hd; mine[2] = o.Ev; mine[3] = o.Ed;
        }
      } else {
        LaLagOut o;
        if (nf == 4) {
          o = la_lags<4>(K, L.c2k, L.uF, L.ut, k64, lane);
        } else {
          const int r0 = nf == 1 ? 1 : 11;
          double c2k[4], uF[4];
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            c2k[f] = sets[(r0 + f) * 64 + lane];
            uF[f] = sets[(r0 + 4 + f) * 64 + lane];
          }
          o = nf == 1 ? la_lags<1>(K, c2k, uF, L.ut, k64, lane) : la_lags<2>(K, c2k, uF, L.ut, k64, lane);
        }
        if (lane == 0) {
          mine[4] = o.Bv; mine[5] = o.Bd; mine[6] = o.Zv; mine[7] = o.Zd; mine[8] = o.cav; mine[9] = o.cad;
        }
      }
    } else if (half == 0) {
      // the pole is beyond the lags' range: the head wave streams the chain's frames (float64, exact); nothing from lags
      if (!have_copy) {
        for (int t = lane; t < T; t += 64) yc[t] = P.y[(size_t)t * N + n];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        have_copy = true;
      }
      const DualD v = la_nll_stream(K, yc, T, L.e0, lane);
      if (lane == 0) {
        mine[0] = 2.0 * v.v; mine[1] = 2.0 * v.d; mine[2] = 0.0; mine[3] = 0.0;
      }
    } else {
      if (lane == 0) {
#pragma unroll
        for (int i = 4; i < 10; ++i) mine[i] = 0.0;
      }
    }
    LAG_STAMP(1);
    __syncthreads();
    double Lv = 0.0, g = 0.0;
    for (int dq = 0; dq < D; ++dq) {
      const double* x = xch[it & 1][dq];
      const DualD v = la_combine(LaHeadOut{x[0], x[1], x[2], x[3]}, LaLagOut{x[4], x[5], x[6], x[7], x[8], x[9]});
      Lv += v.v;
      g += v.d;
    }
    // eks/core.py:650: a non-finite loss is 1e12 with zero gradient; then the step and the stop rule of :652-681
    LAG_STAMP(2);
    const bool fin = isfinite(Lv);
    Lv = fin ? Lv : 1e12;
    g = fin ? g : 0.0;
    const double g_raw = g;
    if (u < P.lo || u > P.hi) g = 0.0;
    g *= P.lr;
    const double cnt = iters + 1.0;
    mom = 0.9 * mom + 0.1 * g;
    vel = 0.999 * vel + 0.001 * g * g;
    b1t *= 0.9;                                                // (0.9^cnt, 0.999^cnt: carried, not recomputed)
    b2t *= 0.999;
    const double mhat = mom * rcp(1.0 - b1t);                   // (Newton reciprocals: within an ulp of the quotients)
    const double vhat = vel * rcp(1.0 - b2t);
    u = u - mhat * rcp(sqrt(vhat) + 1e-8);
    // the stop rule's logarithm only when it can matter: |log p| <= (|e| + 1) ln 2 for p = m 2^e, so a change of the
    // loss above tol times that bound cannot stop the search (all but the last one or two iterations)
    bool stop = false;
    if (isfinite(prev)) {
      const double dL = fabs(Lv - prev), pc = fmax(prev, 1e-12);
      const int ex = ((__double2hiint(pc) >> 20) & 0x7ff) - 1023;
      const double bound = (double)(abs(ex) + 1) * 0.6931471805599453;
      if (!(P.tol >= 0.0) || dL < P.tol * bound + 1e-6) stop = dL < P.tol * fabs(log(pc)) + 1e-6;
    }
    prev = Lv;
    iters = cnt;
    done = stop ? 1.0 : 0.0;
    if (lane == 0 && d == 0 && half == 0) {
      P.nll[k] = Lv;
      P.dnll[k] = g_raw;
    }
    LAG_STAMP(3);
#ifdef EKS_LAG_STAMPS
    lag_acc_[5] += 1;
#endif
    if (stop || !(iters < (double)P.cap)) break;
  }
#ifdef EKS_LAG_STAMPS
  if (lane == 0 && 2 * n + half < 1024) {
    lag_acc_[6] = __builtin_readcyclecounter() - lag_start_;
    lag_acc_[7] = lag_start_;
    for (int i = 0; i < 8; ++i) g_lag_stamps[2 * n + half][i] = lag_acc_[i];
  }
#endif
  if (lane == 0 && d == 0 && half == 0) {
    st[0] = u; st[1] = mom; st[2] = vel; st[3] = prev; st[4] = iters; st[5] = done;
    P.s_keypoint[k] = exp(fmin(fmax(u, P.lo), P.hi));
    if (done == 0.0 && iters < (double)P.cap) atomicAdd(P.n_active, 1);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
bool diag_lag_adam_ok(int T, int K, int D, int n_blocks) {
  return n_blocks == K && T >= kLaMinT && D >= 1 && D <= kLaMaxD && !knob_int(KNOB_ADAM_STREAM, 0);
}

static void lag_geometry(int T, int N, int* nch_out, int* cl_out) {
  const long ntile = (N + 63) / 64;
  const long frames = T - kLaF;
  long nch = (256 + ntile - 1) / ntile;                // one 16-wave block per compute unit (its ring fills the LDS)
  if (nch > (frames + 63) / 64) nch = (frames + 63) / 64;
  if (nch < 1) nch = 1;
  long cl = ((frames + nch - 1) / nch + 63) / 64 * 64;
  // (a chunk's rows are addressed with 32-bit offsets from its first row: very wide sessions take more, shorter chunks)
  while (cl > 64 && (cl + kRgHist + 160) * (long)N * 4 >= (1L << 31)) cl = (cl / 2 + 63) / 64 * 64;
  nch = (frames + cl - 1) / cl;
  *nch_out = (int)nch;
  *cl_out = (int)cl;
}

// [part : nch x 256 x N doubles][ck : N x 256 doubles][yT : N x (T rounded up to 16) floats]
size_t diag_lag_adam_workspace_bytes(int T, int N) {
  if (T < kLaMinT) return 0;
  int nch, cl;
  lag_geometry(T, N, &nch, &cl);
  return align_up((size_t)nch * kLaL * N * sizeof(double), 256) + align_up((size_t)N * kLaL * sizeof(double), 256) +
         align_up((size_t)N * (((size_t)T + 15) / 16 * 16) * sizeof(float), 256);
}

struct LagWs {
  double *part, *ck;
  float* yT;
  int nch, cl;
};
static int lag_ws(int T, int N, void* ws, size_t ws_bytes, LagWs* W) {
  if (ws_bytes < diag_lag_adam_workspace_bytes(T, N)) return EKS_ERR_WORKSPACE;
  lag_geometry(T, N, &W->nch, &W->cl);
  if ((long)(W->cl + kRgHist + 160) * N * 4 >= (1L << 31)) return EKS_ERR_UNSUPPORTED;   // 32-bit row offsets of a chunk
  char* p = static_cast<char*>(ws);
  W->part = reinterpret_cast<double*>(p);
  p += align_up((size_t)W->nch * kLaL * N * sizeof(double), 256);
  W->ck = reinterpret_cast<double*>(p);
  p += align_up((size_t)N * kLaL * sizeof(double), 256);
  W->yT = reinterpret_cast<float*>(p);
  return EKS_OK;
}

// the pass over y: lag sums into the workspace (eks_adam_prepare, or the first thing an eks_adam_run call does).  With an
// optimiser state (F) the tiles none of whose keypoints still runs are skipped.
int diag_lag_sums(const eks_dims_t& d, const float* y, const double* A, const AdamFuse* F, void* ws, size_t ws_bytes,
                  hipStream_t st) {
  const int T = d.n_frames, D = d.state_dim, N = d.n_keypoints * D;
  LagWs W;
  const int rc = lag_ws(T, N, ws, ws_bytes, &W);
  if (rc != EKS_OK) return rc;
  const int ntile = (N + 63) / 64;
  {
    ProfScope ps("lag_sums", st);
    const LagPre P{T, N, D, ntile, W.nch, W.cl, y, A, W.part, F ? F->state : nullptr, F ? F->kp_block : nullptr,
                   F ? F->cap : 0};
    const dim3 grid((unsigned)(ntile * W.nch)), block(64 * kMfWaves);
    if (d.flags & EKS_FLAG_UNIT_AC) hipLaunchKernelGGL(lag_sums_kernel<true>, grid, block, 0, st, P);
    else hipLaunchKernelGGL(lag_sums_kernel<false>, grid, block, 0, st, P);
  }
  {
    ProfScope ps("lag_reduce", st);
    const long lanes = (long)N * kLaL;
    hipLaunchKernelGGL(lag_reduce_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, N, W.nch, W.part, W.ck);
  }
  return hip_status(hipGetLastError());
}

int diag_lag_adam(const eks_dims_t& d, const float* y, const double* rconst, const DiagModel& M, int n_iters, double* nll,
                  double* dnll, const AdamFuse& F, void* ws, size_t ws_bytes, hipStream_t st) {
  const int T = d.n_frames, D = d.state_dim, K = d.n_keypoints, N = K * D;
  LagWs W;
  int rc = lag_ws(T, N, ws, ws_bytes, &W);
  if (rc != EKS_OK) return rc;
  if (!(d.flags & EKS_FLAG_ADAM_PREPARED)) {
    rc = diag_lag_sums(d, y, M.A, &F, ws, ws_bytes, st);
    if (rc != EKS_OK) return rc;
  }
  ProfScope ps("lag_adam", st);
  static const double rho_max = lag_adam_rho_max();
  const int rm = knob_int(KNOB_ADAM_LAG_RHO_PPM, -1);          // (tests: the pole beyond which a chain streams)
  int head = knob_int(KNOB_ADAM_LAG_HEAD, 0);                  // (tests, A/B: the head's length for every evaluation)
  if (head != 64 && head != 128 && head != 256) head = 0;
  const LagAdam P{T, N, D, y, rconst, M.m0, M.S0, M.A, M.C, M.Q, W.ck, W.yT, F.kp_block, F.lr, F.lo, F.hi, F.tol, F.cap,
                  n_iters, rm >= 0 ? 1e-6 * rm : rho_max, head, F.state, F.s_keypoint, nll, dnll, F.n_active_cur};
  hipLaunchKernelGGL(lag_adam_kernel, dim3((unsigned)K), dim3(64, 2 * D), (size_t)D * kLaDynPerChain * sizeof(double), st, P);
  return hip_status(hipGetLastError());
}

}  // namespace eks

EKS_DEFINE_TOUCH(lag_adam)
#ifdef EKS_LAG_STAMPS
extern "C" int eks_debug_lag_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(eks::g_lag_stamps), sizeof(eks::g_lag_stamps));
}
#endif
