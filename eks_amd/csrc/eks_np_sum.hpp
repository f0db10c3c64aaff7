// numpy's float32 summation, operation by operation.  The optimiser's initial guess for s is
// round(numpy.nanstd(differences of the ensemble variances over the first 2 000 frames), 5) (reference
// eks/core.py:104-133) and its float32 rounding seeds the whole Adam trajectory, so the value has to be numpy's to
// the last bit; computing it on the host cost 2 ms per call at 256 keypoints (a quarter of the reference's default
// mode on the C3 session).  numpy sums a contiguous float32 run pairwise (numpy/_core/src/umath/loops_utils.h.src,
// @TYPE@_pairwise_sum): below 8 elements one accumulator from 0, up to 128 elements eight accumulators over blocks
// of eight combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the rest added one by one, above that the halves
// n2 = n/2 - (n/2) % 8 and n - n2 recursively.  nanstd (numpy/lib/_nanfunctions_impl.py: _nanvar) replaces NaNs by
// 0, divides the float32 sum by the count IN FLOAT64 and rounds to float32, subtracts, zeroes the NaN places,
// squares, sums the same way, divides the same way, takes the float32 square root.  tests/test_host_sim.py and
// tests/test_gpu_kernels.py compare against numpy bit for bit; a numpy that summed differently would show there.
#pragma once
#include "eks_math.hpp"

namespace eks {

constexpr int kNpBlock = 128;      // numpy's PW_BLOCKSIZE

// Separately rounded float32 operations: the library is built with -ffp-contract=fast, and a multiply fused into the
// following add is not numpy's arithmetic (a `#pragma clang fp contract(off)` does not reach into lambda bodies: the
// first build of this file had v_fmac_f32 in its squares and was off by one ulp in ~7 % of the rows).
// (__fmul_rn / __fadd_rn are plain operators to this compiler and fuse all the same; the product goes through an
// opaque v_mul_f32 instead - the only multiply in this file - so no add or subtract has one to fuse with.)
#if defined(__HIP_DEVICE_COMPILE__)
EKS_HD float np_add(float a, float b) { return a + b; }
EKS_HD float np_sub(float a, float b) { return a - b; }
EKS_HD float np_mul(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
#else
EKS_HD float np_add(float a, float b) { return a + b; }
EKS_HD float np_sub(float a, float b) { return a - b; }
EKS_HD float np_mul(float a, float b) { return a * b; }
#endif

// one leaf of the recursion: n <= 128 values v(0) ... v(n - 1)
template <typename F>
EKS_HD float np_leaf_sum(int n, F&& v) {
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res = np_add(res, v(i));
    return res;
  }
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = v(j);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = np_add(r[j], v(i + j));
  }
  float res = np_add(np_add(np_add(r[0], r[1]), np_add(r[2], r[3])), np_add(np_add(r[4], r[5]), np_add(r[6], r[7])));
  for (; i < n; ++i) res = np_add(res, v(i));
  return res;
}

// mean as numpy forms it: float32(float64(sum) / float64(count))
EKS_HD float np_divide_by_count(float sum, int count) { return (float)((double)sum / (double)count); }

}  // namespace eks
