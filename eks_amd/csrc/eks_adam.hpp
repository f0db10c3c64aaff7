// The optimiser iteration on u = log s with the reference's stop rule (eks/core.py:652-681, :509-549),
// shared by the stand-alone step kernels (eks_misc.hip) and the fused form at the end of the scalar-chain
// loss assembly (eks_diag_nll.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

namespace eks {

// state of one optimiser block (a set of keypoints sharing one s): {u, mom, vel, prev_loss, iters, done}
constexpr int kAdamState = 6;

__device__ __forceinline__ bool adam_block_running(const double* __restrict__ state, int b, int cap) {
  return state[(size_t)b * kAdamState + 5] == 0.0 && state[(size_t)b * kAdamState + 4] < (double)cap;
}

// one optimiser block b; returns whether it is still running afterwards
__device__ __forceinline__ bool adam_step_block(int b, const int32_t* __restrict__ offs,
                                                const int32_t* __restrict__ members,
                                                const double* __restrict__ nll,
                                                const double* __restrict__ dnll, double lr, double lo,
                                                double hi, double tol, int cap, double* __restrict__ state,
                                                double* __restrict__ s_keypoint) {
  double* st = state + (size_t)b * kAdamState;
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
  return done == 0.0 && iters < (double)cap;
}

// What the loss kernels of one Adam iteration need to (a) skip the keypoints whose optimiser block has
// stopped - a wave whose 64 chains are all finished returns at once, an assembly block of a finished
// keypoint too - and (b) apply the optimiser step at the end of the assembly when every block is a
// single keypoint (no separate launch).  All pointers are device pointers; kp_block[k] = block of
// keypoint k.  n_active_cur counts the blocks still running after this iteration (zeroed by the
// previous iteration or by the host), n_active_next is zeroed for the next one.
struct AdamFuse {
  const int32_t *offs, *members, *kp_block;
  double lr, lo, hi, tol;
  int cap;
  int step_in_kernel;        // 1: every block is one keypoint and the assembly applies the step itself
  double *state, *s_keypoint;
  int32_t *n_active_cur, *n_active_next;
};

}  // namespace eks
