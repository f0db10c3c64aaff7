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

// base^n for the bias corrections 1 - 0.9^t, 1 - 0.999^t (t = the iteration count, a whole number): by squaring, ~20
// dependent multiplications where the library's pow() was most of the step's 2.8 us on the loss kernel's critical path
__device__ __forceinline__ double adam_pow_count(double base, double cnt) {
  unsigned n = (unsigned)cnt;
  double r = 1.0;
  while (n) {
    if (n & 1u) r *= base;
    base *= base;
    n >>= 1;
  }
  return r;
}

__device__ __forceinline__ bool adam_block_running(const double* __restrict__ state, int b, int cap) {
  return state[(size_t)b * kAdamState + 5] == 0.0 && state[(size_t)b * kAdamState + 4] < (double)cap;
}

// one optimiser block b; returns whether it is still running afterwards
__device__ __forceinline__ bool adam_step_block(int b, const int32_t* __restrict__ offs,
                                                const int32_t* __restrict__ members,
                                                const double* __restrict__ nll,
                                                const double* __restrict__ dnll, double lr, double lo,
                                                double hi, double tol, int cap, double* state,
                                                double* s_keypoint) {
  auto put = [](double* p, double v) { *p = v; };
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
    const double mhat = mom / (1.0 - adam_pow_count(0.9, cnt));
    const double vhat = vel / (1.0 - adam_pow_count(0.999, cnt));
    u = u - mhat / (sqrt(vhat) + 1e-8);
    const bool stop = isfinite(prev) &&
                      fabs(L - prev) < tol * fabs(log(fmax(prev, 1e-12))) + 1e-6;
    prev = L;
    iters = cnt;
    done = stop ? 1.0 : 0.0;
    put(st + 0, u); put(st + 1, mom); put(st + 2, vel); put(st + 3, prev); put(st + 4, iters); put(st + 5, done);
  }
  const double s = exp(fmin(fmax(u, lo), hi));
  for (int i = offs[b]; i < offs[b + 1]; ++i) put(s_keypoint + members[i], s);
  return done == 0.0 && iters < (double)cap;
}

// The same step for an optimiser block that is ONE keypoint, its state, loss and gradient already in registers (the
// fused form at the end of the scalar-chain loss kernel: no dependent loads of offs / members / nll on its critical path).
struct AdamRegs {
  double u, mom, vel, prev, iters, done;
};
__device__ __forceinline__ AdamRegs adam_load(const double* state, int b) {
  const double* st = state + (size_t)b * kAdamState;
  return AdamRegs{st[0], st[1], st[2], st[3], st[4], st[5]};
}
template <bool AGENT>
__device__ __forceinline__ bool adam_step_single(int b, int k, AdamRegs a, double L, double g, double lr, double lo, double hi,
                                                 double tol, int cap, double* state, double* s_keypoint, double* s_out) {
  auto put = [](double* p, double v) {
    if (AGENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
  };
  double* st = state + (size_t)b * kAdamState;
  if (a.done == 0.0 && a.iters < (double)cap) {
    if (a.u < lo || a.u > hi) g = 0.0;
    g *= lr;
    const double cnt = a.iters + 1.0;
    a.mom = 0.9 * a.mom + 0.1 * g;
    a.vel = 0.999 * a.vel + 0.001 * g * g;
    const double mhat = a.mom / (1.0 - adam_pow_count(0.9, cnt));
    const double vhat = a.vel / (1.0 - adam_pow_count(0.999, cnt));
    a.u = a.u - mhat / (sqrt(vhat) + 1e-8);
    const bool stop = isfinite(a.prev) && fabs(L - a.prev) < tol * fabs(log(fmax(a.prev, 1e-12))) + 1e-6;
    a.prev = L;
    a.iters = cnt;
    a.done = stop ? 1.0 : 0.0;
    put(st + 0, a.u); put(st + 1, a.mom); put(st + 2, a.vel); put(st + 3, a.prev); put(st + 4, a.iters); put(st + 5, a.done);
  }
  const double s = exp(fmin(fmax(a.u, lo), hi));
  put(s_keypoint + k, s);
  *s_out = s;
  return a.done == 0.0 && a.iters < (double)cap;
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


// ---- two-parameter optimiser of the pupil smoother (eks/ibl_pupil_smoother.py:560-604), one chain k of n.
// state: {u_d, u_c, mom_d, mom_c, vel_d, vel_c, prev_loss, iters, done}.  nll == nullptr: initialisation only.
// Always emits the AR(1) dynamics of the next evaluation and the tangents d/du_d, d/du_c; returns whether the chain
// is still running afterwards.  Shared by the stand-alone step kernel (eks_misc.hip) and the fused finish + step at
// the end of the pupil loss (eks_dense_wave.hip).
__device__ __forceinline__ bool pupil_adam_step_chain(int k, int n, const double* __restrict__ latent_var,
                                                      const double* __restrict__ nll,
                                                      const double* __restrict__ dnll, double lr, double tol,
                                                      int cap, double* __restrict__ state, double* __restrict__ a,
                                                      double* __restrict__ q, double* __restrict__ da,
                                                      double* __restrict__ dq) {
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
  return done == 0.0 && iters < (double)cap;
}

}  // namespace eks
