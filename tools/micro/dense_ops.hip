// Micro-benchmark: latency of the general (D, O) path's building blocks in float64 on ONE wave (the dense
// kernels are depth-bound on BASELINE configs[3]: 4 keypoints): per-frame element step, per-frame filter step,
// RTS step, element composition with / without the log-likelihood term, apply / back.  Prints ns per op.
//   hipcc --offload-arch=gfx950 -O3 -I eks_amd/csrc tools/micro/dense_ops.hip -o dense_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include "eks_dense_lane.hpp"
using namespace eks;
constexpr int D = 3;

template <int MODE>
__global__ void k(double* out, int iters, double seed) {
  DElem<double, D> e = delem_identity<double, D>(), f = delem_identity<double, D>();
  Mat<double, D> sQ = mat_eye<double, D>(), F = mat_eye<double, D>();
  Vec<double, D> m = vec_zero<double, D>(), eta = vec_zero<double, D>();
  Mat<double, D> P = mat_eye<double, D>(), J = mat_eye<double, D>();
  for (int i = 0; i < D; ++i) { sQ.a[i][i] = 0.3 + seed * i; f.C.a[i][i] = 0.5 + seed; f.J.a[i][i] = 0.7; f.A.a[i][i] = 0.9; }
  Vec<double, D> h[4];
  for (int o = 0; o < 4; ++o) for (int i = 0; i < D; ++i) h[o].a[i] = 0.3 + 0.1 * o + 0.05 * i + seed * threadIdx.x;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {            // element step: predict + 4 scalar observations
      delem_predict(e, F, sQ, true);
      for (int o = 0; o < 4; ++o) delem_observe(e, h[o], 1.0 + 0.1 * o + acc * 1e-30, 0.5 + 0.1 * o, false);
    } else if (MODE == 1) {     // filter step
      P = mat_add(P, sQ);
      for (int o = 0; o < 4; ++o) {
        const Vec<double, D> u = mat_vec(P, h[o]);
        const double sigma = 0.5 + dot(h[o], u), g = 1.0 / sigma, d = 1.0 - dot(h[o], m), gd = g * d;
        for (int a = 0; a < D; ++a) { m.a[a] += u.a[a] * gd; for (int b = 0; b < D; ++b) P.a[a][b] -= u.a[a] * u.a[b] * g; }
      }
    } else if (MODE == 2) {     // RTS step (as dense_replay_chunk_obs)
      const Mat<double, D> Pf = f.C;
      const Mat<double, D> Pp = mat_symmetrize(mat_add(Pf, sQ));
      const Mat<double, D> Z = chol_solve_mat(chol_psd(Pp), Pf);
      Vec<double, D> dm; for (int a = 0; a < D; ++a) dm.a[a] = m.a[a] - f.b.a[a];
      const Vec<double, D> Gdm = mat_t_vec(Z, dm);
      for (int a = 0; a < D; ++a) m.a[a] = f.b.a[a] + Gdm.a[a] * 0.5;
      P = mat_symmetrize(mat_add(Pf, mat_mul(mat_mul_tn(Z, mat_sub(P, Pp)), Z)));
    } else if (MODE == 3) {     // composition (with ell)
      e = delem_combine(e, f);
      e.C.a[0][0] += 0.1;
    } else if (MODE == 4) {     // apply + back
      delem_apply(f, m, P);
      delem_back(f, eta, J);
    }
  }
  for (int i = 0; i < D; ++i) acc += e.b.a[i] + e.C.a[i][i] + m.a[i] + P.a[i][i] + eta.a[i] + J.a[i][i] + e.ell;
  out[threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name) {
  double* out; (void)hipMalloc(&out, 8 * 64);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, 10, 1e-3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, iters, 1e-3);
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  printf("%-40s %8.1f ns per op\n", name, ms * 1e6 / iters);
}
int main() {
  run<0>("element step (predict + 4 obs)");
  run<1>("filter step (predict + 4 obs)");
  run<2>("RTS step");
  run<3>("delem_combine (with ell)");
  run<4>("delem_apply + delem_back");
  return 0;
}
