// Wave-level movement of scan elements (gfx950 kernels only: eks_dense_wave.hip, eks_dense_wide.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "eks_dense_lane.hpp"

namespace eks {

// ---- an element moves between lanes as A (D x D), the upper triangles of C and J, b and eta ----------
template <int D>
__device__ __forceinline__ DElem<double, D> delem_shfl_up(const DElem<double, D>& e, int off) {
  DElem<double, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    o.b.a[i] = __shfl_up(e.b.a[i], off);
    o.eta.a[i] = __shfl_up(e.eta.a[i], off);
#pragma unroll
    for (int j = 0; j < D; ++j) o.A.a[i][j] = __shfl_up(e.A.a[i][j], off);
#pragma unroll
    for (int j = i; j < D; ++j) {
      o.C.a[i][j] = o.C.a[j][i] = __shfl_up(e.C.a[i][j], off);
      o.J.a[i][j] = o.J.a[j][i] = __shfl_up(e.J.a[i][j], off);
    }
  }
  o.ell = 0.0;
  return o;
}
template <int D>
__device__ __forceinline__ DElem<double, D> delem_shfl_down(const DElem<double, D>& e, int off) {
  DElem<double, D> o;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    o.b.a[i] = __shfl_down(e.b.a[i], off);
    o.eta.a[i] = __shfl_down(e.eta.a[i], off);
#pragma unroll
    for (int j = 0; j < D; ++j) o.A.a[i][j] = __shfl_down(e.A.a[i][j], off);
#pragma unroll
    for (int j = i; j < D; ++j) {
      o.C.a[i][j] = o.C.a[j][i] = __shfl_down(e.C.a[i][j], off);
      o.J.a[i][j] = o.J.a[j][i] = __shfl_down(e.J.a[i][j], off);
    }
  }
  o.ell = 0.0;
  return o;
}

}  // namespace eks
