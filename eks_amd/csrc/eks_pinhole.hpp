// Calibrated pinhole camera: world point -> pixel, with the analytic 2x3 Jacobian the extended
// Kalman filter linearises with.  Follows the projection of the reference's
// eks/multicam_smoother.py:814-868 (make_jax_projection_fn): world -> camera by (R, t), normalised
// coordinates, POLYNOMIAL radial factor 1 + k1 r^2 + ... + k6 r^12 (the reference does not use
// OpenCV's rational form), tangential and thin-prism terms, intrinsics with skew.  The reference
// obtains the Jacobian by jax.jacfwd; here it is written out.
//
// One camera = kCamDoubles doubles:
//   [0..8] R row-major, [9..11] t, [12] fx, [13] fy, [14] cx, [15] cy, [16] skew,
//   [17..30] distortion in OpenCV order k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4 tx ty (tx, ty are
//   ignored, eks/multicam_smoother.py:801-803), [31] padding.
#pragma once
#include "eks_math.hpp"

namespace eks {

constexpr int kCamDoubles = 32;

// uv = h(X); J[a][i] = d uv[a] / d X[i]
EKS_HD void pinhole_project_jac(const double* __restrict__ cam, const double X[3], double uv[2],
                                double J[2][3]) {
  const double* R = cam;
  const double Xc = R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + cam[9];
  const double Yc = R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + cam[10];
  const double Zc = R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + cam[11];
  const double fx = cam[12], fy = cam[13], cx = cam[14], cy = cam[15], skew = cam[16];
  const double k1 = cam[17], k2 = cam[18], p1 = cam[19], p2 = cam[20], k3 = cam[21], k4 = cam[22],
               k5 = cam[23], k6 = cam[24], s1 = cam[25], s2 = cam[26], s3 = cam[27], s4 = cam[28];
  const double iz = 1.0 / Zc;
  const double x = Xc * iz, y = Yc * iz;
  const double r2 = x * x + y * y;
  // radial(r2) and its derivative by Horner
  const double radial = 1.0 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * (k4 + r2 * (k5 + r2 * k6)))));
  const double drad =
      k1 + r2 * (2.0 * k2 + r2 * (3.0 * k3 + r2 * (4.0 * k4 + r2 * (5.0 * k5 + r2 * 6.0 * k6))));
  const double xd = x * radial + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x) + r2 * (s1 + s2 * r2);
  const double yd = y * radial + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y + r2 * (s3 + s4 * r2);
  uv[0] = fx * xd + skew * yd + cx;
  uv[1] = fy * yd + cy;
  // d(xd, yd) / d(x, y);  d r2 = 2x dx + 2y dy
  const double tpx = s1 + 2.0 * s2 * r2, tpy = s3 + 2.0 * s4 * r2;   // thin prism: d/d r2
  const double xd_x = radial + 2.0 * x * (x * drad + tpx) + 2.0 * p1 * y + 6.0 * p2 * x;
  const double xd_y = 2.0 * y * (x * drad + tpx) + 2.0 * p1 * x + 2.0 * p2 * y;
  const double yd_x = 2.0 * x * (y * drad + tpy) + 2.0 * p1 * x + 2.0 * p2 * y;
  const double yd_y = radial + 2.0 * y * (y * drad + tpy) + 6.0 * p1 * y + 2.0 * p2 * x;
  // d(u, v) / d(x, y)
  const double u_x = fx * xd_x + skew * yd_x, u_y = fx * xd_y + skew * yd_y;
  const double v_x = fy * yd_x, v_y = fy * yd_y;
  // d(x, y) / d(Xc, Yc, Zc) = [[iz, 0, -x iz], [0, iz, -y iz]]
  const double u_c[3] = {u_x * iz, u_y * iz, -(u_x * x + u_y * y) * iz};
  const double v_c[3] = {v_x * iz, v_y * iz, -(v_x * x + v_y * y) * iz};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    J[0][i] = u_c[0] * R[i] + u_c[1] * R[3 + i] + u_c[2] * R[6 + i];
    J[1][i] = v_c[0] * R[i] + v_c[1] * R[3 + i] + v_c[2] * R[6 + i];
  }
}

}  // namespace eks
