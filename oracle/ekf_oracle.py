"""CPU restatement of the reference's NONLINEAR (calibrated multi-camera) path - TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (eks_amd/) never does.  **Parity unpinned**: dynamax / jax / aniposelib are not installed
here, so the extended-filter recursion is restated from SURVEY.md Appendix A.1 and pinned by its own
properties (an affine "camera" reduces it to the linear filter of oracle/eks_oracle.py; the
Jacobian is taken by complex step, independent of the analytic one in the kernels).

What follows the reference, by file:line of /root/reference:
  * rodrigues, make_projection_fn  - eks/multicam_smoother.py:771-868 (projection; the radial
    factor is a polynomial in r^2 up to r^12, thin prism s1..s4, skew)
  * ekf_filter / eks_smoother      - dynamax extended_kalman_filter / _smoother as called at
    eks/core.py:290 (final pass, time-varying diagonal R) and :648 (loss, constant R)
  * run_kalman_smoother_nonlinear  - eks/core.py:159-302 with h_fn (the optimiser of :562-699 with
    the gradient of the loss taken by central differences in log s instead of autodiff)
  * initialize_kalman_filter_geometric - eks/multicam_smoother.py:600-650
  * project_3d_covariance_to_2d    - eks/multicam_smoother.py:924-953
  * triangulate_dlt                - aniposelib CameraGroup.triangulate(fast=True) as called at
    eks/multicam_smoother.py:912-913: undistort, then the homogeneous linear system per point.
"""
from __future__ import annotations

import numpy as np

from oracle import eks_oracle as eo


def rodrigues(rvec):
    """OpenCV-style rotation vector -> matrix, eks/multicam_smoother.py:771-796."""
    rvec = np.asarray(rvec, dtype=np.float64).ravel()
    theta = np.linalg.norm(rvec)
    if theta < 1e-12:
        rx, ry, rz = rvec
        Kx = np.array([[0.0, -rz, ry], [rz, 0.0, -rx], [-ry, rx, 0.0]])
        return np.eye(3) + Kx
    rx, ry, rz = rvec / theta
    Kx = np.array([[0.0, -rz, ry], [rz, 0.0, -rx], [-ry, rx, 0.0]])
    return np.eye(3) + np.sin(theta) * Kx + (1.0 - np.cos(theta)) * (Kx @ Kx)


def make_projection_fn(rot, tvec, Kmat, dist, dtype=np.float64):
    """world (...,3) -> pixels (...,2); accepts complex input (for the complex-step Jacobian).
    `rot` is a rotation vector (3,) or matrix (3,3).  eks/multicam_smoother.py:799-868.
    dtype=np.float32: constants rounded to float32, so that a float32 argument is projected in float32
    arithmetic throughout (the reference's production precision; oracle/f32_forecast.py)."""
    rot = np.asarray(rot, dtype=np.float64)
    R = (rot if rot.shape == (3, 3) else rodrigues(rot)).astype(dtype)
    t = np.asarray(tvec, dtype=np.float64).ravel().astype(dtype)
    Kmat = np.asarray(Kmat, dtype=np.float64).astype(dtype)
    fx, fy, cx, cy, skew = Kmat[0, 0], Kmat[1, 1], Kmat[0, 2], Kmat[1, 2], Kmat[0, 1]
    dc = np.zeros(14, dtype=dtype)
    dist = np.asarray(dist, dtype=np.float64).ravel()
    dc[:len(dist)] = dist[:14]
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = dc[:12]

    def project(Xw):
        Xw = np.asarray(Xw)
        Xc = Xw @ R.T + t
        X, Y, Z = Xc[..., 0], Xc[..., 1], Xc[..., 2]
        x = X / Z
        y = Y / Z
        r2 = x * x + y * y
        r4 = r2 * r2
        r6 = r4 * r2
        r8 = r4 * r4
        r10 = r8 * r2
        r12 = r6 * r6
        radial = 1.0 + k1 * r2 + k2 * r4 + k3 * r6 + k4 * r8 + k5 * r10 + k6 * r12
        x_tan = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        y_tan = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
        xd = x * radial + x_tan + s1 * r2 + s2 * r4
        yd = y * radial + y_tan + s3 * r2 + s4 * r4
        return np.stack([fx * xd + skew * yd + cx, fy * yd + cy], axis=-1)

    return project


def combine_projections(h_cams):
    """h(x) = concat over cameras, eks/multicam_smoother.py:886-896."""
    def h(x):
        return np.concatenate([hc(x) for hc in h_cams], axis=-1)
    return h


def jacobian_cs(h, x, step=1e-30):
    """d h / d x at x (D,) by complex step (exact to rounding for these rational functions)."""
    x = np.asarray(x, dtype=np.float64)
    cols = []
    for i in range(x.shape[0]):
        xc = x.astype(np.complex128)
        xc[i] += 1j * step
        cols.append(np.imag(h(xc)) / step)
    return np.stack(cols, axis=-1)


def ekf_filter(y, Rdiag, m0, S0, A, Q, s, h, jac=None):
    """Extended filter, update-then-predict.  y (T,O); Rdiag (T,O) or (O,).
    Returns (ll, filtered means (T,D), covariances (T,D,D), predicted means (T,D))."""
    y = np.asarray(y, np.float64)
    T, O = y.shape
    Rdiag = np.asarray(Rdiag, np.float64)
    A = np.asarray(A, np.float64)
    sQ = float(s) * np.asarray(Q, np.float64)
    m = np.asarray(m0, np.float64).copy()
    P = np.asarray(S0, np.float64).copy()
    D = m.shape[0]
    mf = np.empty((T, D))
    Pf = np.empty((T, D, D))
    mp = np.empty((T, D))
    ll = 0.0
    for t in range(T):
        mp[t] = m
        H = jac(m) if jac is not None else jacobian_cs(h, m)
        r = Rdiag[t] if Rdiag.ndim == 2 else Rdiag
        S = H @ P @ H.T + np.diag(r)
        e = y[t] - h(m)
        L = np.linalg.cholesky(S)
        z = np.linalg.solve(L, e)
        ll -= 0.5 * (O * np.log(2.0 * np.pi) + 2.0 * np.log(np.diag(L)).sum() + z @ z)
        Kg = np.linalg.solve(S, H @ P).T
        m = m + Kg @ e
        P = P - Kg @ S @ Kg.T
        P = 0.5 * (P + P.T)
        mf[t] = m
        Pf[t] = P
        m = A @ m
        P = A @ P @ A.T + sQ
    return ll, mf, Pf, mp


def eks_smoother(y, Rdiag, m0, S0, A, Q, s, h, jac=None):
    """Extended filter + RTS pass (the dynamics are linear, so the backward pass is the linear
    one of oracle/eks_oracle.py).  Returns (ms (T,D), Vs (T,D,D), ll)."""
    ll, mf, Pf, _ = ekf_filter(y, Rdiag, m0, S0, A, Q, s, h, jac)
    ms, Vs = eo.rts_smoother(mf[None], Pf[None], np.asarray(A, np.float64)[None],
                             np.asarray(Q, np.float64)[None], np.array([float(s)]))
    return ms[0], Vs[0], ll


def ekf_nll(y, Rconst, m0, S0, A, Q, s, h, jac=None):
    return -ekf_filter(y, Rconst, m0, S0, A, Q, s, h, jac)[0]


def run_kalman_smoother_nonlinear(ys, m0s, S0s, As, Qs, ensemble_vars, h_fn, s_frames=None,
                                  smooth_param=None, blocks=None, lr=0.25,
                                  s_bounds_log=(-8.0, 8.0), tol=1e-2, safety_cap=300, fd_step=1e-4):
    """eks/core.py:159-302 with h_fn: choose s per keypoint (or block), then extended filter +
    smoother with time-varying R.  ensemble_vars (T,K,O).  The loss gradient d NLL / d log s is a
    central difference (the reference differentiates through the filter)."""
    ys = np.asarray(ys, np.float64)
    K, T, O = ys.shape
    ev = np.swapaxes(np.asarray(ensemble_vars, np.float64), 0, 1)      # (K,T,O)
    Rd = eo.build_R_from_vars(ev)
    s_finals = np.empty(K)
    info = {}
    if smooth_param is not None:
        s_finals[:] = smooth_param if isinstance(smooth_param, (int, float)) \
            else np.asarray(smooth_param, float)
    else:
        if not blocks:
            blocks = [[k] for k in range(K)]
        guesses = np.array([eo.compute_initial_guess(np.asarray(ensemble_vars)[:, k, :])
                            for k in range(K)])
        y_c = [eo.crop_frames(ys[k], s_frames) if s_frames else ys[k] for k in range(K)]
        R_c = [eo.constant_R_from_timevarying(
            eo.crop_frames(Rd[k], s_frames) if s_frames else Rd[k], 1e-4) for k in range(K)]
        u0 = np.array([np.float32(np.log(np.clip(np.mean([guesses[k] for k in b]), 1e-6, 1e3)))
                       for b in blocks], dtype=np.float64)

        def block_loss(b, u):
            return sum(ekf_nll(y_c[k], R_c[k], m0s[k], S0s[k], As[k], Qs[k], np.exp(u), h_fn)
                       for k in blocks[b])

        def loss_and_grad(u_blocks):
            L = np.array([block_loss(b, u) for b, u in enumerate(u_blocks)])
            G = np.array([(block_loss(b, u + fd_step) - block_loss(b, u - fd_step)) / (2 * fd_step)
                          for b, u in enumerate(u_blocks)])
            return L, G

        u, last, iters = eo.adam_optimize_s(loss_and_grad, u0, lr=lr, s_bounds_log=s_bounds_log,
                                            tol=tol, safety_cap=safety_cap)
        s_b = np.exp(np.clip(u, *s_bounds_log))
        for b, blk in enumerate(blocks):
            s_finals[blk] = s_b[b]
        info = dict(last_loss=last, iters=iters)
    D = np.shape(m0s)[1]
    ms = np.empty((K, T, D))
    Vs = np.empty((K, T, D, D))
    for k in range(K):
        ms[k], Vs[k], _ = eks_smoother(ys[k], Rd[k], m0s[k], S0s[k], As[k], Qs[k], s_finals[k], h_fn)
    return s_finals, ms, Vs, info


def initialize_kalman_filter_geometric(ys):
    """eks/multicam_smoother.py:600-650.  ys (K,T,3) triangulated points."""
    ys = np.asarray(ys, np.float64)
    K, T, D = ys.shape
    m0s = np.stack([ys[k, :10].mean(axis=0) for k in range(K)])
    S0s = np.stack([np.diag([np.nanvar(ys[k, :, d]) + 1e-4 for d in range(D)]) for k in range(K)])
    As = np.tile(np.eye(D), (K, 1, 1))
    Cs = np.tile(np.eye(D), (K, 1, 1))
    Qs = []
    for k in range(K):
        dx = np.diff(ys[k], axis=0)
        med = np.median(dx, axis=0)
        mad = np.median(np.abs(dx - med), axis=0) + 1e-12
        Qs.append(np.diag(np.maximum((1.4826 * mad) ** 2, 1e-8)))
    return m0s, S0s, As, np.stack(Qs), Cs


def project_3d_covariance_to_2d(ms_k, Vs_k, h_cam, vars_k):
    """eks/multicam_smoother.py:924-953 (including its use of the FIRST TWO columns of the
    keypoint's (T, 2V) variance array for every camera)."""
    J = np.stack([jacobian_cs(h_cam, m) for m in np.asarray(ms_k, np.float64)])      # (T,2,3)
    cov = J @ np.asarray(Vs_k, np.float64) @ np.swapaxes(J, 1, 2)
    return cov[:, 0, 0] + vars_k[:, 0], cov[:, 1, 1] + vars_k[:, 1]


def undistort_normalised(uv, Kmat, dist, iters=20):
    """pixels -> undistorted normalised coordinates by fixed-point iteration on the projection's
    own distortion model."""
    Kmat = np.asarray(Kmat, np.float64)
    fx, fy, cx, cy, skew = Kmat[0, 0], Kmat[1, 1], Kmat[0, 2], Kmat[1, 2], Kmat[0, 1]
    dc = np.zeros(14)
    dist = np.asarray(dist, np.float64).ravel()
    dc[:len(dist)] = dist[:14]
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = dc[:12]
    yd = (uv[..., 1] - cy) / fy
    xd = (uv[..., 0] - cx - skew * yd) / fx
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        radial = 1.0 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * (k4 + r2 * (k5 + r2 * k6)))))
        dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x) + r2 * (s1 + s2 * r2)
        dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y + r2 * (s3 + s4 * r2)
        x = (xd - dx) / radial
        y = (yd - dy) / radial
    return np.stack([x, y], axis=-1)


def triangulate_dlt(cams, xy_views):
    """cams: list of dicts(rot, tvec, K, dist); xy_views (C,N,2) pixels -> (N,3) world points."""
    C, N, _ = xy_views.shape
    rows = []
    for c, cam in enumerate(cams):
        rot = np.asarray(cam['rot'], np.float64)
        R = rot if rot.shape == (3, 3) else rodrigues(rot)
        Pm = np.concatenate([R, np.asarray(cam['tvec'], np.float64).reshape(3, 1)], axis=1)   # 3x4
        n = undistort_normalised(np.asarray(xy_views[c], np.float64), cam['K'], cam['dist'])
        rows.append(n[:, 0, None] * Pm[2][None] - Pm[0][None])
        rows.append(n[:, 1, None] * Pm[2][None] - Pm[1][None])
    Amat = np.stack(rows, axis=1)                        # (N, 2C, 4)
    _, _, vh = np.linalg.svd(Amat)
    p = vh[:, -1, :]
    return p[:, :3] / p[:, 3:4]
