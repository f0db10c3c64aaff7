"""Calibrated pinhole cameras for the nonlinear multi-camera smoother.

Host-side mirror of the reference's calibration helpers (eks/multicam_smoother.py:767-953):
`rodrigues`, `make_projection_from_camgroup`, `triangulate_3d_models`,
`project_3d_covariance_to_2d`.  The reference builds a JAX closure h_fn and differentiates it with
jax.jacfwd; a C ABI cannot take a Python callable, so here the projection is DATA - one row of 32
float64 per camera (`pack_camera`, layout in include/eks_hip.h) - and `PinholeProjection` is the
callable wrapper `run_kalman_smoother(h_fn=...)` recognises.  The extended filter itself runs in
the HIP kernels (eks_ekf_smooth); what is evaluated here with NumPy is only the driver's epilogue
(reprojection of the smoothed 3-D means and their covariances) and the triangulation that
initialises the model.

A camera group is anything with a `.cameras` sequence whose items offer aniposelib's getters
(`get_rotation`, `get_translation`, `get_camera_matrix`, `get_distortions`); `Camera` /
`CameraGroup` below are minimal stand-ins for when aniposelib is not installed.
"""
from __future__ import annotations

from typing import Any, Sequence

import numpy as np

CAM_DOUBLES = 32


def rodrigues(rvec) -> np.ndarray:
    """OpenCV-style rotation vector (3,) -> matrix (3,3) (reference :771-796, including its
    first-order branch below 1e-12)."""
    rvec = np.asarray(rvec, dtype=np.float64).ravel()
    theta = float(np.linalg.norm(rvec))
    axis = rvec if theta < 1e-12 else rvec / theta
    rx, ry, rz = axis
    Kx = np.array([[0.0, -rz, ry], [rz, 0.0, -rx], [-ry, rx, 0.0]])
    if theta < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + np.sin(theta) * Kx + (1.0 - np.cos(theta)) * (Kx @ Kx)


def pack_camera(rot, tvec, K, dist) -> np.ndarray:
    """(rotation vector or matrix, translation, 3x3 camera matrix, OpenCV-ordered distortion
    coefficients) -> the 32 doubles eks_ekf_smooth reads."""
    rot = np.asarray(rot, dtype=np.float64)
    R = rot if rot.shape == (3, 3) else rodrigues(rot)
    K = np.asarray(K, dtype=np.float64)
    out = np.zeros(CAM_DOUBLES)
    out[0:9] = R.ravel()
    out[9:12] = np.asarray(tvec, dtype=np.float64).ravel()
    out[12:17] = K[0, 0], K[1, 1], K[0, 2], K[1, 2], K[0, 1]
    d = np.asarray(dist, dtype=np.float64).ravel()[:14]
    out[17:17 + len(d)] = d
    return out


def _distort(cam, x, y):
    """normalised (x, y) -> distorted (xd, yd) and the 2x2 derivative (reference :840-862)."""
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = cam[17:29]
    r2 = x * x + y * y
    radial = 1.0 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * (k4 + r2 * (k5 + r2 * k6)))))
    drad = k1 + r2 * (2 * k2 + r2 * (3 * k3 + r2 * (4 * k4 + r2 * (5 * k5 + r2 * 6 * k6))))
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + r2 * (s1 + s2 * r2)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + r2 * (s3 + s4 * r2)
    tpx, tpy = s1 + 2 * s2 * r2, s3 + 2 * s4 * r2
    xd_x = radial + 2 * x * (x * drad + tpx) + 2 * p1 * y + 6 * p2 * x
    xd_y = 2 * y * (x * drad + tpx) + 2 * p1 * x + 2 * p2 * y
    yd_x = 2 * x * (y * drad + tpy) + 2 * p1 * x + 2 * p2 * y
    yd_y = radial + 2 * y * (y * drad + tpy) + 6 * p1 * y + 2 * p2 * x
    return xd, yd, (xd_x, xd_y, yd_x, yd_y)


def project(cam: np.ndarray, X) -> np.ndarray:
    """world points (..., 3) -> pixels (..., 2) for one packed camera."""
    X = np.asarray(X, dtype=np.float64)
    R = cam[0:9].reshape(3, 3)
    Xc = X @ R.T + cam[9:12]
    x, y = Xc[..., 0] / Xc[..., 2], Xc[..., 1] / Xc[..., 2]
    xd, yd, _ = _distort(cam, x, y)
    fx, fy, cx, cy, skew = cam[12:17]
    return np.stack([fx * xd + skew * yd + cx, fy * yd + cy], axis=-1)


def project_jacobian(cam: np.ndarray, X) -> np.ndarray:
    """d pixels / d world at X (..., 3) -> (..., 2, 3), analytic."""
    X = np.asarray(X, dtype=np.float64)
    R = cam[0:9].reshape(3, 3)
    Xc = X @ R.T + cam[9:12]
    iz = 1.0 / Xc[..., 2]
    x, y = Xc[..., 0] * iz, Xc[..., 1] * iz
    _, _, (xd_x, xd_y, yd_x, yd_y) = _distort(cam, x, y)
    fx, fy, _, _, skew = cam[12:17]
    u_x, u_y = fx * xd_x + skew * yd_x, fx * xd_y + skew * yd_y
    v_x, v_y = fy * yd_x, fy * yd_y
    u_c = np.stack([u_x * iz, u_y * iz, -(u_x * x + u_y * y) * iz], axis=-1)      # d u / d Xc
    v_c = np.stack([v_x * iz, v_y * iz, -(v_x * x + v_y * y) * iz], axis=-1)
    return np.stack([u_c @ R, v_c @ R], axis=-2)


class PinholeProjection:
    """h_fn for run_kalman_smoother: x (3,) or (..., 3) -> concatenated (u, v) of every camera
    (reference make_projection_from_camgroup, :871-898).  Carries the packed cameras the kernels
    read; `heads[c]` is the single-camera projection the driver's epilogue uses."""

    def __init__(self, cams_packed: np.ndarray):
        self.cams = np.ascontiguousarray(np.asarray(cams_packed, dtype=np.float64))
        if self.cams.ndim != 2 or self.cams.shape[1] != CAM_DOUBLES:
            raise ValueError(f'cams must be (n_cameras, {CAM_DOUBLES})')
        self.heads = [_Head(c) for c in self.cams]

    @property
    def n_cameras(self) -> int:
        return self.cams.shape[0]

    def __call__(self, x):
        return np.concatenate([h(x) for h in self.heads], axis=-1)


class _Head:
    def __init__(self, cam: np.ndarray):
        self.cam = cam

    def __call__(self, x):
        return project(self.cam, x)

    def jacobian(self, x):
        return project_jacobian(self.cam, x)


def cameras_of(camgroup: Any) -> np.ndarray:
    """Packed (V, 32) cameras of an aniposelib-style camera group (reference :876-884)."""
    rows = []
    for cam in camgroup.cameras:
        rot = np.asarray(cam.get_rotation(), dtype=np.float64)
        rot = rot if rot.shape == (3, 3) else rot.ravel()
        rows.append(pack_camera(rot, np.asarray(cam.get_translation()).ravel(),
                                np.asarray(cam.get_camera_matrix()),
                                np.asarray(cam.get_distortions()).ravel()))
    return np.stack(rows)


def make_projection_from_camgroup(camgroup: Any) -> tuple[PinholeProjection, list]:
    """-> (combined multi-view h_fn R^3 -> R^{2V}, per-camera heads), reference :871-898."""
    h = PinholeProjection(cameras_of(camgroup))
    return h, h.heads


def undistort_points(cam: np.ndarray, uv, iters: int = 20) -> np.ndarray:
    """pixels (..., 2) -> undistorted normalised image coordinates, by fixed-point iteration on
    the projection's own distortion model."""
    uv = np.asarray(uv, dtype=np.float64)
    fx, fy, cx, cy, skew = cam[12:17]
    yd = (uv[..., 1] - cy) / fy
    xd = (uv[..., 0] - cx - skew * yd) / fx
    x, y = xd.copy(), yd.copy()
    for _ in range(iters):
        fxd, fyd, _ = _distort(cam, x, y)
        r2 = x * x + y * y
        k1, k2, _, _, k3, k4, k5, k6 = cam[17:25]
        radial = 1.0 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * (k4 + r2 * (k5 + r2 * k6)))))
        x = (xd - (fxd - x * radial)) / radial
        y = (yd - (fyd - y * radial)) / radial
    return np.stack([x, y], axis=-1)


def triangulate(cams_packed: np.ndarray, xy_views) -> np.ndarray:
    """xy_views (V, N, 2) pixels -> (N, 3) world points: undistort, then the least-squares
    solution of the homogeneous system x P_3 - P_1 = 0, y P_3 - P_2 = 0 over the cameras (what
    aniposelib's CameraGroup.triangulate solves per point; reference call at :912-913).  Like
    aniposelib, a view whose marker is not finite is left out of THAT point's system, and a point
    seen by fewer than two cameras comes back NaN (a single NaN must not abort the whole run)."""
    xy_views = np.asarray(xy_views, dtype=np.float64)
    ok = np.isfinite(xy_views).all(axis=-1)                              # (V, N)
    rows = []
    for c, cam in enumerate(cams_packed):
        Pm = np.concatenate([cam[0:9].reshape(3, 3), cam[9:12].reshape(3, 1)], axis=1)
        n = undistort_points(cam, np.where(ok[c][:, None], xy_views[c], 0.0))
        w = ok[c].astype(np.float64)[:, None]                            # dropped views: zero rows
        rows.append(w * (n[:, 0, None] * Pm[2][None] - Pm[0][None]))
        rows.append(w * (n[:, 1, None] * Pm[2][None] - Pm[1][None]))
    A = np.stack(rows, axis=1)                                           # (N, 2V, 4)
    enough = ok.sum(axis=0) >= 2
    out = np.full((A.shape[0], 3), np.nan)
    if enough.any():
        # smallest right singular vector = smallest eigenvector of the 4x4 normal matrix
        Ag = A[enough]
        _, vecs = np.linalg.eigh(np.swapaxes(Ag, 1, 2) @ Ag)
        p = vecs[..., 0]
        out[enough] = p[:, :3] / p[:, 3:4]
    return out


def triangulate_3d_models(marker_array, camgroup: Any) -> np.ndarray:
    """Per model, keypoint and frame: (M, K, T, 3) (reference :901-921).  Like the reference this
    calls the camera group's own `triangulate(xy_views (C,T,2), fast=True, disable_64bit=True)`
    when it has one (aniposelib's, or `CameraGroup` below); a bare `.cameras` holder is
    triangulated here."""
    raw = np.asarray(marker_array.get_array())                           # (M,V,T,K,F)
    M, V, T, K, _ = raw.shape
    tri_fn = getattr(camgroup, 'triangulate', None)
    cams = None if tri_fn is not None else cameras_of(camgroup)
    out = np.empty((M, K, T, 3))
    for m in range(M):
        for k in range(K):
            xy = raw[m, :, :, k, :2]
            out[m, k] = tri_fn(xy, fast=True, disable_64bit=True) if tri_fn is not None \
                else triangulate(cams, xy)
    return out


def project_3d_covariance_to_2d(ms_k, Vs_k, h_cam: _Head, inflated_vars_k):
    """Var of the reprojection: diag(J V J^T) + ensemble variance (reference :924-953, which adds
    the FIRST TWO columns of the keypoint's (T, 2V) variance array for every camera)."""
    J = h_cam.jacobian(np.asarray(ms_k, dtype=np.float64))               # (T,2,3)
    cov = J @ np.asarray(Vs_k, dtype=np.float64) @ np.swapaxes(J, 1, 2)
    v = np.asarray(inflated_vars_k)
    return cov[:, 0, 0] + v[:, 0], cov[:, 1, 1] + v[:, 1]


class Camera:
    """Minimal holder with aniposelib's getter names."""

    def __init__(self, rotation, translation, matrix, distortions=(), name: str = ''):
        self._rot = np.asarray(rotation, dtype=np.float64)
        self._t = np.asarray(translation, dtype=np.float64).ravel()
        self._K = np.asarray(matrix, dtype=np.float64)
        self._d = np.asarray(distortions, dtype=np.float64).ravel()
        self.name = name

    def get_rotation(self):
        return self._rot

    def get_translation(self):
        return self._t

    def get_camera_matrix(self):
        return self._K

    def get_distortions(self):
        return self._d


class CameraGroup:
    def __init__(self, cameras: Sequence[Camera]):
        self.cameras = list(cameras)

    @classmethod
    def load(cls, path: str) -> 'CameraGroup':
        """Read an aniposelib calibration TOML: one `[cam_N]` table per camera with `name`,
        `matrix` (3x3), `distortions`, `rotation` (rotation vector), `translation`; other tables
        (`[metadata]`) are ignored.  (aniposelib's CameraGroup.load, used at reference
        eks/multicam_smoother.py:232; neither aniposelib nor a TOML module is in this image, so
        the flat key = value subset those files use is parsed here.)"""
        tables = _read_flat_toml(path)
        cams = []
        # aniposelib orders the tables by their keys sorted AS STRINGS (cam_0, cam_1, cam_10, cam_2 ...)
        for key in sorted(k for k in tables if k.startswith('cam_')):
            t = tables[key]
            cams.append(Camera(np.asarray(t['rotation'], float), t['translation'], t['matrix'],
                               t.get('distortions', ()), name=str(t.get('name', key))))
        if not cams:
            raise ValueError(f'no [cam_N] tables in {path}')
        return cls(cams)

    def triangulate(self, xy_views, **_):
        return triangulate(cameras_of(self), xy_views)


def _read_flat_toml(path: str) -> dict:
    import ast
    tables: dict = {}
    cur = tables.setdefault('', {})
    pending = ''
    with open(path) as f:
        for raw in f:
            line = raw.split('#', 1)[0].strip() if '"' not in raw else raw.strip()
            if not line:
                continue
            if not pending and line.startswith('[') and line.endswith(']') and '=' not in line:
                cur = tables.setdefault(line[1:-1].strip(), {})
                continue
            pending = f'{pending} {line}' if pending else line
            if pending.count('[') != pending.count(']'):
                continue                                  # array continues on the next line
            key, _, val = pending.partition('=')
            val = val.strip()
            val = {'true': 'True', 'false': 'False'}.get(val, val)
            cur[key.strip()] = ast.literal_eval(val)
            pending = ''
    return tables
