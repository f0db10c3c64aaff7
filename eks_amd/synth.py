"""Seeded synthetic pose-ensemble generators for the BASELINE.json configs.

Latent: per (keypoint, coord) random walk x_t = x_{t-1} + N(0, q_k), q_k ~ LogUniform(0.05, 5) px^2,
x_0 ~ U(50, 450).  Ensemble: M members x_t + N(0, sigma2_{k,t}), sigma2 = 0.25 * Gamma(2, 1); 2 % of
(frame, keypoint) pairs are "occluded": sigma2 x100 and likelihood ~ U(0.05, 0.5), otherwise
likelihood ~ Beta(50, 1).  Fields (x, y, likelihood), float32, axis order of the reference's
MarkerArray (n_models, n_cameras, n_frames, n_keypoints, n_fields), eks/marker_array.py:18-20.
"""
from __future__ import annotations

import numpy as np


def singlecam_markers(T: int, K: int, M: int = 5, seed: int = 0) -> np.ndarray:
    """NumPy generator (CPU).  Returns float32 (M, 1, T, K, 3)."""
    rng = np.random.default_rng(seed)
    q = np.exp(rng.uniform(np.log(0.05), np.log(5.0), size=(K, 1)))
    x0 = rng.uniform(50.0, 450.0, size=(1, K, 2))
    steps = rng.standard_normal((T, K, 2)) * np.sqrt(q)[None]
    steps[0] = 0.0
    lat = x0 + np.cumsum(steps, axis=0)                                  # (T,K,2)
    sig2 = 0.25 * rng.gamma(2.0, 1.0, size=(T, K))
    occ = rng.random((T, K)) < 0.02
    sig2 = np.where(occ, sig2 * 100.0, sig2)
    out = np.empty((M, 1, T, K, 3), dtype=np.float32)
    for m in range(M):
        noise = rng.standard_normal((T, K, 2)) * np.sqrt(sig2)[..., None]
        out[m, 0, :, :, 0:2] = lat + noise
        lik = np.where(occ, rng.uniform(0.05, 0.5, size=(T, K)), rng.beta(50.0, 1.0, size=(T, K)))
        out[m, 0, :, :, 2] = lik
    return out


def multicam_markers(T: int, K: int, V: int = 2, M: int = 5, seed: int = 0) -> np.ndarray:
    """Mirror-mouse-like multi-view ensemble: each keypoint follows a 3-D random walk seen through
    V fixed random affine 3-D -> 2-D maps plus per-view ensemble noise.  float32 (M, V, T, K, 3)."""
    rng = np.random.default_rng(seed)
    q = np.exp(rng.uniform(np.log(0.05), np.log(5.0), size=(K, 1)))
    steps = rng.standard_normal((T, K, 3)) * np.sqrt(q)[None]
    steps[0] = 0.0
    lat = np.cumsum(steps, axis=0)                                       # (T,K,3)
    out = np.empty((M, V, T, K, 3), dtype=np.float32)
    for v in range(V):
        P = rng.standard_normal((2, 3))
        P /= np.linalg.norm(P, axis=1, keepdims=True)
        off = rng.uniform(100.0, 400.0, size=(1, K, 2))
        proj = lat @ P.T + off                                           # (T,K,2)
        sig2 = 0.25 * rng.gamma(2.0, 1.0, size=(T, K))
        occ = rng.random((T, K)) < 0.02
        sig2 = np.where(occ, sig2 * 100.0, sig2)
        for m in range(M):
            out[m, v, :, :, 0:2] = proj + rng.standard_normal((T, K, 2)) * np.sqrt(sig2)[..., None]
            out[m, v, :, :, 2] = np.where(occ, rng.uniform(0.05, 0.5, size=(T, K)),
                                          rng.beta(50.0, 1.0, size=(T, K)))
    return out


def singlecam_observations_torch(T: int, K: int, seed: int, device):
    """Device-side generator for bench-sized inputs: returns the smoother's direct inputs
    (centred ensemble averages y and ensemble variances var, both float32 [T, K, 2] on `device`)
    drawn from the generative model above without materialising the M ensemble members: the latent
    random walk, per-frame member variance sig2 (2 % occluded frames x100), `var` = a chi-square-ish
    estimate of sig2 per coordinate (what a 5-member ensemble variance looks like), and
    y = latent + N(0, 0.29 sig2) (the sampling variance of a 5-member median)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))

    def rand(*shape):
        return torch.rand(*shape, device=device, generator=g)

    def randn(*shape):
        return torch.randn(*shape, device=device, generator=g)

    q = torch.exp(rand(K, 1) * float(np.log(5.0 / 0.05)) + float(np.log(0.05)))
    steps = randn(T, K, 2) * torch.sqrt(q)[None]
    steps[0] = 0
    lat = torch.cumsum(steps, dim=0)
    del steps
    lat -= lat.mean(dim=0, keepdim=True)
    sig2 = 0.25 * (-torch.log(rand(T, K).clamp_min(1e-12)) - torch.log(rand(T, K).clamp_min(1e-12)))
    occ = rand(T, K) < 0.02
    sig2 = torch.where(occ, sig2 * 100.0, sig2)
    # ensemble variance of 5 members ~ sig2 * chi2_4 / 4 (mean 1, never exactly 0)
    chi = (randn(T, K, 2, 4) ** 2).mean(dim=-1)
    var = (sig2[..., None] * chi.clamp_min(1e-3)).float().contiguous()
    y = (lat + randn(T, K, 2) * torch.sqrt(0.29 * sig2)[..., None]).float().contiguous()
    return y, var


def pupil_observations(T: int, seed: int, noise: float = 0.6):
    """Synthetic pupil session for the AR(1) path (reference eks/ibl_pupil_smoother.py): latent
    (diameter, com_x, com_y) AR(1) with s = (0.995, 0.97, 0.97), eight observations through the
    fixed 8x3 matrix, per-frame ensemble variances ~ noise * Gamma(2, 0.5) + 0.05.
    Returns float32-rounded float64 (ys (T,8), ensemble_vars (T,8)) and m0 (3,), S0 (3,3),
    latent_vars (3,)."""
    rng = np.random.default_rng(seed)
    C = np.array([[0, 1, 0], [-.5, 0, 1], [0, 1, 0], [.5, 0, 1],
                  [.5, 1, 0], [0, 0, 1], [-.5, 1, 0], [0, 0, 1]], dtype=np.float64)
    a = np.array([0.995, 0.97, 0.97])
    drive = rng.normal(size=(T, 3)) * np.array([0.15, 0.5, 0.5])
    lat = np.empty((T, 3))
    x = np.zeros(3)
    for t in range(T):
        x = a * x + drive[t]
        lat[t] = x
    lat[:, 0] += 12.0
    ev = (noise * rng.gamma(2.0, 0.5, size=(T, 8)) + 0.05).astype(np.float32).astype(np.float64)
    ys = (lat @ C.T + rng.normal(size=(T, 8)) * np.sqrt(ev)).astype(np.float32).astype(np.float64)
    lv = lat.var(axis=0)
    return ys, ev, np.array([lat[:, 0].mean(), 0.0, 0.0]), np.diag(lv), lv


def ring_cameras(V: int, seed: int = 0) -> list[dict]:
    """V calibrated cameras on a ring of radius ~1 m looking at the origin, with mild radial /
    tangential / thin-prism distortion and a little skew: dicts(rot (3,3), tvec, K, dist)."""
    rng = np.random.default_rng(1000 + seed)
    cams = []
    for v in range(V):
        ang = 2.0 * np.pi * v / V + 0.3
        pos = np.array([1000.0 * np.cos(ang), 1000.0 * np.sin(ang), 300.0 + 60.0 * v])
        z = -pos / np.linalg.norm(pos)
        x = np.cross([0.0, 0.0, 1.0], z)
        x /= np.linalg.norm(x)
        R = np.stack([x, np.cross(z, x), z])
        Kmat = np.array([[900.0 + 25.0 * v, 0.4 * v, 320.0], [0.0, 880.0 + 10.0 * v, 240.0],
                         [0.0, 0.0, 1.0]])
        dist = np.array([-0.15, 0.06, 1e-3, -6e-4, 0.012, 0.0, 0.0, 0.0, 3e-4, 0.0, -2e-4, 0.0])
        dist[:2] += rng.normal(0.0, 0.01, 2)
        cams.append(dict(rot=R, tvec=-R @ pos, K=Kmat, dist=dist))
    return cams


def calibrated_multicam(T: int, K: int, V: int = 3, M: int = 5, seed: int = 0) -> dict:
    """Calibrated multi-camera problem: K keypoints follow 3-D
    random walks inside a ~40 cm volume, seen by `ring_cameras(V)`; ensemble noise as in
    `singlecam_markers`.  Returns the markers (M, V, T, K, 3) float32 AND a ready filter problem:
    y_tko / var_tko (T, K, 2V) (ensemble median / variance), m0s, S0s, As, Qs, a smoothing
    parameter per keypoint, the cameras (dicts) and `cams_packed` (V, 32)."""
    from .calibration import pack_camera, project
    rng = np.random.default_rng(seed)
    cams = ring_cameras(V, seed)
    packed = np.stack([pack_camera(c['rot'], c['tvec'], c['K'], c['dist']) for c in cams])
    q = np.exp(rng.uniform(np.log(0.5), np.log(8.0), size=(K, 1)))
    steps = rng.standard_normal((T, K, 3)) * np.sqrt(q)[None]
    steps[0] = 0.0
    lat = rng.uniform(-150.0, 150.0, size=(1, K, 3)) + np.cumsum(steps, axis=0)
    lat = np.clip(lat, -400.0, 400.0)                                     # stay in front of the cameras
    sig2 = 0.25 * rng.gamma(2.0, 1.0, size=(V, T, K))
    occ = rng.random((V, T, K)) < 0.02
    sig2 = np.where(occ, sig2 * 100.0, sig2)
    markers = np.empty((M, V, T, K, 3), dtype=np.float32)
    for v in range(V):
        uv = project(packed[v], lat)                                      # (T,K,2)
        for m in range(M):
            markers[m, v, :, :, 0:2] = uv + rng.standard_normal((T, K, 2)) * np.sqrt(sig2[v])[..., None]
            markers[m, v, :, :, 2] = np.where(occ[v], rng.uniform(0.05, 0.5, size=(T, K)),
                                              rng.beta(50.0, 1.0, size=(T, K)))
    xy = markers[..., :2].astype(np.float64)
    med = np.median(xy, axis=0)                                           # (V,T,K,2)
    var = np.var(xy, axis=0) + 1e-3
    y_tko = np.transpose(med, (1, 2, 0, 3)).reshape(T, K, 2 * V)
    var_tko = np.transpose(var, (1, 2, 0, 3)).reshape(T, K, 2 * V)
    m0s = lat[:10].mean(axis=0) + rng.normal(0.0, 3.0, size=(K, 3))
    S0s = np.stack([np.diag(lat[:, k].var(axis=0) + 1e-4) for k in range(K)])
    Qs = np.stack([np.diag(np.full(3, 1.0) * (0.5 + rng.random(3))) for _ in range(K)])
    return dict(markers=markers, latent=lat, cams=cams, cams_packed=packed, y_tko=y_tko,
                var_tko=var_tko, m0s=m0s, S0s=S0s, As=np.tile(np.eye(3), (K, 1, 1)), Qs=Qs,
                s=np.exp(rng.uniform(np.log(0.05), np.log(20.0), size=K)))
