"""Adam + final smooth on the general path for shapes WITHOUT specialised kernels (n_latent 4 / 5): the generic kernels'
SCORE form against the dual-number losses (EKS_DENSE_DUAL_GRAD=1)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd.core import run_kalman_smoother
for T, K, D, O in ((50000, 4, 4, 4), (50000, 4, 5, 8)):
    rng = np.random.default_rng(4)
    lat = np.cumsum(rng.normal(size=(K, T, D)) * 0.7, axis=1)
    C = rng.normal(size=(K, O, D))
    ev = (0.25 * rng.gamma(2, 1, size=(T, K, O))).clip(1e-3).astype(np.float32)
    y = (np.einsum('kod,ktd->kto', C, lat) + rng.normal(size=(K, T, O)) * np.sqrt(np.swapaxes(ev, 0, 1))).astype(np.float32)
    L = rng.normal(size=(K, D, D)) * 0.3
    Q = L @ np.swapaxes(L, 1, 2) + 0.2 * np.eye(D)
    m0 = np.zeros((K, D)); S0 = np.tile(4 * np.eye(D), (K, 1, 1)); A = np.tile(np.eye(D), (K, 1, 1))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, ms, Vs = run_kalman_smoother(y, m0, S0, A, C, Q, ev)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'T={T} K={K} D={D} O={O}: adam + smooth {dt*1e3:.1f} ms  s={np.round(s, 3)}', flush=True)
