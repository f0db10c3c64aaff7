"""run_kalman_smoother on sessions of the reference's own recordings' size (2 000 frames, a handful of keypoints):
fixed s, grid search and the default Adam mode, NumPy in / NumPy out, steady state (third call on)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eks_amd import synth
from eks_amd.core import run_kalman_smoother
dev = torch.device('cuda', 0)
for T, K in ((2000, 4), (2000, 16), (2000, 64), (10000, 16)):
    y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
    ys = np.ascontiguousarray(y.transpose(0, 1).cpu().numpy().astype(np.float64))
    ev = var.cpu().numpy().astype(np.float64)
    eye = np.tile(np.eye(2), (K, 1, 1)); m0 = np.zeros((K, 2)); S0 = eye * ys.var(axis=1)[:, :, None]
    for label, kw in (('fixed s', dict(smooth_param=10.0)), ('grid', dict(s_mode='grid')), ('adam (default)', dict())):
        ts = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = run_kalman_smoother(ys, m0, S0, eye, eye, eye, ev, **kw)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f'T={T} K={K} {label}: {1e3 * min(ts[2:]):.2f} ms', flush=True)
