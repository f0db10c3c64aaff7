"""PCIe-inclusive rate: run_kalman_smoother called with NumPy inputs (the drop-in boundary), C3 shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eks_amd import synth
from eks_amd.core import run_kalman_smoother
T, K = 100000, 256
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=torch.device('cuda', 0))
ys = np.ascontiguousarray(np.transpose(y.cpu().numpy(), (1, 0, 2)))        # (K,T,2) like upstream
ev = var.cpu().numpy()                                                      # (T,K,2)
eye = np.tile(np.eye(2), (K, 1, 1)); m0 = np.zeros((K, 2)); S0 = eye * ys.var(axis=1)[:, :, None]
for mode, kw in (('grid', dict(s_mode='grid')), ('fixed', dict(smooth_param=10.0)), ('grid, diagonal Vs', dict(s_mode='grid', vs_diag=True))):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, ms, Vs = run_kalman_smoother(ys, m0, S0, eye, eye, eye, ev, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{mode}: {dt*1e3:.1f} ms -> {T*K/dt:.3g} units/s (outputs {ms.nbytes/1e6:.0f} + {Vs.nbytes/1e6:.0f} MB, {ms.dtype})', flush=True)
