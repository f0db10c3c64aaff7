"""PCIe-inclusive rate: run_kalman_smoother called with NumPy inputs (the drop-in boundary), C3 shape - pipelined over
keypoint tiles (default) and as one untiled call (EKS_HOST_UNTILED=1)."""
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
for tiled in (True, False):
    if tiled:
        os.environ.pop('EKS_HOST_UNTILED', None)
    else:
        os.environ['EKS_HOST_UNTILED'] = '1'
    for mode, kw in (('grid', dict(s_mode='grid')), ('fixed', dict(smooth_param=10.0)), ('adam', dict()),
                     ('grid, diagonal Vs', dict(s_mode='grid', vs_diag=True))):
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s, ms, Vs = run_kalman_smoother(ys, m0, S0, eye, eye, eye, ev, **kw)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            nb = (ms.nbytes, Vs.nbytes)
            del s, ms, Vs
        print(f'{"tiled" if tiled else "untiled"} {mode}: {best*1e3:.1f} ms -> {T*K/best:.3g} units/s (outputs {nb[0]/1e6:.0f} + {nb[1]/1e6:.0f} MB)', flush=True)
for tb in (40, 80, 160, 320):
    os.environ.pop('EKS_HOST_UNTILED', None)
    from eks_amd import core
    core._TILE_TARGET_BYTES = tb << 20
    best = 1e9
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, ms, Vs = run_kalman_smoother(ys, m0, S0, eye, eye, eye, ev, s_mode='grid')
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        del s, ms, Vs
    print(f'tiled grid, {tb} MB per tile: {best*1e3:.1f} ms', flush=True)
