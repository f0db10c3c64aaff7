import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from eks_amd import synth, core
from eks_amd.core import run_kalman_smoother
T, K = 100000, 256
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=torch.device('cuda', 0))
ys = np.ascontiguousarray(np.transpose(y.cpu().numpy(), (1, 0, 2))); ev = var.cpu().numpy()
eye = np.tile(np.eye(2), (K, 1, 1)); m0 = np.zeros((K, 2)); S0 = eye * ys.var(axis=1)[:, :, None]
def run(label, reps=5, **kw):
    best = 1e9
    for rep in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, ms, Vs = run_kalman_smoother(ys, m0, S0, eye, eye, eye, ev, **kw)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        del s, ms, Vs
    print(f'{label}: {best*1e3:.1f} ms', flush=True)
for kp in (8, 12, 16, 21, 26, 32, 43, 64):
    os.environ['EKS_HOST_TILE_KP'] = str(kp)
    run(f'equal tiles of {kp} keypoints', s_mode='grid')
os.environ.pop('EKS_HOST_TILE_KP')
core._TILE_TARGET_BYTES = 200 << 20
run('growing tiles from 32', s_mode='grid')
os.environ['EKS_HOST_VAR_WHOLE'] = '1'
core._TILE_TARGET_BYTES = 80 << 20
run('whole variances first, 80 MB tiles', s_mode='grid')
os.environ.pop('EKS_HOST_VAR_WHOLE', None)
run('fixed s, default tiles', smooth_param=10.0)
run('grid, diagonal Vs, default tiles', s_mode='grid', vs_diag=True)
