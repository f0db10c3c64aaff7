"""Six runs of a driver on a realistic session for tools/driver_prof.sh (rocprofv3 kernel + copy statistics):
    python tools/driver_prof.py multicam | multicam_inflate | multicam_adam | singlecam | singlecam_adam"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eks_amd import MarkerArray, synth
from eks_amd.multicam_smoother import ensemble_kalman_smoother_multicam
from eks_amd.singlecam_smoother import ensemble_kalman_smoother_singlecam
which = sys.argv[1] if len(sys.argv) > 1 else 'multicam'
if which.startswith('multicam'):
    T, K, V = 50000, 4, 2
    mk = synth.multicam_markers(T, K, V=V, M=5, seed=4).astype(np.float64)
    ma = MarkerArray(mk, data_fields=['x', 'y', 'likelihood'])
    names, cams = [f'paw{i}' for i in range(K)], ['top', 'bot']
    kw = dict(quantile_keep_pca=95.0, n_latent=3)
    if which != 'multicam_adam':
        kw['smooth_param'] = [10.0]
    if which == 'multicam_inflate':
        kw['inflate_vars'] = True
    run = lambda: ensemble_kalman_smoother_multicam(ma, names, cams, **kw)
else:
    T, K = 100000, 30
    mk = synth.singlecam_markers(T, K, seed=1).astype(np.float64)
    ma = MarkerArray(mk, data_fields=['x', 'y', 'likelihood'])
    names = [f'kp{i}' for i in range(K)]
    kw = {} if which == 'singlecam_adam' else dict(smooth_param=[10.0])
    run = lambda: ensemble_kalman_smoother_singlecam(ma, names, **kw)
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run()
    torch.cuda.synchronize(); print(f'{which}: {(time.perf_counter() - t0) * 1e3:.2f} ms')
