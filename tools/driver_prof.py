import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from eks_amd import MarkerArray, synth
from eks_amd.multicam_smoother import ensemble_kalman_smoother_multicam
T, K, V = 50000, 4, 2
mk2 = synth.multicam_markers(T, K, V=V, M=5, seed=4).astype(np.float64)
ma2 = MarkerArray(mk2, data_fields=['x', 'y', 'likelihood'])
names2, cams = [f'paw{i}' for i in range(K)], ['top', 'bot']
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ensemble_kalman_smoother_multicam(ma2, names2, cams, smooth_param=[10.0], quantile_keep_pca=95.0, n_latent=3)
    torch.cuda.synchronize(); print((time.perf_counter() - t0) * 1e3, 'ms')
