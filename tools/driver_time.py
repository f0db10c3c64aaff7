"""End-to-end driver time on a realistic large session: 5 members x 100k frames x 30 keypoints."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eks_amd import MarkerArray, synth
from eks_amd.singlecam_smoother import ensemble_kalman_smoother_singlecam
T, K = 100000, 30
mk = synth.singlecam_markers(T, K, seed=1).astype(np.float64)
ma = MarkerArray(mk, data_fields=['x', 'y', 'likelihood'])
names = [f'kp{i}' for i in range(K)]
ensemble_kalman_smoother_singlecam(MarkerArray(mk[:, :, :2000], data_fields=['x', 'y', 'likelihood']), names, smooth_param=[10.0])
for kw in (dict(smooth_param=[10.0]), dict(), dict(s_mode='grid')):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    df, s = ensemble_kalman_smoother_singlecam(ma, names, **kw)
    torch.cuda.synchronize(); print(kw, f'{(time.perf_counter()-t0)*1e3:.0f} ms', df.shape, flush=True)
pr = cProfile.Profile(); pr.enable()
ensemble_kalman_smoother_singlecam(ma, names, smooth_param=[10.0])
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
