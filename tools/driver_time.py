"""End-to-end driver times on realistic large sessions (MarkerArray in -> DataFrames out, including
uploads, downloads and the host-side fits):
  singlecam  5 members x 100 000 frames x 30 keypoints
  multicam   BASELINE.json configs[3]: 5 members x 2 views x 50 000 frames x 4 paws (linear path,
             n_latent 3, quantile 95), device-resident pipeline vs the host pipeline
             (EKS_HOST_DRIVER=1), with and without variance inflation
    python tools/driver_time.py [--profile]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eks_amd import MarkerArray, synth
from eks_amd.multicam_smoother import ensemble_kalman_smoother_multicam
from eks_amd.singlecam_smoother import ensemble_kalman_smoother_singlecam


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


T, K = 100000, 30
mk = synth.singlecam_markers(T, K, seed=1).astype(np.float64)
ma = MarkerArray(mk, data_fields=['x', 'y', 'likelihood'])
names = [f'kp{i}' for i in range(K)]
ensemble_kalman_smoother_singlecam(MarkerArray(mk[:, :, :2000], data_fields=['x', 'y', 'likelihood']), names,
                                   smooth_param=[10.0])
for kw in (dict(smooth_param=[10.0]), dict(), dict(s_mode='grid')):
    ms, (df, s) = timed(lambda: ensemble_kalman_smoother_singlecam(ma, names, **kw))
    print(f'singlecam 100k x 30 x 5 {kw}: {ms:.0f} ms', df.shape, flush=True)

T, K, V = 50000, 4, 2
mk2 = synth.multicam_markers(T, K, V=V, M=5, seed=4).astype(np.float64)
ma2 = MarkerArray(mk2, data_fields=['x', 'y', 'likelihood'])
names2, cams = [f'paw{i}' for i in range(K)], ['top', 'bot']
for kw in (dict(smooth_param=[10.0]), dict(smooth_param=[10.0], inflate_vars=True), dict()):
    args = dict(quantile_keep_pca=95.0, n_latent=3, **kw)
    os.environ.pop('EKS_HOST_DRIVER', None)
    ensemble_kalman_smoother_multicam(ma2, names2, cams, **args)
    ms_d, _ = timed(lambda: ensemble_kalman_smoother_multicam(ma2, names2, cams, **args))
    os.environ['EKS_HOST_DRIVER'] = '1'
    ms_h, _ = timed(lambda: ensemble_kalman_smoother_multicam(ma2, names2, cams, **args), reps=2)
    os.environ.pop('EKS_HOST_DRIVER', None)
    print(f'multicam C4 50k x 2 views x 4 paws x 5 {kw}: device pipeline {ms_d:.0f} ms, host pipeline {ms_h:.0f} ms',
          flush=True)

if '--profile' in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    ensemble_kalman_smoother_multicam(ma2, names2, cams, smooth_param=[10.0], quantile_keep_pca=95.0, n_latent=3)
    pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
