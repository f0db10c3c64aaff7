"""Where the wall time of one `bench.py --workload c3adam` step goes on the host: cProfile around run_kalman_smoother
on device tensors (smooth_param=None)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eks_amd import synth
from eks_amd.core import run_kalman_smoother
T, K = 100000, 256
dev = torch.device('cuda', 0)
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = np.tile(np.eye(2), (K, 1, 1)); m0 = np.zeros((K, 2))
S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
y_kt = y.transpose(0, 1)
def step():
    return run_kalman_smoother(y_kt, m0, S0, eye, eye, eye, var, smooth_param=None, return_device=True, return_info=True)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print('step', 1e3 * (time.perf_counter() - t0), 'ms')
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
