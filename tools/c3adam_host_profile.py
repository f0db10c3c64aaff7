"""cProfile of run_kalman_smoother(smooth_param=None) on device tensors, C3 shape: host microseconds per call by function
(what the device waits for between calls)."""
import os, sys, time, cProfile, pstats, io
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
for _ in range(5): step()
torch.cuda.synchronize()
N = 40
pr = cProfile.Profile(); pr.enable()
for _ in range(N): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime')
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:28]
for (fn, ln, name), (cc, nc, tt, ct, _) in rows:
    print(f'{1e6 * tt / N:8.1f} us self  {1e6 * ct / N:8.1f} us cum  {nc / N:6.1f} calls  {os.path.basename(fn)}:{ln} {name}')
