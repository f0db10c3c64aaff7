"""Adam mode on a long sequence (20 000 frames x 4 keypoints) against the NumPy oracle's optimiser."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from oracle import eks_oracle as orc
import test_gpu_kernels as tg
from eks_amd.core import run_kalman_smoother
T, K = 20000, 4
arrs, y, var = tg._singlecam_problem(T, K, seed=123, unit=True)
args = (arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'], arrs['ensemble_vars'])
t0 = time.perf_counter(); s, ms, Vs = run_kalman_smoother(*args, return_device=False); t1 = time.perf_counter()
so, mo, Vo, info = orc.run_kalman_smoother(*args); t2 = time.perf_counter()
print(f'GPU {1e3*(t1-t0):.0f} ms, oracle {t2-t1:.0f} s, iters {info["iters"]}')
print('s gpu', s, '\ns orc', so, '\n|dlog s| max', np.abs(np.log(s) - np.log(so)).max(), '|y| max', np.abs(arrs['ys']).max())
