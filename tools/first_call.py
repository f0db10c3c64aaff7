"""First-call latency of a fresh process: import, library load, the first run_kalman_smoother on
the reference's own data size (2 000 frames x 4 keypoints), the second one; per stage and per path."""
import os, sys, time
t_start = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t0 = time.perf_counter()
import torch
t1 = time.perf_counter()
from eks_amd import _lib, hip_ops
from eks_amd.core import run_kalman_smoother
t2 = time.perf_counter()
mode = sys.argv[1] if len(sys.argv) > 1 else 'diag'
warm = '--warmup' in sys.argv
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
t3 = time.perf_counter()
lib = _lib.load()
t4 = time.perf_counter()
if warm:
    hip_ops.warmup(mode)
    torch.cuda.synchronize()
t5 = time.perf_counter()
rng = np.random.default_rng(0)
T, K = 2000, 4
if mode == 'diag':
    D = O = 2
    ys = np.cumsum(rng.standard_normal((K, T, 2)), axis=1).astype(np.float32)
    ev = (rng.gamma(2.0, 0.3, (T, K, 2)) + 0.05).astype(np.float32)
    eye = np.tile(np.eye(2), (K, 1, 1)); args = (ys, np.zeros((K, 2)), eye * 4.0, eye, eye, eye, ev)
else:
    D, O = 3, 4
    x = np.cumsum(rng.standard_normal((K, T, D)) * 0.5, axis=1)
    C = rng.standard_normal((K, O, D))
    ev = (rng.gamma(2.0, 0.4, (T, K, O)) + 0.02).astype(np.float32)
    ys = (np.einsum('kod,ktd->kto', C, x) + rng.standard_normal((K, T, O)) * np.sqrt(np.swapaxes(ev, 0, 1))).astype(np.float32)
    L = rng.standard_normal((K, D, D)) * 0.3
    Q = L @ np.swapaxes(L, 1, 2) + 0.2 * np.eye(D)
    eye = np.tile(np.eye(D), (K, 1, 1)); args = (ys, np.zeros((K, D)), eye * 3.0, eye, C, Q, ev)
times = []
for rep in range(3):
    for kw in (dict(smooth_param=10.0), dict()):
        torch.cuda.synchronize(); a = time.perf_counter()
        out = run_kalman_smoother(*args, **kw)
        torch.cuda.synchronize(); times.append((rep, 'fixed' if kw else 'adam', 1e3 * (time.perf_counter() - a)))
print(f'{mode}{" +warmup" if warm else ""}: import torch {1e3*(t1-t0):.0f} ms | import eks_amd {1e3*(t2-t1):.0f} | GPU context {1e3*(t3-t2):.0f} | '
      f'load libeks_hip.so {1e3*(t4-t3):.1f} | warmup {1e3*(t5-t4):.1f} | calls: ' + ', '.join(f'{m}#{r} {t:.1f}' for r, m, t in times) + ' ms')
