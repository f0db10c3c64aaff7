import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import synth, hip_ops, _lib
from eks_amd.core import _DeviceProblem, _optimize_on_device
dev = torch.device('cuda', 0)
for (T, K) in ((100000, 256), (10000, 64)):
    y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
    eye = np.tile(np.eye(2), (K, 1, 1)); m0 = np.zeros((K, 2))
    S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
    P = _DeviceProblem(y.transpose(0, 1), m0, S0, eye, eye, eye, var)
    blocks = [[k] for k in range(K)]
    guesses = np.full(K, 0.5)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, info = _optimize_on_device(P, blocks, None, guesses, 0.25, (-8.0, 8.0), 1e-2, 300, 1e-4, 'adam', 0)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = info['state'].cpu().numpy()
        print(f'T={T} K={K} adam: {dt*1e3:.1f} ms, launches={info["launches"]}, iters min/mean/max = {st[:,4].min():.0f}/{st[:,4].mean():.1f}/{st[:,4].max():.0f}, per-iter {dt*1e3/info["launches"]:.3f} ms')
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s2, info2 = _optimize_on_device(P, blocks, None, guesses, 0.25, (-8.0, 8.0), 1e-2, 300, 1e-4, 'grid', 64)
    torch.cuda.synchronize(); print('grid', (time.perf_counter()-t0)*1e3, 'ms', 'median |dlog s| adam vs grid', float(np.median(np.abs(np.log(s.cpu().numpy()) - np.log(s2.cpu().numpy())))))
