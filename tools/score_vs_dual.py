import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from eks_amd import _lib, hip_ops
from test_gpu_kernels import _dense_problem, _dev, _params_dev
for (T, K, D, O) in ((3000, 4, 3, 4), (900, 1200, 3, 4), (500, 3, 2, 8)):
    arrs, y, var = _dense_problem(T, K, D, O, seed=5)
    rconst = hip_ops.const_r(_dev(var), 1e-4)
    for u in (-8.0, -6.0, 0.0, 6.0, 8.0):
        s = np.full(K, np.exp(u))
        args = (_dev(y), rconst, *_params_dev(arrs), _dev(s[:, None]))
        n1, g1 = [a.cpu().numpy()[:, 0] for a in hip_ops.nll(*args, per_keypoint=True, want_grad=True, flags=_lib.FLAG_Q_PD)]
        n0, g0 = [a.cpu().numpy()[:, 0] for a in hip_ops.nll(*args, per_keypoint=True, want_grad=True, flags=0)]
        print(T, K, D, O, 'u', u, 'nll rel', np.abs(n1 - n0).max() / np.abs(n0).max(), 'grad rel to max', np.abs(g1 - g0).max() / np.abs(g0).max(), 'grad rel elementwise', (np.abs(g1 - g0) / np.abs(g0)).max())
