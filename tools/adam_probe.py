import sys, time, numpy as np, torch, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from eks_amd import synth, hip_ops, _lib
from eks_amd.core import _DeviceProblem, _optimize_on_device
dev = torch.device('cuda', 0)
T, K = 100000, 256
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = np.tile(np.eye(2), (K, 1, 1)); m0 = np.zeros((K, 2))
S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
P = _DeviceProblem(y.transpose(0, 1), m0, S0, eye, eye, eye, var)
blocks = [[k] for k in range(K)]
guesses = np.full(K, 0.5)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s, info = _optimize_on_device(P, blocks, None, guesses, 0.25, (-8.0, 8.0), 1e-2, 300, 1e-4, 'adam', 0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = info['state'].cpu().numpy()
    print(f'adam: {dt*1e3:.1f} ms launches={info["launches"]}')
it = st[:, 4]
print('iters sorted', np.sort(it).astype(int).tolist())
print('per-tile (32 kp) max', it.reshape(8, 32).max(axis=1), 'sum of tile maxes / 8', it.reshape(8,32).max(axis=1).mean())
print('active-iteration sum: chain-level', it.sum(), 'of', it.max()*K)
