import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from eks_amd import synth, hip_ops, _lib, core
from eks_amd.core import _DeviceProblem, _optimize_on_device
dev = torch.device('cuda', 0)
T, K = 100000, 256
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = np.tile(np.eye(2), (K, 1, 1)); m0 = np.zeros((K, 2))
S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
P = _DeviceProblem(y.transpose(0, 1), m0, S0, eye, eye, eye, var)
blocks = [[k] for k in range(K)]
guesses = np.full(K, 0.5)
import cProfile, pstats
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s, info = _optimize_on_device(P, blocks, None, guesses, 0.25, (-8.0, 8.0), 1e-2, 300, 1e-4, 'adam', 0)
    torch.cuda.synchronize(); print('total', (time.perf_counter() - t0) * 1e3, 'ms', info['launches'])
# stage timing
orig_run = hip_ops.AdamLoop.run
acc = {'run': 0.0, 'n': 0}
def timed_run(self, n):
    t = time.perf_counter(); orig_run(self, n); acc['run'] += time.perf_counter() - t; acc['n'] += 1
hip_ops.AdamLoop.run = timed_run
torch.cuda.synchronize(); t0 = time.perf_counter()
y_c, var_c = P.cropped(None); torch.cuda.synchronize(); t1 = time.perf_counter()
rc = hip_ops.const_r(var_c, 1e-4); torch.cuda.synchronize(); t2 = time.perf_counter()
print('cropped', (t1 - t0) * 1e3, 'const_r', (t2 - t1) * 1e3)
pr = cProfile.Profile(); pr.enable()
s, info = _optimize_on_device(P, blocks, None, guesses, 0.25, (-8.0, 8.0), 1e-2, 300, 1e-4, 'adam', 0)
torch.cuda.synchronize()
pr.disable(); pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
print('host time inside AdamLoop.run calls', acc['run'] * 1e3, 'ms over', acc['n'], 'calls')
