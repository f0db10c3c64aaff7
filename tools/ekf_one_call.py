import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from eks_amd import hip_ops, synth
T, K, V = 50_000, 16, 4
prob = synth.calibrated_multicam(T, K, V, seed=4)
dev = torch.device('cuda')
t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
y, var = t(prob['y_tko'], torch.float32), t(prob['var_tko'], torch.float32)
m0, S0, A, Q = t(prob['m0s']), t(prob['S0s']), t(prob['As']), t(prob['Qs'])
cams = t(prob['cams_packed'])
cold = m0[:, None, :].expand(K, T, 3).contiguous()
s = torch.full((K,), 0.1, dtype=torch.float64, device=dev)
for _ in range(6):
    x = cold.clone()
    hip_ops.ekf_smooth(y, var, None, m0, S0, A, Q, s, cams, x, max_sweeps=16, tol=1e-10)
torch.cuda.synchronize()
