"""Very long sequences (3 M frames x 8 keypoints): index arithmetic, workspace sizes, parity."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import c_oracle
from eks_amd import hip_ops, synth, _lib
dev = torch.device('cuda', 0)
T, K = 3_000_000, 8
y, var = synth.singlecam_observations_torch(T, K, seed=9, device=dev)
eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
S0 = torch.diag_embed(y.double().var(dim=0, unbiased=False)).contiguous()
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
cand = torch.exp(torch.linspace(-8.0, 8.0, 16, dtype=torch.float64, device=dev))
torch.cuda.synchronize(); t0 = time.perf_counter()
rc = hip_ops.const_r(var, 1e-4)
nll = hip_ops.nll(y, rc, m0, S0, eye, eye, eye, cand, flags=flags)
s, idx = hip_ops.argmin_s(nll, cand)
ms, Vs = hip_ops.smooth(y, var, m0, S0, eye, eye, eye, s, flags=flags, vs_diag=True)
torch.cuda.synchronize(); print(f'GPU: {(time.perf_counter()-t0)*1e3:.1f} ms; |y| max {float(y.abs().max()):.0f}')
yk = np.transpose(y.cpu().numpy().astype(np.float64), (1, 0, 2)).copy()
vk = np.clip(np.transpose(var.cpu().numpy().astype(np.float64), (1, 0, 2)), 1e-12, None).copy()
e = np.tile(np.eye(2), (K, 1, 1)); z = np.zeros((K, 2)); S0n = S0.cpu().numpy()
ref = c_oracle.nll_grid(yk, rc.cpu().numpy(), z, S0n, e, e, e, cand.cpu().numpy())
print('nll rel err', (np.abs(nll.cpu().numpy() - ref) / np.abs(ref)).max(), 'argmin equal', np.array_equal(ref.argmin(1), idx.cpu().numpy()))
mo, Vo, _ = c_oracle.smooth(yk, vk, z, S0n, e, e, e, s.cpu().numpy())
msk = np.transpose(ms.cpu().numpy().astype(np.float64), (1, 0, 2)); Vk = np.transpose(Vs.cpu().numpy().astype(np.float64), (1, 0, 2))
print('ms rel-to-scale', (np.abs(msk - mo) / np.abs(mo).max(axis=(1, 2), keepdims=True)).max(), 'abs px', np.abs(msk - mo).max(),
      'Vs rel', (np.abs(Vk - np.diagonal(Vo, axis1=2, axis2=3)) / np.diagonal(Vo, axis1=2, axis2=3)).max())
