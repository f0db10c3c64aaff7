"""Where an iteration of the lag-sum search goes (diagnostic build: tools/build_alt.sh lagstamps "-DEKS_LAG_STAMPS"
eks_lag_adam.hip; EKS_HIP_LIB=build_alt/lagstamps/libeks_hip.so python tools/lag_stamps.py [T K]).  Lane 0 of every chain's
wave adds up shader-clock cycles per section; printed per iteration for the median and the longest chain."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import _lib, hip_ops, synth                                            # noqa: E402

T, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 256)
dev = torch.device('cuda')
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = np.tile(np.eye(2), (K, 1, 1))
S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)   # noqa: E731
params = [f64(np.zeros((K, 2))), f64(S0), f64(eye), f64(eye), f64(eye)]
flags = hip_ops.model_flags(S0, eye, eye, eye)
rc = hip_ops.const_r(var, 1e-4)
offs = torch.arange(K + 1, dtype=torch.int32, device=dev)
mem = torch.arange(K, dtype=torch.int32, device=dev)
u0 = np.full(K, np.log(8.0))
lib = _lib.load()
for rep in range(3):
    st = np.zeros((K, 6))
    st[:, 0] = u0
    st[:, 3] = np.inf
    st = f64(st)
    s_kp = f64(np.exp(u0))
    loop = hip_ops.AdamLoop(y, rc, *params, offs, mem, st, s_kp, 0.25, -8.0, 8.0, 1e-2, 300, flags=flags)
    loop.run(300)
    torch.cuda.synchronize()
buf = np.zeros((1024, 8), dtype=np.uint64)
fn = lib.eks_debug_lag_stamps
fn.argtypes = [ctypes.c_void_p]
assert fn(buf.ctypes.data) == 0
n = min(1024, 2 * K)
b = buf[:n].astype(np.float64)
its = b[:, 5]
names = ('constants (exp, Riccati)', 'evaluation', 'exchange + barrier', 'Adam step + stop rule', 'loop top')
print(f'T={T} K={K}: {n} chains, iterations {its.min():.0f} .. {its.max():.0f}; cycles per iteration (median chain / longest chain)')
long_ = int(np.argmax(b[:, 6]))
for i, nm in zip((4, 0, 1, 2, 3), (names[4], names[0], names[1], names[2], names[3])):
    print(f'   {nm:28s} {np.median(b[:, i] / its):8.0f} {b[long_, i] / its[long_]:8.0f}')
print(f'   whole wave (cycles)          {np.median(b[:, 6]):8.0f} {b[long_, 6]:8.0f}   prologue + epilogue of the longest: '
      f'{b[long_, 6] - b[long_, :5].sum():.0f}')
