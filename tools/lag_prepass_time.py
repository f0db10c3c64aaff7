"""Times the three kernels of the lag-sum search on the C3 shape (HIP-event profile of one eks_adam_run call)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import _lib, hip_ops, synth
import ctypes
T, K = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda')
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = np.tile(np.eye(2), (K, 1, 1))
S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
params = [f64(np.zeros((K, 2))), f64(S0), f64(eye), f64(eye), f64(eye)]
flags = hip_ops.model_flags(S0, eye, eye, eye)
rc = hip_ops.const_r(var, 1e-4)
offs = torch.arange(K + 1, dtype=torch.int32, device=dev); mem = torch.arange(K, dtype=torch.int32, device=dev)
u0 = np.full(K, np.log(8.0))
lib = _lib.load()
def once():
    st = np.zeros((K, 6)); st[:, 0] = u0; st[:, 3] = np.inf
    st = f64(st); s_kp = f64(np.exp(u0))
    loop = hip_ops.AdamLoop(y, rc, *params, offs, mem, st, s_kp, 0.25, -8.0, 8.0, 1e-2, 300, flags=flags)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loop.run(300); torch.cuda.synchronize()
    return time.perf_counter() - t0, st
for _ in range(3): once()
dts = [once()[0] for _ in range(10)]
print(f'T={T} K={K}: whole eks_adam_run call {1e3*np.median(dts):.3f} ms (min {1e3*min(dts):.3f})')
lib.eks_profile_enable(1)
_, st = once()
lib.eks_profile_enable(0)
import bench
prof = bench.drain_profile(lib)
it = st.cpu().numpy()[:, 4]
print('iterations', it.min(), it.mean(), it.max())
for k, v in prof.items():
    print(k, v)
