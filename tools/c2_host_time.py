"""configs[1] (10 000 x 64, fixed s) is a 20 us GPU problem: is a step bound by the host's enqueue cost?
Prints the host time per hip_ops.smooth call (no synchronisation inside the loop), the same through a raw
pre-built ctypes call, and the GPU time per step (events over a long loop)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eks_amd import hip_ops, synth, _lib

dev = torch.device('cuda:0')
T, K = 10_000, 64
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
S0 = torch.diag_embed(y.double().var(dim=0, unbiased=False)).contiguous()
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
s = torch.full((K,), 10.0, dtype=torch.float64, device=dev)
ms = torch.empty((T, K, 2), dtype=torch.float32, device=dev)
Vs = torch.empty((T, K, 2, 2), dtype=torch.float32, device=dev)
lib = _lib.load()
N = 3000


def loop(fn):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * t_host / N, 1e6 * (time.perf_counter() - t0) / N


a = loop(lambda: hip_ops.smooth(y, var, m0, S0, eye, eye, eye, s, flags=flags, out=(ms, Vs)))
print(f'hip_ops.smooth: host enqueue {a[0]:.1f} us / call, loop incl. drain {a[1]:.1f} us / call')
d = hip_ops._dims(K, T, 2, 2, flags)
ws = hip_ops._workspace(lib.eks_smooth_workspace_bytes(ctypes.byref(d)), dev)
args = (ctypes.byref(d), *(hip_ops._ptr(t) for t in (y, var, m0, S0, eye, eye, eye, s, ms, Vs, ws)), ws.numel(),
        hip_ops._stream())
b = loop(lambda: lib.eks_smooth(*args))
print(f'raw eks_smooth (pre-built arguments): host enqueue {b[0]:.1f} us / call, loop incl. drain {b[1]:.1f} us / call')
g = torch.cuda.CUDAGraph()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    args_s = args[:-1] + (ctypes.c_void_p(st.cuda_stream),)
    lib.eks_smooth(*args_s)
    st.synchronize()
    try:
        with torch.cuda.graph(g, stream=st):
            lib.eks_smooth(*args_s)
        c = loop(g.replay)
        print(f'hipGraph replay of the three launches: host {c[0]:.1f} us / replay, loop incl. drain {c[1]:.1f} us / replay')
    except Exception as e:  # noqa
        print('graph capture failed:', e)
