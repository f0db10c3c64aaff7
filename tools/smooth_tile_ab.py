"""A/B on one box: keypoint-tiled passes of the scalar-chain smoother (EKS_SMOOTH_TILE = 64-chain tiles per
pass) against the single pass, smooth stage alone and inside the whole C3 step (median -> NLL grid -> argmin
-> smooth); outputs must be bit-identical.  usage: python tools/smooth_tile_ab.py [T K]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import _lib, hip_ops, synth

T, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100_000, 256)
dev = torch.device('cuda', 0)
lib = _lib.load()
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
S0 = torch.diag_embed(y.double().var(dim=0, unbiased=False)).contiguous()
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
cand = torch.exp(torch.linspace(-8.0, 8.0, 64, dtype=torch.float64, device=dev))
ms = torch.empty((T, K, 2), dtype=torch.float32, device=dev)
Vs = torch.empty((T, K, 2, 2), dtype=torch.float32, device=dev)
s_fix = torch.exp(torch.linspace(-6, 6, K, dtype=torch.float64, device=dev))


def knobs(**kw):
    for k in ('EKS_SMOOTH_TILE', 'EKS_REPLAY_FORWARD', 'EKS_SUMMARIZE_REVERSE'):
        os.environ.pop(k, None)
    for k, v in kw.items():
        os.environ[k] = str(v)
    lib.eks_knobs_reload()


def smooth_only():
    hip_ops.smooth(y, var, m0, S0, eye, eye, eye, s_fix, flags=flags, out=(ms, Vs))


def step():
    rc = hip_ops.const_r(var, 1e-4)
    nll = hip_ops.nll(y, rc, m0, S0, eye, eye, eye, cand, flags=flags)
    s, _ = hip_ops.argmin_s(nll, cand)
    hip_ops.smooth(y, var, m0, S0, eye, eye, eye, s, flags=flags, out=(ms, Vs))


def timeit(fn, n=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / n * 1e6)
    return float(np.median(out)), min(out)


def stages(fn):
    import ctypes
    lib.eks_profile_drain(None, 0, None, 0)
    lib.eks_profile_enable(1)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    lib.eks_profile_enable(0)
    buf = ctypes.create_string_buffer(1 << 16)
    arr = (ctypes.c_float * 4096)()
    n = lib.eks_profile_drain(buf, len(buf), arr, 4096)
    names = buf.raw.split(b'\0')[:n]
    agg = {}
    for nm, t in zip(names, list(arr)[:n]):
        agg[nm.decode()] = agg.get(nm.decode(), 0.0) + float(t) * 1e3 / 5
    return {k: round(v, 1) for k, v in agg.items()}


knobs()
smooth_only()
ref = (ms.clone(), Vs.clone())
ntile = (2 * K + 63) // 64
configs = [dict()]
for tp in (1, 2, 4):
    if tp < ntile:
        configs += [dict(EKS_SMOOTH_TILE=tp), dict(EKS_SMOOTH_TILE=tp, EKS_REPLAY_FORWARD=1),
                    dict(EKS_SMOOTH_TILE=tp, EKS_SUMMARIZE_REVERSE=1)]
for rnd in range(2):                     # alternate twice: box drift shows up as disagreement between rounds
    for cfg in configs:
        knobs(**cfg)
        smooth_only()
        same = torch.equal(ms, ref[0]) and torch.equal(Vs, ref[1])
        a = timeit(smooth_only)
        b = timeit(step)
        print(f'round {rnd} {str(cfg):60s} identical={same} smooth {a[0]:7.1f} us (min {a[1]:7.1f})  '
              f'step {b[0]:7.1f} us (min {b[1]:7.1f})  stages {stages(step)}', flush=True)
