"""How much of diag_nll_summarize is transient work?  Times eks_nll on the C3 shape for candidate
grids that are all fast (large s), all slow (small s) and the real 64-point grid."""
import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import synth, hip_ops, _lib
dev = torch.device('cuda', 0)
T, K = 100000, 256
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
S0 = torch.diag_embed(y.double().var(dim=0, unbiased=False)).contiguous()
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
rc = hip_ops.const_r(var, 1e-4)
lib = _lib.load()
def drain():
    buf = ctypes.create_string_buffer(1 << 16); ms = (ctypes.c_float * 4096)()
    n = lib.eks_profile_drain(buf, len(buf), ms, 4096)
    names = buf.raw.split(b'\0')[:n]; out = {}
    for nm, t in zip(names, list(ms)[:n]): out.setdefault(nm.decode(), []).append(float(t))
    return {k: float(np.mean(v)) for k, v in out.items()}
for name, lo, hi in (('real grid -8..8', -8, 8), ('fast 2..8', 2, 8), ('mid -2..2', -2, 2), ('slow -8..-6', -8, -6),
                     ('slowest -8', -8, -8), ('-4', -4, -4), ('0', 0, 0), ('2', 2, 2), ('4', 4, 4), ('6', 6, 6), ('8', 8, 8),
                     ('-1..1', -1, 1), ('3..5', 3, 5), ('-6..-4', -6, -4), ('-8..0', -8, 0), ('0..8', 0, 8)):
    cand = torch.exp(torch.linspace(lo, hi, 64, dtype=torch.float64, device=dev))
    for _ in range(3): hip_ops.nll(y, rc, m0, S0, eye, eye, eye, cand, flags=flags)
    torch.cuda.synchronize(); drain(); lib.eks_profile_enable(1)
    for _ in range(10): hip_ops.nll(y, rc, m0, S0, eye, eye, eye, cand, flags=flags)
    torch.cuda.synchronize(); lib.eks_profile_enable(0)
    print(name, drain(), flush=True)

# Is the in-step slowdown memory residency (y served from the 256 MB Infinity Cache when the kernel runs back to back)
# or clock state?  The same launches with a 1 GB streaming kernel between them (evicts y), and with a compute-only
# kernel between them.
cand = torch.exp(torch.linspace(-8, 8, 64, dtype=torch.float64, device=dev))
big = torch.zeros(256 * 1024 * 1024, dtype=torch.float32, device=dev)
small = torch.randn(1 << 20, dtype=torch.float32, device=dev)
for name, between in (('back to back', lambda: None), ('1 GB stream between', lambda: big.add_(1.0)),
                      ('compute-only between', lambda: [small.mul_(1.0001) for _ in range(20)]),
                      ('back to back again', lambda: None)):
    for _ in range(3):
        hip_ops.nll(y, rc, m0, S0, eye, eye, eye, cand, flags=flags); between()
    torch.cuda.synchronize(); drain(); lib.eks_profile_enable(1)
    for _ in range(10):
        hip_ops.nll(y, rc, m0, S0, eye, eye, eye, cand, flags=flags); between()
    torch.cuda.synchronize(); lib.eks_profile_enable(0)
    print(name, drain(), flush=True)
