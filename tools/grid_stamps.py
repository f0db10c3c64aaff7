"""Role timeline of diag_nll_grid_kernel (eks_diag_nll.hip) from in-kernel stamps of the 100 MHz real-time counter
(diagnostic build: tools/build_alt.sh gridstamps -DEKS_GRID_STAMPS eks_diag_nll.hip; run with
EKS_HIP_LIB=build_alt/gridstamps/libeks_hip.so).  One grid search on C3 (T = 100 000, K = 256, 64 candidates)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import synth, hip_ops, _lib
T, K = 100_000, 256
dev = torch.device('cuda', 0)
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
rconst = hip_ops.const_r(var)
cand = torch.exp(torch.linspace(-8.0, 8.0, 64, dtype=torch.float64, device=dev))
lib = _lib.load()
for _ in range(5):
    out = hip_ops.nll_argmin(y, rconst, m0, eye * 4.0, eye, eye, eye, cand, flags=_lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2048 * 4 * 4))()
lib.eks_debug_grid_stamps.restype = ctypes.c_int
assert lib.eks_debug_grid_stamps(buf) == 0
st = np.array(buf, dtype=np.float64).reshape(2048, 4, 4)
ok = st[:, :, 1] > 0
t0 = st[:, :, 0][ok].min() * 0.01
names = {0: 'head', 1: 'lean (round 4)', 2: 'lag form', 3: 'exact-entry fallback'}
print('role: waves, start (min / max), end (median / p90 / max), duration (median / max) in us since the first wave')
for r, nm in names.items():
    m = ok & (st[:, :, 2] == r)
    if not m.any():
        continue
    a = st[:, :, 0][m] * 0.01 - t0
    b = st[:, :, 1][m] * 0.01 - t0
    print(f'  {nm:22s} {m.sum():5d}  start {a.min():6.1f} / {a.max():6.1f}   end {np.median(b):6.1f} / {np.percentile(b, 90):6.1f} / {b.max():6.1f}'
          f'   duration {np.median(b - a):6.1f} / {(b - a).max():6.1f}')
m = ok & (st[:, :, 2] == 0)
if m.any():
    d = (st[:, :, 1] - st[:, :, 0]) * 0.01
    blocks = np.where(m.any(axis=1))[0]
    print('head waves by candidate group (4 candidates each; duration, us):')
    per = d[blocks].reshape(-1)            # wave index hw = block * 4 + w; tile = hw // 16, g = hw % 16
    hw = (blocks[:, None] * 4 + np.arange(4)[None]).reshape(-1)
    for g in range(16):
        sel = (hw % 16) == g
        print(f'    group {g:2d}: {np.median(per[sel]):6.1f}  (max {per[sel].max():6.1f})')
