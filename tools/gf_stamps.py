"""Phase timeline of the single-launch Adam loss kernel (diag_nll_grad_fused_kernel, eks_diag_nll.hip) from
in-kernel stamps of the 100 MHz real-time counter (diagnostic build: tools/build_alt.sh gfstamps -DEKS_GF_STAMPS
eks_diag_nll.hip; run with EKS_HIP_LIB=build_alt/gfstamps/libeks_hip.so).  One evaluation on C3 (T = 100 000,
K = 256), all keypoints running; times are relative to the earliest stamp of the launch."""
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
s = torch.full((K, 1), 0.5, dtype=torch.float64, device=dev)
lib = _lib.load()
for _ in range(5):
    out = hip_ops.nll(y, rconst, m0, eye * 4.0, eye, eye, eye, s, per_keypoint=True, want_grad=True, flags=_lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (256 * 8 * 8))()
lib.eks_debug_gf_stamps.restype = ctypes.c_int
assert lib.eks_debug_gf_stamps(buf) == 0
st = np.array(buf, dtype=np.float64).reshape(256, 8, 8) * 0.01           # microseconds
t0 = st[:, :, 0][st[:, :, 0] > 0].min()
st = np.where(st > 0, st - t0, np.nan)
st[:, :, 4:] = np.where(st[:, :, 4:] < st[:, :, 3:4], np.nan, st[:, :, 4:])      # stale: from an earlier launch
names = ["start", "chunk summarised", "block tree", "ticket taken", "groups composed (last)", "tile tree (last)", "end (last)"]
print('us since the first wave started: median / min / max over blocks x waves')
for i, n in enumerate(names):
    v = st[:, :, i][~np.isnan(st[:, :, i])]
    if v.size:
        print(f'   {n:28s} {np.median(v):7.2f} {v.min():7.2f} {v.max():7.2f}   n={v.size}')
d = st[:, :, 1] - st[:, :, 0]
print('summarise duration per wave: median %.2f min %.2f max %.2f' % (np.nanmedian(d), np.nanmin(d), np.nanmax(d)))
print('per wave index (median):', np.round(np.nanmedian(d, axis=0), 2))
