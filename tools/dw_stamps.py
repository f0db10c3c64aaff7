"""Phase timeline of the narrow-session dense kernels (eks_dense_wave.hip) from in-kernel stamps of the
100 MHz real-time counter (diagnostic build: tools/build_alt.sh stamps -DEKS_DW_STAMPS eks_dense_wave.hip;
run with EKS_HIP_LIB=build_alt/stamps/libeks_hip.so).  Prints per phase the median over the first 64 blocks."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import _lib, hip_ops
T, K, D, O = 50_000, 4, 3, 4
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev); g.manual_seed(4)
lat = torch.cumsum(torch.randn(T, K, D, device=dev, generator=g) * 0.7, dim=0)
C = torch.linalg.qr(torch.randn(K, O, D, device=dev, generator=g, dtype=torch.float64))[0].contiguous()
var = (0.25 * (-torch.log(torch.rand(T, K, O, device=dev, generator=g).clamp_min(1e-12)))).clamp_min(1e-3).float().contiguous()
y = (torch.einsum('kod,tkd->tko', C.float(), lat) + torch.randn(T, K, O, device=dev, generator=g) * var.sqrt()).float().contiguous()
eye = torch.eye(D, dtype=torch.float64, device=dev).expand(K, D, D).contiguous()
m0 = torch.zeros(K, D, dtype=torch.float64, device=dev)
s = torch.full((K,), 10.0, dtype=torch.float64, device=dev)
for _ in range(5):
    hip_ops.smooth(y, var, m0, eye * 4.0, eye, C, eye, s)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (2 * 64 * 16))()
lib.eks_debug_dw_stamps.restype = ctypes.c_int
assert lib.eks_debug_dw_stamps(buf) == 0
st = np.array(buf, dtype=np.float64).reshape(2, 64, 16) * 0.01          # microseconds
names = [['start', 'model loaded', 'element built', 'scanned', 'stored'],
         ['start', 'model + requests', 'prior through earlier blocks', 'barrier passed', 'boundary ops', 'filtered',
          'last frame smoothed', 'RTS + stores done']]
for kern, nm in enumerate(names):
    t = st[kern][:, :len(nm)]
    d = np.diff(t, axis=1)
    print(['dw_summarize', 'dw_replay'][kern], 'wave 0, us per phase (median over blocks; min..max):')
    for i, n in enumerate(nm[1:]):
        print(f'   {n:32s} {np.median(d[:, i]):7.2f}   ({d[:, i].min():.2f} .. {d[:, i].max():.2f})')
    print(f'   total                            {np.median(t[:, -1] - t[:, 0]):7.2f}')
