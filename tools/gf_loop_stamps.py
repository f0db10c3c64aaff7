"""Per-iteration timeline of the in-launch optimiser loop (diag_nll_grad_fused_kernel<., true>, eks_diag_nll.hip: GfLoop)
from in-kernel stamps (diagnostic build: tools/build_alt.sh gfstamps -DEKS_GF_STAMPS eks_diag_nll.hip; run with
EKS_HIP_LIB=build_alt/gfstamps/libeks_hip.so).  C3 (T = 100 000, K = 256) unless `iterations T K` are given; one call."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import synth, hip_ops, _lib
T, K = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (100_000, 256)
NI = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda', 0)
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
rconst = hip_ops.const_r(var)
lib = _lib.load()
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
def fresh():
    state = torch.zeros(K, 6, dtype=torch.float64, device=dev); state[:, 0] = np.log(80.0); state[:, 3] = float('inf')
    s = torch.full((K,), 80.0, dtype=torch.float64, device=dev)
    offs = torch.arange(K + 1, dtype=torch.int32, device=dev); mem = torch.arange(K, dtype=torch.int32, device=dev)
    return hip_ops.AdamLoop(y.view(T, K, 2), rconst, m0, eye * 4.0, eye, eye, eye, offs, mem, state, s, 0.25, -8.0, 8.0, 1e-2, 300, flags=flags)
for _ in range(2):
    lp = fresh(); lp.run(NI); torch.cuda.synchronize()
assert int(lp.n_active.item()) >= 0
buf = (ctypes.c_ulonglong * (256 * 48 * 8))()
lib.eks_debug_gf_iter_stamps.restype = ctypes.c_int
assert lib.eks_debug_gf_iter_stamps(buf) == 0
st = np.array(buf, dtype=np.float64).reshape(256, 48, 8) * 0.01
t0 = st[:, 0, 0].min()
st = np.where(st > 0, st - t0, np.nan)
names = ['start', 'chunk done', 'block sum', 'ticket', 'slots read (last)', 'tile sum (last)', 'stepped (last)', 'next s known']
ntile = (2 * K + 63) // 64
tile = np.arange(256) % ntile
print('iteration: per phase the median over the blocks of tile 0 (us since the launch\'s first stamp); cycle = start(it+1) - start(it)')
for it in range(min(NI, 12)):
    row = [np.nanmedian(st[tile == 0, it, ph]) for ph in range(8)]
    print(f'  it {it:2d} ' + ' '.join(f'{n}={v:7.2f}' for n, v in zip(names, row)))
starts = np.nanmedian(st[tile == 0, :NI, 0], axis=0)
print('cycle per iteration, tile 0:', np.round(np.diff(starts), 1))
for tl in range(ntile):
    stt = np.nanmedian(st[tile == tl, :NI, 0], axis=0)
    print(f'tile {tl}: start of iterations 0, 8, 16, 24: {np.round(stt[[0, 8, 16, 24]], 1)}  mean cycle {np.nanmean(np.diff(stt)):.2f}')
d = st[:, :NI, 1] - st[:, :NI, 0]
print('chunk duration (wave 0): median per iteration', np.round(np.nanmedian(d, axis=0)[:16], 1))
w = st[:, :NI, 7] - st[:, :NI, 3]
print('ticket -> next s known: median per iteration', np.round(np.nanmedian(w, axis=0)[:16], 1))
grp = np.arange(256) // ntile
print('chunk duration of wave 0, group 0 (chunk 0: known entry state) vs the other groups, per iteration:')
print('   group 0 :', np.round(np.nanmedian(d[grp == 0], axis=0)[:20], 1))
print('   others  :', np.round(np.nanmedian(d[grp != 0], axis=0)[:20], 1))
print('   max     :', np.round(np.nanmax(d, axis=0)[:20], 1), 'argmax block', np.nanargmax(d, axis=0)[:20])
