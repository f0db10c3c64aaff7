"""Does the relative placement of y, var, ms, Vs in HBM change the smoother's time?  C5's per-GPU share
(T = 50 000, N = 8192 chains: row strides of 32 and 64 KiB) showed K3 at 1.37 or 1.55 ms from process to process
with the same binary.  One arena, the four arrays carved out of it at chosen byte offsets past their natural
positions; eks_smooth timed per layout (HIP events around 10 calls).
    python tools/placement_probe.py [T K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eks_amd import hip_ops, _lib

T, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50000, 4096)
dev = torch.device('cuda:0')
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
n_y, n_v = T * K * 2, T * K * 4
slack = 1 << 22                                               # floats of slack between arrays
arena = torch.empty(2 * n_y + n_y + n_v + 8 * slack, dtype=torch.float32, device=dev)
print('arena base %#x' % arena.data_ptr())
eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
s = torch.full((K,), 10.0, dtype=torch.float64, device=dev)


def carve(offsets_bytes):
    out, pos = [], 0
    for n, off in zip((n_y, n_y, n_y, n_v), offsets_bytes):
        start = pos + off // 4
        out.append(arena[start:start + n])
        pos = start + n + slack
        pos = (pos + 63) // 64 * 64
    return out


def run(label, offsets):
    y, var, ms, Vs = carve(offsets)
    y = y.view(T, K, 2); var = var.view(T, K, 2); ms = ms.view(T, K, 2); Vs = Vs.view(T, K, 2, 2)
    y.normal_(); var.uniform_(0.5, 1.5)
    for _ in range(2):
        hip_ops.smooth(y, var, m0, eye, eye, eye, eye, s, flags=flags, out=(ms, Vs))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        hip_ops.smooth(y, var, m0, eye, eye, eye, eye, s, flags=flags, out=(ms, Vs))
    b.record(); torch.cuda.synchronize()
    print('%-44s %.3f ms per eks_smooth' % (label, a.elapsed_time(b) / 10), flush=True)


KB = 1024
run('natural (all offsets 0)', (0, 0, 0, 0))
run('var +256 B', (0, 256, 0, 0))
run('var +4 KiB, ms +8 KiB, Vs +12 KiB', (0, 4 * KB, 8 * KB, 12 * KB))
run('var +64 KiB, ms +128 KiB, Vs +192 KiB', (0, 64 * KB, 128 * KB, 192 * KB))
run('var +1 MiB, ms +2 MiB, Vs +3 MiB', (0, 1024 * KB, 2048 * KB, 3072 * KB))
run('var +1.3 KiB, ms +2.8 KiB, Vs +5.1 KiB (odd)', (0, 1280, 2816, 5120 + 256))
run('natural again', (0, 0, 0, 0))
