"""FETCH_SIZE probe of the fused smoother kernels: the same eks_smooth call on all-zero inputs and on random inputs,
at T = 100 000 and T = 400 000 frames (K = 256 keypoints).  Run under
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/fetch_probe.py
and read the per-dispatch counters in dispatch order (tools/micro/fetch_calib.sh does the same for the load micro)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import hip_ops, _lib
FLAGS = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC

dev = torch.device('cuda:0')
K = 256
for T in (100000, 400000):
    eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
    m0 = torch.zeros((K, 2), dtype=torch.float64, device=dev)
    s = torch.full((K,), 10.0, dtype=torch.float64, device=dev)
    for kind in ('zeros', 'random'):
        if kind == 'zeros':
            y = torch.zeros((T, K, 2), dtype=torch.float32, device=dev)
            var = torch.zeros((T, K, 2), dtype=torch.float32, device=dev)
        else:
            y = torch.randn((T, K, 2), dtype=torch.float32, device=dev)
            var = torch.rand((T, K, 2), dtype=torch.float32, device=dev) + 0.5
        for _ in range(2):
            hip_ops.smooth(y, var, m0, eye, eye, eye, eye, s, flags=FLAGS)
        torch.cuda.synchronize()
        print(T, kind, 'done', flush=True)
