"""eks_ensemble on the C3 shape: M=5 members x 100k frames x 256 keypoints (80 B per frame*keypoint)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import hip_ops
dev = torch.device('cuda', 0)
M, V, T, K = 5, 1, 100000, 256
g = torch.Generator(device=dev); g.manual_seed(0)
mk = torch.rand((M, V, T, K, 3), device=dev, generator=g, dtype=torch.float32) * 100
for mode in (('median', 'confidence_weighted_var'), ('mean', 'var')):
    for _ in range(2): out = hip_ops.ensemble(mk, *mode)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = hip_ops.ensemble(mk, *mode)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    by = mk.numel() * 4 + out.numel() * 4
    print(f'{mode}: {dt*1e3:.3f} ms, {by/dt/1e12:.2f} TB/s ({by/1e9:.2f} GB)', flush=True)
