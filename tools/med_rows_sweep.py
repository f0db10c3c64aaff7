"""eks_const_r (the exact time-median of the variances) for the rows-per-wave choices of its full pass
(EKS_MED_ROWS, read once per process: run once per value).  Prints the time of the whole call per shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eks_amd import hip_ops
dev = torch.device('cuda:0')
for T, K in ((10_000, 64), (100_000, 256), (50_000, 4096), (3_000, 30), (400_000, 16)):
    g = torch.Generator(device=dev).manual_seed(1)
    var = torch.rand((T, K, 2), generator=g, device=dev).mul_(0.6).add_(0.05)
    for _ in range(5):
        hip_ops.const_r(var, 1e-4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        hip_ops.const_r(var, 1e-4)
    torch.cuda.synchronize()
    print(f'EKS_MED_ROWS={os.environ.get("EKS_MED_ROWS", "auto")} T={T} K={K}: {1e6 * (time.perf_counter() - t0) / n:.1f} us', flush=True)
