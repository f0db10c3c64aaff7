import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import synth, hip_ops, _lib
dev = torch.device('cuda', 0)
T, K = 100000, 256
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
for _ in range(3): hip_ops.const_r(var, 1e-4)
torch.cuda.synchronize()
