"""Stress of the in-launch optimiser loop (eks_diag_nll.hip: GfLoop): 300 searches on C3, each in calls of 37 + 128 + 135
iterations - optimiser state and s must come out identical every time, nothing may be left running or give up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from eks_amd import synth, hip_ops, _lib
T, K = 100_000, 256
dev = torch.device('cuda', 0)
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
rconst = hip_ops.const_r(var)
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
ref = None
t0 = time.time()
for rep in range(300):
    state = torch.zeros(K, 6, dtype=torch.float64, device=dev); state[:, 0] = np.log(80.0); state[:, 3] = float('inf')
    s = torch.full((K,), 80.0, dtype=torch.float64, device=dev)
    offs = torch.arange(K + 1, dtype=torch.int32, device=dev); mem = torch.arange(K, dtype=torch.int32, device=dev)
    lp = hip_ops.AdamLoop(y.view(T, K, 2), rconst, m0, eye * 4.0, eye, eye, eye, offs, mem, state, s, 0.25, -8.0, 8.0, 1e-2, 300, flags=flags)
    for n in (37, 128, 135): lp.run(n)
    out = (state.clone(), s.clone(), int(lp.n_active.item()))
    assert out[2] == 0, out[2]
    if ref is None: ref = out
    else: assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), rep
print('300 searches on C3 (calls of 37 + 128 + 135 iterations): identical every time,', round(time.time() - t0, 1), 's')
