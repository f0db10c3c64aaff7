import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eks_amd import synth, core
from eks_amd.core import run_kalman_smoother
T, K = 100000, 256
dev = torch.device('cuda', 0)
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
ys = np.ascontiguousarray(np.transpose(y.cpu().numpy(), (1, 0, 2))); ev = var.cpu().numpy()
eye = np.tile(np.eye(2), (K, 1, 1)); m0 = np.zeros((K, 2)); S0 = eye * ys.var(axis=1)[:, :, None]
def sync(): torch.cuda.synchronize()
for kw in (dict(smooth_param=10.0), dict(s_mode='grid')):
    for rep in range(3):
        # one tile's device call, timed stage by stage (synchronised)
        k0, k1 = 0, 32
        sync(); t0 = time.perf_counter()
        y_t = torch.as_tensor(np.ascontiguousarray(ys[k0:k1]), device=dev); sync(); t1 = time.perf_counter()
        v_t = torch.as_tensor(np.ascontiguousarray(ev[:, k0:k1]), device=dev); sync(); t2 = time.perf_counter()
        res = run_kalman_smoother(y_t, m0[k0:k1], S0[k0:k1], eye[k0:k1], eye[k0:k1], eye[k0:k1], v_t, return_device=True, return_info=True, _s_on_device=True, **kw); sync(); t3 = time.perf_counter()
        a = res[1].contiguous(); b = res[2].contiguous(); sync(); t4 = time.perf_counter()
        print(f'{kw}: y up {1e3*(t1-t0):.2f}  var gather+up {1e3*(t2-t1):.2f}  device call {1e3*(t3-t2):.2f}  transposes {1e3*(t4-t3):.2f} ms', flush=True)
    st = torch.cuda.Stream()
    for rep in range(2):
        sync(); t0 = time.perf_counter()
        with torch.cuda.stream(st):
            res = run_kalman_smoother(y_t, m0[k0:k1], S0[k0:k1], eye[k0:k1], eye[k0:k1], eye[k0:k1], v_t, return_device=True, return_info=True, _s_on_device=True, **kw)
        sync(); print(f'   same call on a side stream: {1e3*(time.perf_counter()-t0):.2f} ms', flush=True)
t0 = time.perf_counter(); s3 = [torch.cuda.Stream() for _ in range(3)]; print('3 streams created in', 1e3*(time.perf_counter()-t0), 'ms')
