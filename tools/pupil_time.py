"""Timing of the pupil optimiser + smoother on the GPU (golden ibl-pupil session and synthetic
sessions of growing length).  Usage: python tools/pupil_time.py [cap]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import synth, hip_ops
from eks_amd import ibl_pupil_smoother as ips

cap = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for T in (2000, 20000, 200000):
    ys, ev, m0, S0, lv = synth.pupil_observations(T, seed=1)
    P = ips._PupilProblem(ys, m0, S0, ips.PUPIL_C, ev, lv)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s_d, s_c, info = ips._optimize_on_device(P, None, 5e-3, 1e-6, cap)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'T={T}: optimise {dt*1e3:.1f} ms, iters={info["iters"]} launched={info["launched"]} '
              f'({dt*1e6/info["launched"]:.1f} us/iter) s=({s_d:.6f},{s_c:.6f}) conv={info["converged"]}', flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = ips.run_pupil_kalman_smoother(ys, m0, S0, ips.PUPIL_C, ev, *lv, smooth_params=[s_d, s_c])
    torch.cuda.synchronize(); print(f'   smooth (incl. H2D/D2H) {(time.perf_counter()-t0)*1e3:.1f} ms', flush=True)
    loss = hip_ops.Ar1Loss(P.y, P.var, P.m0, P.S0, P.C, n_tan=2)
    loss.a.copy_(torch.tensor([[s_d, s_c, s_c]])); loss.q.copy_(torch.as_tensor(lv[None] * (1 - np.array([s_d, s_c, s_c]) ** 2)))
    hip_ops._lib.load().eks_profile_enable(1)
    for _ in range(20): loss.evaluate()
    torch.cuda.synchronize()
    import ctypes
    names = ctypes.create_string_buffer(1 << 16); ms = (ctypes.c_float * 4096)()
    n = hip_ops._lib.load().eks_profile_drain(names, len(names), ms, 4096)
    hip_ops._lib.load().eks_profile_enable(0)
    print(f'   ar1_nll kernels (value + 2 tangents): {np.mean(list(ms)[:n])*1e3:.1f} us', flush=True)
