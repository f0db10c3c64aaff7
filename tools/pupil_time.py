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
    for pos, what in ((False, 'dual numbers'), (True, 'smoothing-distribution derivatives')):
        loss = hip_ops.Ar1Loss(P.y, P.var, P.m0, P.S0, P.C, n_tan=2, positive_noise=pos)
        loss.a.copy_(torch.tensor([[s_d, s_c, s_c]])); loss.q.copy_(torch.as_tensor(lv[None] * (1 - np.array([s_d, s_c, s_c]) ** 2)))
        for _ in range(5): loss.evaluate()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): loss.evaluate()
        torch.cuda.synchronize()
        print(f'   eks_ar1_nll (value + 2 tangents), {what}: {(time.perf_counter()-t0)/50*1e6:.1f} us', flush=True)
