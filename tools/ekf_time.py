"""Timings of the calibrated (nonlinear) multicam path on the GPU: cold and warm eks_ekf_smooth,
sweeps to the fixed point for several smoothing parameters, and run_kalman_smoother(h_fn=...) with
the optimiser (Adam by central differences over three chains per keypoint)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from eks_amd import calibration as cal
from eks_amd import hip_ops, synth
from eks_amd.core import run_kalman_smoother


def main():
    T, K, V = 50_000, 16, 4
    prob = synth.calibrated_multicam(T, K, V, seed=4)
    dev = torch.device('cuda')
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    y, var = t(prob['y_tko'], torch.float32), t(prob['var_tko'], torch.float32)
    m0, S0, A, Q = t(prob['m0s']), t(prob['S0s']), t(prob['As']), t(prob['Qs'])
    cams = t(prob['cams_packed'])
    cold = m0[:, None, :].expand(K, T, 3).contiguous()

    def timed(fn, n=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, out

    print(f'T={T} K={K} V={V} (O={2 * V})')
    for sval in (1e-3, 0.1, 10.0, 1e3):
        s = torch.full((K,), sval, dtype=torch.float64, device=dev)
        x = cold.clone()
        dt, out = timed(lambda: (x.copy_(cold), hip_ops.ekf_smooth(y, var, None, m0, S0, A, Q, s, cams, x,
                                                                  max_sweeps=16, tol=1e-10))[1])
        info = out[3].cpu().numpy()
        dtw, outw = timed(lambda: hip_ops.ekf_smooth(y, var, None, m0, S0, A, Q, s, cams, x, max_sweeps=16,
                                                     tol=1e-10))
        print(f's={sval:g}: cold start {dt * 1e3:.2f} ms ({info[0]:.0f} filter sweeps, last change '
              f'{info[1]:.1e}); warm {dtw * 1e3:.2f} ms ({outw[3][0].item():.0f} sweep)')
    h = cal.PinholeProjection(prob['cams_packed'])
    ys = np.swapaxes(prob['y_tko'], 0, 1)
    for mode in ('adam', 'grid'):
        run_kalman_smoother(ys[:2], prob['m0s'][:2], prob['S0s'][:2], prob['As'][:2], None, prob['Qs'][:2],
                            prob['var_tko'][:, :2], h_fn=h, s_mode=mode, n_grid=16)
        t0 = time.perf_counter()
        s, ms, Vs = run_kalman_smoother(ys, prob['m0s'], prob['S0s'], prob['As'], None, prob['Qs'],
                                        prob['var_tko'], h_fn=h, s_mode=mode, n_grid=16, return_device=True)
        torch.cuda.synchronize()
        print(f'run_kalman_smoother(h_fn, s_mode={mode!r}): {time.perf_counter() - t0:.3f} s  '
              f's in [{s.min():.3g}, {s.max():.3g}]')


if __name__ == '__main__':
    main()
