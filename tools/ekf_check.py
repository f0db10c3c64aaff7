"""GPU check of eks_ekf_smooth against the sequential extended filter of oracle/ekf_oracle.py."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from eks_amd import hip_ops, synth
from oracle import ekf_oracle as ek


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    V = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    prob = synth.calibrated_multicam(T, K, V, seed=7)
    dev = torch.device('cuda')
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    y, var = t(prob['y_tko'], torch.float32), t(prob['var_tko'], torch.float32)
    m0, S0, A, Q = t(prob['m0s']), t(prob['S0s']), t(prob['As']), t(prob['Qs'])
    s = t(prob['s'])
    cams = t(prob['cams_packed'])
    xlin = m0[:, None, :].expand(K, T, 3).contiguous()
    for rep in range(3):
        xl = xlin.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ms, Vs, nll, info = hip_ops.ekf_smooth(y, var, None, m0, S0, A, Q, s, cams, xl, max_sweeps=16, tol=1e-10)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f'T={T} K={K} V={V}: {dt*1e3:.2f} ms, sweeps {info[0].item():.0f}, resid {info[1].item():.2e}')
    h = ek.combine_projections([ek.make_projection_fn(c['rot'], c['tvec'], c['K'], c['dist']) for c in prob['cams']])
    ms, Vs, nll = ms.cpu().numpy(), Vs.cpu().numpy(), nll.cpu().numpy()
    yq = prob['y_tko'].astype(np.float32).astype(np.float64)
    vq = np.maximum(prob['var_tko'].astype(np.float32).astype(np.float64), 1e-12)
    for k in range(min(K, 3)):
        mo, Vo, ll = ek.eks_smoother(yq[:, k], vq[:, k], prob['m0s'][k], prob['S0s'][k], prob['As'][k],
                                     prob['Qs'][k], prob['s'][k], h)
        scale = np.abs(mo).max()
        print(k, 'ms rel', np.abs(ms[:, k] - mo).max() / scale, 'Vs rel', np.abs(Vs[:, k] - Vo).max() / np.abs(Vo).max(),
              'nll rel', abs(nll[k] + ll) / abs(ll))


if __name__ == '__main__':
    main()
