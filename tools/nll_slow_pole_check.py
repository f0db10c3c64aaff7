"""NLL grid against the C oracle when the slow candidates' poles sit very close to 1 (large ensemble variances
against the process noise): relative error per candidate, worst over keypoints.  usage: [T] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from oracle import c_oracle
import test_gpu_kernels as tg
from eks_amd import hip_ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for unit in (True, False):
    for scale in (1.0, 30.0, 1000.0):
        arrs, y_tk, var_tk = tg._singlecam_problem(T, K, seed=5, unit=unit)
        var_tk = (var_tk * scale).astype(np.float32)
        arrs['ensemble_vars'] = var_tk.astype(np.float64)
        arrs['S0s'] = np.eye(2) * np.maximum(np.var(arrs['ys'], axis=1), 1e-3)[:, :, None]
        flags = hip_ops.model_flags(arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'])
        rc = hip_ops.const_r(tg._dev(var_tk), 1e-4)
        cand = np.exp(np.linspace(-8, 8, 64))
        nll = hip_ops.nll(tg._dev(y_tk), rc, *tg._params_dev(arrs), tg._dev(cand), flags=flags).cpu().numpy()
        ref = c_oracle.nll_grid(arrs['ys'], rc.cpu().numpy(), arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'], cand)
        rel = np.abs(nll - ref) / np.abs(ref)
        am_g, am_o = nll.argmin(1), ref.argmin(1)
        print(f'unit={unit} var x{scale:g} (median R {np.median(rc.cpu().numpy()):.3g}): worst rel err per candidate '
              f'[0..5] {np.array2string(rel.max(0)[:6], precision=1)}, overall {rel.max():.2e}; argmin equal: '
              f'{int((am_g == am_o).sum())}/{K} (oracle argmin range {am_o.min()}..{am_o.max()})', flush=True)
