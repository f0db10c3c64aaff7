"""Smoother parity at long T / large |y| (scalar-chain path) against the C oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from oracle import c_oracle
import test_gpu_kernels as tg
from eks_amd import hip_ops
T, K = 40000, 32
arrs, y_tk, var_tk = tg._singlecam_problem(T, K, seed=5 + T, unit=True)
flags = hip_ops.model_flags(arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'])
Rd = np.maximum(np.swapaxes(arrs['ensemble_vars'], 0, 1), 1e-12)
for sv in (np.exp(-8.0), 1.0, np.exp(8.0)):
    s = np.full(K, sv)
    ms, Vs = hip_ops.smooth(tg._dev(y_tk), tg._dev(var_tk), *tg._params_dev(arrs), tg._dev(s), flags=flags, vs_diag=True)
    ms = np.transpose(ms.cpu().numpy().astype(np.float64), (1, 0, 2)); Vs = np.transpose(Vs.cpu().numpy().astype(np.float64), (1, 0, 2))
    mo, Vo, _ = c_oracle.smooth(arrs['ys'], Rd, arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'], s)
    Vd = np.diagonal(Vo, axis1=2, axis2=3)
    print(f's={sv:.3g}: |ms| max {np.abs(mo).max():.0f}, abs err max {np.abs(ms-mo).max():.2e} px, rel-to-scale {(np.abs(ms-mo)/np.abs(mo).max(axis=(1,2),keepdims=True)).max():.2e}, '
          f'err/posterior-sd max {(np.abs(ms-mo)/np.sqrt(Vd)).max():.2e}, Vs rel {(np.abs(Vs-Vd)/Vd).max():.2e}', flush=True)
