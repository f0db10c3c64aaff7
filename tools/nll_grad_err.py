import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from oracle import eks_oracle as orc
import test_gpu_kernels as tg
from eks_amd import hip_ops
T, K = 40000, 4
arrs, y_tk, var_tk = tg._singlecam_problem(T, K, seed=5 + T, unit=True)
flags = hip_ops.model_flags(arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'])
rconst = hip_ops.const_r(tg._dev(var_tk), 1e-4)
Rc = rconst.cpu().numpy()
print('|y| max', np.abs(y_tk).max())
for sv in (np.exp(-8.0), 0.01, 1.0, 100.0):
    s = np.full((K, 1), sv)
    nll, g = hip_ops.nll(tg._dev(y_tk), rconst, *tg._params_dev(arrs), tg._dev(s), per_keypoint=True, want_grad=True, flags=flags)
    ref, gr = orc.filter_nll(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'], np.full(K, sv), Rc, want_grad=True)
    print(f's={sv:.3g}: nll rel err {np.abs(nll.cpu().numpy()[:,0]-ref).max()/np.abs(ref).max():.2e}, grad rel err {(np.abs(g.cpu().numpy()[:,0]-gr)/np.abs(gr)).max():.2e} (|g| {np.abs(gr).min():.3g}..{np.abs(gr).max():.3g})', flush=True)
