"""Randomised parity sweep, part 2: the gradient (dual-number) path of eks_nll on scalar chains and
the general (D, O) kernels (smoother, loss, d/dlog s) against the NumPy oracle.
Usage: python tools/fuzz_parity2.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from oracle import eks_oracle as orc
import test_gpu_kernels as tg
from eks_amd import hip_ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(g_nll=0.0, g_grad=0.0, d_ms=0.0, d_Vs=0.0, d_nll=0.0, d_grad=0.0)
for case in range(n_cases):
    # ---- scalar chains, gradient mode
    T = int(rng.choice([1, 2, 5, 9, 100, 511, 512, 513, 1300, 3000]))
    K = int(rng.choice([1, 2, 3, 5]))
    unit = bool(rng.integers(0, 2))
    arrs, y_tk, var_tk = tg._singlecam_problem(max(T, 2), K, seed=int(rng.integers(1 << 30)), unit=unit)
    y_tk, var_tk = y_tk[:T].copy(), (var_tk[:T] * float(np.exp(rng.uniform(-3, 3)))).astype(np.float32)
    arrs['ys'] = np.transpose(y_tk, (1, 0, 2)).astype(np.float64); arrs['ensemble_vars'] = var_tk.astype(np.float64)
    arrs['S0s'] = np.eye(2) * (np.maximum(np.var(arrs['ys'], axis=1), 1e-3) if T > 2 else np.full((K, 2), 3.0))[:, :, None]
    flags = hip_ops.model_flags(arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'])
    rc = hip_ops.const_r(tg._dev(var_tk), 1e-4)
    s = np.exp(rng.uniform(-8, 8, (K, 1)))
    nll, g = hip_ops.nll(tg._dev(y_tk), rc, *tg._params_dev(arrs), tg._dev(s), per_keypoint=True, want_grad=True, flags=flags)
    ref, gr = orc.filter_nll(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'], s[:, 0], rc.cpu().numpy(), want_grad=True)
    e1 = float((np.abs(nll.cpu().numpy()[:, 0] - ref) / np.maximum(np.abs(ref), 10.0)).max())
    e2 = float((np.abs(g.cpu().numpy()[:, 0] - gr) / np.maximum(np.abs(gr), 1e-2 * np.maximum(np.abs(ref), 10.0))).max())
    worst['g_nll'] = max(worst['g_nll'], e1); worst['g_grad'] = max(worst['g_grad'], e2)
    # ---- general (D, O) path
    Td = int(rng.choice([1, 2, 9, 33, 300, 1500])); Kd = int(rng.choice([1, 2, 4])); D = int(rng.choice([1, 2, 3, 4, 6])); O = int(rng.choice([D, D + 1, 2 * D, 8]))
    darr, y, var = tg._dense_problem(Td, Kd, D, O, seed=int(rng.integers(1 << 30)))
    if rng.random() < 0.5:
        darr['As'] = darr['As'] * 0.97 + 0.02 * rng.standard_normal((Kd, D, D))
    if rng.random() < 0.3:
        var[rng.random(var.shape) < 0.1] = 0.0
        darr['ensemble_vars'] = var.astype(np.float64)
    sd = np.exp(rng.uniform(-4, 4, Kd))
    ms, Vs = hip_ops.smooth(tg._dev(y), tg._dev(var), *tg._params_dev(darr), tg._dev(sd), flags=0)
    ms = np.transpose(ms.cpu().numpy().astype(np.float64), (1, 0, 2)); Vs = np.transpose(Vs.cpu().numpy().astype(np.float64), (1, 0, 2, 3))
    Rd = np.maximum(np.swapaxes(darr['ensemble_vars'], 0, 1), 1e-12)
    mo, Vo, _ = orc.kalman_smoother(darr['ys'], darr['m0s'], darr['S0s'], darr['As'], darr['Cs'], darr['Qs'], sd, Rd)
    e3 = float((np.abs(ms - mo) / np.maximum(np.abs(mo).max(axis=(1, 2), keepdims=True), 1e-3)).max())
    e4 = float((np.abs(Vs - Vo) / np.abs(Vo).max(axis=1, keepdims=True)).max())
    rcd = hip_ops.const_r(tg._dev(var), 1e-4)
    nd, gd = hip_ops.nll(tg._dev(y), rcd, *tg._params_dev(darr), tg._dev(sd[:, None].copy()), per_keypoint=True, want_grad=True, flags=0)
    refd, grd = orc.filter_nll(darr['ys'], darr['m0s'], darr['S0s'], darr['As'], darr['Cs'], darr['Qs'], sd, rcd.cpu().numpy(), want_grad=True)
    e5 = float((np.abs(nd.cpu().numpy()[:, 0] - refd) / np.maximum(np.abs(refd), 10.0)).max())
    e6 = float((np.abs(gd.cpu().numpy()[:, 0] - grd) / np.maximum(np.abs(grd), 1e-2 * np.maximum(np.abs(refd), 10.0))).max())
    clipped = bool((var == 0).any())
    if not clipped:      # at the clip the covariance-form oracle is not a valid reference
        worst['d_ms'] = max(worst['d_ms'], e3); worst['d_Vs'] = max(worst['d_Vs'], e4)
    worst['d_nll'] = max(worst['d_nll'], e5); worst['d_grad'] = max(worst['d_grad'], e6)
    bad = max(e1, e2, e5, e6, 0.0 if clipped else max(e3, e4)) > 1e-5
    print(f'case {case}: diag T={T} K={K} unit={unit}: nll {e1:.1e} grad {e2:.1e} | dense T={Td} K={Kd} D={D} O={O} clip={clipped}: '
          f'ms {e3:.1e} Vs {e4:.1e} nll {e5:.1e} grad {e6:.1e}' + ('   <-- above 1e-5' if bad else ''), flush=True)
print('worst', worst)
