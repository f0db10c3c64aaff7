"""Randomised sweep of the pupil driver against the oracle's pupil pipeline (fixed parameters and a
capped optimisation).  Usage: python tools/fuzz_pupil.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import eks_oracle as orc
from eks_amd import MarkerArray
from eks_amd.core import ensemble
from eks_amd.ibl_pupil_smoother import PUPIL_BODYPARTS, ensemble_kalman_smoother_ibl_pupil

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(fixed=0.0, adam_s=0.0, adam_out=0.0)
for case in range(n_cases):
    T, M = int(rng.choice([30, 150, 700, 2500])), int(rng.integers(2, 7))
    # a pupil: centre random walk, diameter AR(1), four points + member noise
    cen = np.cumsum(rng.normal(size=(T, 2)) * 0.3, axis=0) + np.array([80.0, 60.0])
    dia = 12 + np.cumsum(rng.normal(size=T) * 0.05)
    pts = np.stack([cen + np.stack([0 * dia, -dia / 2], 1), cen + np.stack([0 * dia, dia / 2], 1),
                    cen + np.stack([dia / 2, 0 * dia], 1), cen + np.stack([-dia / 2, 0 * dia], 1)], axis=1)   # (T,4,2)
    mk = np.empty((M, 1, T, 4, 3))
    mk[:, 0, :, :, :2] = pts[None] + rng.normal(size=(M, T, 4, 2)) * rng.uniform(0.2, 1.5)
    mk[:, 0, :, :, 2] = rng.uniform(0.5, 1.0, size=(M, T, 4))
    mk = mk.astype(np.float32)
    ma = MarkerArray(mk.astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    avg, varm = str(rng.choice(['median', 'mean'])), str(rng.choice(['confidence_weighted_var', 'var']))
    ens = ensemble(ma, avg_mode=avg, var_mode=varm).array
    arrs = orc.pupil_arrays(mk, avg, varm, ens=ens)
    a = (arrs['ys'].astype(np.float32).astype(np.float64), arrs['m0'], arrs['S0'], arrs['C'],
         arrs['ensemble_vars'].astype(np.float32).astype(np.float64), arrs['latent_vars'])
    sp = [float(rng.uniform(0.3, 0.999)), float(rng.uniform(0.3, 0.999))]
    df, s = ensemble_kalman_smoother_ibl_pupil(ma, list(PUPIL_BODYPARTS), smooth_params=sp, avg_mode=avg, var_mode=varm)
    so, ms, Vs, _ = orc.run_pupil_kalman_smoother(*a, smooth_params=sp)
    ref = orc.pupil_outputs(arrs, ms, Vs)
    e1 = float((np.abs(df.values - ref) / np.maximum(np.abs(ref).max(axis=0), 1e-12)).max())
    cap = int(rng.choice([5, 40]))
    frames = None if rng.random() < 0.5 else [(int(T * 0.1), int(T * 0.8))]
    df2, s2 = ensemble_kalman_smoother_ibl_pupil(ma, list(PUPIL_BODYPARTS), smooth_params=None, s_frames=frames,
                                                 avg_mode=avg, var_mode=varm, safety_cap=cap)
    so2, ms2, Vs2, info = orc.run_pupil_kalman_smoother(*a, s_frames=frames, safety_cap=cap)
    e2 = float(np.abs(np.array(s2) - np.array(so2)).max())
    ref2 = orc.pupil_outputs(arrs, ms2, Vs2)
    e3 = float((np.abs(df2.values - ref2) / np.maximum(np.abs(ref2).max(axis=0), 1e-12)).max())
    worst['fixed'] = max(worst['fixed'], e1); worst['adam_s'] = max(worst['adam_s'], e2); worst['adam_out'] = max(worst['adam_out'], e3)
    print(f'case {case}: T={T} M={M} {avg}/{varm} s={np.round(sp, 3)}: fixed {e1:.1e}; cap {cap} frames {frames}: |ds| {e2:.1e} out {e3:.1e}'
          + ('   <-- above 1e-5' if max(e1, e3) > 1e-5 else ''), flush=True)
print('worst', worst)
