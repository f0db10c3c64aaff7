"""Randomised sweep of run_kalman_smoother in the reference's Adam mode (blocks, s_frames, both
model families) against the oracle's restatement of the optimiser.  Usage: fuzz_adam.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from oracle import eks_oracle as orc
import test_gpu_kernels as tg
from eks_amd.core import run_kalman_smoother

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(dlogs=0.0, ms=0.0, iters=0)
for case in range(n_cases):
    dense = bool(rng.integers(0, 2))
    T = int(rng.choice([60, 250, 600])); K = int(rng.integers(1, 5))
    if not dense and rng.random() < 0.35:      # long enough for the search from cached lag sums (>= 1 024 frames)
        T = int(rng.choice([1100, 1500])); K = min(K, 2)
    if dense:
        D = int(rng.choice([2, 3])); O = int(rng.choice([D + 1, 4, 6]))
        arrs, y, var = tg._dense_problem(T, K, D, O, seed=int(rng.integers(1 << 30)))
    else:
        arrs, y, var = tg._singlecam_problem(T, K, seed=int(rng.integers(1 << 30)), unit=bool(rng.integers(0, 2)))
    blocks = []
    if K >= 2 and rng.random() < 0.5:
        perm = rng.permutation(K); cut = int(rng.integers(1, K))
        blocks = [sorted(perm[:cut].tolist()), sorted(perm[cut:].tolist())]
    s_frames = None
    if rng.random() < 0.5:
        a = int(rng.integers(0, T // 3)); b = int(rng.integers(2 * T // 3, T))
        s_frames = [(a, b)] if rng.random() < 0.5 else [(None, a + 10), (b - 5, None)]
    args = (arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'], arrs['ensemble_vars'])
    s, ms, Vs = run_kalman_smoother(*args, s_frames=s_frames, blocks=blocks)
    so, mo, Vo, info = orc.run_kalman_smoother(*args, s_frames=s_frames, blocks=blocks)
    dl = float(np.abs(np.log(s) - np.log(so)).max())
    # outputs strictly at the SAME s
    s2, ms2, Vs2 = run_kalman_smoother(*args, smooth_param=list(so))
    e = float((np.abs(ms2 - mo) / np.maximum(np.abs(mo).max(axis=(1, 2), keepdims=True), 1e-3)).max())
    worst['dlogs'] = max(worst['dlogs'], dl); worst['ms'] = max(worst['ms'], e)
    print(f'case {case}: dense={dense} T={T} K={K} blocks={blocks} s_frames={s_frames}: |dlog s| {dl:.2e} '
          f'(oracle iters {np.asarray(info["iters"]).max()}), ms at oracle s {e:.1e}' + ('   <-- check' if dl > 0.05 or e > 1e-5 else ''), flush=True)
print('worst', worst)
