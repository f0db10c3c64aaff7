"""Randomised sweep of the reference-shaped drivers (singlecam, multicam linear with and without
variance inflation) against the oracle's restatement of the same pipelines.
Usage: python tools/fuzz_drivers.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sklearn.decomposition import PCA
from oracle import eks_oracle as orc
from eks_amd import MarkerArray, synth
from eks_amd.core import ensemble
from eks_amd.singlecam_smoother import ensemble_kalman_smoother_singlecam
from eks_amd.multicam_smoother import ensemble_kalman_smoother_multicam


def sk(X, n):
    p = PCA(n_components=n).fit(X)
    return p.components_, p.mean_


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = dict(single=0.0, multi=0.0)
for case in range(n_cases):
    # ---- singlecam
    T, K, M = int(rng.choice([40, 333, 1200])), int(rng.integers(1, 6)), int(rng.integers(1, 7))
    avg, varm = str(rng.choice(['median', 'mean'])), str(rng.choice(['confidence_weighted_var', 'var']))
    sp = float(np.exp(rng.uniform(-3, 5)))
    mk = synth.singlecam_markers(T, K, M=M, seed=int(rng.integers(1 << 30)))
    ma = MarkerArray(mk.astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    df, s = ensemble_kalman_smoother_singlecam(ma, [f'k{i}' for i in range(K)], smooth_param=sp, avg_mode=avg, var_mode=varm)
    arrs = orc.singlecam_arrays(mk, avg, varm, ens=ensemble(ma, avg_mode=avg, var_mode=varm).array)
    for key in ('ys', 'ensemble_vars'):
        arrs[key] = arrs[key].astype(np.float32).astype(np.float64)
    so, ms, Vs, _ = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'],
                                            arrs['ensemble_vars'], smooth_param=sp)
    ref = orc.singlecam_outputs(arrs, so, ms, Vs)
    e1 = float((np.abs(df.values - ref) / np.maximum(np.abs(ref).max(axis=0), 1e-12)).max())
    worst['single'] = max(worst['single'], e1)
    # ---- multicam (linear)
    T2, K2, V, M2 = int(rng.choice([120, 500])), int(rng.integers(1, 4)), int(rng.integers(2, 5)), int(rng.integers(2, 6))
    nl = 3 if V == 2 else int(rng.choice([3, 4]))
    qk = float(rng.choice([50.0, 75.0, 95.0]))
    infl = bool(rng.integers(0, 2))
    mk2 = synth.multicam_markers(T2, K2, V=V, M=M2, seed=int(rng.integers(1 << 30)))
    ma2 = MarkerArray(mk2.astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    dfs, s2, df3 = ensemble_kalman_smoother_multicam(ma2, [f'p{i}' for i in range(K2)], [f'c{i}' for i in range(V)],
                                                     smooth_param=sp, n_latent=nl, quantile_keep_pca=qk, inflate_vars=infl)
    a2 = orc.multicam_arrays(mk2, quantile_keep_pca=qk, n_latent=nl, pca_fit=sk, ens=ensemble(ma2).array, inflate_vars=infl)
    for key in ('ys', 'ensemble_vars'):
        a2[key] = a2[key].astype(np.float32).astype(np.float64)
    s_o, ms2, Vs2, _ = orc.run_kalman_smoother(a2['ys'], a2['m0s'], a2['S0s'], a2['As'], a2['Cs'], a2['Qs'], a2['ensemble_vars'],
                                               smooth_param=sp)
    cams, _ = orc.multicam_outputs(a2, ms2, Vs2)
    e2 = max(float((np.abs(dfs[c].values - cams[c]) / np.maximum(np.abs(cams[c]).max(axis=0), 1e-12)).max()) for c in range(V))
    worst['multi'] = max(worst['multi'], e2)
    print(f'case {case}: single T={T} K={K} M={M} {avg}/{varm} s={sp:.3g}: {e1:.1e} | multi T={T2} K={K2} V={V} M={M2} '
          f'n_latent={nl} q={qk} inflate={infl}: {e2:.1e}' + ('   <-- above 1e-5' if max(e1, e2) > 1e-5 else ''), flush=True)
print('worst', worst)
