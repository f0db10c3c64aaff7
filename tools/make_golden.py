#!/usr/bin/env python
"""Generate tests/golden/*.npz (run in the BUILD container, where /root/reference is mounted).

Inputs are the reference's own test data files (data/ibl-pupil, data/mirror-mouse: the inputs of
its integration tests tests/integration/test_singlecam.py:4-20 and test_mirrored_multicam.py:4-30),
stored as float32 marker arrays.  Expected outputs come from THIS repo's float64 oracle
(oracle/eks_oracle.py): the reference itself cannot run here (jax / dynamax absent) and its golden
CSVs are remote, so these vectors are self-generated ("parity unpinned" w.r.t. upstream numbers).
Outputs are stored decimated (every 4th frame + both ends) plus full-array column sums.
"""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import eks_oracle as orc  # noqa: E402

REF = '/root/reference/data'
OUT = os.path.join(ROOT, 'tests', 'golden')


def read_dir(d):
    dfs = []
    for f in sorted(os.listdir(d)):
        if f.endswith('.csv'):
            dfs.append(pd.read_csv(os.path.join(d, f), header=[0, 1, 2], index_col=0))
    return dfs


def keep_idx(T):
    return np.unique(np.concatenate([np.arange(0, 16), np.arange(0, T, 4), np.arange(T - 16, T)]))


def pack(full):
    """decimated rows + column sums / abs-sums of the full array"""
    idx = keep_idx(full.shape[0])
    return dict(rows=full[idx].astype(np.float32), colsum=full.sum(axis=0),
                colabs=np.abs(full).sum(axis=0))


def pupil():
    """ibl-pupil through the pupil smoother (the reference's tests/integration/test_ibl_pupil.py:
    defaults and smooth_params=[0.99, 0.99]).  The marker array is the one already stored in
    ibl_pupil_singlecam.npz, re-ordered to (top, bottom, right, left)."""
    g = np.load(os.path.join(OUT, 'ibl_pupil_singlecam.npz'))
    order = [list(g['keypoints']).index(k) for k in orc.PUPIL_KEYPOINTS]
    mk = g['markers'][:, :, :, order]
    arrs = orc.pupil_arrays(mk)
    out = dict(order=np.array(order), keep_idx=keep_idx(mk.shape[2]), m0=arrs['m0'], S0=arrs['S0'],
               latent_vars=arrs['latent_vars'], means=np.array([arrs['mean_x'], arrs['mean_y']]))
    args = (arrs['ys'], arrs['m0'], arrs['S0'], arrs['C'], arrs['ensemble_vars'], arrs['latent_vars'])
    s, ms, Vs, info = orc.run_pupil_kalman_smoother(*args, smooth_params=[0.99, 0.99])
    out['fixed_s'] = np.array(s)
    out['fixed_nll'] = info['nll']
    for k, v in pack(orc.pupil_outputs(arrs, ms, Vs)).items():
        out[f'fixed_{k}'] = v
    for k, v in pack(np.concatenate([ms, Vs.reshape(len(ms), 9)], axis=1)).items():
        out[f'fixed_state_{k}'] = v
    s, ms, Vs, info = orc.run_pupil_kalman_smoother(*args)
    out['adam_s'] = np.array(s)
    out['adam_iters'] = info['iters']
    out['adam_last_loss'] = info['last_loss']
    for k, v in pack(orc.pupil_outputs(arrs, ms, Vs)).items():
        out[f'adam_{k}'] = v
    # the loss and its gradient at a few points of the (u_diam, u_com) plane
    us = np.array([[4.59511985, 3.8918203], [2.0, 2.0], [6.0, 1.0], [0.0, 5.0], [-2.0, 7.0]])
    lg = [orc.pupil_nll_and_grad(u, *args, use_c=False) for u in us]
    out['probe_u'] = us
    out['probe_nll'] = np.array([v[0] for v in lg])
    out['probe_grad'] = np.array([v[1] for v in lg])
    np.savez_compressed(os.path.join(OUT, 'ibl_pupil_pupil.npz'), **out)
    print('ibl-pupil (pupil smoother): s_adam', s, 'iters', info['iters'], 'loss', info['last_loss'])


def main():
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ['pupil']:
        return pupil()
    # ---------------- ibl-pupil: singlecam, 5 models x 2000 frames x 4 keypoints
    dfs = read_dir(os.path.join(REF, 'ibl-pupil'))
    kps = dfs[0].columns[dfs[0].columns.get_level_values('coords') == 'x'].get_level_values('bodyparts').tolist()
    scorer = dfs[0].columns[0][0]
    mk = np.stack([np.stack([df[(scorer, k, c)].to_numpy() for k in kps for c in ('x', 'y', 'likelihood')],
                            axis=1).reshape(len(df), len(kps), 3) for df in dfs])[:, None]
    mk = mk.astype(np.float32)                                     # (5,1,2000,4,3)
    arrs = orc.singlecam_arrays(mk)
    out = dict(markers=mk, keypoints=np.array(kps), scorer=np.array(scorer),
               index_first=np.array(dfs[0].index[:3]), csv_header=np.array(
                   open(os.path.join(REF, 'ibl-pupil', sorted(os.listdir(os.path.join(REF, 'ibl-pupil')))[0])
                        ).read().split('\n')[:4]))
    T = mk.shape[2]
    out['keep_idx'] = keep_idx(T)
    # fixed smoothing parameter (upstream integration test uses smooth_param=[10.0])
    s, ms, Vs, _ = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'],
                                           arrs['Qs'], arrs['ensemble_vars'], smooth_param=[10.0])
    for k, v in pack(orc.singlecam_outputs(arrs, s, ms, Vs)).items():
        out[f's10_{k}'] = v
    # optimised s: Adam (reference behaviour) and 64-candidate grid
    s_a, ms, Vs, info = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'],
                                                arrs['Cs'], arrs['Qs'], arrs['ensemble_vars'])
    out['adam_s'] = s_a
    out['adam_iters'] = info['iters']
    for k, v in pack(orc.singlecam_outputs(arrs, s_a, ms, Vs)).items():
        out[f'adam_{k}'] = v
    s_g, ms, Vs, info = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'],
                                                arrs['Cs'], arrs['Qs'], arrs['ensemble_vars'], s_mode='grid')
    out['grid_s'] = s_g
    out['grid_idx'] = info['argmin']
    out['grid_nll'] = info['nll']
    for k, v in pack(orc.singlecam_outputs(arrs, s_g, ms, Vs)).items():
        out[f'grid_{k}'] = v
    out['guesses'] = np.array([orc.compute_initial_guess(arrs['ensemble_vars'][:, k]) for k in range(len(kps))])
    np.savez_compressed(os.path.join(OUT, 'ibl_pupil_singlecam.npz'), **out)
    print('ibl-pupil: s_adam', s_a, 'iters', out['adam_iters'], 's_grid', s_g)

    # ---------------- mirror-mouse: mirrored multicam, 2 views, all 4 paws, n_latent 3
    dfs = read_dir(os.path.join(REF, 'mirror-mouse'))
    scorer = dfs[0].columns[0][0]
    cams, paws = ['top', 'bot'], ['paw1LH', 'paw2LF', 'paw3RF', 'paw4RH']
    mk = np.empty((len(dfs), 2, len(dfs[0]), len(paws), 3), np.float32)
    for m, df in enumerate(dfs):
        for v, cam in enumerate(cams):
            for k, paw in enumerate(paws):
                for f, c in enumerate(('x', 'y', 'likelihood')):
                    mk[m, v, :, k, f] = df[(scorer, f'{paw}_{cam}', c)].to_numpy()
    out = dict(markers=mk, keypoints=np.array(paws), cameras=np.array(cams), scorer=np.array(scorer),
               csv_header=np.array(open(os.path.join(REF, 'mirror-mouse', sorted(os.listdir(
                   os.path.join(REF, 'mirror-mouse')))[0])).read().split('\n')[:4]))
    T = mk.shape[2]
    out['keep_idx'] = keep_idx(T)
    from sklearn.decomposition import PCA

    def sk_pca(X, n):
        p = PCA(n_components=n).fit(X)
        return p.components_, p.mean_

    arrs = orc.multicam_arrays(mk, quantile_keep_pca=95.0, n_latent=3, pca_fit=sk_pca)
    out['valid_mask'] = arrs['mask']
    out['good_idx'] = arrs['good_idx']
    for nm in ('Cs', 'Qs', 'S0s'):
        out[nm] = arrs[nm]
    s, ms, Vs, _ = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'],
                                           arrs['Qs'], arrs['ensemble_vars'], smooth_param=[10.0])
    cams_out, lat = orc.multicam_outputs(arrs, ms, Vs)
    for c, co in enumerate(cams_out):
        for k, v in pack(co).items():
            out[f's10_cam{c}_{k}'] = v
    for k, v in pack(lat).items():
        out[f's10_latent_{k}'] = v
    s_a, ms, Vs, info = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'],
                                                arrs['Cs'], arrs['Qs'], arrs['ensemble_vars'])
    out['adam_s'] = s_a
    out['adam_iters'] = info['iters']
    cams_out, lat = orc.multicam_outputs(arrs, ms, Vs)
    for c, co in enumerate(cams_out):
        for k, v in pack(co).items():
            out[f'adam_cam{c}_{k}'] = v
    # the configuration of the reference's integration test (tests/integration/
    # test_mirrored_multicam.py:19-30): two paws, quantile_keep_pca=95, inflate_vars=True, s=10
    arrs = orc.multicam_arrays(mk[:, :, :, :2], quantile_keep_pca=95.0, n_latent=3, pca_fit=sk_pca,
                               inflate_vars=True)
    out['infl_vars'] = np.swapaxes(arrs['ensemble_vars'], 0, 1).astype(np.float32)      # (K,T,2V)
    s, ms, Vs, _ = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'],
                                           arrs['Qs'], arrs['ensemble_vars'], smooth_param=[10.0])
    cams_out, lat = orc.multicam_outputs(arrs, ms, Vs)
    for c, co in enumerate(cams_out):
        for k, v in pack(co).items():
            out[f'infl_s10_cam{c}_{k}'] = v
    np.savez_compressed(os.path.join(OUT, 'mirror_mouse_multicam.npz'), **out)
    print('mirror-mouse: s_adam', s_a, 'iters', out['adam_iters'])
    pupil()


if __name__ == '__main__':
    main()
