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


def fly():
    """data/fly through the CALIBRATED multicam path (reference eks/multicam_smoother.py:367-407 with
    its own data/fly/calibration.toml: 3 cameras x 3 ensemble members x 502 frames x 12 keypoints).
    Expected outputs: oracle/ekf_oracle.py (sequential extended filter + RTS, complex-step
    Jacobians) on the triangulation-initialised model, smooth_param = 10 for all keypoints and the
    Adam search for the first two."""
    import ast
    from oracle import ekf_oracle as ek
    d = os.path.join(REF, 'fly')
    text = open(os.path.join(d, 'calibration.toml')).read()
    cams, cur = [], None
    for line in text.split('\n'):
        line = line.strip()
        if line.startswith('[cam_'):
            cur = {}
            cams.append(cur)
        elif line.startswith('['):
            cur = None
        elif cur is not None and '=' in line:
            k, v = line.split('=', 1)
            cur[k.strip()] = ast.literal_eval(v.strip())
    names = [c['name'] for c in cams]
    files = sorted(f for f in os.listdir(d) if f.endswith('.csv'))
    per_cam = [[pd.read_csv(os.path.join(d, f), header=[0, 1, 2], index_col=0) for f in files if n in f]
               for n in names]
    df0 = per_cam[0][0]
    scorer = df0.columns[0][0]
    kps = df0.columns[df0.columns.get_level_values('coords') == 'x'].get_level_values('bodyparts').tolist()
    M, V, T, K = len(per_cam[0]), len(names), len(df0), len(kps)
    mk = np.empty((M, V, T, K, 3), np.float32)
    for v in range(V):
        for m in range(M):
            for k, kp in enumerate(kps):
                for f, c in enumerate(('x', 'y', 'likelihood')):
                    mk[m, v, :, k, f] = per_cam[v][m][(scorer, kp, c)].to_numpy()
    ocams = [dict(rot=np.array(c['rotation'], float), tvec=np.array(c['translation'], float),
                  K=np.array(c['matrix'], float), dist=np.array(c['distortions'], float)) for c in cams]
    heads = [ek.make_projection_fn(c['rot'], c['tvec'], c['K'], c['dist']) for c in ocams]
    h = ek.combine_projections(heads)
    st = orc.ensemble(mk)[0]                                                     # (V,T,K,5)
    tri = np.stack([np.stack([ek.triangulate_dlt(ocams, mk[m, :, :, k, :2].astype(np.float64))
                              for k in range(K)]) for m in range(M)])            # (M,K,T,3)
    ys3 = tri.mean(axis=0)
    m0s, S0s, As, Qs, _ = ek.initialize_kalman_filter_geometric(ys3)
    ys = np.transpose(st[..., 0:2], (2, 1, 0, 3)).reshape(K, T, 2 * V)
    evs = np.transpose(st[..., 2:4], (2, 1, 0, 3)).reshape(K, T, 2 * V)
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    out = dict(markers=mk, keypoints=np.array(kps), cameras=np.array(names), toml=np.array(text),
               keep_idx=keep_idx(T), ys3=ys3.astype(np.float32), m0s=m0s, S0s=S0s, Qs=Qs)

    def tables(ms, Vs, kk):
        res = []
        for c in range(V):
            tab = np.empty((T, len(kk), 9))
            for j, k in enumerate(kk):
                tab[:, j, 0:2] = heads[c](ms[j])
                tab[:, j, 7], tab[:, j, 8] = ek.project_3d_covariance_to_2d(ms[j], Vs[j], heads[c], evs[k])
                tab[:, j, 2] = st[c, :, k, 4]
                tab[:, j, 3:5] = st[c, :, k, 0:2]
                tab[:, j, 5:7] = st[c, :, k, 2:4]
            res.append(tab.reshape(T, len(kk) * 9))
        lat = np.concatenate([np.swapaxes(ms, 0, 1), np.swapaxes(np.diagonal(Vs, axis1=2, axis2=3), 0, 1)],
                             axis=2).reshape(T, len(kk) * 6)
        return res, lat

    s, ms, Vs, _ = ek.run_kalman_smoother_nonlinear(f32(ys), m0s, S0s, As, Qs, np.swapaxes(f32(evs), 0, 1), h,
                                                    smooth_param=10.0)
    cams_out, lat = tables(ms, Vs, list(range(K)))
    for c, co in enumerate(cams_out):
        for k, v in pack(co).items():
            out[f's10_cam{c}_{k}'] = v
    for k, v in pack(lat).items():
        out[f's10_latent_{k}'] = v
    kk = [0, 1]
    s_a, ms, Vs, info = ek.run_kalman_smoother_nonlinear(f32(ys[kk]), m0s[kk], S0s[kk], As[kk], Qs[kk],
                                                         np.swapaxes(f32(evs[kk]), 0, 1), h)
    out['adam_s'] = s_a
    out['adam_iters'] = info['iters']
    cams_out, lat = tables(ms, Vs, kk)
    for c, co in enumerate(cams_out):
        for k, v in pack(co).items():
            out[f'adam_cam{c}_{k}'] = v
    # the configuration of the reference's integration test (tests/integration/test_multicam.py:
    # 32-58): bodyparts L1A, L1B, quantile_keep_pca=95, inflate_vars=True, smooth_param=[10.0]
    from sklearn.decomposition import PCA

    def sk_pca(X, n):
        p = PCA(n_components=n).fit(X)
        return p.components_, p.mean_

    arrs = orc.multicam_arrays(mk[:, :, :, :2], quantile_keep_pca=95.0, n_latent=3, pca_fit=sk_pca,
                               inflate_vars=True)
    evs_i = np.swapaxes(arrs['ensemble_vars'], 0, 1)                            # (K,T,2V), inflated
    out['infl_vars'] = evs_i.astype(np.float32)
    _, ms, Vs, _ = ek.run_kalman_smoother_nonlinear(f32(ys[kk]), m0s[kk], S0s[kk], As[kk], Qs[kk],
                                                    np.swapaxes(f32(evs_i), 0, 1), h, smooth_param=[10.0])
    evs_all = evs.copy()
    evs[kk] = evs_i                       # tables() adds the (inflated) variances of camera 0
    cams_out, lat = tables(ms, Vs, kk)
    evs[:] = evs_all
    for c, co in enumerate(cams_out):
        for k, v in pack(co).items():
            out[f'infl_s10_cam{c}_{k}'] = v
    # ... and its default-arguments twin (test_multicam_defaults_nonlinear, :32-44): s optimised
    s_i, ms, Vs, info_i = ek.run_kalman_smoother_nonlinear(f32(ys[kk]), m0s[kk], S0s[kk], As[kk], Qs[kk],
                                                           np.swapaxes(f32(evs_i), 0, 1), h)
    out['infl_adam_s'] = s_i
    out['infl_adam_iters'] = info_i['iters']
    evs[kk] = evs_i
    cams_out, lat = tables(ms, Vs, kk)
    evs[:] = evs_all
    for c, co in enumerate(cams_out):
        for k, v in pack(co).items():
            out[f'infl_adam_cam{c}_{k}'] = v
    np.savez_compressed(os.path.join(OUT, 'fly_calibrated_multicam.npz'), **out)
    print('fly: T', T, 'K', K, 'V', V, 'M', M, 's_adam', s_a, 'iters', info['iters'])


def csv_samples():
    """The first 40 and last 5 lines of every recording under the reference's data/ (data files of its own tests) and
    what pandas reads from them: tests/test_csv_ingest.py holds the library's reader to pandas' values on the GPU box,
    where neither the reference nor its data exist."""
    import glob
    import pandas as pd
    import tempfile
    texts, names, values = [], [], []
    for f in sorted(glob.glob(os.path.join(REF, '**', '*.csv'), recursive=True)):
        lines = open(f).read().split('\n')
        body = [ln for ln in lines[3:] if ln]
        keep = lines[:3] + body[:40] + body[-5:]
        txt = '\n'.join(keep) + '\n'
        with tempfile.NamedTemporaryFile('w', suffix='.csv', delete=False) as t:
            t.write(txt)
        df = pd.read_csv(t.name, header=[0, 1, 2], index_col=0)
        os.unlink(t.name)
        texts.append(txt)
        names.append(os.path.relpath(f, REF))
        values.append(df.to_numpy(dtype=np.float64))
    enc = [t.encode('utf-8') for t in texts]                 # (bytes + offsets: nothing in the fixture needs unpickling)
    np.savez_compressed(os.path.join(OUT, 'csv_samples.npz'), names=np.array(names),
                        texts_bytes=np.frombuffer(b''.join(enc), dtype=np.uint8),
                        texts_offsets=np.cumsum([0] + [len(e) for e in enc]).astype(np.int64),
                        **{f'values_{i}': v for i, v in enumerate(values)})
    print('csv_samples:', len(names), 'files')


def main():
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ['csv']:
        return csv_samples()
    if sys.argv[1:] == ['fly']:
        return fly()
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


def reference_available():
    """True when the reference's own arithmetic can run here: jax + dynamax + optax importable."""
    import importlib.util
    return all(importlib.util.find_spec(m) is not None for m in ('jax', 'dynamax', 'optax'))


def from_reference():
    """PIN-WHEN-POSSIBLE HOOK.  When jax / dynamax / optax are importable in the build container,
    run the REFERENCE's own drivers (imported from /root/reference, never copied) on its own data
    with the configurations of its integration tests (tests/integration/test_singlecam.py:4-20,
    test_mirrored_multicam.py:4-30, test_multicam.py:4-30, test_ibl_pupil.py:4-22) and store the
    resulting DataFrames (decimated rows + column sums, like the oracle goldens) as
    tests/golden/ref_*.npz.  Only DATA travels.  tests/test_golden_oracle.py compares the oracle to
    these files when they exist - at that moment the oracle stops being "parity unpinned"."""
    if not reference_available():
        print('parity unpinned: jax / dynamax / optax are not importable here; no ref_*.npz written')
        return 1
    import tempfile
    sys.path.insert(0, '/root/reference')
    from eks.ibl_pupil_smoother import fit_eks_pupil
    from eks.multicam_smoother import fit_eks_mirrored_multicam, fit_eks_multicam
    from eks.singlecam_smoother import fit_eks_singlecam
    tmp = tempfile.mkdtemp()
    jobs = {
        'ref_singlecam_defaults': lambda: fit_eks_singlecam(os.path.join(REF, 'ibl-pupil'),
                                                            os.path.join(tmp, 'a', 'o.csv')),
        'ref_singlecam_s10': lambda: fit_eks_singlecam(os.path.join(REF, 'ibl-pupil'),
                                                       os.path.join(tmp, 'b', 'o.csv'), smooth_param=[10.0]),
        'ref_mirrored_defaults': lambda: fit_eks_mirrored_multicam(
            os.path.join(REF, 'mirror-mouse'), os.path.join(tmp, 'c', 'o.csv'),
            bodypart_list=['paw1LH', 'paw2LF'], camera_names=['top', 'bot'], quantile_keep_pca=95,
            inflate_vars=True),
        'ref_mirrored_s10': lambda: fit_eks_mirrored_multicam(
            os.path.join(REF, 'mirror-mouse'), os.path.join(tmp, 'd', 'o.csv'),
            bodypart_list=['paw1LH', 'paw2LF'], camera_names=['top', 'bot'], smooth_param=[10.0],
            quantile_keep_pca=95, inflate_vars=True),
        'ref_pupil_defaults': lambda: fit_eks_pupil(os.path.join(REF, 'ibl-pupil'),
                                                    os.path.join(tmp, 'e', 'o.csv'), smooth_params=[None, None]),
        'ref_pupil_fixed': lambda: fit_eks_pupil(os.path.join(REF, 'ibl-pupil'),
                                                 os.path.join(tmp, 'f', 'o.csv'), smooth_params=[0.99, 0.99]),
    }
    for name, job in jobs.items():
        res = job()
        df, s = res[0], res[1]
        vals = np.asarray(df.select_dtypes('number').values, dtype=np.float64)
        np.savez_compressed(os.path.join(OUT, f'{name}.npz'), columns=np.array([str(c) for c in df.columns]),
                            s_finals=np.asarray(s, dtype=np.float64), **pack(vals))
        print(name, vals.shape, 's =', np.asarray(s))
    return 0


if __name__ == '__main__':
    if '--from-reference' in sys.argv:
        sys.exit(from_reference())
    main()
