"""CPU: the oracle reproduces the committed golden vectors from the committed inputs (guards the
oracle against drift), and the fixtures carry the reference's data-format pins."""
import os

import numpy as np
import pytest

from oracle import eks_oracle as orc


@pytest.fixture(scope='module')
def pupil(golden_dir):
    return np.load(os.path.join(golden_dir, 'ibl_pupil_singlecam.npz'))


@pytest.fixture(scope='module')
def mouse(golden_dir):
    return np.load(os.path.join(golden_dir, 'mirror_mouse_multicam.npz'))


def _check(full, g, prefix, rtol=2e-6):
    rows = full[g['keep_idx']]
    ref = g[f'{prefix}_rows'].astype(np.float64)
    scale = np.abs(ref).max(axis=0)
    assert (np.abs(rows - ref) / scale).max() < rtol + 1e-7          # rows are stored as float32
    np.testing.assert_allclose(full.sum(axis=0), g[f'{prefix}_colsum'], rtol=1e-9,
                               atol=1e-9 * g[f'{prefix}_colabs'].max())


def test_fixture_shapes_and_format_pins(pupil, mouse):
    assert pupil['markers'].shape == (5, 1, 2000, 4, 3) and pupil['markers'].dtype == np.float32
    assert list(pupil['keypoints']) == ['pupil_top_r', 'pupil_right_r', 'pupil_bottom_r', 'pupil_left_r']
    assert not np.isnan(pupil['markers']).any()
    hdr = list(pupil['csv_header'])
    assert hdr[0].startswith('scorer,') and hdr[1].startswith('bodyparts,') and hdr[2].startswith('coords,x,y,likelihood')
    assert mouse['markers'].shape == (5, 2, 2000, 4, 3)
    assert 'paw1LH_top' in mouse['csv_header'][1] and 'paw1LH_bot' in mouse['csv_header'][1]


def test_oracle_reproduces_singlecam_fixed_s(pupil):
    arrs = orc.singlecam_arrays(pupil['markers'])
    s, ms, Vs, _ = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'],
                                           arrs['Qs'], arrs['ensemble_vars'], smooth_param=[10.0])
    np.testing.assert_array_equal(s, 10.0)
    _check(orc.singlecam_outputs(arrs, s, ms, Vs), pupil, 's10')


def test_oracle_reproduces_singlecam_adam(pupil):
    arrs = orc.singlecam_arrays(pupil['markers'])
    np.testing.assert_allclose([orc.compute_initial_guess(arrs['ensemble_vars'][:, k]) for k in range(4)],
                               pupil['guesses'])
    s, ms, Vs, info = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'],
                                              arrs['Cs'], arrs['Qs'], arrs['ensemble_vars'])
    np.testing.assert_allclose(s, pupil['adam_s'], rtol=1e-9)
    np.testing.assert_array_equal(info['iters'], pupil['adam_iters'])
    _check(orc.singlecam_outputs(arrs, s, ms, Vs), pupil, 'adam')
    # Adam and the 64-point grid land in the same basin of the same loss
    assert np.all(np.abs(np.log(pupil['adam_s']) - np.log(pupil['grid_s'])) < 0.3)


def test_oracle_reproduces_multicam_fixed_s(mouse):
    from sklearn.decomposition import PCA

    def sk_pca(X, n):
        p = PCA(n_components=n).fit(X)
        return p.components_, p.mean_

    arrs = orc.multicam_arrays(mouse['markers'], quantile_keep_pca=95.0, n_latent=3, pca_fit=sk_pca)
    np.testing.assert_array_equal(arrs['mask'], mouse['valid_mask'])          # indices bit-exact
    np.testing.assert_array_equal(arrs['good_idx'], mouse['good_idx'])
    # PCA directions up to sign
    sgn = np.sign(np.einsum('kod,kod->kd', arrs['Cs'], mouse['Cs']))
    np.testing.assert_allclose(arrs['Cs'] * sgn[:, None, :], mouse['Cs'], atol=1e-9)
    s, ms, Vs, _ = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'],
                                           arrs['Qs'], arrs['ensemble_vars'], smooth_param=[10.0])
    cams, lat = orc.multicam_outputs(arrs, ms, Vs)
    for c, co in enumerate(cams):
        _check(co, mouse, f's10_cam{c}')        # observation space: PCA-sign invariant


# ---- second golden tier: vectors produced by the REFERENCE itself (tools/make_golden.py --from-reference)
def test_oracle_against_reference_generated_vectors_when_present(golden_dir):
    """The day jax / dynamax / optax are importable in the build container, `python
    tools/make_golden.py --from-reference` runs the reference's own drivers on its own data and
    commits tests/golden/ref_*.npz; this test then pins the oracle to upstream numbers at the
    reference's own tolerance (atol 1e-4, tests/conftest.py:95-101).  Until then the oracle is
    PARITY UNPINNED and this test says so."""
    import glob
    files = sorted(glob.glob(os.path.join(golden_dir, 'ref_*.npz')))
    if not files:
        pytest.skip('parity unpinned: no reference-generated vectors (jax / dynamax / optax absent when the '
                    'goldens were made; run tools/make_golden.py --from-reference where they import)')
    from oracle import f32_forecast as ff
    pup = np.load(os.path.join(golden_dir, 'ibl_pupil_singlecam.npz'))
    mouse = np.load(os.path.join(golden_dir, 'mirror_mouse_multicam.npz'))

    def rows_of(full):
        idx = np.unique(np.concatenate([np.arange(0, 16), np.arange(0, full.shape[0], 4),
                                        np.arange(full.shape[0] - 16, full.shape[0])]))
        return full[idx]

    def close(out64, out32, ref_rows, missed_labels, name):
        """Labels the forecast (tests/test_f32_forecast.py) predicts the float64 oracle to hold at upstream's own
        bar are held to it (atol 1e-4); the labels it predicts to miss - float32 storage of values beyond 1024,
        float32 covariance-form rounding - are compared through the float32 EMULATION of upstream's arithmetic,
        at upstream's bar plus 8 float32 ulps of the value (the emulation is not a bitwise twin of XLA)."""
        got64, got32 = rows_of(out64), rows_of(np.asarray(out32, np.float64))
        n_lab = len(ff.LABELS)
        lab = np.arange(ref_rows.shape[1]) % n_lab
        miss = np.isin(lab, [ff.LABELS.index(l) for l in missed_labels])
        np.testing.assert_allclose(got64[:, ~miss], ref_rows[:, ~miss], rtol=0, atol=1e-4, err_msg=name)
        tol = 1e-4 + 8.0 * np.spacing(np.abs(ref_rows[:, miss]).astype(np.float32)).astype(np.float64)
        assert (np.abs(got32[:, miss] - ref_rows[:, miss]) <= tol).all(), name

    for f in files:
        ref = np.load(f)
        name = os.path.basename(f)
        fixed = 's10' in name or 'fixed' in name
        if name.startswith('ref_singlecam'):
            arrs = orc.singlecam_arrays(pup['markers'])
            s, ms, Vs, _ = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'],
                                                   arrs['Qs'], arrs['ensemble_vars'],
                                                   smooth_param=[10.0] if fixed else None)
            if fixed:          # outputs at a given s: the reference's own bar (forecast: passes with 20x to spare)
                np.testing.assert_allclose(rows_of(orc.singlecam_outputs(arrs, s, ms, Vs)), ref['rows'], rtol=0,
                                           atol=1e-4, err_msg=name)
            else:              # the optimiser's float32 stop test is chaotic: s within its flat basin
                assert np.all(np.abs(np.log(s) - np.log(ref['s_finals'])) < 0.5), name
        elif name.startswith('ref_mirrored'):
            from sklearn.decomposition import PCA

            def sk_pca(X, n):
                p = PCA(n_components=n).fit(X)
                return p.components_, p.mean_
            kp = [list(mouse['keypoints']).index(k) for k in ('paw1LH', 'paw2LF')]
            arrs = orc.multicam_arrays(mouse['markers'][:, :, :, kp], quantile_keep_pca=95.0, n_latent=3,
                                       pca_fit=sk_pca, inflate_vars=True)
            if not fixed:
                s = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'],
                                            arrs['ensemble_vars'])[0]
                assert np.all(np.abs(np.log(s) - np.log(ref['s_finals'])) < 0.5), name
                continue
            s = np.full(len(kp), 10.0)
            Rd = np.clip(np.swapaxes(arrs['ensemble_vars'], 0, 1), 1e-12, None)
            args = (arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'], s, Rd)
            ms64, Vs64, _ = orc.kalman_smoother(*args)
            ms32, Vs32, _ = orc.kalman_smoother_f32(*args)
            a32 = dict(arrs)
            for k in ('Cs', 'means', 'ens', 'ensemble_vars'):
                a32[k] = np.asarray(arrs[k], np.float32)
            c64, c32 = orc.multicam_outputs(arrs, ms64, Vs64)[0], orc.multicam_outputs(a32, ms32, Vs32)[0]
            # the reference concatenates the per-camera tables column-wise, bodyparts renamed {kp}_{cam}
            # (eks/multicam_smoother.py:141-152)
            close(np.concatenate(c64, axis=1), np.concatenate(c32, axis=1), ref['rows'],
                  ('x', 'y', 'x_ens_var', 'y_ens_var', 'x_posterior_var', 'y_posterior_var'), name)
        elif name.startswith('ref_pupil') and fixed:
            gp = np.load(os.path.join(golden_dir, 'ibl_pupil_pupil.npz'))
            pa = orc.pupil_arrays(pup['markers'][:, :, :, gp['order']])
            s, ms, Vs, _ = orc.run_pupil_kalman_smoother(pa['ys'], pa['m0'], pa['S0'], pa['C'], pa['ensemble_vars'],
                                                         pa['latent_vars'], smooth_params=[0.99, 0.99])
            np.testing.assert_allclose(rows_of(orc.pupil_outputs(pa, ms, Vs)), ref['rows'], rtol=0, atol=1e-4,
                                       err_msg=name)
