"""CPU tests of the host-side mirror of the reference interface (no GPU, no kernels): MarkerArray,
CSV formats, frame cropping, centring, KF initialisation.  Modelled on what the reference's own
tests assert (tests/test_marker_array.py, test_utils.py, test_singlecam_smoother.py:98-140,
test_multicam_smoother.py:284-341)."""
import os

import numpy as np
import pandas as pd
import pytest

from eks_amd.marker_array import (MarkerArray, input_dfs_to_markerArray, mA_to_stacked_array,
                                  stacked_array_to_mA)
from eks_amd import utils
from oracle import eks_oracle as orc


def _write_dlc_csv(path, names, T, rng, scorer='tracker'):
    cols = pd.MultiIndex.from_product([[scorer], names, ['x', 'y', 'likelihood']],
                                      names=['scorer', 'bodyparts', 'coords'])
    df = pd.DataFrame(rng.random((T, len(names) * 3)), columns=cols)
    df.to_csv(path)
    return df


def test_marker_array_basics():
    a = np.arange(2 * 3 * 4 * 5 * 3, dtype=float).reshape(2, 3, 4, 5, 3)
    ma = MarkerArray(a, data_fields=['x', 'y', 'likelihood'])
    assert ma.shape == (2, 3, 4, 5, 3)
    assert (ma.n_models, ma.n_cameras, ma.n_frames, ma.n_keypoints, ma.n_fields) == (2, 3, 4, 5, 3)
    np.testing.assert_array_equal(ma.slice('keypoints', 2).array, a[:, :, :, 2:3])
    np.testing.assert_array_equal(ma.slice('frames', [0, 3]).array, a[:, :, [0, 3]])
    sub = ma.slice_fields('likelihood', 'x')
    assert sub.data_fields == ['likelihood', 'x']
    np.testing.assert_array_equal(sub.array, a[..., [2, 0]])
    re = ma.reorder_data_fields(['y', 'x', 'likelihood'])
    np.testing.assert_array_equal(re.array[..., 0], a[..., 1])
    st = MarkerArray.stack([ma.slice('cameras', 0), ma.slice('cameras', 2)], 'cameras')
    np.testing.assert_array_equal(st.array, a[:, [0, 2]])
    sf = MarkerArray.stack_fields(ma.slice_fields('x'), ma.slice_fields('y', 'likelihood'))
    assert sf.data_fields == ['x', 'y', 'likelihood']
    np.testing.assert_array_equal(sf.array, a)
    assert MarkerArray(shape=(1, 1, 2, 2, 2), data_fields=['x', 'y']).array.dtype == np.float32
    clone = MarkerArray(marker_array=ma)
    assert clone.array is not ma.array and clone.data_fields == ma.data_fields
    assert ma.get_array(squeeze=True).shape == (2, 3, 4, 5, 3)
    assert 'models=2' in repr(ma)
    for bad in (lambda: MarkerArray(), lambda: MarkerArray(np.zeros((2, 2))),
                lambda: ma.slice('nope', 0), lambda: ma.slice_fields('z'),
                lambda: MarkerArray.stack([ma, MarkerArray(a[:, :, :2], data_fields=ma.data_fields)], 'cameras'),
                lambda: ma.reorder_data_fields(['x', 'y'])):
        with pytest.raises(AssertionError):
            bad()


def test_stacked_array_round_trip():
    rng = np.random.default_rng(0)
    a = rng.random((1, 3, 7, 4, 2))
    ma = MarkerArray(a, data_fields=['x', 'y'])
    s = mA_to_stacked_array(ma, 2)
    assert s.shape == (7, 6)
    np.testing.assert_array_equal(s[:, 2:4], a[0, 1, :, 2, :])         # [c0x, c0y, c1x, c1y, ...]
    back = stacked_array_to_mA(s, 3, ['x', 'y'])
    np.testing.assert_array_equal(back.array[0, :, :, 0, :], a[0, :, :, 2, :])
    with pytest.raises(AssertionError):
        mA_to_stacked_array(ma, 4)


def test_format_data_and_marker_array_from_csv(tmp_path):
    rng = np.random.default_rng(1)
    names = ['nose', 'ear']
    dfs = [_write_dlc_csv(tmp_path / f'pred.rng={i}.csv', names, 11, rng) for i in (1, 0)]
    (tmp_path / 'notes.txt').write_text('skip me')
    out, kp = utils.format_data(str(tmp_path))
    assert kp == names and len(out) == 2
    assert list(out[0].columns) == ['nose_x', 'nose_y', 'nose_likelihood', 'ear_x', 'ear_y', 'ear_likelihood']
    # directory listing is sorted: rng=0 (written second) comes first
    np.testing.assert_allclose(out[0]['nose_x'].to_numpy(), dfs[1][('tracker', 'nose', 'x')].to_numpy())
    ma = input_dfs_to_markerArray([out], kp, [''])
    assert ma.shape == (2, 1, 11, 2, 3) and ma.array.dtype == np.float64
    np.testing.assert_allclose(ma.array[1, 0, :, 1, 2], dfs[0][('tracker', 'ear', 'likelihood')].to_numpy())
    out2, _ = utils.format_data([str(tmp_path / 'pred.rng=1.csv'), str(tmp_path / 'pred.rng=0.csv')])
    np.testing.assert_allclose(out2[0].to_numpy(), out[0].to_numpy())
    with pytest.raises(FileNotFoundError):
        utils.format_data(str(tmp_path / 'notes.txt').replace('notes.txt', 'empty') if os.makedirs(
            tmp_path / 'empty', exist_ok=True) is None else '')
    with pytest.raises(ValueError):
        utils.format_data(3)
    with pytest.raises(NotImplementedError):
        utils.format_data([str(tmp_path / 'x.slp')])


def test_format_data_per_camera(tmp_path):
    rng = np.random.default_rng(2)
    for cam in ('top', 'bot'):
        for i in range(2):
            _write_dlc_csv(tmp_path / f'vid_{cam}_rng{i}.csv', ['paw'], 5, rng)
    out, kp = utils.format_data(str(tmp_path), camera_names=['top', 'bot'])
    assert kp == ['paw'] and [len(x) for x in out] == [2, 2]
    out, _ = utils.format_data({'top': [str(tmp_path / 'vid_top_rng0.csv')],
                                'bot': [str(tmp_path / 'vid_bot_rng1.csv')]}, camera_names=['top', 'bot'])
    assert [len(x) for x in out] == [1, 1]
    with pytest.raises(FileNotFoundError):
        utils.format_data(str(tmp_path), camera_names=['side'])


def test_output_index_layout():
    idx = utils.make_dlc_pandas_index(['a', 'b'], labels=['x', 'y'])
    assert idx.names == ['scorer', 'bodyparts', 'coords']
    assert list(idx) == [('ensemble-kalman_tracker', 'a', 'x'), ('ensemble-kalman_tracker', 'a', 'y'),
                         ('ensemble-kalman_tracker', 'b', 'x'), ('ensemble-kalman_tracker', 'b', 'y')]


def test_crop_frames_matches_oracle_semantics():
    y = np.arange(20).reshape(10, 2)
    assert utils.crop_frames(y, None) is y and utils.crop_frames(y, []) is y
    assert utils.crop_frames(y, [(None, None)]) is y
    for spec in ([(None, 3)], [(7, None), (0, 2)], [(2, 5)], [(0, 1), (1, 2), (5, 10)]):
        np.testing.assert_array_equal(utils.crop_frames(y, spec), orc.crop_frames(y, spec))
    for bad in ([(3, 3)], [(0, 11)], [(0, 5), (4, 6)], [(0.0, 3)], [[0, 3]], [(-1, 3)]):
        with pytest.raises(ValueError):
            utils.crop_frames(y, bad)
    with pytest.raises(TypeError):
        utils.crop_frames(y, ((0, 3),))
    R = utils.build_R_from_vars(np.array([[[0.0, 2.0], [3.0, 4.0]]]))
    assert R.shape == (1, 2, 2, 2) and R[0, 0, 0, 0] == 1e-12 and R[0, 1, 1, 1] == 4.0 and R[0, 0, 0, 1] == 0
    assert utils.crop_R(np.tile(np.eye(2), (3, 10, 1, 1)), [(2, 6)]).shape == (3, 4, 2, 2)


@pytest.mark.parametrize('q', [100, 50, 95, 5])
def test_center_predictions_matches_oracle(q):
    rng = np.random.default_rng(3)
    ens = rng.random((1, 2, 60, 4, 5))
    ens[0, :, :, :, 2:4] = np.round(ens[0, :, :, :, 2:4], 1)            # ties at the threshold
    ma = MarkerArray(ens, data_fields=['x', 'y', 'var_x', 'var_y', 'likelihood'])
    mask, cen, good, means = utils.center_predictions(ma, q)
    mask_o, cen_o, good_o, means_o, _ = orc.center_predictions(ens, q)
    np.testing.assert_array_equal(mask, mask_o)                          # indices bit-exact
    np.testing.assert_allclose(cen.array, cen_o, rtol=1e-13)
    np.testing.assert_allclose(good.array, good_o, rtol=1e-13)
    np.testing.assert_allclose(means.array, means_o, rtol=1e-13)
    assert cen.data_fields == ['x', 'y'] and means.shape == (1, 2, 1, 4, 2)
    with pytest.raises(AssertionError):
        utils.center_predictions(MarkerArray(np.zeros((2, 1, 3, 1, 5)), data_fields=ma.data_fields), 50)


def test_initialize_kalman_filter_shapes():
    from eks_amd.singlecam_smoother import initialize_kalman_filter
    rng = np.random.default_rng(4)
    cen = MarkerArray(rng.standard_normal((1, 1, 50, 3, 2)), data_fields=['x', 'y'])
    m0s, S0s, As, Qs, Cs = initialize_kalman_filter(cen)
    assert m0s.shape == (3, 2) and np.all(m0s == 0)
    eye = np.tile(np.eye(2), (3, 1, 1))
    for M in (As, Qs, Cs):
        np.testing.assert_array_equal(M, eye)
    np.testing.assert_allclose(S0s[:, 0, 0], cen.array[0, 0, :, :, 0].var(axis=0))
    assert np.all(S0s[:, 0, 1] == 0)


def test_initialize_kalman_filter_pca_matches_oracle(golden_dir):
    from sklearn.decomposition import PCA
    from eks_amd.multicam_smoother import initialize_kalman_filter_pca
    from eks_amd.stats import compute_pca
    g = np.load(os.path.join(golden_dir, 'mirror_mouse_multicam.npz'))
    ens = orc.ensemble(g['markers'])
    ma = MarkerArray(ens, data_fields=['x', 'y', 'var_x', 'var_y', 'likelihood'])
    mask, cen, good, means = utils.center_predictions(ma, 95.0)
    np.testing.assert_array_equal(mask, g['valid_mask'])
    pcas, good_pcs = compute_pca(mask, cen, good, n_components=3)
    assert all(isinstance(p, PCA) for p in pcas) and good_pcs[0].shape == (int(mask[:, 0].sum()), 3)
    m0s, S0s, As, Qs, Cs = initialize_kalman_filter_pca(good_pcs, pcas, 3)
    assert Cs.shape == (4, 4, 3) and Qs.shape == (4, 3, 3)
    sgn = np.sign(np.einsum('kod,kod->kd', Cs, g['Cs']))
    np.testing.assert_allclose(Cs * sgn[:, None, :], g['Cs'], atol=1e-7)
    np.testing.assert_allclose(S0s, g['S0s'], rtol=1e-6)
    np.testing.assert_allclose(Qs * sgn[:, :, None] * sgn[:, None, :], g['Qs'], atol=1e-7)
    assert np.abs(Qs).max(axis=(1, 2)) == pytest.approx(1.0)


def test_compute_initial_guesses_and_constant_R():
    from eks_amd.core import compute_initial_guesses, constant_R_from_timevarying
    ev = np.arange(40, dtype=float).reshape(20, 2) ** 2
    assert compute_initial_guesses(ev) == round(float(np.std(ev[1:] - ev[:-1])), 5)
    with pytest.raises(ValueError):
        compute_initial_guesses(np.ones((1, 2)))
    Rt = utils.build_R_from_vars(np.array([[1.0, 5e-5], [3.0, 2e-5], [2.0, 9e-5]]))
    np.testing.assert_allclose(np.diag(constant_R_from_timevarying(Rt)), [2.0, 1e-4])


def test_inflate_variance_semantics():
    from eks_amd.multicam_smoother import inflate_variance
    T = 10
    v = np.ones((T, 6))
    calm = {c: np.ones((T, 1)) for c in range(3)}
    out, changed = inflate_variance(v=v, maha_dict=calm, threshold=5, scalar=2)
    assert not changed and np.array_equal(out, v) and out is not v
    out, changed = inflate_variance(v=v, maha_dict=calm, threshold=0.5, scalar=2)
    assert changed and np.array_equal(out, 2 * v)
    mixed = {0: np.ones((T, 1)), 1: 2 * np.ones((T, 1)), 2: 4 * np.ones((T, 1))}
    out, changed = inflate_variance(v=v, maha_dict=mixed, threshold=1.5, scalar=3)
    assert changed and np.array_equal(out[:, :2], v[:, :2]) and np.array_equal(out[:, 2:], 3 * v[:, 2:])
    # exactly two views: one offending view inflates the whole frame
    out, changed = inflate_variance(v=np.ones((T, 4)), maha_dict={0: np.ones((T, 1)), 1: 2 * np.ones((T, 1))},
                                    threshold=1.5, scalar=3)
    assert changed and np.array_equal(out, 3 * np.ones((T, 4)))
    with pytest.raises(AssertionError):
        inflate_variance(v=np.ones((T, 2)), maha_dict={0: np.ones((T, 1))})


def test_compute_mahalanobis_properties_and_oracle():
    from eks_amd.stats import compute_mahalanobis
    rng = np.random.default_rng(0)
    n_t, n_cams, L = 120, 4, 3
    W = rng.standard_normal((2 * n_cams, L))
    x = rng.standard_normal((n_t, L)) @ W.T
    v = np.ones((n_t, 2 * n_cams))
    out = compute_mahalanobis(x, v, n_latent=L, v_quantile_threshold=None)
    assert set(out) == {'mahalanobis', 'posterior_variance', 'reconstructed'}
    assert len(out['mahalanobis']) == n_cams and out['mahalanobis'][0].shape == (n_t, 1)
    assert out['posterior_variance'][2].shape == (n_t, 2, 2)
    np.testing.assert_allclose(out['reconstructed'], x, atol=1e-6)          # exact-rank data is recovered
    assert not np.allclose(compute_mahalanobis(x, v, n_latent=1, v_quantile_threshold=None)['reconstructed'], x)
    # larger observation variance: distance shrinks, predictive variance grows, by the same factor
    x2, v2 = np.vstack([x, x]), np.vstack([v, 10 * v])
    o2 = compute_mahalanobis(x2, v2, n_latent=L - 1)
    assert o2['mahalanobis'][0][0, 0] == pytest.approx(10 * o2['mahalanobis'][0][n_t, 0], rel=1e-4)
    np.testing.assert_allclose(10 * o2['posterior_variance'][0][0], o2['posterior_variance'][0][n_t], rtol=1e-4)
    # supplied loading matrix / mean skip the fit; likelihood filter is accepted
    compute_mahalanobis(x, v, n_latent=L, loading_matrix=rng.standard_normal((2 * n_cams, L)),
                        mean=rng.standard_normal(2 * n_cams))
    compute_mahalanobis(x, v, n_latent=L, likelihoods=rng.random((n_t, n_cams)), likelihood_threshold=0.1,
                        v_quantile_threshold=None)
    # vectorised implementation == loop-style oracle restatement
    xn = x + 0.3 * rng.standard_normal(x.shape)
    vn = rng.gamma(2.0, 0.3, x.shape) + 0.01
    got = compute_mahalanobis(xn, vn, n_latent=2)
    ref = orc.mahalanobis_loop(xn, vn, n_latent=2)
    for c in range(n_cams):
        np.testing.assert_allclose(got['mahalanobis'][c][:, 0], ref[c], rtol=1e-10)


def test_variance_inflation_pipeline_matches_oracle(golden_dir):
    from eks_amd.multicam_smoother import mA_compute_maha
    g = np.load(os.path.join(golden_dir, 'mirror_mouse_multicam.npz'))
    ens = orc.ensemble(g['markers'][:, :, :, :2])
    ma = MarkerArray(ens, data_fields=['x', 'y', 'var_x', 'var_y', 'likelihood'])
    _, cen, _, _ = utils.center_predictions(ma, 95.0)
    kw = {}
    infl = mA_compute_maha(cen, ma.slice_fields('var_x', 'var_y'), ma.slice_fields('likelihood'), 3,
                           inflate_vars_kwargs=kw)
    assert kw == {'likelihood_threshold': 0.9, 'v_quantile_threshold': 50.0}    # defaults written back
    assert infl.shape == (1, 2, 2000, 2, 2) and infl.data_fields == ['var_x', 'var_y']
    got = np.stack([mA_to_stacked_array(infl, k) for k in range(2)])                # (K,T,2V)
    np.testing.assert_allclose(got, g['infl_vars'].astype(np.float64), rtol=1e-6)
    raw = np.stack([mA_to_stacked_array(ma.slice_fields('var_x', 'var_y'), k) for k in range(2)])
    ratio = got / raw
    assert np.all(np.isclose(ratio, 1) | (ratio > 9.99)) and (ratio > 9.99).any()   # only x10^n inflations


@pytest.mark.parametrize('p,k,n', [(4, 3, 20000), (6, 3, 4000), (12, 3, 3000), (4, 2, 500), (13, 3, 2000)])
def test_factor_analysis_from_moments_is_sklearns_fit(p, k, n):
    """stats.factor_analysis_from_moments (the variance-inflation driver's fit, from the fitted rows'
    covariance alone) against sklearn.decomposition.FactorAnalysis(n_components).fit - what the
    reference calls (eks/stats.py:114-117): same number of EM iterations, noise variances and
    loadings (up to the sign of a column) to 1e-9, for every width p <= n_components + 10 where
    sklearn's randomized SVD is exact."""
    from sklearn.decomposition import FactorAnalysis

    from eks_amd.stats import factor_analysis_from_moments
    rng = np.random.default_rng(p * 100 + k)
    Wt = rng.normal(size=(p, k)) * 3
    X = rng.normal(size=(n, k)) @ Wt.T + rng.normal(size=(n, p)) * rng.uniform(0.3, 2, size=p) + rng.normal(size=p) * 10
    fa = FactorAnalysis(n_components=k).fit(X)
    Xc = X - X.mean(0)
    W, psi, it = factor_analysis_from_moments(Xc.T @ Xc / n, n, k)
    assert it == fa.n_iter_
    np.testing.assert_allclose(psi, fa.noise_variance_, rtol=1e-9)
    np.testing.assert_allclose(np.abs(W), np.abs(fa.components_.T), rtol=0, atol=1e-9 * np.abs(W).max())
    # what the Mahalanobis step uses: the column space
    P1, P2 = W @ np.linalg.pinv(W), fa.components_.T @ np.linalg.pinv(fa.components_.T)
    assert np.abs(P1 - P2).max() < 1e-10


def test_page_locked_result_cap_follows_the_arrays_callers_hold(monkeypatch):
    """core._to_host counts page-locked result bytes for as long as the CALLER holds the arrays (or views
    of them) and switches to pageable results beyond the cap (a finaliser on the staging
    tensor fired on return because ndarray.base is another tensor object).  Plain host tensors stand in
    for page-locked ones here (no GPU)."""
    import gc
    import torch
    from eks_amd import core
    made = []

    def fake_pinned(shape, dtype):
        made.append(1)
        return torch.empty(shape, dtype=dtype)

    monkeypatch.setattr(core, '_pinned_empty', fake_pinned)
    monkeypatch.setattr(core, '_PINNED_CAP_BYTES', 3000)
    monkeypatch.setattr(core, '_pinned_live', [0])
    a, = core._to_host(torch.arange(500, dtype=torch.float32))            # 2000 B, staged
    assert len(made) == 1 and core._pinned_live[0] == 2000
    view = np.swapaxes(a.reshape(10, 50), 0, 1)
    del a
    gc.collect()
    assert core._pinned_live[0] == 2000                                   # a view still holds the buffer
    b, = core._to_host(torch.arange(500, dtype=torch.float32))            # 2000 + 2000 > cap: pageable
    assert len(made) == 1 and core._pinned_live[0] == 2000
    np.testing.assert_array_equal(b, np.arange(500, dtype=np.float32))
    del view
    gc.collect()
    assert core._pinned_live[0] == 0
    c, = core._to_host(torch.arange(500, dtype=torch.float32))            # room again
    assert len(made) == 2 and core._pinned_live[0] == 2000


def test_percentile_from_two_order_statistics_is_numpys_bit_for_bit():
    """utils.percentile_ranks + percentile_from_order_stats (what hip_ops.percentile does with the two
    device-selected order statistics) against numpy.percentile on float32 columns: same dtype, same bits -
    NaN slices, duplicates, q = 0 / 100 and single-row matrices included (reference eks/utils.py:318-322,
    eks/stats.py:109-112 call numpy.percentile on float32 arrays)."""
    from eks_amd.utils import percentile_from_order_stats, percentile_ranks
    rng = np.random.default_rng(0)
    for trial in range(600):
        n, K = int(rng.integers(1, 3000)), int(rng.integers(1, 5))
        q = float(rng.choice([0, 100, 50, 95, 25, 37.5, 99.9, rng.uniform(0, 100)]))
        a = (rng.standard_normal((n, K)) * rng.choice([1e-3, 1, 1e4])).astype(np.float32)
        if trial % 7 == 0:
            a = np.abs(a)
        if trial % 11 == 0:
            a[rng.integers(0, n), rng.integers(0, K)] = np.nan
        if trial % 13 == 0:
            a = np.round(a)
        ref = np.percentile(a, q, axis=0)
        lo, hi, g = percentile_ranks(n, q, np.float32)
        srt = np.sort(a, axis=0)                                        # NaNs last, as the device kernel orders them
        got = percentile_from_order_stats(np.stack([srt[lo], srt[hi]], axis=-1), g, np.isnan(a).sum(axis=0))
        assert got.dtype == ref.dtype and np.array_equal(got, ref, equal_nan=True), (n, q)


def test_model_flags_diag_unit_and_positive_definite_q():
    """hip_ops.model_flags from HOST copies of the parameters: scalar-chain model (DIAG_MODEL, UNIT_AC) as before,
    and Q_PD - every Q[k] positive definite with a margin - which lets the general path take the loss gradient
    from the smoothing distribution (include/eks_hip.h); a singular or indefinite Q keeps the dual-number kernels."""
    from eks_amd import _lib, hip_ops
    K, D, O = 3, 3, 4
    rng = np.random.default_rng(0)
    eyeD = np.tile(np.eye(D), (K, 1, 1))
    L = rng.standard_normal((K, D, D))
    Q = L @ np.swapaxes(L, 1, 2) + 0.1 * np.eye(D)
    C = rng.standard_normal((K, O, D))
    assert hip_ops.model_flags(eyeD, eyeD, C, Q) == _lib.FLAG_Q_PD
    Qs = Q.copy()
    Qs[1] = np.outer(L[1, :, 0], L[1, :, 0])                       # rank one
    assert hip_ops.model_flags(eyeD, eyeD, C, Qs) == 0
    Qn = Q.copy()
    Qn[2, 0, 0] = -1.0                                             # indefinite
    assert hip_ops.model_flags(eyeD, eyeD, C, Qn) == 0
    Qz = Q.copy()
    Qz[0] = 0.0
    assert hip_ops.model_flags(eyeD, eyeD, C, Qz) == 0
    # conditioning threshold: cond(Q) <= 1e6 for every keypoint, decided per batch
    Qc = Q.copy()
    Qc[1] = np.diag([1.0, 0.5, 2e-6])
    assert hip_ops.model_flags(eyeD, eyeD, C, Qc) == _lib.FLAG_Q_PD
    Qc[1] = np.diag([1.0, 0.5, 5e-7])
    assert hip_ops.model_flags(eyeD, eyeD, C, Qc) == 0
    eye2 = np.tile(np.eye(2), (K, 1, 1))
    f = hip_ops.model_flags(eye2 * 3.0, eye2, eye2, eye2 * 0.5)
    assert f == _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC | _lib.FLAG_Q_PD
    f = hip_ops.model_flags(eye2 * 3.0, eye2 * 0.9, eye2, eye2 * 0.0)
    assert f == _lib.FLAG_DIAG_MODEL


def test_pca_from_moments_reproduces_sklearn_components_and_signs():
    """stats.pca_from_moments: principal axes from the covariance alone, signs by the installed scikit-learn's rule
    (decided once by pca_sign_rule) - the multi-camera driver reduces the fitted rows to this covariance on the
    device instead of downloading them for a per-keypoint sklearn fit."""
    from sklearn.decomposition import PCA
    from eks_amd import stats
    assert stats.pca_sign_rule() in ('v', 'u')
    rng = np.random.default_rng(3)
    for n, F, L in ((20000, 4, 3), (300, 8, 3), (1000, 12, 5), (60, 4, 2)):
        X = np.cumsum(rng.standard_normal((n, F)), axis=0) @ rng.standard_normal((F, F)) + 50.0
        ref = PCA(n_components=L).fit(X).components_
        Xc = X - X.mean(axis=0)

        def extreme(axes):
            sc = Xc @ axes.T
            return sc[np.argmax(np.abs(sc), axis=0), np.arange(sc.shape[1])]

        comp = stats.pca_from_moments(Xc.T @ Xc / (n - 1), L, extreme)
        assert np.abs(comp - ref).max() < 1e-9
    # the other rule, against a hand-written svd_flip(u_based_decision=True)
    X = rng.standard_normal((200, 5)) @ rng.standard_normal((5, 5))
    Xc = X - X.mean(axis=0)
    U, S, Vt = np.linalg.svd(Xc, full_matrices=False)
    signs = np.sign(U[np.argmax(np.abs(U), axis=0), np.arange(U.shape[1])])
    want = (Vt * signs[:, None])[:3]
    got = stats._apply_sign_rule(stats._pca_axes(Xc.T @ Xc / 199, 3), 'u', Xc @ stats._pca_axes(Xc.T @ Xc / 199, 3).T)
    assert np.abs(got - want).max() < 1e-9


def test_initial_guesses_for_all_keypoints_equal_the_per_keypoint_calls_bit_for_bit():
    """core._initial_guesses_per_keypoint (one nanstd over a (K, (T'-1) O) layout) against the loop of
    compute_initial_guesses calls it replaces in run_kalman_smoother (reference eks/core.py:104-133, :233-236):
    identical floats for float32 and float64 variances, NaNs, an all-NaN keypoint and a constant one (-> 2.0)."""
    import warnings
    from eks_amd import core
    rng = np.random.default_rng(1)
    for dt in (np.float32, np.float64):
        for T, K, O in ((2000, 40, 2), (300, 7, 4), (2, 3, 2)):
            ev = rng.gamma(2, 1, (T, K, O)).astype(dt)
            ev[rng.random(ev.shape) < 0.02] = np.nan
            if K > 3:
                ev[:, 2, :] = np.nan
                ev[:, 3, :] = 1.0
            loop = np.full(K, 2.0)
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                for k in range(K):
                    g = float(core.compute_initial_guesses(ev[:, k, :]) or 2.0)
                    loop[k] = g if (np.isfinite(g) and g > 0) else 2.0
            assert np.array_equal(loop, core._initial_guesses_per_keypoint(ev))


def test_gram_never_builds_the_outer_product_tensor_beyond_its_cap():
    """the PCA covariance / factor-analysis moments were reduced through a (..., n, F, F)
    temporary - K * n * F^2 * 8 bytes, several GB for a long many-camera session.  Above the cap the reduction
    goes column by column (and in row slabs), with the same result."""
    import torch
    from eks_amd import multicam_smoother as mc
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 5000, 12, dtype=torch.float64, generator=g)           # a 6-camera shape, 3 keypoints
    ref = torch.einsum('kni,knj->kij', x, x)
    small = mc._gram(x)                                                       # 3 * 5000 * 144 * 8 = 17 MB: direct form
    assert torch.allclose(small, ref, rtol=1e-12, atol=1e-9)
    peak = []
    orig_sum = torch.Tensor.sum

    def spy(self, *a, **k):
        peak.append(self.numel() * self.element_size())
        return orig_sum(self, *a, **k)
    torch.Tensor.sum = spy
    try:
        for cap in (1 << 20, 100_000, 1):                                     # column form, row slabs, one row at a time
            if cap == 1:
                xx = x[:, :40]
                got, want = mc._gram(xx, max_temp_bytes=cap), torch.einsum('kni,knj->kij', xx, xx)
            else:
                got, want = mc._gram(x, max_temp_bytes=cap), ref
            assert torch.allclose(got, want, rtol=1e-12, atol=1e-9)
            assert max(peak) <= max(cap, 3 * 12 * 8), (cap, max(peak))        # never more than the cap (or one row)
            peak.clear()
    finally:
        torch.Tensor.sum = orig_sum
    # 2-D input (the per-keypoint calls of the prior / process-noise set-up)
    y = torch.randn(7000, 3, dtype=torch.float64, generator=g)
    assert torch.allclose(mc._gram(y, max_temp_bytes=4096), y.T @ y, rtol=1e-12, atol=1e-9)


def test_guess_rows_prepared_by_torch_equal_the_host_preparation_bit_for_bit():
    """The optimiser's initial guesses from a device tensor: the differences and their transposition run in torch
    (on the device in the drivers), the nanstd and the rounding stay numpy's - same bits as the all-host form."""
    import torch
    from eks_amd import core
    rng = np.random.default_rng(3)
    ev = rng.gamma(2.0, 0.3, (2600, 9, 2)).astype(np.float32)
    ev[17, 2, 1] = np.nan
    ev[:, 5] = 0.25                                            # a constant keypoint: std 0 -> the 2.0 fallback
    a = core._initial_guesses_per_keypoint(ev)
    b = core._initial_guesses_per_keypoint(rows=core._guess_rows_from_device(torch.as_tensor(ev)))
    np.testing.assert_array_equal(a, b)
    assert a[5] == 2.0
    with pytest.raises(ValueError, match='Not enough frames'):
        core._guess_rows_from_device(torch.as_tensor(ev[:1]))


def test_np_sum_program_is_numpys_pairwise_summation_order():
    """hip_ops.np_sum_program(n) - the leaves and combine order eks_np_nanstd_rows is given - interpreted in float32
    on the host reproduces numpy.sum of n contiguous float32 values bit for bit (numpy sums pairwise:
    numpy/_core/src/umath/loops_utils.h.src); the device kernel follows the same tables (GPU test)."""
    from eks_amd.hip_ops import np_sum_program
    f = np.float32

    def leaf(a):
        n = len(a)
        if n < 8:
            r = f(0.0)
            for x in a:
                r = f(r + x)
            return r
        r = [f(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = f(r[j] + a[i + j])
            i += 8
        res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
        while i < n:
            res = f(res + a[i])
            i += 1
        return res

    rng = np.random.default_rng(0)
    for n in (1, 7, 8, 9, 128, 129, 255, 256, 257, 1000, 3998, 7996, 8192):
        leaves, ops = np_sum_program(n)
        assert leaves[:, 1].sum() == n and (leaves[:, 1] <= 128).all() and len(ops) == len(leaves) - 1
        for _ in range(4):
            a = (rng.standard_normal(n) * 30).astype(f)
            slot = [leaf(a[s0:s0 + ln]) for s0, ln in leaves] + [None] * len(ops)
            for dst, x, y in ops:
                slot[dst] = f(slot[x] + slot[y])
            assert f(0.0) + slot[-1] == np.sum(a), n


def test_model_flags_in_one_c_pass_decides_what_the_numpy_route_decides():
    """hip_ops.model_flags asks the library (eks_host_model_flags: one pass over the host arrays, in front of every
    call's first launch) when the arrays are C-contiguous float64, and NumPy otherwise (and for a finite Q that is not
    diagonal, whose eigenvalues decide EKS_FLAG_Q_PD).  Both routes, the same answers: diagonal / unit / decaying
    models, an ill-conditioned and a non-finite Q, off-diagonal entries and NaNs in each array, D != O."""
    from eks_amd import hip_ops
    rng = np.random.default_rng(11)

    def both(S0, A, C, Q):
        fast = hip_ops.model_flags(*(np.ascontiguousarray(a, dtype=np.float64) for a in (S0, A, C, Q)))
        slow = hip_ops.model_flags(*(np.asarray(a, dtype=np.float32) for a in (S0, A, C, Q)))   # (not float64: NumPy)
        return fast, slow

    for K, D in ((1, 1), (7, 2), (40, 3)):
        eye = np.tile(np.eye(D), (K, 1, 1))
        S0 = eye * rng.uniform(0.5, 4.0, (K, 1, 1))
        diagQ = eye * rng.uniform(0.5, 2.0, (K, D, 1))
        cases = {
            'unit': (S0, eye, eye, eye),
            'decaying': (S0, eye * 0.5, eye * 0.75, diagQ),
            'ill-conditioned Q': (S0, eye, eye, eye * np.where(np.arange(D) == 0, 1.0, 1e-7 if D > 1 else 1.0)[None, :, None]),
            'Q with a NaN on the diagonal': (S0, eye, eye, np.where(eye > 0, np.nan, 0.0)),
        }
        if D > 1:
            off = np.zeros_like(eye)
            off[K // 2, 0, 1] = off[K // 2, 1, 0] = 0.25
            nan_off = np.zeros_like(eye)
            nan_off[0, 1, 0] = np.nan
            cases.update({
                'S0 off-diagonal': (S0 + off, eye, eye, eye), 'A off-diagonal': (S0, eye + off, eye, eye),
                'C off-diagonal': (S0, eye, eye + off, eye), 'Q off-diagonal (PD)': (S0, eye, eye, eye + off),
                'Q off-diagonal (indefinite)': (S0, eye, eye, eye + 8.0 * off), 'A with a NaN off the diagonal': (S0, eye + nan_off, eye, eye),
                'Q with a NaN off the diagonal': (S0, eye, eye, eye + nan_off),
            })
            C_tall = np.tile(np.vstack([np.eye(D), np.ones((1, D))]), (K, 1, 1))
            cases['D != O'] = (S0, eye, C_tall, eye)
        for name, arrs in cases.items():
            fast, slow = both(*arrs)
            assert fast == slow, (K, D, name, fast, slow)
    fast, _ = both(np.ones((3, 1, 1)), np.ones((3, 1, 1)), np.ones((3, 1, 1)), np.ones((3, 1, 1)))
    assert fast == hip_ops.FLAG_DIAG_MODEL | hip_ops.FLAG_UNIT_AC | hip_ops.FLAG_Q_PD
