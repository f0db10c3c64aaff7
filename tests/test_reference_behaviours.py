"""Host-side behaviours the reference's own unit tests pin (tests/test_utils.py, test_stats.py,
test_multicam_smoother.py, test_ibl_pupil_smoother.py, test_ibl_paw_multicam_smoother.py), restated
against this package: error types and messages of the data-format helpers, the ragged good-frame
handling of center_predictions, compute_pca with a supplied model, compute_mahalanobis with singular
inputs, the pupil geometry helpers on empty / single-row / all-NaN input, the paw camera-mean helpers.
CPU only - these are the callers and data formats either side of the device path."""
import logging
import os
import warnings

import numpy as np
import pandas as pd
import pytest

from eks_amd.marker_array import MarkerArray
from eks_amd.stats import compute_mahalanobis, compute_pca
from eks_amd.utils import center_predictions, crop_frames, format_data


def _write_predictions(folder, name, keypoints, n_frames=6, seed=0):
    """A DLC-style prediction CSV (three header rows: scorer / bodyparts / coords)."""
    rng = np.random.default_rng(seed)
    cols = pd.MultiIndex.from_product([['net'], keypoints, ['x', 'y', 'likelihood']],
                                      names=['scorer', 'bodyparts', 'coords'])
    path = os.path.join(str(folder), name)
    pd.DataFrame(rng.random((n_frames, len(cols))), columns=cols).to_csv(path)
    return path


# ------------------------------------------------------------------------------------------
# crop_frames (reference tests/test_utils.py:17-98)
# ------------------------------------------------------------------------------------------
def test_crop_frames_passthrough_forms_return_the_input_itself():
    y = np.arange(12)
    assert crop_frames(y, None) is y
    assert crop_frames(y, []) is y
    assert crop_frames(y, [(None, None)]) is y


def test_crop_frames_spans_open_ends_and_order():
    y = np.arange(20)
    np.testing.assert_array_equal(crop_frames(y, [(3, 7)]), np.arange(3, 7))
    np.testing.assert_array_equal(crop_frames(y, [(None, 3), (15, None)]), np.r_[0:3, 15:20])
    # spans are applied in sorted order whatever order they were given in
    np.testing.assert_array_equal(crop_frames(y, [(10, 12), (1, 3)]), np.r_[1:3, 10:12])


@pytest.mark.parametrize('spans', [[(1, 3, 5)], [(1, 40)], [(6, 5)], [(2, 6), (5, 10)], [(-1, 4)], [(1.5, 4)]])
def test_crop_frames_rejects_malformed_spans(spans):
    with pytest.raises(ValueError):
        crop_frames(np.arange(20), spans)


def test_crop_frames_rejects_a_non_list():
    with pytest.raises(TypeError):
        crop_frames(np.arange(20), ((1, 3),))


# ------------------------------------------------------------------------------------------
# format_data (reference tests/test_utils.py:101-232)
# ------------------------------------------------------------------------------------------
def test_format_data_directory_and_list_inputs_agree(tmp_path):
    a = _write_predictions(tmp_path, 'model0.csv', ['nose', 'tail'], seed=1)
    b = _write_predictions(tmp_path, 'model1.csv', ['nose', 'tail'], seed=2)
    (tmp_path / 'notes.txt').write_text('not a prediction file')           # skipped
    dfs_dir, names_dir = format_data(str(tmp_path))
    dfs_list, names_list = format_data([b, a])                              # sorted like the directory
    assert names_dir == names_list == ['nose', 'tail'] and len(dfs_dir) == len(dfs_list) == 2
    assert list(dfs_dir[0].columns) == ['nose_x', 'nose_y', 'nose_likelihood', 'tail_x', 'tail_y', 'tail_likelihood']
    for u, v in zip(dfs_dir, dfs_list):
        pd.testing.assert_frame_equal(u, v)


def test_format_data_errors(tmp_path):
    with pytest.raises(ValueError, match='input_source must be'):
        format_data('/nonexistent/path/that/does/not/exist')
    (tmp_path / 'readme.txt').write_text('not a csv')
    with pytest.raises(FileNotFoundError, match='no valid marker input files'):
        format_data(str(tmp_path))


def test_format_data_per_camera_grouping(tmp_path, caplog):
    for m in range(2):
        _write_predictions(tmp_path, f'model{m}_top.csv', ['nose'], seed=m)
    _write_predictions(tmp_path, 'model0_bot.csv', ['nose'], seed=7)
    with caplog.at_level(logging.WARNING, logger='eks_amd.utils'):
        per_cam, names = format_data(str(tmp_path), camera_names=['top', 'bot'])
    assert [len(c) for c in per_cam] == [2, 1] and names == ['nose']
    assert 'unequal number of seed files per camera' in caplog.text
    with pytest.raises(FileNotFoundError, match="no files matching camera 'side'"):
        format_data(str(tmp_path), camera_names=['top', 'side'])


def test_format_data_camera_mapping(tmp_path):
    paths = {cam: [_write_predictions(tmp_path, f'model{m}_{cam}.csv', ['nose'], seed=m) for m in (1, 0)]
             for cam in ('top', 'bot')}
    per_cam, names = format_data(paths, camera_names=['top', 'bot'])
    assert [len(c) for c in per_cam] == [2, 2] and names == ['nose']


# ------------------------------------------------------------------------------------------
# center_predictions with ragged good-frame counts (reference tests/test_multicam_smoother.py:284-341)
# ------------------------------------------------------------------------------------------
def test_center_predictions_truncates_every_keypoint_to_the_shortest_good_run():
    rng = np.random.default_rng(42)
    V, T, K = 2, 24, 5
    xy = rng.normal(size=(1, V, T, K, 2)) * 10
    var = np.abs(rng.normal(size=(1, V, T, K, 2))) * 5
    var[:, :, 0] = 1e6                                  # frame 0 never passes
    for k in range(K):                                  # ties at the threshold differ per keypoint
        tied = rng.choice(np.arange(1, T), size=int(rng.integers(5, T)), replace=False)
        var[:, :, tied, k, :] = 2.0
    lik = rng.random((1, V, T, K, 1))
    ema = MarkerArray(np.concatenate([xy, var, lik], axis=-1), data_fields=['x', 'y', 'var_x', 'var_y', 'likelihood'])
    mask, centered, good, means = center_predictions(ema, 50)
    counts = mask.sum(axis=0)
    assert len(set(counts.tolist())) > 1               # the scenario is ragged
    assert good.array.shape[2] == counts.min()
    assert not mask[0].any()
    assert centered.array.shape[:4] == (1, V, T, K)


# ------------------------------------------------------------------------------------------
# compute_pca / compute_mahalanobis (reference tests/test_stats.py)
# ------------------------------------------------------------------------------------------
def test_compute_pca_fits_per_keypoint_or_reuses_the_given_model():
    from sklearn.decomposition import PCA
    rng = np.random.default_rng(0)
    T, K, V = 30, 4, 2
    mask = np.ones((T, K), dtype=bool)
    mask[::3, 1] = False
    centered = MarkerArray(rng.normal(size=(1, V, T, K, 2)), data_fields=['x', 'y'])
    good = MarkerArray(rng.normal(size=(1, V, T, K, 2)), data_fields=['x', 'y'])
    models, pcs = compute_pca(mask, centered, good, n_components=3)
    assert len(models) == len(pcs) == K and all(isinstance(m, PCA) for m in models)
    assert [p.shape for p in pcs] == [(int(mask[:, k].sum()), 3) for k in range(K)]
    given = PCA(n_components=3).fit(rng.normal(size=(T, 2 * V)))
    models2, pcs2 = compute_pca(mask, centered, good, pca_object=given)
    assert all(m is given for m in models2)
    assert all(np.array_equal(m.components_, given.components_) for m in models2)
    assert [p.shape for p in pcs2] == [p.shape for p in pcs]


def test_compute_mahalanobis_shapes_and_singular_input():
    rng = np.random.default_rng(42)
    N, C = 40, 2
    x = rng.normal(size=(N, 2 * C))
    v = np.abs(rng.normal(size=(N, 2 * C))) + 0.1
    out = compute_mahalanobis(x, v, n_latent=2)
    assert set(out) == {'mahalanobis', 'posterior_variance', 'reconstructed'}
    assert out['reconstructed'].shape == (N, 2 * C)
    for c in range(C):
        assert out['mahalanobis'][c].shape == (N, 1) and (out['mahalanobis'][c] >= 0).all()
        assert out['posterior_variance'][c].shape == (N, 2, 2)
    # all-zero variances: singular with epsilon = 0 (a RuntimeWarning, like upstream), fine with the default
    x5, v0 = rng.normal(size=(5, 2 * C)), np.zeros((5, 2 * C))
    with pytest.warns(RuntimeWarning):
        compute_mahalanobis(x5, v0, epsilon=0, v_quantile_threshold=None)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        ok = compute_mahalanobis(x5, v0, epsilon=1e-6, v_quantile_threshold=None)
    assert np.isfinite(ok['reconstructed']).all()


# ------------------------------------------------------------------------------------------
# pupil geometry helpers (reference tests/test_ibl_pupil_smoother.py)
# ------------------------------------------------------------------------------------------
_PUPIL_KEYS = [f'pupil_{p}_r_{c}' for p in ('top', 'bottom', 'left', 'right') for c in ('x', 'y')]


def test_pupil_location_and_diameter_shapes_and_nans():
    from eks_amd.ibl_pupil_smoother import get_pupil_diameter, get_pupil_location
    rng = np.random.default_rng(0)
    dlc = {k: rng.random(10) for k in _PUPIL_KEYS}
    loc, diam = get_pupil_location(dlc), get_pupil_diameter(dlc)
    assert loc.shape == (10, 2) and diam.shape == (10,)
    assert np.isfinite(loc).all() and np.isfinite(diam).all()
    # a perfect circle of radius r centred at (cx, cy): the centre and 2 r come back
    cx, cy, r = 40.0, 25.0, 6.0
    circle = {'pupil_top_r_x': np.full(4, cx), 'pupil_top_r_y': np.full(4, cy - r),
              'pupil_bottom_r_x': np.full(4, cx), 'pupil_bottom_r_y': np.full(4, cy + r),
              'pupil_left_r_x': np.full(4, cx - r), 'pupil_left_r_y': np.full(4, cy),
              'pupil_right_r_x': np.full(4, cx + r), 'pupil_right_r_y': np.full(4, cy)}
    np.testing.assert_allclose(get_pupil_location(circle), np.tile([cx, cy], (4, 1)))
    np.testing.assert_allclose(get_pupil_diameter(circle), 2 * r)
    nan_dlc = {k: np.full(10, np.nan) for k in _PUPIL_KEYS}
    assert np.isnan(get_pupil_diameter(nan_dlc)).all()


def test_add_mean_to_array_regular_empty_and_single_row():
    from eks_amd.ibl_pupil_smoother import add_mean_to_array
    rng = np.random.default_rng(1)
    arr, keys = rng.normal(size=(10, 4)), ['key1_x', 'key2_y', 'key3_x', 'key4_y']
    out = add_mean_to_array(arr, keys, 2.0, 3.0)
    assert list(out) == keys
    for i, k in enumerate(keys):
        np.testing.assert_allclose(out[k], arr[:, i] + (2.0 if 'x' in k else 3.0))
    assert add_mean_to_array(np.zeros((0, 0)), [], 2.0, 3.0) == {}
    one = add_mean_to_array(np.array([[1.0, 2.0]]), ['a_x', 'a_y'], 2.0, 3.0)
    np.testing.assert_allclose(one['a_x'], [3.0])
    np.testing.assert_allclose(one['a_y'], [5.0])


# ------------------------------------------------------------------------------------------
# paw camera-mean helpers (reference tests/test_ibl_paw_multicam_smoother.py)
# ------------------------------------------------------------------------------------------
def test_paw_camera_mean_helpers_shift_columns_and_invert_each_other():
    from eks_amd.ibl_paw_multicam_smoother import add_camera_means, pca, remove_camera_means
    rng = np.random.default_rng(0)
    stacks = [rng.normal(size=(10, 2)) for _ in range(3)]
    before = [s.copy() for s in stacks]
    means = [3.0, 7.0]
    removed = remove_camera_means(stacks, means)
    for r, b in zip(removed, before):
        np.testing.assert_allclose(r[:, 0], b[:, 0] - 3.0)
        np.testing.assert_allclose(r[:, 1], b[:, 1] - 7.0)
    assert all(r is s for r, s in zip(removed, stacks))      # upstream shifts the caller's arrays in place
    restored = add_camera_means(removed, means)
    for r, b in zip(restored, before):
        np.testing.assert_allclose(r, b)
    model, ratio = pca(rng.normal(size=(50, 4)), 2)
    assert model.components_.shape == (2, 4) and ratio.shape == (2,) and 0 < ratio.sum() <= 1
