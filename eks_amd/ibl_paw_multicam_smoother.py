"""IBL two-camera paw smoother (mirror of the reference's eks/ibl_paw_multicam_smoother.py): the
right camera's markers are interpolated onto the left camera's timestamps, flipped horizontally
into the left camera's frame, and both views of each paw go through the linear multi-camera
smoother (`ensemble_kalman_smoother_multicam`, device-resident pipeline).

    fit_eks_multicam_ibl_paw(input_source, save_dir, smooth_param, s_frames, quantile_keep_pca,
                             avg_mode, var_mode, img_width, inflate_vars, n_latent)
        -> (camera_dfs, s_finals, input_dfs_list, bodypart_list)

Only the file handling lives here (it is IO either side of the path); the interpolation is one
vectorised numpy.interp per column instead of the reference's per-timestamp Python loop
(:178-207).
"""
from __future__ import annotations

import os
from typing import Literal

import numpy as np
import pandas as pd

from .marker_array import MarkerArray, input_dfs_to_markerArray
from .multicam_smoother import ensemble_kalman_smoother_multicam
from .utils import convert_lp_dlc, write_prediction_csv

__all__ = ['fit_eks_multicam_ibl_paw', 'remove_camera_means', 'add_camera_means', 'pca']

BODYPARTS = ['paw_l', 'paw_r']          # the only keypoints this smoother knows (reference :138)
CAMERAS = ['left', 'right']
_KEYS = ['paw_l_x', 'paw_l_y', 'paw_r_x', 'paw_r_y']


def _shift_camera_columns(ensemble_stacks, camera_means, sign):
    out = list(ensemble_stacks)
    for stack in out:
        for cam, mean in enumerate(camera_means):
            stack[:, cam] = stack[:, cam] + sign * mean
    return out


def remove_camera_means(ensemble_stacks: list, camera_means) -> list:
    """Subtract camera c's mean from column c of every keypoint's (T, n_columns) stack (reference
    eks/ibl_paw_multicam_smoother.py:21-39).  As upstream, only the LIST is copied: the arrays are
    shifted in place and the same arrays are returned."""
    return _shift_camera_columns(ensemble_stacks, camera_means, -1.0)


def add_camera_means(ensemble_stacks: list, camera_means) -> list:
    """Inverse of remove_camera_means (reference :42-60), same in-place semantics."""
    return _shift_camera_columns(ensemble_stacks, camera_means, 1.0)


def pca(S: np.ndarray, n_comps: int) -> tuple:
    """(fitted sklearn PCA with n_comps components, its explained-variance ratios) (reference :63-76)."""
    from sklearn.decomposition import PCA
    model = PCA(n_components=n_comps).fit(S)
    return model, model.explained_variance_ratio_


def fit_eks_multicam_ibl_paw(input_source: str, save_dir: str, smooth_param: float | list | None = None,
                             s_frames: list | None = None, quantile_keep_pca: float = 50.0,
                             avg_mode: Literal['mean', 'median'] = 'median',
                             var_mode: Literal['var', 'confidence_weighted_var'] = 'confidence_weighted_var',
                             img_width: int = 128, inflate_vars: bool = False, n_latent: int = 3) -> tuple:
    """Directory of `*left*` / `*right*` prediction CSVs (one per ensemble member and camera) plus
    one `*timestamps*left*.npy` and one `*timestamps*right*.npy` -> per-camera smoothed DataFrames,
    written to `save_dir/multicam_{left,right}_results.csv` (reference :79-256)."""
    left, right = [], []
    ts_left = ts_right = None
    for filename in os.listdir(input_source):                # directory order, like the reference
        path = os.path.join(input_source, filename)
        if 'timestamps' in filename:
            if 'left' in filename:
                ts_left = np.load(path)
            else:
                ts_right = np.load(path)
            continue
        df = convert_lp_dlc(pd.read_csv(path, header=[0, 1, 2], index_col=0), BODYPARTS)
        if 'left' in filename:
            left.append(df)
        else:
            # the right camera sees the animal mirrored: its paw_l is the left camera's paw_r
            swap = {'paw_l_x': 'paw_r_x', 'paw_l_y': 'paw_r_y', 'paw_l_likelihood': 'paw_r_likelihood',
                    'paw_r_x': 'paw_l_x', 'paw_r_y': 'paw_l_y', 'paw_r_likelihood': 'paw_l_likelihood'}
            right.append(df.rename(columns=swap).loc[:, list(swap.keys())])
    if ts_left is None or ts_right is None:
        raise ValueError('Need timestamps for both cameras')
    if len(right) != len(left) or len(left) == 0:
        raise ValueError('Need same number of left and right camera models and >=1 model for each.')
    # left frames whose timestamp lies inside the right camera's recording (reference :190-196)
    keep = (ts_left >= ts_right[0]) & (ts_left <= ts_right[-1])
    t_keep = ts_left[keep]
    # scipy's interp1d (assume_sorted=False, what upstream calls) sorts the abscissa first; np.interp
    # silently assumes it is increasing - dropped / reordered frames would interpolate garbage
    order = np.argsort(ts_right, kind='stable')
    ts_sorted = np.asarray(ts_right)[order]
    per_cam = [[], []]
    for df_l, df_r in zip(left, right):
        lv = df_l.to_numpy()[:, [0, 1, 3, 4]][keep]
        rv_all = df_r.to_numpy()[:, [0, 1, 3, 4]]
        rv = np.stack([np.interp(t_keep, ts_sorted, rv_all[order, j]) for j in range(4)], axis=1)
        rv[:, 0] = img_width - rv[:, 0]                      # flip x into the left camera's frame
        rv[:, 2] = img_width - rv[:, 2]
        per_cam[0].append(pd.DataFrame(lv, columns=_KEYS))
        per_cam[1].append(pd.DataFrame(rv, columns=_KEYS))
    marker_array = input_dfs_to_markerArray(per_cam, BODYPARTS, CAMERAS, data_fields=['x', 'y'])
    # the interpolated markers carry no likelihood: a zero field, as upstream (:226-231)
    lik_shape = list(marker_array.shape)
    lik_shape[-1] = 1
    marker_array = MarkerArray.stack_fields(
        marker_array, MarkerArray(shape=tuple(lik_shape), data_fields=['likelihood']))
    camera_dfs, s_finals, _ = ensemble_kalman_smoother_multicam(
        marker_array=marker_array, keypoint_names=BODYPARTS, smooth_param=smooth_param,
        quantile_keep_pca=quantile_keep_pca, camera_names=CAMERAS, s_frames=s_frames, avg_mode=avg_mode,
        var_mode=var_mode, inflate_vars=inflate_vars, n_latent=n_latent,
        inflate_vars_kwargs={'likelihoods': None})
    os.makedirs(save_dir, exist_ok=True)
    for cam, df in zip(CAMERAS, camera_dfs):
        write_prediction_csv(df, os.path.join(save_dir, f'multicam_{cam}_results.csv'))
    return camera_dfs, s_finals, per_cam, BODYPARTS
