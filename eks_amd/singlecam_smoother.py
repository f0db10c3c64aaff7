"""Single-camera ensemble Kalman smoother (mirror of the reference's eks/singlecam_smoother.py).

    fit_eks_singlecam(input_source, save_file, bodypart_list, smooth_param, s_frames, blocks,
                      avg_mode, var_mode) -> (df_smoothed, s_finals, input_dfs, bodypart_list)
    ensemble_kalman_smoother_singlecam(marker_array, keypoint_names, smooth_param, s_frames,
                      blocks, avg_mode, var_mode) -> (DataFrame, s_finals)
"""
from __future__ import annotations

import logging
import os
from typing import Literal

import numpy as np
import pandas as pd

from .core import ensemble, run_kalman_smoother
from .marker_array import MarkerArray, input_dfs_to_markerArray
from .utils import center_predictions, format_data, make_dlc_pandas_index, write_prediction_csv

__all__ = ['fit_eks_singlecam', 'ensemble_kalman_smoother_singlecam']

logger = logging.getLogger(__name__)

OUTPUT_LABELS = ['x', 'y', 'likelihood', 'x_ens_median', 'y_ens_median', 'x_ens_var', 'y_ens_var',
                 'x_posterior_var', 'y_posterior_var']


def fit_eks_singlecam(input_source, save_file: str, bodypart_list: list | None = None,
                      smooth_param: float | list | None = None, s_frames: list | None = None,
                      blocks: list = [], avg_mode: Literal['mean', 'median'] = 'median',
                      var_mode: Literal['var', 'confidence_weighted_var'] = 'confidence_weighted_var',
                      ) -> tuple:
    """CSV files -> smoothed DataFrame -> CSV (reference eks/singlecam_smoother.py:23-102)."""
    input_dfs, keypoint_names = format_data(input_source)
    if bodypart_list is None:
        bodypart_list = keypoint_names
        logger.info(f'input data loaded for keypoints:\n{bodypart_list}')
    marker_array = input_dfs_to_markerArray([input_dfs], bodypart_list, [''])
    df, s_finals = ensemble_kalman_smoother_singlecam(
        marker_array=marker_array, keypoint_names=bodypart_list, smooth_param=smooth_param,
        s_frames=s_frames, blocks=blocks, avg_mode=avg_mode, var_mode=var_mode)
    os.makedirs(os.path.dirname(save_file), exist_ok=True)
    write_prediction_csv(df, save_file)
    logger.info('dataframes successfully converted to CSV')
    return df, s_finals, input_dfs, bodypart_list


def ensemble_kalman_smoother_singlecam(marker_array: MarkerArray, keypoint_names: list,
                                       smooth_param: float | list | None = None,
                                       s_frames: list | None = None, blocks: list = [],
                                       avg_mode: Literal['mean', 'median'] = 'median',
                                       var_mode: Literal['var', 'confidence_weighted_var'] =
                                       'confidence_weighted_var', **kalman_kwargs) -> tuple:
    """Ensemble -> centre -> Kalman filter / RTS smoother -> 9-label DataFrame
    (reference eks/singlecam_smoother.py:105-243).  `kalman_kwargs` are forwarded to
    run_kalman_smoother (e.g. s_mode='grid')."""
    M, V, T, K, _ = marker_array.shape
    if V == 1 and not os.environ.get('EKS_HOST_DRIVER'):
        out = _singlecam_on_device(marker_array, smooth_param, s_frames, blocks, avg_mode, var_mode,
                                   kalman_kwargs)
        if out is not None:
            table, s_finals = out
            return pd.DataFrame(table.reshape(T, K * 9),
                                columns=make_dlc_pandas_index(keypoint_names, labels=OUTPUT_LABELS)), s_finals
    ens = ensemble(marker_array, avg_mode=avg_mode, var_mode=var_mode)       # (1,1,T,K,5) float32
    _, centered, _, means = center_predictions(ens, quantile_keep_pca=100)
    stats = np.asarray(ens.array)[0, 0]                                       # (T,K,5)
    cen = np.asarray(centered.array)[0, 0]                                    # (T,K,2)
    m0s, S0s, As, Qs, Cs = initialize_kalman_filter(centered)
    s_finals, ms, Vs = run_kalman_smoother(
        ys=np.swapaxes(cen, 0, 1), m0s=m0s, S0s=S0s, As=As, Cs=Cs, Qs=Qs,
        ensemble_vars=stats[:, :, 2:4], s_frames=s_frames, smooth_param=smooth_param,
        blocks=blocks, vs_diag=True, **kalman_kwargs)
    # C = I on this path: observation-space mean / variance are the state's
    # (reference :189-217 computes C m + mean and diag(C V C'))
    mu = np.asarray(means.array)[0, 0, 0]                                     # (K,2)
    out = np.empty((T, K, 9))
    out[:, :, 0:2] = np.swapaxes(ms, 0, 1) + mu[None]
    out[:, :, 2] = stats[:, :, 4]
    out[:, :, 3:5] = stats[:, :, 0:2]
    out[:, :, 5:7] = stats[:, :, 2:4]
    out[:, :, 7:9] = np.swapaxes(Vs, 0, 1)
    df = pd.DataFrame(out.reshape(T, K * 9), columns=make_dlc_pandas_index(keypoint_names,
                                                                           labels=OUTPUT_LABELS))
    return df, s_finals


def _singlecam_on_device(marker_array, smooth_param, s_frames, blocks, avg_mode, var_mode,
                         kalman_kwargs):
    """The same pipeline with every array resident on the device between the upload of the markers
    and the download of the finished (T, K, 9) table (the
    reference's per-keypoint assembly loop, singlecam_smoother.py:183-241, as one gather):
    ensemble kernel -> centring and prior variances (float64 reductions over frames) ->
    run_kalman_smoother on device tensors -> table.  Returns None when the ensemble variances
    contain NaN (the percentile branch of center_predictions applies; the host path handles it)."""
    import torch

    from . import hip_ops
    from .core import _to_host
    dev = hip_ops.require_gpu()
    fields = list(marker_array.data_fields)
    arr = np.asarray(marker_array.array)
    if fields != ['x', 'y', 'likelihood']:
        arr = arr[..., [fields.index(f) for f in ('x', 'y', 'likelihood')]]
    # upload in the caller's dtype, convert on the device (a host-side float32 copy of a large
    # float64 ensemble costs more than the transfer)
    mk = torch.as_tensor(np.ascontiguousarray(arr), device=dev).to(torch.float32)
    stats = hip_ops.ensemble(mk, avg_mode, var_mode, 1000.0)[0]               # (T,K,5) float32
    del mk
    if bool(torch.isnan(stats[..., 2:4]).any()):
        return None
    T, K = stats.shape[0], stats.shape[1]
    xy = stats[..., 0:2].double()
    mu = xy.mean(dim=0)                                                        # (K,2)
    cen = xy - mu
    v = cen.var(dim=0, unbiased=False).cpu().numpy()                           # (K,2)
    eye = np.tile(np.eye(2), (K, 1, 1))
    s_finals, ms, Vs = run_kalman_smoother(
        ys=cen.to(torch.float32).transpose(0, 1), m0s=np.zeros((K, 2)), S0s=eye * v[:, :, None],
        As=eye, Cs=eye.copy(), Qs=eye.copy(), ensemble_vars=stats[..., 2:4].contiguous(),
        s_frames=s_frames, smooth_param=smooth_param, blocks=blocks, vs_diag=True,
        return_device=True, **kalman_kwargs)
    table = torch.empty((T, K, 9), dtype=torch.float64, device=dev)
    table[..., 0:2] = ms.transpose(0, 1).double() + mu
    table[..., 2] = stats[..., 4]
    table[..., 3:7] = stats[..., 0:4]
    table[..., 7:9] = Vs.transpose(0, 1)
    return _to_host(table)[0], s_finals


def initialize_kalman_filter(emA_centered_preds: MarkerArray) -> tuple:
    """m0 = 0, S0 = diag(nanvar_t x, nanvar_t y), A = C = Q = I
    (reference eks/singlecam_smoother.py:246-284).  Returns (m0s, S0s, As, Qs, Cs)."""
    cen = np.asarray(emA_centered_preds.slice_fields('x', 'y').array)[0, 0]   # (T,K,2)
    K = cen.shape[1]
    eye = np.tile(np.eye(2), (K, 1, 1))
    v = np.nanvar(cen.astype(np.float64), axis=0)                             # (K,2)
    return np.zeros((K, 2)), eye * v[:, :, None], eye.copy(), eye.copy(), eye.copy()
