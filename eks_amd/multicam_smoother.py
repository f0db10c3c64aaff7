"""Multi-camera ensemble Kalman smoother - mirror of the reference's eks/multicam_smoother.py:
the linear (PCA-subspace) path and the calibrated nonlinear path (`calibration` / `camgroup`).

    fit_eks_mirrored_multicam(...) -> (final_df, s_finals, input_dfs, bodypart_list)
    fit_eks_multicam(...)          -> (camera_dfs, s_finals, input_dfs, bodypart_list, df_3d)
    ensemble_kalman_smoother_multicam(...) -> (camera_dfs, s_finals, df_3d)
    initialize_kalman_filter_pca(good_pcs_list, ensemble_pca, n_latent)

    mA_compute_maha / inflate_variance: Mahalanobis variance inflation (reference :653-764),
        host-side like upstream but vectorised over frames

    initialize_kalman_filter_geometric(ys_3d)  (calibrated path; camera helpers in calibration.py)
"""
from __future__ import annotations

import logging
import os
import time
from typing import Literal

import numpy as np
import pandas as pd

from .calibration import (CameraGroup, make_projection_from_camgroup, project_3d_covariance_to_2d,
                          triangulate_3d_models)
from .core import ensemble, run_kalman_smoother
from .marker_array import (MarkerArray, input_dfs_to_markerArray, mA_to_stacked_array,
                           stacked_array_to_mA)
from .stats import (compute_mahalanobis, compute_pca, factor_analysis_from_moments, pca_from_moments,
                    pca_sign_rule)
from .utils import center_predictions, format_data, make_dlc_pandas_index, write_prediction_csv

__all__ = ['fit_eks_mirrored_multicam', 'fit_eks_multicam', 'ensemble_kalman_smoother_multicam']

logger = logging.getLogger(__name__)

OUTPUT_LABELS = ['x', 'y', 'likelihood', 'x_ens_median', 'y_ens_median', 'x_ens_var', 'y_ens_var',
                 'x_posterior_var', 'y_posterior_var']


def fit_eks_mirrored_multicam(input_source, save_file: str, bodypart_list: list | None = None,
                              smooth_param: float | list | None = None,
                              s_frames: list | None = None, camera_names: list = [],
                              quantile_keep_pca: float = 50.0,
                              avg_mode: Literal['mean', 'median'] = 'median',
                              var_mode: Literal['var', 'confidence_weighted_var'] =
                              'confidence_weighted_var', inflate_vars: bool = False,
                              n_latent: int = 3) -> tuple:
    """Mirrored data: every CSV holds all views, columns '{bodypart}_{camera}_{coord}'
    (reference eks/multicam_smoother.py:37-153)."""
    input_dfs, keypoint_names = format_data(input_source)
    if bodypart_list is None:
        bodypart_list = list(dict.fromkeys(name.split('_')[0] for name in keypoint_names))
    per_cam = []
    for cam in camera_names:
        tag = f'_{cam}_'
        dfs = []
        for df in input_dfs:
            cols = [c for c in df.columns if tag in c]
            dfs.append(df[cols].rename(columns={c: c.replace(f'_{cam}', '') for c in cols}))
        per_cam.append(dfs)
    marker_array = input_dfs_to_markerArray(per_cam, bodypart_list, camera_names)
    camera_dfs, s_finals, _ = ensemble_kalman_smoother_multicam(
        marker_array=marker_array, keypoint_names=bodypart_list, smooth_param=smooth_param,
        quantile_keep_pca=quantile_keep_pca, camera_names=camera_names, s_frames=s_frames,
        avg_mode=avg_mode, var_mode=var_mode, inflate_vars=inflate_vars, n_latent=n_latent)
    for cam, df in zip(camera_names, camera_dfs):
        df.columns = pd.MultiIndex.from_tuples(
            [(scorer, f'{kp}_{cam}', coord) for scorer, kp, coord in df.columns],
            names=df.columns.names)
    final_df = pd.concat(camera_dfs, axis=1) if len(camera_dfs) > 1 else camera_dfs[0]
    os.makedirs(os.path.dirname(save_file), exist_ok=True)
    write_prediction_csv(final_df, f'{save_file}')
    return final_df, s_finals, input_dfs, bodypart_list


def fit_eks_multicam(input_source, save_dir: str, bodypart_list: list | None = None,
                     smooth_param: float | list | None = None, s_frames: list | None = None,
                     camera_names: list | None = None, quantile_keep_pca: float = 50.0,
                     avg_mode: Literal['mean', 'median'] = 'median',
                     var_mode: Literal['var', 'confidence_weighted_var'] = 'confidence_weighted_var',
                     inflate_vars: bool = False, n_latent: int = 3, calibration: str | None = None,
                     save_3d_outputs: bool = True) -> tuple:
    """One set of CSVs per camera (reference eks/multicam_smoother.py:156-276)."""
    camgroup = None
    if calibration is not None:
        camgroup = CameraGroup.load(calibration)
        if camera_names is not None:
            logger.warning('camera_names argument is ignored when calibration is provided; '
                           'camera names will be read from the calibration file')
        camera_names = [cam.name for cam in camgroup.cameras]
    elif camera_names is None:
        raise ValueError('camera_names must be provided when no calibration file is given')
    input_dfs, keypoint_names = format_data(input_source, camera_names=camera_names)
    if bodypart_list is None:
        bodypart_list = keypoint_names
    marker_array = input_dfs_to_markerArray(input_dfs, bodypart_list, camera_names)
    camera_dfs, s_finals, df_3d = ensemble_kalman_smoother_multicam(
        marker_array=marker_array, keypoint_names=bodypart_list, smooth_param=smooth_param,
        quantile_keep_pca=quantile_keep_pca, camera_names=camera_names, s_frames=s_frames,
        avg_mode=avg_mode, var_mode=var_mode, inflate_vars=inflate_vars, n_latent=n_latent,
        camgroup=camgroup)
    os.makedirs(save_dir, exist_ok=True)
    for cam, df in zip(camera_names, camera_dfs):
        write_prediction_csv(df, os.path.join(save_dir, f'multicam_{cam}_results.csv'))
    if save_3d_outputs and calibration is not None:
        write_prediction_csv(df_3d, os.path.join(save_dir, 'multicam_3d_results.csv'))
    return camera_dfs, s_finals, input_dfs, bodypart_list, df_3d


def ensemble_kalman_smoother_multicam(marker_array: MarkerArray, keypoint_names: list,
                                      camera_names: list, smooth_param: float | list | None = None,
                                      quantile_keep_pca: float = 50.0, s_frames: list | None = None,
                                      avg_mode: Literal['mean', 'median'] = 'median',
                                      var_mode: Literal['var', 'confidence_weighted_var'] =
                                      'confidence_weighted_var', inflate_vars: bool = False,
                                      inflate_vars_kwargs: dict = {}, pca_object=None,
                                      n_latent: int = 3, camgroup=None, **kalman_kwargs) -> tuple:
    """Ensemble -> centre -> per-keypoint PCA subspace (D = n_latent, O = 2V, C = components^T)
    -> Kalman filter / RTS smoother -> per-camera reprojection (reference
    eks/multicam_smoother.py:279-551, linear branch)."""
    if camera_names is None or len(camera_names) == 0:
        raise ValueError('camera_names must be provided')
    M, V, T, K, _ = marker_array.shape
    t_all = time.perf_counter()
    if camgroup is None and not os.environ.get('EKS_HOST_DRIVER') \
            and _device_pipeline_covers(V, n_latent, inflate_vars, kalman_kwargs):
        out = _linear_on_device(marker_array, keypoint_names, smooth_param, quantile_keep_pca, s_frames,
                                avg_mode, var_mode, inflate_vars, inflate_vars_kwargs, pca_object, n_latent,
                                kalman_kwargs)
        logger.debug(f'[profile] ensemble_kalman_smoother_multicam total: '
                     f'{time.perf_counter() - t_all:.3f}s')
        return out
    ens = ensemble(marker_array, avg_mode=avg_mode, var_mode=var_mode)       # (1,V,T,K,5)
    valid_mask, centered, good_centered, means = center_predictions(ens, quantile_keep_pca)
    vars_ma = ens.slice_fields('var_x', 'var_y')
    if inflate_vars:
        if inflate_vars_kwargs.get('mean', None) is not None:
            # the predictions are centred, so a supplied mean becomes zero (reference :355-357)
            inflate_vars_kwargs['mean'] = np.zeros_like(inflate_vars_kwargs['mean'])
        vars_ma = mA_compute_maha(centered, vars_ma, ens.slice_fields('likelihood'), n_latent,
                                  inflate_vars_kwargs=inflate_vars_kwargs)
    if camgroup is not None:
        return _calibrated_branch(marker_array, keypoint_names, camera_names, camgroup, ens, vars_ma,
                                  smooth_param, s_frames, kalman_kwargs, t_all)
    pcas, good_pcs = compute_pca(valid_mask, centered, good_centered, n_components=n_latent,
                                 pca_object=pca_object)
    m0s, S0s, As, Qs, Cs = initialize_kalman_filter_pca(good_pcs, pcas, n_latent)
    ys = np.stack([mA_to_stacked_array(centered, k) for k in range(K)])      # (K,T,2V)
    evs = np.stack([mA_to_stacked_array(vars_ma, k) for k in range(K)])      # (K,T,2V)
    t0 = time.perf_counter()
    s_finals, ms, Vs = run_kalman_smoother(
        ys=ys, m0s=m0s, S0s=S0s, As=As, Qs=Qs, Cs=Cs, ensemble_vars=np.swapaxes(evs, 0, 1),
        s_frames=s_frames, smooth_param=smooth_param, **kalman_kwargs)
    logger.debug(f'[profile] run_kalman_smoother (total): {time.perf_counter() - t0:.3f}s')

    # reprojection: y = C m + mean; posterior variance = diag(C V C') + ensemble variance
    # (the + ensemble variance is the reference's multicam convention, :509-510)
    ym = np.einsum('kod,ktd->tko', Cs, ms)                                   # (T,K,2V)
    yv = np.einsum('kod,ktde,koe->tko', Cs, Vs, Cs) + np.swapaxes(evs, 0, 1)
    stats = np.asarray(ens.array)[0]                                         # (V,T,K,5)
    mu = np.asarray(means.array)[0, :, 0]                                    # (V,K,2)
    index = make_dlc_pandas_index(keypoint_names, labels=OUTPUT_LABELS)
    camera_dfs = []
    for c in range(V):
        out = np.empty((T, K, 9))
        out[:, :, 0:2] = ym[:, :, 2 * c:2 * c + 2] + mu[c][None]
        out[:, :, 2] = stats[c, :, :, 4]
        out[:, :, 3:5] = stats[c, :, :, 0:2]
        # x/y_ens_var carry the (possibly inflated) variances on the linear path (reference :505-508)
        out[:, :, 5:7] = np.swapaxes(evs[:, :, 2 * c:2 * c + 2], 0, 1)
        out[:, :, 7:9] = yv[:, :, 2 * c:2 * c + 2]
        camera_dfs.append(pd.DataFrame(out.reshape(T, K * 9), columns=index))
    # latent states and their posterior variances (reference :529-544; the labels say x/y/z but
    # the state is the n_latent PCA subspace)
    lat = np.concatenate([np.swapaxes(ms, 0, 1),
                          np.swapaxes(np.diagonal(Vs, axis1=2, axis2=3), 0, 1)], axis=2)
    base = ['x', 'y', 'z'] if n_latent == 3 else [f'latent{i}' for i in range(n_latent)]
    labels_3d = base + [f'{b}_posterior_var' for b in base]
    df_3d = pd.DataFrame(lat.reshape(T, K * 2 * n_latent),
                         columns=make_dlc_pandas_index(keypoint_names, labels=labels_3d))
    logger.debug(f'[profile] ensemble_kalman_smoother_multicam total: '
                 f'{time.perf_counter() - t_all:.3f}s')
    return camera_dfs, s_finals, df_3d


def _device_pipeline_covers(V: int, n_latent: int, inflate_vars: bool, kalman_kwargs: dict) -> bool:
    """Shapes / options the device-resident pipeline handles; anything else takes the host pipeline (same
    kernels for the Kalman path, NumPy around them): eks_maha_inflate covers 2..8 views and n_latent <= 6,
    eks_multicam_tables wants the full (T,K,D,D) covariances on the device."""
    if kalman_kwargs.get('vs_diag') or 'return_device' in kalman_kwargs:
        return False
    if inflate_vars and not (2 <= V <= 8 and 1 <= n_latent <= 6):
        return False
    return 1 <= n_latent <= 6


def _linear_on_device(marker_array, keypoint_names, smooth_param, quantile_keep_pca, s_frames, avg_mode,
                      var_mode, inflate_vars, inflate_vars_kwargs, pca_object, n_latent, kalman_kwargs):
    """The linear multi-camera pipeline with every (T, ...) array resident on the device between
    the upload of the markers and the download of the finished per-camera tables (SURVEY.md
    section 8(f) ranks 2 and 4; reference eks/multicam_smoother.py:335-348, :409-443, :481-551):

        eks_ensemble -> worst variance per (frame, keypoint) -> percentile mask and good-frame
        indices (utils.py:318-343; the K thresholds are numpy.percentile's, bit for bit, from two order
        statistics per keypoint selected on the device: hip_ops.percentile) -> means over the
        good frames, centring -> [eks_maha_inflate passes; the factor-analysis fits on the host]
        -> PCA fit on the host from the good-frame subset only (n_good x 2V per keypoint), prior
        variances / process noise from the device-side principal components -> run_kalman_smoother
        on device tensors -> eks_multicam_tables (reprojection + posterior variances, (V, T, K, 9))
        -> ONE download.

    Small host <-> device traffic only: 2 K order statistics, the good-frame subset for the PCA fits,
    the factor-analysis moments, K x (D + D^2) statistics."""
    import torch

    from . import hip_ops
    from .core import _to_host
    from .stats import PCA
    dev = hip_ops.require_gpu()
    M, V, T, K, _ = marker_array.shape
    fields = list(marker_array.data_fields)
    arr = np.asarray(marker_array.array)
    if fields != ['x', 'y', 'likelihood']:
        arr = arr[..., [fields.index(f) for f in ('x', 'y', 'likelihood')]]
    mk = torch.as_tensor(np.ascontiguousarray(arr), device=dev).to(torch.float32)
    stats = hip_ops.ensemble(mk, avg_mode, var_mode, 1000.0)                  # (V,T,K,5) float32
    del mk
    preds = stats[..., 0:2].double()                                          # (V,T,K,2)
    vars32 = stats[..., 2:4]                                                  # (V,T,K,2) float32
    # ---- center_predictions (reference eks/utils.py:293-365)
    if quantile_keep_pca >= 100 and not bool(torch.isnan(vars32).any()):
        mask = torch.ones((T, K), dtype=torch.bool, device=dev)
        n_good = T
        order = torch.arange(T, device=dev)[:, None].expand(T, K)
    else:
        worst = vars32.amax(dim=(0, 3)).contiguous()                          # (T,K) float32
        # numpy.percentile(worst, q, axis=0) without downloading (T, K): the two order statistics per
        # keypoint by exact selection on the device, numpy's interpolation on those 2 K floats
        thr = hip_ops.percentile(worst, quantile_keep_pca)
        mask = worst <= torch.as_tensor(thr, device=dev)
        n_good = int(mask.sum(dim=0).min().item())
        # first n_good kept frames of every keypoint (the reference truncates to the shortest list)
        order = torch.argsort((~mask).to(torch.uint8), dim=0, stable=True)[:n_good]   # (n_good,K)
    kk = torch.arange(K, device=dev)[None, :].expand(n_good, K)
    good = preds[:, order, kk, :]                                             # (V,n_good,K,2)
    means = good.mean(dim=1)                                                  # (V,K,2) float64
    centered = preds - means[:, None]                                         # (V,T,K,2)
    # stacked views per keypoint: (K, T, 2V), order [c0x, c0y, c1x, ...] (marker_array.py:302-324)
    ys = centered.permute(2, 1, 0, 3).reshape(K, T, 2 * V).contiguous()
    evs = vars32.permute(2, 1, 0, 3).reshape(K, T, 2 * V).contiguous()       # float32
    if inflate_vars:
        if inflate_vars_kwargs.get('mean', None) is not None:
            inflate_vars_kwargs['mean'] = np.zeros_like(inflate_vars_kwargs['mean'])
        likes = stats[..., 4].permute(2, 1, 0).contiguous() if inflate_vars_kwargs.get('likelihoods') is not None \
            else None
        evs = _inflate_on_device(ys, evs, likes, n_latent, inflate_vars_kwargs)
    # ---- PCA per keypoint on the good frames (host, n_good x 2V each); reference eks/stats.py:9-64
    good_c = (good - means[:, None]).permute(2, 1, 0, 3).reshape(K, n_good, 2 * V)
    if pca_object is None and pca_sign_rule() != 'unknown' and n_good > 2 * V and \
            not os.environ.get('EKS_HOST_PCA'):
        # PCA(n_latent).fit(rows) from the rows' mean and covariance, reduced on the device (what scikit-learn's
        # own covariance_eigh solver does for tall matrices): 2V + (2V)^2 doubles per keypoint cross the bus
        # instead of n_good x 2V, and no per-keypoint fit on the host (stats.pca_from_moments)
        pmean = good_c.mean(dim=1)                                            # (K,2V)
        xc = good_c - pmean[:, None]
        cov_h = (_gram(xc) / (n_good - 1)).cpu().numpy()

        def extreme(k):
            def fn(axes):                                                     # 'u' sign rule only (old scikit-learn)
                sc = xc[k] @ torch.as_tensor(axes.T, device=dev)              # (n_good, L)
                idx = sc.abs().argmax(dim=0)
                return sc[idx, torch.arange(sc.shape[1], device=dev)].cpu().numpy()
            return fn

        comp_h = np.stack([pca_from_moments(cov_h[k], n_latent, extreme(k)) for k in range(K)])   # (K,L,2V)
    else:
        good_host = good_c.cpu().numpy()
        pcas = [pca_object if pca_object is not None else PCA(n_components=n_latent).fit(good_host[k])
                for k in range(K)]
        comp_h = np.stack([np.asarray(p.components_) for p in pcas])
        pmean = torch.as_tensor(np.stack([np.asarray(p.mean_) for p in pcas]), device=dev)    # (K,2V)
    comp = torch.as_tensor(comp_h, device=dev)                                # (K,L,2V)
    pcs = torch.einsum('kto,klo->ktl', ys - pmean[:, None], comp)             # all frames (K,T,L)
    # initialize_kalman_filter_pca (reference :554-597): statistics over the frames in `mask`
    S0s, Qs = np.zeros((K, n_latent, n_latent)), np.zeros((K, n_latent, n_latent))
    var_d, cov_d = [], []
    for k in range(K):
        gp = pcs[k][mask[:, k]]                                               # (n_valid_k, L)
        var_d.append(gp.var(dim=0, unbiased=False))
        dd = gp[1:] - gp[:-1]
        # np.cov(d.T): unbiased covariance of the frame-to-frame differences, L x L - reduced where the rows are
        cov_d.append(_gram(dd - dd.mean(dim=0)) / (dd.shape[0] - 1) if dd.shape[0] > 1
                     else torch.full((n_latent, n_latent), float('nan'), dtype=pcs.dtype, device=dev))
    var_h, cov_h2 = _to_host(torch.stack(var_d), torch.stack(cov_d), pinned=False)
    for k in range(K):
        S0s[k] = np.diag(var_h[k])
        cov = np.atleast_2d(cov_h2[k])
        top = np.max(np.abs(cov))
        Qs[k] = cov / top if top > 0 else cov
    Cs = np.ascontiguousarray(np.swapaxes(comp_h, 1, 2))                      # (K,2V,L)
    m0s = np.zeros((K, n_latent))
    As = np.tile(np.eye(n_latent), (K, 1, 1))
    t0 = time.perf_counter()
    ev_tk = evs.transpose(0, 1).contiguous()                                  # (T,K,2V)
    s_finals, ms, Vs = run_kalman_smoother(
        ys=ys.to(torch.float32), m0s=m0s, S0s=S0s, As=As, Qs=Qs, Cs=Cs, ensemble_vars=ev_tk,
        s_frames=s_frames, smooth_param=smooth_param, return_device=True, **kalman_kwargs)
    logger.debug(f'[profile] run_kalman_smoother (total): {time.perf_counter() - t0:.3f}s')
    tables, latent = hip_ops.multicam_tables(stats, ev_tk, ms.transpose(0, 1), Vs.transpose(0, 1),
                                             torch.as_tensor(Cs, device=dev), means)
    tables_h, latent_h = _to_host(tables, latent)
    index = make_dlc_pandas_index(keypoint_names, labels=OUTPUT_LABELS)
    camera_dfs = [pd.DataFrame(tables_h[c].reshape(T, K * 9), columns=index) for c in range(V)]
    base = ['x', 'y', 'z'] if n_latent == 3 else [f'latent{i}' for i in range(n_latent)]
    labels_3d = base + [f'{b}_posterior_var' for b in base]
    df_3d = pd.DataFrame(latent_h.reshape(T, K * 2 * n_latent),
                         columns=make_dlc_pandas_index(keypoint_names, labels=labels_3d))
    return camera_dfs, s_finals, df_3d


_GRAM_TEMP_BYTES = 64 << 20


def _gram(x, max_temp_bytes: int | None = None):
    """x^T x over the second-to-last axis of a tall (..., n, F) float64 tensor with a handful of columns, as an
    elementwise product and a sum: rocBLAS picks a 64 x 64 macro-tile float64 GEMM for these shapes - 1.3 ms for
    47 500 x 3 (rocprofv3 on the configs[3] driver) where the reduction takes ~30 us.

    The (..., n, F, F) outer-product temporary is only formed while it stays under `max_temp_bytes` (64 MiB);
    larger problems (long sessions, many cameras: K * n * F^2 * 8 bytes) are reduced one column at a time and,
    where even an x-sized temporary is too large, in slabs of rows - the temporary never exceeds the cap (plus
    one row slab), so a session that fitted before the elementwise form still fits."""
    cap = _GRAM_TEMP_BYTES if max_temp_bytes is None else int(max_temp_bytes)
    n, F = x.shape[-2], x.shape[-1]
    lead = 1
    for d in x.shape[:-2]:
        lead *= int(d)
    item = x.element_size()
    if lead * n * F * F * item <= cap:
        return (x.unsqueeze(-1) * x.unsqueeze(-2)).sum(dim=-3)
    out = x.new_zeros(x.shape[:-2] + (F, F))
    rows = max(1, min(n, cap // max(1, lead * F * item)))        # rows per slab: one x-sized temporary <= cap
    for r0 in range(0, n, rows):
        xs = x[..., r0:r0 + rows, :]
        for j in range(F):
            out[..., :, j] += (xs * xs[..., j:j + 1]).sum(dim=-2)
    return out


def _inflate_on_device(ys, evs, likes, n_latent, inflate_vars_kwargs, threshold: float = 5.0,
                       scalar: float = 10.0):
    """mA_compute_maha (reference eks/multicam_smoother.py:653-721) with the per-frame work on the
    device: every pass, the keypoints still inflating get a factor-analysis fit (rows chosen as
    compute_mahalanobis does, eks/stats.py:103-118, from the CURRENT variances; the fitted rows are
    reduced to their mean and covariance on the device and sklearn's EM loop runs on that 2V x 2V
    matrix on the host, stats.factor_analysis_from_moments) and one eks_maha_inflate launch covers
    them all; a keypoint stops when a pass inflates nothing.
    ys (K,T,2V) float64, evs (K,T,2V) float32 -> inflated evs (new tensor)."""
    import torch
    from sklearn.decomposition import FactorAnalysis

    from . import hip_ops
    K, T, O = ys.shape
    dev = ys.device
    kw = inflate_vars_kwargs
    kw.setdefault('likelihood_threshold', 0.9)           # written INTO the caller's dict, like upstream
    kw.setdefault('v_quantile_threshold', 50.0)
    eps = kw.get('epsilon', 1e-6)
    v = evs.clone()
    fixed = kw.get('loading_matrix') is not None and kw.get('mean') is not None
    # sklearn's default randomized SVD is an exact SVD when its n_latent + 10 random directions
    # span all 2V columns: the fit then only needs the fitted rows' mean and covariance, which are
    # reduced on the device (stats.factor_analysis_from_moments); wider problems use sklearn itself
    moments = not fixed and O <= n_latent + 10
    x_host = None if (fixed or moments) else ys.cpu().numpy()
    rows_ok = torch.ones((K, T), dtype=torch.bool, device=dev)
    if likes is not None and kw.get('likelihood_threshold') is not None and not fixed:
        rows_ok &= likes.amin(dim=2) >= kw['likelihood_threshold']
    W = torch.zeros((K, O, n_latent), dtype=torch.float64, device=dev)
    mu = torch.zeros((K, O), dtype=torch.float64, device=dev)
    if fixed:
        W[:] = torch.as_tensor(np.ascontiguousarray(kw['loading_matrix'], dtype=np.float64), device=dev)
        mu[:] = torch.as_tensor(np.ascontiguousarray(kw['mean'], dtype=np.float64), device=dev)
    active = np.ones(K, dtype=bool)
    while active.any():
        for k in np.flatnonzero(active):
            logger.info(f'inflating keypoint: {k}')
        if not fixed:
            rows = rows_ok.clone()
            if kw.get('v_quantile_threshold') is not None:
                # eks/stats.py:109-112: frames whose worst variance lies below the keypoint's percentile;
                # numpy.percentile's value per keypoint from two device-selected order statistics
                worst = v.amax(dim=2)                                             # (K,T) float32
                thr = hip_ops.percentile(worst.transpose(0, 1).contiguous(), kw['v_quantile_threshold'])
                rows &= worst < torch.as_tensor(thr, device=dev)[:, None]
        if moments:
            act = torch.as_tensor(np.flatnonzero(active), device=dev)
            w = rows.index_select(0, act).to(torch.float64)                       # (Ka,T)
            n_rows = w.sum(dim=1)
            xa = ys.index_select(0, act)
            mean = (w[:, :, None] * xa).sum(dim=1) / n_rows[:, None]
            xc = (xa - mean[:, None, :]) * w[:, :, None]
            cov = (_gram(xc) / n_rows[:, None, None]).cpu().numpy()          # (not a GEMM: see _gram)
            n_host, mean_host = n_rows.cpu().numpy(), mean.cpu().numpy()
            for i, k in enumerate(np.flatnonzero(active)):
                if n_host[i] < 1:
                    raise ValueError(f'Found array with 0 sample(s) (shape=(0, {O})) while a minimum of 1 is '
                                     'required by FactorAnalysis.')
                Wk, _, _ = factor_analysis_from_moments(cov[i], int(n_host[i]), n_latent)
                W[k] = torch.as_tensor(np.ascontiguousarray(Wk), device=dev)
                mu[k] = torch.as_tensor(mean_host[i], device=dev)
        elif not fixed:
            rows_h = rows.cpu().numpy()
            for k in np.flatnonzero(active):
                fa = FactorAnalysis(n_components=n_latent).fit(x_host[k][rows_h[k]])
                W[k] = torch.as_tensor(np.ascontiguousarray(fa.components_.T, dtype=np.float64), device=dev)
                mu[k] = torch.as_tensor(np.ascontiguousarray(fa.mean_, dtype=np.float64), device=dev)
        n_inf, _ = hip_ops.maha_inflate(ys, v, W, mu, torch.as_tensor(active.astype(np.int32), device=dev),
                                        epsilon=eps, threshold=threshold, scalar=scalar)
        active &= n_inf.cpu().numpy() > 0
    return v


def _calibrated_branch(marker_array, keypoint_names, camera_names, camgroup, ens, vars_ma,
                       smooth_param, s_frames, kalman_kwargs, t_all) -> tuple:
    """Nonlinear path (reference eks/multicam_smoother.py:367-407, :450-480): triangulate every
    ensemble member, average -> 3-D initialisation; latent state = the 3-D point, observation =
    its projection into every calibrated camera; extended Kalman smoother (eks_ekf_smooth);
    reprojection of the smoothed points and of their covariances through the projection's
    Jacobian."""
    M, V, T, K, _ = marker_array.shape
    h_fn, h_cams = make_projection_from_camgroup(camgroup)
    t0 = time.perf_counter()
    ys_3d = triangulate_3d_models(marker_array, camgroup).mean(axis=0)      # (K,T,3)
    logger.debug(f'[profile] triangulation: {time.perf_counter() - t0:.3f}s')
    m0s, S0s, As, Qs, Cs = initialize_kalman_filter_geometric(ys_3d)
    stats = np.asarray(ens.array)[0]                                         # (V,T,K,5)
    # the UNCENTRED ensemble averages are the observations here (reference :389-397)
    ys = np.transpose(stats[..., 0:2], (2, 1, 0, 3)).reshape(K, T, 2 * V)
    evs = np.stack([mA_to_stacked_array(vars_ma, k) for k in range(K)])      # (K,T,2V), maybe inflated
    t0 = time.perf_counter()
    s_finals, ms, Vs = run_kalman_smoother(
        ys=ys, m0s=m0s, S0s=S0s, As=As, Qs=Qs, Cs=Cs, ensemble_vars=np.swapaxes(evs, 0, 1),
        s_frames=s_frames, smooth_param=smooth_param, h_fn=h_fn, x_init=ys_3d, **kalman_kwargs)
    logger.debug(f'[profile] run_kalman_smoother (total): {time.perf_counter() - t0:.3f}s')
    index = make_dlc_pandas_index(keypoint_names, labels=OUTPUT_LABELS)
    camera_dfs = []
    for c in range(V):
        out = np.empty((T, K, 9))
        for k in range(K):
            out[:, k, 0:2] = h_cams[c](ms[k])
            # like upstream, the ensemble variance added here is columns 0 and 1 of the keypoint's
            # (T, 2V) array - camera 0's - whatever the camera (reference :466-467, :949-951)
            out[:, k, 7], out[:, k, 8] = project_3d_covariance_to_2d(ms[k], Vs[k], h_cams[c], evs[k])
        out[:, :, 2] = stats[c, :, :, 4]
        out[:, :, 3:5] = stats[c, :, :, 0:2]
        out[:, :, 5:7] = stats[c, :, :, 2:4]          # the UNinflated ensemble variances (:474-477)
        camera_dfs.append(pd.DataFrame(out.reshape(T, K * 9), columns=index))
    lat = np.concatenate([np.swapaxes(ms, 0, 1),
                          np.swapaxes(np.diagonal(Vs, axis1=2, axis2=3), 0, 1)], axis=2)
    labels_3d = ['x', 'y', 'z', 'x_posterior_var', 'y_posterior_var', 'z_posterior_var']
    df_3d = pd.DataFrame(lat.reshape(T, K * 6),
                         columns=make_dlc_pandas_index(keypoint_names, labels=labels_3d))
    logger.debug(f'[profile] ensemble_kalman_smoother_multicam total: '
                 f'{time.perf_counter() - t_all:.3f}s')
    return camera_dfs, s_finals, df_3d


def initialize_kalman_filter_geometric(ys: np.ndarray) -> tuple:
    """Filter parameters of the 3-D latent from the triangulated points ys (K,T,3): m0 = mean of
    the first 10 frames, S0 = diag(var + 1e-4), A = C = I, Q = diag of the squared robust (MAD)
    scale of the frame-to-frame differences (reference eks/multicam_smoother.py:600-650).
    Returns (m0s, S0s, As, Qs, Cs)."""
    ys = np.asarray(ys, dtype=np.float64)
    K, T, D = ys.shape
    m0s = ys[:, :10].mean(axis=1)
    S0s = np.stack([np.diag(np.nanvar(ys[k], axis=0) + 1e-4) for k in range(K)])
    eye = np.tile(np.eye(D), (K, 1, 1))
    dx = np.diff(ys, axis=1)
    mad = np.median(np.abs(dx - np.median(dx, axis=1, keepdims=True)), axis=1) + 1e-12
    Qs = np.stack([np.diag(v) for v in np.maximum((1.4826 * mad) ** 2, 1e-8)])
    return m0s, S0s, eye, Qs, eye.copy()


def mA_compute_maha(centered_emA_preds: MarkerArray, emA_vars: MarkerArray, emA_likes: MarkerArray,
                    n_latent: int, inflate_vars_kwargs: dict = {}, threshold: float = 5.0,
                    scalar: float = 10.0) -> MarkerArray:
    """Per keypoint: inflate the ensemble variances of frames whose per-view Mahalanobis distance
    (factor-analysis residual) exceeds `threshold`, re-fitting until nothing is inflated
    (reference eks/multicam_smoother.py:653-721).  Like upstream, missing defaults are written
    INTO `inflate_vars_kwargs` (likelihood_threshold 0.9, v_quantile_threshold 50.0)."""
    _, V, _, K, _ = centered_emA_preds.shape
    inflate_vars_kwargs.setdefault('likelihood_threshold', 0.9)
    inflate_vars_kwargs.setdefault('v_quantile_threshold', 50.0)
    per_kp = []
    for k in range(K):
        preds = mA_to_stacked_array(centered_emA_preds, k)
        cur = mA_to_stacked_array(emA_vars, k)
        likes = mA_to_stacked_array(emA_likes, k)
        logger.info(f'inflating keypoint: {k}')
        changed = True
        while changed:
            kw = dict(inflate_vars_kwargs)
            if kw.get('likelihoods', None) is not None:
                kw['likelihoods'] = likes
            res = compute_mahalanobis(preds, cur, n_latent=n_latent, **kw)
            cur, changed = inflate_variance(cur, res['mahalanobis'], threshold, scalar)
        per_kp.append(stacked_array_to_mA(cur, V, data_fields=['var_x', 'var_y']))
    return MarkerArray.stack(per_kp, 'keypoints')


def inflate_variance(v: np.ndarray, maha_dict: dict, threshold: float = 5.0,
                     scalar: float = 10.0) -> tuple:
    """Multiply by `scalar` the variances of every (frame, view) whose Mahalanobis distance exceeds
    `threshold`; with exactly two views the whole frame is inflated if either view is
    (reference eks/multicam_smoother.py:724-764).  Returns (new variances, anything inflated?)."""
    assert len(maha_dict) >= 2, 'must have >=2 views to inflate variance'
    n_views = len(maha_dict)
    hit = np.zeros((v.shape[0], n_views), dtype=bool)
    for view, dist in maha_dict.items():
        hit[:, view] = dist[:, 0] > threshold
    mask = np.repeat(hit, 2, axis=1)
    if n_views == 2:
        mask |= mask.any(axis=1, keepdims=True)
    out = v.copy()
    out[mask] *= scalar
    return out, bool(mask.any())


def initialize_kalman_filter_pca(good_pcs_list, ensemble_pca, n_latent: int) -> tuple:
    """m0 = 0, S0 = diag(var of good-frame PCs), A = I, C = components^T (2V x n_latent),
    Q = cov(diff of PCs) / max|cov| (reference eks/multicam_smoother.py:554-597).
    Returns (m0s, S0s, As, Qs, Cs)."""
    K = len(good_pcs_list)
    m0s = np.zeros((K, n_latent))
    S0s = np.stack([np.diag(np.var(p[:, :n_latent], axis=0)) for p in good_pcs_list])
    As = np.tile(np.eye(n_latent), (K, 1, 1))
    Cs = np.stack([np.asarray(p.components_).T for p in ensemble_pca])
    Qs = []
    for pcs in good_pcs_list:
        cov = np.atleast_2d(np.cov((pcs[1:] - pcs[:-1]).T))
        top = np.max(np.abs(cov))
        Qs.append(cov / top if top > 0 else cov)
    return m0s, S0s, As, np.stack(Qs), Cs
