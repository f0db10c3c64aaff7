"""IBL pupil smoother (mirror of the reference's eks/ibl_pupil_smoother.py).

One 3-state chain (diameter, com_x, com_y) observed through eight coordinates (four pupil points),
AR(1) dynamics A = diag(s_d, s_c, s_c), Q = diag(var (1 - s^2)), time-varying R_t from the ensemble
variances in the loss AND the final pass.  The loss, its two sensitivities, the Adam step and the
final filter + RTS smoother are kernels of libeks_hip.so (eks_ar1_nll, eks_pupil_adam_step,
eks_smooth); the per-frame geometry (diameter / centre of mass from four points) is O(T) host NumPy
like upstream.

    fit_eks_pupil(input_source, save_file, smooth_params, s_frames, avg_mode, var_mode)
        -> (df_smoothed, smooth_params, input_dfs, keypoint_names)
    ensemble_kalman_smoother_ibl_pupil(marker_array, keypoint_names, smooth_params, s_frames,
        avg_mode, var_mode) -> (DataFrame, [s_diam, s_com])
    run_pupil_kalman_smoother(ys, m0, S0, C, ensemble_vars, diameters_var, x_var, y_var, ...)
        -> ([s_diam, s_com], ms (T,3), Vs (T,3,3))
"""
from __future__ import annotations

import logging
import os
import warnings
from typing import Literal

import numpy as np
import pandas as pd

from . import hip_ops
from .core import _to_numpy, _torch, ensemble
from .marker_array import MarkerArray, input_dfs_to_markerArray
from .utils import format_data, frame_spans, make_dlc_pandas_index, write_prediction_csv

__all__ = ['fit_eks_pupil', 'ensemble_kalman_smoother_ibl_pupil', 'get_pupil_location',
           'get_pupil_diameter', 'add_mean_to_array', 'run_pupil_kalman_smoother',
           'pupil_optimize_smooth']

logger = logging.getLogger(__name__)

# NOTE: this order MUST be kept (reference eks/ibl_pupil_smoother.py:166-168)
PUPIL_BODYPARTS = ['pupil_top_r', 'pupil_bottom_r', 'pupil_right_r', 'pupil_left_r']
OUTPUT_LABELS = ['x', 'y', 'likelihood', 'x_ens_median', 'y_ens_median', 'x_ens_var', 'y_ens_var',
                 'x_posterior_var', 'y_posterior_var']
# observation matrix: rows top x,y / bottom x,y / right x,y / left x,y; columns diameter, com_x,
# com_y (reference :271-276)
PUPIL_C = np.array([[0, 1, 0], [-.5, 0, 1], [0, 1, 0], [.5, 0, 1],
                    [.5, 1, 0], [0, 0, 1], [-.5, 1, 0], [0, 0, 1]], dtype=np.float64)


# --------------------------------------------------------------------------------------------
# per-frame geometry (host)
# --------------------------------------------------------------------------------------------
def _mid(a, b, tolerate_nan: bool):
    pair = np.stack([np.asarray(a), np.asarray(b)])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        return np.nanmedian(pair, axis=0) if tolerate_nan else np.median(pair, axis=0)


def get_pupil_location(dlc: dict) -> np.ndarray:
    """Pupil centre per frame, (T, 2) (reference eks/ibl_pupil_smoother.py:34-60).

    x: the midpoint of top/bottom x may lose one of the two to NaN, left/right x must both be
    present; y: top/bottom y must both be present, left/right y may lose one; the two estimates of
    each coordinate are then combined NaN-tolerantly."""
    g = {p: (np.asarray(dlc[f'pupil_{p}_r_x']), np.asarray(dlc[f'pupil_{p}_r_y']))
         for p in ('top', 'bottom', 'left', 'right')}
    cx = _mid(_mid(g['top'][0], g['bottom'][0], True), _mid(g['right'][0], g['left'][0], False), True)
    cy = _mid(_mid(g['top'][1], g['bottom'][1], False), _mid(g['right'][1], g['left'][1], True), True)
    out = np.zeros((len(cx), 2))
    out[:, 0], out[:, 1] = cx, cy
    return out


def get_pupil_diameter(dlc: dict) -> np.ndarray:
    """Pupil diameter per frame, (T,) (reference eks/ibl_pupil_smoother.py:63-91): NaN-ignoring
    median of six estimates - top-bottom, left-right, and sqrt(2) x the four adjacent-point
    distances (circle assumption)."""
    pts = {p: np.stack([np.asarray(dlc[f'pupil_{p}_r_x']), np.asarray(dlc[f'pupil_{p}_r_y'])])
           for p in ('top', 'bottom', 'left', 'right')}

    def sep(p, q):
        d = pts[p] - pts[q]
        return np.sqrt(d[0] * d[0] + d[1] * d[1])

    est = [sep('top', 'bottom'), sep('left', 'right')]
    est += [sep(p, q) * 2 ** 0.5 for p in ('top', 'bottom') for q in ('left', 'right')]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        return np.nanmedian(np.stack(est), axis=0)


def add_mean_to_array(pred_arr: np.ndarray, keys: list, mean_x, mean_y) -> dict:
    """{key: column + mean_x if the key names an x coordinate else column + mean_y}
    (reference eks/ibl_pupil_smoother.py:94-117; 'x' in key decides)."""
    return {key: pred_arr[:, i] + (mean_x if 'x' in key else mean_y) for i, key in enumerate(keys)}


# --------------------------------------------------------------------------------------------
# fit wrapper and driver
# --------------------------------------------------------------------------------------------
def fit_eks_pupil(input_source, save_file: str, smooth_params: list | None = None,
                  s_frames: list | None = None, avg_mode: Literal['mean', 'median'] = 'median',
                  var_mode: Literal['var', 'confidence_weighted_var'] = 'confidence_weighted_var',
                  ) -> tuple:
    """CSV files -> smoothed DataFrame -> CSV (reference eks/ibl_pupil_smoother.py:120-194)."""
    bodypart_list = list(PUPIL_BODYPARTS)
    input_dfs, _ = format_data(input_source)
    logger.info(f'input data loaded for keypoints: {bodypart_list}')
    marker_array = input_dfs_to_markerArray([input_dfs], bodypart_list, [''])
    df, s_finals = ensemble_kalman_smoother_ibl_pupil(
        marker_array=marker_array, keypoint_names=bodypart_list, smooth_params=smooth_params,
        s_frames=s_frames, avg_mode=avg_mode, var_mode=var_mode)
    os.makedirs(os.path.dirname(save_file), exist_ok=True)
    write_prediction_csv(df, save_file)
    logger.info('dataframes successfully converted to CSV')
    return df, s_finals, input_dfs, bodypart_list


def ensemble_kalman_smoother_ibl_pupil(marker_array: MarkerArray, keypoint_names: list,
                                       smooth_params: list | None = None,
                                       s_frames: list | None = None,
                                       avg_mode: Literal['mean', 'median'] = 'median',
                                       var_mode: Literal['var', 'confidence_weighted_var'] =
                                       'confidence_weighted_var', **opt_kwargs) -> tuple:
    """Ensemble -> diameter / centre of mass -> optimise (s_diam, s_com) -> filter + RTS ->
    9-label DataFrame (reference eks/ibl_pupil_smoother.py:197-359).  `opt_kwargs` (lr, tol,
    safety_cap) go to run_pupil_kalman_smoother.

    The table layout keeps upstream's: data are gathered in (top, right, bottom, left) order under
    a header built from `keypoint_names`; the i-th likelihood is keypoint_names[i]'s; the posterior
    variances of position i are entries (i, i) and (i+1, i+1) of C V C' (:310-349)."""
    _, _, T, K, _ = marker_array.shape
    ens = ensemble(marker_array, avg_mode=avg_mode, var_mode=var_mode)       # (1,1,T,4,5) float32
    stats = np.asarray(ens.array, dtype=np.float64)[0, 0]                     # (T,4,5) x,y,vx,vy,lik
    preds = stats[:, :, 0:2].reshape(T, -1)
    evars = stats[:, :, 2:4].reshape(T, -1)
    likes = stats[:, :, 4]
    keys = [f'{kp}_{c}' for kp in keypoint_names for c in ('x', 'y')]
    as_dict = {key: preds[:, i] for i, key in enumerate(keys)}
    diam = get_pupil_diameter(as_dict)
    loc = get_pupil_location(as_dict)
    mean_x, mean_y = np.mean(loc[:, 0]), np.mean(loc[:, 1])
    x_obs, y_obs = loc[:, 0] - mean_x, loc[:, 1] - mean_y
    m0 = np.array([np.mean(diam), 0.0, 0.0])
    S0 = np.diag([np.nanvar(diam), np.nanvar(x_obs), np.nanvar(y_obs)])
    ys = preds.copy()
    ys[:, 0::2] -= mean_x
    ys[:, 1::2] -= mean_y
    s_finals, ms, Vs = run_pupil_kalman_smoother(
        ys=ys, m0=m0, S0=S0, C=PUPIL_C, ensemble_vars=evars, diameters_var=np.var(diam),
        x_var=np.var(x_obs), y_var=np.var(y_obs), s_frames=s_frames, smooth_params=smooth_params,
        **opt_kwargs)
    logger.debug(f'diameter_s={s_finals[0]}, com_s={s_finals[1]}')
    y_m = ms @ PUPIL_C.T                                                     # (T,8)
    y_v = np.einsum('od,tde,pe->top', PUPIL_C, Vs, PUPIL_C)                  # (T,8,8)
    smoothed = add_mean_to_array(y_m, keys, mean_x, mean_y)
    out = np.empty((T, 4, 9))
    gather = [('pupil_top_r', 0), ('pupil_right_r', 4), ('pupil_bottom_r', 2), ('pupil_left_r', 6)]
    for i, (name, col) in enumerate(gather):
        out[:, i, 0] = smoothed[f'{name}_x']
        out[:, i, 1] = smoothed[f'{name}_y']
        out[:, i, 2] = likes[:, i]
        out[:, i, 3:5] = preds[:, col:col + 2]
        out[:, i, 5:7] = evars[:, col:col + 2]
        out[:, i, 7] = y_v[:, i, i]
        out[:, i, 8] = y_v[:, i + 1, i + 1]
    df = pd.DataFrame(out.reshape(T, 36),
                      columns=make_dlc_pandas_index(keypoint_names, labels=OUTPUT_LABELS))
    return df, s_finals


# --------------------------------------------------------------------------------------------
# Kalman path (device)
# --------------------------------------------------------------------------------------------
class _PupilProblem:
    """Device copies, frame-major with a chain axis of length 1: y, var (T, 1, 8) float32."""

    def __init__(self, ys, m0, S0, C, ensemble_vars, latent_vars):
        torch = _torch()
        self.dev = hip_ops.require_gpu()
        ys = np.ascontiguousarray(_to_numpy(ys), dtype=np.float32)
        ev = np.ascontiguousarray(_to_numpy(ensemble_vars), dtype=np.float32)
        if ys.ndim != 2 or ev.shape != ys.shape:
            raise ValueError(f'ys and ensemble_vars must both be (T, O); got {ys.shape}, {ev.shape}')
        self.T, self.O = ys.shape
        C = np.ascontiguousarray(_to_numpy(C, np.float64))
        self.D = C.shape[1]
        if C.shape[0] != self.O:
            raise ValueError(f'C must be ({self.O}, D), got {C.shape}')
        dev = self.dev
        self.y = torch.as_tensor(ys, device=dev).unsqueeze(1)
        self.var = torch.as_tensor(ev, device=dev).unsqueeze(1)
        self.m0 = torch.as_tensor(_to_numpy(m0, np.float64).reshape(1, self.D), device=dev)
        self.S0 = torch.as_tensor(_to_numpy(S0, np.float64).reshape(1, self.D, self.D), device=dev)
        self.C = torch.as_tensor(C[None], device=dev)
        self.latent_vars = np.asarray(latent_vars, dtype=np.float64).reshape(self.D)

    def cropped(self, s_frames):
        """(y, var) on the s_frames spans - the loss only (reference :510-518)."""
        if not s_frames or (len(s_frames) == 1 and s_frames[0] == (None, None)):
            return self.y, self.var
        if not isinstance(s_frames, list):
            raise TypeError('s_frames must be a list of (start, end) tuples or None.')
        torch = _torch()
        idx = torch.cat([torch.arange(a, b, device=self.dev) for a, b in frame_spans(self.T, s_frames)])
        return self.y.index_select(0, idx).contiguous(), self.var.index_select(0, idx).contiguous()


def _to_stable_s(u, eps: float = 1e-3):
    """sigmoid(u) (1 - 2 eps) + eps (reference eks/ibl_pupil_smoother.py:506-508)."""
    return 1.0 / (1.0 + np.exp(-np.asarray(u, dtype=np.float64))) * (1.0 - 2 * eps) + eps


def _optimize_on_device(P: _PupilProblem, s_frames, lr, tol, safety_cap, sync_every: int = 8):
    """Adam on u = logit-like reparametrisation of (s_diam, s_com), entirely on the device:
    eks_pupil_adam_run = { eks_ar1_nll (loss + 2 sensitivities) -> eks_pupil_adam_step } x
    `sync_every` per host round trip (steps enqueued after convergence leave the state untouched).
    Returns (s_d, s_c, info)."""
    torch = _torch()
    y_c, var_c = P.cropped(s_frames)
    loss = hip_ops.Ar1Loss(y_c, var_c, P.m0, P.S0, P.C, n_tan=2,
                           positive_noise=bool(np.all(np.asarray(P.latent_vars) > 0.0)))
    s0 = np.array([0.99, 0.98], dtype=np.float32).astype(np.float64)     # :561-562
    state = np.zeros((1, 9))
    state[0, 0:2] = np.log(s0 / (1.0 - s0))
    state[0, 6] = np.inf
    state = torch.as_tensor(state, device=P.dev)
    latent = torch.as_tensor(P.latent_vars[None], device=P.dev)
    n_active = torch.zeros(1, dtype=torch.int32, device=P.dev)
    hip_ops.pupil_adam_step(loss, latent, state, n_active, lr, tol, safety_cap, init=True)
    # The running count after round r is copied to pinned memory behind round r and read only after round r + 1
    # has been enqueued (steps enqueued after convergence leave the state untouched): the device never waits for
    # the host's answer (core._optimize_on_device does the same).
    from .core import _pinned_empty
    cap = int(safety_cap)
    snap = _pinned_empty((max((cap + sync_every - 1) // sync_every, 1),), torch.int32)
    launched, r, pending = 0, 0, None
    while launched < cap:
        n = min(sync_every, cap - launched)
        hip_ops.pupil_adam_run(loss, latent, state, n_active, lr, tol, safety_cap, n)
        launched += n
        snap[r:r + 1].copy_(n_active, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        if pending is not None:
            pending[1].synchronize()
            if int(snap[pending[0]]) == 0:
                break
        pending = (r, ev)
        r += 1
    st = state.cpu().numpy()[0]
    s = _to_stable_s(st[0:2])
    return float(s[0]), float(s[1]), dict(iters=int(st[7]), last_loss=float(st[6]),
                                          converged=bool(st[8]), launched=launched)


def _diag_vars(R):
    """(T, O, O) diagonal covariances (upstream's argument) or (T, O) variances -> (T, O)."""
    R = _to_numpy(R)
    return np.diagonal(R, axis1=-2, axis2=-1) if R.ndim == 3 else R


def pupil_optimize_smooth(ys, m0, S0, C, R, diameters_var, x_var, y_var,
                          s_frames: list | None = None, smooth_params: list | None = None,
                          lr: float = 5e-3, tol: float = 1e-6, safety_cap: int = 5000) -> tuple:
    """(s_diam, s_com) minimising the filter NLL with time-varying R on the (cropped) data
    (reference eks/ibl_pupil_smoother.py:451-607).  `R` is (T, O, O) diagonal like upstream or the
    (T, O) variances.  Both smooth_params given: returned after rounding to float32 and clipping
    to [1e-3, 1 - 1e-3] (:555-557)."""
    fixed = _fixed_params(smooth_params)
    if fixed is not None:
        return fixed
    P = _PupilProblem(ys, m0, S0, C, _diag_vars(R), [diameters_var, x_var, y_var])
    s_d, s_c, info = _optimize_on_device(P, s_frames, lr, tol, safety_cap)
    _log_opt(s_d, s_c, info)
    return s_d, s_c


def _fixed_params(smooth_params):
    if smooth_params is not None and all(v is not None for v in smooth_params):
        s = np.clip(np.asarray(smooth_params, dtype=np.float32), np.float32(1e-3),
                    np.float32(1 - 1e-3))
        return float(s[0]), float(s[1])
    return None


def _log_opt(s_d, s_c, info):
    logger.debug(f"[pupil/hip] iters={info['iters']}  s_diam={s_d:.6f}  s_com={s_c:.6f}  "
                 f"NLL={info['last_loss']:.6f}")


def run_pupil_kalman_smoother(ys, m0, S0, C, ensemble_vars, diameters_var, x_var, y_var,
                              s_frames: list | None = None, smooth_params: list | None = None,
                              lr: float = 5e-3, tol: float = 1e-6, safety_cap: int = 5000,
                              return_info: bool = False) -> tuple:
    """Optimise [s_diam, s_com] on the (cropped) loss, then smooth all frames with A(s), Q(s) and
    R_t (reference eks/ibl_pupil_smoother.py:363-448).  ys, ensemble_vars (T, 8); m0 (3,);
    S0 (3, 3); C (8, 3).  Returns ([s_diam, s_com], ms (T, 3), Vs (T, 3, 3))."""
    torch = _torch()
    P = _PupilProblem(ys, m0, S0, C, ensemble_vars, [diameters_var, x_var, y_var])
    fixed = _fixed_params(smooth_params)
    info = dict(iters=0, last_loss=float('nan'), converged=True, launched=0)
    if fixed is not None:
        s_d, s_c = fixed
    else:
        s_d, s_c, info = _optimize_on_device(P, s_frames, lr, tol, safety_cap)
        _log_opt(s_d, s_c, info)
    a = np.array([s_d, s_c, s_c])
    A = torch.as_tensor(np.diag(a)[None], device=P.dev)
    Q = torch.as_tensor(np.diag(P.latent_vars * (1.0 - a * a))[None], device=P.dev)
    one = torch.ones(1, dtype=torch.float64, device=P.dev)
    ms, Vs = hip_ops.smooth(P.y, P.var, P.m0, P.S0, A, P.C, Q, one, flags=0)
    ms = ms[:, 0].cpu().numpy().astype(np.float64)
    Vs = Vs[:, 0].cpu().numpy().astype(np.float64)
    out = ([s_d, s_c], ms, Vs)
    return out + (info,) if return_info else out
