"""eks_amd - MI355X-native ensemble Kalman smoother: drop-in for the Kalman hot path of
paninski-lab/eks (run_kalman_smoother and its singlecam, multicam - linear and calibrated -,
IBL-pupil and IBL-paw drivers).

The public names mirror the reference's `eks/__init__.py`.  Importing this package needs no GPU;
calling the smoothers does (there is no CPU fallback)."""
__version__ = '0.1.0'

from .marker_array import MarkerArray, input_dfs_to_markerArray  # noqa: F401


def __getattr__(name):          # lazy: pandas/sklearn/torch are only imported when used
    import importlib
    table = {
        'fit_eks_singlecam': 'singlecam_smoother', 'ensemble_kalman_smoother_singlecam': 'singlecam_smoother',
        'fit_eks_mirrored_multicam': 'multicam_smoother', 'fit_eks_multicam': 'multicam_smoother',
        'ensemble_kalman_smoother_multicam': 'multicam_smoother',
        'fit_eks_multicam_ibl_paw': 'ibl_paw_multicam_smoother',
        'fit_eks_pupil': 'ibl_pupil_smoother', 'ensemble_kalman_smoother_ibl_pupil': 'ibl_pupil_smoother',
        'run_kalman_smoother': 'core', 'ensemble': 'core', 'optimize_smooth_param': 'core',
    }
    if name in table:
        return getattr(importlib.import_module(f'.{table[name]}', __name__), name)
    raise AttributeError(name)
