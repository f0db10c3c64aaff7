"""ctypes binding of libeks_hip.so (include/eks_hip.h).  There is no CPU fallback: if the
library is missing, or a call is made without a ROCm device, this raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_size_t, c_uint32, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'lib', 'libeks_hip.so')

EKS_OK = 0
FLAG_DIAG_MODEL = 1
FLAG_VS_DIAG = 2
FLAG_UNIT_AC = 4
FLAG_Q_PD = 8
FLAG_ADAM_PREPARED = 16
EKS_ERR_UNSUPPORTED = -3
# eks_warmup units (include/eks_hip.h: EKS_WARM_*)
WARM = dict(misc=1, diag=2, diag_nll=4, dense=8, dense_wave=16, dense_wide=32, loss=64, loss_ar1=128, multicam=256)
WARM_ALL = 511


class EksDims(ctypes.Structure):
    _fields_ = [('n_keypoints', c_int32), ('n_frames', c_int32), ('state_dim', c_int32),
                ('obs_dim', c_int32), ('flags', c_uint32)]


# name -> (restype, argtypes); the test-suite checks this table against include/eks_hip.h
SIGNATURES = {
    'eks_version': (c_char_p, []),
    'eks_status_string': (c_char_p, [ctypes.c_int]),
    'eks_smooth_workspace_bytes': (c_size_t, [POINTER(EksDims)]),
    'eks_smooth': (ctypes.c_int, [POINTER(EksDims)] + [c_void_p] * 11 + [c_size_t, c_void_p]),
    'eks_const_r_workspace_bytes': (c_size_t, [POINTER(EksDims)]),
    'eks_const_r': (ctypes.c_int, [POINTER(EksDims), c_void_p, c_double, c_void_p, c_void_p,
                                   c_size_t, c_void_p]),
    'eks_nll_workspace_bytes': (c_size_t, [POINTER(EksDims), c_int32]),
    'eks_nll': (ctypes.c_int, [POINTER(EksDims)] + [c_void_p] * 8 + [c_int32, c_int32, c_void_p,
                                                                    c_void_p, c_void_p, c_size_t,
                                                                    c_void_p]),
    'eks_nll_argmin': (ctypes.c_int, [POINTER(EksDims)] + [c_void_p] * 8 + [c_int32, c_void_p, c_void_p, c_void_p,
                                                                           c_void_p, c_size_t, c_void_p]),
    'eks_argmin_s': (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    'eks_np_nanstd_rows': (ctypes.c_int, [c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p,
                                          c_void_p]),
    'eks_np_nanstd_diff_rows': (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32,
                                               c_void_p, c_void_p]),
    'eks_order_stats': (ctypes.c_int, [c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                       c_void_p]),
    'eks_adam_step': (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_double,
                                     c_double, c_double, c_double, c_int32, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    'eks_adam_run': (ctypes.c_int, [POINTER(EksDims)] + [c_void_p] * 7 + [c_int32, c_void_p, c_void_p,
                                                                         c_double, c_double, c_double, c_double,
                                                                         c_int32, c_int32] + [c_void_p] * 6
                     + [c_size_t, c_void_p]),
    'eks_pupil_adam_run': (ctypes.c_int, [POINTER(EksDims)] + [c_void_p] * 6 + [c_double, c_double, c_int32,
                                                                               c_int32] + [c_void_p] * 9
                           + [c_size_t, c_void_p]),
    'eks_ar1_nll_workspace_bytes': (c_size_t, [POINTER(EksDims), c_int32]),
    'eks_ar1_nll': (ctypes.c_int, [POINTER(EksDims)] + [c_void_p] * 9 + [c_int32, c_void_p, c_void_p,
                                                                        c_void_p, c_size_t, c_void_p]),
    'eks_pupil_adam_step': (ctypes.c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_double, c_double,
                                           c_int32] + [c_void_p] * 7),
    'eks_ekf_smooth_workspace_bytes': (c_size_t, [POINTER(EksDims), c_int32]),
    'eks_ekf_smooth': (ctypes.c_int, [POINTER(EksDims), c_int32] + [c_void_p] * 9 + [c_int32, c_void_p,
                                                                                    c_int32, c_double]
                       + [c_void_p] * 5 + [c_size_t, c_void_p]),
    'eks_maha_inflate': (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32] + [c_void_p] * 5
                         + [c_double, c_double, c_double, c_void_p, c_void_p, c_void_p]),
    'eks_multicam_tables': (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32] + [c_void_p] * 9),
    'eks_warmup': (ctypes.c_int, [c_uint32, c_void_p]),
    'eks_profile_enable': (ctypes.c_int, [ctypes.c_int]),
    'eks_knobs_reload': (ctypes.c_int, []),
    'eks_csv_read_numeric': (ctypes.c_int, [c_char_p, c_int32, c_void_p, ctypes.c_int64, c_void_p, c_void_p, c_void_p,
                                            c_int32, c_int32]),
    'eks_csv_write_table': (ctypes.c_int, [c_char_p, c_char_p, ctypes.c_int64, c_void_p, c_void_p, ctypes.c_int64, c_int32,
                                           c_int32]),
    'eks_format_repr': (ctypes.c_int, [c_void_p, ctypes.c_int64, c_void_p, ctypes.c_int64, c_void_p]),
    'eks_host_thread_speedup': (c_double, [c_int32]),
    'eks_host_gather_cols': (ctypes.c_int, [c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                            c_void_p, c_int32]),
    'eks_host_model_flags': (ctypes.c_int, [c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_double]),
    'eks_adam_run_stride': (c_int32, [c_void_p, c_int32]),
    'eks_adam_prepare': (ctypes.c_int, [POINTER(EksDims), c_void_p, c_void_p, c_int32, c_void_p, c_size_t, c_void_p]),
    'eks_profile_drain': (ctypes.c_int, [c_void_p, c_size_t, c_void_p, c_int32]),
    'eks_ensemble': (ctypes.c_int, [c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32,
                                    c_float, c_void_p, c_void_p]),
}

_lib = None


class EksHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libeks_hip.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get('EKS_HIP_LIB', LIB_PATH)      # A/B builds of the same sources (tools/)
    if not os.path.exists(path):
        raise EksHipError(
            f'{path} not found: the HIP extension has not been built. '
            'Run `python -m eks_amd._build` (needs hipcc). There is no CPU fallback.')
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != EKS_OK:
        msg = load().eks_status_string(status).decode()
        raise EksHipError(f'{what} failed: status {status} ({msg})')
