"""Device-tensor wrappers over the C ABI (include/eks_hip.h).  PyTorch is used only for device
memory, streams and dtype/layout checks; every computation is a kernel of libeks_hip.so.

All tensors must live on a ROCm device.  Layout is frame-major: y, var (T, K, O) float32;
ms (T, K, D), Vs (T, K, D, D) or (T, K, D) float32; parameters float64 with the reference's shapes.
"""
from __future__ import annotations

import ctypes

import numpy as np

import torch

from . import _lib
from ._lib import FLAG_ADAM_PREPARED, FLAG_DIAG_MODEL, FLAG_Q_PD, FLAG_UNIT_AC, FLAG_VS_DIAG, EksDims


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.EksHipError('no ROCm device visible: eks_amd has no CPU fallback for the Kalman '
                               'path (the float64 oracle under oracle/ is test infrastructure only)')
    # (a background warm-up of every unit's code object at the first entry was measured SLOWER than letting each unit
    #  load at its first launch - 44 / 33 ms against 23 / 24 ms, tools/first_call.py - and its switch is gone; callers who
    #  want the loads ahead of time call warmup())
    return torch.device('cuda', torch.cuda.current_device())


_WARM_SETS = {
    # what the first call of each driver is going to launch
    'diag': ('misc', 'diag', 'diag_nll'),                                   # singlecam: scalar chains
    'dense': ('misc', 'dense', 'dense_wave', 'dense_wide', 'loss', 'multicam'),   # multicam, linear
    'pupil': ('misc', 'dense', 'dense_wave', 'loss_ar1'),
    'all': tuple(_lib.WARM),
}
_warm_done = set()
_warm_thread = None


def warmup(what: str = 'all', background: bool = False):
    """eks_warmup: load the code objects the named path (`diag` = singlecam, `dense` = linear multicam, `pupil`,
    `all`) is going to need, so that the first smoothing call of the process does not pay for it (tens of
    milliseconds per translation unit: for a 2 000-frame recording that was the whole run time).  With
    background=True the loading runs on a daemon thread - the drivers start it the moment they are entered, so it
    overlaps their host-side set-up (CSV parsing, ensembling bookkeeping); calls into the library made meanwhile just
    wait for the unit they need.  Returns {unit: milliseconds} (foreground) or the thread."""
    import threading
    global _warm_thread
    require_gpu()
    lib = _lib.load()
    units = [u for u in _WARM_SETS[what] if u not in _warm_done]
    mask = 0
    for u in units:
        mask |= _lib.WARM[u]
    if not mask:
        return {} if not background else _warm_thread
    dev = torch.cuda.current_device()

    def work():
        torch.cuda.set_device(dev)
        ms = (ctypes.c_float * 9)()
        _lib.check(lib.eks_warmup(mask, ms), 'eks_warmup')
        _warm_done.update(units)                     # (only once the units really are loaded)
        return {u: float(ms[i]) for i, u in enumerate(_lib.WARM) if mask & _lib.WARM[u]}

    if background:
        _warm_thread = threading.Thread(target=work, name='eks-warmup', daemon=True)
        _warm_thread.start()
        return _warm_thread
    return work()


def _ptr(t: torch.Tensor | None):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream():
    """torch's current stream of the current device as a hipStream_t.  (torch.cuda.current_stream() builds a Stream
    object through three Python layers, ~8 us a call - ten calls a search; the raw accessor is one C call.)"""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, dtype, name: str, shape=None) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.EksHipError(f'{name} must be a device tensor')
    if t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f'{name} must have shape {tuple(shape)}, got {tuple(t.shape)}')
    return t.contiguous()


def _dims(K, T, D, O, flags):
    return EksDims(int(K), int(T), int(D), int(O), int(flags))


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


Q_PD_MIN_EIG_RATIO = 1e-6      # EKS_FLAG_Q_PD is asserted for cond(Q) <= 1e6 (every keypoint)


def model_flags(S0, A, C, Q) -> int:
    """Inspect HOST copies of the parameters (numpy) and return DIAG_MODEL / UNIT_AC / Q_PD flags."""
    import numpy as np
    D, O = A.shape[-1], C.shape[-2]
    # one pass in the library when the arrays are what run_kalman_smoother hands over (C-contiguous float64 (K, D, D) /
    # (K, O, D)); a finite Q that is not diagonal needs eigenvalues and takes the NumPy route below
    arrs = (S0, A, C, Q)
    if all(isinstance(a, np.ndarray) and a.dtype == np.float64 and a.ndim == 3 and a.flags.c_contiguous for a in arrs) \
            and S0.shape == A.shape == Q.shape == (A.shape[0], D, D) and C.shape == (A.shape[0], O, D):
        rc = _lib.load().eks_host_model_flags(A.shape[0], D, O, S0.ctypes.data, A.ctypes.data, C.ctypes.data, Q.ctypes.data,
                                              Q_PD_MIN_EIG_RATIO)
        if rc >= 0:
            return int(rc)
    # Q positive definite with a margin: the smoothing-distribution gradient forms tr((sQ)^-1 E[w w']) - D, a
    # cancellation whose error grows with Q's condition number and is summed over T frames, so it is only taken for
    # cond(Q) <= 1e6 per keypoint (measured against the dual-number kernels at the threshold and T = 50 000:
    # tests/test_gpu_kernels.py::test_score_gradient_at_the_conditioning_threshold); worse-conditioned Q - the
    # normalised covariance of principal-component differences the multicam driver feeds can be - takes the
    # dual-number kernels, which do not invert Q (include/eks_hip.h: EKS_FLAG_Q_PD)
    Qh = np.asarray(Q, dtype=np.float64)
    off = ~np.eye(D, dtype=bool)

    def is_diag(M):
        return bool(np.all(M[..., off] == 0))

    pd = 0
    if np.all(np.isfinite(Qh)):
        if Qh.shape[-1] == Qh.shape[-2] == D and is_diag(Qh):
            # (a diagonal Q's eigenvalues are its diagonal: no batched LAPACK call - 0.1-0.2 ms at 256 keypoints, in front
            #  of every call's first launch)
            dg = np.diagonal(Qh, axis1=-2, axis2=-1)
            lo_e, hi_e = dg.min(axis=-1), dg.max(axis=-1)
        else:
            ev = np.linalg.eigvalsh(0.5 * (Qh + np.swapaxes(Qh, -1, -2)))
            lo_e, hi_e = ev[..., 0], ev[..., -1]
        pd = FLAG_Q_PD if bool(np.all(lo_e > Q_PD_MIN_EIG_RATIO * np.maximum(hi_e, 1e-300))) else 0

    if D != O or not (is_diag(S0) and is_diag(A) and is_diag(C) and is_diag(Q)):
        return pd
    flags = FLAG_DIAG_MODEL | pd
    eye = np.eye(D)
    if np.array_equal(A, np.broadcast_to(eye, A.shape)) and \
            np.array_equal(C, np.broadcast_to(eye, C.shape)):
        flags |= FLAG_UNIT_AC
    return flags


class PreparedSmooth:
    """eks_smooth with everything but the launches done once: argument checks, output and workspace allocation and
    the ctypes argument list.  Calling the object enqueues the smoother on torch's current stream with the tensors
    it was built from (their CONTENTS may change between calls - e.g. new smoothing parameters written into `s`).
    For callers that smooth small sessions in a loop: configs[1] (10 000 x 64, three launches, 16 us of GPU time)
    costs 19 us per `smooth()` call on the host but 12.5 us through this object (tools/c2_host_time.py)."""

    def __init__(self, y, var, m0, S0, A, C, Q, s, flags: int = 0, vs_diag: bool = False, out=None):
        lib = _lib.load()
        T, K, O = y.shape
        D = m0.shape[-1]
        y = _chk(y, torch.float32, 'y')
        var = _chk(var, torch.float32, 'var', (T, K, O))
        m0 = _chk(m0, torch.float64, 'm0', (K, D))
        S0 = _chk(S0, torch.float64, 'S0', (K, D, D))
        A = _chk(A, torch.float64, 'A', (K, D, D))
        C = _chk(C, torch.float64, 'C', (K, O, D))
        Q = _chk(Q, torch.float64, 'Q', (K, D, D))
        s = _chk(s, torch.float64, 's', (K,))
        if vs_diag:
            flags |= FLAG_VS_DIAG
        self._dims = _dims(K, T, D, O, flags)
        if out is None:
            ms = torch.empty((T, K, D), dtype=torch.float32, device=y.device)
            Vs = torch.empty((T, K, D) if vs_diag else (T, K, D, D), dtype=torch.float32, device=y.device)
        else:
            ms, Vs = out
        ws = _workspace(lib.eks_smooth_workspace_bytes(ctypes.byref(self._dims)), y.device)
        self.ms, self.Vs = ms, Vs
        self._keep = (y, var, m0, S0, A, C, Q, s, ws)           # the pointers below stay valid
        self._fn = lib.eks_smooth
        self._args = (ctypes.byref(self._dims), _ptr(y), _ptr(var), _ptr(m0), _ptr(S0), _ptr(A), _ptr(C), _ptr(Q),
                      _ptr(s), _ptr(ms), _ptr(Vs), _ptr(ws), ws.numel())

    def __call__(self):
        rc = self._fn(*self._args, _stream())
        if rc:
            _lib.check(rc, 'eks_smooth')
        return self.ms, self.Vs


def smooth(y, var, m0, S0, A, C, Q, s, flags: int = 0, vs_diag: bool = False, out=None):
    """eks_smooth: fixed-s Kalman filter + RTS smoother.  Returns (ms, Vs)."""
    return PreparedSmooth(y, var, m0, S0, A, C, Q, s, flags, vs_diag, out)()


def const_r(var, min_var: float = 1e-4):
    """eks_const_r: (T, K, O) float32 -> (K, O) float64 floored time-median."""
    lib = _lib.load()
    T, K, O = var.shape
    var = _chk(var, torch.float32, 'var')
    d = _dims(K, T, O, O, 0)
    out = torch.empty((K, O), dtype=torch.float64, device=var.device)
    ws = _workspace(lib.eks_const_r_workspace_bytes(ctypes.byref(d)), var.device)
    rc = lib.eks_const_r(ctypes.byref(d), _ptr(var), float(min_var), _ptr(out), _ptr(ws), ws.numel(),
                         _stream())
    _lib.check(rc, 'eks_const_r')
    return out


_NP_SUM_PROGRAMS: dict = {}


def np_sum_program(n: int):
    """Leaves and combine order of numpy's pairwise float32 summation of n contiguous values
    (numpy/_core/src/umath/loops_utils.h.src): leaves (L, 2) int32 = (start, length <= 128) left to right, ops
    (L - 1, 3) int32 = (dst, a, b) in evaluation order over slots 0 .. L - 1 (leaf sums) and L .. (internal nodes);
    the root is the last slot.  A function of n alone: cached."""
    if n not in _NP_SUM_PROGRAMS:
        leaves, ops = [], []

        def node(start, m):
            if m <= 128:
                leaves.append((start, m))
                return ('leaf', len(leaves) - 1)
            h = m // 2
            h -= h % 8
            a, b = node(start, h), node(start + h, m - h)
            ops.append([None, a, b])
            return ('op', len(ops) - 1)

        node(0, int(n))
        L = len(leaves)
        slot = lambda ref: ref[1] if ref[0] == 'leaf' else L + ref[1]
        prog = np.array([[L + i, slot(a), slot(b)] for i, (_, a, b) in enumerate(ops)], dtype=np.int32).reshape(-1, 3)
        _NP_SUM_PROGRAMS[n] = (np.array(leaves, dtype=np.int32), prog)
    return _NP_SUM_PROGRAMS[n]


_NP_SUM_DEVICE: dict = {}


def np_nanstd_rows(x):
    """eks_np_nanstd_rows: numpy.nanstd(x, axis=1) of a device float32 matrix (K, n), bit for bit, as a float32 device
    tensor (K,); None when the row does not fit the kernel's LDS (the caller reduces on the host then)."""
    lib = _lib.load()
    x = _chk(x, torch.float32, 'x')
    K, n = x.shape
    leaves, ops = np_sum_program(n)
    # (beyond 8 192 elements numpy reduces in buffered pieces of that size - another order; such rows, e.g. three
    #  cameras' 6 x 1 999 differences, go to the host's own numpy)
    if n > 8192 or (n + 2 * len(leaves)) * 4 > 64 * 1024:
        return None
    key = (n, x.device)
    first = key not in _NP_SUM_DEVICE
    if first:
        _NP_SUM_DEVICE[key] = (torch.as_tensor(leaves, device=x.device), torch.as_tensor(ops, device=x.device))
    lv, op = _NP_SUM_DEVICE[key]
    if lv is None:
        return None
    out = torch.empty(K, dtype=torch.float32, device=x.device)
    rc = lib.eks_np_nanstd_rows(K, n, _ptr(x), _ptr(lv), len(leaves), _ptr(op) if len(ops) else None, len(ops),
                                _ptr(out), _stream())
    _lib.check(rc, 'eks_np_nanstd_rows')
    if first:
        # The kernel hard-codes THIS numpy's pairwise float32 summation (block of 128, eight accumulators): a numpy
        # build that reduces in another order would seed the optimiser differently from the host path without anything
        # failing.  Once per (row length, device): one row against numpy itself; on a mismatch the host reduces.
        import numpy as np
        import warnings
        row = x[:1].cpu().numpy()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)
            ref = np.nanstd(row, axis=1)
        got = out[:1].cpu().numpy()
        if not (np.array_equal(got, ref) or (np.isnan(got).all() and np.isnan(ref).all())):
            _NP_SUM_DEVICE[key] = (None, None)
            return None
    return out


def np_nanstd_diff_rows(x):
    """eks_np_nanstd_diff_rows: numpy.nanstd over (frames, coordinates) of the frame-to-frame differences of a frame-major
    device float32 tensor (T', K, O), per keypoint, bit for bit - what np_nanstd_rows gives for the transposed matrix of
    differences, without making it.  None where that function returns None (the caller reduces on the host)."""
    lib = _lib.load()
    x = _chk(x, torch.float32, 'x')
    Tn, K, O = x.shape
    n = (Tn - 1) * O
    key = (n, x.device)
    hit = _NP_SUM_DEVICE.get(key)
    if hit is None:
        # first use of this row length on this device: the row-matrix form checks the summation order against this
        # NumPy once (one row) and leaves the tables behind
        d = (x[1:] - x[:-1]).transpose(0, 1).contiguous()
        return np_nanstd_rows(d.reshape(K, -1))
    lv, op = hit
    if lv is None:
        return None
    out = torch.empty(K, dtype=torch.float32, device=x.device)
    rc = lib.eks_np_nanstd_diff_rows(Tn, K, O, _ptr(x), _ptr(lv), lv.numel() // 2, _ptr(op) if op.numel() else None,
                                     op.numel() // 3, _ptr(out), _stream())
    _lib.check(rc, 'eks_np_nanstd_diff_rows')
    return out


def order_stats(x, rank_lo: int, rank_hi: int):
    """eks_order_stats: x (T, N) float32 -> (vals (N, 2) float32 = the order statistics rank_lo and rank_hi of
    every column with NaNs sorted last, nan_count (N,) int32), device tensors."""
    lib = _lib.load()
    x = _chk(x, torch.float32, 'x')
    T, N = x.shape
    vals = torch.empty((N, 2), dtype=torch.float32, device=x.device)
    nans = torch.empty(N, dtype=torch.int32, device=x.device)
    rc = lib.eks_order_stats(T, N, _ptr(x), int(rank_lo), int(rank_hi), _ptr(vals), _ptr(nans), _stream())
    _lib.check(rc, 'eks_order_stats')
    return vals, nans


def percentile(x, q: float):
    """numpy.percentile(x, q, axis=0) for a device matrix x (T, N) float32, bit for bit: the two order
    statistics come from eks_order_stats, the interpolation is numpy's own arithmetic on 2 N floats
    (utils.percentile_from_order_stats).  Returns a float32 NumPy array (N,); a column with a NaN gives NaN."""
    from .utils import percentile_from_order_stats, percentile_ranks
    T = x.shape[0]
    r_lo, r_hi, gamma = percentile_ranks(T, q, np.float32)
    vals, nans = order_stats(x, r_lo, r_hi)
    return percentile_from_order_stats(vals.cpu().numpy(), gamma, nans.cpu().numpy())


def nll(y, rconst, m0, S0, A, C, Q, s_cand, per_keypoint: bool = False, want_grad: bool = False,
        flags: int = 0):
    """eks_nll: constant-R filter NLL for candidate s values.  Returns nll (K, n_cand) float64 and
    (if want_grad) d nll / d log s of the same shape."""
    lib = _lib.load()
    T, K, O = y.shape
    D = m0.shape[-1]
    y = _chk(y, torch.float32, 'y')
    rconst = _chk(rconst, torch.float64, 'rconst', (K, O))
    m0 = _chk(m0, torch.float64, 'm0', (K, D))
    S0 = _chk(S0, torch.float64, 'S0', (K, D, D))
    A = _chk(A, torch.float64, 'A', (K, D, D))
    C = _chk(C, torch.float64, 'C', (K, O, D))
    Q = _chk(Q, torch.float64, 'Q', (K, D, D))
    s_cand = _chk(s_cand, torch.float64, 's_cand')
    n_cand = s_cand.shape[-1]
    if per_keypoint and tuple(s_cand.shape) != (K, n_cand):
        raise ValueError('per-keypoint s_cand must be (K, n_cand)')
    if not per_keypoint and s_cand.dim() != 1:
        raise ValueError('shared s_cand must be (n_cand,)')
    d = _dims(K, T, D, O, flags)
    out = torch.empty((K, n_cand), dtype=torch.float64, device=y.device)
    grad = torch.empty((K, n_cand), dtype=torch.float64, device=y.device) if want_grad else None
    ws = _workspace(lib.eks_nll_workspace_bytes(ctypes.byref(d), n_cand), y.device)
    rc = lib.eks_nll(ctypes.byref(d), _ptr(y), _ptr(rconst), _ptr(m0), _ptr(S0), _ptr(A), _ptr(C),
                     _ptr(Q), _ptr(s_cand), n_cand, int(per_keypoint), _ptr(out), _ptr(grad),
                     _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, 'eks_nll')
    return (out, grad) if want_grad else out


def nll_argmin(y, rconst, m0, S0, A, C, Q, s_cand, flags: int = 0):
    """eks_nll_argmin: the grid search in one call - the NLL table (K, n_cand), s at each keypoint's argmin and the
    int32 indices (the scalar-chain grid kernels take the argmin inside the table's assembly)."""
    lib = _lib.load()
    T, K, O = y.shape
    D = m0.shape[-1]
    y = _chk(y, torch.float32, 'y')
    rconst = _chk(rconst, torch.float64, 'rconst', (K, O))
    m0 = _chk(m0, torch.float64, 'm0', (K, D))
    S0 = _chk(S0, torch.float64, 'S0', (K, D, D))
    A = _chk(A, torch.float64, 'A', (K, D, D))
    C = _chk(C, torch.float64, 'C', (K, O, D))
    Q = _chk(Q, torch.float64, 'Q', (K, D, D))
    s_cand = _chk(s_cand, torch.float64, 's_cand')
    if s_cand.dim() != 1:
        raise ValueError('s_cand must be (n_cand,)')
    n_cand = s_cand.shape[0]
    d = _dims(K, T, D, O, flags)
    out = torch.empty((K, n_cand), dtype=torch.float64, device=y.device)
    s_out = torch.empty(K, dtype=torch.float64, device=y.device)
    idx = torch.empty(K, dtype=torch.int32, device=y.device)
    ws = _workspace(lib.eks_nll_workspace_bytes(ctypes.byref(d), n_cand), y.device)
    rc = lib.eks_nll_argmin(ctypes.byref(d), _ptr(y), _ptr(rconst), _ptr(m0), _ptr(S0), _ptr(A), _ptr(C), _ptr(Q),
                            _ptr(s_cand), n_cand, _ptr(out), _ptr(s_out), _ptr(idx), _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, 'eks_nll_argmin')
    return out, s_out, idx


def argmin_s(nll_kc, s_cand):
    lib = _lib.load()
    K, n_cand = nll_kc.shape
    nll_kc = _chk(nll_kc, torch.float64, 'nll')
    s_cand = _chk(s_cand, torch.float64, 's_cand', (n_cand,))
    s_out = torch.empty(K, dtype=torch.float64, device=nll_kc.device)
    idx = torch.empty(K, dtype=torch.int32, device=nll_kc.device)
    rc = lib.eks_argmin_s(K, n_cand, _ptr(nll_kc), _ptr(s_cand), _ptr(s_out), _ptr(idx), _stream())
    _lib.check(rc, 'eks_argmin_s')
    return s_out, idx


def adam_step(block_offsets, block_members, nll_k, dnll_k, state, s_keypoint, n_active, lr, lo, hi,
              tol, safety_cap):
    lib = _lib.load()
    nb = block_offsets.numel() - 1
    rc = lib.eks_adam_step(nb, _ptr(block_offsets), _ptr(block_members), _ptr(nll_k), _ptr(dnll_k),
                           float(lr), float(lo), float(hi), float(tol), int(safety_cap),
                           _ptr(state), _ptr(s_keypoint), _ptr(n_active), _stream())
    _lib.check(rc, 'eks_adam_step')


class AdamLoop:
    """eks_adam_run with every buffer allocated once (reference eks/core.py:654-681 / :520-549):
    `run(n)` enqueues n iterations of loss + gradient + Adam step without touching the host."""

    def __init__(self, y, rconst, m0, S0, A, C, Q, block_offsets, block_members, state, s_keypoint, lr,
                 lo, hi, tol, safety_cap, flags: int = 0):
        self.lib = _lib.load()
        T, K, O = y.shape
        D = m0.shape[-1]
        # (rconst may be given later - set_rconst - by a caller that enqueues prepare() before eks_const_r)
        self.bufs = [_chk(y, torch.float32, 'y'), None if rconst is None else _chk(rconst, torch.float64, 'rconst', (K, O)),
                     _chk(m0, torch.float64, 'm0', (K, D)), _chk(S0, torch.float64, 'S0', (K, D, D)),
                     _chk(A, torch.float64, 'A', (K, D, D)), _chk(C, torch.float64, 'C', (K, O, D)),
                     _chk(Q, torch.float64, 'Q', (K, D, D))]
        self.offs = _chk(block_offsets, torch.int32, 'block_offsets')
        self.members = _chk(block_members, torch.int32, 'block_members', (K,))
        self.nb = self.offs.numel() - 1
        self.state = _chk(state, torch.float64, 'state', (self.nb, 6))
        self.s_keypoint = _chk(s_keypoint, torch.float64, 's_keypoint', (K,))
        self.opt = (float(lr), float(lo), float(hi), float(tol), int(safety_cap))
        dev = y.device
        self.nll = torch.empty(K, dtype=torch.float64, device=dev)
        self.dnll = torch.empty(K, dtype=torch.float64, device=dev)
        self.n_active = torch.zeros(1, dtype=torch.int32, device=dev)
        self.dims = _dims(K, T, D, O, flags)
        self.ws = _workspace(self.lib.eks_nll_workspace_bytes(ctypes.byref(self.dims), 1), dev)
        self.prepared = False

    def set_rconst(self, rconst) -> None:
        K, O = self.dims.n_keypoints, self.dims.obs_dim
        self.bufs[1] = _chk(rconst, torch.float64, 'rconst', (K, O))

    def prepare(self, stream=None) -> bool:
        """eks_adam_prepare: the pass over y that does not depend on the optimiser's state, enqueued now (on `stream`, a
        torch stream; default: the current one) - the caller may fill `state` / `s_keypoint` (e.g. from initial guesses
        it is still waiting for) before the first run().  False: this problem's search does not use such a pass
        (nothing was enqueued)."""
        st = _stream() if stream is None else ctypes.c_void_p(stream.cuda_stream)
        rc = self.lib.eks_adam_prepare(ctypes.byref(self.dims), _ptr(self.bufs[0]), _ptr(self.bufs[4]), self.nb,
                                       _ptr(self.ws), self.ws.numel(), st)
        self.prepared = True                                     # (asked once, whatever the answer)
        if rc == _lib.EKS_ERR_UNSUPPORTED:
            return False
        _lib.check(rc, 'eks_adam_prepare')
        self.dims = _dims(self.dims.n_keypoints, self.dims.n_frames, self.dims.state_dim, self.dims.obs_dim,
                          self.dims.flags | FLAG_ADAM_PREPARED)
        return True

    def stride(self) -> int:
        """eks_adam_run_stride: iterations to ask for per run() on this problem (the spacing of the caller's reads of
        n_active)."""
        return int(self.lib.eks_adam_run_stride(ctypes.byref(self.dims), self.nb))

    def run(self, n_iters: int) -> None:
        if self.bufs[1] is None:
            raise ValueError('AdamLoop.run: rconst has not been set (set_rconst)')
        rc = self.lib.eks_adam_run(ctypes.byref(self.dims), *[_ptr(b) for b in self.bufs], self.nb,
                                   _ptr(self.offs), _ptr(self.members), *self.opt, int(n_iters),
                                   _ptr(self.state), _ptr(self.s_keypoint), _ptr(self.nll),
                                   _ptr(self.dnll), _ptr(self.n_active), _ptr(self.ws), self.ws.numel(),
                                   _stream())
        _lib.check(rc, 'eks_adam_run')


def ensemble(markers, avg_mode: str = 'median', var_mode: str = 'confidence_weighted_var',
             nan_replacement: float = 1000.0):
    """eks_ensemble: (M, V, T, K, 3) float32 -> (V, T, K, 5) float32."""
    lib = _lib.load()
    markers = _chk(markers, torch.float32, 'markers')
    M, V, T, K, F = markers.shape
    if F != 3:
        raise ValueError('markers must have fields (x, y, likelihood)')
    out = torch.empty((V, T, K, 5), dtype=torch.float32, device=markers.device)
    am = 0 if avg_mode == 'median' else 1
    vm = 0 if var_mode in ('conf_weighted_var', 'confidence_weighted_var') else 1
    rc = lib.eks_ensemble(M, V, T, K, _ptr(markers), am, vm, float(nan_replacement), _ptr(out),
                          _stream())
    _lib.check(rc, 'eks_ensemble')
    return out


def maha_inflate(x, v, W, mu, active=None, epsilon: float = 1e-6, threshold: float = 5.0,
                 scalar: float = 10.0, want_maha: bool = False):
    """eks_maha_inflate: one pass of the variance-inflation loop for every active keypoint.
    x (K, N, 2C) float64, v (K, N, 2C) float32 (updated in place), W (K, 2C, L), mu (K, 2C) float64,
    active (K,) int32 or None.  Returns (n_inflated (K,) int32 device tensor, maha (K, N, C) float64
    or None)."""
    lib = _lib.load()
    K, N, O = x.shape
    C = O // 2
    L = W.shape[-1]
    x = _chk(x, torch.float64, 'x')
    if not v.is_contiguous():
        raise ValueError('v must be contiguous (it is updated in place)')
    _chk(v, torch.float32, 'v', (K, N, O))
    W = _chk(W, torch.float64, 'W', (K, O, L))
    mu = _chk(mu, torch.float64, 'mu', (K, O))
    if active is not None:
        active = _chk(active, torch.int32, 'active', (K,))
    n_inf = torch.empty(K, dtype=torch.int32, device=x.device)
    maha = torch.empty((K, N, C), dtype=torch.float64, device=x.device) if want_maha else None
    rc = lib.eks_maha_inflate(K, N, C, L, _ptr(x), _ptr(v), _ptr(W), _ptr(mu), _ptr(active), float(epsilon),
                              float(threshold), float(scalar), _ptr(maha), _ptr(n_inf), _stream())
    _lib.check(rc, 'eks_maha_inflate')
    return n_inf, maha


def multicam_tables(stats, ev, ms, Vs, C, mean, want_latent: bool = True):
    """eks_multicam_tables: stats (V, T, K, 5), ev (T, K, 2V), ms (T, K, D), Vs (T, K, D, D) float32;
    C (K, 2V, D), mean (V, K, 2) float64 -> tables (V, T, K, 9) float64 and latent (T, K, 2D)."""
    lib = _lib.load()
    V, T, K, _ = stats.shape
    D = ms.shape[-1]
    stats = _chk(stats, torch.float32, 'stats', (V, T, K, 5))
    ev = _chk(ev, torch.float32, 'ev', (T, K, 2 * V))
    ms = _chk(ms, torch.float32, 'ms', (T, K, D))
    Vs = _chk(Vs, torch.float32, 'Vs', (T, K, D, D))
    C = _chk(C, torch.float64, 'C', (K, 2 * V, D))
    mean = _chk(mean, torch.float64, 'mean', (V, K, 2))
    tables = torch.empty((V, T, K, 9), dtype=torch.float64, device=stats.device)
    latent = torch.empty((T, K, 2 * D), dtype=torch.float64, device=stats.device) if want_latent else None
    rc = lib.eks_multicam_tables(V, T, K, D, _ptr(stats), _ptr(ev), _ptr(ms), _ptr(Vs), _ptr(C), _ptr(mean),
                                 _ptr(tables), _ptr(latent), _stream())
    _lib.check(rc, 'eks_multicam_tables')
    return tables, latent


class Ar1Loss:
    """eks_ar1_nll with every buffer allocated once: the pupil optimiser evaluates this loss (and
    its two sensitivities) thousands of times on the same arrays (reference
    eks/ibl_pupil_smoother.py:540-594).  y, var (T, K, O) float32; m0 (K, D), S0 (K, D, D),
    C (K, O, D) float64.  `a`, `q` (K, D) and the tangents `da`, `dq` (n_tan, K, D) are device
    buffers owned here that the caller (or eks_pupil_adam_step) fills before `evaluate()`."""

    def __init__(self, y, var, m0, S0, C, n_tan: int = 2, positive_noise: bool = False):
        """positive_noise: the caller guarantees q > 0 in every coordinate for every evaluation (the pupil model:
        q = latent_var (1 - s^2) with s < 1, so latent_var > 0) - EKS_FLAG_Q_PD: on the pupil's shape the tangents'
        derivatives then come from the smoothing distribution inside the smoother's wave kernels instead of
        dual numbers."""
        self.lib = _lib.load()
        T, K, O = y.shape
        D = m0.shape[-1]
        self.y = _chk(y, torch.float32, 'y')
        self.var = _chk(var, torch.float32, 'var', (T, K, O))
        self.m0 = _chk(m0, torch.float64, 'm0', (K, D))
        self.S0 = _chk(S0, torch.float64, 'S0', (K, D, D))
        self.C = _chk(C, torch.float64, 'C', (K, O, D))
        self.n_tan, self.K, self.D = int(n_tan), K, D
        dev = y.device
        f64 = dict(dtype=torch.float64, device=dev)
        self.a = torch.zeros((K, D), **f64)
        self.q = torch.zeros((K, D), **f64)
        self.da = torch.zeros((max(n_tan, 1), K, D), **f64)
        self.dq = torch.zeros((max(n_tan, 1), K, D), **f64)
        self.nll = torch.empty(K, **f64)
        self.dnll = torch.empty((max(n_tan, 1), K), **f64)
        self.dims = _dims(K, T, D, O, FLAG_Q_PD if positive_noise else 0)
        self.ws = _workspace(self.lib.eks_ar1_nll_workspace_bytes(ctypes.byref(self.dims), self.n_tan),
                             dev)

    def evaluate(self):
        """Enqueue one evaluation; results land in self.nll (K,) and self.dnll (n_tan, K)."""
        t = self.n_tan > 0
        rc = self.lib.eks_ar1_nll(ctypes.byref(self.dims), _ptr(self.y), _ptr(self.var), _ptr(self.m0),
                                  _ptr(self.S0), _ptr(self.C), _ptr(self.a), _ptr(self.q),
                                  _ptr(self.da if t else None), _ptr(self.dq if t else None),
                                  self.n_tan, _ptr(self.nll), _ptr(self.dnll if t else None),
                                  _ptr(self.ws), self.ws.numel(), _stream())
        _lib.check(rc, 'eks_ar1_nll')
        return self.nll, self.dnll


def pupil_adam_step(loss: Ar1Loss, latent_var, state, n_active, lr, tol, safety_cap,
                    init: bool = False):
    """eks_pupil_adam_step on the buffers of `loss`; init=True only derives a, q, da, dq from the
    state's u (no update)."""
    rc = loss.lib.eks_pupil_adam_step(loss.K, _ptr(latent_var), _ptr(None if init else loss.nll),
                                      _ptr(None if init else loss.dnll), float(lr), float(tol),
                                      int(safety_cap), _ptr(state), _ptr(loss.a), _ptr(loss.q),
                                      _ptr(loss.da), _ptr(loss.dq), _ptr(n_active), _stream())
    _lib.check(rc, 'eks_pupil_adam_step')


def pupil_adam_run(loss: Ar1Loss, latent_var, state, n_active, lr, tol, safety_cap, n_iters: int):
    """eks_pupil_adam_run: n_iters x (loss + 2 sensitivities + Adam step) in one call."""
    rc = loss.lib.eks_pupil_adam_run(ctypes.byref(loss.dims), _ptr(loss.y), _ptr(loss.var), _ptr(loss.m0),
                                     _ptr(loss.S0), _ptr(loss.C), _ptr(latent_var), float(lr), float(tol),
                                     int(safety_cap), int(n_iters), _ptr(state), _ptr(loss.a), _ptr(loss.q),
                                     _ptr(loss.da), _ptr(loss.dq), _ptr(loss.nll), _ptr(loss.dnll),
                                     _ptr(n_active), _ptr(loss.ws), loss.ws.numel(), _stream())
    _lib.check(rc, 'eks_pupil_adam_run')


def ekf_smooth(y, var, rconst, m0, S0, A, Q, s, cams, xlin, max_sweeps: int = 16, tol: float = 1e-9,
               want_smoother: bool = True, vs_diag: bool = False):
    """eks_ekf_smooth: extended Kalman filter (+ RTS smoother) with calibrated pinhole cameras.

    y (T, Kd, O) float32; exactly one of var (T, Kd, O) float32 / rconst (Kd, O) float64; m0 (K, 3),
    S0, A, Q (K, 3, 3), s (K,) float64 per CHAIN (K a multiple of Kd; chain k reads keypoint
    k % Kd); cams (V, 32) float64 (see include/eks_hip.h), O = 2 V; xlin (K, T, 3) float64 in/out.
    Returns (ms, Vs, nll, info): ms / Vs are None when want_smoother is False; info is a device
    tensor (sweeps executed, last relative change of a linearisation point)."""
    lib = _lib.load()
    T, Kd, O = y.shape
    K = m0.shape[0]
    V = cams.shape[0]
    y = _chk(y, torch.float32, 'y')
    if (var is None) == (rconst is None):
        raise ValueError('pass exactly one of var and rconst')
    if var is not None:
        var = _chk(var, torch.float32, 'var', (T, Kd, O))
    else:
        rconst = _chk(rconst, torch.float64, 'rconst', (Kd, O))
    m0 = _chk(m0, torch.float64, 'm0', (K, 3))
    S0 = _chk(S0, torch.float64, 'S0', (K, 3, 3))
    A = _chk(A, torch.float64, 'A', (K, 3, 3))
    Q = _chk(Q, torch.float64, 'Q', (K, 3, 3))
    s = _chk(s, torch.float64, 's', (K,))
    cams = _chk(cams, torch.float64, 'cams', (V, 32))
    if not xlin.is_contiguous():
        raise ValueError('xlin must be contiguous (it is updated in place)')
    _chk(xlin, torch.float64, 'xlin', (K, T, 3))
    d = _dims(K, T, 3, O, FLAG_VS_DIAG if vs_diag else 0)
    ms = Vs = None
    if want_smoother:
        ms = torch.empty((T, K, 3), dtype=torch.float32, device=y.device)
        Vs = torch.empty((T, K, 3) if vs_diag else (T, K, 3, 3), dtype=torch.float32, device=y.device)
    nll_k = torch.empty((K,), dtype=torch.float64, device=y.device)
    info = torch.zeros((2,), dtype=torch.float64, device=y.device)
    ws = _workspace(lib.eks_ekf_smooth_workspace_bytes(ctypes.byref(d), int(want_smoother)), y.device)
    rc = lib.eks_ekf_smooth(ctypes.byref(d), int(Kd), _ptr(y), _ptr(var), _ptr(rconst), _ptr(m0),
                            _ptr(S0), _ptr(A), _ptr(Q), _ptr(s), _ptr(cams), int(V), _ptr(xlin),
                            int(max_sweeps), float(tol), _ptr(ms), _ptr(Vs), _ptr(nll_k), _ptr(info),
                            _ptr(ws), ws.numel(), _stream())
    _lib.check(rc, 'eks_ekf_smooth')
    return ms, Vs, nll_k, info
