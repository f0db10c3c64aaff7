"""Core operators of the ensemble Kalman smoother on MI355X (mirror of the reference's
eks/core.py, same names / arguments / return types).

    ensemble(marker_array, avg_mode, var_mode, nan_replacement) -> MarkerArray (1,V,T,K,5)
    compute_initial_guesses(ensemble_vars) -> float
    run_kalman_smoother(ys, m0s, S0s, As, Cs, Qs, ensemble_vars, s_frames, smooth_param, blocks,
                        lr, s_bounds_log, tol, safety_cap, h_fn) -> (s_finals, ms, Vs)
    optimize_smooth_param(...)           -> fills s_finals in place
    constant_R_from_timevarying(R_t, min_var)

All arithmetic of the Kalman path runs in the HIP kernels of libeks_hip.so (eks_amd/csrc); this
module only moves arrays to the device in the kernels' frame-major layout and drives the search
for the smoothing parameter.  There is no CPU fallback.
"""
from __future__ import annotations

import logging
import os
import warnings
import time
from typing import Callable, Literal

import numpy as np

from . import hip_ops
from .marker_array import MarkerArray
from .utils import frame_spans

logger = logging.getLogger(__name__)


# ------------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


_PINNED_CAP_BYTES = int(os.environ.get('EKS_PINNED_CAP_BYTES', 1 << 30))
_pinned_live = [0]          # bytes of page-locked result buffers callers still hold
# The host layer's process-wide state (this counter - also touched from finalisers, on whatever thread the garbage
# collector runs - and the per-device side streams below) is guarded: several threads may drive sessions, each on its own
# torch stream (INTEGRATION.md "Threads"; tests/test_gpu_drivers.py::test_two_threads_two_streams...).
import threading
_state_lock = threading.Lock()


def _release_pinned(nbytes: int) -> None:
    with _state_lock:
        _pinned_live[0] -= nbytes


def _pinned_reserve(nbytes: int) -> bool:
    """Count nbytes of page-locked results against the cap; False (nothing counted) beyond it."""
    with _state_lock:
        if _pinned_live[0] + nbytes > _PINNED_CAP_BYTES:
            return False
        _pinned_live[0] += nbytes
        return True


def _pinned_empty(shape, dtype):
    return _torch().empty(shape, dtype=dtype, pin_memory=True)


def _to_host(*tensors, pinned: bool | None = None):
    """Device tensors -> NumPy.  Fast path: page-locked staging buffers (torch's caching host
    allocator keeps them across calls), so the D2H copies run at the link rate and overlap each
    other; the arrays alias the staging storage.  Page-locked memory is unswappable, so the bytes of
    such results that callers still hold are counted and capped at EKS_PINNED_CAP_BYTES (1 GiB):
    beyond the cap - a loop that keeps every session's output, `distributed.smooth_sessions` - or
    with pinned=False results come back in ordinary pageable arrays.  The count is released by a
    finaliser on the array's `.base` - the tensor object NumPy itself keeps alive for as long as the
    array or any view of it exists (`tensor.numpy().base` is a fresh tensor object, not the staging
    tensor, so a finaliser on the latter would fire as soon as this function returned)."""
    import weakref
    torch = _torch()
    nbytes = sum(t.numel() * t.element_size() for t in tensors)
    if pinned is None:
        pinned = not os.environ.get('EKS_PAGEABLE_D2H')
    if not pinned or nbytes > (2 << 30) or not _pinned_reserve(nbytes):
        return tuple(t.cpu().numpy() for t in tensors)
    try:
        host = [_pinned_empty(t.shape, t.dtype) for t in tensors]
    except RuntimeError:            # page-locked memory exhausted or unavailable: plain copies
        _release_pinned(nbytes)
        return tuple(t.cpu().numpy() for t in tensors)
    for h, t in zip(host, tensors):
        h.copy_(t, non_blocking=True)
    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()
    out = []
    for h in host:
        arr = h.numpy()
        n = h.numel() * h.element_size()
        owner = arr.base if arr.base is not None else h
        weakref.finalize(owner, _release_pinned, n)            # (reserved above)
        out.append(arr)
    return tuple(out)


def _to_numpy(a, dtype=None) -> np.ndarray:
    if hasattr(a, 'detach'):                      # torch tensor
        a = a.detach().cpu().numpy()
    return np.asarray(a, dtype=dtype)


def ensemble(marker_array: MarkerArray, avg_mode: Literal['mean', 'median'] = 'median',
             var_mode: Literal['var', 'confidence_weighted_var'] = 'confidence_weighted_var',
             nan_replacement: float = 1000.0) -> MarkerArray:
    """Ensemble mean/median and variance over models (reference eks/core.py:25-101), computed by
    the `eks_ensemble` kernel.  Input (M,V,T,K,3) fields x,y,likelihood; output float32
    (1,V,T,K,5) fields x,y,var_x,var_y,likelihood."""
    torch = _torch()
    dev = hip_ops.require_gpu()
    fields = list(marker_array.data_fields)
    arr = _to_numpy(marker_array.array)
    if fields != ['x', 'y', 'likelihood']:
        arr = arr[..., [fields.index(f) for f in ('x', 'y', 'likelihood')]]
    mk = torch.as_tensor(np.ascontiguousarray(arr, dtype=np.float32), device=dev)
    stats = hip_ops.ensemble(mk, avg_mode, var_mode, nan_replacement)
    return MarkerArray(_to_host(stats)[0][None], data_fields=['x', 'y', 'var_x', 'var_y', 'likelihood'])


def compute_initial_guesses(ensemble_vars) -> float:
    """Initial guess for s: std of frame-to-frame changes of the ensemble variance over the first
    2000 frames, rounded to 5 decimals (reference eks/core.py:104-133)."""
    ev = np.asarray(ensemble_vars)[:2000]
    if ev.shape[0] < 2:
        raise ValueError('Not enough frames to compute temporal differences.')
    return float(round(float(np.nanstd(ev[1:] - ev[:-1])), 5))


def _guess_rows_from_device(ev_dev) -> np.ndarray:
    """The (K, (T' - 1) O) matrix of frame-to-frame differences _initial_guesses_per_keypoint reduces, prepared on the
    device from a (T, K, O) tensor: the float32 subtraction is the same IEEE operation wherever it runs and the
    transposition is a copy, so the rows are the host version's bit for bit - without its 2 ms of strided host copies
    in front of the first optimiser launch."""
    ev = ev_dev[:2000]
    if ev.shape[0] < 2:
        raise ValueError('Not enough frames to compute temporal differences.')
    d = (ev[1:] - ev[:-1]).transpose(0, 1).contiguous()
    return d.reshape(d.shape[0], -1).cpu().numpy()


def _guess_std_on_device(ev_dev, lazy: bool = False):
    """numpy.nanstd of every keypoint's frame-to-frame differences, computed ON THE DEVICE with numpy's own summation
    order (hip_ops.np_nanstd_rows: bit for bit) - K floats come back instead of the (K, (T' - 1) O) rows, and the
    2 ms the host reduction cost at 256 keypoints are gone from in front of the optimiser's first launch.  None
    when the rows are not float32 or too long for the kernel (the host reduces then)."""
    torch = _torch()
    ev = ev_dev[:2000]
    if ev.shape[0] < 2:
        raise ValueError('Not enough frames to compute temporal differences.')
    if ev.dtype != torch.float32 or not ev.is_cuda:
        return None
    if ev.dim() == 3 and ev.is_contiguous():
        sd = hip_ops.np_nanstd_diff_rows(ev)                    # (differences formed inside the reduction's launch)
    else:
        d = (ev[1:] - ev[:-1]).transpose(0, 1).contiguous()
        sd = hip_ops.np_nanstd_rows(d.reshape(d.shape[0], -1))
    if sd is None or not lazy:
        return None if sd is None else sd.cpu().numpy()
    # The K floats start their way back NOW, behind the reduction, into page-locked memory: whatever the caller enqueues
    # next (eks_const_r, the lag sums' pass over y) runs beside the host's wait - a `.cpu()` at the time of asking would
    # queue the copy behind all of that.
    try:
        host = _pinned_empty(sd.shape, sd.dtype)
    except RuntimeError:
        return lambda: sd.cpu().numpy()
    host.copy_(sd, non_blocking=True)
    ready = torch.cuda.Event()
    ready.record()

    def fetch():
        ready.synchronize()
        return host.numpy().copy()
    return fetch


def _initial_guesses_per_keypoint(ev_host: np.ndarray = None, rows: np.ndarray | None = None,
                                  sd: np.ndarray | None = None) -> np.ndarray:
    """compute_initial_guesses for every keypoint of a (T', K, O) array at once (missing or non-positive guesses
    become 2.0, as in run_kalman_smoother's loop).  Each keypoint's differences are laid out as one contiguous
    row in the order the 2-D call reduces them, so the values are the per-keypoint calls' bit for bit; a loop of K
    nanstd calls cost 13 ms at K = 256 - twice the optimisation it seeds."""
    if sd is not None:
        d = None                                                 # (reduced on the device: _guess_std_on_device)
    elif rows is not None:
        d = rows                                                 # (prepared on the device: _guess_rows_from_device)
    else:
        ev = np.asarray(ev_host)[:2000]
        if ev.shape[0] < 2:
            raise ValueError('Not enough frames to compute temporal differences.')
        d = np.ascontiguousarray(np.swapaxes(ev[1:] - ev[:-1], 0, 1)).reshape(ev.shape[1], -1)    # (K, (T'-1) O)
    if sd is None:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', RuntimeWarning)      # all-NaN keypoints: nan -> 2.0 below
            sd = np.nanstd(d, axis=1)
    g = np.array([round(v, 5) for v in np.asarray(sd, dtype=np.float64).tolist()])   # Python's round on a Python
                                                                 # float, as the scalar form
    g = np.where(g == 0.0, 2.0, g)                               # (`or 2.0` of the loop form)
    return np.where(np.isfinite(g) & (g > 0.0), g, 2.0)


def constant_R_from_timevarying(R_t_np: np.ndarray, min_var: float = 1e-4) -> np.ndarray:
    """(T', O, O) -> constant diagonal R by median over time, floored (reference
    eks/core.py:702-709).  Host helper kept for API parity; the optimiser uses `eks_const_r`."""
    d = np.diagonal(np.asarray(R_t_np), axis1=-2, axis2=-1)
    return np.diag(np.clip(np.nanmedian(d, axis=0), min_var, np.inf)).astype(R_t_np.dtype)


# ------------------------------------------------------------------------------------------
class _DeviceProblem:
    """Inputs of run_kalman_smoother on the device, frame-major."""

    def __init__(self, ys, m0s, S0s, As, Cs, Qs, ensemble_vars, flags=None):
        torch = _torch()
        self.dev = hip_ops.require_gpu()
        host = {k: np.ascontiguousarray(_to_numpy(v, np.float64))
                for k, v in dict(m0=m0s, S0=S0s, A=As, C=Cs, Q=Qs).items()}
        self.K, self.D = host['m0'].shape
        self.O = host['C'].shape[1]
        # (the tiled host path decides the model's flags ONCE, over all keypoints: a session of which only some keypoints
        #  are diagonal or well conditioned must not take different kernels in different tiles)
        self.flags = hip_ops.model_flags(host['S0'], host['A'], host['C'], host['Q']) if flags is None else int(flags)
        # the five parameter arrays go up as ONE copy (an upload of a few hundred bytes costs ~19 us of host time
        # whatever its size: five of them were a fifth of a 2 000-frame session's whole call)
        keys = ('m0', 'S0', 'A', 'C', 'Q')
        # (page-locked and asynchronous: a pageable copy holds the host until the stream - the previous call's smoother
        #  included - has drained, so back-to-back calls could not overlap their set-up with the device's work)
        n_par = sum(host[k].size for k in keys)
        try:
            stage = _pinned_empty((n_par,), torch.float64)
            np.concatenate([host[k].ravel() for k in keys], out=stage.numpy())
            flat = stage.to(self.dev, non_blocking=True)
        except RuntimeError:
            flat = torch.as_tensor(np.concatenate([host[k].ravel() for k in keys]), device=self.dev)
        self.params, at = [], 0
        for k in keys:
            n = host[k].size
            self.params.append(flat[at:at + n].view(host[k].shape))
            at += n
        # ys arrives (K,T,O) like upstream; a torch tensor that is a transposed view of a
        # frame-major buffer is taken zero-copy, anything else is transposed once
        if hasattr(ys, 'detach'):
            y = ys.to(self.dev, dtype=torch.float32).transpose(0, 1).contiguous()
        else:
            # upload in the caller's (K,T,O) layout and dtype; the float32 conversion and the
            # transposition to frame-major happen on the device (a host-side transpose of a few
            # hundred MB costs more than everything the GPU does)
            y = torch.as_tensor(np.ascontiguousarray(_to_numpy(ys)), device=self.dev).to(torch.float32)
            y = y.transpose(0, 1).contiguous()
        if hasattr(ensemble_vars, 'detach'):
            var = ensemble_vars.to(self.dev, dtype=torch.float32).contiguous()
        else:
            var = torch.as_tensor(np.ascontiguousarray(_to_numpy(ensemble_vars)), device=self.dev)
            var = var.to(torch.float32).contiguous()
        self.T = y.shape[0]
        if tuple(y.shape) != (self.T, self.K, self.O) or tuple(var.shape) != (self.T, self.K, self.O):
            raise ValueError(f'ys must be (K,T,O) and ensemble_vars (T,K,O); got {tuple(ys.shape)} '
                             f'and {tuple(ensemble_vars.shape)}')
        self.y, self.var = y, var

    def cropped(self, s_frames):
        """(y, var) restricted to the s_frames spans (loss only, reference eks/core.py:599-601)."""
        if not s_frames or (len(s_frames) == 1 and s_frames[0] == (None, None)):
            return self.y, self.var
        if not isinstance(s_frames, list):
            raise TypeError('s_frames must be a list of (start, end) tuples or None.')
        torch = _torch()
        spans = frame_spans(self.T, s_frames)
        idx = torch.cat([torch.arange(a, b, device=self.dev) for a, b in spans])
        return self.y.index_select(0, idx).contiguous(), self.var.index_select(0, idx).contiguous()


def _block_csr(blocks, K):
    """CSR form of `blocks` for the device optimiser.  The blocks must partition range(K): the
    reference (eks/core.py:553-554) would leave the s_finals of an uncovered keypoint unset, and a
    stray index would address device memory out of bounds here."""
    if blocks is None:                 # the reference's default (eks/core.py:223-224): every keypoint its own block
        ar = np.arange(K, dtype=np.int64)
        return np.arange(K + 1, dtype=np.int32), ar.astype(np.int32), ar
    members = np.concatenate([np.asarray(b, dtype=np.int64).reshape(-1) for b in blocks]) \
        if len(blocks) else np.zeros(0, dtype=np.int64)
    if members.size != K or not np.array_equal(np.sort(members), np.arange(K)):
        raise ValueError(f'blocks must partition the {K} keypoints (every index 0..{K - 1} exactly '
                         f'once); got {[list(map(int, b)) for b in blocks]}')
    counts = np.fromiter((len(b) for b in blocks), dtype=np.int64, count=len(blocks))
    offs = np.zeros(len(blocks) + 1, dtype=np.int32)
    offs[1:] = np.cumsum(counts)
    of_kp = np.full(K, -1, dtype=np.int64)
    of_kp[members] = np.repeat(np.arange(len(blocks)), counts)       # block of every keypoint
    return offs, members.astype(np.int32), of_kp


def _adam_setup(P: _DeviceProblem, blocks, s_frames, lr, s_bounds_log, tol, safety_cap, min_R_var=1e-4):
    """The optimiser's buffers, and the pass over y that does not depend on its starting point (the lag sums of
    eks_lag_adam.hip) ENQUEUED - before the initial guesses are asked for, before eks_const_r, before the host has done
    anything else for this search: the device has ~0.4 ms of work (BASELINE configs[2]) beside which the host prepares
    the rest.  Returns what _optimize_on_device continues from."""
    torch = _torch()
    y_c, var_c = P.cropped(s_frames)
    lo, hi = float(s_bounds_log[0]), float(s_bounds_log[1])
    offs, members, of_kp = _block_csr(blocks, P.K)
    nb = len(offs) - 1
    packed = torch.empty(nb * 6 + P.K, dtype=torch.float64, device=P.dev)
    state, s_kp = packed[:nb * 6].view(nb, 6), packed[nb * 6:]
    if blocks is None or (nb == P.K and np.array_equal(members, np.arange(P.K))):
        offs_d, mem_d = _identity_blocks(P.K, P.dev)
    else:
        offs_d = torch.as_tensor(offs, device=P.dev)
        mem_d = torch.as_tensor(members, device=P.dev)
    loop = hip_ops.AdamLoop(y_c, None, *P.params, offs_d, mem_d, state, s_kp, lr, lo, hi, tol, safety_cap, flags=P.flags)
    prep_ev = None
    mode = os.environ.get('EKS_ADAM_PREPARE', 'side')        # (A/B runs: 'after_const_r' | 'first')
    if mode == 'first':                                      # the pass on the caller's stream, in front of everything
        loop.prepare()
    elif mode == 'side':
        # on a SIDE stream beside the guesses' reduction and eks_const_r.  The pass fills every compute unit (140 KB of
        # LDS per workgroup) and what runs beside it crawls - the median's sampling kernels take 90-290 us for 12, its
        # full pass (40 KB of LDS) only fits where a workgroup of the pass has gone - so the overlap is worth little:
        # 0.946 ms per C3 step against 0.969 for ONE queue in the order guesses, median, pass ('after_const_r'), 1.02
        # with the pass in front of everything ('first'), and 1.11 with eks_const_r enqueued in front of the pass (its
        # full pass then starves for 298 us and the guesses behind it arrive late).
        cur = torch.cuda.current_stream(P.dev)
        side = _prepare_stream(P.dev)
        side.wait_stream(cur)                                # (y, and whatever produced it, is ready)
        if loop.prepare(side):                               # (the stream by its handle: no context manager to enter)
            prep_ev = torch.cuda.Event()
            prep_ev.record(side)
            for t in (y_c, loop.ws, P.params[2]):            # what the pass reads and writes: y, the workspace, A
                t.record_stream(side)
    # ('after_const_r': _optimize_on_device enqueues the pass right behind eks_const_r)
    return dict(y_c=y_c, var_c=var_c, offs=offs, members=members, of_kp=of_kp, packed=packed, state=state, s_kp=s_kp,
                loop=loop, prep_ev=prep_ev)


def _optimize_on_device(P: _DeviceProblem, blocks, s_frames, s_guess_per_k, lr, s_bounds_log, tol,
                        safety_cap, min_R_var, s_mode, n_grid, sync_every: int | None = None, setup=None):
    """Returns (s per keypoint as a device float64 tensor, info dict)."""
    torch = _torch()
    lo, hi = float(s_bounds_log[0]), float(s_bounds_log[1])
    if s_mode != 'grid' and setup is None:
        setup = _adam_setup(P, blocks, s_frames, lr, s_bounds_log, tol, safety_cap, min_R_var)
    if setup is not None:
        y_c, var_c, offs, members, of_kp = (setup[k] for k in ('y_c', 'var_c', 'offs', 'members', 'of_kp'))
    else:
        y_c, var_c = P.cropped(s_frames)
        offs, members, of_kp = _block_csr(blocks, P.K)
    rconst = hip_ops.const_r(var_c, min_R_var)
    if setup is not None and not setup['loop'].prepared:
        setup['loop'].prepare()                              # the pass over y behind the median, in front of the host's wait
    nb = len(offs) - 1
    if s_mode == 'grid':
        cand = torch.exp(torch.linspace(lo, hi, n_grid, dtype=torch.float64, device=P.dev))
        if nb != P.K:                                   # blocks share one s: sum member losses
            nll = hip_ops.nll(y_c, rconst, *P.params, cand, flags=P.flags)
            blk = torch.zeros((nb, n_grid), dtype=torch.float64, device=P.dev)
            blk.index_add_(0, torch.as_tensor(of_kp, device=P.dev), nll)
            s_blk, idx = hip_ops.argmin_s(blk, cand)
            s = s_blk[torch.as_tensor(of_kp, device=P.dev)]
        else:                                           # table and argmin in one call (eks_nll_argmin)
            nll, s, idx = hip_ops.nll_argmin(y_c, rconst, *P.params, cand, flags=P.flags)
        return s, dict(mode='grid', nll=nll, argmin=idx, candidates=cand)
    # Adam on u = log s (reference eks/core.py:612-613, :439-441: float32 initial value)
    packed, state, s_kp, loop = (setup[k] for k in ('packed', 'state', 's_kp', 'loop'))
    loop.set_rconst(rconst)
    # (block means by one segmented sum over the CSR member list: a Python loop over 256 blocks of np.mean / np.clip
    #  calls held the first launch back by 2 ms)
    if callable(s_guess_per_k):
        s_guess_per_k = s_guess_per_k()
    g = np.asarray(s_guess_per_k, dtype=np.float64)[np.asarray(members, dtype=np.int64)]
    offs64 = np.asarray(offs, dtype=np.int64)
    if np.all(np.diff(offs64) == 1):
        means = g
    else:
        means = np.array([np.mean(g[offs64[b]:offs64[b + 1]]) for b in range(nb)])
    u0 = np.log(np.clip(means, 1e-6, 1e3)).astype(np.float32).astype(np.float64)
    # optimiser state and starting point in ONE upload (four small pageable copies cost ~25 us each in front of the
    # first launch), the block lists generated on the device when every block is a keypoint in order
    # (from page-locked memory and without waiting: a pageable copy would hold the host until the stream - the pass over
    #  y enqueued above included - had drained, and the search's launch would follow an idle gap)
    try:
        stage = _pinned_empty((nb * 6 + P.K,), torch.float64)
    except RuntimeError:          # page-locked memory exhausted or unavailable: a blocking copy
        stage = torch.empty(nb * 6 + P.K, dtype=torch.float64)
    packed_h = stage.numpy()
    packed_h[:] = 0.0
    packed_h[0:nb * 6:6] = u0
    packed_h[3:nb * 6:6] = np.inf
    packed_h[nb * 6:] = np.exp(np.clip(u0, lo, hi))[of_kp]
    packed.copy_(stage, non_blocking=True)
    if setup.get('prep_ev') is not None:
        torch.cuda.current_stream(P.dev).wait_event(setup['prep_ev'])
    iters, cap = 0, int(safety_cap)
    # several iterations per host round trip: a step enqueued after a block has stopped (or reached the
    # cap) leaves that block untouched (its loss waves exit at once), so over-issuing changes nothing.
    # The count of still-running blocks after round r is copied to pinned memory behind round r and read
    # only after round r + 1 has been enqueued: the device never waits for the host's answer, at the price
    # of one round of no-op launches after the last block stops.
    # Iterations per round: scalar chains skip the keypoints that have stopped (an over-issued iteration is a
    # launch that returns at once), the general path evaluates every keypoint whatever the optimiser state - a
    # round issued after the last block stopped costs a round of full evaluations - so its rounds are short.
    if sync_every is None:
        # the library knows which form the loop takes here (eks_adam_run_stride, include/eks_hip.h): one launch per
        # call where the loss kernel keeps its workgroups (long calls cost nothing on the device and spare host round
        # trips), one launch per iteration otherwise
        sync_every = loop.stride()
    if os.environ.get('EKS_ADAM_SYNC_EVERY'):            # (A/B runs)
        sync_every = max(1, int(os.environ['EKS_ADAM_SYNC_EVERY']))
    if sync_every >= 64:
        # One launch per call, and a call issued after the last block has stopped returns at once: every call the cap
        # allows is enqueued now, nothing is waited for, and whatever the caller enqueues next (the final smoothing pass)
        # follows the search without a host round trip in between.  (The search from cached lag sums asks for the whole
        # cap in one call: eks_adam_run_stride.)
        r = 0
        while iters < cap:
            n = min(sync_every, cap - iters)
            loop.run(n)
            iters += n
            r += 1
        return s_kp, dict(mode='adam', state=state, launches=iters, calls=r, deferred_count=True, _keep=stage)
    rounds = (cap + sync_every - 1) // sync_every
    try:
        snap = _pinned_empty((max(rounds, 1),), torch.int32)
    except RuntimeError:          # page-locked memory exhausted or unavailable (as _to_host): blocking reads
        snap = None
    pending = None                                        # (round index, event) of the newest unread count
    r = 0
    while iters < cap:
        n = min(sync_every, cap - iters)
        loop.run(n)
        iters += n
        if snap is None:
            left = int(loop.n_active.item())
            if left == 0:
                break
            continue
        snap[r:r + 1].copy_(loop.n_active, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        if pending is not None:
            pending[1].synchronize()
            left = int(snap[pending[0]])
            if left == 0:
                break
        pending = (r, ev)
        r += 1
    # `launches`: iterations ENQUEUED (the deferred count over-issues one round after the last block stops; on
    # scalar chains those launches return at once); the iterations each block actually took are state[:, 4]
    return s_kp, dict(mode='adam', state=state, launches=iters, deferred_count=snap is not None)


def optimize_smooth_param(ys, m0s, S0s, As, Cs, Qs, Rs, blocks, s_finals, s_frames, s_guess_per_k,
                          lr: float = 0.25, s_bounds_log=(-8.0, 8.0), tol: float = 1e-3,
                          safety_cap: int = 300, min_R_var: float = 1e-4,
                          h_fn_combined: Callable | None = None) -> None:
    """One s per block of keypoints by Adam on log s over the summed constant-R filter NLL
    (reference eks/core.py:306-559, :562-699).  `Rs` is the reference's (K,T,O,O) stack of
    diagonal R_t; only its diagonal is used.  Writes `s_finals` in place."""
    K = np.shape(ys)[0]
    if not blocks:
        blocks = [[k] for k in range(K)]
    Rd = np.diagonal(_to_numpy(Rs), axis1=-2, axis2=-1)                 # (K,T,O)
    if h_fn_combined is not None:       # calibrated projection: extended filter (reference :478-510)
        s, _, _ = _run_kalman_smoother_pinhole(
            ys, m0s, S0s, As, Qs, np.swapaxes(Rd, 0, 1), h_fn_combined, s_frames, None, blocks, lr,
            s_bounds_log, tol, safety_cap, 'adam', 0, True, True, None,
            guesses=np.asarray(s_guess_per_k, float), min_R_var=min_R_var, final_pass=False)
        s_finals[:] = s
        return
    P = _DeviceProblem(ys, m0s, S0s, As, Cs, Qs, np.swapaxes(Rd, 0, 1))
    s, info = _optimize_on_device(P, blocks, s_frames, np.asarray(s_guess_per_k, float), lr,
                                  s_bounds_log, tol, safety_cap, min_R_var, 'adam', 0)
    s_finals[:] = s.cpu().numpy()
    _log_opt(blocks, s_finals, info)


def _run_kalman_smoother_pinhole(ys, m0s, S0s, As, Qs, ensemble_vars, h_fn, s_frames, smooth_param,
                                 blocks, lr, s_bounds_log, tol, safety_cap, s_mode, n_grid, vs_diag,
                                 return_device, x_init, fd_step: float = 1e-3,
                                 lin_tol: float = 1e-10, max_sweeps: int = 16, guesses=None,
                                 min_R_var: float = 1e-4, final_pass: bool = True):
    """run_kalman_smoother with the calibrated multi-camera projection (reference eks/core.py:
    159-302 with h_fn; optimiser :562-699 / :306-559).  Extended filter = eks_ekf_smooth.  The
    reference differentiates the loss through the filter (jax.value_and_grad); here d NLL / d log s
    is a central difference over three chains per keypoint (u, u + h, u - h) that share the data,
    so one launch evaluates loss and gradient; each chain keeps its own linearisation points
    between iterations (warm start: 1-2 sweeps per iteration)."""
    from .calibration import PinholeProjection
    if not isinstance(h_fn, PinholeProjection):
        raise NotImplementedError(
            'h_fn must be an eks_amd.calibration.PinholeProjection (make_projection_from_camgroup): '
            'the HIP kernels cannot call a Python emission function')
    torch = _torch()
    dev = hip_ops.require_gpu()
    f64 = lambda a: torch.as_tensor(np.ascontiguousarray(_to_numpy(a, np.float64)), device=dev)
    m0, S0, A, Q = f64(m0s), f64(S0s), f64(As), f64(Qs)
    K = m0.shape[0]
    if m0.shape[1] != 3:
        raise ValueError('the calibrated path has a 3-D latent state')
    cams = torch.as_tensor(h_fn.cams, device=dev)
    if hasattr(ys, 'detach'):
        y = ys.to(dev, dtype=torch.float32).transpose(0, 1).contiguous()
        var = ensemble_vars.to(dev, dtype=torch.float32).contiguous()
    else:
        y = torch.as_tensor(np.ascontiguousarray(_to_numpy(ys)), device=dev).to(torch.float32)
        y = y.transpose(0, 1).contiguous()
        var = torch.as_tensor(np.ascontiguousarray(_to_numpy(ensemble_vars)), device=dev)
        var = var.to(torch.float32).contiguous()
    T, O = y.shape[0], y.shape[2]
    if T < 2:                       # reference eks/core.py:233-236 (initial guesses, unconditional)
        raise ValueError('Not enough frames to compute temporal differences.')
    if tuple(y.shape) != (T, K, O) or tuple(var.shape) != (T, K, O) or O != 2 * h_fn.n_cameras:
        raise ValueError(f'ys must be (K,T,2V) and ensemble_vars (T,K,2V); got {tuple(y.shape)} '
                         f'(frame-major) and {tuple(var.shape)} for {h_fn.n_cameras} cameras')
    if not blocks:
        blocks = [[k] for k in range(K)]
    if x_init is None:
        xlin = m0[:, None, :].expand(K, T, 3).contiguous()
    else:
        xlin = f64(x_init).reshape(K, T, 3).contiguous()
    worst = torch.zeros((), dtype=torch.float64, device=dev)

    def crop(a):
        if not s_frames or (len(s_frames) == 1 and s_frames[0] == (None, None)):
            return a, None
        if not isinstance(s_frames, list):
            raise TypeError('s_frames must be a list of (start, end) tuples or None.')
        idx = torch.cat([torch.arange(a0, b0, device=dev) for a0, b0 in frame_spans(T, s_frames)])
        return a.index_select(0, idx).contiguous(), idx

    s_finals = np.empty(K, dtype=float)
    info = {}
    if smooth_param is not None:
        s_finals[:] = float(smooth_param) if isinstance(smooth_param, (int, float)) \
            else np.asarray(smooth_param, dtype=float)
        s_dev = torch.as_tensor(s_finals, device=dev)
    else:
        y_c, idx = crop(y)
        var_c = var if idx is None else var.index_select(0, idx).contiguous()
        x_c = xlin if idx is None else xlin.index_select(1, idx).contiguous()
        rconst = hip_ops.const_r(var_c, min_R_var)
        lo, hi = float(s_bounds_log[0]), float(s_bounds_log[1])
        offs, members, of_kp = _block_csr(blocks, K)
        nb = len(blocks)
        if s_mode == 'grid':
            cand = torch.exp(torch.linspace(lo, hi, n_grid, dtype=torch.float64, device=dev))
            rep = lambda a: a.repeat((n_grid,) + (1,) * (a.dim() - 1))
            xg = x_c.repeat(n_grid, 1, 1)
            _, _, nll, inf = hip_ops.ekf_smooth(y_c, None, rconst, rep(m0), rep(S0), rep(A), rep(Q),
                                                cand.repeat_interleave(K), cams, xg, max_sweeps,
                                                lin_tol, want_smoother=False)
            worst = torch.maximum(worst, inf[1])
            nll = nll.view(n_grid, K).transpose(0, 1).contiguous()
            blk = torch.zeros((nb, n_grid), dtype=torch.float64, device=dev)
            blk.index_add_(0, torch.as_tensor(of_kp, device=dev), nll)
            s_blk, amin = hip_ops.argmin_s(blk, cand)
            s_dev = s_blk[torch.as_tensor(of_kp, device=dev)].contiguous()
            info = dict(mode='grid', nll=nll, argmin=amin, candidates=cand)
        else:
            if guesses is None:
                ev_host = _to_numpy(ensemble_vars)[:2000] if not hasattr(ensemble_vars, 'detach') \
                    else ensemble_vars[:2000].detach().cpu().numpy()
                guesses = _initial_guesses_per_keypoint(ev_host)
            u0 = np.array([np.float32(np.log(np.clip(np.mean([guesses[k] for k in b]), 1e-6, 1e3)))
                           for b in blocks], dtype=np.float64)
            state = np.zeros((nb, 6))
            state[:, 0] = u0
            state[:, 3] = np.inf
            state = torch.as_tensor(state, device=dev)
            offs_d, mem_d = torch.as_tensor(offs, device=dev), torch.as_tensor(members, device=dev)
            s_kp = torch.as_tensor(np.exp(np.clip(u0, lo, hi))[of_kp], device=dev)
            n_active = torch.zeros(1, dtype=torch.int32, device=dev)
            rep3 = lambda a: a.repeat((3,) + (1,) * (a.dim() - 1))
            m3, S3, A3, Q3 = rep3(m0), rep3(S0), rep3(A), rep3(Q)
            x3 = x_c.repeat(3, 1, 1)
            step = torch.tensor([1.0, np.exp(fd_step), np.exp(-fd_step)], dtype=torch.float64,
                                device=dev)
            iters, cap = 0, int(safety_cap)
            while iters < cap:
                for _ in range(min(8, cap - iters)):
                    s3 = (step[:, None] * s_kp[None, :]).reshape(-1).contiguous()
                    # warm-started iterations settle in 1-3 sweeps; every enqueued-but-gated sweep
                    # still costs its four empty launches
                    _, _, nll3, inf = hip_ops.ekf_smooth(y_c, None, rconst, m3, S3, A3, Q3, s3, cams, x3,
                                                         max_sweeps if iters == 0 else min(max_sweeps, 8),
                                                         lin_tol, want_smoother=False)
                    worst = torch.maximum(worst, inf[1])
                    nll3 = nll3.view(3, K)
                    dnll = ((nll3[1] - nll3[2]) / (2.0 * fd_step)).contiguous()
                    hip_ops.adam_step(offs_d, mem_d, nll3[0].contiguous(), dnll, state, s_kp, n_active,
                                      lr, lo, hi, tol, cap)
                    iters += 1
                if int(n_active.item()) == 0:
                    break
            s_dev = s_kp
            info = dict(mode='adam', state=state, launches=iters)
            if idx is None:
                xlin = x3[:K].contiguous()          # warm start of the final pass
        s_finals[:] = s_dev.cpu().numpy()
        _log_opt(blocks, s_finals, info)
    if not final_pass:                  # optimize_smooth_param only wants s (reference :306-559)
        _warn_if_unconverged(float(worst.item()), lin_tol, max_sweeps)
        return s_finals, None, None
    ms, Vs, _, inf = hip_ops.ekf_smooth(y, var, None, m0, S0, A, Q, s_dev.contiguous(), cams, xlin,
                                        max_sweeps, lin_tol, want_smoother=True, vs_diag=vs_diag)
    _warn_if_unconverged(float(torch.maximum(worst, inf[1]).item()), lin_tol, max_sweeps)
    if return_device:
        return s_finals, ms.transpose(0, 1), Vs.transpose(0, 1)
    ms_h, Vs_h = _to_host(ms, Vs)
    return s_finals, np.swapaxes(ms_h, 0, 1), np.swapaxes(Vs_h, 0, 1)


def _warn_if_unconverged(worst: float, lin_tol: float, max_sweeps: int) -> None:
    if not worst <= lin_tol:
        logger.warning(f'extended filter: linearisation points still moving by {worst:.2e} after '
                       f'{max_sweeps} sweeps (tolerance {lin_tol:.0e}); the result may differ from '
                       'the sequential extended Kalman filter')


def _log_opt(blocks, s_finals, info) -> None:
    if not logger.isEnabledFor(logging.DEBUG) or info.get('mode') != 'adam':
        return
    st = info['state'].cpu().numpy()
    if blocks is None:
        blocks = [[k] for k in range(st.shape[0])]
    for b, blk in enumerate(blocks):
        logger.debug(f'[opt s | block {list(blk)}] s={s_finals[blk[0]]:.6g}, '
                     f'iters={int(st[b, 4])}, NLL={st[b, 3]:.6f}')


_TILE_MIN_BYTES = int(os.environ.get('EKS_HOST_TILE_MIN_BYTES', 96 << 20))     # below this one untiled call is as fast
_TILE_TARGET_BYTES = int(os.environ.get('EKS_HOST_TILE_BYTES', 128 << 20))     # transfer volume of one tile (in + out)


_TILE_STREAMS: dict = {}


def _tile_streams(dev):
    """Four side streams per device and calling thread (three for the tiles' pipeline, one for the variances' upload), created once: torch's caching allocator keeps a pool per stream, so streams
    made per call would hipMalloc every tile's buffers afresh (measured: 200 - 400 ms per call instead of 15)."""
    key = (dev.index if dev.index is not None else _torch().cuda.current_device(), threading.get_ident())
    with _state_lock:                      # (a set per device AND calling thread: two threads' pipelines do not share)
        if key not in _TILE_STREAMS:
            _TILE_STREAMS[key] = [_torch().cuda.Stream(device=dev) for _ in range(4)]
        return _TILE_STREAMS[key]


_PREPARE_STREAMS: dict = {}


_IDENTITY_BLOCKS: dict = {}


def _identity_blocks(K, dev):
    """(offsets 0..K, members 0..K-1) of the reference's default blocks on the device, made once per K and device (two
    torch.arange launches in front of every search otherwise); read-only to every caller."""
    key = (int(K), str(dev))
    hit = _IDENTITY_BLOCKS.get(key)
    if hit is None:
        torch = _torch()
        ar = torch.arange(K + 1, dtype=torch.int32, device=dev)
        hit = _IDENTITY_BLOCKS[key] = (ar, ar[:K])
        if len(_IDENTITY_BLOCKS) > 16:
            _IDENTITY_BLOCKS.pop(next(iter(_IDENTITY_BLOCKS)))
    return hit


def _prepare_stream(dev):
    """The side stream of the search's pass over y (_adam_setup), one per device and calling thread, created once."""
    key = (dev.index if dev.index is not None else _torch().cuda.current_device(), threading.get_ident())
    with _state_lock:
        if key not in _PREPARE_STREAMS:
            _PREPARE_STREAMS[key] = _torch().cuda.Stream(device=dev)
        return _PREPARE_STREAMS[key]


_TILE_ADAM = False


def _host_tiles(ys, ensemble_vars, K, T, O, D, vs_diag, blocks, h_fn, return_device, searching_adam=False):
    """How to cut a host-array call into keypoint tiles for the pipelined boundary, or None.  Keypoints are
    independent (reference eks/core.py:223-224, :293), so any partition gives the same numbers; tiling needs plain
    NumPy inputs (a device tensor has nothing to upload), singleton blocks and enough bytes to be worth it."""
    if return_device or h_fn is not None or os.environ.get('EKS_HOST_UNTILED'):
        return None
    # The Adam search is ~100 dependent launches of ~45 us whatever the number of keypoints: run per tile it is paid per
    # tile (measured on BASELINE configs[2]: 88 ms tiled against 28.5 ms untiled), so that mode stays one call.
    if searching_adam and not _TILE_ADAM:
        return None
    if hasattr(ys, 'detach') or hasattr(ensemble_vars, 'detach'):
        return None
    if blocks and any(len(b) != 1 for b in blocks):
        return None
    if blocks:
        _block_csr(blocks, K)                 # (what the untiled call would refuse - a list that does not partition the
                                              #  keypoints - is refused here too, before anything is cut)
    per_kp = T * (2 * O + D + (D if vs_diag else D * D)) * 4
    if K * per_kp < _TILE_MIN_BYTES or K < 4:
        return None
    kt = max(1, min(K // 2, int(round(_TILE_TARGET_BYTES / per_kp))))
    if kt >= 32:
        kt = kt // 32 * 32                    # whole 64-chain tiles of the scalar-chain kernels (D = 2)
    if os.environ.get('EKS_HOST_TILE_KP'):
        kt = int(os.environ['EKS_HOST_TILE_KP'])
        return [(k0, min(K, k0 + kt)) for k0 in range(0, K, kt)]
    # (measured on BASELINE configs[2], tools/host_boundary_ab.py: equal tiles of 8 / 16 / 32 / 43 / 64 keypoints 30 / 21 /
    #  14.6 / 15.6 / 15.2 ms - every tile costs ~1 ms of Python and launches on the host, and a first tile twice as large
    #  delays the first download by what it takes to bring it up; tiles that grow 1.3x from 32: 17.6 ms - the three
    #  streams then serialise a large tile behind the download of the third before it)
    return [(k0, min(K, k0 + kt)) for k0 in range(0, K, kt)]


def _run_tiled_from_host(tiles, ys, m0s, S0s, As, Cs, Qs, ensemble_vars, s_frames, smooth_param, lr, s_bounds_log,
                         tol, safety_cap, s_mode, n_grid, vs_diag, return_info):
    """run_kalman_smoother on HOST arrays as a three-stage pipeline over keypoint tiles: while tile i is in the
    kernels, tile i + 1 is on its way up and tile i - 1's results are on their way down (PCIe is full duplex;
    the untiled call moved 16 B per unit up, ran, then moved 24 B per unit down: 19.6 ms on BASELINE configs[2]
    for 0.55 ms of kernels).  Each tile is an ordinary device-tensor call on one of three streams - upload, search,
    smooth, transposition to the reference's (K, T, .) layout, download into its slab of ONE page-locked result
    buffer - so the outputs are the untiled call's bit for bit (keypoints are independent, eks/core.py:293).
    Measured on BASELINE configs[2] (tools/host_path_time.py): 19.4 -> see profiles/r04_*_host_path_time.txt."""
    torch = _torch()
    dev = hip_ops.require_gpu()
    ys_h = np.asarray(ys)
    ev_h = np.asarray(ensemble_vars)
    K, T, O = ys_h.shape
    par = {k: np.asarray(_to_numpy(v, np.float64)) for k, v in dict(m0=m0s, S0=S0s, A=As, C=Cs, Q=Qs).items()}
    D = par['m0'].shape[1]
    if tuple(ev_h.shape) != (T, K, O):
        raise ValueError(f'ys must be (K,T,O) and ensemble_vars (T,K,O); got {tuple(ys_h.shape)} and {tuple(ev_h.shape)}')
    if T < 2:
        raise ValueError('Not enough frames to compute temporal differences.')
    t_entry = time.perf_counter()
    # The ensemble variances arrive (T, K, O), frame-major: a keypoint tile of them is a strided view on the host.
    # numpy gathers one in 1.6 - 5 ms per 25 MB (three to ten times its transfer), so round 4 sent the whole array up
    # before the first tile started (3.6 ms of BASELINE configs[2]'s 16.6); the library's own threaded gather
    # (eks_host_gather_cols) makes a tile contiguous in a page-locked buffer in a fraction of that, and the variances
    # travel tile by tile like the observations - the first kernels start after ONE tile's uploads.
    import ctypes
    from . import _lib
    lib = _lib.load()
    cur = torch.cuda.current_stream(dev)
    streams = _tile_streams(dev)
    for st in streams:
        st.wait_stream(cur)
    ev_c = ev_h if ev_h.flags['C_CONTIGUOUS'] else np.ascontiguousarray(ev_h)
    ev_dtype = torch.as_tensor(ev_c[:1, :1]).dtype
    n_thr = max(1, min(32, (os.cpu_count() or 1) // 2))
    row_bytes = K * O * ev_c.itemsize

    def var_tile(k0, k1):
        """ensemble_vars[:, k0:k1] as a contiguous host tensor (page-locked when possible)"""
        try:
            buf = torch.empty((T, k1 - k0, O), dtype=ev_dtype, pin_memory=True)
        except RuntimeError:
            buf = torch.empty((T, k1 - k0, O), dtype=ev_dtype)
        rc = lib.eks_host_gather_cols(ctypes.c_void_p(ev_c.ctypes.data), T, row_bytes, k0 * O * ev_c.itemsize,
                                      (k1 - k0) * O * ev_c.itemsize, ctypes.c_void_p(buf.data_ptr()), n_thr)
        _lib.check(rc, 'eks_host_gather_cols')
        return buf

    work = streams[:-1]
    # EKS_HOST_VAR_WHOLE=1: round 4's form (the whole array up before the first tile), kept for A/B runs
    whole = bool(os.environ.get('EKS_HOST_VAR_WHOLE'))
    trace = [] if os.environ.get('EKS_HOST_TILE_TRACE') else None
    if whole:
        with torch.cuda.stream(streams[-1]):
            ev_d = torch.as_tensor(ev_c, device=dev)
            ev_ready = torch.cuda.Event()
            ev_ready.record(streams[-1])
    import time as _time
    # the variance tiles are made contiguous AHEAD of the loop below, on a helper thread (the gather releases the GIL):
    # the calling thread's own per-tile work - the pageable upload of ys, the launches - does not wait for it
    import queue
    vq: queue.Queue = queue.Queue(maxsize=3)

    stop = threading.Event()                 # set when the consumer leaves early: the producer must not wait on a full queue

    def offer(item):
        while not stop.is_set():
            try:
                vq.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def produce():
        try:
            for (a, b) in tiles:
                if not offer(var_tile(a, b)):
                    return
        except BaseException as e:          # noqa: BLE001 - handed to the consumer
            offer(e)

    if not whole:
        threading.Thread(target=produce, name='eks-var-tiles', daemon=True).start()
    vshape = (K, T, D) if vs_diag else (K, T, D, D)
    nbytes = (K * T * D + int(np.prod(vshape))) * 4
    pinned = not os.environ.get('EKS_PAGEABLE_D2H') and nbytes <= (2 << 30) and _pinned_reserve(nbytes)
    try:
        ms_h = torch.empty((K, T, D), dtype=torch.float32, pin_memory=pinned)
        Vs_h = torch.empty(vshape, dtype=torch.float32, pin_memory=pinned)
    except RuntimeError:
        if pinned:
            _release_pinned(nbytes)
        pinned = False
        ms_h = torch.empty((K, T, D), dtype=torch.float32)
        Vs_h = torch.empty(vshape, dtype=torch.float32)
    sp = None if smooth_param is None or isinstance(smooth_param, (int, float)) else \
        np.broadcast_to(np.asarray(smooth_param, dtype=float), (K,))
    flags = hip_ops.model_flags(par['S0'], par['A'], par['C'], par['Q'])      # once, for every tile
    s_parts, infos = [], []
    t_loop = time.perf_counter()
    try:
        for i, (k0, k1) in enumerate(tiles):
            st = work[i % len(work)]
            with torch.cuda.stream(st):
                t_a = _time.perf_counter()
                y_t = torch.as_tensor(np.ascontiguousarray(ys_h[k0:k1]), device=dev)           # (Kt,T,O), caller's dtype
                t_b = _time.perf_counter()
                if whole:
                    st.wait_event(ev_ready)
                    v_t = ev_d[:, k0:k1]
                    ev_d.record_stream(st)
                else:
                    v_host = vq.get()
                    if isinstance(v_host, BaseException):
                        raise v_host
                    t_c = _time.perf_counter()
                    v_t = v_host.to(dev, non_blocking=True)                                     # (T,Kt,O)
                    if trace is not None:
                        trace.append((k0, k1, (t_b - t_a) * 1e3, (t_c - t_b) * 1e3, (_time.perf_counter() - t_c) * 1e3))
                res = run_kalman_smoother(
                    y_t, par['m0'][k0:k1], par['S0'][k0:k1], par['A'][k0:k1], par['C'][k0:k1], par['Q'][k0:k1], v_t,
                    s_frames=s_frames, smooth_param=(smooth_param if sp is None else list(sp[k0:k1])), blocks=None,
                    lr=lr, s_bounds_log=s_bounds_log, tol=tol, safety_cap=safety_cap, s_mode=s_mode, n_grid=n_grid,
                    vs_diag=vs_diag, return_device=True, return_info=True, _s_on_device=True, _model_flags=flags)
                s_dev, ms_d, Vs_d, info = res
                ms_c, Vs_c = ms_d.contiguous(), Vs_d.contiguous()    # (Kt,T,D) views of the frame-major buffers ->
                ms_h[k0:k1].copy_(ms_c, non_blocking=True)           # the reference's layout on the device, then down
                Vs_h[k0:k1].copy_(Vs_c, non_blocking=True)
                s_parts.append(s_dev)
                infos.append({k: v for k, v in info.items()})
                # nothing of the tile is kept alive past this point: the copies above are enqueued on `st`, and the
                # caching allocator hands a block back only to work enqueued on the same stream later (a call just over
                # the tiling threshold then peaks at a few tiles' worth of device memory, not at the untiled footprint)
                del y_t, v_t, res, ms_d, Vs_d, ms_c, Vs_c
    except BaseException:
        if pinned:                               # (the finalisers below were never attached: give the bytes back)
            _release_pinned(nbytes)
        raise
    finally:
        stop.set()                               # a producer still gathering stops at its next tile ...
        while True:                              # ... and the tiles it had queued (page-locked buffers) are dropped
            try:
                vq.get_nowait()
            except queue.Empty:
                break
        for st in streams:                       # (also when a tile raised: the side streams rejoin the caller's)
            cur.wait_stream(st)
    cur.synchronize()
    if trace:
        logger.warning(f'tiled boundary: {(t_loop - t_entry) * 1e3:.2f} ms before the first tile, {(time.perf_counter() - t_loop) * 1e3:.2f} ms '
                       'to the end of the downloads; per tile (k0, k1, ys upload ms, variance gather ms, variance upload enqueue ms): '
                       + '; '.join(f'{a}-{b}: {u:.2f} {g:.2f} {e:.2f}' for a, b, u, g, e in trace))
    s_finals = np.concatenate([np.asarray(s.cpu().numpy(), dtype=float) for s in s_parts])
    out_ms, out_Vs = ms_h.numpy(), Vs_h.numpy()
    if pinned:
        import weakref
        for arr, h in ((out_ms, ms_h), (out_Vs, Vs_h)):
            n = h.numel() * h.element_size()
            weakref.finalize(arr.base if arr.base is not None else h, _release_pinned, n)     # (reserved above)
    out = (s_finals, out_ms, out_Vs)
    if return_info:
        out = out + (dict(mode='tiled', tiles=list(tiles), tile_info=infos),)
    if infos and infos[0].get('mode') == 'adam':
        _log_opt([[k] for k in range(K)], s_finals,
                 dict(mode='adam', state=torch.cat([inf['state'] for inf in infos])))
    return out


def run_kalman_smoother(ys, m0s, S0s, As, Cs, Qs, ensemble_vars, s_frames: list | None = None,
                        smooth_param: float | list | None = None,
                        blocks: list[list[int]] | None = None, lr: float = 0.25,
                        s_bounds_log: tuple = (-8.0, 8.0), tol: float = 1e-2, safety_cap: int = 300,
                        h_fn: Callable | None = None, *, s_mode: str = 'adam', n_grid: int = 64,
                        vs_diag: bool = False, return_device: bool = False, x_init=None,
                        return_info: bool = False, _s_on_device: bool = False, _model_flags: int | None = None):
    """Choose (or optimise) the process-noise scale s per keypoint, then run the Kalman filter +
    RTS smoother.  Drop-in for the reference's eks/core.py:159-302.

    ys (K,T,O); m0s (K,D); S0s, As, Qs (K,D,D); Cs (K,O,D); ensemble_vars (T,K,O) (note T-major);
    R_{k,t} = diag(clip(ensemble_vars[t,k], 1e-12)).  Returns (s_finals float64 (K,), ms float32
    (K,T,D), Vs float32 (K,T,D,D)) as NumPy arrays; ms / Vs are transposed views of the kernels'
    frame-major buffers.

    Extensions (keyword-only, not in the reference): s_mode 'adam' (reference behaviour) or 'grid'
    (n_grid candidates exp(linspace(*s_bounds_log)), BASELINE.json config 3); vs_diag returns only
    the diagonal of Vs as (K,T,D); return_device keeps ms / Vs as device tensors; return_info appends the search's
    record (mode, optimiser state / NLL table as device tensors, launches) as a fourth element.

    h_fn: the nonlinear observation model y_t = h_fn(x_t) + v_t (reference :188-190).  The
    accelerated path takes a `calibration.PinholeProjection` (what
    `make_projection_from_camgroup` returns - the only h_fn the reference itself constructs);
    the kernels cannot call back into Python, so any other callable raises NotImplementedError.
    `Cs` is ignored then, D must be 3 and O = 2 * n_cameras.  x_init (K,T,3), optional: a first
    guess of the states (e.g. the triangulated points) that the linearisation starts from.
    """
    if s_mode not in ('adam', 'grid'):
        raise ValueError("s_mode must be 'adam' or 'grid'")
    if h_fn is not None:
        res = _run_kalman_smoother_pinhole(ys, m0s, S0s, As, Qs, ensemble_vars, h_fn, s_frames,
                                           smooth_param, blocks, lr, s_bounds_log, tol, safety_cap,
                                           s_mode, n_grid, vs_diag, return_device, x_init)
        return res + ({},) if return_info else res
    torch = _torch()
    t0 = time.perf_counter()
    if not return_device and not hasattr(ys, 'detach'):
        shp = np.shape(ys)
        if len(shp) == 3:
            tiles = _host_tiles(ys, ensemble_vars, shp[0], shp[1], shp[2], np.shape(m0s)[1], vs_diag, blocks, h_fn,
                                return_device, searching_adam=smooth_param is None and s_mode == 'adam')
            if tiles and len(tiles) > 1:
                return _run_tiled_from_host(tiles, ys, m0s, S0s, As, Cs, Qs, ensemble_vars, s_frames, smooth_param,
                                            lr, s_bounds_log, tol, safety_cap, s_mode, n_grid, vs_diag, return_info)
    P = _DeviceProblem(ys, m0s, S0s, As, Cs, Qs, ensemble_vars, flags=_model_flags)
    K = P.K
    if not blocks:
        blocks = None                  # (= [[k] for k in range(K)], the reference's default: _block_csr's fast path)
    logger.debug(f'correlated keypoint blocks: {blocks if blocks else "every keypoint its own"}')
    logger.debug(f'[profile]   build_R: {time.perf_counter() - t0:.3f}s')   # upload; R is never built

    # the reference computes the initial guesses before it looks at smooth_param (eks/core.py:
    # 233-236), so fewer than two frames raise whether or not s is given
    if P.T < 2:
        raise ValueError('Not enough frames to compute temporal differences.')
    s_finals = np.empty(K, dtype=float)
    info = {}
    if smooth_param is not None:
        if isinstance(smooth_param, (int, float)):
            s_finals[:] = float(smooth_param)
        else:
            s_finals[:] = np.asarray(smooth_param, dtype=float)
        s_dev = torch.as_tensor(s_finals, device=P.dev)
    else:
        t1 = time.perf_counter()
        guesses = np.full(K, 2.0)
        # the search's buffers and its pass over y FIRST (nothing it needs waits for the guesses or for eks_const_r)
        setup = _adam_setup(P, blocks, s_frames, lr, s_bounds_log, tol, safety_cap, 1e-4) if s_mode == 'adam' else None
        if s_mode == 'adam':            # the starting point of the optimiser (reference :233-236)
            if hasattr(ensemble_vars, 'detach') and ensemble_vars.is_cuda and ensemble_vars.dtype == torch.float32:
                ev_d = ensemble_vars.detach()
                sd = _guess_std_on_device(ev_d, lazy=True)
                if sd is None:
                    guesses = _initial_guesses_per_keypoint(rows=_guess_rows_from_device(ev_d))
                else:       # read back (and rounded, on the host) once eks_const_r has been enqueued behind the reduction
                    guesses = lambda: _initial_guesses_per_keypoint(sd=sd())      # noqa: E731
            else:
                src_f32 = (ensemble_vars.dtype == torch.float32) if hasattr(ensemble_vars, 'detach') \
                    else (getattr(ensemble_vars, 'dtype', None) == np.float32)
                sd = _guess_std_on_device(P.var, lazy=True) if src_f32 else None
                if sd is not None:
                    # float32 variances from the host: the copy on the device holds the same values, and the device
                    # reduction is numpy's float32 summation bit for bit (np.nanstd over 256 x 4 000 cost 2-3 ms here)
                    guesses = lambda: _initial_guesses_per_keypoint(sd=sd())      # noqa: E731
                else:
                    ev_host = _to_numpy(ensemble_vars)[:2000] if not hasattr(ensemble_vars, 'detach') \
                        else ensemble_vars[:2000].detach().cpu().numpy()
                    guesses = _initial_guesses_per_keypoint(ev_host)
        s_dev, info = _optimize_on_device(P, blocks, s_frames, guesses, lr, s_bounds_log, tol,
                                          safety_cap, 1e-4, s_mode, n_grid, setup=setup)
        if not _s_on_device:                     # (the tiled boundary reads s once, after the last tile is enqueued)
            # s travels to page-locked memory behind the search and is read after the final pass has been enqueued:
            # the device goes from the search straight into the smoother, the host wakes up beside it
            try:
                s_host = _pinned_empty((K,), torch.float64)
                s_host.copy_(s_dev, non_blocking=True)
                s_ready = torch.cuda.Event()
                s_ready.record()
            except RuntimeError:
                s_host = None
        logger.debug(f'[profile]   optimize_smooth_param (enqueued): {time.perf_counter() - t1:.3f}s')

    t2 = time.perf_counter()
    ms, Vs = hip_ops.smooth(P.y, P.var, *P.params, s_dev.contiguous(), flags=P.flags, vs_diag=vs_diag)
    if smooth_param is None and not _s_on_device:
        if s_host is not None:
            s_ready.synchronize()
            s_finals[:] = s_host.numpy()
        else:
            s_finals[:] = s_dev.cpu().numpy()
        _log_opt(blocks, s_finals, info)
    if return_device:
        out = (s_dev if _s_on_device else s_finals), ms.transpose(0, 1), Vs.transpose(0, 1)
    else:
        ms_h, Vs_h = _to_host(ms, Vs)
        out = s_finals, np.swapaxes(ms_h, 0, 1), np.swapaxes(Vs_h, 0, 1)
    logger.debug(f'[profile]   final smoother pass ({K} keypoints): {time.perf_counter() - t2:.3f}s')
    return out + (info,) if return_info else out
