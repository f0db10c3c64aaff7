"""ctypes wrapper of oracle/eks_oracle.c (float64 C twin of eks_oracle.py).  TEST INFRASTRUCTURE
ONLY (tests, smoke(), bench.py's cpu_baseline leg)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libeks_oracle.so')
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, 'eks_oracle.c')
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        # -march=native is decided on the machine that runs it: rebuild there if the ISA differs
        subprocess.run(['make', '-C', HERE, '-B', 'libeks_oracle.so'], check=True,
                       capture_output=True)
    return LIB


def load():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = ctypes.CDLL(LIB)
            _lib.eksc_max_threads()
        except OSError:
            build(force=True)
            _lib = ctypes.CDLL(LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def max_threads() -> int:
    return int(load().eksc_max_threads())


def smooth(y, Rd, m0, S0, A, C, Q, s, nthreads: int = 0):
    """y (K,T,O), Rd (K,T,O) or (K,O).  Returns ms (K,T,D), Vs (K,T,D,D), nll (K,)."""
    lib = load()
    y, Rd, m0, S0, A, C, Q = map(_c, (y, Rd, m0, S0, A, C, Q))
    K, T, O = y.shape
    D = m0.shape[-1]
    s = _c(np.broadcast_to(s, (K,)))
    ms = np.empty((K, T, D))
    Vs = np.empty((K, T, D, D))
    nll = np.empty(K)
    rc = lib.eksc_smooth(K, T, D, O, _p(y), _p(Rd), int(Rd.ndim == 2), _p(m0), _p(S0), _p(A), _p(C),
                         _p(Q), _p(s), _p(ms), _p(Vs), _p(nll), int(nthreads))
    if rc:
        raise RuntimeError(f'eksc_smooth rc={rc}')
    return ms, Vs, nll


def nll_grid(y, Rc, m0, S0, A, C, Q, s_cand, nthreads: int = 0):
    lib = load()
    y, Rc, m0, S0, A, C, Q, s_cand = map(_c, (y, Rc, m0, S0, A, C, Q, s_cand))
    K, T, O = y.shape
    D = m0.shape[-1]
    out = np.empty((K, len(s_cand)))
    rc = lib.eksc_nll_grid(K, T, D, O, _p(y), _p(Rc), _p(m0), _p(S0), _p(A), _p(C), _p(Q),
                           _p(s_cand), len(s_cand), _p(out), int(nthreads))
    if rc:
        raise RuntimeError(f'eksc_nll_grid rc={rc}')
    return out


def smooth_diag(y, Rd, m0, S0, A, C, Q, s, nthreads: int = 0):
    """The diagonal model (A, C, Q, S0 diagonal, D == O) as scalar chains: y (K,T,D), Rd (K,T,D) or (K,D).
    Returns ms (K,T,D), Vd (K,T,D) (diagonal of the covariance), nll (K,)."""
    lib = load()
    y, Rd, m0, S0, A, C, Q = map(_c, (y, Rd, m0, S0, A, C, Q))
    K, T, D = y.shape
    s = _c(np.broadcast_to(s, (K,)))
    ms = np.empty((K, T, D))
    Vd = np.empty((K, T, D))
    nll = np.empty(K)
    rc = lib.eksc_smooth_diag(K, T, D, _p(y), _p(Rd), int(Rd.ndim == 2), _p(m0), _p(S0), _p(A), _p(C), _p(Q), _p(s),
                              _p(ms), _p(Vd), _p(nll), int(nthreads))
    if rc:
        raise RuntimeError(f'eksc_smooth_diag rc={rc}')
    return ms, Vd, nll


def nll_grid_diag(y, Rc, m0, S0, A, C, Q, s_cand, nthreads: int = 0):
    lib = load()
    y, Rc, m0, S0, A, C, Q, s_cand = map(_c, (y, Rc, m0, S0, A, C, Q, s_cand))
    K, T, D = y.shape
    out = np.empty((K, len(s_cand)))
    rc = lib.eksc_nll_grid_diag(K, T, D, _p(y), _p(Rc), _p(m0), _p(S0), _p(A), _p(C), _p(Q), _p(s_cand),
                                len(s_cand), _p(out), int(nthreads))
    if rc:
        raise RuntimeError(f'eksc_nll_grid_diag rc={rc}')
    return out


def nll_directional(y, Rd, m0, S0, A, C, Q, dA, dQ):
    """One chain: y (T,O), Rd (T,O) or (O,), directions dA, dQ (n_dir,D,D).  Returns
    (nll, dnll (n_dir,)) - complex-step derivatives of the C filter."""
    lib = load()
    y, Rd, m0, S0, A, C, Q, dA, dQ = map(_c, (y, Rd, m0, S0, A, C, Q, dA, dQ))
    T, O = y.shape
    D = m0.shape[-1]
    n_dir = dA.shape[0]
    nll = np.empty(1)
    dnll = np.empty(max(n_dir, 1))
    rc = lib.eksc_nll_directional(T, D, O, _p(y), _p(Rd), int(Rd.ndim == 1), _p(m0), _p(S0), _p(A),
                                  _p(C), _p(Q), _p(dA), _p(dQ), n_dir, _p(nll), _p(dnll))
    if rc:
        raise RuntimeError(f'eksc_nll_directional rc={rc}')
    return float(nll[0]), dnll[:n_dir].copy()
