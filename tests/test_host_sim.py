"""CPU: the float32 lane arithmetic of the scalar-chain kernels, compiled for the host from the
SAME headers the gfx950 kernels include (eks_amd/csrc/eks_diag_lane.hpp, eks_nll_lane.hpp), against
the float64 oracle.  This pins the chunked-scan / three-regime numerics without a GPU; the kernels
themselves are tested under -m gpu.  (tests/host_sim is test infrastructure: eks_amd never loads it.)"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import eks_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def sim():
    src = os.path.join(ROOT, 'tests', 'host_sim', 'diag_sim.cpp')
    lib = os.path.join(ROOT, 'tests', 'host_sim', 'libdiag_sim.so')
    subprocess.run(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'eks_amd', 'csrc'),
                    src, '-o', lib], check=True)
    return ctypes.CDLL(lib)


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _problem(T, K, seed):
    from eks_amd import synth
    arrs = orc.singlecam_arrays(synth.singlecam_markers(T, K, seed=seed))
    y = np.ascontiguousarray(np.transpose(arrs['ys'], (1, 0, 2)).reshape(T, 2 * K).astype(np.float32))
    var = np.ascontiguousarray(arrs['ensemble_vars'].reshape(T, 2 * K).astype(np.float32))
    for k in ('m0s', 'S0s', 'As', 'Cs', 'Qs'):
        arrs[k] = np.ascontiguousarray(arrs[k], dtype=np.float64)
    ys64 = np.transpose(y.reshape(T, K, 2), (1, 0, 2)).astype(np.float64)
    ev64 = np.swapaxes(var.reshape(T, K, 2).astype(np.float64), 0, 1)
    return arrs, y, var, ys64, ev64


@pytest.mark.parametrize('B', [8, 32, 64])
@pytest.mark.parametrize('sval', [np.exp(-8.0), 0.3, 10.0, np.exp(8.0)])
def test_float32_chunked_smoother_matches_oracle(sim, B, sval):
    T, K = 1500, 5
    arrs, y, var, ys64, ev64 = _problem(T, K, seed=2)
    s = np.full(K, sval)
    ms = np.empty((T, 2 * K), np.float32)
    Vd = np.empty((T, 2 * K), np.float32)
    f, d = ctypes.c_float, ctypes.c_double
    rc = sim.sim_diag_smooth(T, 2 * K, 2, B, 1, _p(y, f), _p(var, f), _p(arrs['m0s'], d), _p(arrs['S0s'], d),
                             _p(arrs['As'], d), _p(arrs['Cs'], d), _p(arrs['Qs'], d), _p(s, d), _p(ms, f), _p(Vd, f))
    assert rc == 0
    ms_o, Vs_o, _ = orc.kalman_smoother(ys64, arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'],
                                        s, orc.build_R_from_vars(ev64))
    ms_k = np.transpose(ms.reshape(T, K, 2), (1, 0, 2))
    Vd_k = np.transpose(Vd.reshape(T, K, 2), (1, 0, 2))
    assert (np.abs(ms_k - ms_o) / np.abs(ms_o).max(axis=(1, 2), keepdims=True)).max() < 2e-6
    Vo = np.diagonal(Vs_o, axis1=2, axis2=3)
    assert (np.abs(Vd_k - Vo) / Vo).max() < 3e-6


@pytest.mark.parametrize('T,BN', [(3003, 512), (3003, 2048), (600, 4096)])
def test_float32_three_regime_nll_and_dual_gradient(sim, T, BN):
    K = 3
    arrs, y, var, ys64, ev64 = _problem(T, K, seed=5)
    Rc = orc.constant_R_from_timevarying(orc.build_R_from_vars(ev64))
    cand = np.exp(np.linspace(-8, 8, 16))
    ref = [orc.filter_nll(ys64, arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'],
                          np.full(K, sc), Rc, want_grad=True) for sc in cand]
    ref_nll = np.stack([r[0] for r in ref], axis=1)
    ref_g = np.stack([r[1] for r in ref], axis=1)
    rconst = np.ascontiguousarray(Rc.reshape(-1))
    f, d = ctypes.c_float, ctypes.c_double
    for grad in (0, 1):
        nll = np.zeros((K, 16))
        dn = np.zeros((K, 16))
        sim.sim_diag_nll(T, 2 * K, 2, BN, 1, grad, _p(y, f), _p(rconst, d), _p(arrs['m0s'], d), _p(arrs['S0s'], d),
                         _p(arrs['As'], d), _p(arrs['Cs'], d), _p(arrs['Qs'], d), _p(cand, d), 16, 0,
                         _p(nll, d), _p(dn, d))
        assert (np.abs(nll - ref_nll) / np.abs(ref_nll)).max() < 1e-5
        np.testing.assert_array_equal(nll.argmin(axis=1), ref_nll.argmin(axis=1))
        if grad:
            assert (np.abs(dn - ref_g) / np.abs(ref_g).max(axis=1, keepdims=True)).max() < 2e-5
