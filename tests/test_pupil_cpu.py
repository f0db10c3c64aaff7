"""CPU: the pupil-smoother row (reference eks/ibl_pupil_smoother.py).

* oracle: the loss gradient by forward sensitivities (NumPy) == complex-step derivative of the C
  filter == central differences; Adam reproduces the committed golden run; fixed-parameter outputs
  reproduce the committed golden table.
* lane arithmetic: the dual-number AR(1) loss of the gfx950 kernels, compiled for the host from the
  same headers (tests/host_sim/dense_sim.cpp), against the oracle.
* host logic: per-frame geometry, the checks of the reference's tests/test_ibl_pupil_smoother.py.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle import c_oracle
from oracle import eks_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def synth_pupil(T, seed):
    from eks_amd import synth
    return synth.pupil_observations(T, seed)


@pytest.fixture(scope='module')
def gold(golden_dir):
    g = np.load(os.path.join(golden_dir, 'ibl_pupil_pupil.npz'))
    mk = np.load(os.path.join(golden_dir, 'ibl_pupil_singlecam.npz'))['markers'][:, :, :, g['order']]
    return g, mk


# ---------------------------------------------------------------------------------------------
# oracle
# ---------------------------------------------------------------------------------------------
def test_tangent_sensitivity_matches_central_differences():
    rng = np.random.default_rng(1)
    K, T, D, O = 2, 60, 3, 5
    y = rng.normal(size=(K, T, O))
    m0 = rng.normal(size=(K, D))
    S0 = np.tile(np.eye(D), (K, 1, 1))
    A = np.tile(np.diag([0.9, 0.8, 0.7]), (K, 1, 1)) + 0.02 * rng.normal(size=(K, D, D))
    C = rng.normal(size=(K, O, D))
    L = rng.normal(size=(K, D, D))
    Q = L @ np.swapaxes(L, 1, 2) + np.eye(D)
    R = rng.uniform(0.5, 2, size=(K, T, O))
    dA = rng.normal(size=(K, D, D))
    dL = rng.normal(size=(K, D, D))
    dQ = dL + np.swapaxes(dL, 1, 2)
    an = orc.kalman_filter(y, m0, S0, A, C, Q, 1.0, R, tangent=(dA, dQ))['dll']
    h = 1e-6
    fd = (orc.kalman_filter(y, m0, S0, A + h * dA, C, Q + h * dQ, 1.0, R)['ll']
          - orc.kalman_filter(y, m0, S0, A - h * dA, C, Q - h * dQ, 1.0, R)['ll']) / (2 * h)
    np.testing.assert_allclose(an, fd, rtol=2e-6)
    # the d/dlog s special case is the tangent (0, sQ)
    g1 = orc.kalman_filter(y, m0, S0, A, C, Q, 2.0, R, want_grad=True)['dll']
    g2 = orc.kalman_filter(y, m0, S0, A, C, Q, 2.0, R, tangent=(np.zeros_like(A), 2.0 * Q))['dll']
    np.testing.assert_allclose(g1, g2, rtol=1e-13)


@pytest.mark.parametrize('u', [(4.6, 3.9), (0.0, 0.0), (-3.0, 6.5), (7.5, -2.0)])
def test_complex_step_c_filter_agrees_with_numpy_sensitivities(u):
    ys, ev, m0, S0, lv = synth_pupil(400, seed=3)
    L1, g1 = orc.pupil_nll_and_grad(np.array(u), ys, m0, S0, orc.PUPIL_C, ev, lv, use_c=False)
    L2, g2 = orc.pupil_nll_and_grad(np.array(u), ys, m0, S0, orc.PUPIL_C, ev, lv, use_c=True)
    assert abs(L1 - L2) < 1e-9 * abs(L1)
    np.testing.assert_allclose(g1, g2, rtol=1e-9, atol=1e-9 * np.abs(g1).max())


def test_c_filter_value_matches_numpy_filter_without_directions():
    ys, ev, m0, S0, lv = synth_pupil(300, seed=4)
    A, Q = orc.pupil_dynamics(0.99, 0.95, lv)
    nll, d = c_oracle.nll_directional(ys, ev, m0, S0, A, orc.PUPIL_C, Q, np.zeros((0, 3, 3)), np.zeros((0, 3, 3)))
    ref = orc.filter_nll(ys[None], m0[None], S0[None], A[None], orc.PUPIL_C[None], Q[None], 1.0, ev[None])[0]
    assert d.shape == (0,) and abs(nll - ref) < 1e-10 * abs(ref)


def test_stable_s_map_and_fixed_parameters():
    s, ds = orc.pupil_to_stable_s(np.array([-50.0, 0.0, 50.0]))
    np.testing.assert_allclose(s, [1e-3, 0.5, 1 - 1e-3], atol=1e-12)
    assert ds[1] == pytest.approx(0.25 * 0.998)
    ys, ev, m0, S0, lv = synth_pupil(50, seed=5)
    assert orc.pupil_optimize_smooth(ys, m0, S0, orc.PUPIL_C, ev, lv, smooth_params=[0.5, 0.5])[:2] == (0.5, 0.5)
    s_d, s_c = orc.pupil_optimize_smooth(ys, m0, S0, orc.PUPIL_C, ev, lv, smooth_params=[0.99, 1.5])[:2]
    assert s_d == float(np.float32(0.99)) and s_c == float(np.float32(1 - 1e-3))


def test_adam_decreases_the_loss_and_respects_cap_and_crop():
    ys, ev, m0, S0, lv = synth_pupil(300, seed=6)
    u0 = np.log(np.array([0.99, 0.98]) / (1 - np.array([0.99, 0.98])))
    L0, _ = orc.pupil_nll_and_grad(u0, ys, m0, S0, orc.PUPIL_C, ev, lv)
    s_d, s_c, it, last = orc.pupil_optimize_smooth(ys, m0, S0, orc.PUPIL_C, ev, lv, safety_cap=60)
    assert it == 60 and last < L0 and 1e-3 < s_d < 1 and 1e-3 < s_c < 1
    a = orc.pupil_optimize_smooth(ys, m0, S0, orc.PUPIL_C, ev, lv, s_frames=[(20, 120)], safety_cap=15)
    b = orc.pupil_optimize_smooth(ys[20:120], m0, S0, orc.PUPIL_C, ev[20:120], lv, safety_cap=15)
    assert a == b


def test_oracle_reproduces_golden_pupil_vectors(gold):
    g, mk = gold
    arrs = orc.pupil_arrays(mk)
    np.testing.assert_allclose(arrs['m0'], g['m0'], rtol=1e-12)
    np.testing.assert_allclose(arrs['latent_vars'], g['latent_vars'], rtol=1e-12)
    args = (arrs['ys'], arrs['m0'], arrs['S0'], arrs['C'], arrs['ensemble_vars'], arrs['latent_vars'])
    s, ms, Vs, info = orc.run_pupil_kalman_smoother(*args, smooth_params=[0.99, 0.99])
    assert s == list(g['fixed_s'])
    full = orc.pupil_outputs(arrs, ms, Vs)
    ref = g['fixed_rows'].astype(np.float64)
    assert (np.abs(full[g['keep_idx']] - ref) / np.abs(ref).max(axis=0)).max() < 3e-7
    np.testing.assert_allclose(full.sum(axis=0), g['fixed_colsum'], rtol=1e-9)
    for u, L, gr in zip(g['probe_u'], g['probe_nll'], g['probe_grad']):
        L2, g2 = orc.pupil_nll_and_grad(u, *args, use_c=True)
        assert abs(L2 - L) < 1e-9 * abs(L)
        np.testing.assert_allclose(g2, gr, rtol=1e-8, atol=1e-8 * np.abs(gr).max())


def test_oracle_adam_reproduces_golden_run(gold):
    g, mk = gold
    arrs = orc.pupil_arrays(mk)
    s_d, s_c, it, last = orc.pupil_optimize_smooth(arrs['ys'], arrs['m0'], arrs['S0'], arrs['C'],
                                                   arrs['ensemble_vars'], arrs['latent_vars'])
    assert it == int(g['adam_iters'])
    np.testing.assert_allclose([s_d, s_c], g['adam_s'], rtol=1e-9)
    assert abs(last - float(g['adam_last_loss'])) < 1e-7


def test_output_table_layout_quirks(gold):
    """Column block i holds (top, right, bottom, left)[i]'s coordinates, keypoint_names[i]'s
    likelihood, and C V C' entries (i, i), (i+1, i+1) - the reference's layout, kept verbatim."""
    g, mk = gold
    arrs = orc.pupil_arrays(mk)
    T = arrs['ys'].shape[0]
    ms = np.zeros((T, 3))
    Vs = np.tile(np.diag([4.0, 1.0, 2.0]), (T, 1, 1))
    out = orc.pupil_outputs(arrs, ms, Vs).reshape(T, 4, 9)
    yv = orc.PUPIL_C @ np.diag([4.0, 1.0, 2.0]) @ orc.PUPIL_C.T
    for i, col in enumerate((0, 4, 2, 6)):
        np.testing.assert_allclose(out[:, i, 3], arrs['preds'][:, col])
        np.testing.assert_allclose(out[:, i, 2], arrs['likes'][:, i])
        assert out[0, i, 7] == yv[i, i] and out[0, i, 8] == yv[i + 1, i + 1]
    np.testing.assert_allclose(out[:, :, 0], arrs['mean_x'])
    np.testing.assert_allclose(out[:, :, 1], arrs['mean_y'])


# ---------------------------------------------------------------------------------------------
# lane arithmetic of the kernels on the host
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def sim():
    src = os.path.join(ROOT, 'tests', 'host_sim', 'dense_sim.cpp')
    lib = os.path.join(ROOT, 'tests', 'host_sim', 'libdense_sim.so')
    subprocess.run(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'eks_amd', 'csrc'),
                    src, '-o', lib], check=True)
    return ctypes.CDLL(lib)


@pytest.mark.parametrize('B,tree', [(8, 1), (2, 1), (64, 0), (4096, 0)])
@pytest.mark.parametrize('u', [(4.6, 3.9), (1.0, -1.0), (7.0, 7.0)])
def test_dual_number_ar1_loss_matches_oracle(sim, B, tree, u):
    T = 900
    ys, ev, m0, S0, lv = synth_pupil(T, seed=7)
    L, g = orc.pupil_nll_and_grad(np.array(u), ys, m0, S0, orc.PUPIL_C, ev, lv, use_c=False)
    s, ds = orc.pupil_to_stable_s(np.array(u))
    a = np.array([s[0], s[1], s[1]])
    q = lv * (1 - a * a)
    da = np.array([[ds[0], 0, 0], [0, ds[1], ds[1]]])
    dq = np.array([[-2 * s[0] * ds[0] * lv[0], 0, 0], [0, -2 * s[1] * ds[1] * lv[1], -2 * s[1] * ds[1] * lv[2]]])
    y32, v32 = ys.astype(np.float32), ev.astype(np.float32)
    C = np.ascontiguousarray(orc.PUPIL_C)
    P = lambda x: x.ctypes.data_as(ctypes.c_void_p)      # noqa: E731
    nll, dn = np.zeros(1), np.zeros(2)
    rc = sim.sim_ar1_nll(T, 3, 8, B, tree, P(y32), P(v32), P(m0), P(S0), P(C), P(a), P(q), P(da), P(dq), 2, P(nll), P(dn))
    assert rc == 0
    assert abs(nll[0] - L) < 1e-10 * abs(L)
    np.testing.assert_allclose(dn, g, rtol=1e-9, atol=1e-9 * np.abs(g).max())
    rc = sim.sim_ar1_nll(T, 3, 8, B, tree, P(y32), P(v32), P(m0), P(S0), P(C), P(a), P(q), None, None, 0, P(nll), P(dn))
    assert rc == 0 and abs(nll[0] - L) < 1e-10 * abs(L)


def test_ar1_loss_with_clipped_and_wildly_mixed_variances(sim):
    """Variances below the 1e-12 clip and frames whose variances span > 8 decades (absorbed
    observation by observation instead of through the information matrix)."""
    T = 300
    ys, ev, m0, S0, lv = synth_pupil(T, seed=9)
    ev[::7, 3] = 1e-13
    ev[5::11, :] *= 1e-3
    ev[3::13, 6] = 1e9
    u = np.array([3.0, 2.0])
    ev32 = ev.astype(np.float32)
    L, g = orc.pupil_nll_and_grad(u, ys, m0, S0, orc.PUPIL_C, ev32.astype(np.float64), lv, use_c=False)
    s, ds = orc.pupil_to_stable_s(u)
    a = np.array([s[0], s[1], s[1]])
    q = lv * (1 - a * a)
    da = np.array([[ds[0], 0, 0], [0, ds[1], ds[1]]])
    dq = np.array([[-2 * s[0] * ds[0] * lv[0], 0, 0], [0, -2 * s[1] * ds[1] * lv[1], -2 * s[1] * ds[1] * lv[2]]])
    y32 = ys.astype(np.float32)
    C = np.ascontiguousarray(orc.PUPIL_C)
    P = lambda x: x.ctypes.data_as(ctypes.c_void_p)      # noqa: E731
    nll, dn = np.zeros(1), np.zeros(2)
    assert sim.sim_ar1_nll(T, 3, 8, 8, 1, P(y32), P(ev32), P(m0), P(S0), P(C), P(a), P(q), P(da), P(dq), 2,
                           P(nll), P(dn)) == 0
    assert abs(nll[0] - L) < 1e-9 * abs(L)
    np.testing.assert_allclose(dn, g, rtol=1e-7, atol=1e-8 * np.abs(g).max())


# ---------------------------------------------------------------------------------------------
# host logic (reference tests/test_ibl_pupil_smoother.py)
# ---------------------------------------------------------------------------------------------
def _mock_dlc(n=10, seed=0):
    rng = np.random.default_rng(seed)
    return {f'pupil_{p}_r_{c}': rng.random(n) for p in ('top', 'bottom', 'left', 'right') for c in 'xy'}


def _as_array(dlc):
    return np.stack([dlc[f'{kp}_{c}'] for kp in orc.PUPIL_KEYPOINTS for c in 'xy'], axis=1)


def test_get_pupil_location_shapes_nans_and_oracle():
    from eks_amd.ibl_pupil_smoother import get_pupil_location
    dlc = _mock_dlc()
    dlc['pupil_top_r_x'][2] = np.nan
    dlc['pupil_left_r_y'][5] = np.nan
    c = get_pupil_location(dlc)
    assert isinstance(c, np.ndarray) and c.shape == (10, 2) and np.isfinite(c).all()
    np.testing.assert_array_equal(c, orc.pupil_location(_as_array(dlc)))
    # strict pairs poison their own estimate only: left x NaN -> centre x from top/bottom alone
    dlc['pupil_left_r_x'][7] = np.nan
    c = get_pupil_location(dlc)
    assert c[7, 0] == 0.5 * (dlc['pupil_top_r_x'][7] + dlc['pupil_bottom_r_x'][7])
    np.testing.assert_array_equal(c, orc.pupil_location(_as_array(dlc)))


def test_get_pupil_diameter_shapes_nans_and_oracle():
    from eks_amd.ibl_pupil_smoother import get_pupil_diameter
    dlc = _mock_dlc(seed=1)
    dlc['pupil_top_r_x'][2] = np.nan
    dlc['pupil_left_r_y'][5] = np.nan
    d = get_pupil_diameter(dlc)
    assert isinstance(d, np.ndarray) and d.shape == (10,) and np.isfinite(d).all()
    np.testing.assert_array_equal(d, orc.pupil_diameter(_as_array(dlc)))
    assert np.isnan(get_pupil_diameter({k: np.full(10, np.nan) for k in dlc})).all()
    # a perfect circle of diameter 6: all six estimates agree
    circ = dict(pupil_top_r_x=np.array([0.]), pupil_top_r_y=np.array([3.]), pupil_bottom_r_x=np.array([0.]),
                pupil_bottom_r_y=np.array([-3.]), pupil_left_r_x=np.array([-3.]), pupil_left_r_y=np.array([0.]),
                pupil_right_r_x=np.array([3.]), pupil_right_r_y=np.array([0.]))
    assert get_pupil_diameter(circ)[0] == pytest.approx(6.0, rel=1e-15)


def test_add_mean_to_array():
    from eks_amd.ibl_pupil_smoother import add_mean_to_array
    arr = np.arange(40.0).reshape(10, 4)
    keys = ['key1_x', 'key2_y', 'key3_x', 'key4_y']
    out = add_mean_to_array(arr, keys, 2.0, 3.0)
    assert isinstance(out, dict) and set(out) == set(keys)
    for i, k in enumerate(keys):
        np.testing.assert_array_equal(out[k], arr[:, i] + (2.0 if 'x' in k else 3.0))
    assert add_mean_to_array(np.zeros((0, 0)), [], 2.0, 3.0) == {}
    one = add_mean_to_array(np.array([[1.0, 2.0, 3.0, 4.0]]), keys, 2.0, 3.0)
    assert [float(one[k][0]) for k in keys] == [3.0, 5.0, 5.0, 7.0]


def test_fixed_parameters_bypass_and_no_gpu_refusal():
    import torch
    from eks_amd import _lib
    from eks_amd import ibl_pupil_smoother as ips
    assert ips._fixed_params([0.5, 0.5]) == (0.5, 0.5)
    assert ips._fixed_params([0.99, 2.0]) == (float(np.float32(0.99)), float(np.float32(1 - 1e-3)))
    assert ips._fixed_params(None) is None and ips._fixed_params([None, None]) is None
    assert ips._fixed_params([0.9, None]) is None          # upstream: one fixed parameter unsupported
    np.testing.assert_array_equal(ips.PUPIL_C, orc.PUPIL_C)
    np.testing.assert_allclose(ips._to_stable_s([-1.0, 2.0]), orc.pupil_to_stable_s([-1.0, 2.0])[0], rtol=1e-15)
    if not torch.cuda.is_available():
        ys, ev, m0, S0, lv = synth_pupil(20, seed=8)
        with pytest.raises(_lib.EksHipError):
            ips.run_pupil_kalman_smoother(ys, m0, S0, orc.PUPIL_C, ev, *lv, smooth_params=[0.9, 0.9])
