"""CPU: the calibrated (nonlinear) multi-camera path - SURVEY.md section 8(f) rank 3.

* the oracle's extended filter (oracle/ekf_oracle.py) reduces to the linear oracle for an affine
  camera and its complex-step Jacobian agrees with the analytic one of the product
  (eks_amd/calibration.py) and of the kernels' header (eks_pinhole.hpp, compiled for the host);
* the kernels' fixed-point formulation (scan sweeps over stored linearisation points + extended
  replay per chunk), run from plain loops over the SAME lane headers, reproduces the sequential
  extended filter / smoother;
* host pieces of the driver: triangulation, geometric initialisation, calibration TOML.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from eks_amd import calibration as cal
from eks_amd import synth
from oracle import ekf_oracle as ek
from oracle import eks_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def dense_sim():
    src = os.path.join(ROOT, 'tests', 'host_sim', 'dense_sim.cpp')
    lib = os.path.join(ROOT, 'tests', 'host_sim', 'libdense_sim_ekf.so')
    subprocess.run(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'eks_amd', 'csrc'),
                    src, '-o', lib], check=True)
    return ctypes.CDLL(lib)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _oracle_h(cams):
    return ek.combine_projections([ek.make_projection_fn(c['rot'], c['tvec'], c['K'], c['dist'])
                                   for c in cams])


def test_extended_filter_with_an_affine_camera_is_the_linear_filter():
    rng = np.random.default_rng(0)
    T, D, O = 200, 3, 4
    C = rng.normal(size=(O, D))
    off = rng.normal(size=O) * 10
    y = rng.normal(size=(T, O)).cumsum(axis=0)
    Rd = 0.5 + rng.random((T, O))
    m0, S0 = rng.normal(size=D), np.eye(D) * 4.0
    A = np.eye(D) + 0.05 * rng.normal(size=(D, D))
    Q = np.diag([1.0, 2.0, 0.5])
    ms, Vs, ll = ek.eks_smoother(y, Rd, m0, S0, A, Q, 0.7, lambda x: x @ C.T + off)
    mo, Vo, llo = orc.kalman_smoother((y - off)[None], m0[None], S0[None], A[None], C[None], Q[None],
                                      np.array([0.7]), Rd[None])
    assert np.abs(ms - mo[0]).max() < 1e-9 and np.abs(Vs - Vo[0]).max() < 1e-10
    assert abs(ll + llo[0]) < 1e-10 * abs(ll)       # third output is the NLL


def test_analytic_jacobians_match_complex_step_at_wide_angles(dense_sim):
    cams = synth.ring_cameras(3, seed=2)
    rng = np.random.default_rng(1)
    X = rng.uniform(-350, 350, size=(40, 3))           # up to ~20 degrees off axis: distortion matters
    for c in cams:
        packed = cal.pack_camera(c['rot'], c['tvec'], c['K'], c['dist'])
        packed[22:25] = [0.02, -0.01, 0.005]           # k4..k6
        packed[26], packed[28] = 1e-4, -5e-5           # s2, s4
        dist = packed[17:29]
        h = ek.make_projection_fn(c['rot'], c['tvec'], c['K'], dist)
        Jref = np.stack([ek.jacobian_cs(h, x) for x in X])
        assert np.abs(cal.project(packed, X) - h(X)).max() < 1e-10
        assert np.abs(cal.project_jacobian(packed, X) - Jref).max() < 1e-9 * np.abs(Jref).max()
        uv, J = np.zeros(2), np.zeros(6)
        for x, jr in zip(X, Jref):
            dense_sim.sim_pinhole(_p(packed), _p(x), _p(uv), _p(J))
            assert np.abs(uv - h(x)).max() < 1e-10
            assert np.abs(J.reshape(2, 3) - jr).max() < 1e-9 * np.abs(jr).max()


def test_rodrigues_small_and_general_angles():
    assert np.allclose(cal.rodrigues([0, 0, 0]), np.eye(3))
    R = cal.rodrigues([0.3, -0.2, 0.5])
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and np.isclose(np.linalg.det(R), 1.0)
    assert np.allclose(R, ek.rodrigues([0.3, -0.2, 0.5]))


@pytest.mark.parametrize('init', ['prior', 'triangulated'])
def test_fixed_point_sweeps_reproduce_the_sequential_extended_smoother(dense_sim, init):
    T, K, V = 700, 3, 3
    prob = synth.calibrated_multicam(T, K, V, seed=11)
    y = prob['y_tko'].astype(np.float32)
    var = prob['var_tko'].astype(np.float32)
    var[5, 0, :] = 1e-5                                 # a near-exact frame
    s = np.array([2.0, 0.01, 300.0])
    m0, S0, A, Q = prob['m0s'], prob['S0s'], prob['As'], prob['Qs']
    if init == 'prior':
        xlin = np.repeat(m0[:, None, :], T, axis=1).copy()
    else:
        xy = np.transpose(prob['y_tko'].reshape(T, K, V, 2), (2, 1, 0, 3)).reshape(V, K * T, 2)
        xlin = cal.triangulate(prob['cams_packed'], xy).reshape(K, T, 3).copy()
    ms = np.zeros((T, K, 3), np.float32)
    Vs = np.zeros((T, K, 3, 3), np.float32)
    nll = np.zeros(K)
    resid = ctypes.c_double()
    n = dense_sim.sim_ekf_smooth(T, K, V, 32, _p(y), _p(var), None, _p(m0), _p(S0), _p(A), _p(Q), _p(s),
                                 _p(prob['cams_packed']), _p(xlin), 40, ctypes.c_double(1e-10), _p(ms),
                                 _p(Vs), _p(nll), ctypes.byref(resid))
    assert n <= 8 and resid.value <= 1e-10
    h = _oracle_h(prob['cams'])
    for k in range(K):
        Rk = np.maximum(var[:, k].astype(np.float64), 1e-12)
        mo, Vo, ll = ek.eks_smoother(y[:, k].astype(np.float64), Rk, m0[k], S0[k], A[k], Q[k], s[k], h)
        _, _, _, mp = ek.ekf_filter(y[:, k].astype(np.float64), Rk, m0[k], S0[k], A[k], Q[k], s[k], h)
        assert np.abs(xlin[k] - mp).max() < 1e-7        # linearisation points = predicted means
        assert np.abs(ms[:, k] - mo).max() < 1e-5 * np.abs(mo).max()
        assert np.abs(Vs[:, k] - Vo).max() < 1e-5 * np.abs(Vo).max()
        assert abs(nll[k] + ll) < 1e-10 * abs(ll)


def test_constant_r_loss_through_the_sweeps(dense_sim):
    T, K, V = 400, 2, 2
    prob = synth.calibrated_multicam(T, K, V, seed=3)
    y = prob['y_tko'].astype(np.float32)
    rconst = np.median(prob['var_tko'], axis=0)
    s = np.array([0.3, 4.0])
    xlin = np.repeat(prob['m0s'][:, None, :], T, axis=1).copy()
    nll = np.zeros(K)
    resid = ctypes.c_double()
    dense_sim.sim_ekf_smooth(T, K, V, 32, _p(y), None, _p(rconst), _p(prob['m0s']), _p(prob['S0s']),
                             _p(prob['As']), _p(prob['Qs']), _p(s), _p(prob['cams_packed']), _p(xlin), 40,
                             ctypes.c_double(1e-10), None, None, _p(nll), ctypes.byref(resid))
    h = _oracle_h(prob['cams'])
    for k in range(K):
        ref = ek.ekf_nll(y[:, k].astype(np.float64), rconst[k], prob['m0s'][k], prob['S0s'][k],
                         prob['As'][k], prob['Qs'][k], s[k], h)
        assert abs(nll[k] - ref) < 1e-10 * abs(ref)


def test_triangulation_and_geometric_initialisation():
    prob = synth.calibrated_multicam(300, 2, 4, seed=5)
    xy = np.stack([cal.project(c, prob['latent'][:, 0]) for c in prob['cams_packed']])
    tri = cal.triangulate(prob['cams_packed'], xy)
    assert np.abs(tri - prob['latent'][:, 0]).max() < 1e-6
    assert np.abs(tri - ek.triangulate_dlt(prob['cams'], xy)).max() < 1e-6
    from eks_amd.multicam_smoother import initialize_kalman_filter_geometric
    ys3 = np.swapaxes(prob['latent'], 0, 1)
    got = initialize_kalman_filter_geometric(ys3)
    ref = ek.initialize_kalman_filter_geometric(ys3)
    for g, r in zip(got, ref):
        assert np.allclose(g, r, rtol=1e-12, atol=0)


def test_triangulation_drops_non_finite_views_per_point():
    """One NaN marker must not abort the run (aniposelib drops that view for that point only and
    returns NaN when fewer than two views are left)."""
    prob = synth.calibrated_multicam(200, 1, 4, seed=6)
    xy = np.stack([cal.project(c, prob['latent'][:, 0]) for c in prob['cams_packed']])     # (V,T,2)
    clean = cal.triangulate(prob['cams_packed'], xy)
    bad = xy.copy()
    bad[1, 10, 0] = np.nan                 # one view gone at frame 10
    bad[0, 20] = np.inf                    # another at frame 20
    bad[:3, 30] = np.nan                   # only one view left at frame 30
    bad[:, 40] = np.nan                    # none at frame 40
    tri = cal.triangulate(prob['cams_packed'], bad)
    keep = np.ones(200, bool)
    keep[[10, 20, 30, 40]] = False
    np.testing.assert_array_equal(tri[keep], clean[keep])
    # exact projections: three views triangulate the same point
    assert np.abs(tri[[10, 20]] - prob['latent'][[10, 20], 0]).max() < 1e-6
    assert np.isnan(tri[30]).all() and np.isnan(tri[40]).all()
    ref10 = cal.triangulate(prob['cams_packed'][[0, 2, 3]], xy[[0, 2, 3], 10:11])
    np.testing.assert_allclose(tri[10], ref10[0], rtol=1e-9)


def test_camera_group_orders_tables_like_aniposelib(tmp_path):
    """aniposelib sorts the TOML keys as strings: cam_10 comes before cam_2."""
    fn = tmp_path / 'calibration.toml'
    body = ''
    for i in (0, 1, 2, 10):
        body += (f'[cam_{i}]\nname = "c{i}"\nmatrix = [ [ 900.0, 0.0, 320.0,], [ 0.0, 900.0, 240.0,], '
                 f'[ 0.0, 0.0, 1.0,],]\ndistortions = [ 0.0, 0.0, 0.0, 0.0, 0.0,]\n'
                 f'rotation = [ 0.0, 0.{i}, 0.0,]\ntranslation = [ 0.0, 0.0, 900.0,]\n\n')
    fn.write_text(body)
    assert [c.name for c in cal.CameraGroup.load(str(fn)).cameras] == ['c0', 'c1', 'c10', 'c2']


def test_camera_group_from_calibration_toml(tmp_path):
    fn = tmp_path / 'calibration.toml'
    fn.write_text('''[cam_0]
name = "top"
size = [ 640, 480,]
matrix = [ [ 900.0, 0.5, 320.0,], [ 0.0, 880.0, 240.0,], [ 0.0, 0.0, 1.0,],]
distortions = [ -0.15, 0.06, 0.001, -0.002, 0.01,]
rotation = [ 0.1, -0.2, 0.3,]
translation = [ 1.0, 2.0,
  1000.0,]

[cam_1]
name = "bot"
size = [ 640, 480,]
matrix = [ [ 910.0, 0.0, 320.0,], [ 0.0, 890.0, 240.0,], [ 0.0, 0.0, 1.0,],]
distortions = [ -0.1, 0.0, 0.0, 0.0, 0.0,]
rotation = [ 0.0, 0.5, 0.0,]
translation = [ -10.0, 0.0, 900.0,]

[metadata]
adjusted = false
error = 0.52
''')
    g = cal.CameraGroup.load(str(fn))
    assert [c.name for c in g.cameras] == ['top', 'bot']
    h, heads = cal.make_projection_from_camgroup(g)
    assert h.n_cameras == 2 and h.cams.shape == (2, 32)
    ref = ek.make_projection_fn([0.1, -0.2, 0.3], [1.0, 2.0, 1000.0],
                                np.array([[900.0, 0.5, 320.0], [0, 880.0, 240.0], [0, 0, 1.0]]),
                                [-0.15, 0.06, 0.001, -0.002, 0.01])
    x = np.array([20.0, -35.0, 60.0])
    assert np.allclose(heads[0](x), ref(x), rtol=1e-13)
    assert np.allclose(h(x)[:2], ref(x), rtol=1e-13) and h(x).shape == (4,)


def test_only_calibrated_projections_are_accepted_as_h_fn():
    from eks_amd.core import run_kalman_smoother
    z = np.zeros
    with pytest.raises(NotImplementedError):
        run_kalman_smoother(z((1, 4, 4)), z((1, 3)), z((1, 3, 3)), z((1, 3, 3)), z((1, 4, 3)),
                            z((1, 3, 3)), z((4, 1, 4)), smooth_param=1.0, h_fn=lambda x: x)


# ---- the reference's own calibrated data set (data/fly + calibration.toml), tests/golden ----------
@pytest.fixture(scope='module')
def fly(golden_dir):
    return np.load(os.path.join(golden_dir, 'fly_calibrated_multicam.npz'))


def test_real_calibration_file_loads_and_triangulates_the_fly_markers(fly, tmp_path):
    fn = tmp_path / 'calibration.toml'
    fn.write_text(str(fly['toml']))
    group = cal.CameraGroup.load(str(fn))
    assert [c.name for c in group.cameras] == list(fly['cameras']) == ['Cam-A', 'Cam-B', 'Cam-C']
    packed = cal.cameras_of(group)
    assert packed.shape == (3, 32) and packed[0, 12] == 20003.023008090135 and packed[2, 17] == -6468.60716369641
    from eks_amd import MarkerArray
    from eks_amd.multicam_smoother import initialize_kalman_filter_geometric
    ma = MarkerArray(fly['markers'].astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    ys3 = cal.triangulate_3d_models(ma, group).mean(axis=0)                   # (K,T,3)
    assert np.abs(ys3 - fly['ys3']).max() < 1e-5 * np.abs(fly['ys3']).max()
    m0s, S0s, _, Qs, _ = initialize_kalman_filter_geometric(ys3)
    np.testing.assert_allclose(m0s, fly['m0s'], rtol=1e-6)
    np.testing.assert_allclose(S0s, fly['S0s'], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(Qs, fly['Qs'], rtol=1e-4, atol=1e-14)


def test_oracle_reproduces_the_fly_golden_for_one_keypoint(fly, tmp_path):
    """Guards the committed vectors against drift of the oracle (keypoint 3, smooth_param 10)."""
    import ast
    cams = []
    for block in str(fly['toml']).split('[cam_')[1:]:
        d = {}
        for line in block.split('\n')[1:]:
            if '=' in line and not line.startswith('['):
                k, v = line.split('=', 1)
                d[k.strip()] = ast.literal_eval(v.strip())
            elif line.startswith('['):
                break
        cams.append(dict(rot=np.array(d['rotation']), tvec=np.array(d['translation']),
                         K=np.array(d['matrix']), dist=np.array(d['distortions'])))
    mk = fly['markers']
    M, V, T, K, _ = mk.shape
    k = 3
    heads = [ek.make_projection_fn(c['rot'], c['tvec'], c['K'], c['dist']) for c in cams]
    st = orc.ensemble(mk[:, :, :, k:k + 1])[0]                                  # (V,T,1,5)
    tri = np.stack([ek.triangulate_dlt(cams, mk[m, :, :, k, :2].astype(np.float64)) for m in range(M)]).mean(axis=0)
    m0s, S0s, As, Qs, _ = ek.initialize_kalman_filter_geometric(tri[None])
    f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
    ys = np.transpose(st[..., 0:2], (2, 1, 0, 3)).reshape(1, T, 2 * V)
    evs = np.transpose(st[..., 2:4], (2, 1, 0, 3)).reshape(1, T, 2 * V)
    _, ms, Vs, _ = ek.run_kalman_smoother_nonlinear(f32(ys), m0s, S0s, As, Qs, np.swapaxes(f32(evs), 0, 1),
                                                    ek.combine_projections(heads), smooth_param=10.0)
    idx = fly['keep_idx']
    ref = fly['s10_cam1_rows'].reshape(len(idx), K, 9)[:, k]
    got = heads[1](ms[0])[idx]
    assert np.abs(got - ref[:, 0:2]).max() < 1e-5 * np.abs(ref[:, 0:2]).max()
    lat = fly['s10_latent_rows'].reshape(len(idx), K, 6)[:, k]
    assert np.abs(ms[0][idx] - lat[:, :3]).max() < 1e-6 * np.abs(lat[:, :3]).max()


# ---- what the reference's own helper tests assert (tests/test_multicam_smoother.py:484-640) --------
class _Cam:
    def __init__(self, rotation, translation, K, dist):
        self._r, self._t, self._K, self._d = rotation, translation, K, dist

    def get_rotation(self):
        return self._r

    def get_translation(self):
        return self._t

    def get_camera_matrix(self):
        return self._K

    def get_distortions(self):
        return self._d


class _Group:
    """A camera group whose triangulate() is a recognisable stand-in (mean pixel, z = 1)."""

    def __init__(self, cameras):
        self.cameras = cameras

    def triangulate(self, xy_views, fast=True, disable_64bit=False):
        xy = np.asarray(xy_views)
        return np.stack([xy[:, :, 0].mean(axis=0), xy[:, :, 1].mean(axis=0), np.ones(xy.shape[1])], axis=-1)


def _random_camera(rng, with_dist):
    rvec = rng.normal(size=3) * rng.uniform(0.0, 2.0)
    tvec = rng.normal(size=3) * 0.5
    Km = np.array([[rng.uniform(500, 1500), 0.0, rng.uniform(200, 800)],
                   [0.0, rng.uniform(500, 1500), rng.uniform(200, 800)], [0.0, 0.0, 1.0]])
    dist = np.zeros(14)
    if with_dist:
        dist[:5] = rng.normal(size=5) * np.array([1e-3, 1e-4, 1e-4, 1e-4, 1e-5])
    return rvec, tvec, Km, dist


def test_combined_projection_concatenates_cameras_in_order():
    rng = np.random.default_rng(0)
    rA, tA, KA, dA = _random_camera(rng, True)
    rB, tB, KB, dB = _random_camera(rng, False)
    group = _Group([_Cam(cal.rodrigues(rA), tA, KA, dA), _Cam(rB, tB, KB, dB)])   # matrix and vector forms
    h, heads = cal.make_projection_from_camgroup(group)
    x = rng.normal(size=3)
    x[2] = abs(x[2]) + 0.5
    uv = h(x)
    assert uv.shape == (4,)
    np.testing.assert_allclose(uv, np.concatenate([heads[0](x), heads[1](x)]))
    np.testing.assert_allclose(heads[0](x), ek.make_projection_fn(rA, tA, KA, dA)(x), rtol=1e-12)
    np.testing.assert_allclose(heads[1](x), ek.make_projection_fn(rB, tB, KB, dB)(x), rtol=1e-12)


def test_triangulate_3d_models_uses_the_groups_triangulate_and_shapes():
    class _Markers:
        def __init__(self, a):
            self._a = a
            self.shape = a.shape

        def get_array(self):
            return self._a

    rng = np.random.default_rng(0)
    M, C, T, K = 2, 3, 5, 4
    arr = rng.normal(size=(M, C, T, K, 3))
    group = _Group([_Cam(np.eye(3), np.zeros(3), np.eye(3), np.zeros(14)) for _ in range(C)])
    tri = cal.triangulate_3d_models(_Markers(arr), group)
    assert tri.shape == (M, K, T, 3)
    expected = np.stack([arr[..., 0].mean(axis=1), arr[..., 1].mean(axis=1), np.ones((M, T, K))], axis=-1)
    np.testing.assert_allclose(tri, np.transpose(expected, (0, 2, 1, 3)), atol=1e-12)


def test_projected_covariance_matches_a_finite_difference_linearisation():
    rng = np.random.default_rng(0)
    rvec, tvec, Km, dist = _random_camera(rng, True)
    head = cal.PinholeProjection(cal.pack_camera(rvec, tvec, Km, dist)[None]).heads[0]
    T = 8
    ms = rng.normal(size=(T, 3))
    ms[:, 2] = np.abs(ms[:, 2]) + 0.5
    A = rng.normal(size=(T, 3, 3)) * 1e-2
    Vs = np.einsum('tij,tik->tjk', A, A) + np.eye(3)[None] * 1e-6
    infl = np.abs(rng.normal(size=(T, 3))) * 1e-4
    vx, vy = cal.project_3d_covariance_to_2d(ms, Vs, head, infl)
    assert vx.shape == (T,) and vy.shape == (T,)
    for t in range(T):
        J = np.zeros((2, 3))
        for i in range(3):
            e = np.zeros(3)
            e[i] = 1e-5
            J[:, i] = (head(ms[t] + e) - head(ms[t] - e)) / 2e-5
        cov = J @ Vs[t] @ J.T
        np.testing.assert_allclose(vx[t] - infl[t, 0], cov[0, 0], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(vy[t] - infl[t, 1], cov[1, 1], rtol=1e-4, atol=1e-6)


def test_distortion_padding_order_and_ignored_tilt():
    """OpenCV order k1 k2 p1 p2 k3 ... zero-padded; the tilt terms tx, ty change nothing
    (reference parse_dist, eks/multicam_smoother.py:799-811)."""
    rng = np.random.default_rng(1)
    rvec, tvec, Km, _ = _random_camera(rng, False)
    x = np.array([0.2, -0.1, 1.5])
    five = np.array([0.1, -0.2, 0.01, -0.01, 0.001])
    full = np.zeros(14)
    full[:5] = five
    a = cal.project(cal.pack_camera(rvec, tvec, Km, five), x)
    b = cal.project(cal.pack_camera(rvec, tvec, Km, full), x)
    full[12:] = [0.3, -0.4]
    c = cal.project(cal.pack_camera(rvec, tvec, Km, full), x)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(a, c)
    packed = cal.pack_camera(rvec, tvec, Km, np.arange(14) / 100.0)
    np.testing.assert_array_equal(packed[17:31], np.arange(14) / 100.0)
