"""Pins for the CPU oracle (oracle/eks_oracle.py): closed-form known answers, agreement of three
independent formulations, finite-difference gradients, and the optimiser's control flow.

The reference's own tests hold no numeric known-answer for this path, and
its golden CSVs are remote, so these are the pins ("parity unpinned" w.r.t. upstream numbers)."""
import numpy as np
import pytest

from oracle import eks_oracle as orc


def _rand_model(rng, K, T, D, O, diag=False):
    y = rng.standard_normal((K, T, O)) * 3.0
    m0 = rng.standard_normal((K, D)) * 0.1
    if diag:
        assert D == O
        eye = np.tile(np.eye(D), (K, 1, 1))
        S0 = eye * rng.uniform(0.5, 4.0, (K, D))[:, :, None]
        A, C, Q = eye.copy(), eye.copy(), eye.copy()
    else:
        L = rng.standard_normal((K, D, D))
        S0 = L @ np.swapaxes(L, 1, 2) + 0.5 * np.eye(D)
        A = np.tile(np.eye(D), (K, 1, 1)) + 0.05 * rng.standard_normal((K, D, D))
        C = rng.standard_normal((K, O, D))
        Lq = rng.standard_normal((K, D, D))
        Q = Lq @ np.swapaxes(Lq, 1, 2) + 0.1 * np.eye(D)
    R = rng.gamma(2.0, 0.5, (K, T, O)) + 1e-3
    s = rng.uniform(0.05, 20.0, K)
    return y, m0, S0, A, C, Q, s, R


def test_T1_hand_computed():
    # one frame, scalar: posterior of N(m0,S0) prior after one observation with noise r
    y = np.array([[[2.0]]])
    ms, Vs, nll = orc.kalman_smoother(y, [[0.5]], [[[4.0]]], [[[1.0]]], [[[1.0]]], [[[1.0]]],
                                      [3.0], np.array([[[1.0]]]))
    S = 4.0 + 1.0
    assert ms[0, 0, 0] == pytest.approx(0.5 + 4.0 / S * 1.5, rel=1e-14)
    assert Vs[0, 0, 0, 0] == pytest.approx(4.0 * 1.0 / S, rel=1e-14)
    assert nll[0] == pytest.approx(0.5 * (np.log(2 * np.pi * S) + 1.5 ** 2 / S), rel=1e-14)


def test_T2_hand_computed():
    # two frames, scalar random walk; smoothed x0 by the joint-Gaussian formula
    m0, S0, s, r0, r1 = 0.0, 2.0, 0.7, 0.5, 0.25
    y0, y1 = 1.0, -0.5
    # joint prior over (x0,x1): cov [[S0,S0],[S0,S0+s]]; obs noise diag(r0,r1)
    P = np.array([[S0, S0], [S0, S0 + s]])
    Rm = np.diag([r0, r1])
    Kg = P @ np.linalg.inv(P + Rm)
    post_m = np.array([m0, m0]) + Kg @ (np.array([y0, y1]) - m0)
    post_P = P - Kg @ P
    ms, Vs, nll = orc.kalman_smoother(np.array([[[y0], [y1]]]), [[m0]], [[[S0]]], [[[1.0]]],
                                      [[[1.0]]], [[[1.0]]], [s], np.array([[[r0], [r1]]]))
    np.testing.assert_allclose(ms[0, :, 0], post_m, rtol=1e-13)
    np.testing.assert_allclose(Vs[0, :, 0, 0], np.diag(post_P), rtol=1e-13)
    S = P + Rm
    e = np.array([y0, y1]) - m0
    ref = 0.5 * (2 * np.log(2 * np.pi) + np.log(np.linalg.det(S)) + e @ np.linalg.solve(S, e))
    assert nll[0] == pytest.approx(ref, rel=1e-13)


def test_steady_state_riccati():
    # constant r: predicted variance converges to P = (s + sqrt(s^2 + 4 s r)) / 2
    s, r, T = 0.3, 2.0, 4000
    y = np.zeros((1, T, 1))
    f = orc.kalman_filter(y, [[0.0]], [[[1.0]]], [[[1.0]]], [[[1.0]]], [[[1.0]]], [s],
                          np.full((1, 1), r))
    Pinf_pred = 0.5 * (s + np.sqrt(s * s + 4 * s * r))
    assert f['Pp'][0, -1, 0, 0] == pytest.approx(Pinf_pred, rel=1e-12)
    assert f['Pf'][0, -1, 0, 0] == pytest.approx(Pinf_pred - s, rel=1e-11)


def test_limits_of_s():
    rng = np.random.default_rng(1)
    y = rng.standard_normal((1, 200, 1))
    r = np.full((1, 200, 1), 0.5)
    one = [[[1.0]]]
    ms_big, _, _ = orc.kalman_smoother(y, [[0.0]], [[[1e8]]], one, one, one, [1e9], r)
    np.testing.assert_allclose(ms_big[0, :, 0], y[0, :, 0], atol=1e-6)     # s->inf: follows data
    ms_small, _, _ = orc.kalman_smoother(y, [[0.0]], [[[1e8]]], one, one, one, [1e-12], r)
    np.testing.assert_allclose(ms_small[0, :, 0], y.mean(), atol=1e-5)     # s->0: common mean


@pytest.mark.parametrize('D,O', [(1, 1), (2, 2), (3, 4), (3, 6), (4, 8), (5, 8)])
def test_three_formulations_agree(D, O):
    rng = np.random.default_rng(10 * D + O)
    y, m0, S0, A, C, Q, s, R = _rand_model(rng, 3, 97, D, O)
    ms1, Vs1, nll1 = orc.kalman_smoother(y, m0, S0, A, C, Q, s, R)
    ms2, Vs2, nll2 = orc.info_form_smoother(y, m0, S0, A, C, Q, s, R)
    ms3, Vs3, nll3 = orc.assoc_chunked_smoother(y, m0, S0, A, C, Q, s, R, chunk=16)
    sc = np.abs(ms1).max()
    # three routes through different inverses: agreement to ~1e-9 is conditioning, not algorithm
    assert np.abs(ms1 - ms2).max() / sc < 1e-8
    assert np.abs(ms1 - ms3).max() / sc < 1e-8
    assert np.abs(Vs1 - Vs2).max() / np.abs(Vs1).max() < 1e-8
    assert np.abs(Vs1 - Vs3).max() / np.abs(Vs1).max() < 1e-8
    np.testing.assert_allclose(nll1, nll2, rtol=1e-10)
    np.testing.assert_allclose(nll1, nll3, rtol=1e-10)


def test_xy_decoupling_singlecam():
    # diagonal model == two independent scalar chains
    rng = np.random.default_rng(3)
    y, m0, S0, A, C, Q, s, R = _rand_model(rng, 4, 150, 2, 2, diag=True)
    ms, Vs, nll = orc.kalman_smoother(y, m0, S0, A, C, Q, s, R)
    tot = np.zeros(4)
    for c in range(2):
        ms_c, Vs_c, nll_c = orc.kalman_smoother(
            y[:, :, c:c + 1], m0[:, c:c + 1], S0[:, c:c + 1, c:c + 1], A[:, :1, :1], C[:, :1, :1],
            Q[:, :1, :1], s, R[:, :, c:c + 1])
        np.testing.assert_allclose(ms[:, :, c], ms_c[:, :, 0], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(Vs[:, :, c, c], Vs_c[:, :, 0, 0], rtol=1e-13)
        tot += nll_c
    np.testing.assert_allclose(nll, tot, rtol=1e-13)
    assert np.abs(Vs[:, :, 0, 1]).max() == 0.0


@pytest.mark.parametrize('D,O', [(2, 2), (3, 4)])
def test_gradient_matches_finite_difference(D, O):
    rng = np.random.default_rng(5)
    y, m0, S0, A, C, Q, s, R = _rand_model(rng, 3, 60, D, O)
    Rc = orc.constant_R_from_timevarying(R)
    nll, g = orc.filter_nll(y, m0, S0, A, C, Q, s, Rc, want_grad=True)
    h = 1e-6
    up = orc.filter_nll(y, m0, S0, A, C, Q, s * np.exp(h), Rc)
    dn = orc.filter_nll(y, m0, S0, A, C, Q, s * np.exp(-h), Rc)
    np.testing.assert_allclose(g, (up - dn) / (2 * h), rtol=2e-6)


def test_jitter_switch_is_small_on_realistic_scales():
    # dynamax's psd_solve boost (1e-9) is a numerical regulariser; with S >= 1e-3 its effect is
    # far below the 1e-5 parity tolerance, which is why the kernels implement the exact update.
    rng = np.random.default_rng(6)
    y, m0, S0, A, C, Q, s, R = _rand_model(rng, 3, 120, 2, 2, diag=True)
    a = orc.kalman_smoother(y, m0, S0, A, C, Q, s, R, jitter=0.0)
    b = orc.kalman_smoother(y, m0, S0, A, C, Q, s, R, jitter=1e-9)
    assert np.abs(a[0] - b[0]).max() / np.abs(a[0]).max() < 1e-7
    assert np.abs(a[1] - b[1]).max() / np.abs(a[1]).max() < 1e-6


def test_constant_R_and_initial_guess():
    Rt = np.array([[1.0, 5e-5], [3.0, 2e-5], [2.0, 9e-5]])
    np.testing.assert_allclose(orc.constant_R_from_timevarying(Rt), [2.0, 1e-4])
    ev = np.arange(40, dtype=float).reshape(20, 2) ** 2
    d = ev[1:] - ev[:-1]
    assert orc.compute_initial_guess(ev) == round(float(np.std(d)), 5)
    assert orc.compute_initial_guess(np.ones((10, 2))) == 2.0        # zero std -> fallback 2.0
    with pytest.raises(ValueError):
        orc.compute_initial_guess(np.ones((1, 2)))


def test_crop_frames_semantics():
    y = np.arange(10)
    assert orc.crop_frames(y, None) is y
    assert orc.crop_frames(y, [(None, None)]) is y
    np.testing.assert_array_equal(orc.crop_frames(y, [(None, 3)]), [0, 1, 2])
    np.testing.assert_array_equal(orc.crop_frames(y, [(7, None), (0, 2)]), [0, 1, 7, 8, 9])
    for bad in ([(3, 3)], [(0, 11)], [(0, 5), (4, 6)], [(0.0, 3)], [[0, 3]]):
        with pytest.raises(ValueError):
            orc.crop_frames(y, bad)
    with pytest.raises(TypeError):
        orc.crop_frames(y, ((0, 3),))


def test_adam_control_flow():
    # quadratic bowl in u: L = 100 + (u-1)^2.  Checks bias-corrected first step ~ -sign(g),
    # the stop rule and that the returned u includes the stopping iteration's update.
    calls = []

    def lg(u):
        calls.append(u.copy())
        return 100.0 + (u - 1.0) ** 2, 2.0 * (u - 1.0)

    u, last, iters = orc.adam_optimize_s(lg, np.array([3.0]), tol=1e-2, safety_cap=300)
    assert calls[0][0] == 3.0 and calls[1][0] == pytest.approx(2.0, abs=1e-6)   # unit first step
    assert 1 < iters[0] < 300
    # the loop stopped at the first iteration whose |L - prev| was under the threshold
    Ls = [100.0 + (c[0] - 1.0) ** 2 for c in calls]
    k = int(iters[0]) - 1
    assert abs(Ls[k] - Ls[k - 1]) < 1e-2 * abs(np.log(Ls[k - 1])) + 1e-6
    assert all(abs(Ls[i] - Ls[i - 1]) >= 1e-2 * abs(np.log(Ls[i - 1])) + 1e-6
               for i in range(1, k))
    assert last[0] == Ls[k]
    u_cap, _, it_cap = orc.adam_optimize_s(lg, np.array([3.0]), tol=0.0, safety_cap=5)
    assert it_cap[0] == 5


def test_run_kalman_smoother_modes_and_blocks():
    rng = np.random.default_rng(8)
    K, T = 3, 120
    x = np.cumsum(rng.standard_normal((K, T, 2)) * 0.7, axis=1)
    ev = rng.gamma(2.0, 0.3, (T, K, 2)) + 0.05
    y = x + rng.standard_normal((K, T, 2)) * np.sqrt(np.swapaxes(ev, 0, 1))
    eye = np.tile(np.eye(2), (K, 1, 1))
    m0 = np.zeros((K, 2))
    for sp in (10.0, 7, [10.0], [1.0, 2.0, 3.0]):
        s, ms, Vs, _ = orc.run_kalman_smoother(y, m0, eye, eye, eye, eye, ev, smooth_param=sp)
        np.testing.assert_array_equal(s, np.broadcast_to(np.asarray(sp, float), (K,)))
        assert ms.shape == (K, T, 2) and Vs.shape == (K, T, 2, 2)
    s_a, ms, Vs, info = orc.run_kalman_smoother(y, m0, eye, eye, eye, eye, ev)
    assert np.all(np.isfinite(s_a)) and np.all(s_a > 0) and np.all(info['iters'] <= 300)
    # the optimum should sit near the true process variance 0.49 (within the coarse Adam stop)
    assert np.all(s_a > 0.05) and np.all(s_a < 5.0)
    s_b, *_ = orc.run_kalman_smoother(y, m0, eye, eye, eye, eye, ev, blocks=[[0, 1], [2]],
                                      safety_cap=5)
    assert s_b[0] == s_b[1]
    s_c, *_ = orc.run_kalman_smoother(y, m0, eye, eye, eye, eye, ev, s_frames=[(0, 60)])
    assert np.all(np.isfinite(s_c))
    s_g, _, _, info = orc.run_kalman_smoother(y, m0, eye, eye, eye, eye, ev, s_mode='grid')
    assert info['nll'].shape == (K, 64)
    np.testing.assert_array_equal(s_g, info['candidates'][info['argmin']])
    # grid optimum and Adam optimum bracket the same basin
    assert np.all(np.abs(np.log(s_g) - np.log(s_a)) < 1.5)


def test_ensemble_properties():
    rng = np.random.default_rng(9)
    a = rng.random((4, 2, 6, 3, 3))
    e = orc.ensemble(a)
    assert e.shape == (1, 2, 6, 3, 5) and np.isfinite(e).all()
    a32 = a.astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(e[0, ..., 0], np.median(a32[..., 0], axis=0))
    conf = a32[..., 2].sum(axis=0) / 4
    np.testing.assert_allclose(e[0, ..., 2], a32[..., 0].var(axis=0) / conf)
    e2 = orc.ensemble(a, avg_mode='mean', var_mode='var')
    np.testing.assert_allclose(e2[0, ..., 1], a32[..., 1].mean(axis=0))
    np.testing.assert_allclose(e2[0, ..., 3], a32[..., 1].var(axis=0))
    # all-NaN coordinate -> variance replaced by 1000 (reference tests/test_core.py:60-81)
    b = a.copy()
    b[:, 0, 2, 1, 0] = np.nan
    e3 = orc.ensemble(b)
    assert e3[0, 0, 2, 1, 2] == 1000.0
    # single model -> var = 1/max(conf,1e-5)
    e4 = orc.ensemble(a[:1])
    np.testing.assert_allclose(e4[0, ..., 2], 1.0 / np.maximum(a32[0, ..., 2], 1e-5))
    # partially-NaN ensemble member is ignored by nanmedian / nanvar
    c = a.copy()
    c[0, 1, 3, 2, 1] = np.nan
    e5 = orc.ensemble(c)
    np.testing.assert_allclose(e5[0, 1, 3, 2, 1], np.median(a32[1:, 1, 3, 2, 1]))


def test_center_predictions_semantics():
    rng = np.random.default_rng(11)
    ens = rng.random((1, 2, 50, 3, 5))
    mask, cen, good_c, means, good = orc.center_predictions(ens, 100)
    assert mask.all() and good.shape == (3, 50)
    np.testing.assert_allclose(means[0, :, 0], ens[0, :, :, :, 0:2].mean(axis=1))
    np.testing.assert_allclose(cen, ens[..., 0:2] - means)
    mask, cen, good_c, means, good = orc.center_predictions(ens, 50)
    mv = ens[..., 2:4].max(axis=(0, 1, 4))
    np.testing.assert_array_equal(mask, mv <= np.percentile(mv, 50, axis=0))
    nmin = mask.sum(axis=0).min()
    assert good.shape == (3, nmin) and good_c.shape == (1, 2, nmin, 3, 2)
    for k in range(3):
        np.testing.assert_array_equal(good[k], np.where(mask[:, k])[0][:nmin])
        np.testing.assert_allclose(means[0, :, 0, k], ens[0, :, good[k], k, 0:2].mean(axis=0))


def test_c_port_scalar_chain_form_of_the_diagonal_model_equals_the_general_port():
    """bench.py's like-for-like CPU baseline (eksc_smooth_diag / eksc_nll_grid_diag: the diagonal model as independent
    scalar chains, product forms) against the general-matrix C port and the NumPy oracle: same numbers to 1e-10,
    constant and time-varying R, the 1e-12 variance clip included."""
    from oracle import c_oracle
    rng = np.random.default_rng(12)
    K, T = 5, 700
    y = np.cumsum(rng.standard_normal((K, T, 2)), axis=1) + rng.standard_normal((K, T, 2))
    Rd = rng.gamma(2.0, 0.3, (K, T, 2)) + 0.05
    Rd[::37, :, 0] = 1e-12
    eye = np.tile(np.eye(2), (K, 1, 1))
    A = eye * rng.uniform(0.9, 1.0, (K, 2))[:, :, None]
    C = eye * rng.uniform(0.5, 1.5, (K, 2))[:, :, None]
    Q = eye * rng.uniform(0.5, 2.0, (K, 2))[:, :, None]
    m0, S0 = rng.standard_normal((K, 2)), eye * 3.0
    s = np.exp(rng.uniform(-3, 3, K))
    ms, Vs, nll = c_oracle.smooth(y, Rd, m0, S0, A, C, Q, s)
    ms_d, Vd_d, nll_d = c_oracle.smooth_diag(y, Rd, m0, S0, A, C, Q, s)
    Vd = np.diagonal(Vs, axis1=2, axis2=3)
    assert np.abs(ms_d - ms).max() < 1e-10 * np.abs(ms).max()
    # (at the clip the general port's P - K S K' cancels: 1e-4 of a 3e-12 variance; the product forms agree with the
    #  information-form oracle below to 1e-8)
    assert (np.abs(Vd_d - Vd) / Vd).max() < 1e-3 and (np.abs(Vd_d - Vd) / Vd)[1::37].max() < 1e-9
    assert (np.abs(nll_d - nll) / np.abs(nll)).max() < 1e-10
    ms_o, Vs_o, _ = orc.info_form_smoother(y, m0, S0, A, C, Q, s, Rd)[:3]
    assert (np.abs(Vd_d - np.diagonal(Vs_o, axis1=2, axis2=3)) / Vd_d).max() < 1e-8
    Rc = orc.constant_R_from_timevarying(Rd)
    cand = np.exp(np.linspace(-8, 8, 17))
    g1 = c_oracle.nll_grid(y, Rc, m0, S0, A, C, Q, cand)
    g2 = c_oracle.nll_grid_diag(y, Rc, m0, S0, A, C, Q, cand)
    assert (np.abs(g1 - g2) / np.abs(g1)).max() < 1e-10
