"""GPU: the calibrated (nonlinear) multi-camera path through the C ABI (eks_ekf_smooth) and through
the reference-shaped operator / driver, against the sequential extended Kalman filter / smoother of
oracle/ekf_oracle.py on seeded synthetic calibrated rigs (reference
eks/core.py:188-190, eks/multicam_smoother.py:369-407, :450-480).
Tolerance: 1e-5 relative to the output's magnitude (BASELINE.json); s by Adam: 1e-4 (the loss
gradient is a central difference on both sides, with different steps)."""
import numpy as np
import pandas as pd
import pytest

from eks_amd import calibration as cal
from eks_amd import synth
from oracle import ekf_oracle as ek
from oracle import eks_oracle as orc

pytestmark = pytest.mark.gpu


def _dev(a, dtype=None):
    import torch
    t = torch.as_tensor(np.ascontiguousarray(a), device='cuda')
    return t if dtype is None else t.to(dtype)


def _oracle_h(prob):
    return ek.combine_projections([ek.make_projection_fn(c['rot'], c['tvec'], c['K'], c['dist'])
                                   for c in prob['cams']])


def _f32(a):
    return np.asarray(a, np.float32).astype(np.float64)


@pytest.mark.parametrize('T,K,V', [(1500, 3, 3), (37, 2, 2), (4100, 2, 4), (1, 2, 2)])
def test_kernel_matches_sequential_extended_smoother(T, K, V):
    import torch
    from eks_amd import hip_ops
    prob = synth.calibrated_multicam(max(T, 12), K, V, seed=T)
    y, var = prob['y_tko'][:T], prob['var_tko'][:T]
    s = np.exp(np.linspace(-4, 5, K))
    m0 = _dev(prob['m0s'])
    xlin = m0[:, None, :].expand(K, T, 3).contiguous()
    ms, Vs, nll, info = hip_ops.ekf_smooth(_dev(y, torch.float32), _dev(var, torch.float32), None, m0,
                                           _dev(prob['S0s']), _dev(prob['As']), _dev(prob['Qs']), _dev(s),
                                           _dev(prob['cams_packed']), xlin, max_sweeps=24, tol=1e-10)
    assert info[1].item() <= 1e-10 and 1 <= info[0].item() <= 12
    ms, Vs, nll, xl = ms.cpu().numpy(), Vs.cpu().numpy(), nll.cpu().numpy(), xlin.cpu().numpy()
    h = _oracle_h(prob)
    for k in range(K):
        args = (_f32(y[:, k]), np.maximum(_f32(var[:, k]), 1e-12), prob['m0s'][k], prob['S0s'][k],
                prob['As'][k], prob['Qs'][k], s[k], h)
        mo, Vo, ll = ek.eks_smoother(*args)
        mp = ek.ekf_filter(*args)[3]
        assert np.abs(xl[k] - mp).max() < 1e-7 * max(1.0, np.abs(mp).max())
        assert np.abs(ms[:, k] - mo).max() < 1e-5 * np.abs(mo).max()
        assert np.abs(Vs[:, k] - Vo).max() < 1e-5 * np.abs(Vo).max()
        assert abs(nll[k] + ll) < 1e-9 * abs(ll)


def test_loose_tolerance_still_smooths_at_the_points_it_stopped_at():
    """once the filter sweeps meet the caller's tolerance the smoothing sweep reuses the elements in the
    workspace instead of rebuilding them at the final linearisation points - fine at 1e-10, not at a loose tolerance.
    The shortcut is taken below 1e-8 only: a call with tol = 1e-2 that stops after n sweeps must return exactly what a
    call that is GIVEN n sweeps and never converges (tol = 0: the rebuild always runs) returns."""
    import torch
    from eks_amd import hip_ops
    T, K, V = 1200, 3, 3
    prob = synth.calibrated_multicam(T, K, V, seed=77)
    s = np.exp(np.linspace(-3, 4, K))
    m0 = _dev(prob['m0s'])
    common = (_dev(prob['y_tko'], torch.float32), _dev(prob['var_tko'], torch.float32), None, m0, _dev(prob['S0s']),
              _dev(prob['As']), _dev(prob['Qs']), _dev(s), _dev(prob['cams_packed']))
    # (the sweeps converge fast: a tolerance is searched for at which they stop with a residual between the two thresholds)
    for tol in (3.0, 1.0, 0.3, 0.1, 3e-2, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
        x_a = m0[:, None, :].expand(K, T, 3).contiguous()
        ms_a, Vs_a, _, info_a = hip_ops.ekf_smooth(*common, x_a, max_sweeps=16, tol=tol)
        n = int(info_a[0].item())
        if 1 <= n < 16 and 1e-8 < info_a[1].item() <= tol:
            break
    else:
        pytest.skip('no tolerance leaves the residual between 1e-8 and itself on this problem')
    x_b = m0[:, None, :].expand(K, T, 3).contiguous()
    ms_b, Vs_b, _, _ = hip_ops.ekf_smooth(*common, x_b, max_sweeps=n, tol=0.0)
    assert torch.equal(ms_a, ms_b) and torch.equal(Vs_a, Vs_b)


def test_constant_r_loss_over_replicated_chains_and_vs_diag():
    import torch
    from eks_amd import hip_ops
    T, K, V, n_rep = 900, 2, 3, 3
    prob = synth.calibrated_multicam(T, K, V, seed=21)
    rconst = np.maximum(np.median(_f32(prob['var_tko']), axis=0), 1e-4)
    s = np.exp(np.linspace(-3, 3, n_rep * K))                       # chain c -> keypoint c % K
    rep = lambda a: _dev(np.tile(a, (n_rep,) + (1,) * (a.ndim - 1)))
    xlin = rep(prob['m0s'])[:, None, :].expand(n_rep * K, T, 3).contiguous()
    ms, Vs, nll, info = hip_ops.ekf_smooth(_dev(prob['y_tko'], torch.float32), None, _dev(rconst),
                                           rep(prob['m0s']), rep(prob['S0s']), rep(prob['As']),
                                           rep(prob['Qs']), _dev(s), _dev(prob['cams_packed']), xlin,
                                           max_sweeps=24, tol=1e-10, want_smoother=False)
    assert ms is None and Vs is None and info[1].item() <= 1e-10
    h = _oracle_h(prob)
    nll = nll.cpu().numpy()
    for c in range(n_rep * K):
        k = c % K
        ref = ek.ekf_nll(_f32(prob['y_tko'][:, k]), rconst[k], prob['m0s'][k], prob['S0s'][k],
                         prob['As'][k], prob['Qs'][k], s[c], h)
        assert abs(nll[c] - ref) < 1e-9 * abs(ref)
    # diagonal-only covariance output of the smoothing pass
    x2 = _dev(prob['m0s'])[:, None, :].expand(K, T, 3).contiguous()
    full = hip_ops.ekf_smooth(_dev(prob['y_tko'], torch.float32), _dev(prob['var_tko'], torch.float32),
                              None, _dev(prob['m0s']), _dev(prob['S0s']), _dev(prob['As']),
                              _dev(prob['Qs']), _dev(s[:K]), _dev(prob['cams_packed']), x2.clone())
    diag = hip_ops.ekf_smooth(_dev(prob['y_tko'], torch.float32), _dev(prob['var_tko'], torch.float32),
                              None, _dev(prob['m0s']), _dev(prob['S0s']), _dev(prob['As']),
                              _dev(prob['Qs']), _dev(s[:K]), _dev(prob['cams_packed']), x2.clone(),
                              vs_diag=True)
    assert torch.equal(full[0], diag[0])
    assert torch.equal(torch.diagonal(full[1], dim1=2, dim2=3), diag[1])


def test_abi_rejects_bad_arguments():
    import ctypes
    import torch
    from eks_amd import _lib, hip_ops
    prob = synth.calibrated_multicam(64, 2, 2, seed=1)
    m0 = _dev(prob['m0s'])
    xlin = m0[:, None, :].expand(2, 64, 3).contiguous()
    args = dict(y=_dev(prob['y_tko'], torch.float32), var=_dev(prob['var_tko'], torch.float32), rconst=None,
                m0=m0, S0=_dev(prob['S0s']), A=_dev(prob['As']), Q=_dev(prob['Qs']),
                s=_dev(np.ones(2)), cams=_dev(prob['cams_packed']), xlin=xlin)
    with pytest.raises(_lib.EksHipError):
        hip_ops.ekf_smooth(**args, max_sweeps=0)
    with pytest.raises(_lib.EksHipError):
        hip_ops.ekf_smooth(**args, max_sweeps=65)
    with pytest.raises(_lib.EksHipError):             # observation width must be 2 * n_cams
        hip_ops.ekf_smooth(**{**args, 'cams': _dev(prob['cams_packed'][:1])})
    with pytest.raises(ValueError):
        hip_ops.ekf_smooth(**{**args, 'rconst': _dev(np.ones((2, 4)))})
    lib = _lib.load()
    d = _lib.EksDims(2, 64, 3, 4, 0)
    assert lib.eks_ekf_smooth_workspace_bytes(ctypes.byref(d), 1) > \
        lib.eks_ekf_smooth_workspace_bytes(ctypes.byref(d), 0) > 0


@pytest.mark.parametrize('smooth_param', [4.0, [0.5, 30.0, 2.0]])
def test_run_kalman_smoother_with_projection_fixed_s(smooth_param):
    from eks_amd.core import run_kalman_smoother
    T, K, V = 800, 3, 2
    prob = synth.calibrated_multicam(T, K, V, seed=8)
    h = cal.PinholeProjection(prob['cams_packed'])
    ys = np.swapaxes(prob['y_tko'], 0, 1)
    s, ms, Vs = run_kalman_smoother(ys, prob['m0s'], prob['S0s'], prob['As'], np.tile(np.eye(3), (K, 1, 1)),
                                    prob['Qs'], prob['var_tko'], smooth_param=smooth_param, h_fn=h)
    assert ms.shape == (K, T, 3) and Vs.shape == (K, T, 3, 3) and ms.dtype == np.float32
    so, mo, Vo, _ = ek.run_kalman_smoother_nonlinear(_f32(ys), prob['m0s'], prob['S0s'], prob['As'],
                                                     prob['Qs'], _f32(prob['var_tko']), _oracle_h(prob),
                                                     smooth_param=smooth_param)
    np.testing.assert_array_equal(s, so)
    assert np.abs(ms - mo).max() < 1e-5 * np.abs(mo).max()
    assert np.abs(Vs - Vo).max() < 1e-5 * np.abs(Vo).max()


@pytest.mark.parametrize('blocks,s_frames', [(None, None), ([[0, 1]], [(30, 270)])])
def test_run_kalman_smoother_with_projection_optimises_s(blocks, s_frames):
    from eks_amd.core import run_kalman_smoother
    T, K, V = 300, 2, 2
    prob = synth.calibrated_multicam(T, K, V, seed=13)
    h = cal.PinholeProjection(prob['cams_packed'])
    ys = np.swapaxes(prob['y_tko'], 0, 1)
    s, ms, Vs = run_kalman_smoother(ys, prob['m0s'], prob['S0s'], prob['As'], np.tile(np.eye(3), (K, 1, 1)),
                                    prob['Qs'], prob['var_tko'], h_fn=h, blocks=blocks, s_frames=s_frames)
    so, mo, Vo, info = ek.run_kalman_smoother_nonlinear(_f32(ys), prob['m0s'], prob['S0s'], prob['As'],
                                                        prob['Qs'], _f32(prob['var_tko']), _oracle_h(prob),
                                                        blocks=blocks, s_frames=s_frames)
    assert np.abs(s / so - 1.0).max() < 1e-4
    if blocks:
        assert s[0] == s[1]
    # outputs given the SAME s: strict
    _, mo, Vo, _ = ek.run_kalman_smoother_nonlinear(_f32(ys), prob['m0s'], prob['S0s'], prob['As'],
                                                    prob['Qs'], _f32(prob['var_tko']), _oracle_h(prob),
                                                    smooth_param=list(s))
    assert np.abs(ms - mo).max() < 1e-5 * np.abs(mo).max()
    assert np.abs(Vs - Vo).max() < 1e-5 * np.abs(Vo).max()


def test_grid_mode_with_projection_picks_the_oracle_argmin():
    from eks_amd.core import run_kalman_smoother
    T, K, V = 400, 2, 2
    prob = synth.calibrated_multicam(T, K, V, seed=17)
    h = cal.PinholeProjection(prob['cams_packed'])
    ys = np.swapaxes(prob['y_tko'], 0, 1)
    s, _, _ = run_kalman_smoother(ys, prob['m0s'], prob['S0s'], prob['As'], np.tile(np.eye(3), (K, 1, 1)),
                                  prob['Qs'], prob['var_tko'], h_fn=h, s_mode='grid', n_grid=9)
    cand = np.exp(np.linspace(-8, 8, 9))
    ev = np.swapaxes(_f32(prob['var_tko']), 0, 1)
    oh = _oracle_h(prob)
    for k in range(K):
        rc = orc.constant_R_from_timevarying(np.maximum(ev[k], 1e-12), 1e-4)
        nll = [ek.ekf_nll(_f32(ys[k]), rc, prob['m0s'][k], prob['S0s'][k], prob['As'][k], prob['Qs'][k], c, oh)
               for c in cand]
        assert s[k] == cand[int(np.argmin(nll))]


def test_calibrated_multicam_driver_end_to_end():
    from eks_amd import MarkerArray
    from eks_amd.multicam_smoother import ensemble_kalman_smoother_multicam
    T, K, V = 600, 2, 3
    prob = synth.calibrated_multicam(T, K, V, seed=23)
    group = cal.CameraGroup([cal.Camera(c['rot'], c['tvec'], c['K'], c['dist'], name=f'cam{i}')
                             for i, c in enumerate(prob['cams'])])
    ma = MarkerArray(prob['markers'].astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    names = ['paw', 'nose']
    cams = [c.name for c in group.cameras]
    dfs, s, df3 = ensemble_kalman_smoother_multicam(ma, names, cams, smooth_param=[3.0, 0.7], camgroup=group)
    assert len(dfs) == V and all(isinstance(d, pd.DataFrame) and d.shape == (T, K * 9) for d in dfs)
    assert df3.shape == (T, K * 6)
    assert list(df3.columns.get_level_values('coords')[:6]) == \
        ['x', 'y', 'z', 'x_posterior_var', 'y_posterior_var', 'z_posterior_var']
    np.testing.assert_array_equal(s, [3.0, 0.7])
    # oracle restatement of the same pipeline (eks/multicam_smoother.py:367-407, :450-480)
    ens = orc.ensemble(prob['markers'].astype(np.float64))                 # (1,V,T,K,5)
    st = ens[0]
    tri = np.stack([np.stack([ek.triangulate_dlt(prob['cams'], prob['markers'][m, :, :, k, :2].astype(np.float64))
                              for k in range(K)]) for m in range(prob['markers'].shape[0])])
    m0s, S0s, As, Qs, _ = ek.initialize_kalman_filter_geometric(tri.mean(axis=0))
    ys = np.transpose(st[..., 0:2], (2, 1, 0, 3)).reshape(K, T, 2 * V)
    evs = np.transpose(st[..., 2:4], (2, 1, 0, 3)).reshape(K, T, 2 * V)
    oh = _oracle_h(prob)
    heads = [ek.make_projection_fn(c['rot'], c['tvec'], c['K'], c['dist']) for c in prob['cams']]
    _, mo, Vo, _ = ek.run_kalman_smoother_nonlinear(_f32(ys), m0s, S0s, As, Qs, np.swapaxes(_f32(evs), 0, 1),
                                                    oh, smooth_param=[3.0, 0.7])
    lat = df3.values.reshape(T, K, 6)
    assert np.abs(lat[:, :, :3] - np.swapaxes(mo, 0, 1)).max() < 1e-5 * np.abs(mo).max()
    for c in range(V):
        got = dfs[c].values.reshape(T, K, 9)
        for k in range(K):
            xy = heads[c](mo[k])
            vx, vy = ek.project_3d_covariance_to_2d(mo[k], Vo[k], heads[c], evs[k])
            assert np.abs(got[:, k, 0:2] - xy).max() < 1e-5 * np.abs(xy).max()
            assert np.abs(got[:, k, 7] - vx).max() < 1e-5 * np.abs(vx).max()
            assert np.abs(got[:, k, 8] - vy).max() < 1e-5 * np.abs(vy).max()
            np.testing.assert_allclose(got[:, k, 3:5], st[c, :, k, 0:2], rtol=1e-6)
            np.testing.assert_allclose(got[:, k, 5:7], st[c, :, k, 2:4], rtol=1e-5)
            np.testing.assert_allclose(got[:, k, 2], st[c, :, k, 4], rtol=1e-6)


def test_optimize_smooth_param_with_projection_writes_s_in_place():
    from eks_amd.core import optimize_smooth_param, run_kalman_smoother
    T, K, V = 300, 2, 2
    prob = synth.calibrated_multicam(T, K, V, seed=13)
    h = cal.PinholeProjection(prob['cams_packed'])
    ys = np.swapaxes(prob['y_tko'], 0, 1)
    Rs = np.stack([[np.diag(r) for r in np.maximum(prob['var_tko'][:, k], 1e-12)] for k in range(K)])
    guesses = [orc.compute_initial_guess(prob['var_tko'][:, k, :]) for k in range(K)]
    s_finals = np.zeros(K)
    optimize_smooth_param(ys, prob['m0s'], prob['S0s'], prob['As'], None, prob['Qs'], Rs, None, s_finals,
                          None, guesses, tol=1e-2, h_fn_combined=h)
    s_ref, _, _ = run_kalman_smoother(ys, prob['m0s'], prob['S0s'], prob['As'], None, prob['Qs'],
                                      prob['var_tko'], h_fn=h)
    np.testing.assert_allclose(s_finals, s_ref, rtol=1e-12)


def _rotation_vector(R):
    """Inverse of Rodrigues for the synthetic rig (angles well inside (0, pi))."""
    ang = np.arccos(np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0))
    axis = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / (2.0 * np.sin(ang))
    return axis * ang


def test_fit_eks_multicam_with_calibration_file(tmp_path):
    """CSV files + aniposelib-style calibration TOML -> per-camera CSVs and the 3-D table
    (reference eks/multicam_smoother.py:156-276 with `calibration`)."""
    from eks_amd import MarkerArray
    from eks_amd.multicam_smoother import ensemble_kalman_smoother_multicam, fit_eks_multicam
    T, K, V, M = 300, 2, 3, 3
    prob = synth.calibrated_multicam(T, K, V, M=M, seed=31)
    names, cams = ['paw', 'nose'], ['camA', 'camB', 'camC']
    toml = []
    for i, (c, cam) in enumerate(zip(prob['cams'], cams)):
        fmt = lambda a: '[ ' + ', '.join(repr(float(x)) for x in np.ravel(a)) + ',]'
        mat = '[ ' + ', '.join(fmt(row) for row in c['K']) + ',]'
        toml.append(f'[cam_{i}]\nname = "{cam}"\nsize = [ 640, 480,]\nmatrix = {mat}\n'
                    f'distortions = {fmt(c["dist"])}\nrotation = {fmt(_rotation_vector(c["rot"]))}\n'
                    f'translation = {fmt(c["tvec"])}\n')
    calib = tmp_path / 'calibration.toml'
    calib.write_text('\n'.join(toml) + '\n[metadata]\nadjusted = false\n')
    cols = pd.MultiIndex.from_product([['net'], names, ['x', 'y', 'likelihood']],
                                      names=['scorer', 'bodyparts', 'coords'])
    src = tmp_path / 'in'
    src.mkdir()
    for m in range(M):
        for v, cam in enumerate(cams):
            pd.DataFrame(prob['markers'][m, v].reshape(T, K * 3).astype(np.float64), columns=cols).to_csv(
                src / f'seed{m}_{cam}.csv')
    out = tmp_path / 'out'
    dfs, s, input_dfs, bodyparts, df3 = fit_eks_multicam(str(src), str(out), smooth_param=[2.0, 5.0],
                                                         calibration=str(calib))
    assert bodyparts == names and len(dfs) == V and len(input_dfs) == V and len(input_dfs[0]) == M
    for cam in cams:
        assert (out / f'multicam_{cam}_results.csv').exists()
    assert (out / 'multicam_3d_results.csv').exists()
    back = pd.read_csv(out / 'multicam_camB_results.csv', header=[0, 1, 2], index_col=0)
    np.testing.assert_allclose(back.values, dfs[1].values, rtol=1e-12)
    # the same answer as the in-memory driver on the same markers and cameras
    group = cal.CameraGroup.load(str(calib))
    ma = MarkerArray(prob['markers'].astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    dfs2, s2, df3b = ensemble_kalman_smoother_multicam(ma, names, cams, smooth_param=[2.0, 5.0], camgroup=group)
    np.testing.assert_array_equal(s, s2)
    for a, b in zip(dfs, dfs2):
        assert np.abs(a.values - b.values).max() < 1e-4 * np.abs(b.values).max()   # CSV text round trip of inputs
    # the 3-D track follows the latent points the markers were generated from
    lat = df3.values.reshape(T, K, 6)[:, :, :3]
    assert np.abs(lat - prob['latent']).max() < 5.0


# ---- the reference's own calibrated data set (data/fly + calibration.toml), tests/golden ----------
def _fly_against_golden(df_values, g, prefix, K_cols, tol=1e-5):
    rows = df_values[g['keep_idx']]
    ref = g[f'{prefix}_rows'].astype(np.float64)
    scale = np.abs(ref).max(axis=0)
    assert (np.abs(rows - ref) / scale).max() < tol
    assert (np.abs(df_values.sum(axis=0) - g[f'{prefix}_colsum']) / g[f'{prefix}_colabs']).max() < tol


def test_fly_calibrated_multicam_matches_golden(golden_dir, tmp_path):
    """Inputs: the reference's data/fly predictions (3 cameras x 3 members x 500 frames x 12
    keypoints) and its calibration.toml; expected: oracle/ekf_oracle.py (tools/make_golden.py fly)."""
    import os
    from eks_amd import MarkerArray
    from eks_amd.multicam_smoother import ensemble_kalman_smoother_multicam
    g = np.load(os.path.join(golden_dir, 'fly_calibrated_multicam.npz'))
    fn = tmp_path / 'calibration.toml'
    fn.write_text(str(g['toml']))
    group = cal.CameraGroup.load(str(fn))
    names, cams = list(g['keypoints']), list(g['cameras'])
    ma = MarkerArray(g['markers'].astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    dfs, s, df3 = ensemble_kalman_smoother_multicam(ma, names, cams, smooth_param=10.0, camgroup=group)
    np.testing.assert_array_equal(s, 10.0)
    for c in range(3):
        assert list(dfs[c].columns.get_level_values('bodyparts')[::9]) == names
        _fly_against_golden(dfs[c].values, g, f's10_cam{c}', len(names))
    _fly_against_golden(df3.values, g, 's10_latent', len(names), tol=2e-5)
    # the reference's default: optimise s (first two keypoints)
    ma2 = MarkerArray(g['markers'][:, :, :, :2].astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    dfs, s, _ = ensemble_kalman_smoother_multicam(ma2, names[:2], cams, camgroup=group)
    assert np.abs(s / g['adam_s'] - 1.0).max() < 1e-3
    for c in range(3):
        _fly_against_golden(dfs[c].values, g, f'adam_cam{c}', 2, tol=1e-4)


def test_fly_integration_configuration_with_variance_inflation(golden_dir, tmp_path):
    """The reference's integration test on this data (tests/integration/test_multicam.py:46-58):
    bodyparts L1A, L1B, quantile_keep_pca=95, inflate_vars=True, smooth_param=[10.0]."""
    import os
    from eks_amd import MarkerArray
    from eks_amd.multicam_smoother import ensemble_kalman_smoother_multicam
    g = np.load(os.path.join(golden_dir, 'fly_calibrated_multicam.npz'))
    fn = tmp_path / 'calibration.toml'
    fn.write_text(str(g['toml']))
    group = cal.CameraGroup.load(str(fn))
    names, cams = list(g['keypoints'][:2]), list(g['cameras'])
    assert names == ['L1A', 'L1B']
    ma = MarkerArray(g['markers'][:, :, :, :2].astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    dfs, s, _ = ensemble_kalman_smoother_multicam(ma, names, cams, smooth_param=[10.0], quantile_keep_pca=95,
                                                  inflate_vars=True, camgroup=group)
    np.testing.assert_array_equal(s, 10.0)
    for c in range(3):
        _fly_against_golden(dfs[c].values, g, f'infl_s10_cam{c}', 2)


def test_fly_integration_defaults_optimise_s_with_variance_inflation(golden_dir, tmp_path):
    """tests/integration/test_multicam.py:32-44 of the reference: the same call with the default
    smooth_param=None (Adam on log s through the extended filter)."""
    import os
    from eks_amd import MarkerArray
    from eks_amd.multicam_smoother import ensemble_kalman_smoother_multicam
    g = np.load(os.path.join(golden_dir, 'fly_calibrated_multicam.npz'))
    fn = tmp_path / 'calibration.toml'
    fn.write_text(str(g['toml']))
    group = cal.CameraGroup.load(str(fn))
    ma = MarkerArray(g['markers'][:, :, :, :2].astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    dfs, s, _ = ensemble_kalman_smoother_multicam(ma, ['L1A', 'L1B'], list(g['cameras']), quantile_keep_pca=95,
                                                  inflate_vars=True, camgroup=group)
    assert np.abs(s / g['infl_adam_s'] - 1.0).max() < 1e-3
    for c in range(3):
        _fly_against_golden(dfs[c].values, g, f'infl_adam_cam{c}', 2, tol=1e-4)
