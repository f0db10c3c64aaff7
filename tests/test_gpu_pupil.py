"""GPU: the pupil-smoother row (reference eks/ibl_pupil_smoother.py) through
the C ABI (eks_ar1_nll, eks_pupil_adam_step, eks_smooth) and through the reference-shaped drivers,
against the float64 oracle and the committed golden vectors (inputs = the reference's
data/ibl-pupil files).  Tolerances: loss / gradient are float64 kernels -> 1e-9 relative; smoothed
outputs are float32 arrays -> 1e-5 relative to the column's magnitude (BASELINE.json)."""
import os

import numpy as np
import pandas as pd
import pytest

from oracle import eks_oracle as orc

pytestmark = pytest.mark.gpu


def synth_pupil(T, seed):
    from eks_amd import synth
    return synth.pupil_observations(T, seed)


LABELS = ['x', 'y', 'likelihood', 'x_ens_median', 'y_ens_median', 'x_ens_var', 'y_ens_var',
          'x_posterior_var', 'y_posterior_var']


@pytest.fixture(scope='module')
def gold(golden_dir):
    g = np.load(os.path.join(golden_dir, 'ibl_pupil_pupil.npz'))
    src = np.load(os.path.join(golden_dir, 'ibl_pupil_singlecam.npz'))
    return g, src['markers'][:, :, :, g['order']], str(src['scorer'])


def _loss(ys_list, ev_list, m0_list, S0_list, n_tan, positive_noise=False):
    """Ar1Loss over K independent chains of equal length."""
    import torch
    from eks_amd import hip_ops
    dev = hip_ops.require_gpu()
    y = torch.as_tensor(np.stack(ys_list, axis=1).astype(np.float32), device=dev)
    var = torch.as_tensor(np.stack(ev_list, axis=1).astype(np.float32), device=dev)
    K = len(ys_list)
    m0 = torch.as_tensor(np.stack(m0_list), device=dev)
    S0 = torch.as_tensor(np.stack(S0_list), device=dev)
    C = torch.as_tensor(np.tile(orc.PUPIL_C, (K, 1, 1)), device=dev)
    return hip_ops.Ar1Loss(y, var, m0, S0, C, n_tan=n_tan, positive_noise=positive_noise)


def _fill(loss, us, lvs):
    import torch
    K = len(us)
    a, q = np.empty((K, 3)), np.empty((K, 3))
    da, dq = np.zeros((2, K, 3)), np.zeros((2, K, 3))
    for k, (u, lv) in enumerate(zip(us, lvs)):
        s, ds = orc.pupil_to_stable_s(np.asarray(u))
        a[k] = [s[0], s[1], s[1]]
        q[k] = lv * (1 - a[k] ** 2)
        da[0, k, 0] = ds[0]
        da[1, k, 1:] = ds[1]
        dq[0, k, 0] = -2 * s[0] * ds[0] * lv[0]
        dq[1, k, 1:] = -2 * s[1] * ds[1] * lv[1:]
    loss.a.copy_(torch.as_tensor(a))
    loss.q.copy_(torch.as_tensor(q))
    if loss.n_tan:
        loss.da.copy_(torch.as_tensor(da))
        loss.dq.copy_(torch.as_tensor(dq))


@pytest.mark.parametrize('positive_noise', [False, True])     # dual numbers / smoothing-distribution derivatives
@pytest.mark.parametrize('T', [1, 2, 7, 64, 1000, 5003])
def test_ar1_nll_and_sensitivities_match_oracle(T, positive_noise):
    probs = [synth_pupil(T, seed=10 + k) for k in range(3)]
    for p in probs:                                   # T = 1: the variance over one frame is 0
        p[3][np.diag_indices(3)] = np.maximum(np.diag(p[3]), 0.3)
    lvs = [np.maximum(p[4], 0.3) for p in probs]
    us = [(4.6, 3.9), (0.5, -1.0), (-2.0, 6.0)]
    loss = _loss([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs], [p[3] for p in probs], 2,
                 positive_noise=positive_noise)
    _fill(loss, us, lvs)
    nll, dnll = loss.evaluate()
    nll, dnll = nll.cpu().numpy(), dnll.cpu().numpy()
    for k, (p, u, lv) in enumerate(zip(probs, us, lvs)):
        L, g = orc.pupil_nll_and_grad(np.asarray(u), p[0], p[2], p[3], orc.PUPIL_C, p[1], lv, use_c=False)
        assert abs(nll[k] - L) < 1e-9 * max(abs(L), 1.0)
        np.testing.assert_allclose(dnll[:, k], g, rtol=1e-8, atol=1e-9 * max(np.abs(g).max(), 1.0))
    # value-only variant (n_tan = 0) returns the same loss
    loss0 = _loss([p[0] for p in probs], [p[1] for p in probs], [p[2] for p in probs], [p[3] for p in probs], 0)
    _fill(loss0, us, lvs)
    np.testing.assert_allclose(loss0.evaluate()[0].cpu().numpy(), nll, rtol=1e-12)


@pytest.mark.parametrize('positive_noise', [False, True])
def test_ar1_nll_on_golden_probe_points(gold, positive_noise):
    g, mk, _ = gold
    arrs = orc.pupil_arrays(mk)
    loss = _loss([arrs['ys']], [arrs['ensemble_vars']], [arrs['m0']], [arrs['S0']], 2, positive_noise=positive_noise)
    y32 = arrs['ys'].astype(np.float32).astype(np.float64)      # what the device buffers hold
    v32 = arrs['ensemble_vars'].astype(np.float32).astype(np.float64)
    for u, L, gr in zip(g['probe_u'], g['probe_nll'], g['probe_grad']):
        _fill(loss, [u], [arrs['latent_vars']])
        nll, dnll = loss.evaluate()
        nll, dnll = float(nll[0]), dnll[:, 0].cpu().numpy()
        Lo, go = orc.pupil_nll_and_grad(u, y32, arrs['m0'], arrs['S0'], orc.PUPIL_C, v32, arrs['latent_vars'])
        assert abs(nll - Lo) < 1e-9 * abs(Lo)
        np.testing.assert_allclose(dnll, go, rtol=1e-7, atol=1e-8 * np.abs(go).max())
        # the committed vectors were computed from the unrounded float64 inputs
        assert abs(nll - L) < 1e-6 * abs(L)
        np.testing.assert_allclose(dnll, gr, rtol=1e-4, atol=1e-5 * np.abs(gr).max())


def test_ar1_nll_rejects_bad_arguments():
    import ctypes

    import torch
    from eks_amd import _lib
    lib = _lib.load()
    d = _lib.EksDims(1, 10, 3, 8, 0)
    assert lib.eks_ar1_nll_workspace_bytes(ctypes.byref(d), 2) > 0
    assert lib.eks_ar1_nll_workspace_bytes(ctypes.byref(d), -1) == 0
    z = ctypes.c_void_p(0)
    one = torch.zeros(8, device='cuda')
    p = ctypes.c_void_p(one.data_ptr())
    assert lib.eks_ar1_nll(ctypes.byref(d), z, z, z, z, z, z, z, z, z, 0, z, z, z, 0, z) == -1
    assert lib.eks_ar1_nll(ctypes.byref(d), p, p, p, p, p, p, p, p, p, 2, p, z, p, 1 << 20, z) == -1   # dnll missing
    assert lib.eks_ar1_nll(ctypes.byref(d), p, p, p, p, p, p, p, z, z, 0, p, z, p, 16, z) == -4         # workspace
    d7 = _lib.EksDims(1, 10, 7, 8, 0)
    assert lib.eks_ar1_nll(ctypes.byref(d7), p, p, p, p, p, p, p, z, z, 0, p, z, p, 1 << 30, z) == -3
    assert lib.eks_pupil_adam_step(0, p, p, p, 5e-3, 1e-6, 10, p, p, p, p, p, p, z) == -2
    assert lib.eks_pupil_adam_step(1, p, z, z, 5e-3, 1e-6, 10, z, p, p, p, p, p, z) == -1


@pytest.mark.parametrize('cap', [1, 37])
def test_device_adam_follows_the_oracle_trajectory(cap):
    from eks_amd import ibl_pupil_smoother as ips
    ys, ev, m0, S0, lv = synth_pupil(600, seed=21)
    P = ips._PupilProblem(ys, m0, S0, orc.PUPIL_C, ev, lv)
    s_d, s_c, info = ips._optimize_on_device(P, None, 5e-3, 1e-6, cap, sync_every=5)
    o = orc.pupil_optimize_smooth(ys, m0, S0, orc.PUPIL_C, ev, lv, safety_cap=cap)
    assert info['iters'] == o[2] == cap and not info['converged']
    np.testing.assert_allclose([s_d, s_c], o[:2], rtol=1e-10)
    assert abs(info['last_loss'] - o[3]) < 1e-9 * abs(o[3])


def test_device_adam_stops_like_the_oracle_with_loose_tolerance_and_crop():
    from eks_amd import ibl_pupil_smoother as ips
    ys, ev, m0, S0, lv = synth_pupil(900, seed=22)
    frames = [(100, 400), (500, None)]
    s_d, s_c = ips.pupil_optimize_smooth(ys, m0, S0, orc.PUPIL_C, orc.build_R_from_vars(ev), *lv,
                                         s_frames=frames, tol=1e-4, lr=2e-2)
    o = orc.pupil_optimize_smooth(ys, m0, S0, orc.PUPIL_C, ev, lv, s_frames=frames, tol=1e-4, lr=2e-2)
    assert 1 < o[2] < 5000
    np.testing.assert_allclose([s_d, s_c], o[:2], rtol=1e-9)
    with pytest.raises(ValueError):
        ips.pupil_optimize_smooth(ys, m0, S0, orc.PUPIL_C, ev, *lv, s_frames=[(0, 5000)])


@pytest.mark.parametrize('sp', [[0.99, 0.99], [0.5, 0.5], [0.999, 0.2]])
def test_run_pupil_kalman_smoother_fixed_params(sp):
    from eks_amd.ibl_pupil_smoother import run_pupil_kalman_smoother
    ys, ev, m0, S0, lv = synth_pupil(3001, seed=23)
    s, ms, Vs = run_pupil_kalman_smoother(ys, m0, S0, orc.PUPIL_C, ev, *lv, smooth_params=sp)
    so, mo, Vo, _ = orc.run_pupil_kalman_smoother(ys, m0, S0, orc.PUPIL_C, ev, lv, smooth_params=sp)
    assert s == so and isinstance(s, list) and ms.shape == (3001, 3) and Vs.shape == (3001, 3, 3)
    assert (np.abs(ms - mo) / np.abs(mo).max(axis=0)).max() < 1e-5
    assert (np.abs(Vs - Vo) / np.abs(Vo).max(axis=0)).max() < 1e-5


def test_driver_fixed_params_matches_golden(gold):
    from eks_amd import MarkerArray
    from eks_amd.ibl_pupil_smoother import PUPIL_BODYPARTS, ensemble_kalman_smoother_ibl_pupil
    g, mk, _ = gold
    ma = MarkerArray(mk.astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    df, s = ensemble_kalman_smoother_ibl_pupil(ma, list(PUPIL_BODYPARTS), smooth_params=[0.99, 0.99])
    assert isinstance(df, pd.DataFrame) and df.shape == (2000, 36) and s == list(g['fixed_s'])
    assert list(df.columns.get_level_values('coords')[:9]) == LABELS
    assert list(df.columns.get_level_values('bodyparts')[::9]) == PUPIL_BODYPARTS
    rows = df.values[g['keep_idx']]
    ref = g['fixed_rows'].astype(np.float64)
    assert (np.abs(rows - ref) / np.abs(ref).max(axis=0)).max() < 1e-5
    assert (np.abs(df.values.sum(axis=0) - g['fixed_colsum']) / g['fixed_colabs']).max() < 1e-5


def test_driver_optimised_matches_golden_and_upstream_unit_test_contract(gold):
    from eks_amd import MarkerArray
    from eks_amd.ibl_pupil_smoother import PUPIL_BODYPARTS, ensemble_kalman_smoother_ibl_pupil
    g, mk, _ = gold
    ma = MarkerArray(mk.astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    df, s = ensemble_kalman_smoother_ibl_pupil(ma, list(PUPIL_BODYPARTS), smooth_params=[None, None])
    np.testing.assert_allclose(s, g['adam_s'], rtol=1e-6)
    rows = df.values[g['keep_idx']]
    ref = g['adam_rows'].astype(np.float64)
    assert (np.abs(rows - ref) / np.abs(ref).max(axis=0)).max() < 1e-5
    # reference tests/test_ibl_pupil_smoother.py:176-222: random 100-frame inputs, two members,
    # s_frames, mean / var modes; fixed params pass through, optimised ones are < 1
    rng = np.random.default_rng(0)
    small = MarkerArray(rng.normal(size=(2, 1, 100, 4, 3)), data_fields=['x', 'y', 'likelihood'])
    for sp in ([0.5, 0.5], [None, None], None):
        df, s = ensemble_kalman_smoother_ibl_pupil(small, list(PUPIL_BODYPARTS), sp, [(1, 20)],
                                                   avg_mode='mean', var_mode='var', safety_cap=50)
        assert df.shape[0] == 100 and len(s) == 2 and s[0] < 1 and s[1] < 1
        if sp == [0.5, 0.5]:
            assert s == sp


def test_fit_eks_pupil_from_csv(gold, tmp_path):
    from eks_amd import fit_eks_pupil
    g, mk, scorer = gold
    names = list(orc.PUPIL_KEYPOINTS)
    # file column order differs from the smoother's required order: input_dfs_to_markerArray
    # selects by name
    file_order = [0, 2, 1, 3]
    cols = pd.MultiIndex.from_product([[scorer], [names[i] for i in file_order], ['x', 'y', 'likelihood']],
                                      names=['scorer', 'bodyparts', 'coords'])
    for m in range(5):
        pd.DataFrame(mk[m, 0][:, file_order].reshape(2000, 12).astype(np.float64), columns=cols
                     ).to_csv(tmp_path / f'pupil.rng={m}.csv')
    save = tmp_path / 'out' / 'eks_pupil.csv'
    df, s, input_dfs, bps = fit_eks_pupil(str(tmp_path), str(save), smooth_params=[0.99, 0.99])
    assert bps == names and len(input_dfs) == 5 and save.exists()
    ref = g['fixed_rows'].astype(np.float64)
    assert (np.abs(df.values[g['keep_idx']] - ref) / np.abs(ref).max(axis=0)).max() < 1e-5
    back = pd.read_csv(save, header=[0, 1, 2], index_col=0)
    np.testing.assert_allclose(back.values, df.values, rtol=1e-12)
