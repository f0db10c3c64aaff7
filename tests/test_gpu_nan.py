"""GPU: observations that are NaN.  When every ensemble member is NaN at a frame the reference's `ensemble` yields
x = NaN, var = 1000 (eks/core.py:56, :64-65, :82-83) and dynamax has no NaN handling: the filtered mean is NaN from that
frame on, the RTS pass carries it back to frame 0, the covariances - which never see y - stay finite, the loss is
non-finite -> 1e12 for every s (eks/core.py:650), and under `vmap` no other keypoint is touched (eks/core.py:293, :684).
The kernels must reproduce exactly that: the poisoned keypoint's NaN pattern and its finite outputs match the oracle,
and every OTHER keypoint - in particular the other 63 lanes of its 64-chain tile, which share waves, ballots and
wave-uniform regime decisions with it - is bit-identical to a run in which that keypoint is healthy."""
import warnings

import numpy as np
import pytest

from oracle import eks_oracle as orc

pytestmark = pytest.mark.gpu


def _session(T, K, seed, dense):
    import torch
    from eks_amd import synth
    rng = np.random.default_rng(seed)
    if dense:
        D, O = 3, 4
        x = np.cumsum(rng.standard_normal((K, T, D)) * 0.5, axis=1)
        Cs = rng.standard_normal((K, O, D))
        ev = (rng.gamma(2.0, 0.4, (T, K, O)) + 0.02).astype(np.float32)
        ys = (np.einsum('kod,ktd->kto', Cs, x) + rng.standard_normal((K, T, O)) * np.sqrt(np.swapaxes(ev, 0, 1))).astype(np.float32)
        L = rng.standard_normal((K, D, D)) * 0.3
        Qs = L @ np.swapaxes(L, 1, 2) + 0.2 * np.eye(D)
        return [ys, np.zeros((K, D)), np.tile(np.eye(D) * 3.0, (K, 1, 1)), np.tile(np.eye(D), (K, 1, 1)), Cs, Qs, ev]
    y, var = synth.singlecam_observations_torch(T, K, seed=seed, device=torch.device('cuda', 0))
    ys = np.ascontiguousarray(np.transpose(y.cpu().numpy(), (1, 0, 2)))
    ev = var.cpu().numpy().copy()
    eye = np.tile(np.eye(2), (K, 1, 1))
    return [ys, np.zeros((K, 2)), eye * ys.var(axis=1)[:, :, None], eye, eye, eye, ev]


@pytest.mark.parametrize('mode,T,K', [('fixed', 12_000, 70), ('grid', 7_000, 70), ('grid_nolag', 7_000, 70),
                                      ('adam', 6_000, 70), ('adam_small', 2_000, 6), ('dense_fixed', 4_000, 36),
                                      ('fixed_small', 700, 5)])
def test_nan_observations_poison_only_their_keypoint(mode, T, K, set_knob):
    from eks_amd.core import run_kalman_smoother
    dense = mode == 'dense_fixed'
    k0 = 5 if K > 5 else 2                     # a lane in the middle of the first 64-chain tile
    healthy = _session(T, K, seed=31, dense=dense)
    gaps = [(T // 3, T // 3 + 4), (T // 2, T // 2 + 1), (T - 40, T - 37)]
    for a, b in gaps:
        healthy[6][a:b, k0, :] = 1000.0        # nan_replacement (eks/core.py:82-83): identical in both runs
    poisoned = [np.array(x, copy=True) for x in healthy]
    for a, b in gaps:
        poisoned[0][k0, a:b, :] = np.nan
    if mode == 'grid_nolag':
        set_knob('EKS_NLL_NOLAG', '1')             # the round-4 summaries (every candidate by the recursion)
    kw = {'fixed': dict(smooth_param=7.5), 'fixed_small': dict(smooth_param=2.0), 'dense_fixed': dict(smooth_param=4.0),
          'grid': dict(s_mode='grid', n_grid=64), 'grid_nolag': dict(s_mode='grid', n_grid=64),
          'adam': dict(safety_cap=60), 'adam_small': dict(safety_cap=60)}[mode]
    s1, ms1, Vs1 = run_kalman_smoother(*poisoned, **kw)
    s0, ms0, Vs0 = run_kalman_smoother(*healthy, **kw)
    others = np.arange(K) != k0
    # (a) nobody else is touched: bit for bit
    np.testing.assert_array_equal(s1[others], s0[others])
    np.testing.assert_array_equal(ms1[others], ms0[others])
    np.testing.assert_array_equal(Vs1[others], Vs0[others])
    assert np.isfinite(ms1[others]).all() and np.isfinite(Vs1[others]).all()
    # (b) the poisoned keypoint is what the reference's arithmetic makes of it
    sl = slice(k0, k0 + 1)
    one = [poisoned[0][sl].astype(np.float64)] + [np.asarray(x[sl], np.float64) for x in poisoned[1:6]] + \
          [poisoned[6][:, sl].astype(np.float64)]
    okw = {k: v for k, v in kw.items()}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        s_o, ms_o, Vs_o, info = orc.run_kalman_smoother(*one, **okw)
    assert np.isnan(ms_o).all() and np.isfinite(Vs_o).all()           # (what this test is about)
    assert np.isnan(ms1[k0]).all()
    assert np.isfinite(Vs1[k0]).all()
    if mode.startswith('grid'):
        assert info['argmin'][0] == 0                                  # every loss is 1e12: numpy's argmin takes the first
        assert s1[k0] == np.exp(-8.0)
    if mode.startswith('adam'):
        assert info['iters'][0] == 2                                   # loss 1e12, gradient 0: the stop rule fires at once
    np.testing.assert_allclose(s1[k0], s_o[0], rtol=1e-6 if mode.startswith('adam') else 1e-12)
    scale = np.abs(Vs_o).max()
    assert np.abs(Vs1[k0] - Vs_o[0]).max() / scale < 1e-5
