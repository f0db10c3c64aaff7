"""The identities behind the Adam search from cached lag sums (eks_amd/csrc/eks_lag_adam.hip), stated in NumPy float64
(tools/lag_adam_proto.py) and held to the oracle's filter on the CPU: closed-form variances of the head, the linear scan
of the innovations, the steady remainder as a polynomial in the pole with the lag sums as coefficients."""
import os
import sys

import numpy as np
import pytest

from oracle import eks_oracle as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
import lag_adam_proto as lp          # noqa: E402


def _chains(T, N, seed, unit):
    rng = np.random.default_rng(seed)
    q_true = np.exp(rng.uniform(-2, 1, N))
    y = (np.cumsum(rng.standard_normal((T, N)) * np.sqrt(q_true), axis=0) + rng.uniform(50, 300, N)
         + rng.standard_normal((T, N)) * 0.6).astype(np.float32).astype(np.float64)
    a = np.ones(N) if unit else rng.uniform(0.95, 1.0, N)
    c = np.ones(N) if unit else rng.uniform(0.6, 1.4, N)
    q = np.ones(N) if unit else rng.uniform(0.5, 2.0, N)
    r = rng.uniform(0.2, 1.5, N)
    return y, rng.standard_normal(N) + y[0], rng.uniform(1.0, 500.0, N), a, c, q, r


def _oracle_nll(y, m0, P0, a, c, q, r, s):
    """oracle/eks_oracle.py: filter_nll with its forward-mode gradient, one 1-D model per chain."""
    N = y.shape[1]
    one = np.ones((N, 1, 1))
    return orc.filter_nll(y.T[:, :, None], m0[:, None], P0[:, None, None] * one, a[:, None, None] * one,
                          c[:, None, None] * one, q[:, None, None] * one, s, r[:, None], want_grad=True)


@pytest.mark.parametrize('unit', [True, False])
def test_lag_form_is_the_filters_loss_and_gradient(unit):
    T, N = 1500, 12
    y, m0, P0, a, c, q, r = _chains(T, N, 5 + unit, unit)
    pre = lp.precompute(y, a)
    th = np.random.default_rng(0).uniform(-2.5, 3.0, N)
    v, g, rho = lp.lag_loss(th, pre, m0, P0, a, c, q, r)
    v_o, g_o = _oracle_nll(y, m0, P0, a, c, q, r, np.exp(th))
    assert np.abs(rho).max() < 0.9                       # (inside the range 256 lag sums cover)
    assert np.abs(v / v_o - 1).max() < 1e-11
    assert np.abs(g - g_o).max() < 1e-8 * np.abs(g_o).max()


def test_lag_sums_truncation_is_bounded_as_the_kernel_assumes():
    """|rho|^256 <= 1e-10 (1 - |rho|) bounds the dropped tail 2 sum_{k>=256} rho^k c_k against c_0 whatever the data
    (|c_k| <= c_0): at the edge of the range the loss still agrees to 1e-9; well beyond it (rho ~ 0.97) it does not -
    which is why the kernel streams those evaluations instead."""
    T, N = 6000, 6
    rng = np.random.default_rng(3)
    # smooth trajectories: the inputs u are positively correlated over hundreds of frames (the worst case for truncation)
    y = (np.cumsum(np.cumsum(rng.standard_normal((T, N)) * 0.01, axis=0), axis=0) + 100).astype(np.float32).astype(np.float64)
    m0, P0, a, c, q = y[0].copy(), np.full(N, 25.0), np.ones(N), np.ones(N), np.ones(N)
    r = np.full(N, 1.0)
    pre = lp.precompute(y, a, L=255)
    for rho_t, bar, inside in ((0.90, 1e-9, True), (0.97, 1e-9, False)):
        # s q such that the steady pole is rho_t:  s q = r (1 - p)^2 / p  for a = c = 1
        th = np.log(np.full(N, (1 - rho_t) ** 2 / rho_t))
        v, g, rho = lp.lag_loss(th, pre, m0, P0, a, c, q, r)
        assert np.allclose(rho, rho_t, atol=1e-12)
        v_o, _ = _oracle_nll(y, m0, P0, a, c, q, r, np.exp(th))
        err = np.abs(v / v_o - 1).max()
        assert (err < bar) == inside, (rho_t, err)
