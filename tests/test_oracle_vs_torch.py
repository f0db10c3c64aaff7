"""CPU: the oracle's restatements of the THIRD-PARTY pieces of the reference's recipe against PyTorch's own,
independently written implementations of the same published definitions:

  reference (eks/core.py:640-675)                          here, from torch
  ------------------------------------------------------   -----------------------------------------------
  tfp MultivariateNormalFullCovariance.log_prob            torch.distributions.MultivariateNormal.log_prob
    (inside dynamax's extended_kalman_filter, :648)
  jax.value_and_grad (reverse mode) of the NLL, :652       torch.autograd through the whole filter
  optax.adam(learning_rate=1.0) on lr-scaled grads, :654   torch.optim.Adam(lr=1.0, betas=(.9,.999), eps=1e-8)

The filter recursion itself (update-then-predict) is written out again below
with torch.linalg.solve - a third statement of it besides oracle/eks_oracle.py and oracle/eks_oracle.c.
This does NOT pin parity with upstream (only reference-produced numbers could, and jax / dynamax / optax
are absent here); it removes "shared misreading of the MVN log-density, of reverse-mode AD through the
filter, or of Adam's bias correction / epsilon placement" from the list of ways oracle and kernels
could agree with each other and still be wrong.
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import eks_oracle as orc

torch.set_num_threads(2)
F64 = torch.float64


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=F64)


def torch_filter_nll(y, m0, S0, A, C, Q, u, R, lo=-8.0, hi=8.0):
    """-marginal log-likelihood of ONE keypoint's linear-Gaussian model at s = exp(clip(u, lo, hi))
    (eks/core.py:640-650): y (T,O), R (O,O) constant or (T,O,O); u a 0-d tensor (may require grad)."""
    s = torch.exp(torch.clamp(u, lo, hi))
    m, P = m0, S0
    ll = torch.zeros((), dtype=F64)
    T = y.shape[0]
    for t in range(T):
        Rt = R if R.dim() == 2 else R[t]
        S = C @ P @ C.T + Rt
        pred = C @ m
        ll = ll + torch.distributions.MultivariateNormal(pred, covariance_matrix=S).log_prob(y[t])
        Kg = torch.linalg.solve(S, C @ P).T                  # psd_solve(S, H P)^T without the float32 jitter
        P = P - Kg @ S @ Kg.T
        P = 0.5 * (P + P.T)
        m = m + Kg @ (y[t] - pred)
        m = A @ m
        P = A @ P @ A.T + s * Q
    nll = -ll
    return nll if bool(torch.isfinite(nll)) else torch.full((), 1e12, dtype=F64)


def _loss_and_grad(prob, k, u):
    ut = torch.tensor(float(u), dtype=F64, requires_grad=True)
    L = torch_filter_nll(prob['y'][k], prob['m0'][k], prob['S0'][k], prob['A'][k], prob['C'][k], prob['Q'][k],
                         ut, prob['R'][k])
    g, = torch.autograd.grad(L, ut)
    return float(L.detach()), float(g)


def _singlecam_problem(golden_dir, T):
    g = np.load(os.path.join(golden_dir, 'ibl_pupil_singlecam.npz'))
    arrs = orc.singlecam_arrays(g['markers'][:, :, :T])
    Rd = orc.build_R_from_vars(np.swapaxes(arrs['ensemble_vars'], 0, 1))          # (K,T,O)
    Rc = np.stack([orc.constant_R_from_timevarying(Rd[k]) for k in range(Rd.shape[0])])
    return arrs, Rd, Rc


def _multicam_problem(golden_dir, T):
    from sklearn.decomposition import PCA

    def sk_pca(X, n):
        p = PCA(n_components=n).fit(X)
        return p.components_, p.mean_

    g = np.load(os.path.join(golden_dir, 'mirror_mouse_multicam.npz'))
    arrs = orc.multicam_arrays(g['markers'][:, :, :T], quantile_keep_pca=95.0, n_latent=3, pca_fit=sk_pca)
    Rd = orc.build_R_from_vars(np.swapaxes(arrs['ensemble_vars'], 0, 1))
    Rc = np.stack([orc.constant_R_from_timevarying(Rd[k]) for k in range(Rd.shape[0])])
    return arrs, Rd, Rc


def _as_torch(arrs, R_diag):
    """R_diag (K,O) constant or (K,T,O) time-varying -> dense matrices, like build_R_from_vars upstream."""
    return dict(y=_t(arrs['ys']), m0=_t(arrs['m0s']), S0=_t(arrs['S0s']), A=_t(arrs['As']), C=_t(arrs['Cs']),
                Q=_t(arrs['Qs']), R=torch.diag_embed(_t(R_diag)))


@pytest.mark.parametrize('family', ['singlecam', 'multicam'])
def test_nll_and_reverse_mode_gradient_match_the_oracle(golden_dir, family):
    """MVN log-prob summed over frames + autograd d/d log s  ==  the oracle's NLL and its forward-mode
    sensitivity, on the reference's own recordings (constant R of the loss AND time-varying R of the final
    pass), across the whole range of s including the clipped ends (zero gradient outside the bounds)."""
    T = 200
    arrs, Rd, Rc = (_singlecam_problem if family == 'singlecam' else _multicam_problem)(golden_dir, T)
    K = arrs['ys'].shape[0]
    for R_diag in (Rc, Rd):
        prob = _as_torch(arrs, R_diag)
        for u in (-9.0, -6.5, 0.0, 2.3, 7.9):
            uc = float(np.clip(u, -8.0, 8.0))
            nll_o, g_o = orc.filter_nll(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'],
                                        np.full(K, math.exp(uc)), R_diag, want_grad=True)
            for k in range(K):
                L, g = _loss_and_grad(prob, k, u)
                assert abs(L - nll_o[k]) <= 1e-10 * abs(nll_o[k]), (family, u, k, L, nll_o[k])
                if abs(u) > 8.0:
                    assert g == 0.0                                   # jnp.clip's derivative outside the bounds
                else:
                    assert abs(g - g_o[k]) <= 1e-7 * max(abs(g_o[k]), 1e-3 * abs(nll_o[k])), (family, u, k, g, g_o[k])


def _torch_adam_trajectory(loss_and_grad, u0, lr=0.25, lo=-8.0, hi=8.0, tol=1e-2, cap=300):
    """The reference's loop (eks/core.py:654-681) around torch.optim.Adam: value_and_grad at u, gradient
    scaled by lr, one Adam(1.0) step, stop rule on the loss BEFORE the update; returns (u, last loss, iters,
    the visited u's)."""
    u = torch.tensor(float(u0), dtype=F64, requires_grad=True)
    opt = torch.optim.Adam([u], lr=1.0, betas=(0.9, 0.999), eps=1e-8)
    prev, iters, done, last, path = math.inf, 0, False, math.nan, []
    while not done and iters < cap:
        L, g = loss_and_grad(float(u))
        opt.zero_grad()
        u.grad = torch.tensor(lr * g, dtype=F64)
        opt.step()
        done = math.isfinite(prev) and abs(L - prev) < tol * abs(math.log(max(prev, 1e-12))) + 1e-6
        prev, last, iters = L, L, iters + 1
        path.append(float(u))
    return float(u), last, iters, path


def test_adam_iterates_and_stopping_iteration_match_the_oracle(golden_dir):
    """torch.optim.Adam driven by autograd gradients reproduces oracle.adam_optimize_s (fed by the oracle's
    own forward-mode gradient) on the ibl-pupil keypoints: same number of iterations, same final log s,
    same last loss - optimiser and differentiation both independent of the oracle's."""
    T = 300
    arrs, Rd, Rc = _singlecam_problem(golden_dir, T)
    K = arrs['ys'].shape[0]
    prob = _as_torch(arrs, Rc)
    guesses = [orc.compute_initial_guess(arrs['ensemble_vars'][:, k]) or 2.0 for k in range(K)]
    u0 = np.array([np.float32(np.log(np.clip(g, 1e-6, 1e3))) for g in guesses], dtype=np.float64)

    def oracle_lg(uc):
        return orc.filter_nll(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'],
                              np.exp(uc), Rc, want_grad=True)

    u_o, last_o, it_o = orc.adam_optimize_s(oracle_lg, u0, lr=0.25, tol=1e-2, safety_cap=300)
    for k in range(K):
        u_t, last_t, it_t, path = _torch_adam_trajectory(lambda u: _loss_and_grad(prob, k, u), u0[k])
        assert it_t == it_o[k], (k, it_t, it_o[k])
        assert abs(u_t - u_o[k]) < 1e-8 and abs(last_t - last_o[k]) <= 1e-9 * abs(last_o[k])
        assert abs(path[0] - u0[k]) == pytest.approx(1.0, abs=1e-6)      # Adam's first step is +-1 in log s


def test_adam_on_a_scripted_gradient_sequence_matches_torch():
    """Optimiser alone: both fed the same synthetic loss (a tilted quartic in u with a kink), no filter."""
    def lg(u):
        u = np.asarray(u, dtype=np.float64)
        return 3.0 + (u - 1.7) ** 4 + 0.3 * np.abs(u + 2.0), 4.0 * (u - 1.7) ** 3 + 0.3 * np.sign(u + 2.0)

    for u0, tol, cap in ((-5.0, 1e-2, 300), (6.0, 1e-4, 300), (0.3, 1e-6, 40)):
        u_o, last_o, it_o = orc.adam_optimize_s(lambda uc: lg(uc), np.array([u0]), lr=0.25, tol=tol,
                                                safety_cap=cap)
        u_t, last_t, it_t, _ = _torch_adam_trajectory(lambda u: tuple(map(float, lg(np.clip(u, -8.0, 8.0)))),
                                                      u0, tol=tol, cap=cap)
        assert it_t == it_o[0] and abs(u_t - u_o[0]) < 1e-10 and abs(last_t - last_o[0]) < 1e-10
