"""Host-side statistics around the multicam smoother (reference eks/stats.py).

compute_pca: one small PCA per keypoint on the low-variance frames; stays on the host (it is
set-up for the observation matrices C, not part of the Kalman path).
compute_mahalanobis: factor-analysis reconstruction residuals per view, used by the variance
inflation that precedes the multicam smoother (vectorised over frames; the reference loops)."""
from __future__ import annotations

import numpy as np
from sklearn.decomposition import PCA, FactorAnalysis

from .marker_array import MarkerArray, mA_to_stacked_array


def compute_pca(valid_frames_mask: np.ndarray, emA_centered_preds: MarkerArray,
                emA_good_centered_preds: MarkerArray, n_components: int = 3,
                pca_object: PCA | None = None):
    """Per keypoint: fit PCA on the variance-filtered frames (stacked views (n_good, 2V)),
    transform all frames, keep the components of the frames in `valid_frames_mask`
    (reference eks/stats.py:9-64).  Returns (list of PCA, list of (n_valid_k, n_components))."""
    M, V, T, K, _ = emA_centered_preds.shape
    assert M == 1, 'MarkerArray should have n_models = 1 after ensembling.'
    models, good_pcs = [], []
    for k in range(K):
        all_frames = mA_to_stacked_array(emA_centered_preds, k)
        model = pca_object if pca_object is not None else \
            PCA(n_components=n_components).fit(mA_to_stacked_array(emA_good_centered_preds, k))
        models.append(model)
        good_pcs.append(model.transform(all_frames)[np.flatnonzero(valid_frames_mask[:, k])])
    return models, good_pcs


_PCA_SIGN_RULE = None      # 'v' | 'u' | 'unknown': which entry the installed sklearn makes positive (svd_flip)


def pca_sign_rule() -> str:
    """How the installed scikit-learn fixes the sign of a principal axis: 'v' - the largest-magnitude entry of the
    AXIS itself is positive (svd_flip(u_based_decision=False), scikit-learn >= 1.5) - or 'u' - the largest-magnitude
    SCORE of the fitted rows is positive (older releases).  Decided once by fitting a small matrix on which the two
    rules disagree; 'unknown' (neither reproduces sklearn) sends callers back to sklearn itself."""
    global _PCA_SIGN_RULE
    if _PCA_SIGN_RULE is None:
        from sklearn.decomposition import PCA
        rng = np.random.default_rng(12345)
        X = rng.standard_normal((400, 4)) @ np.diag([5.0, 2.0, 1.0, 0.3]) @ rng.standard_normal((4, 4))
        ref = PCA(n_components=3).fit(X).components_
        Xc = X - X.mean(axis=0)
        rule = 'unknown'
        for cand in ('v', 'u'):
            comp = _pca_axes(Xc.T @ Xc / (len(X) - 1), 3)
            scores = Xc @ comp.T if cand == 'u' else None
            comp = _apply_sign_rule(comp, cand, scores)
            if np.allclose(comp, ref, rtol=0, atol=1e-8):
                rule = cand
                break
        _PCA_SIGN_RULE = rule
    return _PCA_SIGN_RULE


def _pca_axes(cov: np.ndarray, n_components: int) -> np.ndarray:
    """Leading eigenvectors of a covariance matrix as rows, largest eigenvalue first (signs as eigh leaves them)."""
    lam, vec = np.linalg.eigh(np.asarray(cov, dtype=np.float64))
    return np.ascontiguousarray(vec[:, ::-1][:, :n_components].T)


def _apply_sign_rule(comp: np.ndarray, rule: str, scores_extreme: np.ndarray | None) -> np.ndarray:
    """comp (L, F) axes as rows.  'v': flip each axis so that its largest-magnitude entry is positive; 'u': so that
    the largest-magnitude score of the fitted rows is - `scores_extreme` is either the (n, L) score matrix or the
    (L,) signed extreme scores themselves."""
    if rule == 'v':
        idx = np.argmax(np.abs(comp), axis=1)
        sign = np.sign(comp[np.arange(comp.shape[0]), idx])
    else:
        ext = np.asarray(scores_extreme)
        if ext.ndim == 2:
            ext = ext[np.argmax(np.abs(ext), axis=0), np.arange(ext.shape[1])]
        sign = np.sign(ext)
    sign = np.where(sign == 0, 1.0, sign)
    return comp * sign[:, None]


def pca_from_moments(cov: np.ndarray, n_components: int, extreme_scores=None) -> np.ndarray:
    """sklearn.decomposition.PCA(n_components).fit(X).components_ from the covariance of X alone (X_c' X_c / (n - 1)):
    what scikit-learn's own `covariance_eigh` solver computes for tall matrices, and the right singular vectors of
    X_c otherwise.  The sign of every axis follows the installed scikit-learn (`pca_sign_rule`); the 'u' rule needs
    the signed largest-magnitude score of the fitted rows along each UNSIGNED axis: `extreme_scores(axes) -> (L,)`,
    a callable so that the rows can stay where they are (the multi-camera driver keeps them on the device)."""
    rule = pca_sign_rule()
    if rule == 'unknown':
        raise RuntimeError('pca_from_moments: the installed scikit-learn follows neither sign rule')
    comp = _pca_axes(cov, n_components)
    ext = extreme_scores(comp) if rule == 'u' else None
    return _apply_sign_rule(comp, rule, ext)


def factor_analysis_from_moments(cov: np.ndarray, n_samples: int, n_components: int, tol: float = 1e-2,
                                 max_iter: int = 1000) -> tuple:
    """sklearn.decomposition.FactorAnalysis(n_components).fit(X) from X's second moments alone.

    The estimator's EM loop (default arguments: tol 1e-2, max_iter 1000, unit initial noise) only ever
    looks at X through the singular values and right singular vectors of X_c / (sqrt(psi) sqrt(n)),
    X_c = X - mean: the eigen-decomposition of D cov D, D = diag(1 / sqrt(psi)), cov = X_c' X_c / n
    (p x p, p = 2 x views).  Its default randomized SVD draws n_components + 10 directions, which
    spans all p columns whenever p <= n_components + 10 - it is an exact SVD there, which is the
    condition the caller checks before using this routine; the loop below is then the same
    sequence of iterates up to rounding.  The (T, p) matrix never has to leave the device: the
    variance-inflation driver reduces it to `cov` there and runs this p x p loop on the host
    (C4 shape, 4 keypoints x 25 000 fitted rows: 2.8 s of sklearn fits per call -> milliseconds).

    Returns (W (p, n_components), psi (p,), n_iter).  The sign of W's columns is the eigenvector
    routine's; everything the variance inflation computes from W (reconstruction, posterior
    predictive variance, Mahalanobis distance: eks/stats.py:119-151) depends on its column space only."""
    cov = np.asarray(cov, dtype=np.float64)
    p = cov.shape[0]
    var = np.diag(cov).copy()
    llconst = p * np.log(2.0 * np.pi) + n_components
    psi = np.ones(p)
    old_ll = -np.inf
    small = 1e-12
    W = np.zeros((n_components, p))
    it = 0
    for it in range(max_iter):
        sqrt_psi = np.sqrt(psi) + small
        scaled = cov / np.outer(sqrt_psi, sqrt_psi)
        lam, vec = np.linalg.eigh(scaled)                       # ascending
        lam, vec = lam[::-1][:n_components], vec[:, ::-1][:, :n_components]
        lam = np.maximum(lam, 0.0)
        unexp_var = np.trace(scaled) - lam.sum()
        W = (np.sqrt(np.maximum(lam - 1.0, 0.0))[:, None] * vec.T) * sqrt_psi
        with np.errstate(divide='ignore'):
            ll = llconst + np.sum(np.log(lam))
        ll += unexp_var + np.sum(np.log(psi))
        ll *= -n_samples / 2.0
        if (ll - old_ll) < tol:
            break
        old_ll = ll
        psi = np.maximum(var - np.sum(W ** 2, axis=0), small)
    else:
        import warnings

        from sklearn.exceptions import ConvergenceWarning
        warnings.warn('FactorAnalysis did not converge. You might want to increase the number of iterations.',
                      ConvergenceWarning)
    return W.T, psi, it + 1


def compute_mahalanobis(x: np.ndarray, v: np.ndarray, n_latent: int = 3,
                        v_quantile_threshold: float | None = 50.0,
                        likelihoods: np.ndarray | None = None,
                        likelihood_threshold: float | None = 0.9, epsilon: float | None = 1e-6,
                        loading_matrix: np.ndarray | None = None,
                        mean: np.ndarray | None = None) -> dict:
    """Mahalanobis distance of each view's 2-D residual under a linear latent-variable model
    x = W z + mu + noise(v) (reference eks/stats.py:67-157).

    W, mu come from sklearn FactorAnalysis fitted on the rows whose largest variance is below the
    `v_quantile_threshold` percentile (and whose smallest likelihood is >= `likelihood_threshold`
    when likelihoods are given), unless `loading_matrix` / `mean` are supplied.  Returns the same
    dict as the reference: 'mahalanobis' {view: (N,1)}, 'posterior_variance' {view: (N,2,2)},
    'reconstructed' (N,2C)."""
    x = np.asarray(x)
    v = np.asarray(v)
    if loading_matrix is None or mean is None:
        rows = np.ones(x.shape[0], dtype=bool)
        if likelihoods is not None and likelihood_threshold is not None:
            rows &= np.min(likelihoods, axis=1) >= likelihood_threshold
        if v_quantile_threshold is not None:
            worst = v.max(axis=1)
            rows &= worst < np.percentile(worst, v_quantile_threshold)
        fa = FactorAnalysis(n_components=n_latent).fit(x[rows])
        W, mu = fa.components_.T, fa.mean_
    else:
        W, mu = np.asarray(loading_matrix), np.asarray(mean)
    prec = 1.0 / (v + epsilon)                                        # (N, 2C)
    WtP = W.T[None, :, :] * prec[:, None, :]                          # (N, L, 2C) = W' diag(prec)
    B = np.linalg.inv(WtP @ W)                                        # (N, L, L)
    z = (B @ (WtP @ (x - mu)[:, :, None]))[:, :, 0]                   # (N, L)
    xhat = z @ W.T + mu
    diff = x - xhat
    n_views = x.shape[1] // 2
    Q, M = {}, {}
    for c in range(n_views):
        Wc = W[2 * c:2 * c + 2]                                       # (2, L)
        Qc = Wc[None] @ B @ Wc.T[None]                                # (N, 2, 2)
        Qc[:, 0, 0] += v[:, 2 * c]
        Qc[:, 1, 1] += v[:, 2 * c + 1]
        dc = diff[:, 2 * c:2 * c + 2]
        M[c] = np.einsum('ni,nij,nj->n', dc, np.linalg.inv(Qc), dc)[:, None]
        Q[c] = Qc
    return {'mahalanobis': M, 'posterior_variance': Q, 'reconstructed': xhat}
