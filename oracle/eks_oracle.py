"""CPU oracle (float64 NumPy) for the eks Kalman hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement of the algorithm the reference runs on its hot path
(`run_kalman_smoother`, /root/reference/eks/core.py:159-302) and of the host-side stages either
side of it.  It exists to check the HIP kernels; nothing under ``eks_amd/`` may import it.  Only
``tests/`` (and the parity-sweep / check scripts under ``tools/`` that ``tests/test_gpu_fuzz.py``
drives), ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it - always as
the checker, never as the thing measured or shipped.

PARITY UNPINNED: the reference's arithmetic lives in third-party ``dynamax`` (pyproject.toml:39-46
pins ``dynamax<=1.0.1``; jax/jaxlib/optax unpinned), which together with ``jax`` is absent from this
image and from /root/reference, and the reference's only numeric pins (golden CSVs,
tests/conftest.py:12) are downloaded at test time from a URL that is unreachable here.  The oracle
is therefore pinned by (1) closed-form known answers, (2) three independent formulations agreeing
(covariance form, information form, associative-scan form), (3) finite-difference gradients and
(4) the properties the reference's own unit tests assert.  See tests/test_oracle_*.py.

What each function follows (file:line relative to /root/reference):

* ``kalman_filter`` / ``kalman_smoother``  - dynamax ``extended_kalman_filter`` /
  ``extended_kalman_smoother`` with linear f,h as called at eks/core.py:290, :469, :648
  (update-then-predict filter, RTS backward pass).
* ``constant_R_from_timevarying``          - eks/core.py:702-709
* ``compute_initial_guess``                - eks/core.py:104-133 and its call site :233-236
* ``crop_frames``                          - eks/utils.py:235-290
* ``adam_optimize_s``                      - eks/core.py:562-699 (singletons) and :403-559 (blocks)
* ``run_kalman_smoother``                  - eks/core.py:159-302
* ``ensemble``                             - eks/core.py:25-101
* ``center_predictions``                   - eks/utils.py:293-365
* ``singlecam_arrays`` / ``multicam_arrays`` - eks/singlecam_smoother.py:140-243, :246-284 and
  eks/multicam_smoother.py:335-348, :409-443, :481-551, :554-597
* ``pupil_*`` / ``run_pupil_kalman_smoother`` - eks/ibl_pupil_smoother.py:34-91, :233-359, :363-607
"""
from __future__ import annotations

import math

import numpy as np

LOG2PI = math.log(2.0 * math.pi)


# --------------------------------------------------------------------------------------------
# small batched helpers (leading axis = keypoints)
# --------------------------------------------------------------------------------------------
def _sym(P):
    return 0.5 * (P + np.swapaxes(P, -1, -2))


def _bmm(*mats):
    out = mats[0]
    for m in mats[1:]:
        out = np.matmul(out, m)
    return out


def _T(M):
    return np.swapaxes(M, -1, -2)


def _as_R_getter(R, K, T, O):
    """R may be (K,T,O) time-varying diagonal, (K,O) constant diagonal, or full matrices
    (K,T,O,O) / (K,O,O).  Returns f(t) -> (K,O,O)."""
    R = np.asarray(R, dtype=np.float64)
    eye = np.eye(O)
    if R.shape == (K, T, O):
        return lambda t: R[:, t, :, None] * eye
    if R.shape == (K, O):
        Rc = R[:, :, None] * eye
        return lambda t: Rc
    if R.shape == (K, T, O, O):
        return lambda t: R[:, t]
    if R.shape == (K, O, O):
        return lambda t: R
    raise ValueError(f'bad R shape {R.shape} for K={K},T={T},O={O}')


# --------------------------------------------------------------------------------------------
# forward filter (covariance form; the dynamax recursion)
# --------------------------------------------------------------------------------------------
def kalman_filter(y, m0, S0, A, C, Q, s, R, *, jitter=0.0, symmetrize=True, want_grad=False,
                  tangent=None):
    """Update-then-predict Kalman filter, batched over keypoints.

    y (K,T,O); m0 (K,D); S0,A,Q (K,D,D); C (K,O,D); s (K,) process-noise scale (cov = s*Q);
    R see ``_as_R_getter``.  ``jitter`` reproduces dynamax ``psd_solve``'s diagonal boost (1e-9
    upstream) applied ONLY inside the gain solve; ``symmetrize`` its post-update symmetrisation.

    Returns dict with filtered means/covs (K,T,D)/(K,T,D,D), predicted ones, ``ll`` (K,) marginal
    log-likelihood, and if ``want_grad`` also ``dll`` = d ll / d log s (forward sensitivity).
    ``tangent=(dA, dQ)`` (each (K,D,D)) instead returns ``dll`` = the directional derivative of
    ll for the perturbation A + h dA, s*Q + h dQ (used by the pupil loss, whose two parameters
    enter both A and Q: eks/ibl_pupil_smoother.py:540-552).
    """
    y = np.asarray(y, np.float64)
    K, T, O = y.shape
    m0 = np.asarray(m0, np.float64)
    D = m0.shape[-1]
    S0 = np.asarray(S0, np.float64)
    A = np.asarray(A, np.float64)
    C = np.asarray(C, np.float64)
    Q = np.asarray(Q, np.float64)
    s = np.broadcast_to(np.asarray(s, np.float64), (K,))
    Rt = _as_R_getter(R, K, T, O)
    sQ = s[:, None, None] * Q
    eyeO = np.eye(O)

    mf = np.empty((K, T, D))
    Pf = np.empty((K, T, D, D))
    mp = np.empty((K, T, D))
    Pp = np.empty((K, T, D, D))
    ll = np.zeros(K)
    m = m0.copy()
    P = S0.copy()
    if tangent is not None:
        want_grad = True
        dA = np.asarray(tangent[0], np.float64)
        dsQ = np.asarray(tangent[1], np.float64)
    else:
        dA = None
        dsQ = sQ                                # d(sQ)/d log s = sQ
    if want_grad:
        dm = np.zeros_like(m)
        dP = np.zeros_like(P)
        dll = np.zeros(K)
    At = _T(A)
    Ct = _T(C)
    for t in range(T):
        mp[:, t] = m
        Pp[:, t] = P
        R_ = Rt(t)
        S = _bmm(C, P, Ct) + R_
        e = y[:, t] - np.einsum('kod,kd->ko', C, m)
        Sinv = np.linalg.inv(S)
        _, logdet = np.linalg.slogdet(S)
        quad = np.einsum('ko,kop,kp->k', e, Sinv, e)
        ll += -0.5 * (O * LOG2PI + logdet + quad)
        Sj_inv = np.linalg.inv(_sym(S) + jitter * eyeO) if jitter else Sinv
        PCt = np.matmul(P, Ct)
        Kg = np.matmul(PCt, Sj_inv)                      # (K,D,O)
        if want_grad:
            dS = _bmm(C, dP, Ct)
            de = -np.einsum('kod,kd->ko', C, dm)
            dll += -0.5 * (np.einsum('kop,kpo->k', Sinv, dS)
                           + 2.0 * np.einsum('ko,kop,kp->k', e, Sinv, de)
                           - np.einsum('ko,kop,kpq,kqr,kr->k', e, Sinv, dS, Sinv, e))
            dKg = np.matmul(np.matmul(dP, Ct), Sj_inv) - _bmm(Kg, dS, Sj_inv)
            dm = dm + np.einsum('kdo,ko->kd', dKg, e) + np.einsum('kdo,ko->kd', Kg, de)
            dP = dP - _bmm(dKg, S, _T(Kg)) - _bmm(Kg, dS, _T(Kg)) - _bmm(Kg, S, _T(dKg))
        m = m + np.einsum('kdo,ko->kd', Kg, e)
        P = P - _bmm(Kg, S, _T(Kg))
        if symmetrize:
            P = _sym(P)
            if want_grad:
                dP = _sym(dP)
        mf[:, t] = m
        Pf[:, t] = P
        if want_grad:
            dm = np.einsum('kde,ke->kd', A, dm)
            dP = _bmm(A, dP, At) + dsQ
            if dA is not None:
                dm = dm + np.einsum('kde,ke->kd', dA, m)
                APdAt = _bmm(A, P, _T(dA))
                dP = dP + APdAt + _T(APdAt)
        m = np.einsum('kde,ke->kd', A, m)
        P = _bmm(A, P, At) + sQ
    out = dict(mf=mf, Pf=Pf, mp=mp, Pp=Pp, ll=ll, m_next=m, P_next=P)
    if want_grad:
        out['dll'] = dll
    return out


def rts_smoother(mf, Pf, A, Q, s, *, jitter=0.0):
    """RTS backward pass (dynamax ``extended_kalman_smoother``'s second scan)."""
    K, T, D = mf.shape
    s = np.broadcast_to(np.asarray(s, np.float64), (K,))
    sQ = s[:, None, None] * np.asarray(Q, np.float64)
    A = np.asarray(A, np.float64)
    At = _T(A)
    ms = np.empty_like(mf)
    Vs = np.empty_like(Pf)
    ms[:, -1] = mf[:, -1]
    Vs[:, -1] = Pf[:, -1]
    eye = np.eye(D)
    for t in range(T - 2, -1, -1):
        m_pred = np.einsum('kde,ke->kd', A, mf[:, t])
        S_pred = sQ + _bmm(A, Pf[:, t], At)
        G = np.matmul(np.matmul(Pf[:, t], At), np.linalg.inv(_sym(S_pred) + jitter * eye))
        ms[:, t] = mf[:, t] + np.einsum('kde,ke->kd', G, ms[:, t + 1] - m_pred)
        Vs[:, t] = Pf[:, t] + _bmm(G, Vs[:, t + 1] - S_pred, _T(G))
    return ms, Vs


def kalman_smoother(y, m0, S0, A, C, Q, s, R, *, jitter=0.0, symmetrize=True):
    f = kalman_filter(y, m0, S0, A, C, Q, s, R, jitter=jitter, symmetrize=symmetrize)
    ms, Vs = rts_smoother(f['mf'], f['Pf'], A, Q, s, jitter=jitter)
    return ms, Vs, -f['ll']


def filter_nll(y, m0, S0, A, C, Q, s, R, *, jitter=0.0, symmetrize=True, want_grad=False):
    """NLL (K,) of eks/core.py:640-650: -marginal_loglik, non-finite -> 1e12."""
    f = kalman_filter(y, m0, S0, A, C, Q, s, R, jitter=jitter, symmetrize=symmetrize,
                      want_grad=want_grad)
    nll = -f['ll']
    bad = ~np.isfinite(nll)
    nll = np.where(bad, 1e12, nll)
    if want_grad:
        g = np.where(bad, 0.0, -f['dll'])
        return nll, g
    return nll


# --------------------------------------------------------------------------------------------
# float32 EMULATION of the upstream recursion: what the reference's own numbers
# look like.  The reference runs dynamax in float32 (jax's default; eks/ never enables x64: SURVEY.md
# D5 / A.4), in the covariance form, with dynamax's operation order (recalled from
# dynamax <= 1.0.1 `_condition_on` / `_predict` / `extended_kalman_smoother`):
#     S = R + H P H^T ; ll += MVN(h(m), S).log_prob(y)   (tfp: Cholesky of S, triangular solve)
#     K = psd_solve(S, H P)^T                            (Cholesky solve of sym(S) + 1e-9 I)
#     P <- P - K S K^T ; m <- m + K (y - h(m)) ; P <- (P + P^T) / 2
#     m <- A m ; P <- A P A^T + s Q
#     backward: G = psd_solve(s Q + A P_f A^T, A P_f)^T ; m_s = m_f + G (m_s+ - A m_f) ;
#               P_s = P_f + G (P_s+ - S_pred) G^T
# Every array and every intermediate is float32 here (NumPy keeps float32 through matmul / Cholesky;
# the triangular solves are written out so that nothing is silently promoted).  XLA's fusion and
# FMA choices differ from NumPy's in the last bit of each operation, so this is an emulation of the
# MAGNITUDE of upstream's rounding - the forecast of tests/test_f32_forecast.py - not a bitwise twin.
# --------------------------------------------------------------------------------------------
def _chol_solve_f32(S, B, jitter):
    """psd_solve(S, B): X with (sym(S) + jitter I) X = B by Cholesky, float32 throughout.
    S (K,n,n), B (K,n,m) -> (K,n,m)."""
    f = np.float32
    n = S.shape[-1]
    Sj = (f(0.5) * (S + np.swapaxes(S, -1, -2)) + f(jitter) * np.eye(n, dtype=f)).astype(f)
    L = np.linalg.cholesky(Sj).astype(f)
    Z = np.empty_like(B)
    for i in range(n):                                   # L Z = B
        acc = B[:, i, :].copy()
        for j in range(i):
            acc = acc - L[:, i, j, None] * Z[:, j, :]
        Z[:, i, :] = acc / L[:, i, i, None]
    X = np.empty_like(B)
    for i in range(n - 1, -1, -1):                       # L^T X = Z
        acc = Z[:, i, :].copy()
        for j in range(i + 1, n):
            acc = acc - L[:, j, i, None] * X[:, j, :]
        X[:, i, :] = acc / L[:, i, i, None]
    return X


def _mvn_logpdf_f32(e, S):
    """tfp MultivariateNormalFullCovariance(loc, S).log_prob(y) with e = y - loc, float32: scale_tril =
    cholesky(S), -0.5 |L^-1 e|^2 - sum log diag L - 0.5 n log 2 pi."""
    f = np.float32
    n = S.shape[-1]
    L = np.linalg.cholesky(S).astype(f)
    z = np.empty_like(e)
    for i in range(n):
        acc = e[:, i].copy()
        for j in range(i):
            acc = acc - L[:, i, j] * z[:, j]
        z[:, i] = acc / L[:, i, i]
    logdet_half = np.log(np.diagonal(L, axis1=-2, axis2=-1)).astype(f).sum(axis=-1, dtype=f)
    return (f(-0.5) * (z * z).sum(axis=-1, dtype=f) - logdet_half - f(0.5 * n * LOG2PI)).astype(f)


def kalman_smoother_f32(y, m0, S0, A, C, Q, s, R, *, jitter=1e-9, symmetrize=True, emission=None):
    """The upstream recursion in float32 (see the block comment above).  Arguments as kalman_smoother (they are
    rounded to float32 on entry, as `jnp.asarray` does upstream).  `emission(m (K,D) float32) -> (h(m) (K,O),
    dh/dx (K,O,D))` in float32 replaces the linear C (the extended filter of eks/core.py:188-190 linearised at the
    predicted mean; C is then ignored).  Returns (ms, Vs, nll) as float32 arrays."""
    f = np.float32
    y = np.asarray(y, f)
    K, T, O = y.shape
    m0 = np.asarray(m0, f)
    D = m0.shape[-1]
    A = np.asarray(A, f)
    C = None if emission is not None else np.asarray(C, f)
    At = _T(A)
    sQ = (np.broadcast_to(np.asarray(s, f), (K,))[:, None, None] * np.asarray(Q, f)).astype(f)   # eks/core.py:152
    R = np.asarray(R, f)
    eyeO = np.eye(O, dtype=f)
    if R.shape == (K, T, O):
        Rt = lambda t: R[:, t, :, None] * eyeO
    elif R.shape == (K, O):
        Rc = R[:, :, None] * eyeO
        Rt = lambda t: Rc
    else:
        raise ValueError(f'bad R shape {R.shape}')
    mf = np.empty((K, T, D), f)
    Pf = np.empty((K, T, D, D), f)
    ll = np.zeros(K, f)
    m, P = m0.copy(), np.asarray(S0, f).copy()
    for t in range(T):
        if emission is None:
            H, yhat = C, np.einsum('kod,kd->ko', C, m)
        else:
            yhat, H = emission(m)
            assert yhat.dtype == f and H.dtype == f
        HP = np.matmul(H, P)                                        # (K,O,D)
        S = (Rt(t) + np.matmul(HP, _T(H))).astype(f)
        e = y[:, t] - yhat
        ll = (ll + _mvn_logpdf_f32(e, S)).astype(f)
        Kg = _T(_chol_solve_f32(S, HP, jitter))                     # (K,D,O)
        P = P - _bmm(Kg, S, _T(Kg))
        m = m + np.einsum('kdo,ko->kd', Kg, e)
        if symmetrize:
            P = f(0.5) * (P + _T(P))
        mf[:, t], Pf[:, t] = m, P
        m = np.einsum('kde,ke->kd', A, m)
        P = _bmm(A, P, At) + sQ
    ms, Vs = np.empty_like(mf), np.empty_like(Pf)
    ms[:, -1], Vs[:, -1] = mf[:, -1], Pf[:, -1]
    for t in range(T - 2, -1, -1):
        m_pred = np.einsum('kde,ke->kd', A, mf[:, t])
        AP = np.matmul(A, Pf[:, t])
        S_pred = sQ + np.matmul(AP, At)
        G = _T(_chol_solve_f32(S_pred, AP, jitter))
        ms[:, t] = mf[:, t] + np.einsum('kde,ke->kd', G, ms[:, t + 1] - m_pred)
        Vs[:, t] = Pf[:, t] + _bmm(G, Vs[:, t + 1] - S_pred, _T(G))
    assert ms.dtype == f and Vs.dtype == f and ll.dtype == f
    return ms, Vs, -ll


# --------------------------------------------------------------------------------------------
# independent formulation #2: information-form filter + two-filter smoother (diagonal R only)
# --------------------------------------------------------------------------------------------
def info_form_smoother(y, m0, S0, A, C, Q, s, Rdiag):
    """Same posterior by a different route: information-form measurement update
    P = (P^-1 + C' R^-1 C)^-1 and RTS written with explicit solves.  Used only to cross-check
    ``kalman_smoother``."""
    y = np.asarray(y, np.float64)
    K, T, O = y.shape
    D = np.asarray(m0).shape[-1]
    Rt = np.broadcast_to(np.asarray(Rdiag, np.float64)[:, None, :] if np.ndim(Rdiag) == 2
                         else np.asarray(Rdiag, np.float64), (K, T, O))
    s = np.broadcast_to(np.asarray(s, np.float64), (K,))
    sQ = s[:, None, None] * np.asarray(Q, np.float64)
    A = np.asarray(A, np.float64)
    C = np.asarray(C, np.float64)
    m = np.asarray(m0, np.float64).copy()
    P = np.asarray(S0, np.float64).copy()
    mf = np.empty((K, T, D))
    Pf = np.empty((K, T, D, D))
    ll = np.zeros(K)
    for t in range(T):
        W = 1.0 / Rt[:, t]                                     # (K,O)
        J = np.einsum('kod,ko,koe->kde', C, W, C)
        eta = np.einsum('kod,ko,ko->kd', C, W, y[:, t])
        Pinv = np.linalg.inv(P)
        Pn = np.linalg.inv(Pinv + J)
        mn = np.einsum('kde,ke->kd', Pn, np.einsum('kde,ke->kd', Pinv, m) + eta)
        # log-lik through the matrix determinant lemma |S| = |R| |P| / |Pn|
        e = y[:, t] - np.einsum('kod,kd->ko', C, m)
        _, ldP = np.linalg.slogdet(P)
        _, ldPn = np.linalg.slogdet(Pn)
        logdetS = np.sum(np.log(Rt[:, t]), axis=1) + ldP - ldPn
        # e' S^-1 e = e' W e - (C'We)' Pn (C'We)
        cwe = np.einsum('kod,ko,ko->kd', C, W, e)
        quad = np.einsum('ko,ko,ko->k', e, W, e) - np.einsum('kd,kde,ke->k', cwe, Pn, cwe)
        ll += -0.5 * (O * LOG2PI + logdetS + quad)
        mf[:, t] = mn
        Pf[:, t] = Pn
        m = np.einsum('kde,ke->kd', A, mn)
        P = _bmm(A, Pn, _T(A)) + sQ
    ms, Vs = rts_smoother(mf, Pf, A, Q, s)
    return ms, Vs, -ll


# --------------------------------------------------------------------------------------------
# independent formulation #3: associative-scan elements (what the HIP kernels compose)
# --------------------------------------------------------------------------------------------
def assoc_elements(y, A, C, Q, s, Rdiag):
    """Per-step filtering elements (A_e, b, C_e, eta, J, ell) for the update-then-predict
    ordering: element t maps a belief over x_t (before y_t) to a belief over x_{t+1}."""
    y = np.asarray(y, np.float64)
    K, T, O = y.shape
    D = np.asarray(A).shape[-1]
    W = 1.0 / np.broadcast_to(np.asarray(Rdiag, np.float64), (K, T, O))
    Cm = np.asarray(C, np.float64)
    J = np.einsum('kod,kto,koe->ktde', Cm, W, Cm)
    eta = np.einsum('kod,kto,kto->ktd', Cm, W, y)
    s = np.broadcast_to(np.asarray(s, np.float64), (K,))
    sQ = s[:, None, None] * np.asarray(Q, np.float64)
    Ae = np.broadcast_to(np.asarray(A, np.float64)[:, None], (K, T, D, D)).copy()
    b = np.zeros((K, T, D))
    Ce = np.broadcast_to(sQ[:, None], (K, T, D, D)).copy()
    # ell: log of the x-independent factor of p(y_t | x_t) = N(y; Cx, R)
    ell = -0.5 * (O * LOG2PI - np.sum(np.log(W), axis=2) + np.einsum('kto,kto,kto->kt', y, W, y))
    return Ae, b, Ce, eta, J, ell


def assoc_combine(ei, ej):
    """Compose element i (earlier) with element j (later).  Batched over leading axes."""
    Ai, bi, Ci, etai, Ji, li = ei
    Aj, bj, Cj, etaj, Jj, lj = ej
    D = Ai.shape[-1]
    eye = np.eye(D)
    M = np.linalg.inv(eye + np.matmul(Ci, Jj))             # (I + C_i J_j)^-1
    AjM = np.matmul(Aj, M)
    A = np.matmul(AjM, Ai)
    b = np.einsum('...de,...e->...d', AjM, bi + np.einsum('...de,...e->...d', Ci, etaj)) + bj
    Cc = _bmm(AjM, Ci, _T(Aj)) + Cj
    Mt = _T(M)                                             # (I + J_j C_i)^-1
    AitMt = np.matmul(_T(Ai), Mt)
    eta = np.einsum('...de,...e->...d', AitMt,
                    etaj - np.einsum('...de,...e->...d', Jj, bi)) + etai
    J = _bmm(AitMt, Jj, Ai) + Ji
    # log-normaliser of  int N(x; mu, Ci) exp(eta_j'x - x'J_j x/2) dx  at mu = b_i (x_in = 0 part)
    _, ld = np.linalg.slogdet(eye + np.matmul(Ci, Jj))
    v = etaj - np.einsum('...de,...e->...d', Jj, bi)
    l = li + lj - 0.5 * ld + np.einsum('...d,...d->...', bi, etaj) \
        - 0.5 * np.einsum('...d,...de,...e->...', bi, Jj, bi) \
        + 0.5 * np.einsum('...d,...de,...e->...', v, np.matmul(M, Ci), v)
    return A, b, Cc, eta, J, l


def assoc_apply(elem, m, P):
    """Push a Gaussian belief N(m,P) on x_in through a (composite) element: returns the belief on
    x_out and the log marginal likelihood of the element's observations under that prior."""
    A, b, Cc, eta, J, l = elem
    D = A.shape[-1]
    eye = np.eye(D)
    N = np.linalg.inv(eye + np.matmul(P, J))               # (I + P J)^-1
    m_in = np.einsum('...de,...e->...d', N, m + np.einsum('...de,...e->...d', P, eta))
    P_in = np.matmul(N, P)
    m_out = np.einsum('...de,...e->...d', A, m_in) + b
    P_out = _bmm(A, P_in, _T(A)) + Cc
    _, ld = np.linalg.slogdet(eye + np.matmul(P, J))
    v = eta - np.einsum('...de,...e->...d', J, m)
    ll = l - 0.5 * ld + np.einsum('...d,...d->...', m, eta) \
        - 0.5 * np.einsum('...d,...de,...e->...', m, J, m) \
        + 0.5 * np.einsum('...d,...de,...e->...', v, P_in, v)
    return m_out, P_out, ll


def assoc_chunked_smoother(y, m0, S0, A, C, Q, s, Rdiag, chunk):
    """The three-phase chunked scan the HIP kernels implement, restated in float64:
    (1) per-chunk composite element, (2) prefix scan -> incoming prior per chunk and suffix scan
    -> information (eta,J) about the state at the start of the NEXT chunk from all later data,
    (3) per chunk: exact filter replay from the incoming prior, combine the outgoing predicted
    belief with the suffix information, RTS backward inside the chunk."""
    y = np.asarray(y, np.float64)
    K, T, O = y.shape
    D = np.asarray(m0).shape[-1]
    Rd = np.broadcast_to(np.asarray(Rdiag, np.float64)[:, None, :] if np.ndim(Rdiag) == 2
                         else np.asarray(Rdiag, np.float64), (K, T, O))
    el = assoc_elements(y, A, C, Q, s, Rd)
    bounds = list(range(0, T, chunk)) + [T]
    nchunk = len(bounds) - 1
    summ = []
    for c in range(nchunk):
        lo, hi = bounds[c], bounds[c + 1]
        acc = tuple(e[:, lo] for e in el)
        for t in range(lo + 1, hi):
            acc = assoc_combine(acc, tuple(e[:, t] for e in el))
        summ.append(acc)
    # prefix: incoming prior of each chunk + total log-lik
    m_in = [np.asarray(m0, np.float64)]
    P_in = [np.asarray(S0, np.float64)]
    ll = np.zeros(K)
    for c in range(nchunk):
        mo, Po, l = assoc_apply(summ[c], m_in[c], P_in[c])
        ll += l
        m_in.append(mo)
        P_in.append(Po)
    # suffix: (eta,J) about x at the start of chunk c+1 from chunks c+1..end
    suf = [None] * nchunk
    eta = np.zeros((K, D))
    J = np.zeros((K, D, D))
    for c in range(nchunk - 1, -1, -1):
        suf[c] = (eta, J)
        Ac, bc, Cc, etac, Jc, _ = summ[c]
        # information about x_in(c) = elem_c's own (eta,J) + what flows back through it
        Mt = np.linalg.inv(np.eye(D) + np.matmul(J, Cc))
        AtMt = np.matmul(_T(Ac), Mt)
        eta_new = np.einsum('kde,ke->kd', AtMt, eta - np.einsum('kde,ke->kd', J, bc)) + etac
        J_new = _bmm(AtMt, J, Ac) + Jc
        eta, J = eta_new, J_new
    ms = np.empty((K, T, D))
    Vs = np.empty((K, T, D, D))
    s_arr = np.broadcast_to(np.asarray(s, np.float64), (K,))
    for c in range(nchunk):
        lo, hi = bounds[c], bounds[c + 1]
        f = kalman_filter(y[:, lo:hi], m_in[c], P_in[c], A, C, Q, s_arr, Rd[:, lo:hi],
                          symmetrize=False)
        eta_s, J_s = suf[c]
        mN, PN = f['m_next'], f['P_next']
        N = np.linalg.inv(np.eye(D) + np.matmul(PN, J_s))
        m_s = np.einsum('kde,ke->kd', N, mN + np.einsum('kde,ke->kd', PN, eta_s))
        P_s = np.matmul(N, PN)
        sQ = s_arr[:, None, None] * np.asarray(Q, np.float64)
        Am = np.asarray(A, np.float64)
        for t in range(hi - 1, lo - 1, -1):
            i = t - lo
            S_pred = sQ + _bmm(Am, f['Pf'][:, i], _T(Am))
            G = np.matmul(np.matmul(f['Pf'][:, i], _T(Am)), np.linalg.inv(S_pred))
            m_s = f['mf'][:, i] + np.einsum('kde,ke->kd', G,
                                            m_s - np.einsum('kde,ke->kd', Am, f['mf'][:, i]))
            P_s = f['Pf'][:, i] + _bmm(G, P_s - S_pred, _T(G))
            ms[:, t] = m_s
            Vs[:, t] = P_s
    return ms, Vs, -ll


# --------------------------------------------------------------------------------------------
# pieces of eks/core.py around the filter
# --------------------------------------------------------------------------------------------
def build_R_from_vars(ev):
    """eks/utils.py:368-377 restated for diagonals only: clip(var, 1e-12, inf)."""
    return np.clip(np.asarray(ev, np.float64), 1e-12, None)


def constant_R_from_timevarying(Rdiag_t, min_var=1e-4):
    """eks/core.py:702-709: median over time of diag R_t, floored.  Rdiag_t (..., T, O)."""
    med = np.nanmedian(np.asarray(Rdiag_t, np.float64), axis=-2)
    return np.clip(med, min_var, np.inf)


def compute_initial_guess(ensemble_vars_k):
    """eks/core.py:104-133 with the call-site post-processing of :233-236.
    ensemble_vars_k (T,O) for one keypoint."""
    ev = np.asarray(ensemble_vars_k)[:2000]
    if ev.shape[0] < 2:
        raise ValueError('Not enough frames to compute temporal differences.')
    d = ev[1:] - ev[:-1]
    g = float(round(float(np.nanstd(d)), 5)) or 2.0
    return g if (np.isfinite(g) and g > 0.0) else 2.0


def crop_frames(y, s_frames):
    """eks/utils.py:235-290 (0-based half-open spans, ascending non-overlapping)."""
    n = len(y)
    if s_frames is None or len(s_frames) == 0 or \
            (len(s_frames) == 1 and tuple(s_frames[0]) == (None, None)):
        return y
    if not isinstance(s_frames, list):
        raise TypeError('s_frames must be a list of (start, end) tuples or None.')
    spans = []
    for i, fr in enumerate(s_frames):
        if not (isinstance(fr, tuple) and len(fr) == 2):
            raise ValueError(f's_frames[{i}] must be a (start, end) tuple, got {fr!r}')
        a, b = fr
        for nm, v in (('start', a), ('end', b)):
            if v is not None and not isinstance(v, int):
                raise ValueError(f's_frames[{i}].{nm} must be int or None, got {v!r}')
        a = 0 if a is None else a
        b = n if b is None else b
        if a < 0 or b > n:
            raise ValueError(f'Range ({a}, {b}) out of bounds for length {n}.')
        if a >= b:
            raise ValueError(f'Invalid range ({a}, {b}).')
        spans.append((a, b))
    spans.sort(key=lambda ab: ab[0])
    for i in range(1, len(spans)):
        if spans[i][0] < spans[i - 1][1]:
            raise ValueError(f'Overlapping or out-of-order intervals: {spans[i-1]} and {spans[i]}')
    return np.concatenate([y[a:b] for a, b in spans], axis=0)


def adam_optimize_s(loss_and_grad, u0, *, lr=0.25, s_bounds_log=(-8.0, 8.0), tol=1e-2,
                    safety_cap=300):
    """Adam on u = log s exactly as eks/core.py:652-681 arranges it (per lane, vectorised):
    grad scaled by lr, optax.adam(1.0) (b1 .9, b2 .999, eps 1e-8, eps_root 0, bias-corrected),
    stop when |L - prev| < tol*|log(max(prev,1e-12))| + 1e-6 (prev finite), cap ``safety_cap``.
    The returned u includes the update of the stopping iteration.  ``loss_and_grad(u_clipped)``
    must return (L, dL/du) arrays; the clip of eks/core.py:642 (gradient zero outside the
    bounds, as jnp.clip differentiates) is applied here."""
    lo, hi = s_bounds_log
    u = np.array(u0, dtype=np.float64, copy=True)
    n = u.shape[0]
    mom = np.zeros(n)
    vel = np.zeros(n)
    prev = np.full(n, np.inf)
    iters = np.zeros(n, dtype=np.int64)
    done = np.zeros(n, dtype=bool)
    last = np.full(n, np.nan)
    b1, b2, eps = 0.9, 0.999, 1e-8
    while True:
        act = (~done) & (iters < safety_cap)
        if not act.any():
            break
        uc = np.clip(u, lo, hi)
        L, g = loss_and_grad(uc)
        g = np.where((u < lo) | (u > hi), 0.0, g) * lr
        cnt = iters + 1
        mom_n = b1 * mom + (1 - b1) * g
        vel_n = b2 * vel + (1 - b2) * g * g
        mhat = mom_n / (1 - b1 ** cnt)
        vhat = vel_n / (1 - b2 ** cnt)
        u_n = u - mhat / (np.sqrt(vhat) + eps)
        with np.errstate(invalid='ignore', divide='ignore'):
            rel = tol * np.abs(np.log(np.maximum(prev, 1e-12)))
            stop = np.isfinite(prev) & (np.abs(L - prev) < rel + 1e-6)
        u = np.where(act, u_n, u)
        mom = np.where(act, mom_n, mom)
        vel = np.where(act, vel_n, vel)
        prev = np.where(act, L, prev)
        last = np.where(act, L, last)
        iters = np.where(act, cnt, iters)
        done = np.where(act, stop, done)
    return u, last, iters


def optimize_smooth_param(ys, m0s, S0s, As, Cs, Qs, ensemble_vars_KTO, blocks, s_frames,
                          s_guess_per_k, *, lr=0.25, s_bounds_log=(-8.0, 8.0), tol=1e-2,
                          safety_cap=300, min_R_var=1e-4, jitter=0.0):
    """eks/core.py:306-559 + :562-699: one s per block by Adam on the summed constant-R NLL."""
    ys = np.asarray(ys, np.float64)
    K = ys.shape[0]
    if not blocks:
        blocks = [[k] for k in range(K)]
    Rd = build_R_from_vars(ensemble_vars_KTO)
    y_c, R_c = [], []
    for k in range(K):
        yk = crop_frames(ys[k], s_frames) if s_frames else ys[k]
        Rk = crop_frames(Rd[k], s_frames) if s_frames else Rd[k]
        y_c.append(yk)
        R_c.append(constant_R_from_timevarying(Rk, min_R_var))
    y_c = np.stack(y_c)
    R_c = np.stack(R_c)
    nb = len(blocks)
    u0 = np.empty(nb)
    for b, blk in enumerate(blocks):
        s0 = float(np.mean([s_guess_per_k[k] for k in blk])) if len(blk) > 1 \
            else float(s_guess_per_k[blk[0]])
        u0[b] = np.float32(np.log(np.clip(s0, 1e-6, 1e3)))      # float32 init, core.py:441/:622
    member_block = np.empty(K, dtype=int)
    for b, blk in enumerate(blocks):
        for k in blk:
            member_block[k] = b
    order = [k for blk in blocks for k in blk]

    def loss_and_grad(u_blocks):
        s_k = np.exp(u_blocks[member_block])
        nll, g = filter_nll(y_c[order], np.asarray(m0s)[order], np.asarray(S0s)[order],
                            np.asarray(As)[order], np.asarray(Cs)[order], np.asarray(Qs)[order],
                            s_k[order], R_c[order], jitter=jitter, want_grad=True)
        L = np.zeros(nb)
        G = np.zeros(nb)
        np.add.at(L, member_block[order], nll)
        np.add.at(G, member_block[order], g)
        return L, G

    u, last, iters = adam_optimize_s(loss_and_grad, u0, lr=lr, s_bounds_log=s_bounds_log,
                                     tol=tol, safety_cap=safety_cap)
    s_blocks = np.exp(np.clip(u, *s_bounds_log))
    return s_blocks[member_block], last, iters


def nll_grid(ys, m0s, S0s, As, Cs, Qs, ensemble_vars_KTO, s_candidates, s_frames=None,
             min_R_var=1e-4, jitter=0.0):
    """Build addition (BASELINE.json config 3): the same constant-R loss of eks/core.py:640-650
    evaluated on a grid of candidates.  Returns nll (K, n_cand)."""
    ys = np.asarray(ys, np.float64)
    K = ys.shape[0]
    Rd = build_R_from_vars(ensemble_vars_KTO)
    y_c = np.stack([crop_frames(ys[k], s_frames) if s_frames else ys[k] for k in range(K)])
    R_c = np.stack([constant_R_from_timevarying(
        crop_frames(Rd[k], s_frames) if s_frames else Rd[k], min_R_var) for k in range(K)])
    out = np.empty((K, len(s_candidates)))
    for j, sc in enumerate(s_candidates):
        out[:, j] = filter_nll(y_c, m0s, S0s, As, Cs, Qs, np.full(K, float(sc)), R_c,
                               jitter=jitter)
    return out


def run_kalman_smoother(ys, m0s, S0s, As, Cs, Qs, ensemble_vars, s_frames=None,
                        smooth_param=None, blocks=None, lr=0.25, s_bounds_log=(-8.0, 8.0),
                        tol=1e-2, safety_cap=300, jitter=0.0, s_mode='adam', n_grid=64):
    """eks/core.py:159-302.  ``ensemble_vars`` is (T,K,O) like upstream."""
    ys = np.asarray(ys, np.float64)
    K, T, O = ys.shape
    ev = np.swapaxes(np.asarray(ensemble_vars, np.float64), 0, 1)      # (K,T,O)
    Rd = build_R_from_vars(ev)
    guesses = np.array([compute_initial_guess(np.asarray(ensemble_vars)[:, k, :])
                        for k in range(K)])
    s_finals = np.empty(K)
    info = {}
    if smooth_param is not None:
        if isinstance(smooth_param, (int, float)):
            s_finals[:] = float(smooth_param)
        else:
            s_finals[:] = np.asarray(smooth_param, dtype=float)
    elif s_mode == 'adam':
        s_finals[:], last, iters = optimize_smooth_param(
            ys, m0s, S0s, As, Cs, Qs, ev, blocks, s_frames, guesses, lr=lr,
            s_bounds_log=s_bounds_log, tol=tol, safety_cap=safety_cap, jitter=jitter)
        info = dict(last_loss=last, iters=iters)
    else:
        cand = np.exp(np.linspace(s_bounds_log[0], s_bounds_log[1], n_grid))
        nll = nll_grid(ys, m0s, S0s, As, Cs, Qs, ev, cand, s_frames, jitter=jitter)
        idx = np.argmin(nll, axis=1)
        s_finals[:] = cand[idx]
        info = dict(nll=nll, argmin=idx, candidates=cand)
    ms, Vs, _ = kalman_smoother(ys, m0s, S0s, As, Cs, Qs, s_finals, Rd, jitter=jitter)
    return s_finals, ms, Vs, info


# --------------------------------------------------------------------------------------------
# stages either side of the filter
# --------------------------------------------------------------------------------------------
def ensemble(arr, avg_mode='median', var_mode='confidence_weighted_var', nan_replacement=1000.0):
    """eks/core.py:25-101.  arr (M,V,T,K,3) fields x,y,likelihood -> (1,V,T,K,5) fields
    x,y,var_x,var_y,likelihood.  Computed in float64 on float32-rounded inputs (upstream rounds
    the inputs to float32 at core.py:90-92)."""
    a = np.asarray(arr, np.float32).astype(np.float64)
    M = a.shape[0]
    x, y, lh = a[..., 0], a[..., 1], a[..., 2]
    avg = np.nanmedian if avg_mode == 'median' else np.nanmean
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ax = avg(x, axis=0)
            ay = avg(y, axis=0)
            conf = np.sum(lh, axis=0) / M
            if M == 1:
                vx = vy = 1.0 / np.maximum(conf, 1e-5)
            elif var_mode in ('conf_weighted_var', 'confidence_weighted_var'):
                vx = np.nanvar(x, axis=0) / conf
                vy = np.nanvar(y, axis=0) / conf
            else:
                vx = np.nanvar(x, axis=0)
                vy = np.nanvar(y, axis=0)
    vx = np.where(np.isnan(vx), nan_replacement, vx)
    vy = np.where(np.isnan(vy), nan_replacement, vy)
    # jnp.nan_to_num also maps +-inf to the dtype's largest finite value (float32 upstream)
    fmax = float(np.finfo(np.float32).max)
    vx = np.clip(vx, -fmax, fmax)
    vy = np.clip(vy, -fmax, fmax)
    return np.stack([ax, ay, vx, vy, conf], axis=-1)[None]


def center_predictions(ens, quantile_keep_pca):
    """eks/utils.py:293-365.  ens (1,V,T,K,5).  Returns (valid_mask (T,K) bool,
    centered (1,V,T,K,2), good_centered (1,V,min_frames,K,2), means (1,V,1,K,2),
    good_idx (K,min_frames) int)."""
    preds = ens[..., 0:2]
    vars_ = ens[..., 2:4]
    max_var = np.max(vars_, axis=(0, 1, 4))                      # (T,K)
    thr = np.percentile(max_var, quantile_keep_pca, axis=0)
    mask = max_var <= thr
    Kp = ens.shape[3]
    good = [np.where(mask[:, k])[0] for k in range(Kp)]
    nmin = min(len(g) for g in good)
    good = np.stack([g[:nmin] for g in good])                    # (K,nmin)
    means = np.stack([preds[:, :, good[k], k, :].mean(axis=2) for k in range(Kp)], axis=2)
    means = means[:, :, None]                                    # (1,V,1,K,2)
    centered = preds - means
    good_c = np.stack([centered[:, :, good[k], k, :] for k in range(Kp)], axis=3)
    return mask, centered, good_c, means, good


def singlecam_arrays(marker, avg_mode='median', var_mode='confidence_weighted_var', ens=None):
    """Inputs of run_kalman_smoother as eks/singlecam_smoother.py:140-181 + :246-284 build them.
    marker (M,1,T,K,3).  Everything downstream of `ensemble` is float64 here.  `ens` (1,1,T,K,5)
    overrides the ensemble stage (to isolate the stages after it)."""
    ens = ensemble(marker, avg_mode, var_mode) if ens is None else np.asarray(ens, np.float64)
    _, centered, _, means, _ = center_predictions(ens, 100)
    ys = np.transpose(centered[0, 0], (1, 0, 2))                 # (K,T,2)
    K = ys.shape[0]
    ev = ens[0, 0, :, :, 2:4]                                    # (T,K,2)
    m0s = np.zeros((K, 2))
    S0s = np.zeros((K, 2, 2))
    S0s[:, 0, 0] = np.nanvar(ys[:, :, 0], axis=1)
    S0s[:, 1, 1] = np.nanvar(ys[:, :, 1], axis=1)
    eye = np.tile(np.eye(2), (K, 1, 1))
    return dict(ys=ys, m0s=m0s, S0s=S0s, As=eye.copy(), Cs=eye.copy(), Qs=eye.copy(),
                ensemble_vars=ev, means=means, ens=ens)


def singlecam_outputs(arrs, s_finals, ms, Vs):
    """The 9 output fields per keypoint of eks/singlecam_smoother.py:183-241 as a (T, K*9) array
    (keypoint-major, label order x,y,likelihood,x_ens_median,y_ens_median,x_ens_var,y_ens_var,
    x_posterior_var,y_posterior_var)."""
    ens = arrs['ens']
    means = arrs['means']
    K, T, _ = ms.shape
    out = np.empty((T, K, 9))
    Cs = arrs['Cs']
    ym = np.einsum('kod,ktd->kto', Cs, ms)
    yv = np.einsum('kod,ktde,kpe->ktop', Cs, Vs, Cs)
    out[:, :, 0] = ym[:, :, 0].T + means[0, 0, 0, :, 0]
    out[:, :, 1] = ym[:, :, 1].T + means[0, 0, 0, :, 1]
    out[:, :, 2] = ens[0, 0, :, :, 4]
    out[:, :, 3] = ens[0, 0, :, :, 0]
    out[:, :, 4] = ens[0, 0, :, :, 1]
    out[:, :, 5] = ens[0, 0, :, :, 2]
    out[:, :, 6] = ens[0, 0, :, :, 3]
    out[:, :, 7] = yv[:, :, 0, 0].T
    out[:, :, 8] = yv[:, :, 1, 1].T
    return out.reshape(T, K * 9)


def stacked_views(a, k):
    """eks/marker_array.py:302-324: (1,V,T,K,F) -> (T, V*F) for keypoint k, order [c0f0,c0f1,c1f0..]."""
    sel = a[0, :, :, k, :]                                       # (V,T,F)
    return np.transpose(sel, (1, 0, 2)).reshape(sel.shape[1], -1)


def multicam_arrays(marker, quantile_keep_pca=50.0, n_latent=3, avg_mode='median',
                    var_mode='confidence_weighted_var', pca_fit=None, ens=None, inflate_vars=False):
    """Linear multicam inputs as eks/multicam_smoother.py:342-348, :412-430 and :554-597 build
    them (no variance inflation).  ``pca_fit(X, n) -> (components (n,F), mean (F,))`` defaults to
    an SVD PCA with sklearn's sign convention left to the caller.  `ens` (1,V,T,K,5) overrides the
    ensemble stage."""
    ens = ensemble(marker, avg_mode, var_mode) if ens is None else np.asarray(ens, np.float64)
    mask, centered, good_c, means, good_idx = center_predictions(ens, quantile_keep_pca)
    V, T, K = ens.shape[1], ens.shape[2], ens.shape[3]
    if pca_fit is None:
        def pca_fit(X, n):
            mu = X.mean(axis=0)
            U, S, Vt = np.linalg.svd(X - mu, full_matrices=False)
            # sklearn svd_flip(u_based_decision=False): largest |entry| of each row of Vt positive
            sgn = np.sign(Vt[np.arange(Vt.shape[0]), np.argmax(np.abs(Vt), axis=1)])
            return (Vt * sgn[:, None])[:n], mu
    vars_used = ens[..., 2:4]
    if inflate_vars:
        vars_used = inflate_variances(centered, vars_used, n_latent)
    ys, evs, Cs, S0s, Qs = [], [], [], [], []
    for k in range(K):
        Xg = stacked_views(good_c, k)
        Xa = stacked_views(centered, k)
        comp, mu = pca_fit(Xg, n_latent)
        pcs = (Xa - mu) @ comp.T
        good_pcs = pcs[np.where(mask[:, k])[0]]
        S0s.append(np.diag(np.var(good_pcs, axis=0)))
        d = good_pcs[1:] - good_pcs[:-1]
        cov = np.atleast_2d(np.cov(d.T))
        mx = np.max(np.abs(cov))
        Qs.append(cov / mx if mx > 0 else cov)
        Cs.append(comp.T)
        ys.append(Xa)
        evs.append(stacked_views(vars_used, k))
    ys = np.stack(ys)
    evs = np.stack(evs)                                          # (K,T,2V)
    eye = np.tile(np.eye(n_latent), (K, 1, 1))
    return dict(ys=ys, m0s=np.zeros((K, n_latent)), S0s=np.stack(S0s), As=eye,
                Cs=np.stack(Cs), Qs=np.stack(Qs), ensemble_vars=np.swapaxes(evs, 0, 1),
                means=means, ens=ens, mask=mask, good_idx=good_idx)


def mahalanobis_loop(x, v, n_latent=3, v_quantile_threshold=50.0, epsilon=1e-6):
    """Loop-style restatement of eks/stats.py:67-157 (default call: no likelihood filter, no
    supplied loading matrix).  Returns {view: (N,) Mahalanobis distance}."""
    from sklearn.decomposition import FactorAnalysis
    worst = v.max(axis=1)
    rows = worst < np.percentile(worst, v_quantile_threshold)
    fa = FactorAnalysis(n_components=n_latent).fit(x[rows])
    W, mu = fa.components_.T, fa.mean_
    N, n_views = x.shape[0], x.shape[1] // 2
    out = {c: np.zeros(N) for c in range(n_views)}
    for i in range(N):
        Dinv = np.diag(1.0 / (v[i] + epsilon))
        B = np.linalg.inv(W.T @ Dinv @ W)
        z = B @ W.T @ Dinv @ (x[i] - mu)
        diff = x[i] - (W @ z + mu)
        for c in range(n_views):
            sl = slice(2 * c, 2 * c + 2)
            Qc = np.diag(v[i, sl]) + W[sl] @ B @ W[sl].T
            out[c][i] = diff[sl] @ np.linalg.inv(Qc) @ diff[sl]
    return out


def inflate_variances(centered, vars_, n_latent=3, threshold=5.0, scalar=10.0):
    """eks/multicam_smoother.py:653-764 with the default kwargs: per keypoint, inflate x10 the
    variances of (frame, view) pairs with Mahalanobis distance > 5 (whole frame when there are two
    views), refit, repeat until nothing changes.  centered, vars_ (1,V,T,K,2) -> inflated vars."""
    V, K = centered.shape[1], centered.shape[3]
    out = np.array(vars_, dtype=np.float64, copy=True)
    for k in range(K):
        x = stacked_views(centered, k)
        cur = stacked_views(vars_, k).astype(np.float64)
        while True:
            M = mahalanobis_loop(x, cur, n_latent)
            hit = np.stack([M[c] > threshold for c in range(V)], axis=1)
            mask = np.repeat(hit, 2, axis=1)
            if V == 2:
                mask = mask | mask.any(axis=1, keepdims=True)
            if not mask.any():
                break
            cur = np.where(mask, cur * scalar, cur)
        out[0, :, :, k, :] = np.transpose(cur.reshape(cur.shape[0], V, 2), (1, 0, 2))
    return out


def multicam_outputs(arrs, ms, Vs):
    """Per-camera (T, K*9) arrays of eks/multicam_smoother.py:481-527 (posterior var includes
    + ensemble var, :509-510) and the latent (T, K*2D) array of :529-544."""
    ens, means = arrs['ens'], arrs['means']
    V = ens.shape[1]
    K, T, D = ms.shape
    Cs = arrs['Cs']
    ev = np.swapaxes(arrs['ensemble_vars'], 0, 1)               # (K,T,2V)
    ym = np.einsum('kod,ktd->kto', Cs, ms)
    yv = np.einsum('kod,ktde,kpe->ktop', Cs, Vs, Cs)
    cams = []
    for c in range(V):
        xi, yi = 2 * c, 2 * c + 1
        out = np.empty((T, K, 9))
        out[:, :, 0] = ym[:, :, xi].T + means[0, c, 0, :, 0]
        out[:, :, 1] = ym[:, :, yi].T + means[0, c, 0, :, 1]
        out[:, :, 2] = ens[0, c, :, :, 4]
        out[:, :, 3] = ens[0, c, :, :, 0]
        out[:, :, 4] = ens[0, c, :, :, 1]
        out[:, :, 5] = ev[:, :, xi].T           # (inflated) ensemble variances, :505-508
        out[:, :, 6] = ev[:, :, yi].T
        out[:, :, 7] = (yv[:, :, xi, xi] + ev[:, :, xi]).T
        out[:, :, 8] = (yv[:, :, yi, yi] + ev[:, :, yi]).T
        cams.append(out.reshape(T, K * 9))
    lat = np.empty((T, K, 2 * D))
    lat[:, :, :D] = np.transpose(ms, (1, 0, 2))
    lat[:, :, D:] = np.transpose(np.diagonal(Vs, axis1=2, axis2=3), (1, 0, 2))
    return cams, lat.reshape(T, K * 2 * D)


# --------------------------------------------------------------------------------------------
# IBL pupil smoother: eks/ibl_pupil_smoother.py:34-607
# --------------------------------------------------------------------------------------------
PUPIL_KEYPOINTS = ('pupil_top_r', 'pupil_bottom_r', 'pupil_right_r', 'pupil_left_r')
# rows: top x,y / bottom x,y / right x,y / left x,y ; columns: diameter, com_x, com_y
# (eks/ibl_pupil_smoother.py:271-276)
PUPIL_C = np.array([[0, 1, 0], [-.5, 0, 1], [0, 1, 0], [.5, 0, 1],
                    [.5, 1, 0], [0, 0, 1], [-.5, 1, 0], [0, 0, 1]], dtype=np.float64)


def _median2(a, b, skip_nan):
    """Median of two numbers per frame: their mean; with skip_nan the non-NaN one survives
    (np.nanmedian) - otherwise a NaN poisons the result (np.median)."""
    both = 0.5 * (a + b)
    if not skip_nan:
        return both
    return np.where(np.isnan(a), b, np.where(np.isnan(b), a, both))


def pupil_location(p):
    """eks/ibl_pupil_smoother.py:34-60.  p (T,8) in PUPIL_KEYPOINTS x/y order -> centre (T,2).
    x: nan-tolerant over (top, bottom), strict over (right, left), then nan-tolerant over the two;
    y: strict over (top, bottom), nan-tolerant over (right, left), then nan-tolerant."""
    tx, ty, bx, by, rx, ry, lx, ly = (p[:, i] for i in range(8))
    cx = _median2(_median2(tx, bx, True), _median2(rx, lx, False), True)
    cy = _median2(_median2(ty, by, False), _median2(ry, ly, True), True)
    return np.stack([cx, cy], axis=1)


def pupil_diameter(p):
    """eks/ibl_pupil_smoother.py:63-91: nanmedian of six estimates - the two direct diameters and
    sqrt(2) x the four adjacent-point distances (circle assumption)."""
    pts = {n: p[:, 2 * i:2 * i + 2] for i, n in enumerate(('top', 'bottom', 'right', 'left'))}

    def dist(a, b):
        d = pts[a] - pts[b]
        return np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])

    est = [dist('top', 'bottom'), dist('left', 'right')]
    for a, b in (('top', 'left'), ('top', 'right'), ('bottom', 'left'), ('bottom', 'right')):
        est.append(dist(a, b) * 2 ** 0.5)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=RuntimeWarning)
        return np.nanmedian(np.stack(est), axis=0)


def pupil_arrays(marker, avg_mode='median', var_mode='confidence_weighted_var', ens=None):
    """Inputs of run_pupil_kalman_smoother as eks/ibl_pupil_smoother.py:233-289 builds them.
    marker (M,1,T,4,3) with keypoints in PUPIL_KEYPOINTS order."""
    ens = ensemble(marker, avg_mode, var_mode) if ens is None else np.asarray(ens, np.float64)
    T = ens.shape[2]
    preds = ens[0, 0, :, :, 0:2].reshape(T, 8)
    evars = ens[0, 0, :, :, 2:4].reshape(T, 8)
    likes = ens[0, 0, :, :, 4]
    diam = pupil_diameter(preds)
    loc = pupil_location(preds)
    mean_x, mean_y = np.mean(loc[:, 0]), np.mean(loc[:, 1])
    xo, yo = loc[:, 0] - mean_x, loc[:, 1] - mean_y
    m0 = np.array([np.mean(diam), 0.0, 0.0])
    S0 = np.diag([np.nanvar(diam), np.nanvar(xo), np.nanvar(yo)])
    ys = preds.copy()
    ys[:, 0::2] -= mean_x
    ys[:, 1::2] -= mean_y
    return dict(ys=ys, m0=m0, S0=S0, C=PUPIL_C.copy(), ensemble_vars=evars, likes=likes,
                preds=preds, mean_x=mean_x, mean_y=mean_y,
                latent_vars=np.array([np.var(diam), np.var(xo), np.var(yo)]))


def pupil_dynamics(s_d, s_c, latent_vars):
    """A = diag(s_d, s_c, s_c), Q = diag(var * (1 - s^2)); eks/ibl_pupil_smoother.py:427-432."""
    a = np.array([s_d, s_c, s_c], dtype=np.float64)
    return np.diag(a), np.diag(np.asarray(latent_vars, np.float64) * (1.0 - a * a))


def pupil_to_stable_s(u, eps=1e-3):
    """sigmoid(u) * (1 - 2 eps) + eps and its derivative; eks/ibl_pupil_smoother.py:506-508."""
    sig = 1.0 / (1.0 + np.exp(-np.asarray(u, np.float64)))
    return sig * (1.0 - 2 * eps) + eps, sig * (1.0 - sig) * (1.0 - 2 * eps)


def pupil_nll_and_grad(u, ys, m0, S0, C, ensemble_vars, latent_vars, use_c=True):
    """Loss of eks/ibl_pupil_smoother.py:540-552 (filter NLL with the time-varying R_t) and its
    gradient w.r.t. u = (u_diam, u_com).  Gradient by forward sensitivities through the filter
    (``kalman_filter(tangent=...)``), or - with use_c and the C twin built - by complex-step
    differentiation of the C filter (an independent route to the same number)."""
    s, ds = pupil_to_stable_s(u)
    lv = np.asarray(latent_vars, np.float64)
    A, Q = pupil_dynamics(s[0], s[1], lv)
    dA = [np.diag([ds[0], 0.0, 0.0]), np.diag([0.0, ds[1], ds[1]])]
    dQ = [np.diag([-2 * s[0] * ds[0] * lv[0], 0.0, 0.0]),
          np.diag([0.0, -2 * s[1] * ds[1] * lv[1], -2 * s[1] * ds[1] * lv[2]])]
    Rd = np.maximum(np.asarray(ensemble_vars, np.float64), 1e-12)
    if use_c:
        from . import c_oracle
        return c_oracle.nll_directional(ys, Rd, m0, S0, A, C, Q, dA, dQ)
    g = np.empty(2)
    for i in range(2):
        f = kalman_filter(ys[None], m0[None], S0[None], A[None], C[None], Q[None], 1.0, Rd[None],
                          tangent=(dA[i][None], dQ[i][None]))
        g[i] = -f['dll'][0]
    return -f['ll'][0], g


def pupil_optimize_smooth(ys, m0, S0, C, ensemble_vars, latent_vars, s_frames=None,
                          smooth_params=None, lr=5e-3, tol=1e-6, safety_cap=5000, use_c=True):
    """eks/ibl_pupil_smoother.py:451-607.  Both parameters given: clip to [1e-3, 1-1e-3] after
    rounding to float32 (:555-557).  Otherwise Adam (optax.adam(lr): b1 .9, b2 .999, eps 1e-8,
    bias-corrected) on u from s0 = (0.99, 0.98), one loss for both parameters, stop when
    |L - prev| < tol*|log(max(prev,1e-12))| + 1e-6 (prev finite) or at ``safety_cap`` (:571-594).
    Returns (s_diam, s_com, iters, last_loss)."""
    if smooth_params is not None and all(v is not None for v in smooth_params):
        s = np.clip(np.asarray(smooth_params, dtype=np.float32), np.float32(1e-3),
                    np.float32(1 - 1e-3))
        return float(s[0]), float(s[1]), 0, float('nan')
    y_loss = crop_frames(np.asarray(ys, np.float64), s_frames)
    v_loss = crop_frames(np.asarray(ensemble_vars, np.float64), s_frames)
    s0 = np.array([0.99, 0.98], dtype=np.float32).astype(np.float64)
    u = np.log(s0 / (1.0 - s0))
    mom = np.zeros(2)
    vel = np.zeros(2)
    prev, iters, last = np.inf, 0, np.nan
    b1, b2, eps = 0.9, 0.999, 1e-8
    while iters < safety_cap:
        L, g = pupil_nll_and_grad(u, y_loss, m0, S0, C, v_loss, latent_vars, use_c=use_c)
        iters += 1
        mom = b1 * mom + (1 - b1) * g
        vel = b2 * vel + (1 - b2) * g * g
        u = u - lr * (mom / (1 - b1 ** iters)) / (np.sqrt(vel / (1 - b2 ** iters)) + eps)
        with np.errstate(invalid='ignore', divide='ignore'):
            stop = np.isfinite(prev) and \
                abs(L - prev) < tol * abs(np.log(max(prev, 1e-12))) + 1e-6
        prev = last = L
        if stop:
            break
    s, _ = pupil_to_stable_s(u)
    return float(s[0]), float(s[1]), iters, float(last)


def run_pupil_kalman_smoother(ys, m0, S0, C, ensemble_vars, latent_vars, s_frames=None,
                              smooth_params=None, **opt):
    """eks/ibl_pupil_smoother.py:363-448: optimise (s_diam, s_com) on the (cropped) loss, then the
    smoother over all frames with A(s), Q(s) and the time-varying R_t.
    Returns ([s_d, s_c], ms (T,3), Vs (T,3,3), info)."""
    s_d, s_c, iters, last = pupil_optimize_smooth(ys, m0, S0, C, ensemble_vars, latent_vars,
                                                  s_frames, smooth_params, **opt)
    A, Q = pupil_dynamics(s_d, s_c, latent_vars)
    Rd = np.maximum(np.asarray(ensemble_vars, np.float64), 1e-12)
    ms, Vs, nll = kalman_smoother(np.asarray(ys, np.float64)[None], m0[None], S0[None], A[None],
                                  C[None], Q[None], 1.0, Rd[None])
    return [s_d, s_c], ms[0], Vs[0], dict(iters=iters, last_loss=last, nll=float(nll[0]))


def pupil_outputs(arrs, ms, Vs):
    """The (T, 36) output table of eks/ibl_pupil_smoother.py:303-359, column order = the
    DataFrame's: for each position i in (top, right, bottom, left) the nine fields x, y,
    likelihood, x/y_ens_median, x/y_ens_var, x/y_posterior_var.  Upstream quirks kept because
    the CSV is the contract: the data are gathered in (top, right, bottom, left) order while the
    column header is built from keypoint_names (top, bottom, right, left); likelihood i is
    keypoint_names[i]'s; the posterior variances are entries (i, i) and (i+1, i+1) of C V C'."""
    C = arrs['C']
    ym = ms @ C.T
    ym[:, 0::2] += arrs['mean_x']
    ym[:, 1::2] += arrs['mean_y']
    yv = np.einsum('od,tde,pe->top', C, Vs, C)
    T = ms.shape[0]
    out = np.empty((T, 4, 9))
    for i, (cx, cy) in enumerate(((0, 1), (4, 5), (2, 3), (6, 7))):
        out[:, i, 0] = ym[:, cx]
        out[:, i, 1] = ym[:, cy]
        out[:, i, 2] = arrs['likes'][:, i]
        out[:, i, 3] = arrs['preds'][:, cx]
        out[:, i, 4] = arrs['preds'][:, cy]
        out[:, i, 5] = arrs['ensemble_vars'][:, cx]
        out[:, i, 6] = arrs['ensemble_vars'][:, cy]
        out[:, i, 7] = yv[:, i, i]
        out[:, i, 8] = yv[:, i + 1, i + 1]
    return out.reshape(T, 36)
