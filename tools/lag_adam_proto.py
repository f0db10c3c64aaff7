"""NumPy float64 prototype of the Adam search's loss from CACHED LAG SUMS (round 6; the device form is
eks_amd/csrc/eks_lag_adam.hip).  Test / design tool: nothing in eks_amd imports it.

Loss of eks/core.py:640-650 on a scalar chain (x' = a x + N(0, s q), y = c x + N(0, r), r constant):

    frames [0, B0)   : exactly, time-parallel.  The predicted variance obeys a Moebius map with fixed points
                       P_inf > 0 > P_-; (P_t - P_inf) / (P_t - P_-) = kappa^t w_0 with kappa = S_- / S_inf = rho^2,
                       so S_t = S_inf + (S_inf - S_-) w_t / (1 - w_t) for every t at once, and the innovations follow
                       e_{t+1} = rho_t e_t + u_{t+1},  rho_t = a r / S_t,  u_t = y_t - a y_{t-1}   (a linear scan).
    frames [B0, T)   : converged variance.  d_t = Dz_t + rho^(t-B0) E with E = e_B0 and Dz the zero-start recursion on
                       the inputs u_t, t >= F = B0 + 1, which do NOT depend on s or r:
                         sum Dz_t^2 = [c_0 + 2 sum_{k=1..L} rho^k c_k - rho^2 Dz_{T-1}^2] / (1 - rho^2),
                         c_k = sum_{t >= F + k} u_t u_{t-k}                                  (cached once per search)
                         sum_t rho^(t-B0) Dz_t = rho Z / (1 - rho^2),  Z = sum_m rho^m u_{F+m}   (head inputs)
                         Dz_{T-1} = sum_m rho^m u_{T-1-m}                                       (tail inputs)

`python tools/lag_adam_proto.py T K` checks value and gradient against the sequential filter on the bench's synthetic
session and runs the optimiser, reporting the poles it visits.
"""
from __future__ import annotations

import sys

import numpy as np


class Dual:
    __slots__ = ('v', 'd')
    __array_ufunc__ = None           # ndarray * Dual -> Dual.__rmul__

    def __init__(self, v, d=None):
        self.v = np.asarray(v, float)
        self.d = np.zeros_like(self.v) if d is None else np.asarray(d, float)

    @staticmethod
    def of(o):
        return o if isinstance(o, Dual) else Dual(o)

    def __add__(self, o):
        o = Dual.of(o)
        return Dual(self.v + o.v, self.d + o.d)
    __radd__ = __add__

    def __sub__(self, o):
        o = Dual.of(o)
        return Dual(self.v - o.v, self.d - o.d)

    def __rsub__(self, o):
        return Dual.of(o) - self

    def __mul__(self, o):
        o = Dual.of(o)
        return Dual(self.v * o.v, self.d * o.v + self.v * o.d)
    __rmul__ = __mul__

    def __truediv__(self, o):
        o = Dual.of(o)
        return Dual(self.v / o.v, (self.d * o.v - self.v * o.d) / (o.v * o.v))

    def __rtruediv__(self, o):
        return Dual.of(o) / self

    def row(self):
        return Dual(self.v[None], self.d[None])


def dlog(x):
    return Dual(np.log(x.v), x.d / x.v)


def dsqrt(x):
    r = np.sqrt(x.v)
    return Dual(r, 0.5 * x.d / r)


def dpow(x, n):
    """x ** n for whole n >= 0 (arrays broadcast)."""
    with np.errstate(all='ignore'):
        return Dual(x.v ** n, np.where(n > 0, n * x.v ** np.maximum(n - 1, 0) * x.d, 0.0))


def sequential_nll(y, m0, P0, a, c, q, r, s):
    """The filter as the reference runs it (update, then predict), one chain per column of y."""
    T, N = y.shape
    m, P, ll = m0.copy(), P0.copy(), np.zeros(N)
    for t in range(T):
        S = c * c * P + r
        e = y[t] - c * m
        ll += -0.5 * (np.log(2 * np.pi * S) + e * e / S)
        m = a * (m + P * c / S * e)
        P = a * a * P * r / S + s * q
    return -ll


def precompute(y, a, B0=256, L=255):
    """What one streaming pass leaves behind: lag sums, head rows, tail rows."""
    T, N = y.shape
    F = B0 + 1
    u = np.zeros_like(y)
    u[1:] = y[1:] - a * y[:-1]
    uz = u.copy()
    uz[:F] = 0.0
    ck = np.stack([np.einsum('tn,tn->n', uz[k:], uz[:T - k]) for k in range(L + 1)])
    m = np.arange(L + 1)
    return dict(ck=ck, u_head=u[:B0 + 1], u_F=uz[np.minimum(F + m, T - 1)] * (F + m < T)[:, None],
                u_tail=uz[np.maximum(T - 1 - m, 0)] * (T - 1 - m >= 0)[:, None], T=T, B0=B0, L=L, y0=y[0])


def lag_loss(theta, pre, m0, P0, a, c, q, r):
    """nll, d nll / d theta, rho for theta = log s per chain."""
    T, B0, L = pre['T'], pre['B0'], pre['L']
    N = theta.shape[0]
    s = np.exp(theta)
    sq = Dual(s * q, s * q)
    c2 = c * c
    beta = r * (1 - a * a) - sq * c2                 # c2 P^2 + beta P - sq r = 0
    disc = dsqrt(beta * beta + 4 * c2 * r * sq)
    Pinf = (disc - beta) / (2 * c2)
    Sinf = c2 * Pinf + r
    Sm = (a * a * r * r) / Sinf
    Pm = (Sm - r) / c2
    rho = (a * r) / Sinf
    kap = Sm / Sinf
    w0 = (P0 - Pinf) / (P0 - Pm)
    t = np.arange(B0)[:, None]
    wt = w0.row() * dpow(kap.row(), t)
    St = Sinf.row() + (Sinf - Sm).row() * wt / (1 - wt)
    St.v[0], St.d[0] = c2 * P0 + r, 0.0
    rt = (a * r) / St
    # the scan, written as a loop here (64-lane scans + LDS on the device)
    e = Dual(pre['y0'] - c * m0)
    terms = Dual(np.zeros(N))
    uh = pre['u_head']
    for tt in range(B0):
        Stt = Dual(St.v[tt], St.d[tt])
        terms = terms + dlog(Stt) + e * e / Stt
        e = Dual(rt.v[tt], rt.d[tt]) * e + (uh[tt + 1] if tt + 1 <= B0 else 0.0)
    E = e                                            # innovation of frame B0
    n = T - B0
    k = np.arange(L + 1)[:, None]
    rk = dpow(rho.row(), k)
    cc = pre['ck'].copy()
    cc[1:] *= 2

    def dot(x):
        return Dual((rk.v * x).sum(0), (rk.d * x).sum(0))
    poly, Dl, Z = dot(cc), dot(pre['u_tail']), dot(pre['u_F'])
    om = 1 - rho * rho
    SS = (poly - rho * rho * Dl * Dl) / om
    X1 = rho * Z / om
    tot = SS + 2 * E * X1 + E * E * (1 - dpow(rho, 2 * n)) / om
    nll = 0.5 * (T * np.log(2 * np.pi) + terms + n * dlog(Sinf) + tot / Sinf)
    return nll.v, nll.d, rho.v


def adam(loss, u0, lr=0.25, lo=-8.0, hi=8.0, tol=1e-2, cap=300):
    """eks/core.py:652-681 per chain group (here: per keypoint = D consecutive chains handled by `loss`)."""
    u = u0.copy()
    mom = np.zeros_like(u)
    vel = np.zeros_like(u)
    prev = np.full_like(u, np.inf)
    iters = np.zeros_like(u)
    done = np.zeros(u.shape, bool)
    trace = []
    for it in range(cap):
        if done.all():
            break
        L, g, rho = loss(np.clip(u, lo, hi))
        g = np.where((u < lo) | (u > hi), 0.0, g) * lr
        run = ~done
        cnt = iters + 1
        mom_n = 0.9 * mom + 0.1 * g
        vel_n = 0.999 * vel + 0.001 * g * g
        un = u - (mom_n / (1 - 0.9 ** cnt)) / (np.sqrt(vel_n / (1 - 0.999 ** cnt)) + 1e-8)
        with np.errstate(all='ignore'):
            stop = np.isfinite(prev) & (np.abs(L - prev) < tol * np.abs(np.log(np.maximum(prev, 1e-12))) + 1e-6)
        u = np.where(run, un, u)
        mom = np.where(run, mom_n, mom)
        vel = np.where(run, vel_n, vel)
        prev = np.where(run, L, prev)
        iters = np.where(run, cnt, iters)
        trace.append(np.where(run, rho, np.nan))
        done = done | (run & stop)
    return u, iters, np.array(trace)


def main():
    sys.path.insert(0, __file__.rsplit('/tools/', 1)[0])
    from eks_amd import synth
    T, K = int(sys.argv[1]), int(sys.argv[2])
    y, var = synth.singlecam_observations_torch(T, K, seed=3, device='cpu')
    y = y.numpy().reshape(T, -1).astype(np.float64)
    var = var.numpy().reshape(T, -1).astype(np.float64)
    N = y.shape[1]
    r = np.maximum(np.median(var, axis=0), 1e-4)
    a = np.ones(N)
    c = np.ones(N)
    q = np.ones(N)
    m0 = np.zeros(N)
    P0 = y.var(axis=0)
    pre = precompute(y, a)
    rng = np.random.default_rng(0)
    for trial in range(2):
        th = rng.uniform(-3.5, 2.0, size=N)
        v, g, rho = lag_loss(th, pre, m0, P0, a, c, q, r)
        ref = sequential_nll(y, m0, P0, a, c, q, r, np.exp(th))
        h = 1e-5
        fd = (sequential_nll(y, m0, P0, a, c, q, r, np.exp(th + h)) -
              sequential_nll(y, m0, P0, a, c, q, r, np.exp(th - h))) / (2 * h)
        ok = rho < 0.906
        print(f'trial {trial}: rho {rho.min():.3f}..{rho.max():.3f}  in range {ok.sum()}/{N}  '
              f'max |nll - seq| / nll {np.max(np.abs(v - ref)[ok] / ref[ok]):.2e}  '
              f'abs {np.max(np.abs(v - ref)[ok]):.2e}  grad rel {np.max(np.abs(g - fd)[ok] / np.abs(fd)[ok]):.2e}  '
              f'out of range: value rel {np.max(np.abs(v - ref)[~ok] / ref[~ok], initial=0):.2e}')
    # the search itself, one s per keypoint (D = 2 chains): losses and gradients add
    ev = var.reshape(T, K, 2)[:2000]
    d = np.swapaxes(ev[1:] - ev[:-1], 0, 1).reshape(K, -1)
    g0 = np.array([round(float(x), 5) for x in np.nanstd(d.astype(np.float32), axis=1)])
    u0 = np.log(np.clip(np.where(g0 > 0, g0, 2.0), 1e-6, 1e3)).astype(np.float32).astype(np.float64)

    def kp_loss(u):
        v, g, rho = lag_loss(np.repeat(u, 2), pre, m0, P0, a, c, q, r)
        return v.reshape(K, 2).sum(1), g.reshape(K, 2).sum(1), rho.reshape(K, 2).max(1)
    u, iters, trace = adam(kp_loss, u0)
    print(f'search: iterations {iters.min():.0f}..{iters.max():.0f}; s {np.exp(u).min():.4f}..{np.exp(u).max():.3f}; '
          f'largest pole visited {np.nanmax(trace):.4f}; at the end {np.nanmax(trace[-1]) if len(trace) else 0:.4f}')
    worst = np.nanmax(trace, axis=0)
    print('keypoints whose search ever leaves rho <= 0.906:', int((worst > 0.906).sum()), 'of', K,
          '; <= 0.76:', int((worst <= 0.76).sum()))


if __name__ == '__main__':
    main()
