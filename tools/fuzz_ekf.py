"""Randomised parity sweep of eks_ekf_smooth against the sequential extended Kalman smoother of
oracle/ekf_oracle.py: random lengths, camera counts, smoothing parameters over eight decades,
variance scales with occlusion spikes, cold / noisy starts of the linearisation points, constant
and time-varying R.  usage: fuzz_ekf.py [n_cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from eks_amd import hip_ops, synth
from oracle import ekf_oracle as ek


def run_case(rng, i, verbose=True):
    T = int(rng.choice([1, 2, 3, 31, 32, 33, 64, 65, int(rng.integers(100, 3000))]))
    K = int(rng.integers(1, 5))
    V = int(rng.integers(2, 6))
    prob = synth.calibrated_multicam(max(T, 12), K, V, seed=int(rng.integers(1 << 30)))
    y = prob['y_tko'][:T].astype(np.float32)
    var = (prob['var_tko'][:T] * np.exp(rng.uniform(-5, 5))).astype(np.float32)
    const = bool(rng.random() < 0.4)
    s = np.exp(rng.uniform(np.log(1e-4), np.log(1e4), size=K))
    dev = torch.device('cuda')
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    m0 = t(prob['m0s'])
    start = rng.choice(['prior', 'noisy'])
    xlin = m0[:, None, :].expand(K, T, 3).contiguous()
    if start == 'noisy':
        xlin = xlin + 30.0 * torch.randn_like(xlin)
    rconst = np.maximum(np.median(var.astype(np.float64), axis=0), 1e-4) if const else None
    ms, Vs, nll, info = hip_ops.ekf_smooth(t(y, torch.float32), None if const else t(var, torch.float32),
                                           t(rconst) if const else None, m0, t(prob['S0s']), t(prob['As']),
                                           t(prob['Qs']), t(s), t(prob['cams_packed']), xlin,
                                           max_sweeps=48, tol=1e-10)
    info = info.cpu().numpy()
    ms, Vs, nll = ms.cpu().numpy(), Vs.cpu().numpy(), nll.cpu().numpy()
    h = ek.combine_projections([ek.make_projection_fn(c['rot'], c['tvec'], c['K'], c['dist'])
                                for c in prob['cams']])
    worst = dict(ms=0.0, Vs=0.0, nll=0.0)
    for k in range(K):
        Rk = rconst[k] if const else np.maximum(var[:, k].astype(np.float64), 1e-12)
        mo, Vo, ll = ek.eks_smoother(y[:, k].astype(np.float64), Rk, prob['m0s'][k], prob['S0s'][k],
                                     prob['As'][k], prob['Qs'][k], s[k], h)
        worst['ms'] = max(worst['ms'], np.abs(ms[:, k] - mo).max() / max(np.abs(mo).max(), 1.0))
        worst['Vs'] = max(worst['Vs'], np.abs(Vs[:, k] - Vo).max() / np.abs(Vo).max())
        worst['nll'] = max(worst['nll'], abs(nll[k] + ll) / abs(ll))
    bad = max(worst.values()) > 1e-5 or not info[1] <= 1e-10
    if verbose or bad:
        print(f'case {i}: T={T} K={K} V={V} const_R={const} start={start} sweeps={info[0]:.0f} '
              f'resid={info[1]:.1e} ms {worst["ms"]:.1e} Vs {worst["Vs"]:.1e} nll {worst["nll"]:.1e}'
              + ('   <-- above 1e-5' if bad else ''), flush=True)
    return worst, info, bad


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    tot = dict(ms=0.0, Vs=0.0, nll=0.0)
    sweeps, n_bad = 0, 0
    for i in range(n):
        w, info, bad = run_case(rng, i)
        tot = {k: max(tot[k], w[k]) for k in tot}
        sweeps = max(sweeps, int(info[0]))
        n_bad += bad
    print('worst', tot, 'max sweeps', sweeps, 'cases above tolerance', n_bad)


if __name__ == '__main__':
    main()
