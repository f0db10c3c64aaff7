"""GPU check of the Adam search from cached lag sums (eks_lag_adam.hip) against the streaming forms of the same
library (EKS_ADAM_STREAM=1) and, on a few keypoints, the oracle's optimiser fed by the C port's complex-step gradient.
Prints per shape: iterations, max |d log s| (lag form vs streaming form, in-block streaming fallback vs both) and the
time of the search.  `python tools/lag_adam_check.py [quick]`."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import _lib, hip_ops                                                    # noqa: E402
from tests.test_gpu_kernels import _dev, _params_dev, _singlecam_problem            # noqa: E402


def knob(name, value):
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = value
    _lib.load().eks_knobs_reload()


def search(y, rc, params, flags, K, u0, cap=300, stride=None, time_it=False):
    offs = torch.arange(K + 1, dtype=torch.int32, device='cuda')
    mem = torch.arange(K, dtype=torch.int32, device='cuda')

    def once():
        state = np.zeros((K, 6))
        state[:, 0] = u0
        state[:, 3] = np.inf
        state = _dev(state)
        s_kp = _dev(np.exp(np.clip(u0, -8, 8)))
        loop = hip_ops.AdamLoop(y, rc, *params, offs, mem, state, s_kp, 0.25, -8.0, 8.0, 1e-2, cap, flags=flags)
        n = stride or loop.stride()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = 0
        while it < cap:
            loop.run(min(n, cap - it))
            it += n
            if int(loop.n_active.item()) == 0:
                break
        torch.cuda.synchronize()
        return time.perf_counter() - t0, loop.stride(), state.cpu().numpy(), s_kp.cpu().numpy(), loop.nll.cpu().numpy()
    out = once()
    if time_it:
        dts = [once()[0] for _ in range(5)]
        out = (float(np.median(dts)),) + out[1:]
    return out


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
    shapes = [(30_000, 70, True), (20_011, 33, False), (4_500, 40, True), (1_024, 3, True), (100_000, 256, True)]
    if quick:
        shapes = shapes[:3]
    for T, K, unit in shapes:
        arrs, y_tk, var_tk = _singlecam_problem(T, K, seed=31 + T, unit=unit)
        flags = hip_ops.model_flags(arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'])
        y, rc = _dev(y_tk), hip_ops.const_r(_dev(var_tk), 1e-4)
        params = _params_dev(arrs)
        u0 = np.log(np.random.default_rng(T).uniform(0.05, 50.0, K))
        knob('EKS_ADAM_STREAM', None)
        knob('EKS_ADAM_LAG_RHO_PPM', None)
        dt_l, n_l, st_l, s_l, nll_l = search(y, rc, params, flags, K, u0, time_it=True)
        knob('EKS_ADAM_LAG_RHO_PPM', '0')            # every chain streams its own frames inside the search kernel
        dt_f, n_f, st_f, s_f, nll_f = search(y, rc, params, flags, K, u0, time_it=True)
        knob('EKS_ADAM_LAG_RHO_PPM', None)
        knob('EKS_ADAM_STREAM', '1')                 # the chip-wide streaming forms
        dt_s, n_s, st_s, s_s, nll_s = search(y, rc, params, flags, K, u0, time_it=True)
        knob('EKS_ADAM_STREAM', None)
        same_it = int((st_l[:, 4] == st_s[:, 4]).sum())
        same_it_f = int((st_f[:, 4] == st_s[:, 4]).sum())
        print(f'T={T} K={K} unit={unit}: strides {n_l}/{n_f}/{n_s}; iterations {st_s[:, 4].min():.0f}..{st_s[:, 4].max():.0f}; '
              f'same stopping iteration lag {same_it}/{K} fallback {same_it_f}/{K}; '
              f'max|dlog s| lag-vs-stream {np.abs(np.log(s_l) - np.log(s_s)).max():.2e} '
              f'fallback-vs-stream {np.abs(np.log(s_f) - np.log(s_s)).max():.2e}; '
              f'last loss rel {np.abs(nll_l / nll_s - 1).max():.2e} / {np.abs(nll_f / nll_s - 1).max():.2e}; '
              f'search ms lag {1e3 * dt_l:.3f} fallback {1e3 * dt_f:.3f} streaming {1e3 * dt_s:.3f}', flush=True)
        if T <= 30_000:
            from oracle import c_oracle, eks_oracle as orc
            ks = list(range(min(K, 3)))
            ys = np.transpose(y_tk, (1, 0, 2)).astype(np.float64)
            Rc = rc.cpu().numpy()
            zero = np.zeros((1, 2, 2))

            def loss_and_grad(u):
                out = []
                for j, k in enumerate(ks):
                    sQ = np.exp(np.clip(u[j], -8, 8)) * arrs['Qs'][k]
                    L, g = c_oracle.nll_directional(ys[k], Rc[k], arrs['m0s'][k], arrs['S0s'][k], arrs['As'][k],
                                                    arrs['Cs'][k], sQ, zero, sQ[None])
                    out.append((L, g[0]))
                return np.array([o[0] for o in out]), np.array([o[1] for o in out])
            u_o, _, it_o = orc.adam_optimize_s(loss_and_grad, u0[ks], tol=1e-2, safety_cap=300)
            print(f'   oracle ({len(ks)} keypoints): iterations {it_o} vs {st_l[ks, 4].astype(int)}; '
                  f'max|dlog s| lag {np.abs(np.log(s_l[ks]) - np.clip(u_o, -8, 8)).max():.2e} '
                  f'fallback {np.abs(np.log(s_f[ks]) - np.clip(u_o, -8, 8)).max():.2e} '
                  f'streaming {np.abs(np.log(s_s[ks]) - np.clip(u_o, -8, 8)).max():.2e}', flush=True)


if __name__ == '__main__':
    main()
