"""The Adam search (one eks_adam_run call: lag sums + their reduction + the search kernel) across the session shapes of
profiles/r05_probes.txt section 11, beside the kernels that read y every iteration (EKS_ADAM_STREAM=1).
`python tools/lag_adam_shapes.py` prints one line per shape: ms per search (median of 7), iterations, same-s check."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import _lib, hip_ops, synth                                            # noqa: E402

SHAPES = ((100_000, 256), (100_000, 128), (100_000, 64), (100_000, 32), (30_000, 256), (30_000, 64), (30_000, 30),
          (10_000, 64), (300_000, 64))


def knob(name, value):
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = value
    _lib.load().eks_knobs_reload()


def main():
    dev = torch.device('cuda')
    f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)   # noqa: E731
    for T, K in SHAPES:
        y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
        eye = np.tile(np.eye(2), (K, 1, 1))
        S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
        params = [f64(np.zeros((K, 2))), f64(S0), f64(eye), f64(eye), f64(eye)]
        flags = hip_ops.model_flags(S0, eye, eye, eye)
        rc = hip_ops.const_r(var, 1e-4)
        offs = torch.arange(K + 1, dtype=torch.int32, device=dev)
        mem = torch.arange(K, dtype=torch.int32, device=dev)
        sd = torch.diff(var[:2000], dim=0).double().transpose(0, 1).reshape(K, -1).std(dim=1, unbiased=False)
        u0 = np.log(np.clip(sd.cpu().numpy(), 1e-6, 1e3))

        def once():
            st = np.zeros((K, 6))
            st[:, 0] = u0
            st[:, 3] = np.inf
            st = f64(st)
            s_kp = f64(np.exp(u0))
            loop = hip_ops.AdamLoop(y, rc, *params, offs, mem, st, s_kp, 0.25, -8.0, 8.0, 1e-2, 300, flags=flags)
            n = loop.stride()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            it = 0
            while it < 300:
                loop.run(min(n, 300 - it))
                it += n
                if int(loop.n_active.item()) == 0:
                    break
            torch.cuda.synchronize()
            return time.perf_counter() - t0, st.cpu().numpy(), s_kp.cpu().numpy()

        out = {}
        for mode in ('lag', 'stream'):
            knob('EKS_ADAM_STREAM', '1' if mode == 'stream' else None)
            for _ in range(2):
                once()
            runs = [once() for _ in range(7)]
            out[mode] = (float(np.median([r[0] for r in runs])), runs[-1][1], runs[-1][2])
        knob('EKS_ADAM_STREAM', None)
        (dl, stl, sl), (ds, sts, ss) = out['lag'], out['stream']
        print(f'{T:>7} x {K:<3}  from lag sums {1e3 * dl:6.3f} ms | a launch per iteration {1e3 * ds:6.3f} ms | iterations '
              f'{stl[:, 4].min():.0f}..{stl[:, 4].max():.0f} | same stopping iteration {int((stl[:, 4] == sts[:, 4]).sum())}/{K} | '
              f'max |d log s| {np.abs(np.log(sl) - np.log(ss)).max():.1e}', flush=True)
        del y, var


if __name__ == '__main__':
    main()
