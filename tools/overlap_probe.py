"""Probe (round 2): how much of the C3 step can be hidden by running keypoint slices of the
session on several HIP streams - the VALU-bound NLL kernel of one slice beside the HBM-bound
median / summarize / replay kernels of another - and what the 256 MiB Infinity Cache gives when a
slice's y, var (2 x 205 MB / S) stay resident between its passes.  Uses the per-stage C ABI on
separately allocated slices; prints ms per whole step for each schedule."""
from __future__ import annotations

import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from eks_amd import _lib, hip_ops, synth  # noqa: E402


def main():
    T, K = 100_000, 256
    dev = hip_ops.require_gpu()
    flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
    y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
    cand = torch.exp(torch.linspace(-8.0, 8.0, 64, dtype=torch.float64, device=dev))

    def problem(k0, k1):
        ys, vs = y[:, k0:k1].contiguous(), var[:, k0:k1].contiguous()
        Kk = k1 - k0
        eye = torch.eye(2, dtype=torch.float64, device=dev).expand(Kk, 2, 2).contiguous()
        m0 = torch.zeros(Kk, 2, dtype=torch.float64, device=dev)
        S0 = torch.diag_embed(ys.double().var(dim=0, unbiased=False)).contiguous()
        ms = torch.empty((T, Kk, 2), dtype=torch.float32, device=dev)
        Vs = torch.empty((T, Kk, 2, 2), dtype=torch.float32, device=dev)
        return dict(y=ys, var=vs, eye=eye, m0=m0, S0=S0, ms=ms, Vs=Vs)

    def stage_a(p):                      # median + NLL grid + argmin
        rc = hip_ops.const_r(p['var'], 1e-4)
        nll = hip_ops.nll(p['y'], rc, p['m0'], p['S0'], p['eye'], p['eye'], p['eye'], cand, flags=flags)
        p['s'], _ = hip_ops.argmin_s(nll, cand)

    def stage_cr(p):
        p['rc'] = hip_ops.const_r(p['var'], 1e-4)

    def stage_nll(p):
        nll = hip_ops.nll(p['y'], p['rc'], p['m0'], p['S0'], p['eye'], p['eye'], p['eye'], cand, flags=flags)
        p['s'], _ = hip_ops.argmin_s(nll, cand)

    def stage_b(p):
        hip_ops.smooth(p['y'], p['var'], p['m0'], p['S0'], p['eye'], p['eye'], p['eye'], p['s'],
                       flags=flags, out=(p['ms'], p['Vs']))

    def timeit(fn, reps=20, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps

    full = problem(0, K)
    print(f'monolithic, one stream           : {timeit(lambda: (stage_a(full), stage_b(full))):.3f} ms', flush=True)
    print(f'  stage a (median+nll+argmin)    : {timeit(lambda: stage_a(full)):.3f} ms')
    print(f'  stage b (smooth)               : {timeit(lambda: stage_b(full)):.3f} ms', flush=True)

    for S in (2, 4, 8):
        ps = [problem(i * K // S, (i + 1) * K // S) for i in range(S)]

        def serial():
            for p in ps:
                stage_a(p)
                stage_b(p)
        print(f'S={S} slices, one stream           : {timeit(serial):.3f} ms', flush=True)
        print(f'  slice stage a                  : {timeit(lambda: stage_a(ps[0])):.3f} ms')
        print(f'  slice stage b                  : {timeit(lambda: stage_b(ps[0])):.3f} ms', flush=True)

        streams = [torch.cuda.Stream() for _ in range(S)]
        main_s = torch.cuda.current_stream()

        def free_for_all():
            ev0 = torch.cuda.Event()
            ev0.record(main_s)
            for p, st in zip(ps, streams):
                st.wait_event(ev0)
                with torch.cuda.stream(st):
                    stage_a(p)
                    stage_b(p)
            for st in streams:
                ev = torch.cuda.Event()
                ev.record(st)
                main_s.wait_event(ev)
        print(f'S={S} slices, {S} streams unordered  : {timeit(free_for_all):.3f} ms', flush=True)

        two = [torch.cuda.Stream(), torch.cuda.Stream()]

        def staggered():
            # stream A carries every slice's VALU-bound stage (median + nll), in slice order;
            # stream B carries every slice's smooth, each waiting for its slice's s
            ev0 = torch.cuda.Event()
            ev0.record(main_s)
            two[0].wait_event(ev0)
            two[1].wait_event(ev0)
            for p in ps:
                with torch.cuda.stream(two[0]):
                    stage_a(p)
                    ev = torch.cuda.Event()
                    ev.record(two[0])
                two[1].wait_event(ev)
                with torch.cuda.stream(two[1]):
                    stage_b(p)
            for st in two:
                ev = torch.cuda.Event()
                ev.record(st)
                main_s.wait_event(ev)
        print(f'S={S} slices, 2 streams (a | b)    : {timeit(staggered):.3f} ms', flush=True)

        three = [torch.cuda.Stream() for _ in range(3)]

        def staggered3():
            # median | nll | smooth on three streams
            ev0 = torch.cuda.Event()
            ev0.record(main_s)
            for st in three:
                st.wait_event(ev0)
            for p in ps:
                with torch.cuda.stream(three[0]):
                    stage_cr(p)
                    e1 = torch.cuda.Event()
                    e1.record(three[0])
                three[1].wait_event(e1)
                with torch.cuda.stream(three[1]):
                    stage_nll(p)
                    e2 = torch.cuda.Event()
                    e2.record(three[1])
                three[2].wait_event(e2)
                with torch.cuda.stream(three[2]):
                    stage_b(p)
            for st in three:
                ev = torch.cuda.Event()
                ev.record(st)
                main_s.wait_event(ev)
        print(f'S={S} slices, 3 streams (cr|nll|b) : {timeit(staggered3):.3f} ms', flush=True)


if __name__ == '__main__':
    main()
