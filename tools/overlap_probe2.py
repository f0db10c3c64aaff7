"""Probe: do the (VALU-bound) NLL grid kernels and the (HBM-bound) smoother kernels of two
independent problems overlap when enqueued on two streams?  Prints ms for each alone and for both
together.  Env knobs of the NLL geometry (EKS_NLL_*) apply."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import _lib, hip_ops, synth

T, K = 100_000, 256
dev = hip_ops.require_gpu()
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
cand = torch.exp(torch.linspace(-8.0, 8.0, 64, dtype=torch.float64, device=dev))


def problem(seed):
    y, var = synth.singlecam_observations_torch(T, K, seed=seed, device=dev)
    eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
    m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
    S0 = torch.diag_embed(y.double().var(dim=0, unbiased=False)).contiguous()
    return dict(y=y, var=var, eye=eye, m0=m0, S0=S0, rc=hip_ops.const_r(var, 1e-4),
                s=torch.full((K,), 10.0, dtype=torch.float64, device=dev),
                ms=torch.empty((T, K, 2), dtype=torch.float32, device=dev),
                Vs=torch.empty((T, K, 2, 2), dtype=torch.float32, device=dev))


A, B = problem(1), problem(2)
nll = lambda p: hip_ops.nll(p['y'], p['rc'], p['m0'], p['S0'], p['eye'], p['eye'], p['eye'], cand, flags=flags)
smooth = lambda p: hip_ops.smooth(p['y'], p['var'], p['m0'], p['S0'], p['eye'], p['eye'], p['eye'], p['s'],
                                  flags=flags, out=(p['ms'], p['Vs']))
cr = lambda p: hip_ops.const_r(p['var'], 1e-4)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def both(f, g, first='f'):
    def run():
        ev = torch.cuda.Event()
        ev.record()
        s1.wait_event(ev)
        s2.wait_event(ev)
        order = [(s1, f, A), (s2, g, B)] if first == 'f' else [(s2, g, B), (s1, f, A)]
        for st, fn, p in order:
            with torch.cuda.stream(st):
                fn(p)
        for st in (s1, s2):
            e = torch.cuda.Event()
            e.record(st)
            torch.cuda.current_stream().wait_event(e)
    return run


print('nll alone      %.3f' % timeit(lambda: nll(A)))
print('smooth alone   %.3f' % timeit(lambda: smooth(B)))
print('const_r alone  %.3f' % timeit(lambda: cr(B)))
print('nll || smooth  %.3f (nll enqueued first)' % timeit(both(nll, smooth, 'f')))
print('nll || smooth  %.3f (smooth enqueued first)' % timeit(both(nll, smooth, 'g')))
print('nll || const_r %.3f' % timeit(both(nll, cr, 'f')))
print('smooth||smooth %.3f' % timeit(both(smooth, smooth, 'f')))
