"""Adversarial inputs for the exact time-median (eks_const_r) against numpy.nanmedian, bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eks_amd import hip_ops
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = torch.device('cuda', 0)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    T = int(rng.choice([1025, 1026, 2047, 2048, 4097, 8191, 8193, 30011, 100000]))
    N = int(rng.choice([1, 3, 64, 65, 130]))
    v = np.empty((T, N, 1), np.float32)
    kinds = []
    for n in range(N):
        kind = int(rng.integers(0, 9)); kinds.append(kind)
        if kind == 0: col = rng.gamma(2.0, 0.3, T)
        elif kind == 1: col = np.full(T, rng.uniform(0.01, 3))                       # constant
        elif kind == 2: col = np.round(rng.gamma(2.0, 0.3, T), int(rng.integers(0, 3)))  # heavily quantised
        elif kind == 3: col = np.where(rng.random(T) < 0.5, rng.uniform(0, 1e-3, T), rng.uniform(5, 6, T))  # gap at the median
        elif kind == 4: col = np.where(rng.random(T) < rng.uniform(0.5, 0.999), np.nan, rng.gamma(2.0, 0.3, T))  # mostly NaN
        elif kind == 5: col = np.exp(rng.uniform(-40, 40, T))                        # 35 decades
        elif kind == 6: col = np.sort(rng.gamma(2.0, 0.3, T))                        # sorted in time (sample = quantiles)
        elif kind == 7: col = np.where(np.arange(T) % 24 == 0, 1e-6, rng.gamma(2.0, 0.3, T))  # aliases with the sample stride
        else: col = np.where(rng.random(T) < 0.3, 0.0, rng.gamma(2.0, 0.3, T))       # zeros below the clip
        v[:, n, 0] = col
    if rng.random() < 0.2:
        v[:, 0, 0] = np.nan                                                          # an all-NaN chain
    got = hip_ops.const_r(torch.as_tensor(v, device=dev), 1e-4).cpu().numpy()[:, 0]
    clipped = np.clip(v[:, :, 0].astype(np.float64), 1e-12, None)
    with np.errstate(all='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ref = np.maximum(np.nanmedian(clipped, axis=0), 1e-4)
    ok = np.array_equal(got, ref, equal_nan=True)
    bad += not ok
    print(f'case {case}: T={T} N={N} kinds={sorted(set(kinds))}: {"ok" if ok else "MISMATCH " + str(np.flatnonzero(~((got == ref) | (np.isnan(got) & np.isnan(ref)))))}', flush=True)
print('mismatches', bad)
