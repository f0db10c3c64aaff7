import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
T, K = 100000, 256
dev = torch.device('cuda', 0)
ys = np.random.default_rng(0).standard_normal((K, T, 2)).astype(np.float32)
ev = np.random.default_rng(1).random((T, K, 2)).astype(np.float32)
print('torch threads', torch.get_num_threads(), 'cpus', len(os.sched_getaffinity(0)))
def tm(f, n=3):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3, r
ev_t = torch.from_numpy(ev)
k0, k1 = 0, 32
print('strided gather (T,32,2) contiguous():', tm(lambda: ev_t[:, k0:k1].contiguous())[0], 'ms')
print('numpy ascontiguousarray strided:', tm(lambda: np.ascontiguousarray(ev[:, k0:k1]))[0], 'ms')
g = ev_t[:, k0:k1].contiguous()
print('pageable H2D 25 MB:', tm(lambda: g.to(dev))[0], 'ms')
print('pageable H2D y tile (as_tensor):', tm(lambda: torch.as_tensor(np.ascontiguousarray(ys[k0:k1]), device=dev))[0], 'ms')
print('pageable H2D whole var 205 MB:', tm(lambda: torch.as_tensor(ev, device=dev))[0], 'ms')
print('pinned alloc 205 MB:', tm(lambda: torch.empty((K, T, 2), dtype=torch.float32, pin_memory=True), 2)[0], 'ms')
h = torch.empty((K, T, 2), dtype=torch.float32, pin_memory=True)
d = torch.empty((32, T, 2), dtype=torch.float32, device=dev)
print('D2H 25 MB into pinned slab:', tm(lambda: h[k0:k1].copy_(d, non_blocking=True))[0], 'ms')
dd = torch.empty((K, T, 2), dtype=torch.float32, device=dev)
print('D2H 205 MB into pinned:', tm(lambda: h.copy_(dd, non_blocking=True))[0], 'ms')
s1 = torch.cuda.Stream(); 
def on_stream():
    with torch.cuda.stream(s1):
        x = torch.empty((T, 64), device=dev); return x
print('alloc on side stream:', tm(on_stream)[0], 'ms')
# device gather from whole var on device
evd = torch.as_tensor(ev, device=dev)
print('device slice contiguous:', tm(lambda: evd[:, k0:k1].contiguous())[0], 'ms')
