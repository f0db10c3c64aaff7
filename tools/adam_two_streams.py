"""Feasibility probe: the Adam search of C3 as ONE loop over 256 keypoints against the same search split into
`parts` keypoint groups, each an independent eks_adam_run loop on its own stream (the tail of one group's iteration -
a tile's last block composing its groups, ~12 us with 4-8 blocks busy - overlaps the streaming of the others).
Fixed number of iterations (no stop rule interplay): 120 per group."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eks_amd import synth, hip_ops, _lib
dev = torch.device('cuda', 0)
T, K = 100_000, 256
y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
rc = hip_ops.const_r(var, 1e-4)
ITERS = 120


def make(ks):
    k = len(ks)
    idx = torch.as_tensor(ks, device=dev)
    yk = y.index_select(1, idx).contiguous()
    eye = torch.eye(2, dtype=torch.float64, device=dev).expand(k, 2, 2).contiguous()
    m0 = torch.zeros(k, 2, dtype=torch.float64, device=dev)
    S0 = torch.diag_embed(yk.double().var(dim=0, unbiased=False)).contiguous()
    state = torch.zeros((k, 6), dtype=torch.float64, device=dev)
    state[:, 0] = np.log(0.5)
    state[:, 3] = float('inf')
    offs = torch.arange(k + 1, dtype=torch.int32, device=dev)
    mem = torch.arange(k, dtype=torch.int32, device=dev)
    s_kp = torch.full((k,), 0.5, dtype=torch.float64, device=dev)
    # tol = 0: nobody stops, every iteration does the full work
    return hip_ops.AdamLoop(yk, rc.index_select(0, idx).contiguous(), m0, S0, eye, eye, eye, offs, mem, state, s_kp,
                            0.25, -8.0, 8.0, 0.0, 100000, flags=flags)


for parts in (1, 2, 4):
    groups = np.array_split(np.arange(K), parts)
    loops = [make(list(g)) for g in groups]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(ITERS // 8):
            for lp, st in zip(loops, streams):
                with torch.cuda.stream(st):
                    lp.run(8)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f'{parts} group(s) on {parts} stream(s): {1e3 * dt:.2f} ms for {ITERS} iterations = {1e6 * dt / ITERS:.1f} us per iteration of all 256 keypoints', flush=True)
