"""Two-rank run of the eks_amd.distributed drivers on the real GPU kernels (both ranks may share
one GPU: backend gloo; every rank on its own GPU: backend nccl = RCCL):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_smoke.py
Checks every rank's shard against the oracle: session-at-a-time, sessions stacked along the
keypoint axis on the device, and one session sharded by keypoint blocks."""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import synth
from eks_amd.distributed import (smooth_sessions, smooth_sessions_batched,
                                 smooth_session_keypoint_sharded)
from oracle import eks_oracle as orc

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
ndev = torch.cuda.device_count()
torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % max(1, ndev))
backend = os.environ.get('EKS_BENCH_BACKEND') or ('nccl' if ndev >= world else 'gloo')
dist.init_process_group(backend, rank=rank, world_size=world)


def load(i, T=1500, K=3):
    arrs = orc.singlecam_arrays(synth.singlecam_markers(T, K, seed=100 + i))
    return dict(ys=arrs['ys'], m0s=arrs['m0s'], S0s=arrs['S0s'], As=arrs['As'], Cs=arrs['Cs'], Qs=arrs['Qs'],
                ensemble_vars=arrs['ensemble_vars'])


def rel(a, b):
    return (np.abs(a - b) / np.abs(b).max(axis=(1, 2), keepdims=True)).max()


local, s_all = smooth_sessions(load, 5, s_mode='grid')
assert sorted(local) == list(range(rank, 5, world)) and len(s_all) == 5
ref = {}
for i, (s, ms, Vs) in local.items():
    a = load(i)
    ref[i] = orc.run_kalman_smoother(a['ys'], a['m0s'], a['S0s'], a['As'], a['Cs'], a['Qs'], a['ensemble_vars'],
                                     s_mode='grid')
    np.testing.assert_array_equal(s, s_all[i])
    assert rel(ms, ref[i][1]) < 1e-5
print(f'rank {rank}: sessions {sorted(local)} ok, s[0]={np.round(s_all[0], 4)}', flush=True)

# the same sessions stacked along the keypoint axis: one launch sequence per rank, same results
batched, s_all_b = smooth_sessions_batched(load, 5, s_mode='grid', return_device=True)
assert sorted(batched) == sorted(local)
for i, (s, ms, Vs) in batched.items():
    assert ms.is_cuda and tuple(ms.shape) == (3, 1500, 2)
    np.testing.assert_array_equal(s, local[i][0])
    assert rel(ms.cpu().numpy().astype(np.float64), local[i][1].astype(np.float64)) < 1e-6   # same kernels
    np.testing.assert_array_equal(s_all_b[i], s_all[i])
print(f'rank {rank}: batched sessions {sorted(batched)} ok', flush=True)

# one session of 7 keypoints sharded by keypoint blocks (Adam mode, block [1, 4] shares one s)
big = load(77, T=1200, K=7)
blocks = [[0], [1, 4], [2], [3], [5], [6]]
s_k, own, ms_k, _ = smooth_session_keypoint_sharded(**big, blocks=blocks, safety_cap=5)
s_o, ms_o, _, _ = orc.run_kalman_smoother(big['ys'], big['m0s'], big['S0s'], big['As'], big['Cs'], big['Qs'],
                                          big['ensemble_vars'], blocks=blocks, safety_cap=5)
np.testing.assert_allclose(s_k, s_o, rtol=1e-3)
assert s_k[1] == s_k[4] and (1 in own) == (4 in own)
so2, ms_o2, _, _ = orc.run_kalman_smoother(big['ys'], big['m0s'], big['S0s'], big['As'], big['Cs'], big['Qs'],
                                           big['ensemble_vars'], smooth_param=list(s_k))
assert rel(ms_k, ms_o2[own]) < 1e-5
print(f'rank {rank}: keypoint shard {list(own)} ok ({backend})', flush=True)

# the reference's default mode on every rank at once: sessions long enough for the search from
# cached lag sums - a pass over y, then ONE launch of a workgroup per keypoint in which nothing waits for another
# workgroup - while the other rank's search shares the device; every session equals its own single-process result
from eks_amd.core import run_kalman_smoother


def load_long(i):
    return load(200 + i, T=3000, K=5)


adam_local, adam_s = smooth_sessions(load_long, 4)
assert sorted(adam_local) == list(range(rank, 4, world)) and len(adam_s) == 4
dist.barrier()                                          # (the single-process repeats run after both ranks' searches)
for i, (s, ms, Vs) in adam_local.items():
    a = load_long(i)
    s1, ms1, Vs1 = run_kalman_smoother(a['ys'], a['m0s'], a['S0s'], a['As'], a['Cs'], a['Qs'], a['ensemble_vars'])
    np.testing.assert_array_equal(s, s1)
    np.testing.assert_array_equal(ms, ms1)
    np.testing.assert_array_equal(s, adam_s[i])
print(f'rank {rank}: adam sessions {sorted(adam_local)} ok', flush=True)
dist.barrier()
dist.destroy_process_group()
