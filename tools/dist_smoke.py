"""Two-rank run of eks_amd.distributed.smooth_sessions on the real GPU kernels (both ranks may share
one GPU; backend gloo, or nccl when every rank has its own GPU):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_smoke.py"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eks_amd import synth
from eks_amd.distributed import smooth_sessions
from oracle import eks_oracle as orc

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % max(1, torch.cuda.device_count()))
dist.init_process_group(os.environ.get('EKS_BENCH_BACKEND', 'gloo'), rank=rank, world_size=world)


def load(i):
    arrs = orc.singlecam_arrays(synth.singlecam_markers(1500, 3, seed=100 + i))
    return dict(ys=arrs['ys'], m0s=arrs['m0s'], S0s=arrs['S0s'], As=arrs['As'], Cs=arrs['Cs'], Qs=arrs['Qs'],
                ensemble_vars=arrs['ensemble_vars'])


local, s_all = smooth_sessions(load, 5, s_mode='grid')
assert sorted(local) == list(range(rank, 5, world)) and len(s_all) == 5
for i, (s, ms, Vs) in local.items():
    a = load(i)
    so, mo, Vo, _ = orc.run_kalman_smoother(a['ys'], a['m0s'], a['S0s'], a['As'], a['Cs'], a['Qs'], a['ensemble_vars'],
                                            s_mode='grid')
    np.testing.assert_array_equal(s, s_all[i])
    assert (np.abs(ms - mo) / np.abs(mo).max(axis=(1, 2), keepdims=True)).max() < 1e-5
print(f'rank {rank}: sessions {sorted(local)} ok, s[0]={np.round(s_all[0], 4)}', flush=True)
dist.destroy_process_group()
