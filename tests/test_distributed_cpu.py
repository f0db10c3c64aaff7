"""world_size-2 gloo test of the N > 1 path on CPU: sessions shard round-robin with no data-path
collective, only s_finals are gathered.  The GPU operator is replaced by the float64 oracle here
(tests may use the oracle); the sharding / gather logic is the code under test."""
import os
import socket

import numpy as np
import pytest

from eks_amd import distributed as D


def test_shards_partition_the_work():
    for n, w in ((10, 4), (3, 8), (1024, 8), (0, 2)):
        got = sorted(i for r in range(w) for i in D.session_shard(n, w, r))
        assert got == list(range(n))
    blocks = [[0, 1, 2], [3], [4, 5], [6], [7], [8, 9, 10, 11]]
    owned = [D.keypoint_block_shard(blocks, 3, r) for r in range(3)]
    assert sorted(i for o in owned for i in o) == list(range(len(blocks)))
    loads = [sum(len(blocks[i]) for i in o) for o in owned]
    assert max(loads) - min(loads) <= 2


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_sessions, out_dir):
    import torch.distributed as dist
    from oracle import eks_oracle as orc
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)

    def load(i):
        rng = np.random.default_rng(100 + i)
        K, T = 2, 60
        ev = rng.gamma(2.0, 0.3, (T, K, 2)) + 0.05
        y = np.cumsum(rng.standard_normal((K, T, 2)), axis=1)
        eye = np.tile(np.eye(2), (K, 1, 1))
        return dict(ys=y, m0s=np.zeros((K, 2)), S0s=eye, As=eye, Cs=eye, Qs=eye, ensemble_vars=ev)

    def cpu_smooth(**kw):
        s, ms, Vs, _ = orc.run_kalman_smoother(**kw)
        return s, ms, Vs

    mine, all_s = D.smooth_sessions(load, n_sessions, smooth_fn=cpu_smooth, smooth_param=None, safety_cap=3)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), owned=np.array(sorted(mine)), all_s=np.stack(all_s),
             **{f'ms{i}': r[1] for i, r in mine.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    n_sessions, world = 5, 2
    mp.spawn(_worker, args=(world, _free_port(), n_sessions, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / f'rank{r}.npz') for r in range(world))
    assert list(r0['owned']) == [0, 2, 4] and list(r1['owned']) == [1, 3]
    # every rank ends with the s_finals of all sessions, identical and in session order
    np.testing.assert_array_equal(r0['all_s'], r1['all_s'])
    assert r0['all_s'].shape == (n_sessions, 2) and np.all(r0['all_s'] > 0)
    # smoothed arrays stay sharded: a rank holds only what it produced
    assert 'ms1' not in r0.files and 'ms0' in r0.files
    # and equal what a single process computes for that session
    from oracle import eks_oracle as orc
    rng = np.random.default_rng(100 + 3)
    ev = rng.gamma(2.0, 0.3, (60, 2, 2)) + 0.05
    y = np.cumsum(rng.standard_normal((2, 60, 2)), axis=1)
    eye = np.tile(np.eye(2), (2, 1, 1))
    s, ms, _, _ = orc.run_kalman_smoother(y, np.zeros((2, 2)), eye, eye, eye, eye, ev, safety_cap=3)
    np.testing.assert_allclose(r1['all_s'][3], s, rtol=1e-12)
    np.testing.assert_allclose(r1['ms3'], ms, rtol=1e-12)
