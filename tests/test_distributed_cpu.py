"""world_size-2 gloo tests of the N > 1 paths on CPU: sessions shard round-robin (one by one, or
stacked along the keypoint axis into batches), one large session shards by keypoint blocks; no
data-path collective, only s_finals are gathered (tensor collectives).  The GPU operator is
replaced by the float64 oracle here (tests may use the oracle); the sharding / stacking / gather
logic is the code under test."""
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


def test_session_stacking():
    rng = np.random.default_rng(0)
    sess = [dict(ys=rng.standard_normal((k, 7, 2)), ensemble_vars=rng.random((7, k, 2)),
                 m0s=np.zeros((k, 2)), S0s=np.tile(np.eye(2), (k, 1, 1)), As=np.tile(np.eye(2), (k, 1, 1)),
                 Cs=np.tile(np.eye(2), (k, 1, 1)), Qs=np.tile(np.eye(2), (k, 1, 1))) for k in (2, 3)]
    kw, offs, blocks = D.stack_sessions(sess, [None, [[0, 2], [1]]])
    assert list(offs) == [0, 2, 5] and kw['ys'].shape == (5, 7, 2) and kw['ensemble_vars'].shape == (7, 5, 2)
    assert blocks == [[0], [1], [2, 4], [3]]
    np.testing.assert_array_equal(kw['ys'][2:], sess[1]['ys'])
    np.testing.assert_array_equal(kw['ensemble_vars'][:, :2], sess[0]['ensemble_vars'])


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


def _session(i, K=2, T=60):
    rng = np.random.default_rng(100 + i)
    ev = rng.gamma(2.0, 0.3, (T, K, 2)) + 0.05
    y = np.cumsum(rng.standard_normal((K, T, 2)), axis=1)
    eye = np.tile(np.eye(2), (K, 1, 1))
    return dict(ys=y, m0s=np.zeros((K, 2)), S0s=eye, As=eye, Cs=eye, Qs=eye, ensemble_vars=ev)


def _cpu_smooth(**kw):
    from oracle import eks_oracle as orc
    s, ms, Vs, _ = orc.run_kalman_smoother(**kw)
    return s, ms, Vs


def _worker_batched(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    # 7 sessions; session 5 is longer (cannot be stacked with the others), session 6 has 3 keypoints
    load = lambda i: _session(i, K=3 if i == 6 else 2, T=80 if i == 5 else 60)
    calls = []

    def counting(**kw):
        calls.append(np.shape(kw['ys']))
        return _cpu_smooth(**kw)

    mine, all_s = D.smooth_sessions_batched(load, 7, smooth_fn=counting, max_batch_keypoints=5,
                                            session_blocks=lambda i: [[0, 1], [2]] if i == 6 else None,
                                            smooth_param=None, safety_cap=3)
    np.savez(os.path.join(out_dir, f'b{rank}.npz'), owned=np.array(sorted(mine)),
             all_s=np.concatenate(all_s), calls=np.array(calls),
             **{f'ms{i}': r[1] for i, r in mine.items()}, **{f's{i}': r[0] for i, r in mine.items()})
    # one large session, sharded by keypoint blocks (block [1, 4] shares one s and stays whole)
    big = _session(50, K=6, T=70)
    blocks = [[0], [1, 4], [2], [3], [5]]
    s_all, own, ms, Vs = D.smooth_session_keypoint_sharded(**big, blocks=blocks, smooth_fn=_cpu_smooth,
                                                           safety_cap=3)
    np.savez(os.path.join(out_dir, f'k{rank}.npz'), s_all=s_all, own=own, ms=ms)
    s_fix, own2, _, _ = D.smooth_session_keypoint_sharded(**big, smooth_fn=_cpu_smooth,
                                                          smooth_param=[1., 2., 3., 4., 5., 6.])
    assert list(s_fix) == [1., 2., 3., 4., 5., 6.]
    # per-rank loader: a rank is only ever handed the keypoints it owns (never the whole session)
    asked = []

    def load_kp(idx):
        asked.append(list(map(int, idx)))
        return dict(ys=big['ys'][idx], m0s=big['m0s'][idx], S0s=big['S0s'][idx], As=big['As'][idx],
                    Cs=big['Cs'][idx], Qs=big['Qs'][idx], ensemble_vars=big['ensemble_vars'][:, idx])

    s_ld, own_ld, ms_ld, _ = D.smooth_session_keypoint_sharded(load_keypoints=load_kp, n_keypoints=6, blocks=blocks,
                                                               smooth_fn=_cpu_smooth, safety_cap=3)
    assert asked == [list(own)] and 0 < len(own) < 6
    np.testing.assert_array_equal(s_ld, s_all)
    np.testing.assert_array_equal(ms_ld, ms)
    # a per-keypoint smooth_param list is per session: tiled over a batch of equal sessions, refused otherwise
    mine_l, _ = D.smooth_sessions_batched(lambda i: _session(i), 4, smooth_fn=_cpu_smooth, smooth_param=[3.0, 7.0])
    assert all(list(r[0]) == [3.0, 7.0] for r in mine_l.values()) and len(mine_l) == 2
    with pytest.raises(ValueError, match='smooth_param has 2 entries'):
        D.smooth_sessions_batched(lambda i: _session(i, K=3), 4, smooth_fn=_cpu_smooth, smooth_param=[3.0, 7.0])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_gloo_batched_sessions_and_keypoint_shards(tmp_path):
    import torch.multiprocessing as mp
    from oracle import eks_oracle as orc
    world = 2
    mp.spawn(_worker_batched, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    b0, b1 = (np.load(tmp_path / f'b{r}.npz') for r in range(world))
    assert list(b0['owned']) == [0, 2, 4, 6] and list(b1['owned']) == [1, 3, 5]
    np.testing.assert_array_equal(b0['all_s'], b1['all_s'])
    assert b0['all_s'].shape == (6 * 2 + 3,)
    # rank 0: sessions 0, 2 stacked (4 keypoints), then 4 + 6 (5 keypoints); rank 1: 1, 3 stacked, 5 alone
    assert [tuple(c) for c in b0['calls']] == [(4, 60, 2), (5, 60, 2)]
    assert [tuple(c) for c in b1['calls']] == [(4, 60, 2), (2, 80, 2)]
    # a batch equals its sessions smoothed one by one
    for r, b in ((0, b0), (1, b1)):
        for i in b['owned']:
            kw = _session(int(i), K=3 if i == 6 else 2, T=80 if i == 5 else 60)
            s, ms, _, _ = orc.run_kalman_smoother(**kw, blocks=[[0, 1], [2]] if i == 6 else None, safety_cap=3)
            np.testing.assert_allclose(b[f's{i}'], s, rtol=1e-12)
            np.testing.assert_allclose(b[f'ms{i}'], ms, rtol=1e-10, atol=1e-12)
    assert b0['s6'][0] == b0['s6'][1]                      # the block of session 6 shares one s
    k0, k1 = (np.load(tmp_path / f'k{r}.npz') for r in range(world))
    np.testing.assert_array_equal(k0['s_all'], k1['s_all'])
    assert sorted(list(k0['own']) + list(k1['own'])) == list(range(6))
    assert (1 in k0['own']) == (4 in k0['own'])             # the block stays on one rank
    big = _session(50, K=6, T=70)
    s, ms, _, _ = orc.run_kalman_smoother(**big, blocks=[[0], [1, 4], [2], [3], [5]], safety_cap=3)
    np.testing.assert_allclose(k0['s_all'], s, rtol=1e-12)
    assert k0['s_all'][1] == k0['s_all'][4]
    for k in (k0, k1):
        np.testing.assert_allclose(k['ms'], ms[k['own']], rtol=1e-10, atol=1e-12)


def _worker_ws8(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    K = 250                                              # ragged: 250 keypoints do not divide by 8
    big = _session(77, K=K, T=24)
    blocks = [[3, 4, 5, 200], [17, 249]] + [[k] for k in range(K) if k not in (3, 4, 5, 200, 17, 249)]
    s_all, own, ms, Vs = D.smooth_session_keypoint_sharded(**big, blocks=blocks, smooth_fn=_cpu_smooth, safety_cap=2)
    ids = D.gather_rank_identities()
    D.check_distinct_devices(ids)                        # gloo: ranks may share a device, nothing to check
    np.savez(os.path.join(out_dir, f'w{rank}.npz'), s_all=s_all, own=own, ms=ms,
             ranks=np.array([d['rank'] for d in ids]), pids=np.array([d['pid'] for d in ids]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_eight_ranks_gloo_ragged_keypoint_shards(tmp_path):
    """World size 8 (the node the scaling bench runs on), K = 250 keypoints (not a multiple of 8) with two
    multi-keypoint blocks: every keypoint is owned exactly once, blocks stay whole, every rank ends with the same
    s_finals, the shards equal the single-process result; the ranks' identities are gathered and distinct."""
    import torch.multiprocessing as mp
    from oracle import eks_oracle as orc
    world, K = 8, 250
    mp.spawn(_worker_ws8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f'w{r}.npz') for r in range(world)]
    owned = np.concatenate([r['own'] for r in res])
    assert sorted(owned.tolist()) == list(range(K))
    sizes = [len(r['own']) for r in res]
    assert max(sizes) - min(sizes) <= 3, sizes           # greedy balance: blocks of 4 and 2 among singletons
    for r in res:
        np.testing.assert_array_equal(r['s_all'], res[0]['s_all'])
        own = set(r['own'].tolist())
        assert ({3, 4, 5, 200} <= own) or not ({3, 4, 5, 200} & own)
        assert ({17, 249} <= own) or not ({17, 249} & own)
        assert list(r['ranks']) == list(range(world)) and len(set(r['pids'].tolist())) == world
    big = _session(77, K=K, T=24)
    blocks = [[3, 4, 5, 200], [17, 249]] + [[k] for k in range(K) if k not in (3, 4, 5, 200, 17, 249)]
    s, ms, _, _ = orc.run_kalman_smoother(**big, blocks=blocks, safety_cap=2)
    np.testing.assert_allclose(res[0]['s_all'], s, rtol=1e-12)
    assert s[3] == s[4] == s[5] == s[200] and s[17] == s[249]
    for r in res:
        np.testing.assert_allclose(r['ms'], ms[r['own']], rtol=1e-10, atol=1e-12)
