"""GPU: the sharded path end to end - two ranks (sharing the box's one GPU, gloo rendezvous) run
eks_amd.distributed.smooth_sessions on the real kernels and check their shards against the oracle
(tools/dist_smoke.py); and bench.py's N > 1 code path with the same arrangement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_args, port, backend='gloo', nproc=2, **extra_env):
    env = dict(os.environ, EKS_BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(port)] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_two_ranks_smooth_their_session_shards():
    r = _torchrun([os.path.join('tools', 'dist_smoke.py')], 29611)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'rank 0: sessions [0, 2, 4] ok' in r.stdout and 'rank 1: sessions [1, 3] ok' in r.stdout
    assert 'rank 0: batched sessions [0, 2, 4] ok' in r.stdout and 'rank 1: batched sessions [1, 3] ok' in r.stdout
    assert r.stdout.count('keypoint shard') == 2
    # (the reference's default mode on both ranks at once, sharing the GPU: tools/dist_smoke.py)
    assert 'rank 0: adam sessions [0, 2] ok' in r.stdout and 'rank 1: adam sessions [1, 3] ok' in r.stdout


def test_two_ranks_over_rccl_when_two_gpus_are_visible():
    """The same drivers with backend nccl (= RCCL over xGMI), one GPU per rank.  The 1-GPU test
    box skips this; the multi-GPU node runs it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL: one device per rank)')
    r = _torchrun([os.path.join('tools', 'dist_smoke.py')], 29613, backend='nccl')
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count('(nccl)') == 2


def test_bench_two_rank_code_path():
    r = _torchrun(['bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--workload', 'c2',
                   '--no-cpu-baseline'], 29612)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['scaling'] == 'weak' and out['value'] > 0
    # who ran where is on record: two ranks, here sharing this box's one GPU over gloo
    assert [d['rank'] for d in out['ranks']] == [0, 1] and all(d['backend'] == 'gloo' for d in out['ranks'])
    assert all(d['device_name'] and d['pci_bus_id'] for d in out['ranks'])


def test_bench_eight_ranks_self_launched():
    """The node shape of the scaling bench - `python bench.py --gpus 8` - as far as a 1-GPU box can take it: eight
    self-launched ranks share the GPU over gloo, every rank smooths its own configs[1] session, the s_finals
    all-gather and the MAX-reduced regions run with a world of eight, and the line records all eight ranks."""
    env = dict(os.environ, EKS_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '8', '--workload', 'c2', '--steps', '3', '--warmup', '1',
                        '--regions', '2', '--no-cpu-baseline'], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['n_gpus'] == 8 and out['scaling'] == 'weak' and out['value'] > 0
    assert [d['rank'] for d in out['ranks']] == list(range(8))
    assert out['config']['parallelism'] == 'sessions x8'


def test_rccl_distinct_device_check_refuses_two_ranks_on_one_gpu():
    """Under RCCL the bench aborts before timing when two ranks report the same GPU (the check the first real
    multi-GPU run relies on): unit-level, on fabricated identities plus this process's real one."""
    from eks_amd import distributed as D
    me = D.rank_identity()
    assert me['device_name'] and me['pci_bus_id'] and me['visible_devices'] >= 1
    a = dict(me, rank=0, backend='nccl')
    b = dict(me, rank=1, backend='nccl')
    with pytest.raises(RuntimeError, match='drive the same GPU'):
        D.check_distinct_devices([a, b])
    D.check_distinct_devices([a, dict(b, pci_bus_id='ffff:ff:1f', device=1)])
    D.check_distinct_devices([dict(a, backend='gloo'), dict(b, backend='gloo')])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` the way the round-end driver invokes it - NO outer torchrun: the parent
    (which never touches the GPU) starts torch.distributed.run as a child, the two ranks share this box's GPU
    (gloo rendezvous), rank 0's JSON line comes through the parent's stdout and the exit code is the child's."""
    env = dict(os.environ, EKS_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--workload', 'c2', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['value'] > 0
    assert out['regions'] >= 5 and out['ms_per_step_min'] <= out['ms_per_step'] <= out['ms_per_step_max']
    # strong scaling (ONE session, its keypoints dealt to the ranks) through the same self-launch
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--workload', 'c2', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline', '--scaling', 'strong'], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and out['config']['keypoints'] == 64


def test_bench_refuses_more_rccl_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('EKS_BENCH_BACKEND', None)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', str(n + 1), '--workload', 'c2', '--steps', '2',
                        '--no-cpu-baseline'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'RCCL needs one GPU per rank' in (r.stdout + r.stderr)


def test_rccl_branch_with_a_world_of_one():
    """What a 1-GPU box can exercise of the RCCL path: a process group of ONE rank on backend nccl.  The
    distributed drivers' tensor all-gathers (eks_amd.distributed.all_gather_ragged) and bench.py's
    collectives (async all_gather of s_finals, all_gather_into_tensor of ms / Vs, MAX all_reduce of the
    time, barriers) all run through RCCL on the device; results are checked as in the two-rank tests."""
    r = _torchrun([os.path.join('tools', 'dist_smoke.py')], 29614, backend='nccl', nproc=1)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'rank 0: sessions [0, 1, 2, 3, 4] ok' in r.stdout and '(nccl)' in r.stdout
    r = _torchrun(['bench.py', '--gpus', '1', '--steps', '3', '--warmup', '1', '--workload', 'c2',
                   '--no-cpu-baseline', '--scaling', 'strong', '--gather-outputs'], 29615, backend='nccl',
                  nproc=1, EKS_BENCH_FORCE_DIST='1')
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['n_gpus'] == 1 and out['scaling'] == 'strong' and out['value'] > 0
    assert out['ms_per_step_with_output_gather'] >= out['ms_per_step'] * 0.5
