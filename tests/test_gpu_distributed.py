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


def _torchrun(script_args, port, backend='gloo'):
    env = dict(os.environ, EKS_BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port)] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_two_ranks_smooth_their_session_shards():
    r = _torchrun([os.path.join('tools', 'dist_smoke.py')], 29611)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'rank 0: sessions [0, 2, 4] ok' in r.stdout and 'rank 1: sessions [1, 3] ok' in r.stdout
    assert 'rank 0: batched sessions [0, 2, 4] ok' in r.stdout and 'rank 1: batched sessions [1, 3] ok' in r.stdout
    assert r.stdout.count('keypoint shard') == 2


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
