"""Multi-GPU sharding of the smoother: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path partitions into independent units - sessions, and keypoints (or blocks of keypoints that
share one s) inside a session (reference eks/core.py:223-224, vmapped at :293/:684) - so ranks
exchange nothing while smoothing.  The only collective is a terminal all-gather of the
per-keypoint s_finals (K float64 per session: a few hundred bytes); smoothed means / covariances
stay on the rank that produced them (gathering them would be root-ingress bound over xGMI,
SURVEY.md 8e).
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np


def session_shard(n_sessions: int, world_size: int, rank: int) -> list[int]:
    """Round-robin session ids owned by `rank`."""
    return list(range(rank, n_sessions, world_size))


def keypoint_block_shard(blocks: Sequence[Sequence[int]], world_size: int, rank: int) -> list[int]:
    """Indices (into `blocks`) of the keypoint blocks owned by `rank`: greedy balance by block
    size, blocks never split (their members share one optimiser state)."""
    load = [0] * world_size
    owner = []
    for i in sorted(range(len(blocks)), key=lambda i: -len(blocks[i])):
        r = int(np.argmin(load))
        load[r] += len(blocks[i])
        owner.append((i, r))
    return sorted(i for i, r in owner if r == rank)


def gather_session_results(local: dict[int, np.ndarray], n_sessions: int, group=None) -> list[np.ndarray]:
    """All-gather {session id: s_finals} from every rank; returns the list ordered by session id."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    bucket = [None] * world
    dist.all_gather_object(bucket, {int(k): np.asarray(v, dtype=np.float64) for k, v in local.items()},
                           group=group)
    merged: dict[int, np.ndarray] = {}
    for part in bucket:
        for k, v in part.items():
            if k in merged:
                raise RuntimeError(f'session {k} was produced by two ranks')
            merged[k] = v
    missing = [i for i in range(n_sessions) if i not in merged]
    if missing:
        raise RuntimeError(f'sessions {missing} were produced by no rank')
    return [merged[i] for i in range(n_sessions)]


def smooth_sessions(load_session: Callable[[int], dict], n_sessions: int, smooth_fn: Callable | None = None,
                    group=None, **kalman_kwargs):
    """Smooth `n_sessions` independent sessions across the ranks of `group`.

    load_session(i) returns the keyword arguments of run_kalman_smoother for session i
    (ys, m0s, S0s, As, Cs, Qs, ensemble_vars).  Each rank processes its round-robin shard on its
    own GPU and keeps the smoothed arrays; returns (local results {i: (s_finals, ms, Vs)},
    s_finals of ALL sessions gathered on every rank)."""
    import torch.distributed as dist
    if smooth_fn is None:
        from .core import run_kalman_smoother as smooth_fn
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = {}
    for i in session_shard(n_sessions, world, rank):
        mine[i] = smooth_fn(**load_session(i), **kalman_kwargs)
    all_s = gather_session_results({i: r[0] for i, r in mine.items()}, n_sessions, group)
    return mine, all_s
