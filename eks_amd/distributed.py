"""Multi-GPU sharding of the smoother: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path partitions into independent units - sessions, and keypoints (or blocks of keypoints that
share one s) inside a session (reference eks/core.py:223-224, vmapped at :293/:684; the only
coupling is the summed loss of a block, :474-476) - so ranks exchange nothing while smoothing.
The only collective is a terminal TENSOR all-gather of the per-keypoint s_finals (K float64 per
session); smoothed means / covariances stay on the rank that produced them (gathering them would
be root-ingress bound over xGMI) unless a caller asks for them.

    smooth_sessions_batched      many sessions: each rank stacks its sessions along the keypoint
                                 axis ON THE DEVICE and issues one kernel sequence per batch
                                 (BASELINE.json configs[4]: 128 sessions x 32 keypoints per GPU
                                 become one 4096-keypoint launch)
    smooth_session_keypoint_sharded   one large session: keypoint blocks are dealt to the ranks,
                                 blocks kept whole (configs[2] at 32 keypoints per GPU)
    smooth_sessions              the simple session-at-a-time loop (kept for callers whose
                                 sessions differ in length)
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np


# ------------------------------------------------------------------------------------------
# partitioning (pure functions)
# ------------------------------------------------------------------------------------------
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


# ------------------------------------------------------------------------------------------
# collectives
# ------------------------------------------------------------------------------------------
def _dist():
    import torch.distributed as dist
    return dist


def _rank_world(group=None) -> tuple[int, int]:
    """(rank, world size); a process that never initialised torch.distributed is a world of one,
    so the drivers below also serve a single GPU."""
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _collective_device(group=None):
    """Tensors handed to the collective live where the backend wants them: the rank's GPU for
    nccl (RCCL), host memory for gloo."""
    import torch
    dist = _dist()
    if dist.get_backend(group) == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


def all_gather_ragged(values: np.ndarray, group=None) -> list[np.ndarray]:
    """All-gather one float64 vector per rank (lengths may differ) with two tensor collectives
    (lengths, then zero-padded payloads) - no pickling.  Returns the per-rank vectors."""
    import torch
    dist = _dist()
    values = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    world = _rank_world(group)[1]
    if world == 1:
        return [values.copy()]
    dev = _collective_device(group)
    n = torch.tensor([values.size], dtype=torch.int64, device=dev)
    lens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, n, group=group)
    lens = [int(x.item()) for x in lens]
    cap = max(max(lens), 1)
    buf = torch.zeros(cap, dtype=torch.float64, device=dev)
    buf[:values.size] = torch.as_tensor(values, device=dev)
    parts = [torch.empty(cap, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return [p[:n_r].cpu().numpy() for p, n_r in zip(parts, lens)]


def gather_session_results(local: dict[int, np.ndarray], n_sessions: int, group=None) -> list[np.ndarray]:
    """All-gather {session id: s_finals} from every rank (tensor collectives: a header vector of
    (id, K) pairs and the concatenated values); returns the list ordered by session id."""
    ids = sorted(int(i) for i in local)
    header = np.array([[i, np.asarray(local[i]).size] for i in ids], dtype=np.float64).reshape(-1)
    flat = np.concatenate([np.asarray(local[i], dtype=np.float64).reshape(-1) for i in ids]) \
        if ids else np.zeros(0)
    headers = all_gather_ragged(header, group)
    payloads = all_gather_ragged(flat, group)
    merged: dict[int, np.ndarray] = {}
    for h, p in zip(headers, payloads):
        off = 0
        for sid, k in h.reshape(-1, 2).astype(np.int64):
            if int(sid) in merged:
                raise RuntimeError(f'session {int(sid)} was produced by two ranks')
            merged[int(sid)] = p[off:off + int(k)].copy()
            off += int(k)
    missing = [i for i in range(n_sessions) if i not in merged]
    if missing:
        raise RuntimeError(f'sessions {missing} were produced by no rank')
    return [merged[i] for i in range(n_sessions)]


def rank_identity(group=None) -> dict:
    """Who this rank is and which device it drives: rank, local rank, host, backend, device ordinal, device
    name and PCI address (domain:bus:device) - what a multi-GPU run needs on record to show that N ranks
    really ran on N distinct GPUs.  No device context is created for a process without a visible GPU."""
    import os
    import socket
    import torch
    dist = _dist()
    rank, world = _rank_world(group)
    ident = dict(rank=rank, world_size=world, local_rank=int(os.environ.get('LOCAL_RANK', 0)),
                 host=socket.gethostname(), pid=os.getpid(),
                 backend=dist.get_backend(group) if (dist.is_available() and dist.is_initialized()) else None,
                 device=None, device_name=None, pci_bus_id=None)
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        prop = torch.cuda.get_device_properties(i)
        dom, bus, devid = (getattr(prop, a, None) for a in ('pci_domain_id', 'pci_bus_id', 'pci_device_id'))
        ident.update(device=i, device_name=prop.name,
                     pci_bus_id=None if bus is None else f'{int(dom or 0):04x}:{int(bus):02x}:{int(devid or 0):02x}',
                     device_uuid=str(getattr(prop, 'uuid', '')) or None,
                     visible_devices=torch.cuda.device_count())
    return ident


def gather_rank_identities(group=None) -> list[dict]:
    """rank_identity() of every rank, on every rank (one all_gather_object: a few hundred bytes per rank,
    outside any timed region)."""
    dist = _dist()
    me = rank_identity(group)
    world = me['world_size']
    if world == 1 or not (dist.is_available() and dist.is_initialized()):
        return [me]
    out = [None] * world
    dist.all_gather_object(out, me, group=group)
    return sorted(out, key=lambda d: d['rank'])


def check_distinct_devices(identities: list[dict]) -> None:
    """Under RCCL (`nccl`) every rank must drive its own GPU: raise if two ranks of one host report the same
    device (PCI address where the runtime exposes it, else the ordinal).  Under gloo ranks may share a GPU
    (the CPU / one-GPU test configurations) and nothing is checked."""
    if not identities or identities[0].get('backend') != 'nccl':
        return
    seen = {}
    for d in identities:
        ident = d.get('pci_bus_id') or d.get('device_uuid')
        if not ident:
            # neither a PCI address nor a uuid: ranks isolated with HIP_VISIBLE_DEVICES all report ordinal 0 - nothing
            # here positively identifies a duplicate, so nothing is refused
            continue
        key = (d['host'], ident)
        if key in seen:
            raise RuntimeError(f'ranks {seen[key]} and {d["rank"]} drive the same GPU {key}: RCCL needs one GPU per '
                               'rank (launch with one process per device; LOCAL_RANK selects it)')
        seen[key] = d['rank']


# ------------------------------------------------------------------------------------------
# drivers
# ------------------------------------------------------------------------------------------
def _default_smooth_fn():
    from .core import run_kalman_smoother
    return run_kalman_smoother


def smooth_sessions(load_session: Callable[[int], dict], n_sessions: int, smooth_fn: Callable | None = None,
                    group=None, **kalman_kwargs):
    """Smooth `n_sessions` independent sessions across the ranks of `group`, one session per call.

    load_session(i) returns the keyword arguments of run_kalman_smoother for session i
    (ys, m0s, S0s, As, Cs, Qs, ensemble_vars).  Each rank processes its round-robin shard on its
    own GPU and keeps the smoothed arrays; returns (local results {i: (s_finals, ms, Vs)},
    s_finals of ALL sessions gathered on every rank).  Sessions of equal length are smoothed
    far faster by `smooth_sessions_batched`."""
    if smooth_fn is None:
        smooth_fn = _default_smooth_fn()
    rank, world = _rank_world(group)
    mine = {}
    for i in session_shard(n_sessions, world, rank):
        mine[i] = smooth_fn(**load_session(i), **kalman_kwargs)
    all_s = gather_session_results({i: r[0] for i, r in mine.items()}, n_sessions, group)
    return mine, all_s


def _cat(parts, axis):
    """Concatenate NumPy arrays or torch tensors (device tensors stay on the device)."""
    if hasattr(parts[0], 'detach'):
        import torch
        return torch.cat(list(parts), dim=axis)
    return np.concatenate([np.asarray(p) for p in parts], axis=axis)


def stack_sessions(sessions: Sequence[dict], blocks: Sequence[Sequence[Sequence[int]] | None] | None = None,
                   device=None):
    """Stack run_kalman_smoother inputs of several sessions of equal (T, D, O) along the keypoint
    axis.  Returns (kwargs of ONE run_kalman_smoother call, keypoint offsets [n+1], blocks of the
    stacked problem - each session's blocks shifted by its offset, singletons where a session gave
    none).  With `device`, host arrays of ys / ensemble_vars are uploaded session by session (as
    float32) and concatenated ON THE DEVICE - no host-side copy of the whole batch."""
    offs = np.zeros(len(sessions) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([np.shape(s['m0s'])[0] for s in sessions])

    def big(a):
        if device is None or hasattr(a, 'detach'):
            return a
        import torch
        return torch.as_tensor(np.ascontiguousarray(a), device=device).to(torch.float32)

    kw = dict(ys=_cat([big(s['ys']) for s in sessions], 0),
              ensemble_vars=_cat([big(s['ensemble_vars']) for s in sessions], 1))
    for name in ('m0s', 'S0s', 'As', 'Cs', 'Qs'):
        kw[name] = np.concatenate([np.asarray(s[name], dtype=np.float64) for s in sessions], axis=0)
    stacked_blocks = []
    for j, s in enumerate(sessions):
        b = blocks[j] if blocks is not None and blocks[j] else [[k] for k in range(int(offs[j + 1] - offs[j]))]
        stacked_blocks += [[int(offs[j]) + int(k) for k in blk] for blk in b]
    return kw, offs, stacked_blocks


def smooth_sessions_batched(load_session: Callable[[int], dict], n_sessions: int,
                            smooth_fn: Callable | None = None, group=None,
                            max_batch_keypoints: int = 8192, session_blocks: Callable | None = None,
                            **kalman_kwargs):
    """Many independent sessions across the ranks of `group`, batched on the device.

    Each rank takes its round-robin shard, stacks CONSECUTIVE sessions of equal (T, D, O) along the
    keypoint axis (at most `max_batch_keypoints` keypoints per batch: 8192 x 50 000 frames = 16 GB
    of inputs and outputs at 40 B per keypoint-frame) and runs ONE run_kalman_smoother per batch -
    keypoints are independent (reference eks/core.py:293), so a batch equals the sessions smoothed
    one by one up to float32 rounding (the NLL kernels take wave-uniform regime decisions that see
    the neighbouring chains of a 64-chain tile).  `smooth_param` may be a scalar, a per-keypoint list
    (every session must then have that many keypoints), or a callable i -> per-session value(s).
    `session_blocks(i)` optionally returns session i's keypoint blocks.

    Returns (local {i: (s_finals, ms, Vs)} - views into the batch outputs, on the device when
    return_device=True is passed through - and s_finals of ALL sessions, gathered with tensor
    collectives, on every rank)."""
    device = None
    if smooth_fn is None:
        smooth_fn = _default_smooth_fn()
        from . import hip_ops
        device = hip_ops.require_gpu()         # host sessions go up one by one, stacked on the device
    rank, world = _rank_world(group)
    ids = session_shard(n_sessions, world, rank)
    sp = kalman_kwargs.pop('smooth_param', None)
    mine = {}

    def flush(batch):                       # batch: [(session id, its loaded arrays)]
        if not batch:
            return
        blk = [session_blocks(i) for i, _ in batch] if session_blocks is not None else None
        kw, offs, blocks = stack_sessions([sess for _, sess in batch], blk, device)
        if callable(sp):
            per = [np.broadcast_to(np.asarray(sp(i), dtype=float), (int(offs[j + 1] - offs[j]),))
                   for j, (i, _) in enumerate(batch)]
            kw['smooth_param'] = list(np.concatenate(per))
        elif sp is None or isinstance(sp, (int, float)):
            kw['smooth_param'] = sp
        else:
            # a per-keypoint list is per SESSION (run_kalman_smoother's meaning): tiled over the batch's
            # sessions, which must then all have that many keypoints
            spa = np.asarray(sp, dtype=float).reshape(-1)
            sizes = [int(offs[j + 1] - offs[j]) for j in range(len(batch))]
            if spa.size == 1:
                kw['smooth_param'] = float(spa[0])
            elif all(n == spa.size for n in sizes):
                kw['smooth_param'] = list(np.tile(spa, len(batch)))
            else:
                raise ValueError(f'smooth_param has {spa.size} entries but the sessions of this batch have '
                                 f'{sorted(set(sizes))} keypoints; pass a scalar or a callable i -> value(s)')
        s, ms, Vs = smooth_fn(**kw, blocks=blocks, **kalman_kwargs)
        for j, (i, _) in enumerate(batch):
            a, b = int(offs[j]), int(offs[j + 1])
            mine[i] = (np.asarray(s[a:b]), ms[a:b], Vs[a:b])

    # sessions are loaded one at a time and a batch is smoothed as soon as it is full (or the next
    # session has another shape), so at most one batch of inputs is alive beside its stacked copy
    batch, shape, n_kp = [], None, 0
    for i in ids:
        sess = load_session(i)
        K_i, T_i, O_i = np.shape(sess['ys'])
        sh = (T_i, np.shape(sess['m0s'])[1], O_i)
        if batch and (sh != shape or n_kp + K_i > max_batch_keypoints):
            flush(batch)
            batch, n_kp = [], 0
        shape = sh
        batch.append((i, sess))
        n_kp += K_i
    flush(batch)
    all_s = gather_session_results({i: r[0] for i, r in mine.items()}, n_sessions, group)
    return mine, all_s


def smooth_session_keypoint_sharded(ys=None, m0s=None, S0s=None, As=None, Cs=None, Qs=None, ensemble_vars=None,
                                    blocks: Sequence[Sequence[int]] | None = None,
                                    smooth_fn: Callable | None = None, group=None,
                                    smooth_param=None, load_keypoints: Callable | None = None,
                                    n_keypoints: int | None = None, **kalman_kwargs):
    """ONE large session across the ranks of `group`: keypoint blocks are dealt to the ranks
    (greedy balance, a block - whose members share one s, reference eks/core.py:474-476 - is never
    split), every rank smooths its keypoints on its own GPU with no exchange, and the per-keypoint
    s_finals are all-gathered (tensor collective).  Arguments as run_kalman_smoother.  Either every
    rank passes the full arrays, or - so that no rank ever holds the whole session (C3: 410 MB of
    inputs per rank otherwise) - `load_keypoints(idx) -> dict(ys (len(idx),T,O), m0s, S0s, As, Cs, Qs,
    ensemble_vars (T,len(idx),O))` with `n_keypoints` = K: a rank then loads only the keypoints it
    owns (idx sorted ascending).

    Returns (s_finals of all K keypoints, owned keypoint indices (sorted), ms, Vs of the owned
    keypoints in that order)."""
    if smooth_fn is None:
        smooth_fn = _default_smooth_fn()
    rank, world = _rank_world(group)
    if load_keypoints is not None:
        if n_keypoints is None:
            raise ValueError('load_keypoints needs n_keypoints (the session\'s K)')
        K = int(n_keypoints)
    else:
        K = np.shape(m0s)[0]
    if not blocks:
        blocks = [[k] for k in range(K)]
    flat = sorted(int(k) for b in blocks for k in b)
    if flat != list(range(K)):
        raise ValueError(f'blocks must partition the {K} keypoints')
    own_blocks = [list(map(int, blocks[i])) for i in keypoint_block_shard(blocks, world, rank)]
    own = sorted(k for b in own_blocks for k in b)
    local_of = {k: j for j, k in enumerate(own)}
    if own:
        idx = np.asarray(own)
        take = (lambda a, axis: a.index_select(axis, _index_like(a, idx)) if hasattr(a, 'detach')
                else np.take(np.asarray(a), idx, axis=axis))
        sp = smooth_param
        if sp is not None and not isinstance(sp, (int, float)):
            sp = np.broadcast_to(np.asarray(sp, dtype=float), (K,))[idx]
            sp = list(sp)
        if load_keypoints is not None:
            part = load_keypoints(idx)
            arrays = {n: part[n] for n in ('ys', 'm0s', 'S0s', 'As', 'Cs', 'Qs', 'ensemble_vars')}
            if np.shape(arrays['m0s'])[0] != len(own):
                raise ValueError(f'load_keypoints returned {np.shape(arrays["m0s"])[0]} keypoints for '
                                 f'{len(own)} requested')
        else:
            arrays = dict(ys=take(ys, 0), m0s=np.asarray(m0s)[idx], S0s=np.asarray(S0s)[idx],
                          As=np.asarray(As)[idx], Cs=np.asarray(Cs)[idx], Qs=np.asarray(Qs)[idx],
                          ensemble_vars=take(ensemble_vars, 1))
        s, ms, Vs = smooth_fn(**arrays, smooth_param=sp,
                              blocks=[[local_of[k] for k in b] for b in own_blocks], **kalman_kwargs)
        s = np.asarray(s, dtype=np.float64)
    else:
        s, ms, Vs = np.zeros(0), None, None
    keys = all_gather_ragged(np.asarray(own, dtype=np.float64), group)
    vals = all_gather_ragged(s, group)
    s_all = np.full(K, np.nan)
    for kk, vv in zip(keys, vals):
        s_all[kk.astype(np.int64)] = vv
    if np.isnan(s_all).any():
        raise RuntimeError('some keypoints were smoothed by no rank')
    return s_all, np.asarray(own, dtype=np.int64), ms, Vs


def _index_like(t, idx):
    import torch
    return torch.as_tensor(idx, dtype=torch.int64, device=t.device)
