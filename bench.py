#!/usr/bin/env python
"""bench.py - headline benchmark of the MI355X-native ensemble Kalman smoother.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): singlecam,
T = 100 000 frames x K = 256 keypoints, smoothing parameter chosen per keypoint on a 64-candidate
NLL grid (candidates exp(linspace(-8, 8, 64)), constant-R loss of eks/core.py:602/:640-650), then
the final fixed-s filter + RTS smoother with the full `ms (T,K,2)`, `Vs (T,K,2,2)` contract.
One "step" = one pass of that whole path over one session already resident in HBM:
    eks_const_r -> eks_nll_argmin (64 candidates: table + argmin) -> eks_smooth.
1 unit = one keypoint at one frame.  Inputs are synthetic (seeded, generated on device).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c3adam|c2|c5|c4|c4w|c4adam|pupil|ekf]
                  [--scaling weak|strong] [--gather-outputs]
N > 1 is launched by torch.distributed.run, one rank per GPU over RCCL.  Default ("weak"): every
rank smooths its own independent session of the workload's shape (sessions shard with no data-path
collective; `--workload c5` is BASELINE configs[4]: 128 sessions x 32 keypoints stacked along K per
GPU) and the per-keypoint s_finals are all-gathered inside the timed region.  `--scaling strong`:
ONE session of the workload's shape, its keypoints dealt to the GPUs (configs[2] at 32 keypoints
per GPU on 8 GPUs); `--gather-outputs` adds a second timed loop that also all-gathers ms / Vs to
every rank (compute + gather beside compute only).  Rank 0 prints ONE JSON line: `value` is the
whole-job rate with inputs resident in HBM; `roofline` is the step's DOMINANT (longest) kernel - on the
headline workload since round 5 the smoother's HBM-bound replay kernel (`frac` = `frac_hbm` = achieved / 8 TB/s, counter
traffic beside it), with the VALU-bound NLL grid kernel as `roofline_valu_kernel` (whichever of the two is longer on
the run is `roofline`; the other keeps its own key) and `whole_step_frac` for the step - `cpu_baseline` the C port of the reference
recursion on the host cores, `cpu_baseline_numpy` the NumPy restatement, `host_boundary` the same
step through host arrays (PCIe included; never `value`), `ranks` who ran where (rank, device, PCI address,
backend: under RCCL the ranks must sit on distinct GPUs or the run aborts).  A `parity_vs_cpu_port` figure
beyond BASELINE.json's 1e-5 bar makes the exit code non-zero (3) after the line is printed.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (T, K, n_cand)     singlecam (D = O = 2); n_cand = 0 -> fixed smoothing parameter s = 10
    'c3': (100_000, 256, 64),    # BASELINE.json configs[2]: the headline metric
    'c3adam': (100_000, 256, 0), # configs[2]'s session in the REFERENCE's default mode (smooth_param=None: Adam on
                                 # log s, eks/core.py:562-699) - one step = the whole search + the final smooth
    'c2': (10_000, 64, 0),       # configs[1]
    'c5': (50_000, 128 * 32, 0), # configs[4], one GPU's share: 128 sessions x 32 keypoints batched
    'c4': (50_000, 4, 0),        # configs[3]: mirrored multicam, 2 views x 4 paws, D = 3, O = 4 (dense path)
    'c4w': (50_000, 256, 0),     # the same model on a WIDE session (256 keypoints): can the dense kernels stream?
    'c4adam': (50_000, 4, 0),    # configs[3] in the reference's default mode: one loss + gradient evaluation per step
    'pupil': (100_000, 1, 0),    # IBL pupil AR(1) session, one optimiser iteration per step
    'ekf': (50_000, 16, 0),      # calibrated multicam, 4 cameras, D = 3, O = 8, fixed s
}
SMOOTH_BYTES_PER_UNIT = 40       # y 8 + var 8 in, ms 8 + Vs 16 out  
NLL_BYTES_PER_UNIT = 8           # y read once regardless of candidate count
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 peak, packed vector FMAs and f32-input MFMA alike
PARITY_BAR = 1e-5                # BASELINE.json: smoothed means / covariances (and the NLL table) within 1e-5 relative


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--warmup', type=int, default=3)
    p.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-kernel-events', action='store_true',
                   help='do not bracket kernels with HIP events inside the timed region')
    p.add_argument('--cpu-seconds', type=float, default=15.0)
    p.add_argument('--scaling', default='weak', choices=['weak', 'strong'],
                   help="N > 1: 'weak' = one session of the workload's shape per GPU (sessions shard, "
                        "BASELINE configs[4] style); 'strong' = ONE session, its keypoints dealt to the GPUs "
                        "(configs[2] at K / N keypoints per GPU)")
    p.add_argument('--gather-outputs', action='store_true',
                   help='strong scaling: also time a second loop that all-gathers ms / Vs to every rank')
    p.add_argument('--regions', type=int, default=5,
                   help='how many back-to-back timed regions of --steps steps each; ms_per_step / value are the '
                        'MEDIAN region (max over ranks per region), min / max are reported beside it')
    p.add_argument('--cpu-max-keypoints', type=int, default=0,
                   help='cap on the keypoints of the CPU baseline / parity sample (0: only --cpu-seconds bounds it)')
    p.add_argument('--lean', action='store_true',
                   help='skip the NumPy baseline and the NumPy-boundary timing (the extras\' child legs)')
    p.add_argument('--no-extras', action='store_true',
                   help='headline line only: skip the short legs of the other BASELINE configurations that the default '
                        'single-GPU run reports under `extras`')
    p.add_argument('--master-port', type=int, default=0,
                   help='self-launch (N > 1 without an outer torchrun): rendezvous port, 0 = pick a free one')
    return p.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no outer torchrun: start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a
    CHILD process (never an exec: a process that has touched the GPU must not be replaced, and this one
    has not even imported torch), pass rank 0's JSON line through on stdout and return the child's exit
    code.  Keypoints and sessions are independent (reference eks/core.py:223-224, :293), so the ranks
    only meet in the barriers and the terminal gather of s_finals."""
    import socket
    import subprocess
    port = args.master_port
    if not port:
        with socket.socket() as sock:
            sock.bind(('127.0.0.1', 0))
            port = sock.getsockname()[1]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; with the legacy mode RCCL's
    # peer-to-peer set-up (and any CUDA-tensor sharing across processes) fails with `hipIpcGetMemHandle: invalid
    # argument`.  The images export it already; it is pinned here so that a caller's scrubbed environment cannot
    # lose it for the ranks (a value the caller did set is kept).
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:                    # stream: the JSON line appears as soon as rank 0 prints it
        # only rank 0's result line belongs on stdout; library chatter (gloo's connection notes) goes to stderr
        out = sys.stdout if line.lstrip().startswith('{') else sys.stderr
        out.write(line)
        out.flush()
    return child.wait()


TRAFFIC_FILES = ('r06_traffic.json', 'r05_traffic.json')
# the sources whose kernels the traffic summary describes: a summary taken before any of them changed
# is STALE and is not reported (tests/test_abi_surface.py fails on a stale committed summary)
TRAFFIC_SOURCES = ('eks_diag.hip', 'eks_diag_lane.hpp', 'eks_math.hpp', 'eks_diag_nll.hip', 'eks_nll_lane.hpp', 'eks_nll_lag.hpp')


def kernel_sources_sha16():
    import hashlib
    h = hashlib.sha256()
    for name in TRAFFIC_SOURCES:
        with open(os.path.join(ROOT, 'eks_amd', 'csrc', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` on the C3 shape.  NOT measured in this run (hardware
    counters need rocprofv3's own passes): read from the committed summary of separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this same command
    (profiles/r02_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
    MI355X_MICROARCH.md).  Returns (bytes or None, source file or None)."""
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                doc = json.load(f)
        except Exception:
            continue
        if doc.get('kernel_sources_sha16') != kernel_sources_sha16():
            print(f'bench.py: profiles/{name} was measured on other kernel sources '
                  f'({doc.get("kernel_sources_sha16")} != {kernel_sources_sha16()}): roofline.traffic = null; '
                  'rerun tools/collect_evidence.sh + tools/make_profiles.py', file=sys.stderr, flush=True)
            return None, f'profiles/{name} is STALE (kernel sources changed since it was measured)'
        v = doc['hbm_bytes_per_launch'].get(kernel)
        if v is not None:
            return v, f'profiles/{name}'
    return None, None


def drain_profile(lib):
    buf = ctypes.create_string_buffer(1 << 16)
    ms = (ctypes.c_float * 4096)()
    n = lib.eks_profile_drain(buf, len(buf), ms, 4096)
    names = buf.raw.split(b'\0')[:n]
    out = {}
    for nm, t in zip(names, list(ms)[:n]):
        out.setdefault(nm.decode(), []).append(float(t))
    return out


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _physical_cores(cpus):
    """Distinct (socket, core) pairs among the logical CPUs this process may run on."""
    seen, cur = set(), {}
    try:
        with open('/proc/cpuinfo') as f:
            for line in f.read().split('\n') + ['']:
                if not line.strip():
                    if 'processor' in cur and int(cur['processor']) in cpus:
                        seen.add((cur.get('physical id', '0'), cur.get('core id', cur['processor'])))
                    cur = {}
                elif ':' in line:
                    k, v = line.split(':', 1)
                    cur[k.strip()] = v.strip()
    except OSError:
        pass
    return len(seen) or len(cpus)


def numpy_baseline(y_dev, var_dev, T, n_cand, budget_s=6.0):
    """SURVEY.md 8(d) baseline (i): the NumPy float64 restatement (oracle/eks_oracle.py), vectorised
    over keypoints and sequential in time - the structure of the reference's jit(vmap(scan)) - on
    one process, on a bounded sample: all keypoints, the first Tn frames, fixed s (one filter + RTS
    pass; the grid would multiply the time by 64 / 2.5)."""
    from oracle import eks_oracle as orc
    K = y_dev.shape[1]
    Tn = 1500
    y = np.transpose(y_dev[:Tn].cpu().numpy().astype(np.float64), (1, 0, 2)).copy()
    Rd = np.clip(np.transpose(var_dev[:Tn].cpu().numpy().astype(np.float64), (1, 0, 2)), 1e-12, None)
    eye = np.tile(np.eye(2), (K, 1, 1))
    t0 = time.perf_counter()
    orc.kalman_smoother(y, np.zeros((K, 2)), eye.copy(), eye, eye, eye, np.full(K, 10.0), Rd)
    dt = time.perf_counter() - t0
    return dict(value=Tn * K / dt, unit='frames*keypoints/s', cores=1, threads=1, kind='port',
                sample=f'first {Tn} frames x all {K} keypoints, fixed s, one filter + RTS pass in NumPy float64 '
                       f'(oracle/eks_oracle.py: vectorised over keypoints, sequential in time), {dt:.1f} s')


def cpu_baseline(y_dev, var_dev, T, n_cand, budget_s, max_keypoints=0):
    """Time the C twin of the oracle (general-matrix port of the reference recursion, OpenMP over
    keypoints) on a bounded sample of the same workload: the first Kc keypoints, all T frames."""
    from oracle import c_oracle, eks_oracle as orc
    cpus = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else set(range(os.cpu_count() or 1))
    cores = max(1, min(len(cpus), c_oracle.max_threads()))          # OpenMP threads used
    physical = _physical_cores(cpus)
    # ~0.3 us per frame per filter pass per core for the 2x2 general-matrix port (measured here)
    per_kp = T * 0.3e-6 * (max(n_cand, 0) + 2.5)
    Kc = int(max(cores, min(y_dev.shape[1], round(budget_s * cores / per_kp))))
    Kc = min(Kc, y_dev.shape[1])
    if max_keypoints > 0:
        Kc = min(Kc, max_keypoints)
    y = np.transpose(y_dev[:, :Kc].cpu().numpy().astype(np.float64), (1, 0, 2)).copy()
    var = np.transpose(var_dev[:, :Kc].cpu().numpy().astype(np.float64), (1, 0, 2)).copy()
    eye = np.tile(np.eye(2), (Kc, 1, 1))
    m0 = np.zeros((Kc, 2))
    S0 = eye * np.var(y, axis=1)[:, :, None]
    c_oracle.smooth(y[:1, :64], np.clip(var[:1, :64], 1e-12, None), m0[:1], S0[:1], eye[:1], eye[:1],
                    eye[:1], np.ones(1))                                   # build / warm the .so
    t0 = time.perf_counter()
    Rd = np.clip(var, 1e-12, None)
    if n_cand:
        Rc = orc.constant_R_from_timevarying(Rd)
        cand = np.exp(np.linspace(-8.0, 8.0, n_cand))
        nll = c_oracle.nll_grid(y, Rc, m0, S0, eye, eye, eye, cand, nthreads=cores)
        s = cand[np.argmin(nll, axis=1)]
    else:
        s = np.full(Kc, 10.0)
    ms_cpu, Vs_cpu, _ = c_oracle.smooth(y, Rd, m0, S0, eye, eye, eye, s, nthreads=cores)
    dt = time.perf_counter() - t0
    ref = dict(nll=nll if n_cand else None, ms=ms_cpu, Vs=Vs_cpu)
    # like for like: the same sample through the scalar-chain form of the diagonal model (what the GPU path
    # exploits and the reference does not: eksc_smooth_diag / eksc_nll_grid_diag, float64, product forms)
    t1 = time.perf_counter()
    if n_cand:
        nll_d = c_oracle.nll_grid_diag(y, Rc, m0, S0, eye, eye, eye, cand, nthreads=cores)
        s_d = cand[np.argmin(nll_d, axis=1)]
    else:
        s_d = s
    ms_d, Vd_d, _ = c_oracle.smooth_diag(y, Rd, m0, S0, eye, eye, eye, s_d, nthreads=cores)
    dt_d = time.perf_counter() - t1
    ref['diag'] = dict(value=T * Kc / dt_d, unit='frames*keypoints/s', cores=physical, threads=cores, kind='port',
                       sample=f'the same {Kc} keypoints x {T} frames through the SCALAR-CHAIN form of the diagonal model '
                              f'(oracle/eks_oracle.c: eksc_nll_grid_diag + eksc_smooth_diag, float64, OpenMP over chains) - '
                              f'the structure the GPU kernels exploit and the reference does not; {dt_d:.1f} s',
                       agrees_with_general_port=bool(np.array_equal(s_d, s) and
                                                     np.abs(ms_d - ms_cpu).max() <= 1e-9 * np.abs(ms_cpu).max()))
    return dict(value=T * Kc / dt, unit='frames*keypoints/s', cores=physical, threads=cores, kind='port',
                cpu_model=_cpu_model(),
                sample=f'first {Kc} of the keypoints x all {T} frames of the same workload '
                       f'({n_cand}-candidate NLL grid + smooth), float64 C port of the reference '
                       f'recursion (oracle/eks_oracle.c), OpenMP over keypoints, {dt:.1f} s'), s, ref


def bench_dense(args, T, K, dev, rank, world, lib, grad=False):
    """The general (D, O) kernels on the mirrored-multicam model (2 views, n_latent 3, fixed s): configs[3]
    itself (K = 4 paws: depth-bound by construction - two launches of float64 3x3 chains, no bytes to speak of)
    and, as `c4w`, the same model on a wide session (K = 256) to see whether the kernels can stream.  `grad`
    (`c4adam`): a step is one evaluation of the optimiser's loss and its gradient at one s per keypoint - what the
    reference's default mode (smooth_param=None, eks/core.py:562-699) repeats ~100 times per session."""
    import torch
    from eks_amd import _lib, hip_ops
    D, O = 3, 4
    g = torch.Generator(device=dev)
    g.manual_seed(4 + rank)
    lat = torch.cumsum(torch.randn(T, K, D, device=dev, generator=g) * 0.7, dim=0)
    C = torch.linalg.qr(torch.randn(K, O, D, device=dev, generator=g, dtype=torch.float64))[0].contiguous()
    var = (0.25 * (-torch.log(torch.rand(T, K, O, device=dev, generator=g).clamp_min(1e-12))
                   - torch.log(torch.rand(T, K, O, device=dev, generator=g).clamp_min(1e-12)))
           ).clamp_min(1e-3).float().contiguous()
    y = (torch.einsum('kod,tkd->tko', C.float(), lat)
         + torch.randn(T, K, O, device=dev, generator=g) * var.sqrt()).float().contiguous()
    L = torch.randn(K, D, D, device=dev, generator=g, dtype=torch.float64) * 0.3
    Q = L @ L.transpose(1, 2) + 0.2 * torch.eye(D, dtype=torch.float64, device=dev)
    Q = (Q / Q.abs().amax(dim=(1, 2), keepdim=True)).contiguous()
    eye = torch.eye(D, dtype=torch.float64, device=dev).expand(K, D, D).contiguous()
    m0 = torch.zeros(K, D, dtype=torch.float64, device=dev)
    S0 = (eye * 4.0).contiguous()
    s = torch.full((K,), 10.0, dtype=torch.float64, device=dev)
    ms = torch.empty((T, K, D), dtype=torch.float32, device=dev)
    Vs = torch.empty((T, K, D, D), dtype=torch.float32, device=dev)

    def step():
        hip_ops.smooth(y, var, m0, S0, eye, C, Q, s, out=(ms, Vs))

    if grad:
        rconst = hip_ops.const_r(var)
        s_col = s[:, None].contiguous()

        def step():                                   # noqa: F811  (Q above is positive definite)
            hip_ops.nll(y, rconst, m0, S0, eye, C, Q, s_col, per_keypoint=True, want_grad=True, flags=_lib.FLAG_Q_PD)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    region_dt = []
    for _ in range(max(1, args.regions)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        region_dt.append(time.perf_counter() - t0)
    dt = float(np.median(region_dt))
    lib.eks_profile_drain(None, 0, None, 0)
    lib.eks_profile_enable(1)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    lib.eks_profile_enable(0)
    prof = {k: float(np.mean(v)) for k, v in drain_profile(lib).items()}
    narrow = K <= 16
    longest = max(prof, key=prof.get) if prof else None
    bytes_per_unit = 4 * O * 2 + 4 * D + 4 * D * D           # y, var in; ms, Vs out: 80 B at D = 3, O = 4
    if grad:
        bytes_per_unit = 4 * O                               # y in; two doubles per keypoint out
    whole = bytes_per_unit * T * K / (dt / args.steps) / 1e9
    what = 'through one loss + gradient evaluation' if grad else 'smoothed'
    out = {'metric': f'frames*keypoints {what}/s, mirrored multicam {T // 1000}k x {K} (D={D}, O={O})',
           'value': args.steps * T * K / dt, 'unit': 'frames*keypoints/s', 'n_gpus': 1,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
           'regions': len(region_dt), 'ms_per_step_min': 1e3 * min(region_dt) / args.steps,
           'ms_per_step_max': 1e3 * max(region_dt) / args.steps,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
           'data': 'synthetic',
           'config': {'workload': (f'multicam linear T={T} x K={K} keypoints, D={D}, O={O}, ' +
                                   ('constant-R filter NLL and d/d log s at s=10 (the Adam loop body)' if grad
                                    else 'fixed s=10, full ms/Vs'))},
           'roofline': {'bound': 'hbm', 'kernel': 'whole step (' + ' + '.join(sorted(prof)) + ')', 'unit': 'GB/s',
                        'peak': HBM_PEAK_GBS, 'achieved': whole, 'frac': whole / HBM_PEAK_GBS, 'traffic': None,
                        'algorithmic_bytes_per_unit': bytes_per_unit, 'stage_avg_ms': prof,
                        'longest_stage': longest,
                        'note': ('depth-bound: 4 keypoints x 3 125 chunks = 196 workgroups of dependent float64 3x3 '
                                 'algebra; the bytes (16 MB) are irrelevant, the fraction is reported for the record'
                                 if narrow else
                                 'float64 3x3 algebra per frame (about 800 FMAs per keypoint-frame through summarize, '
                                 'scan and replay) against 80 B: the FP64 vector rate (78.6 TFLOP/s peak) bounds this '
                                 'shape at about the same level as HBM does')}}
    parity_failed = False
    if not grad and not args.no_cpu_baseline:
        # the timed path's outputs against the float64 C port (oracle/eks_oracle.c: eksc_smooth, the general-matrix
        # recursion) on a bounded sample of keypoints, every frame - never timed into `value`
        try:
            from oracle import c_oracle
            Kc = min(K, args.cpu_max_keypoints or 16)
            step()
            torch.cuda.synchronize()
            f64 = lambda t: t[:Kc].cpu().numpy().astype(np.float64)
            t0 = time.perf_counter()
            ms_o, Vs_o, _ = c_oracle.smooth(np.transpose(y[:, :Kc].cpu().numpy().astype(np.float64), (1, 0, 2)),
                                            np.clip(np.transpose(var[:, :Kc].cpu().numpy().astype(np.float64), (1, 0, 2)),
                                                    1e-12, None), f64(m0), f64(S0), f64(eye), f64(C), f64(Q), f64(s))
            dt_cpu = time.perf_counter() - t0
            ms_g = np.transpose(ms[:, :Kc].cpu().numpy().astype(np.float64), (1, 0, 2))
            Vs_g = np.transpose(Vs[:, :Kc].cpu().numpy().astype(np.float64), (1, 0, 2, 3))
            par = {'ms_max_rel_err': float((np.abs(ms_g - ms_o) / np.abs(ms_o).max(axis=(1, 2), keepdims=True)).max()),
                   'Vs_max_rel_err': float((np.abs(Vs_g - Vs_o) / np.abs(Vs_o).max(axis=(1, 2, 3), keepdims=True)).max()),
                   'keypoints_compared': Kc, 'bar': PARITY_BAR}
            par['ok'] = bool(par['ms_max_rel_err'] < PARITY_BAR and par['Vs_max_rel_err'] < PARITY_BAR)
            out['parity_vs_cpu_port'] = par
            out['cpu_baseline'] = {'value': T * Kc / dt_cpu, 'unit': 'frames*keypoints/s', 'cores': 1, 'kind': 'port',
                                   'sample': f'{Kc} keypoints x all {T} frames, fixed s, float64 C port of the general-matrix '
                                             f'recursion (oracle/eks_oracle.c), {dt_cpu:.1f} s'}
            parity_failed = not par['ok']
        except Exception as e:                          # the baseline must never sink the bench line
            out['cpu_baseline'] = {'value': None, 'unit': 'frames*keypoints/s', 'cores': 0, 'kind': 'port',
                                   'sample': f'failed: {e!r}'}
    print(json.dumps(out), flush=True)
    if parity_failed:
        print(f'bench.py: parity_vs_cpu_port beyond {PARITY_BAR:g}: {out["parity_vs_cpu_port"]}', file=sys.stderr)
        raise SystemExit(3)


def bench_c3adam(args, T, K, dev, lib, ranks_info):
    """BASELINE configs[2]'s session in the reference's DEFAULT mode (smooth_param=None): one step = the whole
    run_kalman_smoother call on device tensors - initial guesses (eks/core.py:233-236), eks_const_r, the Adam
    search on log s per keypoint (eks/core.py:562-699: every iteration is ONE launch of
    lag_adam_kernel after one pass for the lag sums - eks_amd/csrc/eks_lag_adam.hip), then the final fixed-s
    smooth with the full ms / Vs contract.  Fresh optimiser state every step."""
    import torch
    from eks_amd import hip_ops, synth
    from eks_amd.core import run_kalman_smoother
    y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
    eye = np.tile(np.eye(2), (K, 1, 1))
    m0 = np.zeros((K, 2))
    S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
    y_kt = y.transpose(0, 1)                      # the reference's (K,T,O) view of the frame-major buffer: zero copy
    last = {}

    def step():
        s, ms, Vs, info = run_kalman_smoother(y_kt, m0, S0, eye, eye, eye, var, smooth_param=None,
                                              return_device=True, return_info=True)
        last.update(s=s, ms=ms, Vs=Vs, info=info)

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    lib.eks_profile_drain(None, 0, None, 0)
    region_dt = []
    for _ in range(max(1, args.regions)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        region_dt.append(time.perf_counter() - t0)
    dt = float(np.median(region_dt))
    ms_step = 1e3 * dt / args.steps
    st = last['info']['state'].cpu().numpy()
    iters = st[:, 4]
    # per-launch durations of the loss kernel: one extra, untimed step with every launch bracketed by HIP events
    # (288 event records per step would perturb a launch-bound loop inside the timed region)
    prof = {}
    if not args.no_kernel_events:
        lib.eks_profile_enable(1)
        step()
        torch.cuda.synchronize()
        lib.eks_profile_enable(0)
        prof = drain_profile(lib)
    out = {'metric': 'frames*keypoints smoothed/s, singlecam 100k x 256 in the reference\'s default mode '
                     '(Adam search for s per keypoint + smooth)',
           'value': args.steps * T * K / dt, 'unit': 'frames*keypoints/s', 'n_gpus': 1, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': ms_step, 'regions': len(region_dt),
           'ms_per_step_min': 1e3 * min(region_dt) / args.steps, 'ms_per_step_max': 1e3 * max(region_dt) / args.steps,
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
           'config': {'workload': f'singlecam T={T} x K={K} keypoints (D=O=2), smooth_param=None: run_kalman_smoother on '
                                  'device tensors = initial guesses + eks_const_r + Adam on log s (lr 0.25, tol 1e-2, cap '
                                  '300; eks/core.py:562-699) + final smooth, full ms/Vs outputs',
                      'frames': T, 'keypoints': K, 'parallelism': 'single GPU',
                      'adam_iterations': {'min': float(iters.min()), 'mean': float(iters.mean()),
                                          'max': float(iters.max()),
                                          'iterations_enqueued': int(last['info']['launches']),
                                          'calls_enqueued': int(last['info'].get('calls', 0)),
                                          'all_stopped_by_rule': bool(np.all(st[:, 5] == 1.0))}},
           'ranks': ranks_info}
    if prof.get('lag_sums'):
        # round 6: the search is one streaming pass for 256 lag sums per chain (the step's longest kernel), a reduction
        # of the chunks' partial sums, and ONE launch of a workgroup per keypoint that reads no frame per iteration
        sums_ms, adam_ms = float(np.sum(prof['lag_sums'])), float(np.sum(prof.get('lag_adam', [0.0])))
        flops = 2.0 * 256 * T * 2 * K                                 # one FMA per lag, chain and frame
        other = {k: float(np.sum(v)) for k, v in prof.items() if k not in ('lag_sums', 'lag_adam')}
        out['roofline'] = {
            'bound': 'mfma', 'kernel': 'lag_sums_kernel', 'unit': 'TFLOP/s', 'peak': FP32_VECTOR_PEAK_TFLOPS,
            'achieved': flops / (sums_ms * 1e-3) / 1e12, 'frac': flops / (sums_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
            'traffic': None, 'kernel_avg_ms': sums_ms,
            'algorithmic_flops_per_launch': flops, 'algorithmic_bytes_per_launch': T * 2 * K * 4,
            'hbm_view': {'achieved_GBps': T * 2 * K * 4 / (sums_ms * 1e-3) / 1e9,
                         'frac': T * 2 * K * 4 / (sums_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            'search_kernel': 'lag_adam_kernel', 'search_kernel_ms': adam_ms,
            'search_us_per_iteration_of_the_longest_keypoint': 1e3 * adam_ms / float(iters.max()),
            'other_stage_ms': other,
            'kernel_avg_ms_source': 'HIP events on the launch stream, one untimed step after the timed regions',
            'note': 'y is read ONCE per search (4 B per chain-frame) by the pass that forms 256 lag sums per chain: a '
                    '16 x 16 x T matrix product per chain on v_mfma_f32_16x16x4_f32 (exact float32, 512 flops per '
                    'chain-frame; dense f32-input MFMA peak = the fp32 vector peak), so the pass is bound by the matrix '
                    'pipe, not by HBM; the iterations that follow touch no frame (closed-form head + lag polynomial, '
                    'float64 duals)'}
    if not args.no_cpu_baseline:
        try:
            out['cpu_baseline'] = cpu_baseline_adam(y, var, m0, S0, T, K, args.cpu_seconds, last)
            if out['cpu_baseline'].get('value'):
                out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        except Exception as e:
            out['cpu_baseline'] = {'value': None, 'unit': 'frames*keypoints/s', 'cores': 0, 'kind': 'port',
                                   'sample': f'failed: {e!r}'}
    print(json.dumps(out), flush=True)
    par = out.get('cpu_baseline', {}).get('parity')
    if par and not par.get('ok', True):
        print(f'bench.py: c3adam parity beyond the bars: {par}', file=sys.stderr)
        raise SystemExit(3)


def cpu_baseline_adam(y_dev, var_dev, m0, S0, T, K, budget_s, last):
    """The reference's default mode on the host cores: the oracle's optimiser (oracle/eks_oracle.py: adam_optimize_s,
    the loop of eks/core.py:652-681) fed by the C port's complex-step gradient, one keypoint per thread, all T
    frames, then the C port's smoother at the result - on a sample of keypoints sized to the budget.  Also the
    parity of the timed path against it: stopping iteration, log s, smoothed outputs."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import c_oracle, eks_oracle as orc
    from eks_amd.core import _initial_guesses_per_keypoint
    cpus = os.sched_getaffinity(0) if hasattr(os, 'sched_getaffinity') else set(range(os.cpu_count() or 1))
    threads = max(1, min(len(cpus), c_oracle.max_threads()))
    # ~1.2 us per frame and evaluation (complex-step filter of the 2x2 general-matrix port), ~90 evaluations
    per_kp = T * 1.2e-6 * 90
    Kc = int(min(K, max(1, threads * max(1, int(budget_s / per_kp)))))
    sel = np.linspace(0, K - 1, Kc).round().astype(int)
    sel = np.unique(sel)
    Kc = len(sel)
    y_s = np.transpose(y_dev[:, sel].cpu().numpy().astype(np.float64), (1, 0, 2)).copy()
    Rd = np.clip(np.transpose(var_dev[:, sel].cpu().numpy().astype(np.float64), (1, 0, 2)), 1e-12, None)
    guesses = _initial_guesses_per_keypoint(var_dev[:2000].cpu().numpy())[sel]
    eye = np.eye(2)
    zero = np.zeros((1, 2, 2))
    t0 = time.perf_counter()
    Rc = orc.constant_R_from_timevarying(Rd)
    u0 = np.array([np.float32(np.log(np.clip(g, 1e-6, 1e3))) for g in guesses], dtype=np.float64)

    def one(k, u):
        sQ = np.exp(u) * eye
        L, g = c_oracle.nll_directional(y_s[k], Rc[k], m0[sel[k]], S0[sel[k]], eye, eye, sQ, zero, sQ[None])
        return L, g[0]

    with ThreadPoolExecutor(max_workers=min(Kc, threads)) as pool:
        def loss_and_grad(u):
            res = list(pool.map(lambda k: one(k, u[k]), range(Kc)))
            return np.array([r[0] for r in res]), np.array([r[1] for r in res])
        u_o, last_o, it_o = orc.adam_optimize_s(loss_and_grad, u0)
    s_o = np.exp(np.clip(u_o, -8.0, 8.0))
    eyeK = np.tile(eye, (Kc, 1, 1))
    ms_o, Vs_o, _ = c_oracle.smooth(y_s, Rd, m0[sel], S0[sel], eyeK, eyeK, eyeK, s_o, nthreads=threads)
    dt = time.perf_counter() - t0
    st = last['info']['state'].cpu().numpy()
    s_g = np.asarray(last['s'])[sel]
    import torch
    sel_d = torch.as_tensor(sel, device=y_dev.device)
    ms_g = last['ms'].index_select(0, sel_d).cpu().numpy().astype(np.float64)
    Vs_g = last['Vs'].index_select(0, sel_d).cpu().numpy().astype(np.float64)
    # smoothed outputs are compared where both searches stopped at the same iteration (then s agrees to 1e-3 and
    # the outputs to the bar); a flipped stop test moves s by O(1) and is reported, not hidden
    same = st[sel, 4].astype(int) == it_o
    par = {'same_stopping_iteration': float(np.mean(same)),
           'max_abs_dlog_s': float(np.abs(np.log(s_g[same]) - np.log(s_o[same])).max()) if same.any() else None}
    if same.any():
        # at slightly different s (<= 1e-3 in log s) the outputs differ by that much times their sensitivity: the
        # strict comparison is at identical s (tests/test_gpu_configs.py); here the bar is 1e-3 * 1e-1
        par['ms_max_rel_err'] = float((np.abs(ms_g[same] - ms_o[same])
                                       / np.abs(ms_o[same]).max(axis=(1, 2), keepdims=True)).max())
    par['ok'] = bool(same.mean() >= 0.99 and (par['max_abs_dlog_s'] or 0.0) <= 2e-6
                     and par.get('ms_max_rel_err', 0.0) <= 1e-4)
    return dict(value=T * Kc / dt, unit='frames*keypoints/s', cores=_physical_cores(cpus), threads=threads, kind='port',
                cpu_model=_cpu_model(), parity=par,
                sample=f'{Kc} of the {K} keypoints (evenly spaced) x all {T} frames: Adam on log s with the reference\'s '
                       f'stop rule (oracle/eks_oracle.py: adam_optimize_s; {int(it_o.min())}-{int(it_o.max())} iterations) '
                       f'on the float64 C port\'s complex-step gradient (oracle/eks_oracle.c), one keypoint per thread, '
                       f'then the C port\'s smoother; {dt:.1f} s')


def bench_pupil(args, T, dev, lib):
    """IBL pupil path (reference eks/ibl_pupil_smoother.py): one chain, D=3, O=8, time-varying R.
    One step = one iteration of the two-parameter optimiser: eks_ar1_nll (loss + 2 forward
    sensitivities) -> eks_pupil_adam_step.  Depth-bound (a single chain offers only time
    parallelism): reported for the record, not the headline."""
    import torch
    from eks_amd import hip_ops, synth
    from eks_amd import ibl_pupil_smoother as ips
    ys, ev, m0, S0, lv = synth.pupil_observations(T, seed=1)
    P = ips._PupilProblem(ys, m0, S0, ips.PUPIL_C, ev, lv)
    loss = hip_ops.Ar1Loss(P.y, P.var, P.m0, P.S0, P.C, n_tan=2, positive_noise=bool(np.all(lv > 0)))   # as the driver does
    state = np.zeros((1, 9))
    state[0, 0:2] = np.log(np.array([0.99, 0.98]) / (1 - np.array([0.99, 0.98])))
    state[0, 6] = np.inf
    state = torch.as_tensor(state, device=dev)
    latent = torch.as_tensor(lv[None], device=dev)
    n_active = torch.zeros(1, dtype=torch.int32, device=dev)
    hip_ops.pupil_adam_step(loss, latent, state, n_active, 5e-3, 0.0, 1 << 30, init=True)

    def step():
        loss.evaluate()
        hip_ops.pupil_adam_step(loss, latent, state, n_active, 5e-3, 0.0, 1 << 30)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    lib.eks_profile_drain(None, 0, None, 0)
    lib.eks_profile_enable(0 if args.no_kernel_events else 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.eks_profile_enable(0)
    prof = {k: float(np.mean(v)) for k, v in drain_profile(lib).items()}
    # (dual-number form: one 'ar1_nll' scope; smoothing-distribution form: the wave kernels' two scopes)
    nll_ms = prof['ar1_nll'] if 'ar1_nll' in prof else (sum(prof.values()) if prof else None)
    hbm = None if not nll_ms else 64 * T / (nll_ms * 1e-3) / 1e9
    out = {'metric': 'frames x optimiser iterations / s, IBL pupil AR(1) session (D=3, O=8, time-varying R)',
           'value': args.steps * T / dt, 'unit': 'frames*iterations/s', 'n_gpus': 1, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': f'pupil AR(1) T={T} frames, one chain; step = loss + 2 sensitivities + Adam'},
           'roofline': {'bound': 'hbm', 'kernel': ' + '.join(sorted(prof)) if prof else 'eks_ar1_nll', 'unit': 'GB/s',
                        'peak': HBM_PEAK_GBS, 'achieved': hbm,
                        'frac': None if hbm is None else hbm / HBM_PEAK_GBS, 'traffic': None,
                        'stage_avg_ms': prof,
                        'note': 'depth-bound, not HBM-bound: 64 B/frame (y, var) is nothing; the time is a chunk of '
                                'frames + log2(T/8) element compositions of dependent float64 arithmetic per launch'}}
    if not args.no_cpu_baseline:
        from oracle import eks_oracle as orc
        Tc = min(T, 20_000)
        u = np.array([4.6, 3.9])
        t1 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t1 < min(args.cpu_seconds, 10.0):
            orc.pupil_nll_and_grad(u, ys[:Tc], m0, S0, orc.PUPIL_C, ev[:Tc], lv, use_c=True)
            reps += 1
        cdt = (time.perf_counter() - t1) / reps
        out['cpu_baseline'] = {'value': Tc / cdt, 'unit': 'frames*iterations/s', 'cores': 1, 'kind': 'port',
                               'sample': f'first {Tc} frames, loss + 2 directional derivatives by complex-step '
                                         f'through the C port of the reference recursion '
                                         f'(oracle/eks_oracle.c), {reps} evaluations'}
    print(json.dumps(out), flush=True)


def bench_ekf(args, T, K, dev, lib, V=4):
    """Calibrated multi-camera path (reference eks/core.py:188-190 with the pinhole h_fn of
    eks/multicam_smoother.py:814-898): one step = eks_ekf_smooth from a cold start (linearisation
    points = the prior mean) to the smoothed outputs - gated filter sweeps to the fixed point, then
    the smoothing sweep.  Small and latency-bound (K chains x T/32 chunks of float64 3x3 algebra):
    reported for the record, not the headline."""
    import torch
    from eks_amd import hip_ops, synth
    prob = synth.calibrated_multicam(T, K, V, seed=4)
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    y, var = t(prob['y_tko'], torch.float32), t(prob['var_tko'], torch.float32)
    m0, S0, A, Q, s = t(prob['m0s']), t(prob['S0s']), t(prob['As']), t(prob['Qs']), t(prob['s'])
    cams = t(prob['cams_packed'])
    cold = m0[:, None, :].expand(K, T, 3).contiguous()
    xlin = cold.clone()
    info = None

    def step():
        nonlocal info
        xlin.copy_(cold)
        info = hip_ops.ekf_smooth(y, var, None, m0, S0, A, Q, s, cams, xlin, max_sweeps=8, tol=1e-10)[3]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    lib.eks_profile_drain(None, 0, None, 0)
    lib.eks_profile_enable(0 if args.no_kernel_events else 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.eks_profile_enable(0)
    prof = {k: float(np.mean(v)) for k, v in drain_profile(lib).items()}
    O = 2 * V
    bytes_per_unit = 4 * O * 2 + 4 * 3 + 4 * 9            # y, var in; ms, Vs out (float32)
    sm = prof.get('ekf_smooth_sweep', float('nan'))
    out = {'metric': 'frames*keypoints smoothed/s, calibrated multicam extended Kalman smoother '
                     f'({V} cameras, D=3, O={O})',
           'value': args.steps * T * K / dt, 'unit': 'frames*keypoints/s', 'n_gpus': 1, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': f'calibrated multicam T={T} x K={K} keypoints, {V} pinhole cameras with '
                                  'distortion, fixed s, cold start (prior-mean linearisation)',
                      'filter_sweeps': float(info[0].item()), 'last_change': float(info[1].item())},
           'roofline': {'bound': 'hbm', 'kernel': 'dense_replay_kernel<3, true, PinholeObs> (smoothing sweep)',
                        'unit': 'GB/s', 'peak': HBM_PEAK_GBS,
                        'achieved': bytes_per_unit * T * K / (sm * 1e-3) / 1e9,
                        'frac': bytes_per_unit * T * K / (sm * 1e-3) / 1e9 / HBM_PEAK_GBS, 'traffic': None,
                        'stage_avg_ms': prof,
                        'note': 'latency-bound, not HBM-bound: K*T/32 lanes of dependent float64 3x3 '
                                'algebra do not fill the device at these sizes'}}
    if not args.no_cpu_baseline:
        from oracle import ekf_oracle as ek
        h = ek.combine_projections([ek.make_projection_fn(c['rot'], c['tvec'], c['K'], c['dist'])
                                    for c in prob['cams']])
        Tc = min(T, 4000)
        t1 = time.perf_counter()
        ek.eks_smoother(prob['y_tko'][:Tc, 0], np.maximum(prob['var_tko'][:Tc, 0], 1e-12), prob['m0s'][0],
                        prob['S0s'][0], prob['As'][0], prob['Qs'][0], prob['s'][0], h)
        cdt = time.perf_counter() - t1
        out['cpu_baseline'] = {'value': Tc / cdt, 'unit': 'frames*keypoints/s', 'cores': 1, 'kind': 'port',
                               'sample': f'first {Tc} frames of keypoint 0: sequential extended filter + RTS '
                                         'in NumPy float64 (oracle/ekf_oracle.py), complex-step Jacobians'}
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))      # before torch is imported: the parent never touches a GPU
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU')
    if (world > torch.cuda.device_count() and os.environ.get('EKS_BENCH_BACKEND', 'nccl') == 'nccl'):
        raise SystemExit(f'--gpus {world} but only {torch.cuda.device_count()} device(s) visible: RCCL needs one GPU '
                         'per rank (EKS_BENCH_BACKEND=gloo lets ranks share a GPU: a code-path test, not a '
                         'measurement)')
    from eks_amd import _lib, hip_ops, synth
    # the rank's own device FIRST (device_count does not create a context): nothing of this process touches GPU 0
    # unless it is its own
    n_dev = torch.cuda.device_count()
    local_rank = local_rank % max(1, n_dev)
    if n_dev:
        torch.cuda.set_device(local_rank)
    hip_ops.require_gpu()
    dev = torch.device('cuda', local_rank)
    backend = os.environ.get('EKS_BENCH_BACKEND', 'nccl')       # nccl == RCCL over xGMI on ROCm
    # EKS_BENCH_FORCE_DIST=1: build the process group and run every collective of the multi-rank path
    # with a world of one too (the 1-GPU box's way of exercising the RCCL branch: tests/test_gpu_distributed.py)
    use_dist = world > 1 or os.environ.get('EKS_BENCH_FORCE_DIST') == '1'
    if use_dist:
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:                                                    # test hook: ranks may share a GPU
            dist.init_process_group(backend, rank=rank, world_size=world)
    lib = _lib.load()
    # who runs where, gathered before anything is timed: under RCCL two ranks on one GPU abort the run here
    from eks_amd.distributed import check_distinct_devices, gather_rank_identities
    identities = gather_rank_identities()
    check_distinct_devices(identities)
    keep = ('rank', 'local_rank', 'host', 'backend', 'device', 'device_name', 'pci_bus_id', 'visible_devices')
    ranks_info = [{k: d.get(k) for k in keep} for d in identities]

    T, K, n_cand = WORKLOADS[args.workload]
    if args.workload == 'c3adam':
        return bench_c3adam(args, T, K, dev, lib, ranks_info)
    if args.workload in ('c4', 'c4w', 'c4adam'):
        return bench_dense(args, T, K, dev, rank, world, lib, grad=args.workload == 'c4adam')
    if args.workload == 'pupil':
        return bench_pupil(args, T, dev, lib)
    if args.workload == 'ekf':
        return bench_ekf(args, T, K, dev, lib)
    strong = args.scaling == 'strong' and use_dist
    if strong:
        # ONE session (the same on every rank: seed 3), its keypoints dealt to the ranks in
        # contiguous blocks (keypoints are independent, reference eks/core.py:223-224, :293)
        from eks_amd.distributed import keypoint_block_shard
        K_total = K
        own = [k for b in keypoint_block_shard([[k] for k in range(K_total)], world, rank) for k in [b]]
        y_all, var_all = synth.singlecam_observations_torch(T, K_total, seed=3, device=dev)
        idx = torch.as_tensor(own, device=dev)
        y, var = y_all.index_select(1, idx).contiguous(), var_all.index_select(1, idx).contiguous()
        del y_all, var_all
        K = len(own)
    else:
        # every rank owns an independent session of the same shape (seed = 3 + rank)
        K_total = K * world
        y, var = synth.singlecam_observations_torch(T, K, seed=3 + rank, device=dev)
    eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
    m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
    S0 = torch.diag_embed(y.double().var(dim=0, unbiased=False)).contiguous()
    flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
    cand = torch.exp(torch.linspace(-8.0, 8.0, max(n_cand, 1), dtype=torch.float64, device=dev))
    s_fixed = torch.full((K,), 10.0, dtype=torch.float64, device=dev)
    ms = torch.empty((T, K, 2), dtype=torch.float32, device=dev)
    Vs = torch.empty((T, K, 2, 2), dtype=torch.float32, device=dev)
    Kmax = (K_total + world - 1) // world if strong else K
    s_pad = torch.zeros(Kmax, dtype=torch.float64, device=dev)
    gathered = [torch.empty(Kmax, dtype=torch.float64, device=dev) for _ in range(world)]
    gathered_host = [torch.empty(Kmax, dtype=torch.float64) for _ in range(world)]

    # fixed s (configs[1]): a step is 16 us of GPU time - the call is prepared once (argument checks, workspace,
    # ctypes argument list), so that the host's enqueue cost (19 us per plain hip_ops.smooth call, 12.5 us
    # prepared; tools/c2_host_time.py) is not what the line reports
    prepared = None if n_cand else hip_ops.PreparedSmooth(y, var, m0, S0, eye, eye, eye, s_fixed, flags=flags,
                                                          out=(ms, Vs))

    def step(gather_outputs=False):
        if prepared is not None:
            s = s_fixed
            prepared()
        elif n_cand:
            rc = hip_ops.const_r(var, 1e-4)
            nll, s, _ = hip_ops.nll_argmin(y, rc, m0, S0, eye, eye, eye, cand, flags=flags)
        if prepared is None:
            hip_ops.smooth(y, var, m0, S0, eye, eye, eye, s, flags=flags, out=(ms, Vs))
        if use_dist:
            # the gather of s_finals (K float64 per rank) is asynchronous: it runs on the collective
            # stream while this rank's next step is being enqueued, and is waited for before the
            # timed region closes
            s_pad[:K] = s
            if backend == 'nccl':
                pending.append(dist.all_gather(gathered, s_pad, async_op=True))
            else:
                pending.append(dist.all_gather(gathered_host, s_pad.cpu(), async_op=True))
            if gather_outputs and backend == 'nccl':
                # every rank ends with the whole session's ms / Vs (24 B per keypoint-frame over xGMI)
                pending.append(dist.all_gather_into_tensor(ms_all, ms, async_op=True))
                pending.append(dist.all_gather_into_tensor(Vs_all, Vs, async_op=True))
        return s

    pending = []

    def sync():
        while pending:
            pending.pop(0).wait()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    lib.eks_profile_drain(None, 0, None, 0)
    events_on = not args.no_kernel_events
    # inside the timed region only the roofline kernel (diag_replay) is bracketed by HIP events:
    # two event records per step; the stage breakdown comes from a short untimed pass afterwards
    # (not for the 16 us steps of a fixed-s small session: two event records per step would be a tenth of it -
    # there the kernel time comes from the untimed pass below)
    region_events = events_on and not (prepared is not None and T * K < 5_000_000)
    lib.eks_profile_enable(2 if region_events else 0)
    # `--regions` back-to-back timed regions of exactly `--steps` steps, each closed by the barrier +
    # synchronize of sync(); the headline is the MEDIAN region (box-to-box and run-to-run spread of a
    # 12 ms region is larger than most kernel changes), min / max beside it
    region_dt = []
    for _ in range(max(1, args.regions)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            s_last = step()
        sync()
        region_dt.append(time.perf_counter() - t0)
    lib.eks_profile_enable(0)
    prof = drain_profile(lib) if region_events else {}
    stages = {}
    if events_on and rank == 0:
        lib.eks_profile_enable(1)
        for _ in range(5):
            step()
        sync()
        lib.eks_profile_enable(0)
        raw = drain_profile(lib)
        stages = {k: float(np.mean(v)) for k, v in raw.items()}
        if not region_events:
            prof = raw
    elif events_on and use_dist:
        for _ in range(5):          # keep the ranks' collectives matched
            step()
        sync()
    dt_gather = None
    if strong and args.gather_outputs and backend == 'nccl' and K * world == K_total:
        ms_all = torch.empty((world, T, K, 2), dtype=torch.float32, device=dev)
        Vs_all = torch.empty((world, T, K, 2, 2), dtype=torch.float32, device=dev)
        step(True)
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        sync()
        dt_gather = time.perf_counter() - t1
    if use_dist:                                                  # every region: MAX over the ranks
        tt = region_dt + ([dt_gather] if dt_gather is not None else [])
        t = torch.tensor(tt, dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_dt = [float(v) for v in t[:len(region_dt)].tolist()]
        if dt_gather is not None:
            dt_gather = float(t[-1].item())
    dt = float(np.median(region_dt))

    units_per_step = T * K_total if strong else T * K * world      # all ranks together
    value = args.steps * units_per_step / dt
    shape = f'singlecam T={T} x K={K_total if strong else K} keypoints (D=O=2)'
    headline = args.workload == 'c3'
    metric = ('frames*keypoints smoothed/s + achieved HBM GB/s fraction, singlecam 100k x 256' if headline else
              f'frames*keypoints smoothed/s + achieved HBM GB/s fraction, {shape}'
              + (' [configs[1]: 10k x 64, fixed s]' if args.workload == 'c2' else
                 ' [configs[4]: one GPU\'s share, 128 sessions x 32 keypoints stacked along K]'
                 if args.workload == 'c5' else ''))
    out = {
        'metric': metric,
        'value': value, 'unit': 'frames*keypoints/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
        'scaling': 'strong' if strong else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'regions': len(region_dt), 'ms_per_step_min': 1e3 * min(region_dt) / args.steps,
        'ms_per_step_max': 1e3 * max(region_dt) / args.steps,
        'timing': f'median of {len(region_dt)} back-to-back regions of {args.steps} steps (max over ranks per region)',
        'config': {'workload': shape + ', '
                               + (f'{n_cand}-candidate NLL grid + ' if n_cand else 'fixed s=10, ')
                               + 'filter+RTS smooth, full ms/Vs outputs; '
                               + (f'one session, {K} keypoints per GPU, all-gather of s_finals' if strong
                                  else 'one session per GPU'),
                   'frames': T, 'keypoints': K_total if strong else K, 'candidates': n_cand,
                   'parallelism': (f'keypoints x{world}' if strong else f'sessions x{world}') if world > 1
                   else 'single GPU'},
    }
    if dt_gather is not None:
        out['ms_per_step_with_output_gather'] = 1e3 * dt_gather / args.steps
        out['value_with_output_gather'] = args.steps * units_per_step / dt_gather
    out['ranks'] = ranks_info
    parity_failed = False
    if rank == 0:
        if prof:
            local_units = T * K
            k3 = float(np.mean(prof['diag_replay']))
            avg = dict(stages)
            smooth_ms = sum(avg.get(k, 0.0) for k in ('diag_summarize', 'diag_scan', 'diag_replay'))
            achieved = SMOOTH_BYTES_PER_UNIT * local_units / (k3 * 1e-3) / 1e9
            traffic, traffic_src = measured_traffic('diag_replay_blk_kernel') if headline else (None, None)
            hbm_roof = {
                'bound': 'hbm', 'kernel': 'diag_replay_blk_kernel', 'achieved': achieved,
                'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                'traffic_source': (f'{traffic_src}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this '
                                   'command (not measured in this run)') if traffic_src else None,
                'algorithmic_bytes_per_launch': SMOOTH_BYTES_PER_UNIT * local_units,
                'kernel_avg_ms': k3, 'launches_timed': len(prof.get('diag_replay', [])),
                'kernel_avg_ms_source': ('HIP events on the launch stream, every launch inside the timed regions'
                                         if region_events else
                                         'HIP events on the launch stream, 5 untimed steps after the timed regions '
                                         '(no event records inside the regions of a 16 us step)'),
            }
            stage_info = {
                'stage_avg_ms': avg,
                'stage_avg_ms_source': 'HIP events, 5 untimed steps after the timed region',
                'smooth_stage_frac': SMOOTH_BYTES_PER_UNIT * local_units / (smooth_ms * 1e-3)
                                     / 1e9 / HBM_PEAK_GBS,
                'whole_step_frac': ((SMOOTH_BYTES_PER_UNIT + (NLL_BYTES_PER_UNIT if n_cand else 0))
                                    * local_units / (dt / args.steps) / 1e9 / HBM_PEAK_GBS),
                'whole_step_algorithmic_bytes': (SMOOTH_BYTES_PER_UNIT + (NLL_BYTES_PER_UNIT if n_cand else 0))
                                                * local_units,
            }
            nll_live = prof.get('diag_nll_summarize', [])
            if n_cand and nll_live:
                # the DOMINANT (longest) kernel of the step is not HBM-bound: 2 FMAs per frame, chain and
                # candidate on the vector ALUs (no MFMA: scalar recursions); y is read once for all candidates
                flops = 2.0 * 2.0 * local_units * 2 * n_cand
                t_nll = float(np.mean(nll_live)) * 1e-3
                nll_bytes = NLL_BYTES_PER_UNIT * local_units
                ntraffic, ntraffic_src = measured_traffic('diag_nll_grid_kernel') if headline else (None, None)
                valu_roof = {
                    'bound': 'valu', 'kernel': 'diag_nll_grid_kernel', 'unit': 'TFLOP/s',
                    'achieved': flops / t_nll / 1e12, 'peak': 157.3, 'frac': flops / t_nll / 1e12 / 157.3,
                    'frac_hbm': nll_bytes / t_nll / 1e9 / HBM_PEAK_GBS,
                    'kernel_avg_ms': float(np.mean(nll_live)), 'launches_timed': len(nll_live),
                    'kernel_avg_ms_source': 'HIP events on the launch stream, every launch inside the timed regions',
                    'algorithmic_flops_per_launch': flops,
                    'traffic': ntraffic,
                    'traffic_source': (f'{ntraffic_src}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                                       'this command (not measured in this run)') if ntraffic_src else None,
                    'hbm_view': {'algorithmic_bytes_per_launch': nll_bytes,
                                 'achieved_GBps': nll_bytes / t_nll / 1e9,
                                 'frac_of_hbm_peak': nll_bytes / t_nll / 1e9 / HBM_PEAK_GBS},
                    'share_of_step': float(np.mean(nll_live)) / (1e3 * dt / args.steps),
                    'note': 'ALGORITHMIC FMA flops (2 per frame x chain x candidate: what the loss of 64 candidates costs '
                            'by the recursion) against the fp32 vector peak (157.3 TFLOP/s = 2 cycles per v_fma_f32 at '
                            '2.4 GHz).  Round 5: the kernel ISSUES fewer - the candidates whose pole is below 0.345 share 16 '
                            'lag sums per chain and chunk (36 of 64 on this shape: ~81 FMAs per chain-frame instead of '
                            '128); it stays bound by the packed FMAs it issues at the 1.7-1.9 GHz the chip holds under them'}
                hbm_roof['frac_hbm'] = hbm_roof['frac']
                hbm_roof['share_of_step'] = k3 / (1e3 * dt / args.steps)
                # `roofline` is the step's DOMINANT (longest) kernel, whichever that is on this run
                if k3 >= float(np.mean(nll_live)):
                    out['roofline'] = {**hbm_roof, **stage_info}
                    out['roofline_valu_kernel'] = valu_roof
                else:
                    out['roofline'] = {**valu_roof, **stage_info}
                    out['roofline_hbm_kernel'] = hbm_roof
            else:
                hbm_roof['frac_hbm'] = hbm_roof['frac']
                out['roofline'] = {**hbm_roof, **stage_info}
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb, s_cpu, ref = cpu_baseline(y, var, T, n_cand, args.cpu_seconds, args.cpu_max_keypoints)
                out['cpu_baseline'] = cb
                out['gpu_over_cpu'] = value / cb['value']
                out['cpu_baseline_diag'] = ref.pop('diag')
                out['gpu_over_cpu_diag'] = value / out['cpu_baseline_diag']['value']
                if not args.lean:
                    try:
                        out['cpu_baseline_numpy'] = numpy_baseline(y, var, T, n_cand)
                    except Exception as e:
                        out['cpu_baseline_numpy'] = {'value': None, 'sample': f'failed: {e!r}'}
                if n_cand:
                    # compare grid INDICES (the candidate values differ by an ulp between
                    # torch.linspace and numpy.linspace)
                    cand_np = np.exp(np.linspace(-8.0, 8.0, n_cand))
                    s_gpu = s_last[:len(s_cpu)].cpu().numpy()
                    i_gpu = np.abs(np.log(s_gpu)[:, None] - np.log(cand_np)[None]).argmin(1)
                    i_cpu = np.abs(np.log(s_cpu)[:, None] - np.log(cand_np)[None]).argmin(1)
                    out['cpu_baseline']['argmin_index_agreement'] = float(np.mean(i_gpu == i_cpu))
                    same = i_gpu == i_cpu
                else:
                    same = np.ones(len(s_cpu), dtype=bool)
                # parity of the timed path's own outputs against the float64 port, on the sample
                # (untimed): relative to the keypoint's magnitude, BASELINE.json's 1e-5 bar
                Kc = len(s_cpu)
                par = {}
                if n_cand:
                    nll_gpu = hip_ops.nll(y, hip_ops.const_r(var, 1e-4), m0, S0, eye, eye, eye, cand,
                                          flags=flags)[:Kc].cpu().numpy()
                    par['nll_max_rel_err'] = float((np.abs(nll_gpu - ref['nll']) / np.abs(ref['nll'])).max())
                ms_g = np.transpose(ms[:, :Kc].cpu().numpy().astype(np.float64), (1, 0, 2))[same]
                Vs_g = np.transpose(Vs[:, :Kc].cpu().numpy().astype(np.float64), (1, 0, 2, 3))[same]
                ms_c, Vs_c = ref['ms'][same], ref['Vs'][same]
                par['ms_max_rel_err'] = float((np.abs(ms_g - ms_c) / np.abs(ms_c).max(axis=(1, 2), keepdims=True)).max())
                Vd_g, Vd_c = np.diagonal(Vs_g, axis1=2, axis2=3), np.diagonal(Vs_c, axis1=2, axis2=3)
                par['Vs_max_rel_err'] = float((np.abs(Vd_g - Vd_c) / Vd_c).max())
                par['keypoints_compared'] = int(same.sum())
                par['bar'] = PARITY_BAR
                par['ok'] = bool(all(par[k] < PARITY_BAR for k in par if k.endswith('_max_rel_err'))
                                 and par['keypoints_compared'] >= max(1, int(0.9 * Kc)))
                parity_failed = not par['ok']
                out['parity_vs_cpu_port'] = par
            except Exception as e:                      # the baseline must never sink the bench line
                out['cpu_baseline'] = {'value': None, 'unit': 'frames*keypoints/s', 'cores': 0,
                                       'kind': 'port', 'sample': f'failed: {e!r}'}
            if not args.lean:
                try:
                    out['host_boundary'] = host_boundary_rate(y, var, T, K, n_cand)
                except Exception as e:
                    out['host_boundary'] = {'value': None, 'note': f'failed: {e!r}'}
        if headline and world == 1 and not args.no_extras and not args.no_cpu_baseline:
            out['extras'] = collect_extras()
            parity_failed = parity_failed or any(isinstance(v, dict) and v.get('exit_code') == 3
                                                 for v in out['extras'].values())
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
    if parity_failed:                 # the line is out; a timed path that disagrees with the float64 port is an error
        bad = {k: v.get('parity_vs_cpu_port') for k, v in (out.get('extras') or {}).items()
               if isinstance(v, dict) and v.get('exit_code') == 3}
        print(f'bench.py: parity beyond the bars ({PARITY_BAR:g} on outputs): headline {out.get("parity_vs_cpu_port")}; '
              f'extras {bad}', file=sys.stderr)
        raise SystemExit(3)


EXTRA_LEGS = (      # (workload, extra arguments): short runs of the other BASELINE configurations, each a child process
    ('c3adam', ['--steps', '5', '--warmup', '2', '--regions', '3', '--cpu-seconds', '8']),
    ('c2', ['--steps', '50', '--warmup', '5', '--regions', '3', '--cpu-seconds', '3', '--lean']),
    ('c4', ['--steps', '50', '--warmup', '5', '--regions', '3', '--cpu-seconds', '3']),
    ('c5', ['--steps', '5', '--warmup', '2', '--regions', '3', '--cpu-seconds', '4', '--cpu-max-keypoints', '256', '--lean']),
)


def collect_extras(budget_s=75.0, per_leg_s=45.0):
    """The other BASELINE configurations beside the headline: `python bench.py --workload X` as a
    CHILD process per leg (this process keeps its device context; nothing is exec'ed over it), its JSON line cut down
    to what a reader of the headline line needs - time per step, rate, fraction of HBM, the parity block against the CPU
    port on a bounded sample.  Never part of `value`; a leg that fails or runs out of time says so and the line goes out."""
    import subprocess
    keep = ('ms_per_step', 'ms_per_step_min', 'ms_per_step_max', 'value', 'unit', 'steps', 'regions', 'gpu_over_cpu')
    extras, t_start = {}, time.perf_counter()
    for wl, extra in EXTRA_LEGS:
        left = budget_s - (time.perf_counter() - t_start)
        if left < 10.0:
            extras[wl] = {'skipped': f'the extras\' time budget ({budget_s:.0f} s) was used up by the legs before it'}
            continue
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), '--workload', wl, '--no-extras'] + extra,
                               capture_output=True, text=True, timeout=min(per_leg_s, left))
            line = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith('{')]
            if not line:
                extras[wl] = {'failed': f'exit code {r.returncode}', 'stderr_tail': r.stderr[-300:]}
                continue
            d = json.loads(line[-1])
            e = {k: d[k] for k in keep if k in d}
            e['workload'] = d.get('config', {}).get('workload')
            roof = d.get('roofline') or {}
            for k in ('whole_step_frac', 'frac', 'kernel', 'bound', 'kernel_avg_ms', 'search_kernel_ms'):
                if k in roof:
                    e['roofline_' + k] = roof[k]
            if 'adam_iterations' in d.get('config', {}):
                e['adam_iterations'] = d['config']['adam_iterations']
            par = d.get('parity_vs_cpu_port') or (d.get('cpu_baseline') or {}).get('parity')
            if par is not None:
                e['parity_vs_cpu_port'] = par
            cb = d.get('cpu_baseline') or {}
            e['cpu_baseline'] = {k: cb.get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample')}
            e['exit_code'] = r.returncode
            e['wall_s'] = round(time.perf_counter() - t0, 1)
            extras[wl] = e
        except subprocess.TimeoutExpired:
            extras[wl] = {'failed': f'no line within {min(per_leg_s, left):.0f} s'}
        except Exception as ex:                        # noqa: BLE001 - the headline line must go out
            extras[wl] = {'failed': repr(ex)}
    extras['note'] = ('short legs run after the headline was timed, one child process each (`--workload X --no-extras`); '
                      'never part of `value`; c3adam = BASELINE configs[2] in the reference\'s default mode (Adam on log s)')
    return extras


def host_boundary_rate(y, var, T, K, n_cand, reps=3):
    """The same step through the reference-shaped NumPy boundary (run_kalman_smoother on HOST arrays:
    upload of y, var, download of ms, Vs over PCIe included).  Never `value`: reported beside it."""
    import torch
    from eks_amd.core import run_kalman_smoother
    ys_h = np.ascontiguousarray(y.transpose(0, 1).cpu().numpy())              # (K,T,2)
    var_h = var.cpu().numpy()
    eye = np.tile(np.eye(2), (K, 1, 1))
    S0 = eye * ys_h.astype(np.float64).var(axis=1)[:, :, None]
    kw = dict(s_mode='grid', n_grid=n_cand) if n_cand else dict(smooth_param=10.0)
    best = 1e9
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run_kalman_smoother(ys_h, np.zeros((K, 2)), S0, eye, eye, eye, var_h, **kw)
        best = min(best, time.perf_counter() - t0)
        del out
    return {'value': T * K / best, 'unit': 'frames*keypoints/s', 'ms_per_step': 1e3 * best,
            'note': 'run_kalman_smoother on host NumPy arrays, best of 3: includes the H2D copy of y, var (16 B per '
                    'unit) and the D2H copy of ms, Vs (24 B per unit) over PCIe - transfer-bound, not the kernels'}


if __name__ == '__main__':
    main()
