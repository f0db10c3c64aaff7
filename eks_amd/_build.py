"""Build libeks_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by
`python -m eks_amd._build`.  The .so is git-ignored but travels to the GPU box with the snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libeks_hip.so')
SOURCES = ['eks_api.hip', 'eks_diag.hip', 'eks_diag_nll.hip', 'eks_lag_adam.hip', 'eks_dense.hip', 'eks_dense_wave.hip', 'eks_dense_wide.hip', 'eks_loss.hip',
           'eks_loss_ar1.hip', 'eks_misc.hip', 'eks_multicam.hip', 'eks_profile.hip', 'eks_host.hip']
ARCH = 'gfx950'


def hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (ROCm toolchain required to build libeks_hip.so)')


def _digest(paths: list[str], flags: list[str]) -> str:
    """sha256 over the contents of `paths` and the compiler flags: what an object file was built FROM.  (Modification
    times do not survive the copy to the GPU box and say nothing about contents; round 3 skipped by mtime.)"""
    import hashlib
    h = hashlib.sha256(' '.join(flags).encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp(obj: str) -> str:
    try:
        with open(obj + '.sha256') as f:
            return f.read().strip()
    except OSError:
        return ''


# The SLP vectoriser packs adjacent scalar f32 operations of the candidate filters into v_pk_*_f32
# plus v_mov shuffles; on gfx950 a packed instruction issues over twice the cycles, so the moves
# are pure overhead (MI355X_MICROARCH.md: 'an anti-lever ... when the compiler SLP-packs').
# Measured on the NLL kernel (C3): 0.268 -> 0.245 ms; on the smoother's K1 / K3 the packed forms cost 90 / 340 extra
# VALU instructions per wave for nothing (step 0.609 -> 0.604 ms, same box, alternating runs).
PER_FILE_FLAGS = {'eks_diag_nll.hip': ['-fno-slp-vectorize'], 'eks_lag_adam.hip': ['-fno-slp-vectorize'], 'eks_diag.hip': ['-fno-slp-vectorize']}


def build(force: bool = False, verbose: bool = False, prove: bool = True) -> str:
    """Compile what is out of date (by CONTENT: every object carries the sha256 of its source, the headers and the
    flags it was built from) and link.  prove=True additionally compiles the smallest unit afresh into a temporary
    file on every call - so that a box that was shipped up-to-date objects still shows that its hipcc builds this
    tree - and the whole record goes to eks_amd/lib/BUILD_INFO.json.  EKS_FORCE_REBUILD=1 rebuilds everything."""
    import json
    import tempfile
    import time
    os.makedirs(LIBDIR, exist_ok=True)
    force = force or bool(os.environ.get('EKS_FORCE_REBUILD'))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hpp')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'eks_hip.h'))
    cc = hipcc()
    flags = ['-O3', '-std=c++17', f'--offload-arch={ARCH}', '-fPIC', '-ffp-contract=fast',
             '-Wno-unused-result', '-I', CSRC] + os.environ.get('EKS_EXTRA_HIPCC_FLAGS', '').split()
    objs = []
    jobs = []
    digests = {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace('.hip', '.o'))
        objs.append(o)
        fl = flags + PER_FILE_FLAGS.get(src, [])
        digests[o] = _digest([s] + headers, fl)
        if force or not os.path.exists(o) or _stamp(o) != digests[o]:
            jobs.append((o, [cc, *fl, '-c', s, '-o', o]))

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed: {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        return r

    def compile_one(job):
        o, cmd = job
        run(cmd)
        with open(o + '.sha256', 'w') as f:
            f.write(digests[o])

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=min(7, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    link_digest = _digest(objs, [])
    relinked = bool(jobs) or force or not os.path.exists(LIB) or _stamp(LIB) != link_digest
    if relinked:
        run([cc, '-shared', '-fPIC', f'--offload-arch={ARCH}', *objs, '-lpthread', '-o', LIB])
        with open(LIB + '.sha256', 'w') as f:
            f.write(link_digest)
    proof = None
    if prove:
        src = os.path.join(CSRC, 'eks_profile.hip')
        with tempfile.TemporaryDirectory() as tmp:
            t1 = time.time()
            run([cc, *flags, '-c', src, '-o', os.path.join(tmp, 'proof.o')])
            proof = dict(unit='eks_profile.hip', seconds=round(time.time() - t1, 2),
                         bytes=os.path.getsize(os.path.join(tmp, 'proof.o')))
    info = dict(hipcc=cc, arch=ARCH, compiled=[os.path.basename(o) for o, _ in jobs], relinked=relinked,
                reused=[os.path.basename(o) for o in objs if o not in {j[0] for j in jobs}],
                seconds=round(time.time() - t0, 2), toolchain_proof=proof, library_bytes=os.path.getsize(LIB),
                when=time.strftime('%Y-%m-%dT%H:%M:%S'))
    try:
        with open(os.path.join(LIBDIR, 'BUILD_INFO.json'), 'w') as f:
            json.dump(info, f, indent=1)
    except OSError:
        pass
    if verbose:
        print(json.dumps(info))
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True, prove='--no-proof' not in sys.argv))
