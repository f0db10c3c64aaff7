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
SOURCES = ['eks_api.hip', 'eks_diag.hip', 'eks_diag_nll.hip', 'eks_dense.hip', 'eks_dense_wave.hip', 'eks_dense_wide.hip', 'eks_loss.hip',
           'eks_loss_ar1.hip', 'eks_misc.hip', 'eks_multicam.hip', 'eks_profile.hip']
ARCH = 'gfx950'


def hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (ROCm toolchain required to build libeks_hip.so)')


def _newer(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


# The SLP vectoriser packs adjacent scalar f32 operations of the candidate filters into v_pk_*_f32
# plus v_mov shuffles; on gfx950 a packed instruction issues over twice the cycles, so the moves
# are pure overhead (MI355X_MICROARCH.md: 'an anti-lever ... when the compiler SLP-packs').
# Measured on the NLL kernel (C3): 0.268 -> 0.245 ms; on the smoother's K1 / K3 the packed forms cost 90 / 340 extra
# VALU instructions per wave for nothing (step 0.609 -> 0.604 ms, same box, alternating runs).
PER_FILE_FLAGS = {'eks_diag_nll.hip': ['-fno-slp-vectorize'], 'eks_diag.hip': ['-fno-slp-vectorize']}


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hpp')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'eks_hip.h'))
    cc = hipcc()
    flags = ['-O3', '-std=c++17', f'--offload-arch={ARCH}', '-fPIC', '-ffp-contract=fast',
             '-Wno-unused-result', '-I', CSRC] + os.environ.get('EKS_EXTRA_HIPCC_FLAGS', '').split()
    objs = []
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace('.hip', '.o'))
        objs.append(o)
        if force or not _newer(o, [s] + headers):
            jobs.append([cc, *flags, *PER_FILE_FLAGS.get(src, []), '-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed: {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        return r

    with ThreadPoolExecutor(max_workers=min(7, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not _newer(LIB, objs):
        run([cc, '-shared', '-fPIC', f'--offload-arch={ARCH}', *objs, '-o', LIB])
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
