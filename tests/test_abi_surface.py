"""CPU-only checks of the C-ABI boundary: libeks_hip.so builds/loads without a GPU and exports
every function include/eks_hip.h declares; the ctypes table covers exactly those functions."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'eks_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(eks_[a-z_0-9]+)\s*\(', txt)))


@pytest.fixture(scope='module')
def lib():
    from eks_amd import _build, _lib
    _build.build()                      # hipcc cross-compiles gfx950 on a CPU-only box
    return _lib.load()


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ('eks_smooth', 'eks_nll', 'eks_const_r', 'eks_argmin_s', 'eks_adam_step',
                 'eks_ensemble', 'eks_smooth_workspace_bytes'):
        assert must in names


def test_library_exports_every_declared_symbol(lib):
    for name in _declared():
        assert hasattr(lib, name), f'{name} declared in include/eks_hip.h but not exported'


def test_ctypes_table_matches_header(lib):
    from eks_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_no_gpu_calls_needed_for_metadata(lib):
    from eks_amd import _lib
    assert lib.eks_version().startswith(b'eks_hip')
    assert lib.eks_status_string(-3).startswith(b'unsupported')
    d = _lib.EksDims(256, 100000, 2, 2, _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC)
    ws = lib.eks_smooth_workspace_bytes(ctypes.byref(d))
    # 9 planes of [ceil(T/32)][N] floats
    assert ws >= 9 * 3125 * 512 * 4
    assert lib.eks_nll_workspace_bytes(ctypes.byref(d), 64) > 0
    bad = _lib.EksDims(0, 10, 2, 2, 0)
    assert lib.eks_smooth_workspace_bytes(ctypes.byref(bad)) == 0


def test_product_path_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from eks_amd import _lib, hip_ops
    with pytest.raises(_lib.EksHipError):
        hip_ops.require_gpu()
    with pytest.raises(_lib.EksHipError):
        hip_ops.smooth(torch.zeros(4, 1, 2), torch.ones(4, 1, 2), torch.zeros(1, 2, dtype=torch.float64),
                       torch.eye(2, dtype=torch.float64)[None], torch.eye(2, dtype=torch.float64)[None],
                       torch.eye(2, dtype=torch.float64)[None], torch.eye(2, dtype=torch.float64)[None],
                       torch.ones(1, dtype=torch.float64))


def test_product_package_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'eks_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M) or 'host_sim' in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, f'product code must not touch oracle/ or tests/host_sim: {bad}'


def test_committed_traffic_summary_matches_the_kernel_sources():
    """bench.py's roofline.traffic comes from profiles/r03_traffic.json (separate rocprofv3 --pmc passes); the
    summary records a hash of the smoother's kernel sources and must be re-measured when they change."""
    import json
    import bench
    path = os.path.join(ROOT, 'profiles', bench.TRAFFIC_FILES[0])
    if not os.path.exists(path):
        pytest.skip(f'{bench.TRAFFIC_FILES[0]} not measured yet this round (bench reports traffic = null)')
    with open(path) as f:
        doc = json.load(f)
    assert doc.get('kernel_sources_sha16') == bench.kernel_sources_sha16(), \
        'profiles traffic summary is stale: rerun tools/collect_evidence.sh + tools/make_profiles.py'
