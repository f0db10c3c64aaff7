"""pytest config: registers the ``gpu`` marker and makes the repo root importable."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests never run silently on a CPU-only box: they are skipped with a loud reason unless
    # selected on a machine that really has the device.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU visible (gpu-marked test)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def set_knob(monkeypatch):
    """set_knob('EKS_SMOOTH_UNFUSED', '1'): set one of the library's A/B variables for the rest of the
    test.  The library reads its EKS_* variables once (never per call), so the change is followed by
    eks_knobs_reload(), and again when the test's environment is restored."""
    from eks_amd import _lib
    lib = _lib.load()

    def _set(name, value):
        if value is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, value)
        lib.eks_knobs_reload()

    yield _set
    monkeypatch.undo()
    lib.eks_knobs_reload()
