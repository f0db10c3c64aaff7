"""GPU: short runs of the randomised parity sweeps under tools/ (each prints one line per case and
flags anything above the 1e-5 bar; the long runs are quoted in DESIGN.md section 2)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', script), *map(str, args)], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize('seed', [303, 202])
def test_scalar_chain_kernels_random_shapes(seed):
    out = _run('fuzz_parity.py', 14, seed)
    assert 'above 1e-5' not in out and "'med': 0.0" in out, out[-2000:]


def test_gradient_path_and_general_kernels_random_shapes():
    out = _run('fuzz_parity2.py', 18, 303)
    assert 'above 1e-5' not in out and 'worst' in out, out[-2000:]


def test_drivers_random_configurations():
    out = _run('fuzz_drivers.py', 7, 404)
    assert 'above 1e-5' not in out and 'worst' in out, out[-2000:]


def test_adam_mode_random_blocks_and_crops():
    out = _run('fuzz_adam.py', 4, 505)
    assert '<-- check' not in out and 'worst' in out, out[-2000:]


def test_pupil_driver_random_sessions():
    out = _run('fuzz_pupil.py', 5, 606)
    assert 'above 1e-5' not in out and 'worst' in out, out[-2000:]


def test_exact_median_adversarial_inputs():
    out = _run('fuzz_median.py', 707, 20)
    assert 'mismatches 0' in out, out[-2000:]


def test_extended_filter_random_calibrated_rigs():
    out = _run('fuzz_ekf.py', 7, 808)
    assert 'above 1e-5' not in out and 'cases above tolerance 0' in out, out[-2000:]
