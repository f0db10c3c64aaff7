"""CPU: the boundary's file formats.  `read_prediction_csv` (the library's own reader of the numeric body,
eks_csv_read_numeric: mmap, threads, pandas' decimal -> double conversion restated) must return what
`pd.read_csv(path, header=[0, 1, 2], index_col=0)` returns (reference eks/utils.py:188) - values bit for bit, dtypes,
index, column index - and `write_prediction_csv` must write the bytes `DataFrame.to_csv` writes (reference
eks/singlecam_smoother.py:98-99).  pandas is the CHECKER here, never the path."""
import os

import numpy as np
import pandas as pd
import pytest

from eks_amd import _lib
from eks_amd.utils import make_dlc_pandas_index, read_prediction_csv, write_prediction_csv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason='libeks_hip.so not built')


def _write_dlc(path, K, rows):
    with open(path, 'w') as f:
        f.write('scorer,' + ','.join(['net'] * (3 * K)) + '\n')
        f.write('bodyparts,' + ','.join(f'kp{i // 3}' for i in range(3 * K)) + '\n')
        f.write('coords,' + ','.join(['x', 'y', 'likelihood'][i % 3] for i in range(3 * K)) + '\n')
        for r, row in enumerate(rows):
            f.write(str(r) + ',' + ','.join(row) + '\n')


def _token(rng, v):
    k = rng.integers(0, 15)
    if k == 0: return repr(float(v))                                         # shortest round trip: up to 17 digits
    if k == 1: return repr(float(np.float32(v)))                             # what a float32 network output looks like
    if k == 2: return '%.25f' % v                                            # more digits than a double holds
    if k == 3: return '%.3e' % v
    if k == 4: return '%.20E' % v
    if k == 5: return '%.17g' % (v * 10.0 ** rng.integers(-320, 300))        # subnormals ... 1e300
    if k == 6: return ('%.12f' % v).rstrip('0')                              # "12." and friends
    if k == 7: return '+' + repr(float(abs(v)))
    if k == 8: return ' ' + repr(float(v)) + '  '
    if k == 9: return '000' + '%.8f' % abs(v)                                # leading zeros count as digits upstream
    if k == 10: return str(int(v * 1000))
    if k == 11: return ['', 'NaN', 'nan', 'NA', 'inf', '-inf', 'Infinity', '-Infinity', 'null', 'N/A', '-nan', '#N/A'][rng.integers(0, 12)]
    if k == 12: return '%d.' % int(v * 7)
    if k == 13: return '12345678901234567890123.5'                           # digits past the 17th move the exponent
    return '.%09d' % rng.integers(0, 10 ** 9)


def _same(path, **kw):
    ref = pd.read_csv(path, header=[0, 1, 2], index_col=0)
    got = read_prediction_csv(path, **kw)
    pd.testing.assert_frame_equal(ref, got, check_exact=True)
    if all(dt.kind in 'fi' for dt in ref.dtypes):
        a, b = ref.to_numpy(dtype=float), got.to_numpy(dtype=float)
        assert np.array_equal(a.view(np.int64)[~np.isnan(a)], b.view(np.int64)[~np.isnan(b)])   # bit for bit
    return ref


@pytest.mark.parametrize('n_threads', [1, 5])
def test_reader_equals_pandas_on_a_fuzz_corpus_of_decimal_strings(tmp_path, n_threads):
    rng = np.random.default_rng(n_threads)
    K = 12
    rows = [[_token(rng, rng.normal() * 10.0 ** rng.integers(-8, 8)) for _ in range(3 * K)] for _ in range(6000)]
    p = str(tmp_path / 'fuzz.csv')
    _write_dlc(p, K, rows)
    ref = _same(p, n_threads=n_threads)
    assert ref.shape == (6000, 3 * K) and np.isnan(ref.to_numpy(dtype=float)).any()


def test_reader_dtypes_blank_lines_crlf_and_the_fallback(tmp_path):
    rng = np.random.default_rng(3)
    # integer columns (one with a missing value: float64 upstream), blank lines, CRLF line ends
    rows = [[str(rng.integers(-10 ** 6, 10 ** 6)) if c % 3 else repr(float(rng.normal())) for c in range(6)] for _ in range(300)]
    rows[7][1] = ''
    p = str(tmp_path / 'ints.csv')
    _write_dlc(p, 2, rows)
    ref = _same(p)
    assert [str(d) for d in ref.dtypes] == ['float64', 'float64', 'int64', 'float64', 'int64', 'int64']
    txt = open(p).read().replace('\n', '\r\n').replace('\r\n5,', '\r\n\r\n5,')
    p2 = str(tmp_path / 'crlf.csv')
    open(p2, 'w', newline='').write(txt)
    _same(p2)
    # text in the body, quotes, a ragged line: pandas' business - the wrapper hands the file over, same result
    for bad in ('3,' + ','.join(['0.5'] * 5) + ',abc\n', '3,"0.5",' + ','.join(['0.5'] * 5) + '\n'):
        p3 = str(tmp_path / 'text.csv')
        lines = open(p).read().split('\n')
        lines[6] = bad.rstrip('\n')
        open(p3, 'w').write('\n'.join(lines))
        _same(p3)
    # an empty table
    p4 = str(tmp_path / 'empty.csv')
    _write_dlc(p4, 2, [])
    assert read_prediction_csv(p4).shape == pd.read_csv(p4, header=[0, 1, 2], index_col=0).shape


def test_reader_equals_pandas_on_the_reference_recordings(golden_dir, tmp_path):
    """tests/golden/csv_samples.npz (tools/make_golden.py csv) keeps the header rows and the first 40 / last 5 lines of
    every recording under the reference's data/ - data files of its own tests - with the values pandas read from them
    in the build container: the reader returns those bits, and what pandas returns here.  In the build container the
    whole recordings are compared as well."""
    import glob
    g = np.load(os.path.join(golden_dir, 'csv_samples.npz'))          # plain arrays: the texts are bytes + offsets
    assert len(g['names']) >= 30
    raw, offs = g['texts_bytes'].tobytes(), g['texts_offsets']
    texts = [raw[offs[i]:offs[i + 1]].decode('utf-8') for i in range(len(offs) - 1)]
    for i, (name, txt) in enumerate(zip(g['names'], texts)):
        p = str(tmp_path / f'sample_{i}.csv')
        open(p, 'w').write(str(txt))
        got = read_prediction_csv(p)
        ref = g[f'values_{i}']
        a, b = got.to_numpy(dtype=np.float64), ref
        assert a.shape == b.shape, name
        assert np.array_equal(a.view(np.int64)[~np.isnan(a)], b.view(np.int64)[~np.isnan(b)]), name
        assert np.array_equal(np.isnan(a), np.isnan(b)), name
        _same(p)
    if os.path.isdir('/root/reference/data'):
        for f in sorted(glob.glob('/root/reference/data/**/*.csv', recursive=True)):
            _same(f)


@pytest.mark.parametrize('with_nan', [False, True])
def test_writer_writes_pandas_bytes(tmp_path, with_nan):
    rng = np.random.default_rng(5)
    T, K = 400, 5
    labels = ['x', 'y', 'likelihood', 'x_ens_median', 'y_ens_median', 'x_ens_var', 'y_ens_var', 'x_posterior_var', 'y_posterior_var']
    arr = (rng.normal(size=(T, K * 9)) * 10.0 ** rng.integers(-6, 17, size=(T, K * 9))).astype(np.float32).astype(np.float64)
    arr[::3] += rng.normal(size=arr[::3].shape)                  # doubles that are not float32 values
    arr[1, 2], arr[2, 3], arr[3, 4], arr[4, 5], arr[5, 6] = 1e16, 1e-5, 0.0001, -0.0, 5.0
    arr[6, 7], arr[7, 8] = np.inf, -np.inf
    if with_nan:
        arr[8, 9] = np.nan
    df = pd.DataFrame(arr, columns=make_dlc_pandas_index([f'kp{i}' for i in range(K)], labels))
    a, b = str(tmp_path / 'a.csv'), str(tmp_path / 'b.csv')
    df.to_csv(a)
    write_prediction_csv(df, b)
    assert open(a, 'rb').read() == open(b, 'rb').read()
    # and it reads back as pandas reads it back (NOT always as it was: pandas' default conversion is not the correctly
    # rounded one - one more reason the reader restates pandas' arithmetic instead of calling strtod)
    _same(b)
    # a table with an integer column, one with a string column (pandas' path), an empty one
    df2 = pd.DataFrame({'a': [1.5, 2.5], 'b': [3, 4]})
    df3 = pd.DataFrame({'a': [1.5, 2.5], 'b': ['u', 'v']})
    for d in (df2, df3, df2.iloc[:0]):
        d.to_csv(a)
        write_prediction_csv(d, b)
        assert open(a, 'rb').read() == open(b, 'rb').read()


def test_library_writer_is_pandas_bytes_and_pythons_repr(tmp_path, monkeypatch):
    """eks_csv_write_table (threads; taken where a process's threads run side by side): forced on here, the file is
    DataFrame.to_csv's byte for byte; and its number formatting is Python's repr on a corpus that covers what the search
    for the shortest digits can get wrong - every power of two (lopsided rounding interval), their neighbours, random
    bit patterns, float32 values, short decimals, halves (double rounding), subnormals, the notation thresholds."""
    import ctypes
    from eks_amd import utils
    lib = _lib.load()
    rng = np.random.default_rng(11)
    p2 = np.array([2.0 ** e for e in range(-1074, 1024)])
    bits = rng.integers(0, 2 ** 63, size=60000, dtype=np.int64).view(np.float64)
    corpus = np.concatenate([
        p2, -p2[::5], np.nextafter(p2[500:1600], np.inf), np.nextafter(p2[500:1600], -np.inf), bits[np.isfinite(bits)],
        (rng.normal(size=60000) * 10.0 ** rng.integers(-6, 6, size=60000)).astype(np.float32).astype(np.float64),
        rng.normal(size=60000) * 10.0 ** rng.integers(-6, 6, size=60000),
        np.round(rng.normal(size=30000) * 100, 3), np.arange(4000) / 8.0,
        np.array([float(f'{a}.{b}5') for a in range(40) for b in range(100)]),
        np.array([0.0, -0.0, 1e16, 1e15, 9999999999999998.0, 1e-5, 1e-4, 123456789012345680.0, 5e-324,
                  1.7976931348623157e308, np.inf, -np.inf, np.nan, 1.0, 0.1, 1 / 3, 1e22, 1e23, 9.999999999999999e22,
                  0.9999999999999999, 2.675, 1e-7])])
    n = len(corpus)
    out = ctypes.create_string_buffer(n * 32 + 64)
    offs = np.zeros(n + 1, np.int64)
    assert lib.eks_format_repr(corpus.ctypes.data_as(ctypes.c_void_p), n, out, len(out), offs.ctypes.data_as(ctypes.c_void_p)) == 0
    raw = out.raw
    for i, v in enumerate(corpus.tolist()):
        assert raw[offs[i]:offs[i + 1]].decode() == ('' if v != v else repr(v)), (v, raw[offs[i]:offs[i + 1]])
    # the table writer, forced on
    monkeypatch.setattr(utils, '_WRITER_THREADS', [3])
    T, K = 3000, 8
    labels = ['x', 'y', 'likelihood', 'x_ens_median', 'y_ens_median', 'x_ens_var', 'y_ens_var', 'x_posterior_var', 'y_posterior_var']
    arr = corpus[rng.integers(0, n, size=T * K * 9)].reshape(T, K * 9)
    df = pd.DataFrame(arr, columns=make_dlc_pandas_index([f'kp{i}' for i in range(K)], labels))
    a, b = str(tmp_path / 'a.csv'), str(tmp_path / 'b.csv')
    df.to_csv(a)
    write_prediction_csv(df, b)
    assert open(a, 'rb').read() == open(b, 'rb').read()


def test_host_gather_cols_is_the_strided_copy():
    import ctypes
    lib = _lib.load()
    rng = np.random.default_rng(0)
    src = rng.normal(size=(3000, 40, 2)).astype(np.float32)
    for k0, k1 in ((0, 7), (7, 40), (13, 14)):
        dst = np.empty((3000, k1 - k0, 2), np.float32)
        rc = lib.eks_host_gather_cols(src.ctypes.data_as(ctypes.c_void_p), 3000, 40 * 2 * 4, k0 * 2 * 4, (k1 - k0) * 2 * 4,
                                      dst.ctypes.data_as(ctypes.c_void_p), 4)
        assert rc == 0
        np.testing.assert_array_equal(dst, src[:, k0:k1])
    assert lib.eks_host_gather_cols(src.ctypes.data_as(ctypes.c_void_p), 3000, 320, 300, 40, src.ctypes.data_as(ctypes.c_void_p), 1) != 0


def test_host_code_is_clean_under_address_sanitizer(tmp_path):
    """The library's host code (CSV body reader, table writer, number formatter, column gather) built for the CPU with
    AddressSanitizer + UBSan and driven over well-formed and adversarial files (tools/host_asan/: truncated, ragged, CRLF,
    NUL bytes, 400-digit fields, random bytes, short output capacities): no memory error, no undefined behaviour, no
    overrun guard touched.  CPU only: tools/host_asan/ does not travel to the GPU boxes (.gpurunignore; sanitizer builds
    are not allowed there)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, 'tools', 'host_asan', 'run.sh')
    if not os.path.exists(script):
        pytest.skip('tools/host_asan is not in this copy of the tree')
    r = subprocess.run(['bash', script, str(tmp_path / 'work')], capture_output=True, text=True, timeout=600)
    if r.returncode == 77:
        pytest.skip('no g++ with the sanitizer runtimes here')
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert ', 0 problems' in r.stdout
