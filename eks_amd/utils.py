"""Host-side data formats either side of the Kalman path (reference eks/utils.py).

CSV in: DLC / Lightning Pose prediction files with a 3-row header (scorer / bodyparts / coords).
DataFrame out: MultiIndex columns ('ensemble-kalman_tracker', keypoint, coord).
"""
from __future__ import annotations

import logging
import os

import numpy as np
import pandas as pd

from .marker_array import MarkerArray

logger = logging.getLogger(__name__)

SCORER = 'ensemble-kalman_tracker'


def make_dlc_pandas_index(keypoint_names, labels=('x', 'y', 'likelihood')) -> pd.MultiIndex:
    """scorer / bodyparts / coords column index (reference eks/utils.py:15-32)."""
    return pd.MultiIndex.from_product([[SCORER], list(keypoint_names), list(labels)],
                                      names=['scorer', 'bodyparts', 'coords'])


def get_keypoint_names(df: pd.DataFrame) -> list:
    """Body-part names in file order (one per 'x' column), reference eks/utils.py:125-135."""
    is_x = df.columns.get_level_values('coords') == 'x'
    return df.columns[is_x].get_level_values('bodyparts').tolist()


def convert_lp_dlc(df_lp: pd.DataFrame, keypoint_names, model_name=None) -> pd.DataFrame:
    """3-level header DataFrame -> flat columns '{keypoint}_{x|y|likelihood}' (reference
    eks/utils.py:35-69).  Missing columns and 'Unnamed' levels are skipped."""
    if model_name is None:
        model_name = str(df_lp.columns[0][0])
    flat = {}
    for kp in keypoint_names:
        for coord in ('x', 'y', 'likelihood'):
            key = (model_name, kp, coord)
            if any(isinstance(lv, str) and lv.startswith('Unnamed') for lv in key):
                continue
            if key in df_lp.columns:
                flat[f'{kp}_{coord}'] = df_lp[key]
    return pd.DataFrame(flat, index=df_lp.index)


_HEADER_CACHE: dict = {}


def _header_frame(path):
    """The column index and index name pandas builds from the file's three header rows (an empty frame), cached by the
    rows' text: the members of an ensemble share their header, and pandas takes 20 ms to read three lines."""
    try:
        with open(path, 'rb') as f:
            rows = [f.readline() for _ in range(3)]
    except OSError:
        return None
    key = b''.join(rows)
    if key not in _HEADER_CACHE:
        if len(_HEADER_CACHE) > 64:
            _HEADER_CACHE.clear()
        _HEADER_CACHE[key] = pd.read_csv(path, header=[0, 1, 2], index_col=0, nrows=0)
    return _HEADER_CACHE[key]


def read_prediction_csv(path: str, n_threads: int | None = None) -> pd.DataFrame:
    """`pd.read_csv(path, header=[0, 1, 2], index_col=0)` (reference eks/utils.py:188) with the numeric body parsed by
    the library's own reader (`eks_csv_read_numeric`: mmap, one thread per block of lines, pandas' own decimal ->
    double conversion restated - the values are pandas' bit for bit, tests/test_csv_ingest.py).  The three header rows
    still go through pandas (three lines).  Files that hold anything but numbers and missing values, and any file when
    EKS_PANDAS_CSV is set or the library is not built, are read by pandas as before."""
    if os.environ.get('EKS_PANDAS_CSV'):
        return pd.read_csv(path, header=[0, 1, 2], index_col=0)
    try:
        import ctypes
        from . import _lib
        lib = _lib.load()
    except Exception:                       # no library here (a CPU-only checkout): pandas
        return pd.read_csv(path, header=[0, 1, 2], index_col=0)
    if n_threads is None:
        n_threads = max(1, min(16, os.cpu_count() or 1))
    n_rows, n_cols = ctypes.c_int64(0), ctypes.c_int32(0)
    bpath = os.fsencode(path)
    rc = lib.eks_csv_read_numeric(bpath, 3, None, 0, ctypes.byref(n_rows), ctypes.byref(n_cols), None, 0, n_threads)
    if rc != 0 or n_cols.value < 2:
        return pd.read_csv(path, header=[0, 1, 2], index_col=0)
    head = _header_frame(path)
    if head is None or len(head.columns) != n_cols.value - 1:
        return pd.read_csv(path, header=[0, 1, 2], index_col=0)
    body = np.empty((n_rows.value, n_cols.value), dtype=np.float64)
    is_int = np.zeros(n_cols.value, dtype=np.uint8)
    rc = lib.eks_csv_read_numeric(bpath, 3, body.ctypes.data_as(ctypes.c_void_p), body.size, ctypes.byref(n_rows),
                                  ctypes.byref(n_cols), is_int.ctypes.data_as(ctypes.c_void_p), is_int.size, n_threads)
    if rc != 0 or not is_int[0]:            # (text somewhere, ragged lines, a non-integer index column: pandas' call)
        return pd.read_csv(path, header=[0, 1, 2], index_col=0)
    index = pd.Index(body[:, 0].astype(np.int64), name=head.index.name)
    df = pd.DataFrame(body[:, 1:], index=index, columns=head.columns)
    ints = np.flatnonzero(is_int[1:])
    if ints.size:                           # columns written as integers throughout: int64, as pandas infers
        df = df.astype({df.columns[i]: np.int64 for i in ints})
    return df


_WRITER_THREADS = [None]


def _writer_threads() -> int:
    """Threads for the library's table writer, or 0 = format in Python.  One number costs the C library's printf /
    strtod pair ~2 us on one thread against 0.3 us for Python's repr, so the threaded writer only pays where the
    process's threads really run side by side (measured once, ~20 ms: eks_host_thread_speedup; on the MI355X box's 256
    cores 32 threads write BASELINE configs[1]'s table in 0.31 s - Python's repr 1.57 s, pandas 5.9 s)."""
    if _WRITER_THREADS[0] is None:
        n = 0
        try:
            from . import _lib
            want = max(1, min(32, (os.cpu_count() or 1) // 2))
            lib = _lib.load()
            if want >= 8 and not os.environ.get('EKS_PY_CSV_WRITER') and \
                    max(lib.eks_host_thread_speedup(8), lib.eks_host_thread_speedup(8)) >= 4.5:
                n = want
        except Exception:
            n = 0
        _WRITER_THREADS[0] = n
    return _WRITER_THREADS[0]


def write_prediction_csv(df: pd.DataFrame, path) -> None:
    """`df.to_csv(path)` (reference eks/singlecam_smoother.py:98-99, eks/multicam_smoother.py:151-152, :270-275) for
    the result tables, byte for byte: the header comes from pandas itself (the empty slice's to_csv), every number is
    Python's own `repr` - the shortest string that reads back as the same double, which is what pandas writes - and a
    missing value is the empty field.  pandas spends 2.2 us per number in its object-array formatter (13 s for BASELINE
    configs[1]'s 10 000 x 576 table); `map(repr, row)` does the same text in a quarter of that.  Tables that are not
    plain float64 / int64 with a plain integer index, and any table when EKS_PANDAS_CSV is set, go through pandas
    (tests/test_csv_ingest.py compares the bytes)."""
    plain = (not os.environ.get('EKS_PANDAS_CSV') and len(df) > 0 and df.shape[1] > 0
             and all(dt == np.float64 or dt == np.int64 for dt in df.dtypes)
             and df.index.nlevels == 1 and df.index.dtype == np.int64)
    if not plain:
        df.to_csv(path)
        return
    head = df.iloc[:0].to_csv()
    all_float = all(dt == np.float64 for dt in df.dtypes)
    n_thr = _writer_threads() if all_float and df.size >= 200_000 else 0
    if n_thr:
        # the library's own writer (eks_csv_write_table): the same text, row blocks formatted on n_thr threads
        import ctypes
        from . import _lib
        vals = np.ascontiguousarray(df.to_numpy(), dtype=np.float64)
        idx64 = np.ascontiguousarray(df.index.to_numpy(), dtype=np.int64)
        hb = head.encode()
        rc = _lib.load().eks_csv_write_table(os.fsencode(os.fspath(path)), hb, len(hb), idx64.ctypes.data_as(ctypes.c_void_p),
                                             vals.ctypes.data_as(ctypes.c_void_p), vals.shape[0], vals.shape[1], n_thr)
        _lib.check(rc, 'eks_csv_write_table')
        return
    idx = df.index.tolist()
    if all_float:
        vals = df.to_numpy()
        fmt = (lambda x: '' if x != x else repr(x)) if np.isnan(vals).any() else repr
        rows = vals.tolist()
        body = '\n'.join([str(i) + ',' + ','.join(map(fmt, r)) for i, r in zip(idx, rows)])
    else:
        cols = [df.iloc[:, c].tolist() for c in range(df.shape[1])]
        fmt = lambda x: '' if x != x else repr(x)          # (repr of a Python int is its decimal string)
        body = '\n'.join([str(i) + ',' + ','.join(map(fmt, r)) for i, r in zip(idx, zip(*cols))])
    with open(path, 'w', newline='') as f:
        f.write(head)
        f.write(body)
        f.write('\n')


def _read_prediction_file(path: str):
    if path.endswith('.slp'):
        raise NotImplementedError(
            'SLEAP .slp input needs the sleap_io reader (reference eks/utils.py:72-122); '
            'convert to CSV first - it is outside the accelerated path')
    raw = read_prediction_csv(path)
    names = get_keypoint_names(raw)
    return convert_lp_dlc(raw, names), names


def format_data(input_source, camera_names=None):
    """Load prediction files (reference eks/utils.py:138-232).

    input_source: directory, list of paths, or {camera: [paths]}.  Paths are sorted
    lexicographically; with `camera_names`, files are matched to a camera when the camera name is a
    substring of the file's basename.  Returns (list of DataFrames | list per camera of lists,
    keypoint_names)."""
    if isinstance(input_source, str) and os.path.isdir(input_source):
        paths = sorted(os.path.join(input_source, f) for f in os.listdir(input_source))
    elif isinstance(input_source, list):
        paths = sorted(input_source)
    elif isinstance(input_source, dict):
        paths = input_source
    else:
        raise ValueError('input_source must be a directory path, a list of file paths, or a map '
                         'from camera names to list of file paths')
    out, names = [], None
    if camera_names is None:
        for p in paths:
            if not (p.endswith('.csv') or p.endswith('.slp')):
                continue
            df, names = _read_prediction_file(p)
            out.append(df)
    else:
        for cam in camera_names:
            files = paths if isinstance(paths, list) else paths.get(cam, [])
            hits = [p for p in files if cam in os.path.basename(p)
                    and (p.endswith('.csv') or p.endswith('.slp'))]
            if not hits:
                raise FileNotFoundError(
                    f"no files matching camera '{cam}' found in {input_source}. "
                    f'ensure the camera name appears as a substring of each filename.')
            per_cam = []
            for p in hits:
                df, names = _read_prediction_file(p)
                per_cam.append(df)
            out.append(per_cam)
        counts = [len(x) for x in out]
        if len(set(counts)) > 1:
            logger.warning('unequal number of seed files per camera (%s)',
                           ', '.join(f'{c}: {n}' for c, n in zip(camera_names, counts)))
    if not out:
        raise FileNotFoundError(f'no valid marker input files found in {input_source}')
    assert names is not None
    return out, names


def crop_frames(y, s_frames):
    """Keep the frames in `s_frames` (list of 0-based half-open (start, end) tuples, None = open
    end); None / [] / [(None, None)] returns `y` itself.  Reference eks/utils.py:235-290."""
    if s_frames is None or len(s_frames) == 0 or \
            (len(s_frames) == 1 and s_frames[0] == (None, None)):
        return y
    if not isinstance(s_frames, list):
        raise TypeError('s_frames must be a list of (start, end) tuples or None.')
    idx = frame_spans(len(y), s_frames)
    if len(idx) == 1:
        return y[idx[0][0]:idx[0][1]]
    return np.concatenate([y[a:b] for a, b in idx], axis=0)


def frame_spans(n: int, s_frames) -> list[tuple[int, int]]:
    """Validated, sorted (start, end) spans of `crop_frames` for a sequence of length n."""
    spans = []
    for i, fr in enumerate(s_frames):
        if not (isinstance(fr, tuple) and len(fr) == 2):
            raise ValueError(f's_frames[{i}] must be a (start, end) tuple, got {fr!r}')
        lo, hi = fr
        if lo is not None and not isinstance(lo, int):
            raise ValueError(f's_frames[{i}].start must be int or None, got {lo!r}')
        if hi is not None and not isinstance(hi, int):
            raise ValueError(f's_frames[{i}].end must be int or None, got {hi!r}')
        lo = 0 if lo is None else lo
        hi = n if hi is None else hi
        if lo < 0 or hi > n:
            raise ValueError(f'Range ({lo}, {hi}) out of bounds for length {n}.')
        if lo >= hi:
            raise ValueError(f'Invalid range ({lo}, {hi}).')
        spans.append((lo, hi))
    spans.sort(key=lambda ab: ab[0])
    for prev, cur in zip(spans, spans[1:]):
        if cur[0] < prev[1]:
            raise ValueError(f'Overlapping or out-of-order intervals: {prev} and {cur}')
    return spans


def build_R_from_vars(ev: np.ndarray) -> np.ndarray:
    """(..., T, O) variances -> (..., T, O, O) diagonal matrices, variances clipped at 1e-12
    (reference eks/utils.py:368-377).  Kept for API parity; the kernels read the O variances and
    never materialise R."""
    v = np.clip(np.asarray(ev), 1e-12, None)
    return v[..., :, None] * np.eye(v.shape[-1], dtype=v.dtype)


def crop_R(R: np.ndarray, s_frames) -> np.ndarray:
    """Crop (..., T, O, O) along T with `crop_frames` semantics (reference eks/utils.py:380-398)."""
    R = np.asarray(R)
    if not s_frames:
        return R
    lead = R.shape[:-3]
    T, O, O2 = R.shape[-3:]
    assert O == O2, 'R_tv must be square in its last two dims'
    flat = R.reshape((-1, T, O, O))
    cropped = np.stack([crop_frames(b, s_frames) for b in flat], axis=0)
    return cropped.reshape((*lead, -1, O, O))


def center_predictions(ensemble_marker_array: MarkerArray, quantile_keep_pca: float):
    """Per-keypoint low-variance frame mask, mean over the kept frames, centred predictions
    (reference eks/utils.py:293-365).

    Returns (valid_frames_mask (T,K) bool, emA_centered_preds (1,V,T,K,2),
    emA_good_centered_preds (1,V,min_frames,K,2), emA_means (1,V,1,K,2))."""
    M, V, T, K, _ = ensemble_marker_array.shape
    assert M == 1, 'MarkerArray should have n_models = 1 after ensembling.'
    # float64 on the host: the reference does this in float32 (its ensemble arrays are float32,
    # SURVEY.md A.4); the means feed every output column, so the extra digits are kept
    preds = np.asarray(ensemble_marker_array.slice_fields('x', 'y').array, dtype=np.float64)
    vars_ = np.asarray(ensemble_marker_array.slice_fields('var_x', 'var_y').array)
    if quantile_keep_pca >= 100 and not np.isnan(vars_).any():
        # every frame is kept (the single-camera drivers): no percentile, no ordering
        mask = np.ones((T, K), dtype=bool)
        good = preds
    else:
        worst = vars_.max(axis=(0, 1, 4))                               # (T, K)
        mask = worst <= np.percentile(worst, quantile_keep_pca, axis=0)
        n_good = int(mask.sum(axis=0).min())
        # first n_good kept frames of every keypoint (the reference truncates to the shortest list)
        order = np.argsort(~mask, axis=0, kind='stable')[:n_good]       # (n_good, K) frame indices
        kk = np.arange(K)[None, :]
        good = preds[:, :, order, kk, :]                                # (1, V, n_good, K, 2)
    means = good.mean(axis=2, keepdims=True)                            # (1, V, 1, K, 2)
    centered = preds - means
    fields = ['x', 'y']
    return (mask, MarkerArray(centered, data_fields=fields),
            MarkerArray(good - means, data_fields=fields), MarkerArray(means, data_fields=fields))


def percentile_ranks(n: int, q: float, dtype=np.float32):
    """What numpy.percentile(a, q, axis=0) (method 'linear') does before it touches the data, for `n`
    values of floating `dtype`: the indices of the two order statistics it interpolates between and the
    weight `gamma`, formed with numpy's own expressions (numpy/lib/_function_base_impl.py: _quantile_unchecked
    -> _quantile -> _get_indexes / _get_gamma) so that the result is bit-identical under whatever promotion
    rules the installed numpy applies.  Returns (prev_index, next_index, gamma as a 0-d array)."""
    dtype = np.dtype(dtype)
    qq = np.asanyarray(np.true_divide(q, dtype.type(100)))      # percentile -> quantile, as numpy.percentile
    if not (0.0 <= float(qq) <= 1.0):
        raise ValueError('Percentiles must be in the range [0, 100]')
    vi = np.asanyarray((n - 1) * qq)                            # 'linear': virtual index (n - 1) q
    prev = int(np.floor(vi))
    nxt = prev + 1
    if vi >= n - 1:
        prev = nxt = n - 1
    if vi < 0:
        prev = nxt = 0
    gamma = np.asanyarray(np.asanyarray(vi - np.intp(prev)), dtype=vi.dtype)
    return prev, min(nxt, n - 1), gamma


def percentile_from_order_stats(vals, gamma, nan_count=None):
    """The interpolation step of numpy.percentile ('linear'; numpy's _lerp) from the two order statistics
    vals (..., 2) of each slice; slices with NaNs give NaN like numpy (its partition puts NaNs last and the
    result is overwritten).  Same operations in the same dtype as numpy, hence bit-identical."""
    vals = np.asarray(vals)
    a, b = vals[..., 0], vals[..., 1]
    t = gamma
    diff = np.subtract(b, a)
    out = np.asanyarray(np.add(a, diff * t))
    np.subtract(b, diff * (1 - t), out=out, where=t >= 0.5, casting='unsafe', dtype=type(out.dtype))
    if nan_count is not None:
        out = np.where(np.asarray(nan_count) > 0, np.asarray(np.nan, dtype=out.dtype), out)
    return out
