import os, sys, time, ctypes
sys.path.insert(0, os.getcwd())
import numpy as np, pandas as pd
from eks_amd import _lib, utils
lib = _lib.load()
for n in (2, 4, 8, 16, 32, 64):
    print('speedup', n, [round(lib.eks_host_thread_speedup(n), 2) for _ in range(3)])
print('writer threads chosen:', utils._writer_threads())
rng = np.random.default_rng(0)
T, K = 10000, 64
arr = (rng.normal(size=(T, K * 9)) * 100).astype(np.float32).astype(np.float64) + rng.normal(size=(T, K * 9))
df = pd.DataFrame(arr, columns=utils.make_dlc_pandas_index([f'kp{i}' for i in range(K)], ['x', 'y', 'likelihood', 'x_ens_median', 'y_ens_median', 'x_ens_var', 'y_ens_var', 'x_posterior_var', 'y_posterior_var']))
for thr in (0, 1, 8, 32, 64, 128):
    utils._WRITER_THREADS[0] = thr
    t0 = time.time(); utils.write_prediction_csv(df, '/tmp/w.csv'); t1 = time.time()
    print('writer threads', thr, '%.3f s' % (t1 - t0))
vals = rng.normal(size=(T, 3 * K)).astype(np.float32) * 100
d2 = pd.DataFrame(vals.astype(np.float64), columns=utils.make_dlc_pandas_index([f'kp{i}' for i in range(K)]))
utils._WRITER_THREADS[0] = 0
utils.write_prediction_csv(d2, '/tmp/r.csv')
n_rows, n_cols = ctypes.c_int64(0), ctypes.c_int32(0)
for thr in (1, 4, 16, 32, 64):
    t0 = time.time(); lib.eks_csv_read_numeric(b'/tmp/r.csv', 3, None, 0, ctypes.byref(n_rows), ctypes.byref(n_cols), None, 0, thr); t1 = time.time()
    body = np.empty((n_rows.value, n_cols.value)); ii = np.zeros(n_cols.value, np.uint8)
    t2 = time.time(); lib.eks_csv_read_numeric(b'/tmp/r.csv', 3, body.ctypes.data_as(ctypes.c_void_p), body.size, ctypes.byref(n_rows), ctypes.byref(n_cols), ii.ctypes.data_as(ctypes.c_void_p), ii.size, thr); t3 = time.time()
    print('reader threads', thr, 'query %.4f parse %.4f' % (t1 - t0, t3 - t2))
for rep in range(3):
    t0 = time.time(); a = utils.read_prediction_csv('/tmp/r.csv'); t1 = time.time()
    print('read_prediction_csv %.4f' % (t1 - t0))
t0 = time.time(); b = pd.read_csv('/tmp/r.csv', header=[0, 1, 2], index_col=0); print('pandas %.4f' % (time.time() - t0))
