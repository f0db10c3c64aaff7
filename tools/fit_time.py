"""Where the wall clock of a `fit_eks_singlecam` call goes: the reference's file-in / file-out
surface (eks/singlecam_smoother.py:23-102) on synthetic prediction files of BASELINE configs[1]'s size (5 members x
10 000 frames x 64 keypoints) and of a long session (5 x 100 000 x 30), read / ensemble + smooth (device-resident
driver, tables included) / write, with pandas' own reader and writer beside the library's (EKS_PANDAS_CSV=1).
Usage: python tools/fit_time.py [small]"""
import os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandas as pd
import torch
from eks_amd import singlecam_smoother as sc, utils


def write_inputs(d, M, T, K, seed):
    rng = np.random.default_rng(seed)
    x = np.cumsum(rng.normal(size=(T, K, 2)), axis=0) + 200.0
    for m in range(M):
        obs = (x + rng.normal(size=(T, K, 2)) * 0.7).astype(np.float32)
        lik = rng.beta(50, 1, size=(T, K, 1)).astype(np.float32)
        arr = np.concatenate([obs, lik], axis=2).reshape(T, K * 3)
        cols = utils.make_dlc_pandas_index([f'kp{i}' for i in range(K)])
        df = pd.DataFrame(arr.astype(np.float64), columns=cols)
        df.columns = df.columns.set_levels(['net'], level=0)
        utils.write_prediction_csv(df, os.path.join(d, f'pred_{m}.csv'))


def timed(fn):
    t0 = time.perf_counter()
    out = fn()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def run(M, T, K, label):
    d = tempfile.mkdtemp(prefix='eks_fit_')
    try:
        write_inputs(d, M, T, K, seed=T + K)
        mb = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / 1e6
        print(f'== {label}: {M} files x {T} frames x {K} keypoints, {mb:.0f} MB of CSV')
        for mode in ('library', 'pandas'):
            if mode == 'pandas':
                os.environ['EKS_PANDAS_CSV'] = '1'
            else:
                os.environ.pop('EKS_PANDAS_CSV', None)
            best = None
            for rep in range(2):
                (dfs, names), t_read = timed(lambda: utils.format_data(d))
                ma, t_ma = timed(lambda: sc.input_dfs_to_markerArray([dfs], names, ['']))
                (df, s), t_smooth = timed(lambda: sc.ensemble_kalman_smoother_singlecam(ma, names, smooth_param=10.0))
                out = os.path.join(d, 'out', 'eks.csv')
                os.makedirs(os.path.dirname(out), exist_ok=True)
                _, t_write = timed(lambda: utils.write_prediction_csv(df, out))
                tot = t_read + t_ma + t_smooth + t_write
                if best is None or tot < best[-1]:
                    best = (t_read, t_ma, t_smooth, t_write, tot)
            print(f'   {mode:8s} read {best[0]*1e3:8.1f} ms | marker array {best[1]*1e3:7.1f} | ensemble + smooth + tables '
                  f'{best[2]*1e3:8.1f} | write {best[3]*1e3:8.1f} | total {best[4]*1e3:8.1f} ms'
                  f'   (output {os.path.getsize(out)/1e6:.0f} MB)')
        os.environ.pop('EKS_PANDAS_CSV', None)
        # the whole call, as the CLI makes it
        _, t_fit = timed(lambda: sc.fit_eks_singlecam(d, os.path.join(d, 'out', 'fit.csv'), smooth_param=10.0))
        print(f'   fit_eks_singlecam end to end (library reader / writer): {t_fit*1e3:.1f} ms')
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == '__main__':
    print('host cores:', os.cpu_count())
    run(5, 10_000, 64, 'BASELINE configs[1] size')
    if 'small' not in sys.argv:
        run(5, 100_000, 30, 'long session')
