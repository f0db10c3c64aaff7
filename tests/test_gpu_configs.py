"""GPU: every BASELINE.json GPU configuration at its STATED size, through the reference-shaped
entry points, against the float64 C port of the reference recursion (oracle/eks_oracle.c) on the
same inputs - whole-output comparison in the spirit of the reference's integration tests
(/root/reference tests/conftest.py:86-101).

  C2  singlecam 10 000 frames x 64 keypoints x 5 members, fixed s        ensemble_kalman_smoother_singlecam
  C3  singlecam 100 000 x 256, 64-candidate NLL grid + smooth            run_kalman_smoother(s_mode='grid')
  C4  mirrored multicam 2 views x 4 paws, 50 000 frames, 3-D state       ensemble_kalman_smoother_multicam
  C5  one GPU's share of 1024 sessions x 50 000 x 32 keypoints           distributed.smooth_sessions_batched
      (128 sessions -> one 4096-keypoint batch, full (K,T,2,2) covariances)

Bars (BASELINE.json): smoothed means / variances within 1e-5 relative to the keypoint's magnitude
(variances elementwise - every bar in this file is the stated 1e-5), NLL within 1e-5 relative, grid
argmin indices bit-exact.  C2 and C4 compare every output column of every keypoint; C3's grid compares ALL 256
keypoints (median, NLL table, argmin, every frame of the smoothed outputs); C3's Adam mode a 16-keypoint sample
(the oracle's trajectory costs ~10 core-seconds per keypoint); C5 one keypoint of each of its 128 sessions; all
check every keypoint for finiteness and for the size-independent identities of the
smoother (posterior variance below both the prior-predictive and the observation variance,
smoothed path inside the observations' envelope)."""
import os

import numpy as np
import pytest

from oracle import c_oracle
from oracle import eks_oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

TOL = 1e-5


def _threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    return max(1, min(n, c_oracle.max_threads()))


def _kp_rel(a, b):
    """max |a - b| relative to each keypoint's largest |b| (a, b: (K, T, ...))."""
    axes = tuple(range(1, b.ndim))
    return float((np.abs(a - b) / np.abs(b).max(axis=axes, keepdims=True)).max())


# ------------------------------------------------------------------------------------------
def test_c2_singlecam_10k_x_64_x_5_fixed_s():
    from eks_amd import MarkerArray, synth
    from eks_amd.singlecam_smoother import ensemble_kalman_smoother_singlecam
    T, K, M = 10_000, 64, 5
    mk = synth.singlecam_markers(T, K, M=M, seed=2)
    names = [f'kp{k}' for k in range(K)]
    df, s = ensemble_kalman_smoother_singlecam(
        MarkerArray(mk.astype(np.float64), data_fields=['x', 'y', 'likelihood']), names, smooth_param=10.0)
    assert df.shape == (T, K * 9) and np.all(s == 10.0)
    arrs = orc.singlecam_arrays(mk)
    Rd = orc.build_R_from_vars(np.swapaxes(arrs['ensemble_vars'], 0, 1))
    ms, Vs, _ = c_oracle.smooth(arrs['ys'], Rd, arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'],
                                np.full(K, 10.0), nthreads=_threads())
    ref = orc.singlecam_outputs(arrs, s, ms, Vs)
    got = df.values
    col_scale = np.abs(ref).max(axis=0)
    assert (np.abs(got - ref) / col_scale).max() < TOL                    # every column, every frame
    pv = slice(7, None, 9)                                                # posterior variances: elementwise
    assert (np.abs(got[:, pv] - ref[:, pv]) / ref[:, pv]).max() < TOL
    pv2 = slice(8, None, 9)
    assert (np.abs(got[:, pv2] - ref[:, pv2]) / ref[:, pv2]).max() < TOL


# ------------------------------------------------------------------------------------------
def test_c3_singlecam_100k_x_256_grid_search_and_smooth():
    from eks_amd import _lib, hip_ops, synth
    from eks_amd.core import run_kalman_smoother
    T, K, NC = 100_000, 256, 64
    dev = hip_ops.require_gpu()
    y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)          # (T,K,2) float32
    eye = np.tile(np.eye(2), (K, 1, 1))
    m0 = np.zeros((K, 2))
    S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
    s, ms, Vs = run_kalman_smoother(y.transpose(0, 1), m0, S0, eye, eye, eye, var, s_mode='grid', n_grid=NC,
                                    return_device=True)
    assert s.shape == (K,) and tuple(ms.shape) == (K, T, 2) and tuple(Vs.shape) == (K, T, 2, 2)
    assert bool(torch.isfinite(ms).all()) and bool(torch.isfinite(Vs).all())
    cand = np.exp(np.linspace(-8.0, 8.0, NC))
    idx = np.abs(np.log(s)[:, None] - np.log(cand)[None]).argmin(axis=1)
    assert np.allclose(s, cand[idx], rtol=1e-12)                          # s is a grid candidate

    # the NLL table of the same path, for the value comparison
    t64 = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)
    flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
    rc = hip_ops.const_r(var, 1e-4)
    nll = hip_ops.nll(y, rc, t64(m0), t64(S0), t64(eye), t64(eye), t64(eye), t64(cand), flags=flags)
    np.testing.assert_array_equal(nll.argmin(dim=1).cpu().numpy(), idx)

    # ---- oracle on ALL 256 keypoints, all frames (64 keypoints at a time to bound the host copies)
    rc_h, nll_h = rc.cpu().numpy(), nll.cpu().numpy()
    n_clear = 0
    for k0 in range(0, K, 64):
        sel = np.arange(k0, min(K, k0 + 64))
        y_s = np.transpose(y[:, sel].cpu().numpy().astype(np.float64), (1, 0, 2)).copy()   # (64,T,2)
        Rd = np.clip(np.transpose(var[:, sel].cpu().numpy().astype(np.float64), (1, 0, 2)), 1e-12, None)
        Rc = orc.constant_R_from_timevarying(Rd)
        np.testing.assert_array_equal(rc_h[sel], Rc)                          # the exact median, bit for bit
        # (the C port's scalar-chain form of the diagonal model - held to its general-matrix form to 1e-10 in
        #  tests/test_oracle_core.py - takes a fifth of the time; the first 64 keypoints go through BOTH)
        nll_o = c_oracle.nll_grid_diag(y_s, Rc, m0[sel], S0[sel], eye[sel], eye[sel], eye[sel], cand, nthreads=_threads())
        if k0 == 0:
            nll_g = c_oracle.nll_grid(y_s, Rc, m0[sel], S0[sel], eye[sel], eye[sel], eye[sel], cand, nthreads=_threads())
            assert (np.abs(nll_g - nll_o) / np.abs(nll_g)).max() < 1e-9
        assert (np.abs(nll_h[sel] - nll_o) / np.abs(nll_o)).max() < TOL
        idx_o = nll_o.argmin(axis=1)
        srt = np.sort(nll_o, axis=1)
        clear = (srt[:, 1] - srt[:, 0]) > 4 * TOL * np.abs(srt[:, 0])         # the oracle's own margin
        n_clear += int(clear.sum())
        np.testing.assert_array_equal(idx[sel][clear], idx_o[clear])          # indices bit-exact
        # a near-tie may legitimately fall either way: then the two losses agree within the bar
        pick = nll_o[np.arange(len(sel)), idx[sel]]
        assert (np.abs(pick - srt[:, 0]) <= 4 * TOL * np.abs(srt[:, 0])).all()

        smooth_o = c_oracle.smooth if k0 == 0 else c_oracle.smooth_diag
        ms_o, Vs_o, _ = smooth_o(y_s, Rd, m0[sel], S0[sel], eye[sel], eye[sel], eye[sel], s[sel], nthreads=_threads())
        ms_g = ms[k0:k0 + len(sel)].cpu().numpy().astype(np.float64)
        Vs_g = Vs[k0:k0 + len(sel)].cpu().numpy().astype(np.float64)
        assert _kp_rel(ms_g, ms_o) < TOL
        Vd_g = np.diagonal(Vs_g, axis1=2, axis2=3)
        Vd_o = np.diagonal(Vs_o, axis1=2, axis2=3) if Vs_o.ndim == 4 else Vs_o   # (smooth_diag returns the diagonal)
        assert (np.abs(Vd_g - Vd_o) / Vd_o).max() < TOL                        # elementwise
        assert np.all(Vs_g[..., 0, 1] == 0) and np.all(Vs_g[..., 1, 0] == 0)
        del y_s, Rd, ms_o, Vs_o, ms_g, Vs_g
    assert n_clear >= K // 2

    # ---- identities that hold at any size, on the device for every keypoint
    Vd = torch.diagonal(Vs, dim1=2, dim2=3)                                # (K,T,2)
    r = var.clamp_min(1e-12).transpose(0, 1)
    assert bool((Vd > 0).all()) and bool((Vd <= r * (1 + 1e-5)).all())    # never worse than the observation
    lo, hi = y.amin(dim=0), y.amax(dim=0)                                  # (K,2)
    assert bool((ms >= (lo - 1e-3)[:, None]).all()) and bool((ms <= (hi + 1e-3)[:, None]).all())


# ------------------------------------------------------------------------------------------
def test_c3_singlecam_100k_x_256_adam():
    """configs[2]'s session in the REFERENCE's own mode: smooth_param=None -> Adam on log s per keypoint
    (eks/core.py:562-699), loss over all 100 000 frames.  The device loop (loss + forward-mode gradient,
    finished keypoints skipped, the step applied at the end of the loss assembly) against the oracle's
    optimiser (oracle/eks_oracle.py: adam_optimize_s) fed by the C port's complex-step gradient, on a
    64-keypoint sample (16 on hosts with fewer than 64 cores): same stopping iteration, |d log s| <= 2e-6; then the smoothed outputs at the
    device's s within 1e-5 of the C port on every frame."""
    from concurrent.futures import ThreadPoolExecutor
    from eks_amd import hip_ops, synth
    from eks_amd.core import (_DeviceProblem, _optimize_on_device, compute_initial_guesses,
                              run_kalman_smoother)
    T, K = 100_000, 256
    KS = 64 if _threads() >= 64 else 16        # (the oracle runs a keypoint per host thread: ~12 s each)
    dev = hip_ops.require_gpu()
    y, var = synth.singlecam_observations_torch(T, K, seed=3, device=dev)
    eye = np.tile(np.eye(2), (K, 1, 1))
    m0 = np.zeros((K, 2))
    S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
    ev_head = var[:2000].cpu().numpy()
    guesses = np.array([compute_initial_guesses(ev_head[:, k]) or 2.0 for k in range(K)])
    P = _DeviceProblem(y.transpose(0, 1), m0, S0, eye, eye, eye, var)
    s_dev, info = _optimize_on_device(P, [[k] for k in range(K)], None, guesses, 0.25, (-8.0, 8.0), 1e-2, 300,
                                      1e-4, 'adam', 0)
    s = s_dev.cpu().numpy()
    st = info['state'].cpu().numpy()
    assert np.all(st[:, 5] == 1.0) and st[:, 4].max() < 300           # every keypoint stopped by the rule
    assert np.all(np.isfinite(s)) and np.all((s > np.exp(-8.0) * 0.999) & (s < np.exp(8.0) * 1.001))

    # ---- the oracle's trajectory on a sample (all frames; keypoints in parallel threads: ctypes drops the GIL)
    sel = np.linspace(0, K - 1, KS).round().astype(int)
    y_s = np.transpose(y[:, sel].cpu().numpy().astype(np.float64), (1, 0, 2)).copy()
    Rd = np.clip(np.transpose(var[:, sel].cpu().numpy().astype(np.float64), (1, 0, 2)), 1e-12, None)
    Rc = orc.constant_R_from_timevarying(Rd)
    u0 = np.array([np.float32(np.log(np.clip(g, 1e-6, 1e3))) for g in guesses[sel]], dtype=np.float64)
    zero = np.zeros((1, 2, 2))

    def one(k, u):
        sQ = np.exp(u) * np.eye(2)
        L, g = c_oracle.nll_directional(y_s[k], Rc[k], m0[sel[k]], S0[sel[k]], np.eye(2), np.eye(2), sQ, zero,
                                        sQ[None])
        return L, g[0]

    with ThreadPoolExecutor(max_workers=min(KS, _threads())) as pool:
        def loss_and_grad(u):
            res = list(pool.map(lambda k: one(k, u[k]), range(KS)))
            return np.array([r[0] for r in res]), np.array([r[1] for r in res])
        u_o, last_o, it_o = orc.adam_optimize_s(loss_and_grad, u0)
    s_o = np.exp(np.clip(u_o, -8.0, 8.0))
    np.testing.assert_array_equal(st[sel, 4].astype(int), it_o)        # same stopping iteration
    assert np.abs(np.log(s[sel]) - np.log(s_o)).max() < 2e-6   # (round 6: the search from cached lag sums)
    assert (np.abs(st[sel, 3] - last_o) / np.abs(last_o)).max() < TOL   # the last loss each saw

    # ---- final pass at the device's s, sample against the C port on every frame
    s2, ms, Vs = run_kalman_smoother(y.transpose(0, 1), m0, S0, eye, eye, eye, var, smooth_param=list(s),
                                     return_device=True)
    np.testing.assert_array_equal(s2, s)
    ms_o, Vs_o, _ = c_oracle.smooth(y_s, Rd, m0[sel], S0[sel], eye[sel], eye[sel], eye[sel], s[sel],
                                    nthreads=_threads())
    sel_d = torch.as_tensor(sel, device=dev)
    ms_g = ms.index_select(0, sel_d).cpu().numpy().astype(np.float64)
    Vs_g = Vs.index_select(0, sel_d).cpu().numpy().astype(np.float64)
    assert _kp_rel(ms_g, ms_o) < TOL
    Vd_g, Vd_o = np.diagonal(Vs_g, axis1=2, axis2=3), np.diagonal(Vs_o, axis1=2, axis2=3)
    assert (np.abs(Vd_g - Vd_o) / Vd_o).max() < TOL
    assert np.all(Vs_g[..., 0, 1] == 0) and np.all(Vs_g[..., 1, 0] == 0)


# ------------------------------------------------------------------------------------------
def _sk(X, n):
    from sklearn.decomposition import PCA
    p = PCA(n_components=n).fit(X)
    return p.components_, p.mean_


def test_c4_mirrored_multicam_2_views_x_4_paws_50k_frames():
    from eks_amd import MarkerArray, synth
    from eks_amd.core import ensemble
    from eks_amd.multicam_smoother import ensemble_kalman_smoother_multicam
    T, K, V, M = 50_000, 4, 2, 5
    mk = synth.multicam_markers(T, K, V=V, M=M, seed=4)
    ma = MarkerArray(mk.astype(np.float64), data_fields=['x', 'y', 'likelihood'])
    names, cams = [f'paw{k}' for k in range(K)], ['top', 'bot']
    dfs, s, df_3d = ensemble_kalman_smoother_multicam(ma, names, cams, smooth_param=10.0,
                                                      quantile_keep_pca=95.0, n_latent=3)
    assert len(dfs) == V and dfs[0].shape == (T, K * 9) and df_3d.shape == (T, K * 6) and np.all(s == 10.0)
    # oracle with the same float32 ensemble statistics (the ensemble kernel has its own parity
    # test): isolates centring + percentile mask + PCA set-up + 3-D Kalman path + reprojection
    arrs = orc.multicam_arrays(mk, quantile_keep_pca=95.0, n_latent=3, pca_fit=_sk, ens=ensemble(ma).array)
    for key in ('ys', 'ensemble_vars'):
        arrs[key] = arrs[key].astype(np.float32).astype(np.float64)      # the boundary is float32
    Rd = orc.build_R_from_vars(np.swapaxes(arrs['ensemble_vars'], 0, 1))
    ms, Vs, _ = c_oracle.smooth(arrs['ys'], Rd, arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'],
                                np.full(K, 10.0), nthreads=_threads())
    cams_o, lat_o = orc.multicam_outputs(arrs, ms, Vs)
    for c in range(V):
        ref = cams_o[c]
        assert (np.abs(dfs[c].values - ref) / np.abs(ref).max(axis=0)).max() < TOL
    lat = df_3d.values.reshape(T, K, 6)
    lat_o = lat_o.reshape(T, K, 6)
    assert (np.abs(lat[..., :3] - lat_o[..., :3]) / np.abs(lat_o[..., :3]).max(axis=(0, 2), keepdims=True)).max() < TOL
    assert (np.abs(lat[..., 3:] - lat_o[..., 3:]) / lat_o[..., 3:]).max() < TOL          # elementwise

    # the reference's default (smooth_param=None -> Adam, eks/core.py:562-699) on the same session,
    # loss on a 6 000-frame crop (s_frames): the device optimiser must reproduce the oracle's
    # float64 Adam trajectory - same stopping iteration, |d log s| <= 1e-3
    crop = [(0, 6000)]
    _, s_a, _ = ensemble_kalman_smoother_multicam(ma, names, cams, quantile_keep_pca=95.0, n_latent=3,
                                                  s_frames=crop)
    s_o = _oracle_adam_c(arrs, crop)
    assert np.abs(np.log(s_a) - np.log(s_o)).max() < 1e-3


def _oracle_adam_c(arrs, s_frames):
    """The reference's optimiser (oracle/eks_oracle.py: adam_optimize_s, eks/core.py:652-681) with
    the loss and d loss / d log s from the C port by complex step (oracle/eks_oracle.c) - the
    NumPy filter would need minutes at these lengths."""
    ys = arrs['ys']
    K = ys.shape[0]
    Rd = orc.build_R_from_vars(np.swapaxes(arrs['ensemble_vars'], 0, 1))
    y_c = [orc.crop_frames(ys[k], s_frames) for k in range(K)]
    R_c = [orc.constant_R_from_timevarying(orc.crop_frames(Rd[k], s_frames)) for k in range(K)]
    guess = [orc.compute_initial_guess(arrs['ensemble_vars'][:, k]) for k in range(K)]
    u0 = np.array([np.float32(np.log(np.clip(g, 1e-6, 1e3))) for g in guess], dtype=np.float64)

    def loss_and_grad(u):
        L, G = np.empty(K), np.empty(K)
        zero = np.zeros((1,) + arrs['As'][0].shape)
        for k in range(K):
            sQ = np.exp(u[k]) * arrs['Qs'][k]
            L[k], g = c_oracle.nll_directional(y_c[k], R_c[k], arrs['m0s'][k], arrs['S0s'][k], arrs['As'][k],
                                               arrs['Cs'][k], sQ, zero, sQ[None])
            G[k] = g[0]
        return L, G

    u, _, _ = orc.adam_optimize_s(loss_and_grad, u0)
    return np.exp(np.clip(u, -8.0, 8.0))


# ------------------------------------------------------------------------------------------
def test_c5_share_128_sessions_x_50k_x_32_keypoints_batched():
    """One GPU's share of configs[4]: 128 independent sessions of 50 000 frames x 32 keypoints,
    stacked by smooth_sessions_batched into ONE 4096-keypoint launch sequence (a single process is
    a world of one; the 8-GPU job runs this per rank)."""
    from eks_amd import hip_ops, synth
    from eks_amd.distributed import smooth_sessions_batched
    T, KS, NS = 50_000, 32, 128
    dev = hip_ops.require_gpu()
    eye = np.tile(np.eye(2), (KS, 1, 1))
    loaded = {}

    def load(i):
        y, var = synth.singlecam_observations_torch(T, KS, seed=5000 + i, device=dev)
        S0 = eye * y.double().var(dim=0, unbiased=False).cpu().numpy()[:, :, None]
        loaded[i] = S0
        return dict(ys=y.transpose(0, 1), m0s=np.zeros((KS, 2)), S0s=S0, As=eye, Cs=eye, Qs=eye,
                    ensemble_vars=var)

    calls = []
    from eks_amd.core import run_kalman_smoother

    def counted(**kw):
        calls.append(tuple(kw['ys'].shape))
        return run_kalman_smoother(**kw)

    s_of = lambda i: 2.0 + 0.25 * (i % 8)                 # a different fixed s per session
    # the full (K,T,2,2) covariance contract, as `bench.py --workload c5` times it
    mine, all_s = smooth_sessions_batched(load, NS, smooth_fn=counted, smooth_param=s_of, return_device=True)
    assert calls == [(NS * KS, T, 2)]                     # ONE batch of 4096 keypoints
    assert sorted(mine) == list(range(NS)) and len(all_s) == NS
    for i in range(NS):
        assert np.all(all_s[i] == s_of(i)) and tuple(mine[i][1].shape) == (KS, T, 2)
        assert tuple(mine[i][2].shape) == (KS, T, 2, 2)
        assert bool(torch.isfinite(mine[i][1]).all())
        assert bool((torch.diagonal(mine[i][2], dim1=2, dim2=3) > 0).all())
        assert bool((mine[i][2][..., 0, 1] == 0).all()) and bool((mine[i][2][..., 1, 0] == 0).all())
    # one keypoint of EVERY session (128 keypoints spread over the whole batch) against the
    # oracle, all frames
    for i in range(NS):
        kp = np.array([(5 * i + 3) % KS])
        y, var = synth.singlecam_observations_torch(T, KS, seed=5000 + i, device=dev)
        y_s = np.transpose(y[:, kp].cpu().numpy().astype(np.float64), (1, 0, 2)).copy()
        Rd = np.clip(np.transpose(var[:, kp].cpu().numpy().astype(np.float64), (1, 0, 2)), 1e-12, None)
        ms_o, Vs_o, _ = c_oracle.smooth(y_s, Rd, np.zeros((len(kp), 2)), loaded[i][kp], eye[kp], eye[kp], eye[kp],
                                        np.full(len(kp), s_of(i)), nthreads=1)
        kp_d = torch.as_tensor(kp, device=dev)
        ms_g = mine[i][1].index_select(0, kp_d).cpu().numpy().astype(np.float64)
        Vs_g = mine[i][2].index_select(0, kp_d).cpu().numpy().astype(np.float64)
        assert _kp_rel(ms_g, ms_o) < TOL, i
        Vd_g, Vd_o = np.diagonal(Vs_g, axis1=2, axis2=3), np.diagonal(Vs_o, axis1=2, axis2=3)
        assert (np.abs(Vd_g - Vd_o) / Vd_o).max() < TOL, i                     # elementwise
