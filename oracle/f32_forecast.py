"""Forecast of the day the oracle can be pinned - TEST INFRASTRUCTURE.

The reference's only numeric pins are golden CSVs compared at `assert_allclose(rtol=0, atol=1e-4)`
(/root/reference tests/conftest.py:86-101) and produced by its float32 dynamax recursion
(eks/core.py:290, :469, :648).  This build's oracle (and kernels) compute the same model
in float64 / cancellation-free float32.  Would the float64 oracle pass upstream's tolerance on upstream's
own numbers?  Nobody can run upstream here - but its ARITHMETIC can be emulated: `eks_oracle.
kalman_smoother_f32` runs dynamax's operation order (covariance form `P - K S K^T`, Cholesky solves with
the 1e-9 boost, symmetrisation, float32 running log-likelihood) entirely in float32.

`forecast(golden_dir)` runs, on the reference's own recordings (the committed fixtures under tests/golden:
data/ibl-pupil, data/mirror-mouse, data/fly with its calibration) and with the configurations of the
reference's integration tests (tests/integration/test_{singlecam,mirrored_multicam,ibl_pupil,multicam}.py),
  (a) the float64 oracle and
  (b) the float32 emulation on the same (float32-rounded) inputs, output tables assembled in float32,
and reports per case and output label max |(b) - (a)| - to be read against atol = 1e-4.  The optimiser
(smooth_param=None) is not emulated: its float32 stop test is chaotic; the default-mode
cases are compared at the oracle's own s ("loose on s, strict on outputs given s").

Not a bitwise twin of XLA (fusion and FMA contraction differ in the last bit of every operation): the
numbers are the MAGNITUDE of upstream's rounding, which is what a pass / fail forecast needs.
"""
from __future__ import annotations

import ast
import os

import numpy as np

from oracle import ekf_oracle as ek
from oracle import eks_oracle as orc

ATOL = 1e-4                                  # /root/reference tests/conftest.py:95-100
LABELS = ('x', 'y', 'likelihood', 'x_ens_median', 'y_ens_median', 'x_ens_var', 'y_ens_var',
          'x_posterior_var', 'y_posterior_var')


def _f32(a):
    return np.asarray(a, np.float32)


def _per_label(a32, a64, n_labels):
    """(T, K * n_labels) tables -> {label index: max |difference|}"""
    d = np.abs(np.asarray(a32, np.float64) - a64).reshape(a64.shape[0], -1, n_labels)
    return d.max(axis=(0, 1))


def _linear_case(arrs, s, outputs, kind):
    """float64 oracle vs float32 emulation of one run_kalman_smoother call + output assembly."""
    Rd = np.clip(np.swapaxes(arrs['ensemble_vars'], 0, 1), 1e-12, None)              # eks/utils.py:373
    args = (arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'])
    ms64, Vs64, nll64 = orc.kalman_smoother(*args, s, Rd)
    ms32, Vs32, nll32 = orc.kalman_smoother_f32(*args, s, Rd)
    a32 = dict(arrs)
    for k in ('Cs', 'means', 'ens', 'ensemble_vars'):
        a32[k] = _f32(arrs[k])
    if kind == 'singlecam':
        o64, o32 = outputs(arrs, s, ms64, Vs64), outputs(a32, s, ms32, Vs32)
        tables = {'table': (o32, o64)}
    else:
        (c64, l64), (c32, l32) = outputs(arrs, ms64, Vs64), outputs(a32, ms32, Vs32)
        tables = {f'cam{c}': (c32[c], c64[c]) for c in range(len(c64))}
    rows = {}
    for name, (o32, o64) in tables.items():
        rows[name] = _per_label(o32, o64, 9)
    state = dict(ms=float(np.abs(ms32 - ms64).max()), Vs=float(np.abs(Vs32 - Vs64).max()),
                 Vs_rel=float((np.abs(Vs32 - Vs64) / np.abs(Vs64).max(axis=(1, 2, 3), keepdims=True)).max()),
                 nll_rel=float((np.abs(nll32 - nll64) / np.abs(nll64)).max()))
    return rows, state


def _sk_pca(X, n):
    from sklearn.decomposition import PCA
    p = PCA(n_components=n).fit(X)
    return p.components_, p.mean_


def _fly_problem(g):
    """data/fly as tools/make_golden.py: fly() sets it up (calibration parsed from the stored TOML text)."""
    cams, cur = [], None
    for line in str(g['toml']).split('\n'):
        line = line.strip()
        if line.startswith('[cam_'):
            cur = {}
            cams.append(cur)
        elif line.startswith('['):
            cur = None
        elif cur is not None and '=' in line:
            k, v = line.split('=', 1)
            cur[k.strip()] = ast.literal_eval(v.strip())
    ocams = [dict(rot=np.array(c['rotation'], float), tvec=np.array(c['translation'], float),
                  K=np.array(c['matrix'], float), dist=np.array(c['distortions'], float)) for c in cams]
    mk = g['markers']
    st = orc.ensemble(mk)[0]                                                          # (V,T,K,5)
    V, T, K = st.shape[:3]
    ys = np.transpose(st[..., 0:2], (2, 1, 0, 3)).reshape(K, T, 2 * V)
    evs = np.transpose(st[..., 2:4], (2, 1, 0, 3)).reshape(K, T, 2 * V)
    return ocams, st, ys, evs


def _ekf_case(ocams, st, ys, evs, m0s, S0s, Qs, s, kk):
    """Calibrated path on keypoints kk: float64 sequential extended filter + RTS vs the float32 emulation with the
    projection evaluated in float32 and its Jacobian rounded to float32 (jax.jacfwd in float32 is the analytic
    derivative evaluated in float32: within a few ulps of that)."""
    V, T = st.shape[0], st.shape[1]
    heads64 = [ek.make_projection_fn(c['rot'], c['tvec'], c['K'], c['dist']) for c in ocams]
    heads32 = [ek.make_projection_fn(c['rot'], c['tvec'], c['K'], c['dist'], dtype=np.float32) for c in ocams]
    h64, h32 = ek.combine_projections(heads64), ek.combine_projections(heads32)
    f32in = lambda a: np.asarray(a, np.float32).astype(np.float64)                    # inputs rounded as upstream
    eye = np.tile(np.eye(3), (len(kk), 1, 1))
    Rd = np.clip(f32in(evs[kk]), 1e-12, None)
    ms64 = np.empty((len(kk), T, 3))
    Vs64 = np.empty((len(kk), T, 3, 3))
    for j, k in enumerate(kk):
        ms64[j], Vs64[j], _ = ek.eks_smoother(f32in(ys[k]), Rd[j], m0s[k], S0s[k], eye[j], Qs[k], s[j], h64)

    def emission(m):
        yhat = h32(m)
        H = np.stack([ek.jacobian_cs(h64, mm.astype(np.float64)) for mm in m]).astype(np.float32)
        return yhat.astype(np.float32), H

    ms32, Vs32, _ = orc.kalman_smoother_f32(f32in(ys[kk]), m0s[kk], S0s[kk], eye, None, Qs[kk], s, Rd,
                                            emission=emission)
    rows = {}
    for c in range(V):
        t64 = np.empty((T, len(kk), 9))
        t32 = np.empty((T, len(kk), 9))
        for j, k in enumerate(kk):
            t64[:, j, 0:2] = heads64[c](ms64[j])
            t32[:, j, 0:2] = heads32[c](ms32[j])
            t64[:, j, 7], t64[:, j, 8] = ek.project_3d_covariance_to_2d(ms64[j], Vs64[j], heads64[c], evs[k])
            J = np.stack([ek.jacobian_cs(heads64[c], m.astype(np.float64)) for m in ms32[j]]).astype(np.float32)
            cov = J @ Vs32[j] @ np.swapaxes(J, 1, 2)
            t32[:, j, 7] = cov[:, 0, 0] + _f32(evs[k][:, 0])
            t32[:, j, 8] = cov[:, 1, 1] + _f32(evs[k][:, 1])
            for tt in (t64, t32):
                tt[:, j, 2] = st[c, :, k, 4]
                tt[:, j, 3:5] = st[c, :, k, 0:2]
                tt[:, j, 5:7] = st[c, :, k, 2:4]
        rows[f'cam{c}'] = _per_label(t32.reshape(T, -1), t64.reshape(T, -1), 9)
    state = dict(ms=float(np.abs(ms32 - ms64).max()), Vs=float(np.abs(Vs32 - Vs64).max()),
                 Vs_rel=float((np.abs(Vs32 - Vs64) / np.abs(Vs64).max(axis=(1, 2, 3), keepdims=True)).max()))
    return rows, state


def forecast(golden_dir, cases=None):
    """Returns a list of dicts: case, config, table, per-label max |float32 emulation - float64 oracle|, the
    labels that exceed upstream's atol, and the state-level differences."""
    out = []

    def add(case, config, rows, state):
        for name, d in rows.items():
            out.append(dict(case=case, config=config, table=name,
                            max_abs_diff={LABELS[i]: float(v) for i, v in enumerate(d)},
                            exceeds_atol=[LABELS[i] for i, v in enumerate(d) if v > ATOL], state=state))

    want = lambda c: cases is None or c in cases
    if want('singlecam'):
        g = np.load(os.path.join(golden_dir, 'ibl_pupil_singlecam.npz'))
        arrs = orc.singlecam_arrays(g['markers'])                 # tests/integration/test_singlecam.py:4-20
        add('ibl-pupil singlecam', 'smooth_param=[10.0]',
            *_linear_case(arrs, np.full(4, 10.0), orc.singlecam_outputs, 'singlecam'))
        add('ibl-pupil singlecam', "defaults (outputs at the oracle's optimised s)",
            *_linear_case(arrs, np.asarray(g['adam_s'], float), orc.singlecam_outputs, 'singlecam'))
    if want('mirrored'):
        g = np.load(os.path.join(golden_dir, 'mirror_mouse_multicam.npz'))
        # tests/integration/test_mirrored_multicam.py:4-30: paw1LH, paw2LF; top, bot; quantile 95; inflation on
        kp = [list(g['keypoints']).index(k) for k in ('paw1LH', 'paw2LF')]
        arrs = orc.multicam_arrays(g['markers'][:, :, :, kp], quantile_keep_pca=95.0, n_latent=3, pca_fit=_sk_pca,
                                   inflate_vars=True)
        add('mirror-mouse mirrored multicam', 'smooth_param=[10.0], quantile 95, inflate_vars',
            *_linear_case(arrs, np.full(2, 10.0), orc.multicam_outputs, 'multicam'))
        s_opt = orc.run_kalman_smoother(arrs['ys'], arrs['m0s'], arrs['S0s'], arrs['As'], arrs['Cs'], arrs['Qs'],
                                        arrs['ensemble_vars'])[0]
        add('mirror-mouse mirrored multicam', "defaults (outputs at the oracle's optimised s), quantile 95, inflate_vars",
            *_linear_case(arrs, s_opt, orc.multicam_outputs, 'multicam'))
    if want('pupil'):
        g = np.load(os.path.join(golden_dir, 'ibl_pupil_singlecam.npz'))
        gp = np.load(os.path.join(golden_dir, 'ibl_pupil_pupil.npz'))
        pa = orc.pupil_arrays(g['markers'][:, :, :, gp['order']])  # tests/integration/test_ibl_pupil.py
        Rd = np.clip(pa['ensemble_vars'], 1e-12, None)[None]
        for label, sp in (('smooth_params=[0.99, 0.99]', gp['fixed_s']),
                          ("defaults (outputs at the oracle's optimised parameters)", gp['adam_s'])):
            A, Q = orc.pupil_dynamics(sp[0], sp[1], pa['latent_vars'])
            args = (pa['ys'][None], pa['m0'][None], pa['S0'][None], A[None], pa['C'][None], Q[None], np.ones(1), Rd)
            ms64, Vs64, _ = orc.kalman_smoother(*args)
            ms32, Vs32, _ = orc.kalman_smoother_f32(*args)
            o64 = orc.pupil_outputs(pa, ms64[0], Vs64[0])
            p32 = {k: (_f32(v) if isinstance(v, np.ndarray) else np.float32(v)) for k, v in pa.items()}
            o32 = orc.pupil_outputs(p32, ms32[0], Vs32[0])
            state = dict(ms=float(np.abs(ms32 - ms64).max()), Vs=float(np.abs(Vs32 - Vs64).max()),
                         Vs_rel=float((np.abs(Vs32 - Vs64) / np.abs(Vs64).max(axis=(1, 2, 3), keepdims=True)).max()))
            add('ibl-pupil pupil smoother', label, {'table': _per_label(o32, o64, 9)}, state)
    if want('fly'):
        g = np.load(os.path.join(golden_dir, 'fly_calibrated_multicam.npz'))
        ocams, st, ys, evs = _fly_problem(g)
        kk = [0, 1]                                               # tests/integration/test_multicam.py:32-58: L1A, L1B
        evs_i = evs.copy()
        evs_i[kk] = g['infl_vars'].astype(np.float64)             # quantile 95 + variance inflation, as stored
        add('fly calibrated multicam', 'smooth_param=[10.0], quantile 95, inflate_vars',
            *_ekf_case(ocams, st, ys, evs_i, g['m0s'], g['S0s'], g['Qs'], np.full(2, 10.0), kk))
        add('fly calibrated multicam', "defaults (outputs at the oracle's optimised s), quantile 95, inflate_vars",
            *_ekf_case(ocams, st, ys, evs_i, g['m0s'], g['S0s'], g['Qs'], np.asarray(g['infl_adam_s'], float), kk))
    return out


def format_table(rows):
    lines = [f'# float32 emulation of the upstream recursion vs the float64 oracle: max |difference| per output label, '
             f'to be read against upstream\'s atol = {ATOL:g} (tests/conftest.py:95-100)',
             f"{'case':34s} {'configuration':78s} {'table':6s} " + ' '.join(f'{l[:15]:>15s}' for l in LABELS)
             + '  exceeds atol']
    for r in rows:
        lines.append(f"{r['case']:34s} {r['config'][:78]:78s} {r['table']:6s} "
                     + ' '.join(f"{r['max_abs_diff'][l]:15.2e}" for l in LABELS)
                     + '  ' + (', '.join(r['exceeds_atol']) or '-'))
        st = r['state']
        lines.append(f"{'':34s}   state: max|ms32 - ms64| = {st['ms']:.2e}, max|Vs32 - Vs64| = {st['Vs']:.2e} "
                     f"({st['Vs_rel']:.2e} of the keypoint's largest covariance)"
                     + (f", NLL relative {st['nll_rel']:.2e}" if 'nll_rel' in st else ''))
    return '\n'.join(lines)


if __name__ == '__main__':
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    print(format_table(forecast(os.path.join(here, 'tests', 'golden'))))
