"""Maximum sizes: problems of MORE THAN 2^31 ELEMENTS per array (the kernels index with 64-bit offsets and 32-bit
buffer offsets per chunk; nothing at the BASELINE sizes gets near the 32-bit element range - configs[4]'s per-GPU share
is 4.1e8).  No oracle finishes at this size, so the check is the size-independent property the domain offers: keypoints
are independent (reference eks/core.py:293, the vmap over keypoints), so any subset of keypoints run as its own small
problem must reproduce the corresponding slice of the big one - and the small problem's kernels are the ones the
oracle parity tests pin.  Frames beyond t = 2^31 / N lie above the 32-bit element range for every chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip('torch')


def _need_gb(gb):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    if torch.cuda.get_device_properties(0).total_memory < gb * (1 << 30):
        pytest.skip(f'needs {gb} GB of device memory')


def _subsets(K, w=32):
    return [slice(0, w), slice(K // 2 - w // 2, K // 2 + w // 2), slice(K - w, K)]


def test_scalar_chains_above_2_31_elements_reproduce_their_keypoint_subsets():
    """T = 40 000 x K = 32 768 keypoints (65 536 chains): 2.62e9 elements in y, var, ms; 5.2e9 in Vs.  Median,
    64-candidate NLL grid (the grid kernel), argmin, per-keypoint loss + gradient (the Adam evaluation) and the
    smoother, each against separate runs of three 32-keypoint subsets.  The exact median and the indices must be
    bit-identical; float32 kernels whose chunk geometry depends on the width agree to their rounding (1e-6 of
    the gross size of the quantity), the smoother - same chunks at every width - bit for bit."""
    _need_gb(150)
    from eks_amd import hip_ops
    T, K = 40_000, 32_768
    assert T * K * 2 > 2 ** 31
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(20)
    y = torch.randn((T, K, 2), generator=g, device=dev)
    y.mul_(1.5).add_(torch.linspace(-40.0, 40.0, T, device=dev)[:, None, None])
    var = torch.rand((T, K, 2), generator=g, device=dev).mul_(0.6).add_(0.05)
    var[::97] *= 50.0                                           # occluded frames
    eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
    m0 = torch.zeros((K, 2), dtype=torch.float64, device=dev)
    S0 = eye * 30.0
    flags = hip_ops.model_flags(S0[:4].cpu().numpy(), eye[:4].cpu().numpy(), eye[:4].cpu().numpy(), eye[:4].cpu().numpy())
    cand = torch.exp(torch.linspace(-6.0, 6.0, 64, dtype=torch.float64, device=dev))

    rc = hip_ops.const_r(var, 1e-4)
    nll = hip_ops.nll(y, rc, m0, S0, eye, eye, eye, cand, flags=flags)
    s_best, idx = hip_ops.argmin_s(nll, cand)
    s_kp = torch.exp(torch.linspace(-5.0, 5.0, K, dtype=torch.float64, device=dev))[:, None].contiguous()
    val, grad = hip_ops.nll(y, rc, m0, S0, eye, eye, eye, s_kp, per_keypoint=True, want_grad=True, flags=flags)
    ms, Vs = hip_ops.smooth(y, var, m0, S0, eye, eye, eye, s_best, flags=flags)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(nll).all()) and bool(torch.isfinite(ms[-1]).all()) and bool(torch.isfinite(Vs[-1]).all())

    gross = T * (torch.log(rc).abs() + 1.0).sum(dim=1, keepdim=True)        # size of the NLL's parts (fuzz_parity.py)
    for sel in _subsets(K):
        ys, vs = y[:, sel].contiguous(), var[:, sel].contiguous()
        k = ys.shape[1]
        rc_s = hip_ops.const_r(vs, 1e-4)
        assert torch.equal(rc_s, rc[sel]), 'median'
        nll_s = hip_ops.nll(ys, rc_s, m0[:k], S0[:k], eye[:k], eye[:k], eye[:k], cand, flags=flags)
        err = ((nll_s - nll[sel]).abs() / torch.maximum(nll_s.abs(), 1e-2 * gross[sel])).max().item()
        assert err < 2e-6, ('nll grid', sel, err)
        sb, ib = hip_ops.argmin_s(nll[sel].contiguous(), cand)
        assert torch.equal(ib, idx[sel]) and torch.equal(sb, s_best[sel])
        v_s, g_s = hip_ops.nll(ys, rc_s, m0[:k], S0[:k], eye[:k], eye[:k], eye[:k], s_kp[sel].contiguous(),
                               per_keypoint=True, want_grad=True, flags=flags)
        err = ((v_s - val[sel]).abs() / torch.maximum(v_s.abs(), 1e-2 * gross[sel])).max().item()
        assert err < 2e-6, ('loss', sel, err)
        gerr = ((g_s - grad[sel]).abs() / torch.maximum(g_s.abs(), 1e-3 * gross[sel])).max().item()
        assert gerr < 2e-5, ('gradient', sel, gerr)
        ms_s, Vs_s = hip_ops.smooth(ys, vs, m0[:k], S0[:k], eye[:k], eye[:k], eye[:k], s_best[sel].contiguous(),
                                    flags=flags)
        assert torch.equal(ms_s, ms[:, sel]), ('ms', sel, (ms_s - ms[:, sel]).abs().max().item())
        assert torch.equal(Vs_s, Vs[:, sel]), ('Vs', sel)


def test_general_path_above_2_31_elements_reproduces_its_keypoint_subsets():
    """T = 50 000 x K = 8 192 keypoints, D = 3, O = 8 (3.3e9 elements in y and var, 3.7e9 in Vs) on the wide
    general-path kernels; subsets of 64 keypoints run on the narrow-session kernels (other chunking, same float64
    algebra): float32 outputs within 2e-6 of the per-keypoint scale."""
    _need_gb(150)
    from eks_amd import hip_ops
    T, K, D, O = 50_000, 8_192, 3, 8
    assert T * K * O > 2 ** 31 and T * K * D * D > 2 ** 31
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(21)
    rng = np.random.default_rng(21)
    C1 = rng.standard_normal((O, D)) / np.sqrt(D)
    A1 = np.eye(D) * 0.995 + 0.002 * rng.standard_normal((D, D))
    t64 = lambda a: torch.as_tensor(a, dtype=torch.float64, device=dev)
    C = (t64(C1)[None] * torch.linspace(0.7, 1.3, K, dtype=torch.float64, device=dev)[:, None, None]).contiguous()
    A = t64(A1).expand(K, D, D).contiguous()
    Q = (t64(np.eye(D)) * torch.linspace(0.5, 2.0, K, dtype=torch.float64, device=dev)[:, None, None]).contiguous()
    S0 = t64(np.eye(D) * 4.0).expand(K, D, D).contiguous()
    m0 = torch.zeros((K, D), dtype=torch.float64, device=dev)
    s = torch.exp(torch.linspace(-3.0, 3.0, K, dtype=torch.float64, device=dev))
    y = torch.randn((T, K, O), generator=g, device=dev).mul_(2.0)
    y.add_(torch.sin(torch.linspace(0.0, 60.0, T, device=dev))[:, None, None] * 10.0)
    var = torch.rand((T, K, O), generator=g, device=dev).mul_(0.8).add_(0.05)
    ms, Vs = hip_ops.smooth(y, var, m0, S0, A, C, Q, s)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(ms[-1]).all()) and bool(torch.isfinite(Vs[-1]).all())
    for sel in _subsets(K, 64):
        ms_s, Vs_s = hip_ops.smooth(y[:, sel].contiguous(), var[:, sel].contiguous(), m0[sel].contiguous(),
                                    S0[sel].contiguous(), A[sel].contiguous(), C[sel].contiguous(),
                                    Q[sel].contiguous(), s[sel].contiguous())
        sc = ms_s.abs().amax(dim=(0, 2), keepdim=True)
        assert ((ms_s - ms[:, sel]).abs() / sc).max().item() < 2e-6, ('ms', sel)
        scv = Vs_s.abs().amax(dim=(0, 2, 3), keepdim=True)
        assert ((Vs_s - Vs[:, sel]).abs() / scv).max().item() < 2e-6, ('Vs', sel)
