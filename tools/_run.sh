timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py tests/test_gpu_drivers.py -x -q -m gpu 2>&1 | tail -5
for c in 1 0; do echo "== c2 coop=$c"; EKS_SMOOTH_COOP=$c timeout 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --workload c2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('ms_per_step %.4f' % o['ms_per_step'], {k: round(v, 4) for k, v in r['stage_avg_ms'].items()})
"; done
python - <<'PY'
import os, time, torch, sys
sys.path.insert(0, '.')
from eks_amd import _lib, hip_ops, synth
dev = hip_ops.require_gpu()
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
for T, K in ((10000, 64), (3000, 32), (20000, 64), (10000, 200), (50000, 30)):
    y, var = synth.singlecam_observations_torch(T, K, seed=1, device=dev)
    eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
    m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
    s = torch.full((K,), 10.0, dtype=torch.float64, device=dev)
    ms = torch.empty((T, K, 2), dtype=torch.float32, device=dev); Vs = torch.empty((T, K, 2), dtype=torch.float32, device=dev)
    out = []
    for u in ('0', '1'):
        os.environ['EKS_SMOOTH_COOP'] = u
        f = lambda: hip_ops.smooth(y, var, m0, eye, eye, eye, eye, s, flags=flags, vs_diag=True, out=(ms, Vs))
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): f()
        torch.cuda.synchronize(); out.append(1e6 * (time.perf_counter() - t0) / 200)
    print(f'T={T} K={K}: three launches {out[0]:.1f} us  one launch {out[1]:.1f} us')
PY
