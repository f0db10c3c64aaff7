for i in 1 2; do for u in 1 0; do echo "== c5 unfused=$u"; EKS_SMOOTH_UNFUSED=$u python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload c5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('ms_per_step %.4f' % o['ms_per_step'], {k: round(v, 4) for k, v in r['stage_avg_ms'].items()})
"; done; done
python - <<'PY'
import os, time, torch, sys
sys.path.insert(0, '.')
from eks_amd import _lib, hip_ops, synth
dev = hip_ops.require_gpu()
flags = _lib.FLAG_DIAG_MODEL | _lib.FLAG_UNIT_AC
for T, K in ((50000, 1024), (50000, 2048), (50000, 4096), (20000, 8192)):
    y, var = synth.singlecam_observations_torch(T, K, seed=1, device=dev)
    eye = torch.eye(2, dtype=torch.float64, device=dev).expand(K, 2, 2).contiguous()
    m0 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
    s = torch.full((K,), 10.0, dtype=torch.float64, device=dev)
    ms = torch.empty((T, K, 2), dtype=torch.float32, device=dev); Vs = torch.empty((T, K, 2, 2), dtype=torch.float32, device=dev)
    out = []
    for u in ('1', '0', '1', '0'):
        os.environ['EKS_SMOOTH_UNFUSED'] = u
        f = lambda: hip_ops.smooth(y, var, m0, eye, eye, eye, eye, s, flags=flags, out=(ms, Vs))
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t0) / 20)
    print(f'T={T} K={K}: unfused {out[0]:.4f} {out[2]:.4f} ms  fused {out[1]:.4f} {out[3]:.4f} ms')
    del y, var, ms, Vs
PY
