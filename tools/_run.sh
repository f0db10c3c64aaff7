timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q -m gpu -k "const_r or median or c3 or fuzz" 2>&1 | tail -3
python tools/fuzz_median.py 2>&1 | tail -3
python - <<'PY'
import os, time, torch, sys
sys.path.insert(0, '.')
from eks_amd import hip_ops, synth
dev = hip_ops.require_gpu()
y, var = synth.singlecam_observations_torch(100000, 256, seed=3, device=dev)
f = lambda: hip_ops.const_r(var, 1e-4)
for rep in range(3):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): f()
    torch.cuda.synchronize(); print('%.1f us per const_r' % (1e6 * (time.perf_counter() - t0) / 50))
PY
python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('ms_per_step %.4f' % o['ms_per_step'], {k: round(v, 4) for k, v in r['stage_avg_ms'].items()})
"
