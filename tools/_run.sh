b() { echo "== $*"; env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('ms_per_step %.4f' % o['ms_per_step'], {k: round(v, 4) for k, v in r['stage_avg_ms'].items()})
"; }
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
b A=1
b A=2
b A=3
for w in c5 c2; do echo "== $w"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload $w 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('ms_per_step %.4f' % o['ms_per_step'], {k: round(v, 4) for k, v in r['stage_avg_ms'].items()})
"; done
