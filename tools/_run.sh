b() { echo "== $*"; env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('ms_per_step %.4f' % o['ms_per_step'], {k: round(v, 4) for k, v in r['stage_avg_ms'].items()})
"; }
for i in 1 2 3; do
b EKS_NLL_ASSEMBLE_SEQ=1
b EKS_NLL_ASSEMBLE_SEQ=0
done
