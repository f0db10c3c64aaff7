# A/B of builds on one box, alternating runs: bash tools/_ab.sh <libA> <libB> ...
for r in 1 2; do for lib in "$@"; do
EKS_HIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('roofline',{}).get('stage_avg_ms',{})
print('%-24s %.4f ' % ('$lib'.split('/')[-1], d['ms_per_step']), {a:round(b,4) for a,b in k.items()})"
done; done
