b() { echo "== $*"; env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('ms_per_step %.4f' % o['ms_per_step'], {k: round(v, 4) for k, v in r['stage_avg_ms'].items()})
"; }
for i in 1 2 3; do
b EKS_HIP_LIB=$PWD/build_alt/libeks_hip_alt.so
b EKS_DUMMY=1
done
for i in 1 2; do
for l in build_alt/libeks_hip_alt.so eks_amd/lib/libeks_hip.so; do echo "== c5 $l"; EKS_HIP_LIB=$PWD/$l python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload c5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        o = json.loads(l); r = o['roofline']
        print('ms_per_step %.4f' % o['ms_per_step'], {k: round(v, 4) for k, v in r['stage_avg_ms'].items()})
"; done; done
