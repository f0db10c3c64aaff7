# A/B of alternative builds (tools/build_alt.sh) inside a bench step, same box, alternating runs:
#   tools/lib_ab.sh default NAME ...        (WORKLOAD=c4 tools/lib_ab.sh default NAME for another workload)
W=${WORKLOAD:-c3}
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = default ]; then L=""; else L="EKS_HIP_LIB=build_alt/$v/libeks_hip.so"; fi
  echo -n "$v: "; env $L python bench.py --workload $W --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k: round(v*1e3,1) for k,v in d['roofline']['stage_avg_ms'].items()})"
done; done
