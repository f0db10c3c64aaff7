cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for l in eks_amd/lib dc24 dc16 dc40; do
  if [ "$l" = "eks_amd/lib" ]; then L=$R/eks_amd/lib/libeks_hip.so; else L=$R/build_alt/$l/libeks_hip.so; fi
  EKS_HIP_LIB=$L python3 $R/bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline']['stage_avg_ms']; print('$l', round(d['ms_per_step'],4), {k: round(1e3*v,1) for k,v in s.items() if k.startswith('diag_s') or k=='diag_replay'})"
done; done
