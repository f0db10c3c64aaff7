#!/bin/bash
# kernel durations of the Adam loop on C3 under rocprofv3 (tools/adam_time.py), per variant:
#   persist | periter | unfused | <chunk length> (per-iteration launches with that EKS_NLL_GRAD_CHUNK)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in "$@"; do
  unset EKS_NLL_GRAD_UNFUSED EKS_NLL_GRAD_CHUNK EKS_ADAM_LAUNCH_PER_ITER
  case $c in
    persist) ;;
    periter) export EKS_ADAM_LAUNCH_PER_ITER=1;;
    unfused) export EKS_NLL_GRAD_UNFUSED=1;;
    *) export EKS_ADAM_LAUNCH_PER_ITER=1 EKS_NLL_GRAD_CHUNK=$c;;
  esac
  rm -rf /tmp/ap_$c
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ap_$c -- python3 $R/tools/adam_time.py > /tmp/ap_$c.log 2>&1
  echo "== $c"; grep "T=100000" /tmp/ap_$c.log
  f=$(find /tmp/ap_$c -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'eks::' in r['Kernel_Name'] and ('nll' in r['Kernel_Name'] or 'adam' in r['Kernel_Name'])]
# the second C3 repetition: the launches between the 2nd and 3rd const_r... simpler: group by grid size
by = collections.defaultdict(list)
for r in rows:
    by[(r['Kernel_Name'][:60], r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', ''))].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for (name, grid), v in by.items():
    d = [(b - a) / 1e3 for a, b in v]
    gaps = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
    gaps = [g for g in gaps if g < 500]
    print(f"{name:60s} grid={grid:>8s} n={len(d):4d} dur avg={sum(d)/len(d):8.1f} max={max(d):8.1f} sum={sum(d)/1e3:7.2f} ms   gap median={sorted(gaps)[len(gaps)//2] if gaps else 0:6.1f}")
PY
done
