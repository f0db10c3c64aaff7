#!/bin/bash
# kernel and copy statistics of a driver under rocprofv3: tools/driver_prof.sh [multicam|multicam_inflate|multicam_adam|singlecam|singlecam_adam] ...
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in "${@:-multicam}"; do
rm -rf /tmp/drvp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/drvp -- python3 $R/tools/driver_prof.py $w > /tmp/drvp.log 2>&1
grep " ms" /tmp/drvp.log | tail -3
for f in $(find /tmp/drvp -name "*memory_copy_stats.csv" -o -name "*kernel_stats.csv"); do
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[1].split('_', 1)[-1])
for r in rows[:10]:
    print(f"  {r['Name'][:96]:96s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
done
done
