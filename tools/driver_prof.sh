#!/bin/bash
# kernel statistics of the multi-camera driver on configs[3] (fixed s) under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/drvp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/drvp -- python3 $R/tools/driver_prof.py > /tmp/drvp.log 2>&1
grep " ms" /tmp/drvp.log
for f in $(find /tmp/drvp -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv"); do
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(sys.argv[1].split('/')[-1])
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
done
