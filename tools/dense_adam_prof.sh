#!/bin/bash
# kernel statistics of the Adam loop on the general (D, O) path (tools/dense_adam_time.py) under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/tools/dense_adam_time.py 2>&1 | grep -v amdgpu.ids
rm -rf /tmp/dap
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dap -- python3 $R/tools/dense_adam_time.py > /tmp/dap.log 2>&1
f=$(find /tmp/dap -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
