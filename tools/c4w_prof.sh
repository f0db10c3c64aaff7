#!/bin/bash
# per-launch durations of the wide dense path (bench.py --workload c4w) from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/c4wp
rocprofv3 --kernel-trace --output-format csv -d /tmp/c4wp -- python3 $R/bench.py --workload c4w --no-cpu-baseline --steps 20 --warmup 3 --no-kernel-events > /tmp/c4wp.log 2>&1
f=$(find /tmp/c4wp -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'eks::d' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last full step: the launches after the last-but-one dwide_replay
idx = [i for i, r in enumerate(rows) if 'replay' in r['Kernel_Name']]
step = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
for r in step:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{r['Kernel_Name'][:60]:60s} grid={r.get('Grid_Size_X', r.get('Grid_Size','')):>8s} start={(a - t0) / 1e3:8.1f} dur={(b - a) / 1e3:8.1f} us")
PY
