cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/k16p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k16p -- python3 /root/repo/tools/k16_probe.py > /tmp/k16.log 2>&1
cat /tmp/k16.log | grep " ms"
f=$(find /tmp/k16p -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
