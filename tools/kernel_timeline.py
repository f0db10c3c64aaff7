"""Print the last N kernel launches of a rocprofv3 --kernel-trace CSV as a timeline (start relative
to the first of them, duration, grid, VGPRs) - shows which launches overlapped.
    python tools/kernel_timeline.py <kernel_trace.csv> [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%-58s start %8.1f us  dur %7.1f us  grid %-8s wg %-5s vgpr %s' % (
        r['Kernel_Name'].replace('void eks::', '').replace('eks::', '')[:58], (a - t0) / 1e3, (b - a) / 1e3,
        r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?'), r.get('VGPR_Count', '?')))
