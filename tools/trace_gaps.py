"""Reads a rocprofv3 kernel-trace CSV and prints, for the LAST `n` kernels, name, duration and the idle gap in front of
each (device timeline of one bench step): python tools/trace_gaps.py trace.csv [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-n:]
prev = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print(f"gap {gap:8.1f} us  dur {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:90]}")
    prev = e
