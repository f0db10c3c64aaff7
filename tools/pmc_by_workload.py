"""Summarise one workload's rocprofv3 passes (tools/pmc_by_workload.sh): per eks kernel the average duration, the
bytes read (2 x FETCH_SIZE KiB on gfx950, MI355X_MICROARCH.md) and written, and the implied rate."""
import collections, csv, glob, sys

wl, d_t, d_f, d_w = sys.argv[1:5]


def one(d, pat):
    f = glob.glob(f'{d}/**/*{pat}', recursive=True)
    return f[0] if f else None


def counter(d, name):
    out = collections.defaultdict(list)
    f = one(d, 'counter_collection.csv')
    if f:
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name:
                out[r['Kernel_Name']].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch, write = counter(d_f, 'FETCH_SIZE'), counter(d_w, 'WRITE_SIZE')
print(f'# workload {wl}: kernel, calls, avg us, read MB (2 x FETCH_SIZE), written MB, (read + written) / time')
f = one(d_t, 'kernel_stats.csv')
for r in csv.DictReader(open(f)) if f else []:
    k = r['Name']
    if 'eks::' not in k:
        continue
    us = float(r['AverageNs']) / 1e3
    rd = 2 * fetch.get(k, float('nan')) * 1024 / 1e6
    wr = write.get(k, float('nan')) * 1024 / 1e6
    print(f'{k[:90]:90s} {int(r["Calls"]):5d} {us:9.1f} us  read {rd:9.2f} MB  written {wr:9.2f} MB  {(rd + wr) / us:6.2f} TB/s')
