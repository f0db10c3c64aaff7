#!/usr/bin/env python
"""Turn gpurun_out/evidence/ (written by tools/collect_evidence.sh on the GPU box) into the committed
summaries profiles/<tag>_kernel_stats.txt, <tag>_pmc_summary.txt, <tag>_bench*.json and
profiles/<round>_traffic.json.  Usage: python tools/make_profiles.py r01_f"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ev = os.path.join(root, 'gpurun_out', 'evidence')
prof = os.path.join(root, 'profiles')
b = json.load(open(os.path.join(ev, 'bench_c3.json')))
def newest(pattern):
    """gpurun merges into gpurun_out/: earlier runs' pid-prefixed files may still be there"""
    return max(glob.glob(pattern), key=os.path.getmtime)


rows = list(csv.DictReader(open(newest(os.path.join(ev, 'stats', '*', '*_kernel_stats.csv')))))
out = [f'# {tag}: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 20 --warmup 3 '
       '--no-cpu-baseline --no-kernel-events',
       '# C3 workload: singlecam T=100000 x K=256, 64-candidate NLL grid + smooth (23 steps incl. warm-up).',
       f"{'kernel':92s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}"]
tot = 0.0
for r in rows:
    if 'eks::' in r['Name']:
        out.append(f"{r['Name'][:92]:92s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} "
                   f"{float(r['MinNs'])/1e3:10.1f} {float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):7.2f}")
        tot += float(r['AverageNs']) / 1e3
out.append(f"# sum of per-kernel averages = {tot:.1f} us per step (bench.py wall clock: {b['ms_per_step']*1e3:.1f} us per step)")
open(os.path.join(prof, f'{tag}_kernel_stats.txt'), 'w').write('\n'.join(out) + '\n')
# the Adam-mode session (c3adam): one pass for the lag sums, then the whole search in ONE launch (round 6)
try:
    rows_a = list(csv.DictReader(open(newest(os.path.join(ev, 'stats_c3adam', '*', '*_kernel_stats.csv')))))
    out_a = [f'# {tag}: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload c3adam --steps 5 '
             '--warmup 2 --no-cpu-baseline --no-kernel-events',
             '# singlecam T=100000 x K=256, smooth_param=None: 7 steps, each = eks_adam_prepare (lag_sums + lag_reduce), ONE',
             '# eks_adam_run call (lag_adam_kernel: the whole search, 55-118 iterations per keypoint) and the final smooth.',
             f"{'kernel':92s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}"]
    for r in rows_a:
        if 'eks::' in r['Name']:
            out_a.append(f"{r['Name'][:92]:92s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} "
                         f"{float(r['MinNs'])/1e3:10.1f} {float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):7.2f}")
    open(os.path.join(prof, f'{tag}_kernel_stats_c3adam.txt'), 'w').write('\n'.join(out_a) + '\n')
except (ValueError, OSError) as e:
    print('no c3adam stats:', e)


def load(d):
    return list(csv.DictReader(open(newest(os.path.join(ev, d, '*', '*_counter_collection.csv')))))


lines = [f'# {tag}: PMC counters from separate rocprofv3 --pmc passes of: python3 bench.py --steps 3 --warmup 1 '
         '--no-cpu-baseline --no-kernel-events',
         '# averages per dispatch.  FETCH_SIZE / WRITE_SIZE in KiB as reported; gfx950 correction: '
         'read bytes = 2 * FETCH_SIZE * 1024']
agg_all = {}
for d in ('pmc_sq', 'pmc_fetch', 'pmc_write'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in load(d):
        if 'eks::' in r['Kernel_Name']:
            nm = r['Kernel_Name'].split('(')[0].replace('void ', '')
            agg[nm][r['Counter_Name']].append(float(r['Counter_Value']))
    agg_all[d] = agg
    lines.append(f'== {d}')
    for k, v in sorted(agg.items()):
        lines.append(f"{k[:70]:70s} " + ' '.join(f'{c}={sum(x)/len(x):.4g}' for c, x in sorted(v.items())))
open(os.path.join(prof, f'{tag}_pmc_summary.txt'), 'w').write('\n'.join(lines) + '\n')
traffic = {}
for k in agg_all['pmc_fetch']:
    f = agg_all['pmc_fetch'][k]['FETCH_SIZE']
    w = agg_all['pmc_write'].get(k, {}).get('WRITE_SIZE', [0])
    traffic[k.replace('eks::', '').split('<')[0]] = int((2 * sum(f) / len(f) + sum(w) / len(w)) * 1024)
json.dump({'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 '
                     f'--warmup 1 --no-cpu-baseline --no-kernel-events; see {tag}_pmc_summary.txt',
           'correction': 'hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reads half of a '
                         'coalesced stream, MI355X_MICROARCH.md HBM section)',
           'kernel_sources_sha16': open(os.path.join(ev, 'kernel_sources_sha16.txt')).read().strip(),
           'hbm_bytes_per_launch': traffic}, open(os.path.join(prof, f"{tag.split('_')[0]}_traffic.json"), 'w'), indent=1)
for name in ('c3', 'c3_default_run', 'c3adam', 'c3adam_streaming', 'c3adam_all_chains_streamed', 'c3_legacy_nll', 'c3_nolag', 'c4', 'c4w', 'c4adam', 'c5', 'c2', 'pupil', 'ekf', 'c3_2ranks_gloo', 'c3_2ranks_gloo_strong'):
    src = os.path.join(ev, f'bench_{name}.json')
    if os.path.exists(src) and os.path.getsize(src):
        shutil.copy(src, os.path.join(prof, f'{tag}_bench_{name}.json'))
for name in ('pytest_gpu.txt', 'smoke.txt', 'adam_time.txt', 'dense_adam_time.txt', 'dense_adam_time_d.txt',
             'pupil_time.txt', 'ekf_time.txt', 'driver_time.txt', 'host_path_time.txt', 'first_call.txt', 'nll_lean2.txt', 'nll_lag.txt',
             'fit_time.txt', 'host_boundary_ab.txt', 'grid_stamps.txt', 'lag_adam_check.txt', 'lag_prepass_time.txt', 'lag_adam_shapes.txt', 'c3adam_prepare_order.txt',
             'ekf_chunk_trade.txt', 'c3adam_timeline.txt'):
    src = os.path.join(ev, name)
    if os.path.exists(src):
        shutil.copy(src, os.path.join(prof, f'{tag}_{name}'))
print(b['ms_per_step'], b['value'], b['roofline']['frac'], b['roofline']['stage_avg_ms'])
print(open(os.path.join(prof, f'{tag}_kernel_stats.txt')).read())
