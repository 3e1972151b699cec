"""Condense the rocprofv3 output of tools/profile_round.sh into kernel_stats.csv + pmc_summary.json (what profiles/<round>/ holds)."""
import csv, glob, json, os, shutil, sys

out = sys.argv[1]
KERNEL = os.environ.get('PROFILE_KERNEL', 'k_skyvis_rec')            # matches k_skyvis_rec<...> and k_skyvis_rec_f32pk<...>

stats = glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True)
summary = {}
avg_ms = None
if stats:
    shutil.copy(stats[0], os.path.join(out, 'kernel_stats.csv'))
    with open(stats[0]) as f:
        for row in csv.DictReader(f):
            if KERNEL in row['Name'] and (avg_ms is None or float(row['AverageNs']) * 1e-6 > avg_ms):
                avg_ms = float(row['AverageNs']) * 1e-6
                summary['_kernel'] = {'Kernel_Name': row['Name'], 'Calls': int(row['Calls']), 'TotalDurationNs': float(row['TotalDurationNs'])}
    # PROFILE_LAUNCHES_PER_STEP = n: a step runs the kernel n times over source runs of different length (config 3 + diffuse: the point
    # sources, then the pixels), so the figure to hold against the hipEvent time of a step is n x the average over all calls
    lps = int(os.environ.get('PROFILE_LAUNCHES_PER_STEP', '1'))
    if avg_ms is not None and lps > 1:
        summary['_kernel']['launches_per_step'] = lps
        avg_ms *= lps
for path in sorted(glob.glob(os.path.join(out, 'pmc*', '**', '*counter_collection.csv'), recursive=True)):
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if KERNEL not in row['Kernel_Name'] or 'reduce' in row['Kernel_Name']:
                continue
            key = (row['Counter_Name'], row['Dispatch_Id'])
            acc[key] = acc.get(key, 0.0) + float(row['Counter_Value'])
    per_counter = {}
    for (name, _), v in acc.items():
        per_counter.setdefault(name, []).append(v)
    for name, vals in per_counter.items():
        big = [v for v in vals if v >= 0.5 * max(vals)] or vals     # ignore tiny launches (parity spot check)
        summary[name] = {'launches': len(big), 'mean_per_launch': sum(big) / len(big)}
        lps_ = int(os.environ.get('PROFILE_LAUNCHES_PER_STEP', '1'))
        if lps_ > 1 and len(vals) % lps_ == 0:                      # several launches per step: the figure per STEP (all of them summed)
            summary[name] = {'launches': len(vals), 'mean_per_launch': sum(vals) / (len(vals) // lps_), 'per': 'step of %d launches' % lps_}
d = {}
if avg_ms:
    d['avg_kernel_ms (rocprofv3 --kernel-trace --stats)'] = avg_ms
if 'FETCH_SIZE' in summary and 'WRITE_SIZE' in summary:
    # MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 64 B of every 128 B request -> x2
    d['hbm_bytes_per_launch (2*FETCH_SIZE + WRITE_SIZE, KiB->B, gfx950 FETCH correction)'] = \
        (2.0 * summary['FETCH_SIZE']['mean_per_launch'] + summary['WRITE_SIZE']['mean_per_launch']) * 1024.0
if 'GRBM_GUI_ACTIVE' in summary and avg_ms:
    d['clock_GHz (GRBM_GUI_ACTIVE/8/avg kernel time)'] = summary['GRBM_GUI_ACTIVE']['mean_per_launch'] / 8.0 / (avg_ms * 1e-3) * 1e-9
try:
    with open(os.path.join(out, 'bench_under_rocprof.json')) as f:
        recs = [json.loads(l) for l in f if l.startswith('{')]
    # (tools/config2_batch.py prints the batched call and then the per-snapshot chain it replaces: the profiled kernel is the batched one)
    b = ([r for r in recs if str(r.get('mode', '')).startswith('batch')] or recs)[-1]
    terms = b['roofline']['terms_per_launch']
    d['algorithmic_bytes_per_launch'] = b['roofline_hbm']['algorithmic_bytes_per_launch']
    if 'SQ_INSTS_VALU' in summary:
        d['valu_wave_instructions_per_wave_term'] = summary['SQ_INSTS_VALU']['mean_per_launch'] / (terms / 64.0)
    d['hipEvent_avg_kernel_ms_under_rocprof'] = b['roofline']['avg_kernel_ms']
except Exception as e:      # noqa
    d['bench_line_error'] = repr(e)
summary['_derived'] = d
try:
    with open(os.path.join(out, 'csrc_hash.txt')) as f:
        summary['_csrc_hash'] = f.read().strip()      # bench.csrc_hash() of the sources this profile was taken with
except Exception:
    pass
with open(os.path.join(out, 'pmc_summary.json'), 'w') as f:
    json.dump(summary, f, indent=1)
print(json.dumps(d, indent=1))
