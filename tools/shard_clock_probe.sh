#!/bin/bash
# Shader clock of the sky-sum kernel on a baseline shard against the whole array: kernel durations (rocprofv3 --kernel-trace) and
# GRBM_GUI_ACTIVE (a separate --pmc pass), per grid size.  Run through gpurun from the repo root.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/shard_clock
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/shard_clock_probe.py > $OUT/trace.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 $R/tools/shard_clock_probe.py > $OUT/pmc.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
dur = collections.defaultdict(list)
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_skyvis_rec_f32pk' in r['Kernel_Name']:
            dur[r['Grid_Size_X']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
cyc = collections.defaultdict(list)
for f in glob.glob(out + '/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_skyvis_rec_f32pk' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            cyc[r['Grid_Size']].append(float(r['Counter_Value']))
for g in sorted(dur, key=int):
    d = dur[g][2:]                      # skip the first two launches (clock ramp after the idle of set_sky)
    c = cyc.get(g, [])[2:]
    ms = sum(d) / len(d)
    res = {'grid_size': int(g), 'launches': len(d), 'avg_kernel_ms': ms}
    if c:
        res['GRBM_GUI_ACTIVE_per_launch'] = sum(c) / len(c)
        res['cycles_per_xcd_per_launch'] = sum(c) / len(c) / 8.0
    print(json.dumps(res))
PY
