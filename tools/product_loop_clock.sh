#!/bin/bash
# does a baseline shard's PRODUCT loop (InterferometerArray.observe_batch on a (RA, Dec) sky model, config 4, rank 0 of 8)
# hold the clock the queued kernel-only loop holds?  Sky-sum kernel durations from rocprofv3 --kernel-trace and GRBM_GUI_ACTIVE from a
# separate --pmc pass of the same program, launch by launch: clock = cycles per XCD / duration.  Run through gpurun from the repo root;
# writes gpurun_out/product_loop_clock.txt (copy to profiles/r05_product_loop_clock.txt).
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/product_loop_clock
rm -rf $OUT && mkdir -p $OUT
CMD="python3 $R/tools/product_loop.py --config 4 --nranks 8 --n-acc 32 --modes batch"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- $CMD > $OUT/pmc.log 2>&1
python3 - $OUT <<'PY' | tee $R/gpurun_out/product_loop_clock.txt
import csv, glob, sys, json
out = sys.argv[1]
dur = []
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'k_skyvis_rec_f32pk' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6 for r in rows]
    starts = [int(r['Start_Timestamp']) for r in rows]
cyc = []
for f in glob.glob(out + '/pmc/**/*counter_collection.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'k_skyvis_rec_f32pk' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE']
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    cyc = [float(r['Counter_Value']) for r in rows]
n = min(len(dur), len(cyc))
print('# config 4, rank 0 of 8 (1016 baselines x 768 channels, nside-64 sky, external beam), fp32; 2 repetitions of [32 snapshots through')
print('# InterferometerArray.observe_batch | 32 kernel-only compute() queued]: launches 0-31 / 64-95 product loop, 32-63 / 96-127 kernel-only')
print('# launch  kernel_ms(trace run)  GRBM_GUI_ACTIVE(pmc run)  GHz = cycles / 8 XCDs / duration  period_ms')
for i in range(n):
    ghz = cyc[i] / 8.0 / (dur[i] * 1e-3) / 1e9
    per = (starts[i + 1] - starts[i]) * 1e-6 if i + 1 < n else float('nan')
    print('%4d  %8.3f  %14.0f  %6.3f  %8.3f' % (i, dur[i], cyc[i], ghz, per))
def seg(lo, hi):
    d = dur[lo:hi]; c = cyc[lo:hi]
    return {'launches': [lo, hi], 'avg_kernel_ms': sum(d) / len(d), 'avg_GHz': sum(ci / 8.0 / (di * 1e-3) / 1e9 for ci, di in zip(c, d)) / len(d),
            'first4_GHz': [round(ci / 8.0 / (di * 1e-3) / 1e9, 3) for ci, di in zip(c[:4], d[:4])]}
if n >= 128:
    print(json.dumps({'product_loop_rep2': seg(64, 96), 'product_loop_rep2_last16': seg(80, 96), 'kernel_only_rep2': seg(96, 128)}))
PY
