#!/bin/bash
# Where do the waves of the taper kernels wait?  Scalar data cache, instruction cache and SMEM-level counters of one config-5 launch,
# split form against the unsplit kernel (separate --pmc passes; run through gpurun from the repo root).
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_wait_diag
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for variant in split unsplit; do
  if [ $variant = unsplit ]; then export PRISIM_HIP_TAPER_SPLIT=0; else unset PRISIM_HIP_TAPER_SPLIT; fi
  i=0
  for group in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" \
               "SQ_WAIT_INST_ANY SQ_INST_LEVEL_SMEM SQ_INST_CYCLES_SMEM SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    rocprofv3 --pmc $group --output-format csv -d "$OUT/${variant}_$i" -- python3 $REPO/bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> "$OUT/${variant}_$i.err" || echo "group $i failed"
  done
done
cd "$REPO"
python3 - <<'PY'
import csv, glob, os
out = os.path.join('gpurun_out', 'pmc_wait_diag')
for variant in ('split', 'unsplit'):
    tot = {}
    for path in glob.glob(os.path.join(out, variant + '_*', '**', '*counter_collection.csv'), recursive=True):
        acc = {}
        for row in csv.DictReader(open(path)):
            if 'k_skyvis_rec_f32pk' not in row['Kernel_Name']:
                continue
            key = (row['Counter_Name'], row['Dispatch_Id'])
            acc[key] = acc.get(key, 0.0) + float(row['Counter_Value'])
        per = {}
        for (name, _), v in acc.items():
            per.setdefault(name, []).append(v)
        for name, vals in per.items():
            big = [v for v in vals if v >= 0.5 * max(vals)] or vals
            tot[name] = sum(big) / len(big)
    print(variant, {k: '%.4g' % v for k, v in sorted(tot.items())})
PY
