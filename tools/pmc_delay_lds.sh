#!/bin/bash
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_delay_lds
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -- python3 $REPO/tools/profile_delay.py 4 > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p2 -- python3 $REPO/tools/profile_delay.py 4 > $OUT/p2.json 2> $OUT/p2.err
cd $REPO
python3 - <<'PY'
import csv, glob
for path in sorted(glob.glob('gpurun_out/prof_delay_lds/p*/**/*counter_collection.csv', recursive=True)):
    acc={}
    for row in csv.DictReader(open(path)):
        if 'k_delay_fft' not in row['Kernel_Name']: continue
        acc.setdefault(row['Counter_Name'],[]).append(float(row['Counter_Value']))
    for k,v in acc.items():
        # sum per dispatch: rows are per-dimension; aggregate by dispatch count 5
        print(k, sum(v)/5.0)
PY
