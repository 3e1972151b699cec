#!/bin/bash
# Does the 45x HBM traffic of the packed fp32 taper kernel on config 5 cost time?  (run on the MI355X box from the repo root.)
# Builds a second library whose packed kernels read row (s mod 8192) of their slab -- 2 MiB per channel tile, L2-resident whatever the
# blocks' drift; results wrong, instruction stream identical -- and times both libraries alternately on one LST of config 5, each also
# with a single flush (PRISIM_HIP_FLUSH_SRC huge: no read-modify-write passes over the cube).
set -e
REPO=$(pwd)
OUT=$REPO/gpurun_out/taper_traffic
mkdir -p "$OUT" build/variants
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/opt/rocm/include -DPRISIM_EXPERIMENT_ROW_WRAP=8192 -c prisim_amd/csrc/skyvis_kernels.hip -o build/variants/skyvis_rowwrap.o
hipcc -shared -fPIC --offload-arch=gfx950 build/variants/skyvis_rowwrap.o build/csrc/aux_kernels.o build/csrc/delay_kernels.o build/csrc/capi.o -o build/variants/libprisim_rowwrap.so -ldl
echo "variant built"
cat > "$OUT/one.py" <<'PY'
import sys, os, json
sys.path.insert(0, os.environ['REPO'])
import numpy as NP
from prisim_amd import _abi
_abi.LIB_PATH = sys.argv[1]
from prisim_amd import workloads as W
cfg = W.config5(n_acc=1)
bl, ch, sky = cfg['baselines'], cfg['channels'], cfg['sky']
zen = NP.array([0.0, 0.0, 1.0])
ctx = _abi.Context(0)
ctx.set_array(bl, ch, nt_max=1)
ctx.set_sky_analytic(sky['dircos'], sky['flux_ref'], sky['spindex'], sky['ref_freq'], _abi.PRISIM_BEAM_AIRY, 14.0, zen, zen, fwhm_deg=sky['fwhm_deg'])
ts = []
for rep in range(int(sys.argv[2])):
    ctx.compute(precision=_abi.PRISIM_FP32); ctx.sync(); ts.append(ctx.timing()['last_kernel_ms'])
print(json.dumps({'lib': os.path.basename(sys.argv[1]), 'flush_src': os.environ.get('PRISIM_HIP_FLUSH_SRC', 'default (16384)'), 'kernel_ms': ts}), flush=True)
PY
export REPO
PROD=$REPO/prisim_amd/lib/libprisim_hip.so
WRAP=$REPO/build/variants/libprisim_rowwrap.so
for rnd in 1 2; do
  for lib in $PROD $WRAP; do
    python3 "$OUT/one.py" $lib 2 >> "$OUT/timing.jsonl"
    PRISIM_HIP_FLUSH_SRC=1000000000 python3 "$OUT/one.py" $lib 2 >> "$OUT/timing.jsonl"
  done
done
echo "timing done"; cat "$OUT/timing.jsonl"
# (The counters of these cases -- FETCH_SIZE / WRITE_SIZE / SQ_WAIT_ANY / SQC_DCACHE_MISSES -- are in profiles/r03_taper_f32_cfg5 and
# r03_pmc_wait_diag.txt for the production build: 118 GB per launch, 46 % scalar-cache misses.  A rocprofv3 --pmc pass of the row-wrap
# variant aborted inside the profiler (signal 6) on the round-4 box and is not repeated here: the timing above is the answer.)
