#!/bin/bash
# rocprofv3 recipe behind profiles/<tag>/ (run on the MI355X box through gpurun, from the repo root):
#   tools/profile_round.sh r02_taper_f32 --workload cfg5 --steps 2
#     -> gpurun_out/prof_<tag>/{kernel_stats.csv, pmc_summary.json, bench_under_rocprof.json}
# Kernel timing and every counter group are separate passes (gpurun refuses --pmc combined with API tracing).
set -e
TAG=${1:-rXX}
shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STEPS="--steps 5"
case " $* " in *" --steps "*) STEPS="";; esac
BENCH="python3 $REPO/bench.py $STEPS --warmup 1 --no-cpu-baseline $*"
# PROFILE_CMD: another program to profile instead of bench.py (e.g. "python3 @REPO@/tools/profile_delay.py 4"); PROFILE_KERNEL: the kernel
# whose counters tools/summarize_pmc.py condenses (default k_skyvis_rec)
if [ -n "$PROFILE_CMD" ]; then BENCH="${PROFILE_CMD//@REPO@/$REPO}"; fi     # the passes run from /tmp: write @REPO@ for the repo root
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
echo "trace done"
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" \
             "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  rocprofv3 --pmc $group --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err" || echo "pmc group $i failed: $group"
  echo "pmc $i done"
done
# PROFILE_EXTRA_PMC: one more counter group, e.g. "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" for the MFMA gradient kernel
if [ -n "$PROFILE_EXTRA_PMC" ]; then
  i=$((i+1))
  rocprofv3 --pmc $PROFILE_EXTRA_PMC --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.json" 2> "$OUT/pmc$i.err" || echo "pmc group $i failed: $PROFILE_EXTRA_PMC"
  echo "pmc $i done"
fi
cd "$REPO"
python3 -c "import bench; print(bench.csrc_hash())" > "$OUT/csrc_hash.txt"
python3 tools/summarize_pmc.py "$OUT"
rm -rf "$OUT"/trace "$OUT"/pmc[0-9]
