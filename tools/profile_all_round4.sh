#!/bin/bash
# Every rocprofv3 summary of round 4, one after the other (run through gpurun from the repo root; ~14 min of box time), then the manifest
# tests/test_host_logic.py checks: profiles/r04_MANIFEST.json names the directories taken with the sources that are committed.
#   tools/profile_all_round4.sh [tags...]      default: all
set -e
TAGS=${@:-"headline fp64 taper64_3d taper64_5 taper32_5 taper32_3d grad64 grad32 delay cfg2"}
DONE=""
for t in $TAGS; do
  case $t in
    headline) tools/profile_round.sh r04_headline_f32; DONE="$DONE r04_headline_f32" ;;
    fp64) tools/profile_round.sh r04_fp64 --precision fp64; DONE="$DONE r04_fp64" ;;
    taper64_3d) PROFILE_KERNEL=k_skyvis_taper tools/profile_round.sh r04_taper_f64_cfg3d --workload cfg3d --precision fp64 --steps 2; DONE="$DONE r04_taper_f64_cfg3d" ;;
    taper64_5) PROFILE_KERNEL=k_skyvis_taper tools/profile_round.sh r04_taper_f64_cfg5 --workload cfg5 --precision fp64 --steps 2; DONE="$DONE r04_taper_f64_cfg5" ;;
    taper32_5) tools/profile_round.sh r04_taper_f32_cfg5 --workload cfg5 --steps 2; DONE="$DONE r04_taper_f32_cfg5" ;;
    taper32_3d) tools/profile_round.sh r04_taper_f32_cfg3d --workload cfg3d --steps 3; DONE="$DONE r04_taper_f32_cfg3d" ;;
    grad64) PROFILE_KERNEL=k_skyvis_grad PROFILE_EXTRA_PMC="SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" tools/profile_round.sh r04_grad_f64 --precision fp64 --want-grad --steps 3; DONE="$DONE r04_grad_f64" ;;
    grad32) PROFILE_KERNEL=k_skyvis_grad PROFILE_EXTRA_PMC="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" tools/profile_round.sh r04_grad_f32 --want-grad --steps 3; DONE="$DONE r04_grad_f32" ;;
    delay) PROFILE_KERNEL=k_delay_fft PROFILE_CMD="python3 @REPO@/tools/profile_delay.py 4" tools/profile_round.sh r04_delay_fft; DONE="$DONE r04_delay_fft" ;;
    cfg2) PROFILE_KERNEL=k_skyvis_taper PROFILE_CMD="python3 @REPO@/tools/profile_cfg2.py" tools/profile_round.sh r04_cfg2_fp64; DONE="$DONE r04_cfg2_fp64" ;;
  esac
  echo "== $t done"
done
python3 - $DONE <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
dirs = sys.argv[1:]
with open(os.path.join('gpurun_out', 'r04_MANIFEST.json'), 'w') as f:
    json.dump({'csrc_hash': bench.csrc_hash(), 'dirs': dirs, 'what': 'rocprofv3 summaries (tools/profile_round.sh) taken with the committed kernel sources; '
               'copied from gpurun_out/prof_<dir>/ to profiles/<dir>/'}, f, indent=1)
print('manifest:', dirs)
PY
