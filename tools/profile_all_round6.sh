#!/bin/bash
# Every rocprofv3 summary of round 6, one after the other (run through gpurun from the repo root; ~20 min of box time), then the manifest
# tests/test_host_logic.py checks: profiles/r06_MANIFEST.json names the directories taken with the sources that are committed.
#   tools/profile_all_round6.sh [tags...]      default: all
set -e
TAGS=${@:-"headline fp64 taper64_3d taper64_5 taper32_5 taper32_3d grad64 grad32 grad64_taper delay cfg2 cfg2_batch cfg2_batch256 cfg2_grad_batch256"}
DONE=""
for t in $TAGS; do
  case $t in
    headline) tools/profile_round.sh r06_headline_f32; DONE="$DONE r06_headline_f32" ;;
    fp64) tools/profile_round.sh r06_fp64 --precision fp64; DONE="$DONE r06_fp64" ;;
    taper64_3d) PROFILE_KERNEL=k_skyvis_taper tools/profile_round.sh r06_taper_f64_cfg3d --workload cfg3d --precision fp64 --steps 2; DONE="$DONE r06_taper_f64_cfg3d" ;;
    taper64_5) PROFILE_KERNEL=k_skyvis_taper tools/profile_round.sh r06_taper_f64_cfg5 --workload cfg5 --precision fp64 --steps 2; DONE="$DONE r06_taper_f64_cfg5" ;;
    taper32_5) tools/profile_round.sh r06_taper_f32_cfg5 --workload cfg5 --steps 2; DONE="$DONE r06_taper_f32_cfg5" ;;
    taper32_3d) tools/profile_round.sh r06_taper_f32_cfg3d --workload cfg3d --steps 3; DONE="$DONE r06_taper_f32_cfg3d" ;;
    grad64) PROFILE_KERNEL=k_skyvis_grad PROFILE_EXTRA_PMC="SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" tools/profile_round.sh r06_grad_f64 --precision fp64 --want-grad --steps 3; DONE="$DONE r06_grad_f64" ;;
    grad32) PROFILE_KERNEL=k_skyvis_grad PROFILE_EXTRA_PMC="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" tools/profile_round.sh r06_grad_f32 --want-grad --steps 3; DONE="$DONE r06_grad_f32" ;;
    grad64_taper) PROFILE_KERNEL=k_skyvis_grad_taper PROFILE_EXTRA_PMC="SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" tools/profile_round.sh r06_grad_f64_taper_cfg3d --workload cfg3d --precision fp64 --want-grad --steps 2; DONE="$DONE r06_grad_f64_taper_cfg3d" ;;
    delay) PROFILE_KERNEL=k_delay_fft PROFILE_CMD="python3 @REPO@/tools/profile_delay.py 4" tools/profile_round.sh r06_delay_fft; DONE="$DONE r06_delay_fft" ;;
    cfg2) PROFILE_KERNEL=k_skyvis_taper_f64_wave PROFILE_CMD="python3 @REPO@/tools/profile_cfg2.py" tools/profile_round.sh r06_cfg2_fp64; DONE="$DONE r06_cfg2_fp64" ;;
    cfg2_batch) PROFILE_KERNEL=k_skyvis_taper_f64_wave_batch PROFILE_CMD="python3 @REPO@/tools/config2_batch.py 64" tools/profile_round.sh r06_cfg2_batch64_fp64; DONE="$DONE r06_cfg2_batch64_fp64" ;;
    cfg2_batch256) PROFILE_KERNEL=k_skyvis_taper_f64_wave_batch PROFILE_CMD="python3 @REPO@/tools/config2_batch.py 256" tools/profile_round.sh r06_cfg2_batch256_fp64; DONE="$DONE r06_cfg2_batch256_fp64" ;;
    cfg2_grad_batch256) PROFILE_KERNEL=k_skyvis_grad_taper_f64_batch PROFILE_EXTRA_PMC="SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" PROFILE_CMD="python3 @REPO@/tools/config2_batch.py 256 grad" tools/profile_round.sh r06_cfg2_grad_batch256_fp64; DONE="$DONE r06_cfg2_grad_batch256_fp64" ;;
  esac
  echo "== $t done"
done
python3 - $DONE <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
dirs = sys.argv[1:]
with open(os.path.join('gpurun_out', os.environ.get('MANIFEST_NAME', 'r06_MANIFEST.json')), 'w') as f:
    json.dump({'csrc_hash': bench.csrc_hash(), 'dirs': dirs, 'what': 'rocprofv3 summaries (tools/profile_round.sh) taken with the committed kernel sources; '
               'copied from gpurun_out/prof_<dir>/ to profiles/<dir>/'}, f, indent=1)
print('manifest:', dirs)
PY
