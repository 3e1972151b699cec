#!/bin/bash
# Every rocprofv3 summary of the round, one after the other (run through gpurun from the repo root; ~12 min of box time):
#   tools/profile_all_round3.sh [tags...]      default: all
set -e
TAGS=${@:-"headline fp64 taper taper_unsplit cfg3d grad64 grad32 delay cfg2"}
for t in $TAGS; do
  case $t in
    headline) tools/profile_round.sh r03_headline_f32 ;;
    fp64) tools/profile_round.sh r03_fp64 --precision fp64 ;;
    taper) tools/profile_round.sh r03_taper_f32_cfg5 --workload cfg5 --steps 2 ;;
    taper_unsplit) PRISIM_HIP_TAPER_SPLIT=0 tools/profile_round.sh r03_taper_f32_cfg5_unsplit --workload cfg5 --steps 2 ;;
    cfg3d) tools/profile_round.sh r03_taper_f32_cfg3d --workload cfg3d --steps 3 ;;
    grad64) PROFILE_KERNEL=k_skyvis_grad PROFILE_EXTRA_PMC="SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" tools/profile_round.sh r03_grad_f64 --precision fp64 --want-grad --steps 3 ;;
    grad32) PROFILE_KERNEL=k_skyvis_grad PROFILE_EXTRA_PMC="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" tools/profile_round.sh r03_grad_f32 --want-grad --steps 3 ;;
    delay) PROFILE_KERNEL=k_delay_fft PROFILE_CMD="python3 @REPO@/tools/profile_delay.py 4" tools/profile_round.sh r03_delay_fft ;;
    cfg2) PROFILE_CMD="python3 @REPO@/tools/profile_cfg2.py" tools/profile_round.sh r03_cfg2_fp64 ;;
  esac
  echo "== $t done"
done
