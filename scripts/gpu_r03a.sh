#!/bin/bash
# round 3, call A: parity of the staged string stores + LDS-budget sweep
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03a/pytest.log
tail -5 gpurun_out/r03a/pytest.log
bash scripts/gpu_env_ab.sh r03a \
  "nostage:RUHVRO_HIP_VARIANT=NOSTAGE" \
  "stage0:RUHVRO_HIP_STAGE_BYTES=0" \
  "stage2816:RUHVRO_HIP_STAGE_BYTES=2816" \
  "stage3584:RUHVRO_HIP_STAGE_BYTES=3584" \
  "stage2048:RUHVRO_HIP_STAGE_BYTES=2048" \
  "stage1024:RUHVRO_HIP_STAGE_BYTES=1024" \
  "w107_stage0:RUHVRO_HIP_WIN_PCT=107 RUHVRO_HIP_WIN_PAD=256 RUHVRO_HIP_STAGE_BYTES=0" \
  "w107_stage1792:RUHVRO_HIP_WIN_PCT=107 RUHVRO_HIP_WIN_PAD=256 RUHVRO_HIP_STAGE_BYTES=1792" \
  "w107_stage1024:RUHVRO_HIP_WIN_PCT=107 RUHVRO_HIP_WIN_PAD=256 RUHVRO_HIP_STAGE_BYTES=1024" \
  "w107_stage2816:RUHVRO_HIP_WIN_PCT=107 RUHVRO_HIP_WIN_PAD=256 RUHVRO_HIP_STAGE_BYTES=2816"
