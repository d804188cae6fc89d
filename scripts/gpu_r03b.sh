#!/bin/bash
bash scripts/gpu_env_ab.sh r03b \
  "nostage_l0:RUHVRO_HIP_VARIANT=NOSTAGE RUHVRO_HIP_STAGE_BYTES=0" \
  "stage0:RUHVRO_HIP_STAGE_BYTES=0" \
  "stage2816:RUHVRO_HIP_STAGE_BYTES=2816" \
  "nostage_l0_again:RUHVRO_HIP_VARIANT=NOSTAGE RUHVRO_HIP_STAGE_BYTES=0"
RUHVRO_HIP_PROFILE=1 RUHVRO_HIP_VARIANT=NOSTAGE RUHVRO_HIP_STAGE_BYTES=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2>&1 | grep -a "profile" | tail -2 > gpurun_out/r03b/phase_nostage.txt
RUHVRO_HIP_PROFILE=1 RUHVRO_HIP_STAGE_BYTES=2816 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2>&1 | grep -a "profile" | tail -2 > gpurun_out/r03b/phase_stage2816.txt
RUHVRO_HIP_PROFILE=1 RUHVRO_HIP_STAGE_BYTES=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end 2>&1 | grep -a "profile" | tail -2 > gpurun_out/r03b/phase_stage0.txt
cat gpurun_out/r03b/phase_*.txt
bash scripts/gpu_pmc_env.sh r03b/pmc_nostage "RUHVRO_HIP_VARIANT=NOSTAGE RUHVRO_HIP_STAGE_BYTES=0" > gpurun_out/r03b/pmc_nostage.txt 2>&1
bash scripts/gpu_pmc_env.sh r03b/pmc_stage2816 "RUHVRO_HIP_STAGE_BYTES=2816" > gpurun_out/r03b/pmc_stage2816.txt 2>&1
