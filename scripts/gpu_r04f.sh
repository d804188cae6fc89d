#!/bin/bash
# r04f: emit latency diet -- batched string copy (one LDS round trip per string column; RH_V_NOBATCH = off) and head prefetch
# (RUHVRO_HIP_NO_HEAD_PREFETCH=1 = off): parity on the GPU suite's core files, then the 2 x 2 A/B, then PMC of the new default
OUT=gpurun_out/r04f; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_async_device.py tests/test_engine_branches.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
STEPS=20 bash scripts/gpu_env_ab.sh r04f "new:" "nobatch:RUHVRO_HIP_VARIANT=NOBATCH" "nopf:RUHVRO_HIP_NO_HEAD_PREFETCH=1" "neither:RUHVRO_HIP_VARIANT=NOBATCH RUHVRO_HIP_NO_HEAD_PREFETCH=1" "new2:" "nobatch2:RUHVRO_HIP_VARIANT=NOBATCH" "nopf2:RUHVRO_HIP_NO_HEAD_PREFETCH=1" "neither2:RUHVRO_HIP_VARIANT=NOBATCH RUHVRO_HIP_NO_HEAD_PREFETCH=1"
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/p_new -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs > $OUT/p_new.log 2>&1; echo "pmc rc=$?"
for f in $(find $OUT/p_new -name "*.db"); do python scripts/rocpd_summary.py $f 2>&1 | grep -E "^rh_spec" > $OUT/pmc_new.txt; done
rm -rf $OUT/p_new; cat $OUT/pmc_new.txt
