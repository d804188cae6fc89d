#!/bin/bash
# r04p: phase clock of the single-pass kernel + kernel stats + PMC
OUT=gpurun_out/r04p; mkdir -p $OUT; export TMPDIR=/tmp; export RUHVRO_HIP_SINGLE_PASS=1
B="--no-cpu-baseline --no-end-to-end --no-projection --no-other-configs --overlap-streams 0"
RUHVRO_HIP_PROFILE=1 timeout 200 python bench.py --steps 4 --warmup 2 $B > $OUT/prof.json 2> $OUT/prof.err; grep "ruhvro_hip profile" $OUT/prof.err | tail -2
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/p1 -o p -- python bench.py --steps 2 --warmup 2 $B > $OUT/p1.log 2>&1
for f in $(find $OUT/p1 -name "*.db"); do python scripts/rocpd_summary.py $f 2>&1 | grep -E "^rh_" ; done
rm -rf $OUT/p1
