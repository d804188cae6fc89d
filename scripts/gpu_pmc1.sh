#!/bin/bash
# one SQ PMC pass (instruction mix) over the specialised kernels
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -d $OUT/p1 -o p1 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --kernel specialized > $OUT/p1.log 2>&1; echo "rc=$?"
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --kernel specialized | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['kernel_ms'], d['config']['emit_lds_bytes_per_workgroup'])"
