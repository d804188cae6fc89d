#!/bin/bash
# latency / i-cache PMC passes over the specialised kernels (short timeouts: a bad counter set can hang rocprofv3)
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_IFETCH_LEVEL SQ_IFETCH -d $OUT/p1 -o p1 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --kernel specialized > $OUT/p1.log 2>&1; echo "rc=$?"
timeout 120 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES -d $OUT/p2 -o p2 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --kernel specialized > $OUT/p2.log 2>&1; echo "rc=$?"
