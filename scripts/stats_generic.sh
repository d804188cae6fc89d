#!/bin/bash
# rocprofv3 --kernel-trace --stats of the GENERIC kernels (rh_k_size / rh_k_emit) over the bench's single-stream region, 10M records:
#   gpurun -- bash scripts/stats_generic.sh   -> gpurun_out/stats_generic/kernel_stats_full10m_generic.txt
cd "$(dirname "$0")/.." && OUT=gpurun_out/stats_generic && mkdir -p $OUT && export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs --no-cold-start"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p -o stats -- python bench.py --workload full10m --kernel generic --steps 20 --warmup 5 $B > $OUT/bench.log 2>&1; echo "stats rc=$?"
grep "^{" $OUT/bench.log > $OUT/bench_full10m_generic.json
for f in $(find $OUT/p -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -vE "^$" > $OUT/kernel_stats_full10m_generic.txt
rm -rf $OUT/p; head -8 $OUT/kernel_stats_full10m_generic.txt
python scripts/bench_summary.py $OUT/bench_full10m_generic.json | head -3
