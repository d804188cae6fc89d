#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r06_q && export TMPDIR=/tmp
O=gpurun_out/r06_q
RUHVRO_HIP_HOSTPROF=1 timeout 600 python scripts/workload_probe.py full_skewed 10000000 --reps 3 --no-parity > $O/skewed_hostprof.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_rn -o stats -- python scripts/workload_probe.py full_realistic_nogiant 10000000 --reps 10 --no-parity > $O/rn_probe.txt 2>&1
for f in $(find $O/p_rn -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -vE "^$" > $O/kernel_stats_realistic_nogiant.txt; rm -rf $O/p_rn
timeout 600 rocprofv3 --kernel-trace --stats -d $O/p_sk -o stats -- python scripts/workload_probe.py full_skewed 10000000 --reps 10 --no-parity > $O/sk_probe.txt 2>&1
for f in $(find $O/p_sk -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -vE "^$" > $O/kernel_stats_skewed.txt; rm -rf $O/p_sk
RUHVRO_HIP_HOSTPROF=1 timeout 600 python scripts/host_path_profile.py > $O/host_path_profile.txt 2>&1
tail -12 $O/skewed_hostprof.txt | cut -c1-400; head -8 $O/kernel_stats_realistic_nogiant.txt; head -8 $O/kernel_stats_skewed.txt; grep -v hostprof $O/host_path_profile.txt | tail -12 | cut -c1-400
