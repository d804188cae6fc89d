# SQ counter groups of the GENERIC kernels (rh_k_size / rh_k_emit) on the 10M-record workload, one --pmc pass each:
#   gpurun -- bash scripts/pmc_generic.sh   -> gpurun_out/r06s5b/pmc_generic_all.txt
OUT=gpurun_out/r06s5b; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs --no-cold-start"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o p$i -- python bench.py --workload full10m --kernel generic --steps 2 --warmup 1 $B > $OUT/pmc$i.log 2>&1; echo "pass $i rc=$?"
  for f in $(find $OUT/pmc$i -name "*.db"); do python scripts/rocpd_summary.py $f; done 2>&1 | grep -E "^(rh_|kernel)" > $OUT/pmc$i.txt; rm -rf $OUT/pmc$i
done
cat $OUT/pmc*.txt > $OUT/pmc_generic_all.txt; grep -E "^rh_k_(size|emit)" $OUT/pmc_generic_all.txt
