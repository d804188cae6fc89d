#!/bin/bash
# The one runner for everything that goes to the MI355X box:
#
#   gpurun --timeout T -- bash scripts/gpu_run.sh <tag> <step> [<step> ...]
#
# Outputs go to gpurun_out/<tag>/ (copied back by gpurun); what is worth keeping is copied to profiles/<tag>_*.
# Every step runs under its own `timeout`, so a hung kernel cannot eat the GPU budget.  Steps:
#
#   suite                 whole GPU suite (pytest -m gpu)
#   tests:<args>          pytest -m gpu <args>          e.g. tests:tests/test_round5.py   "tests:tests/test_gpu_parity.py -k golden"
#   notrust               parity core again with RUHVRO_HIP_NO_TRUST=1
#   smoke                 __graft_entry__.smoke()
#   bench[:<args>]        bench.py <args> -> bench.json + a short summary
#   stats[:<workload>]    rocprofv3 --kernel-trace --stats over the bench's single-stream region -> kernel_stats_<workload>.txt
#   stamp                 FETCH_SIZE / WRITE_SIZE passes (separate) -> hbm_traffic.json stamped with the kernel key, copied to profiles/
#   stamp_encode          the same for the Arrow -> Avro direction -> encode_hbm_traffic.json
#   pmc[:<workload>]      SQ / TCP / GRBM counter groups, one --pmc pass each
#   py:<script args>      python <script args>          e.g. py:scripts/gather_scaling.py
#   env:<NAME=value>      export for the steps that follow
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs --no-cold-start"
summary() { for f in $(find $1 -name "*.db"); do python scripts/rocpd_summary.py $f; done; }
n=0
for step in "$@"; do
  n=$((n+1)); name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "=== [$n] $step"
  case $name in
    env) export "$arg";;
    suite) timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log;;
    tests) timeout 1500 python -m pytest -m gpu -q -x $arg > $OUT/tests_$n.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/tests_$n.log; tail -15 $OUT/tests_$n.log;;
    notrust) RUHVRO_HIP_NO_TRUST=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_async_device.py tests/test_single_pass.py -m gpu -q > $OUT/pytest_no_trust.log 2>&1
             echo "pytest NO_TRUST rc=$?" | tee -a $OUT/pytest_no_trust.log; tail -2 $OUT/pytest_no_trust.log;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log;;
    bench) timeout 1200 python bench.py $arg > $OUT/bench_$n.json 2> $OUT/bench_$n.err; echo "bench rc=$?"; tail -2 $OUT/bench_$n.err
           cp $OUT/bench_$n.json $OUT/bench.json; python scripts/bench_summary.py $OUT/bench_$n.json;;
    stats) W=${arg:-full10m}
           timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_stats -o stats -- python bench.py --workload $W --steps 20 --warmup 5 $B > $OUT/p_stats_$W.log 2>&1; echo "stats rc=$?"
           grep "^{" $OUT/p_stats_$W.log > $OUT/bench_stats_$W.json
           summary $OUT/p_stats | grep -vE "^$" > $OUT/kernel_stats_$W.txt; [ "$W" = full10m ] && cp $OUT/kernel_stats_$W.txt $OUT/kernel_stats.txt; head -9 $OUT/kernel_stats_$W.txt; rm -rf $OUT/p_stats;;
    stamp) timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p_fetch -o fetch -- python bench.py --steps 3 --warmup 2 $B > $OUT/p_fetch.log 2>&1; echo "fetch rc=$?"
           timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p_write -o write -- python bench.py --steps 3 --warmup 2 $B > $OUT/p_write.log 2>&1; echo "write rc=$?"
           KEY=$(python -c "from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS; print(cabi.kernel_key(SCHEMAS['full']))")
           python scripts/rocpd_summary.py --traffic-json $(find $OUT/p_fetch -name "*.db" | head -1) $(find $OUT/p_write -name "*.db" | head -1) $KEY > $OUT/hbm_traffic.json
           summary $OUT/p_fetch | grep -E "FETCH_SIZE" > $OUT/fetch.txt; summary $OUT/p_write | grep -E "WRITE_SIZE" > $OUT/write.txt
           rm -rf $OUT/p_fetch $OUT/p_write; cat $OUT/hbm_traffic.json | head -30;;
    stamp_encode) E="--direction encode $B"
           timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pe_fetch -o fetch -- python bench.py --steps 3 --warmup 2 $E > $OUT/pe_fetch.log 2>&1; echo "fetch rc=$?"
           timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pe_write -o write -- python bench.py --steps 3 --warmup 2 $E > $OUT/pe_write.log 2>&1; echo "write rc=$?"
           KEY=$(python -c "from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS; print(cabi.kernel_key(SCHEMAS['full'], True))")
           python scripts/rocpd_summary.py --traffic-json $(find $OUT/pe_fetch -name "*.db" | head -1) $(find $OUT/pe_write -name "*.db" | head -1) $KEY > $OUT/encode_hbm_traffic.json
           rm -rf $OUT/pe_fetch $OUT/pe_write; cat $OUT/encode_hbm_traffic.json | head -30;;
    pmc) W=${arg:-full10m}; i=0
         # (the TA_* group hung twice in round 2 until its timeout: not collected)
         for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
                    "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS" \
                    "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" \
                    "TCP_TCP_TA_DATA_STALL_CYCLES TCP_PENDING_STALL_CYCLES TCP_TOTAL_WRITE TCP_TCC_WRITE_REQ" \
                    "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
           i=$((i+1))
           timeout 120 rocprofv3 --kernel-trace --pmc $set -d $OUT/pmc$i -o p$i -- python bench.py --workload $W --steps 2 --warmup 1 $B > $OUT/pmc$i.log 2>&1; echo "pass $i rc=$?"
           summary $OUT/pmc$i 2>&1 | grep -E "^(rh_|kernel)" > $OUT/pmc$i.txt; rm -rf $OUT/pmc$i
         done; cat $OUT/pmc*.txt > $OUT/pmc_all.txt; grep -E "^rh_spec" $OUT/pmc_all.txt | head -40;;
    py) timeout 1500 python $arg > $OUT/py_$n.log 2>&1; echo "py rc=$?"; tail -40 $OUT/py_$n.log;;
    *) echo "unknown step $step";;
  esac
done
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -type d -empty -delete 2>/dev/null
true
