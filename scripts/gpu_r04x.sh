#!/bin/bash
# r04x: the round's confirmation pass on the tree as it ships: whole GPU suite, the parity core again with RUHVRO_HIP_NO_TRUST=1, smoke(),
# traffic re-stamp (FETCH / WRITE passes) incl. the opt-in single-pass kernel, kernel stats, SQ / TCP / GRBM counter groups, default bench line
OUT=gpurun_out/r04x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
RUHVRO_HIP_NO_TRUST=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_async_device.py tests/test_single_pass.py tests/test_round4.py -m gpu -q > $OUT/pytest_no_trust.log 2>&1; echo "pytest NO_TRUST rc=$?" | tee -a $OUT/pytest_no_trust.log; tail -2 $OUT/pytest_no_trust.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/p_fetch -o fetch -- python bench.py --steps 3 --warmup 2 $B > $OUT/p_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/p_write -o write -- python bench.py --steps 3 --warmup 2 $B > $OUT/p_write.log 2>&1; echo "write rc=$?"
KEY=$(python -c "from pyruhvro_amd import cabi; from avrogen.schemas import SCHEMAS; print(cabi.kernel_key(SCHEMAS['full']))")
python scripts/rocpd_summary.py --traffic-json $(find $OUT/p_fetch -name "*.db" | head -1) $(find $OUT/p_write -name "*.db" | head -1) $KEY > $OUT/hbm_traffic.json
for f in $(find $OUT/p_fetch -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -E "FETCH_SIZE" > $OUT/full10m_fetch.txt
for f in $(find $OUT/p_write -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -E "WRITE_SIZE" > $OUT/full10m_write.txt
cp $OUT/hbm_traffic.json profiles/hbm_traffic.json
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_stats -o stats -- python bench.py --steps 20 --warmup 5 $B > $OUT/p_stats.log 2>&1; echo "stats rc=$?"
for f in $(find $OUT/p_stats -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -vE "^$" > $OUT/full10m_kernel_stats.txt; head -9 $OUT/full10m_kernel_stats.txt
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
bash scripts/gpu_pmc_r02.sh r04x_pmc > $OUT/pmc_all.txt 2>&1; grep -E "^rh_spec" $OUT/pmc_all.txt | grep -E "INSTS_VALU|INSTS_LDS|WAVE_CYCLES|INSTS_VMEM|TA_BUSY|GUI_ACTIVE" | head -30
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -2 $OUT/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04x/bench_default.json")); r=d["roofline"]
print("ms/step", round(d["ms_per_step"],4), "value", d["value"], d["config"]["kernel_ms"], d["config"]["kernel_form"])
print("roofline", {k: r[k] for k in ("kernel","frac","path_frac","traffic","read_frac","hbm_frac","path_traffic")})
print("single_pass", d.get("single_pass"))
print("overlapped", d["overlapped"]["ms_per_step"], "sync", d["config"]["sync_call_ms"])
print("proj", {g:(round(v['ms_per_step'],4), round(v['implied_efficiency'],3)) for g,v in d['config5_projection']['g'].items()})
print("parity", d.get("parity_check"))
print({k: (round(v["ms_per_step"],4), round(v.get("sync_call_ms",0),4), round(v["emit_frac"],3), v.get("traffic")) for k,v in d["other_configs"].items()})
e=d["end_to_end"]; print({k: round(e[k]["value"]/1e6,1) for k in ("packed_pageable","record_slices","packed_8_logical_shards")}, {m: (round(v["value"]/1e6,1), v["gil_held_ms"], round(v["vs_record_slices"],3)) for m,v in e["python_list_bytes"].items()}, e["config1_python_10k"]["wall_ms"])
PY
