#!/bin/bash
# r04zz: confirmation pass of the tree as it ships at the end of round 4: whole GPU suite, the parity core again with RUHVRO_HIP_NO_TRUST=1,
# smoke(), rocprofv3 kernel stats of the default bench command's single-stream region, the default bench line
OUT=gpurun_out/r04zz; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
RUHVRO_HIP_NO_TRUST=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_async_device.py tests/test_single_pass.py -m gpu -q > $OUT/pytest_no_trust.log 2>&1; echo "pytest NO_TRUST rc=$?" | tee -a $OUT/pytest_no_trust.log; tail -2 $OUT/pytest_no_trust.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
B="--no-cpu-baseline --no-end-to-end --no-projection --overlap-streams 0 --no-other-configs"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_stats -o stats -- python bench.py --steps 20 --warmup 5 $B > $OUT/p_stats.log 2>&1; echo "stats rc=$?"
for f in $(find $OUT/p_stats -name "*.db"); do python scripts/rocpd_summary.py $f; done | grep -vE "^$" > $OUT/full10m_kernel_stats.txt; head -9 $OUT/full10m_kernel_stats.txt
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -2 $OUT/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04zz/bench_default.json")); r=d["roofline"]
print("ms/step", round(d["ms_per_step"],4), "value", d["value"], d["config"]["kernel_ms"], d["config"].get("kernel_form"))
print("roofline", {k: r.get(k) for k in ("kernel","frac","path_frac","traffic","read_frac","hbm_frac","path_traffic")})
print("single_pass", d.get("single_pass"))
print("overlapped", d["overlapped"]["ms_per_step"], "sync", d["config"]["sync_call_ms"])
print("proj", {g:(round(v['ms_per_step'],4), round(v['implied_efficiency'],3)) for g,v in d['config5_projection']['g'].items()})
print("parity", d.get("parity_check"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print({k: (round(v["ms_per_step"],4), round(v.get("sync_call_ms",0),4), round(v["emit_frac"],3), v.get("traffic")) for k,v in d["other_configs"].items()})
e=d["end_to_end"]; print({k: round(e[k]["value"]/1e6,1) for k in ("packed_pageable","record_slices","packed_8_logical_shards")}, {m: (round(v["value"]/1e6,1), v["gil_held_ms"], round(v["vs_record_slices"],3)) for m,v in e["python_list_bytes"].items()}, e["config1_python_10k"]["wall_ms"])
PY
