#!/bin/bash
# r04k: window bounds from a compact per-tile table (rh_k_tilepos, one extra ~5 us launch per call) vs from the offsets array
OUT=gpurun_out/r04k; mkdir -p $OUT; export TMPDIR=/tmp
STEPS=30 bash scripts/gpu_env_ab.sh r04k "tilepos:" "offsets:RUHVRO_HIP_NO_TILEPOS=1" "tilepos2:" "offsets2:RUHVRO_HIP_NO_TILEPOS=1" "tilepos3:" "offsets3:RUHVRO_HIP_NO_TILEPOS=1"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04k/bench_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], "ms/step", round(d["ms_per_step"],4), "sync", round(d["config"]["sync_call_ms"],4), "overlapped", round(d["overlapped"]["ms_per_step"],4), {k: (round(v["ms_per_step"],4), round(v["sync_call_ms"],4)) for k,v in d.get("other_configs",{}).items() if "sync_call_ms" in v})
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p -o t -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-projection --no-other-configs --overlap-streams 0 > $OUT/st.json 2> $OUT/st.err
for f in $(find $OUT/p -name "*.db"); do python scripts/rocpd_summary.py $f | head -8; done
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
