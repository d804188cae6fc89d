#!/bin/bash
# r04b: the round's new tests (timed configuration at 10M incl. in-call internal streams, one-rank RCCL bench line, parity_check in
# the line, NO_TRUST), then the in-call overlap A/B: RUHVRO_HIP_INTERNAL_STREAMS = 1 / 2 / 3 / 4 on the default bench loop without
# stats steps (a stats step is never split), two rounds
OUT=gpurun_out/r04b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_round4.py -m gpu -x -q > $OUT/pytest_round4.log 2>&1; echo "round4 rc=$?"; tail -15 $OUT/pytest_round4.log
STEPS=30 BENCH_ARGS="--stats-every 1000 --no-projection --no-other-configs" bash scripts/gpu_env_ab.sh r04b "s1:RUHVRO_HIP_INTERNAL_STREAMS=1" "s2:RUHVRO_HIP_INTERNAL_STREAMS=2" "s3:RUHVRO_HIP_INTERNAL_STREAMS=3" "s4:RUHVRO_HIP_INTERNAL_STREAMS=4" "s8:RUHVRO_HIP_INTERNAL_STREAMS=8" "s1b:RUHVRO_HIP_INTERNAL_STREAMS=1" "s2b:RUHVRO_HIP_INTERNAL_STREAMS=2" "s3b:RUHVRO_HIP_INTERNAL_STREAMS=3" "s4b:RUHVRO_HIP_INTERNAL_STREAMS=4"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04b/bench_s*.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], "ms/step", round(d["ms_per_step"],4), "sync_call", round(d["config"]["sync_call_ms"],4), "overlapped", round(d["overlapped"]["ms_per_step"],4) if d.get("overlapped") else None)
    except Exception as e: print(f, e)
PY
