#!/bin/bash
RT=$(python -c "from pyruhvro_amd._build import asan_runtime; print(asan_runtime())")
export LD_PRELOAD="$RT $(gcc -print-file-name=libstdc++.so.6)" RUHVRO_HIP_LIB=$PWD/pyruhvro_amd/_san/libruhvro_hip.so LD_LIBRARY_PATH=$PWD/pyruhvro_amd/_san:$LD_LIBRARY_PATH
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:protect_shadow_gap=0:verbosity=0" UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
timeout 300 python -c "
import sys; sys.path.insert(0,'tests')
print('start', flush=True)
import pyruhvro_amd as P
print('devices', P.device_count(), flush=True)
from avrogen import synth
from avrogen.schemas import SCHEMAS
r = P.deserialize_array_threaded(synth.records('full', 500), SCHEMAS['full'], 4)
print('decoded', [b.num_rows for b in r], flush=True)
" 2>&1 | tail -30
echo "rc=$?"
timeout 600 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_device_export.py 2>&1 | tail -30
