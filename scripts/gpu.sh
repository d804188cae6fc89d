#!/bin/bash
# Local front of scripts/gpu_run.sh: build, warm the kernel cache (a rebuilt library prunes it -- a cold cache on the box means
# generic kernels plus compile helpers eating the box's CPU quota under the measurement), then send the tree to the MI355X box.
#   bash scripts/gpu.sh <timeout-seconds> <tag> <step> [<step> ...]
T=${1:?timeout}; shift
cd "$(dirname "$0")/.." || exit 1
python -c "from pyruhvro_amd._build import build_all; build_all()" || exit 1
python -c "from oracle.build import build; build(); from avrogen.fastgen import build as b; b()" || exit 1
python scripts/known_schemas.py | tail -1 || exit 1
exec /usr/local/graft/bin/gpurun --timeout "$T" -- bash scripts/gpu_run.sh "$@"
