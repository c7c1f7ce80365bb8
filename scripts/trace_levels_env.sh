#!/bin/bash
# Kernel-by-kernel timeline of one step under the given environment settings (through gpurun):
#   scripts/trace_levels_env.sh <tag> VAR=value ...   ->  gpurun_out/<tag>_step_timeline.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace_$TAG -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-scale-modes > gpurun_out/trace_$TAG.log 2>&1 || exit 1
python3 scripts/step_timeline.py gpurun_out/prof_trace_$TAG 3 > gpurun_out/${TAG}_step_timeline.txt
