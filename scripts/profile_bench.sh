#!/bin/bash
# Profiles bench.py on the GPU box.  Usage: scripts/profile_bench.sh <tag>   (run through gpurun)
# Produces under gpurun_out/profile_<tag>/:
#   stats/      rocprofv3 --kernel-trace --stats (per-kernel durations)
#   pmc_fetch/  rocprofv3 --pmc FETCH_SIZE   (own pass: TCC has 4 slots, FETCH_SIZE takes 3)
#   pmc_write/  rocprofv3 --pmc WRITE_SIZE
#   bench.json  the bench line of the same command, un-profiled
set -e
TAG=${1:-r01}
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 5 --warmup 2 --no-cpu-baseline"
python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1
ls -R $OUT | head -40
