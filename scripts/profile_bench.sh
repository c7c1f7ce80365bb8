#!/bin/bash
# Profiles a command on the GPU box with rocprofv3, four separate passes (never --pmc together with a trace).
# Usage (through gpurun):  scripts/profile_bench.sh <tag> [bench|cfg3|cfg5]
#   bench (default): python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --sequential-solves   (cfg2, the headline
#                    workload, every sweep single-column so that per-solve traffic is well defined)
#   cfgN:            python3 scripts/bench_one_config.py N 5
# Produces under gpurun_out/profile_<tag>/:
#   stats/      --kernel-trace --stats (per-kernel durations)
#   pmc_fetch/  --pmc FETCH_SIZE   (own pass: TCC has 4 slots, FETCH_SIZE takes 3)
#   pmc_write/  --pmc WRITE_SIZE
#   pmc_mfma/   --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
#   bench.json  the bench line of the same command, un-profiled (bench only)
set -e
TAG=${1:-r02}
WHAT=${2:-bench}
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$WHAT" = "bench" ]; then
  ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --sequential-solves --no-scale-modes"
  python3 $ARGS > $OUT/bench.json 2> $OUT/bench.err
else
  ARGS="scripts/bench_one_config.py ${WHAT#cfg} 5"
fi
echo "$ARGS" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
# Counter collection serialises the kernels, and the factorisation's overlap mode needs two streams running side by side
# (it would time out once and fall back by itself): the counter passes run with it off.  Traffic differs from the default
# mode only by the write-through flag of the top fronts' stores.
export HIPKKT_FACTOR_OVERLAP=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ARGS > $OUT/pmc_mfma.log 2>&1
ls $OUT
