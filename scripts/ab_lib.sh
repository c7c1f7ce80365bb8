#!/bin/bash
# A/B timing of two builds of libhipkkt.so on ONE box: scripts/ab_lib.sh <other.so> [bench args...]
# prints steps/s, ms/step and the phase times for the in-tree build (A) and the other build (B), twice each, interleaved
other=$1; shift
for rep in 1 2; do
  for v in A B; do
    if [ $v = A ]; then unset HIPKKT_LIB; else export HIPKKT_LIB=$other; fi
    python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v$rep', round(d['value'],1), round(d['ms_per_step'],3), round(d['ms_per_step_sequential_solves'],3), {k:round(v['avg_ms'],4) for k,v in d['phases'].items()})
" || exit 1
  done
done
