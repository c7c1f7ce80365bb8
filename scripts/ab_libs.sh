#!/bin/bash
# timing of several library builds on ONE box, interleaved, twice: scripts/ab_libs.sh <a.so> <b.so> ...  ("-" = the in-tree build)
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset HIPKKT_LIB; else export HIPKKT_LIB=$lib; fi
    python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-32s' % '$lib', round(d['value'],1), round(d['ms_per_step'],3), round(d['ms_per_step_sequential_solves'],3), {k:round(v['avg_ms'],4) for k,v in d['phases'].items()})
" || exit 1
  done
done
