#!/bin/bash
# A/B timing of environment settings on ONE box: scripts/ab_env.sh "<ENV=..>" "<ENV=..>" ... ("-" = no setting), two rounds, interleaved
for rep in 1 2; do
  for e in "$@"; do
    if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
    env $ee python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-48s' % '$e', round(d['value'],1), round(d['ms_per_step'],3), round(d['ms_per_step_sequential_solves'],3), {k:round(v['avg_ms'],4) for k,v in d['phases'].items()})
" || exit 1
  done
done
