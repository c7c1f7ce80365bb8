#!/bin/bash
# A/B of environment settings on one box over BASELINE configurations: scripts/ab_cfg.sh 3,5 "<ENV=..>" "-" ...
cfgs=$1; shift
for rep in 1 2; do
  for e in "$@"; do
    if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
    env $ee python bench.py --configs $cfgs --steps 10 --cfg-cpu-units 1 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    print('cfg%-3s %-28s' % (d['config'], '$e'), 'unit %.3f ms (level C %.3f)  factor %.3f  trisolve %.3f  TF %.2f  fallbacks %s' % (d['gpu_ms_per_unit'], d.get('system_ms_per_unit', 0), d['factor_ms'], d['trisolve_ms'], d['factor_TFLOPs'], d.get('fallbacks')))
" || exit 1
  done
done
