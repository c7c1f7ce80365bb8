#!/bin/bash
# A/B of the sibling-bundle threshold (HIPKKT_BUNDLE_KIDS; 0 = no bundles) on one box: the headline workload, cfg3 / cfg5 and the zoo
for b in "$@"; do
  export HIPKKT_BUNDLE_KIDS=$b
  echo "== HIPKKT_BUNDLE_KIDS=$b"
  bash scripts/ab_env.sh - | head -1
  python bench.py --configs 3,5,4b --steps 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    print('cfg%-3s' % d['config'], 'unit %.3f ms (level C %.3f)  factor %.3f  trisolve %.3f  TF %.2f  fallbacks %s' % (d['gpu_ms_per_unit'], d.get('system_ms_per_unit', 0), d['factor_ms'], d['trisolve_ms'], d['factor_TFLOPs'], d.get('fallbacks')))
"
  python scripts/bench_zoo.py --reps 8 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if not line.startswith('{'): continue
    d=json.loads(line)
    print('zoo %-16s unit %8.3f ms factor %8.3f sweep %7.3f nsuper %7d stored %9d fallbacks %s ok %s' % (d['name'], d['unit_ms'], d['factor_ms'], d['sweep_pair_ms'], d['nsuper'], d['nnzL_stored'], d['fallbacks'], d['ok']))
"
done
