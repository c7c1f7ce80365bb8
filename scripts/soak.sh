for i in 1 2 3 4 5 6; do python bench.py --configs 3,5,2 --steps 10 --cfg-cpu-units 1 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('run $i cfg', d['config'], round(d['gpu_ms_per_unit'],3), round(d['system_ms_per_unit'],3), d['fallbacks'], d['fallbacks_reported'])
"; done
for i in 1 2; do python bench.py --mode problems --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('problems', round(d['value'],1), d['config'].get('fallbacks'))
"; done
