#!/bin/bash
# After scripts/final_profiles_1.sh and _2.sh (and bench.py --configs 2lr) ran on the GPU box: summaries into profiles/
set -e
for t in r03 r03_cfg3 r03_cfg5; do python scripts/summarize_profile.py gpurun_out/profile_$t profiles $t > /dev/null; done
cp gpurun_out/r03_step_timeline.txt gpurun_out/r03_system.json gpurun_out/r03_multirhs.json gpurun_out/r03_bench_rhs.json \
   gpurun_out/r03_bench_problems.json gpurun_out/r03_configs.jsonl gpurun_out/r03_config_2lr.jsonl profiles/
# (the default bench line is only copied when it was measured against THIS traffic summary: run bench.py once more after
#  the summary exists, scripts/final_profiles_1.sh's own run predates it)
python3 - <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
print('traffic summary matches sources:', bench.traffic_summary() is not None)
d = json.loads([x for x in open('gpurun_out/bench_r03_default.json') if x.startswith('{')][-1])
print('default bench line: %.1f steps/s, traffic %s' % (d['value'], d['roofline']['traffic']))
PY
