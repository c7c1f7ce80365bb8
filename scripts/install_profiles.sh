#!/bin/bash
# After scripts/final_profiles_1.sh and _2.sh (and bench.py --configs 2lr) ran on the GPU box: summaries into profiles/
set -e
TAG=${1:-r04}
for t in ${TAG} ${TAG}_cfg3 ${TAG}_cfg5; do python scripts/summarize_profile.py gpurun_out/profile_$t profiles $t > /dev/null; done
cp gpurun_out/${TAG}_step_timeline.txt gpurun_out/${TAG}_system.json gpurun_out/${TAG}_multirhs.json gpurun_out/${TAG}_bench_rhs.json \
   gpurun_out/${TAG}_bench_problems.json gpurun_out/${TAG}_configs.jsonl gpurun_out/${TAG}_config_2lr.jsonl gpurun_out/${TAG}_hops_*.txt profiles/
# (the default bench line is only copied when it was measured against THIS traffic summary: run bench.py once more after
#  the summary exists, scripts/final_profiles_1.sh's own run predates it)
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
sys.path.insert(0, '.')
import bench
print('traffic summary matches sources:', bench.traffic_summary() is not None)
d = json.loads([x for x in open('gpurun_out/bench_%s_default.json' % tag) if x.startswith('{')][-1])
print('default bench line: %.1f steps/s, traffic %s' % (d['value'], d['roofline']['traffic']))
PY
