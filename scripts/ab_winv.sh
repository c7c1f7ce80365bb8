#!/bin/bash
# A/B of the W-formation fork points / grid (knobs.hpp: HIPKKT_WINV_*) on the headline workload; through gpurun.
# Prints ms_per_step (level C lazy) and the level-B figures per setting.
set -e
OUT=gpurun_out/ab_winv
mkdir -p $OUT
run() {
  tag=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-scale-modes > $OUT/$tag.json 2> $OUT/$tag.err
  python3 - "$tag" "$OUT/$tag.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms_per_step %.4f" % d["ms_per_step"], "seq %.4f" % d["ms_per_step_sequential_solves"], "levelB", {k: round(v, 4) for k, v in d["level_B_ms_per_step"].items()},
      "factor %.4f" % d["phases"]["factor"]["avg_ms"], "trisolve %.4f" % d["phases"]["trisolve"]["avg_ms"], flush=True)
PY
}
run base HIPKKT_VERBOSE=0
run early2 HIPKKT_WINV_EARLY=2
run early3 HIPKKT_WINV_EARLY=3
run blocks128 HIPKKT_WINV_BLOCKS=128
run blocks160 HIPKKT_WINV_BLOCKS=160
run early2_b128 HIPKKT_WINV_EARLY=2 HIPKKT_WINV_BLOCKS=128
run base2 HIPKKT_VERBOSE=0
