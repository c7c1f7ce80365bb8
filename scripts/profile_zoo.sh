#!/bin/bash
# Per-kernel time of one zoo structure's units (scripts/bench_zoo.py --only <name>) on the GPU box:
#   scripts/profile_zoo.sh <name> [--big]   ->  gpurun_out/zoo_prof_<name>/ (rocprofv3 --kernel-trace --stats) and a top-25 table
NAME=$1
shift
OUT=gpurun_out/zoo_prof_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/bench_zoo.py --reps 5 --only $NAME "$@" > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(sys.argv[1] + "/top.txt", "w") as out:
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
        out.write("%-70s calls %6s avg %9.1f us total %9.1f us %5.1f%%\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                           float(r["TotalDurationNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat $OUT/top.txt
