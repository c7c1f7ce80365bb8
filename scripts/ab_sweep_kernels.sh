#!/bin/bash
# Per-kernel average durations of the sweep kernels under several environment settings, on ONE box:
#   scripts/ab_sweep_kernels.sh "<ENV=..>" ... ("-" = no setting).  rocprofv3 --kernel-trace --stats of a short bench run each.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for e in "$@"; do
  i=$((i+1))
  if [ "$e" = "-" ]; then ee=""; else ee="$e"; fi
  d=gpurun_out/swk_$i
  rm -rf $d
  env $ee rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $d.log 2>&1
  echo "== $e"
  python3 - "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].replace('hipkkt::', '').replace('void ', '').split('(')[0]
    if any(k in n for k in ('fwd', 'bwd', 'top_solve', 'chain')):
        print('  %-36s calls %5s avg %8.1f us  total %9.1f us' % (n[:36], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
  rm -rf $d
done
