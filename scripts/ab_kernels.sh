#!/bin/bash
# per-kernel time of two library builds on one box: scripts/ab_kernels.sh <other.so>  (rocprofv3 --kernel-trace --stats of bench.py, each)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in A B; do
  if [ $v = A ]; then unset HIPKKT_LIB; else export HIPKKT_LIB=$1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abk_$v -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/abk_$v.log 2>&1
done
python3 - <<'PY'
import csv, glob
def load(v):
    f = glob.glob('gpurun_out/abk_%s/*/*_kernel_stats.csv' % v)[0]
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f))}
a, b = load('A'), load('B')
rows = []
for k in set(a) | set(b):
    ta = a.get(k, (0, 0.0)); tb = b.get(k, (0, 0.0))
    rows.append((abs(ta[1] - tb[1]), k, ta, tb))
for d, k, ta, tb in sorted(rows, reverse=True)[:14]:
    print('%-60s A %5d calls %9.1f us   B %5d calls %9.1f us   A-B %+8.1f us' % (k.replace('hipkkt::', '')[:60], ta[0], ta[1] / 1e3, tb[0], tb[1] / 1e3, (ta[1] - tb[1]) / 1e3))
PY
