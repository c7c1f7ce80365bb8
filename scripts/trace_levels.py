#!/usr/bin/env python3
"""Per-launch timeline of the last factorisation (or the last solve) in a rocprofv3 --kernel-trace directory:
trace_levels.py <dir> [factor|solve]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
which = sys.argv[2] if len(sys.argv) > 2 else 'factor'
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].replace('hipkkt::', '').replace('void ', '').split('(')[0], int(r['Start_Timestamp']), int(r['End_Timestamp']),
        int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) for r in rows]
names = ('k_front', 'k_panel', 'k_schur', 'k_winv', 'k_subtree', 'k_factor') if which == 'factor' else ('k_fwd', 'k_bwd', 'k_top_solve', 'k_sub')
idx = [i for i, s in enumerate(seq) if s[0].startswith(names)]
end = idx[-1]
start = end
while start - 1 in idx:
    start -= 1
tot = collections.defaultdict(float)
t0 = seq[start][1]
for s in seq[start:end + 1]:
    tot[s[0]] += (s[2] - s[1]) / 1000.0
    print('%-34s start %8.1f  dur %7.1f us  wgs %6d' % (s[0], (s[1] - t0) / 1000.0, (s[2] - s[1]) / 1000.0, s[3]))
print({k: round(v, 1) for k, v in tot.items()}, 'sum', round(sum(tot.values()), 1), 'wall', round((seq[end][2] - t0) / 1000.0, 1))
