#!/usr/bin/env python3
"""Kernel-by-kernel timeline of ONE step (value update .. last kernel before the next update) from a rocprofv3
--kernel-trace directory: step_timeline.py <dir> [which step, default 4]"""
import csv
import glob
import sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].replace('hipkkt::', '').replace('void ', '').split('(')[0], int(r['Start_Timestamp']), int(r['End_Timestamp']),
        int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)) for r in rows]
idx = [i for i, s in enumerate(seq) if s[0].startswith(('k_cone_elementwise', 'k_cone_scaling'))]
i0, i1 = idx[which], idx[which + 1]
t0, prev_end = seq[i0][1], seq[i0][1]
for s in seq[i0:i1]:
    print('%-40s start %8.1f dur %7.1f wgs %6d gap %6.1f' % (s[0][:40], (s[1] - t0) / 1000, (s[2] - s[1]) / 1000, s[3], (s[1] - prev_end) / 1000))
    prev_end = max(prev_end, s[2])
print('step wall', (seq[i1][1] - t0) / 1000)
