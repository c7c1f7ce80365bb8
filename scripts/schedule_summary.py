#!/usr/bin/env python3
"""Print the schedule summary (HIPKKT_VERBOSE) of every benchmark configuration: schedule_summary.py [cfg ...]."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import sys; sys.path.insert(0, %r)\n"
        "from cuclarabel_amd import problems\n"
        "from cuclarabel_amd.kktsolver import HipKKTSolver\n"
        "pb = getattr(problems, 'config' + sys.argv[1])()\n"
        "ks = HipKKTSolver(pb.P, pb.A, pb.cones)\n") % ROOT
for cfg in (sys.argv[1:] or ["1", "2", "3", "4", "5"]):
    r = subprocess.run([sys.executable, "-c", CODE, cfg], env=dict(os.environ, HIPKKT_VERBOSE="1"), capture_output=True, text=True)
    for line in r.stderr.splitlines():
        if line.startswith("[hipkkt]"):
            print("cfg%s %s" % (cfg, line), flush=True)
