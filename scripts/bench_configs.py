#!/usr/bin/env python3
"""All BASELINE.json configurations at full size on one MI355X (results table of BASELINE.md section 3).
Thin wrapper: the work is `bench.py --configs ...` (whose cpu_baseline leg is the only place outside tests/ and
smoke() that loads the oracle).

Usage: python scripts/bench_configs.py [--configs 1,2,3,4,4b,5] [--steps 10] [--cpu-units 2] [--no-cpu] [--concurrent]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2,3,4,4b,5")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu-units", type=int, default=2)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--concurrent", action="store_true")
    a = ap.parse_args()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--configs", a.configs, "--steps", str(a.steps),
           "--cfg-cpu-units", str(a.cpu_units)]
    if a.no_cpu:
        cmd.append("--no-cpu-baseline")
    if a.concurrent:
        cmd.append("--concurrent")
    sys.exit(subprocess.call(cmd))


if __name__ == "__main__":
    main()
