#!/usr/bin/env python3
"""Replay a problem file written by Clarabel.save_to_file (json.jl:25-55) through the C ABI on the
MI355X: the IPM test driver (cuclarabel_amd/ipm.py, Zero / Nonnegative / SecondOrder cones) with
libhipkkt.so as its KKT backend.  Prints status, objective, iterations and KKT refinement rounds.

Usage: python scripts/solve_json.py problem.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from cuclarabel_amd import ipm, jsonio
    P, q, A, b, cones, _ = jsonio.load_problem(sys.argv[1])
    res = ipm.solve(P, q, A, b, cones, ipm.HipBackend(P, A, cones))
    print(json.dumps(dict(status=res.status, obj_val=res.obj_val, obj_val_dual=res.obj_val_dual,
                          iterations=res.iterations, kkt_ir_rounds=res.kkt_ir_rounds,
                          x_head=[float(v) for v in res.x[:8]])))


if __name__ == "__main__":
    main()
