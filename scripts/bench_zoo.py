#!/usr/bin/env python3
"""The structure zoo (cuclarabel_amd.problems.ZOO: sparsity patterns outside the five BASELINE configurations) timed
on the GPU: one JSON line per structure with handle creation, factorisation and sweep-pair times from the handle's own
per-phase HIP-event timers, the rates they imply against SURVEY.md 8(d)'s algorithmic bytes / flops, and the fall-back
counters.  bench_zoo.py [--reps N] [--only name,name] [--big]   (--big: the larger variants, no oracle involved).

This is the evidence behind DESIGN.md's "outside the five configurations" table: the thresholds of the schedule were
fitted on cfg1-cfg5; here is what the same defaults do elsewhere."""
import argparse
import json
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
import torch

import bench
from cuclarabel_amd import problems
from cuclarabel_amd.kktsolver import HipKKTSolver

BIG = [
    ("grid2d_400", lambda: problems.zoo_grid((400, 400))),
    ("grid3d_36", lambda: problems.zoo_grid((36, 36, 36), seed=2011)),
    ("chain_400k", lambda: problems.zoo_chain(n=400_000)),
    ("arrow_200k", lambda: problems.zoo_arrow(n=200_000)),
    ("powerlaw_60k", lambda: problems.zoo_powerlaw(n=60_000)),
    ("diag_500k", lambda: problems.zoo_diag(n=500_000)),
    ("lp_600x700", lambda: problems.zoo_lp_transport(600, 700)),
    ("equality_200k", lambda: problems.zoo_equality_heavy(n=200_000)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--big", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cases = BIG if args.big else problems.ZOO
    only = set(filter(None, args.only.split(",")))
    for name, mk in cases:
        if only and name not in only:
            continue
        pb = mk()
        t0 = time.perf_counter()
        ks = HipKKTSolver(pb.P, pb.A, pb.cones)
        setup_s = time.perf_counter() - t0
        ks.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        rng = np.random.default_rng(0)
        s, z = torch.from_numpy(pb.s0).to(dev), torch.from_numpy(pb.z0).to(dev)
        rx, rz = torch.from_numpy(rng.standard_normal(pb.n)).to(dev), torch.from_numpy(rng.standard_normal(pb.m)).to(dev)
        lx = torch.zeros(pb.n, dtype=torch.float64, device=dev)
        lz = torch.zeros(pb.m, dtype=torch.float64, device=dev)

        def unit():
            ok = ks.kktsolver_update_from_sz_dev(s.data_ptr(), z.data_ptr())
            for _ in range(3):
                ks.kktsolver_setrhs_dev(rx.data_ptr(), rz.data_ptr())
                ok = ks.kktsolver_solve_dev(lx.data_ptr(), lz.data_ptr()) and ok
            return ok

        ok = True
        for _ in range(3):
            ok = unit() and ok
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ok = unit() and ok
        torch.cuda.synchronize()
        unit_ms = (time.perf_counter() - t0) / args.reps * 1e3
        ks.profile_enable(True)
        ks.profile_reset()
        for _ in range(args.reps):
            ok = unit() and ok
        torch.cuda.synchronize()
        prof = ks.profile()
        ks.profile_enable(False)
        info = ks.info
        B = bench.algorithmic_bytes(info)
        f_ms = prof["factor_ms"] / max(prof["n_factor"], 1)
        t_ms = prof["trisolve_ms"] / max(prof["n_trisolve"], 1)
        print(json.dumps(dict(
            name=name, ok=bool(ok), n=pb.n, m=pb.m, N=info["N"], nnzK=info["nnzK"], nnzL=info["nnzL"], nnzL_stored=info["nnzL_stored"],
            nsuper=info["nsuper"], levels=info["nlevels"], max_front=info["max_front"], factor_flops=info["factor_flops"],
            setup_s=round(setup_s, 3), unit_ms=round(unit_ms, 4), factor_ms=round(f_ms, 4), sweep_pair_ms=round(t_ms, 4),
            update_ms=round(prof["update_ms"] / max(prof["n_update"], 1), 4),
            ir_rounds_per_unit=prof["ir_iterations"] / args.reps,
            factor_TFLOPs=round(info["factor_flops"] / (f_ms * 1e-3) / 1e12, 4),
            sweep_GBs=round(B["solve"] / (t_ms * 1e-3) / 1e9, 1), sweep_frac_hbm=round(B["solve"] / (t_ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 4),
            fallbacks=list(ks.fallbacks))), flush=True)
        del ks


if __name__ == "__main__":
    main()
