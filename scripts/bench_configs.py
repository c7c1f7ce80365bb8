#!/usr/bin/env python3
"""All BASELINE.json configurations at full size on one MI355X (results table of BASELINE.md §3).

For each configuration: the unit of work of SURVEY.md 8(d) (1 update + 1 refactor + 3 solves with
refinement, inputs resident in HBM) timed on the GPU through the C ABI, the solution of the last
solve compared with the CPU oracle on the same K, b (relative inf-norm error), and the oracle timed
on this box's host (1 thread, AMD ordering) on a bounded sample.  cfg4 is run as the per-GPU share
of the batch: 8 independent problems on one GPU, back to back on one stream (--concurrent: one HIP
stream and host thread per problem; measured 284 vs 252 units/s).

Usage: python scripts/bench_configs.py [--configs 1,2,3,4,4b,5] [--steps 10] [--cpu-units 2]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2,3,4,4b,5")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cpu-units", type=int, default=2)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--concurrent", action="store_true",
                    help="cfg4: one HIP stream + host thread per problem instead of back to back on one stream")
    args = ap.parse_args()
    import numpy as np
    import torch
    from cuclarabel_amd import _lib, problems
    from cuclarabel_amd.kktsolver import HipKKTSolver
    from tests.oracle_bindings import make_oracle

    dev = torch.device("cuda", 0)
    makers = {"1": lambda: [problems.config1()], "2": lambda: [problems.config2()], "3": lambda: [problems.config3()],
              "4": lambda: [problems.config4(j=j) for j in range(8)], "5": lambda: [problems.config5()],
              # the per-GPU share of cfg4 as ONE block-diagonal problem: all 8 in the same per-level launches
              "4b": lambda: [problems.block_diagonal([problems.config4(j=j) for j in range(8)])]}
    for c in args.configs.split(","):
        pbs = makers[c]()
        per_unit = 8 if c == "4b" else 1
        t0 = time.perf_counter()
        sol = [HipKKTSolver(pb.P, pb.A, pb.cones) for pb in pbs]
        setup_s = time.perf_counter() - t0
        rng = np.random.default_rng(0)
        state = []
        concurrent = args.concurrent and len(sol) > 1
        for pb, ks in zip(pbs, sol):
            if not concurrent:
                ks.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            rhs = [(rng.standard_normal(pb.n), rng.standard_normal(pb.m)) for _ in range(3)]
            state.append(dict(s=torch.from_numpy(pb.s0).to(dev), z=torch.from_numpy(pb.z0).to(dev),
                              rhs=[(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in rhs],
                              rhs_host=rhs, lx=torch.zeros(pb.n, dtype=torch.float64, device=dev),
                              lz=torch.zeros(pb.m, dtype=torch.float64, device=dev)))

        def unit(ks, st):
            assert ks.kktsolver_update_from_sz_dev(st["s"].data_ptr(), st["z"].data_ptr())
            for rx, rz in st["rhs"]:
                ks.kktsolver_setrhs_dev(rx.data_ptr(), rz.data_ptr())
                assert ks.kktsolver_solve_dev(st["lx"].data_ptr(), st["lz"].data_ptr())

        for ks, st in zip(sol, state):
            unit(ks, st)
            unit(ks, st)
        for ks in sol:
            ks.profile_enable(True)
            ks.profile_reset()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if concurrent:
            import threading

            def worker(ks, st):
                for _ in range(args.steps):
                    unit(ks, st)
                ks.synchronize()
            th = [threading.Thread(target=worker, args=(ks, st)) for ks, st in zip(sol, state)]
            for t in th:
                t.start()
            for t in th:
                t.join()
        else:
            for _ in range(args.steps):
                for ks, st in zip(sol, state):
                    unit(ks, st)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / (args.steps * len(sol))
        prof = sol[0].profile()
        info = sol[0].info
        row = dict(config=c, problems_on_gpu=len(sol), N=info["N"], nnzK=info["nnzK"], nnzL=info["nnzL"],
                   nnzL_stored=info["nnzL_stored"], levels=info["nlevels"], max_front=info["max_front"],
                   factor_gflop=info["factor_flops"] / 1e9, setup_s=setup_s / len(sol),
                   gpu_ms_per_unit=dt * 1e3 / per_unit, gpu_units_per_s=per_unit / dt, problems_per_handle=per_unit,
                   factor_ms=prof["factor_ms"] / max(prof["n_factor"], 1),
                   trisolve_ms=prof["trisolve_ms"] / max(prof["n_trisolve"], 1),
                   update_ms=prof["update_ms"] / max(prof["n_update"], 1),
                   ir_rounds_per_unit=prof["ir_iterations"] / args.steps,
                   factor_TFLOPs=info["factor_flops"] / (prof["factor_ms"] / max(prof["n_factor"], 1) * 1e-3) / 1e12)
        if not args.no_cpu:
            pb, ks, st = pbs[0], sol[0], state[0]
            o0 = make_oracle(pb, perm=np.arange(info["N"]))
            perm, _ = _lib.symbolic_analyse(o0.K(), ordering=_lib.ORDER_AMD)
            del o0
            o = make_oracle(pb, perm=perm)
            times = []
            for it in range(args.cpu_units + 1):
                t0 = time.perf_counter()
                assert o.update_scaling(pb.s0, pb.z0) and o.kktsolver_update()
                for rx, rz in st["rhs_host"]:
                    o.kktsolver_setrhs(rx, rz)
                    ok, xo, zo = o.kktsolver_solve()
                    assert ok
                times.append(time.perf_counter() - t0)
            med = sorted(times[1:])[len(times[1:]) // 2]
            x = st["lx"].cpu().numpy()
            z = st["lz"].cpu().numpy()
            scale = max(np.abs(xo).max(), np.abs(zo).max())
            row.update(cpu_ms_per_unit=med * 1e3 / per_unit, cpu_units_per_s=per_unit / med, cpu_nnzL=int(o.nnzL), cpu_cores=1,
                       speedup=med / dt, rel_err_vs_oracle=float(max(np.abs(x - xo).max(), np.abs(z - zo).max()) / scale))
        print(json.dumps(row), flush=True)
        del sol, state


if __name__ == "__main__":
    main()
