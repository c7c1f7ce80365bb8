#!/usr/bin/env python3
"""Batched right-hand sides against one factorisation ("batched RHS" mode).

Factorises the BASELINE workload (cfg2, n=100k) once and times hipkkt_kkt_solve_multi_dev for
several column counts: complete solves (with the reference's refinement rule per column) per
second, and the bare triangular sweeps per second with the section 8(d) multi-RHS byte count
B_solve(k) = 2 nnz(L) 12 + k 6 N 8.

N ranks (torch.distributed.run): the columns are dealt round-robin to the ranks, every rank
factorises the same K itself (4 ms, bit-identical; cheaper than broadcasting ~0.4 GB of factors
over one xGMI link) and solves its share; the solutions are combined with ONE all_gather of
N*k/world doubles per rank -- the only exchange this mode has.  Strong scaling: total k is fixed.

Usage: python scripts/bench_multirhs.py [--nrhs 1,8,64,512] [--n 100000] [--reps 5]
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
    ap.add_argument("--nrhs", default="1,8,64,512")
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--reps", type=int, default=9)
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    from cuclarabel_amd import _lib, problems
    from cuclarabel_amd.distributed import gather_columns, shard_columns
    from cuclarabel_amd.kktsolver import HipKKTSolver

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    pb = problems.config2(seed=1002, n=args.n)               # the same K on every rank
    ks = HipKKTSolver(pb.P, pb.A, pb.cones, settings=_lib.default_settings(device=local_rank))
    ks.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    assert ks.kktsolver_update_from_sz(pb.s0, pb.z0)
    info = ks.info
    N, nnzL = info["N"], info["nnzL"]
    rows = []
    for k in [int(t) for t in args.nrhs.split(",")]:
        mine = shard_columns(k, world, rank)
        km = len(mine)
        g = torch.Generator(device="cpu").manual_seed(7)
        RX = torch.randn(k, pb.n, dtype=torch.float64, generator=g)[mine].to(dev)      # row j = column j (contiguous)
        RZ = torch.randn(k, pb.m, dtype=torch.float64, generator=g)[mine].to(dev)
        LX = torch.zeros(max(km, 1), pb.n, dtype=torch.float64, device=dev)
        LZ = torch.zeros(max(km, 1), pb.m, dtype=torch.float64, device=dev)

        def run():
            if km:
                ok, ir = ks.kktsolver_solve_multi_dev(km, RX.data_ptr(), RZ.data_ptr(), LX.data_ptr(), LZ.data_ptr())
                assert ok
                return int(ir.sum())
            return 0

        def gather():
            if world > 1:
                gather_columns(LX, k)                         # RCCL: the one exchange of this mode

        run(); gather()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        # per-call times, MEDIAN reported: on this stack a call now and then takes 20-80 ms longer when the script has
        # just freed large host arrays (the per-k right-hand sides); the library's own loop shows no such calls
        irs = 0
        per_call = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            irs += run()
            gather()
            torch.cuda.synchronize(dev)
            per_call.append(time.perf_counter() - t0)
        if world > 1:
            dist.barrier()
        dt = float(np.median(per_call))
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        sweeps = 1.0 + irs / args.reps / max(km, 1)            # triangular sweeps per column incl. refinement
        B = 2 * nnzL * 12 + max(km, 1) * 6 * N * 8
        rows.append(dict(nrhs=k, per_rank=km, ms_per_call=dt * 1e3, solves_per_s=k / dt,
                         sweeps_per_column=sweeps, algorithmic_GBs_per_rank=B * sweeps / dt / 1e9))
    if rank == 0:
        print(json.dumps(dict(metric="KKT solves/s against one factorisation (fp64, with refinement)", n_gpus=world,
                              scaling="strong", N=N, nnzL=nnzL, rows=rows)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
