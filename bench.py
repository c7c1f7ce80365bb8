#!/usr/bin/env python3
"""bench.py -- KKT factorize+solve throughput (fp64) per IPM iteration on MI355X.

One "step" = one interior-point iteration's KKT work on the BASELINE.json workload
(configs[1]: random sparse SOCP, n=100k, m=200k, NN(100k)+1000xSOC(100)):
    1 x cone scaling + value scatter + static regularisation   (kktsolver_update!, a5-a9)
    1 x numeric LDL^T                                            (refactor!, a11)
    3 x solve with iterative refinement                          (kktsolver_solve!, a13-a14:
        constant RHS, affine RHS, combined RHS -- kktsystem.jl:87-88,170-171)
issued the way solver.jl:278-319 issues it -- kkt_update!, kkt_solve!(:affine), kkt_solve!(:combined), three SEPARATE
calls on the reduced-system layer (level C of the C ABI, hipkkt_kkt_system_*), including its right-hand-side
construction and step recovery -- with the iterate and the right-hand sides already resident in HBM.  The handle is in
lazy mode (hipkkt_kkt_system_set_lazy): kkt_update! leaves the constant-RHS solve to the affine kkt_solve!, which
sends both right-hand sides through the triangular sweeps as one 2-column solve.  N > 1 GPUs: every rank
runs the same workload on its own problem instance (seed + rank), no data-path collective
(independent problems shard: SURVEY.md section 8e) -> weak scaling; one tiny RCCL all-reduce
of the timing at the end.

Usage:  python bench.py --gpus N --steps K --warmup W
        (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6       # MI355X_MICROARCH.md: dense FP64 matrix peak
FP64_MFMA_MEASURED_TF = 49.1   # scripts/fp64_mfma_peak.hip on an MI355X box (profiles/r02a_fp64_peak.json)


def algorithmic_bytes(info):
    """bytes per unit of work on the scalar CSC layout the reference
    uses (8 B value + 4 B index per entry of L), with QDLDL's structural nnz(L)."""
    N, nnzK, nnzL = info["N"], info["nnzK"], info["nnzL"]
    nHs, slen = info["nHs"], info["sparse_soc_len"]
    B_solve = 2 * nnzL * 12 + 2 * (N + 1) * 4 + 6 * N * 8
    B_spmv = nnzK * 12 + (N + 1) * 4 + 3 * N * 8
    B_upd = (nHs + 2 * slen) * 20 + N * 28
    B_fact = nnzK * 12 + nnzL * 12 + N * 8
    return dict(solve=B_solve, spmv=B_spmv, update=B_upd, factor=B_fact)


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def native_oracle():
    """Build the oracle -O3 -march=native ON THIS BOX (BASELINE.md section 2's protocol) into a scratch directory and
    point tests/oracle_bindings at it; falls back to the portable in-tree build if the compiler is missing.
    Must run before tests.oracle_bindings is first imported.  Returns the build's description."""
    import subprocess
    import tempfile
    if "tests.oracle_bindings" in sys.modules:
        return "as loaded earlier"
    out = os.path.join(tempfile.mkdtemp(prefix="kktoracle_"), "libkktoracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", "NATIVE_OUT=" + out])
        os.environ["KKT_ORACLE_SO"] = out
        return "gcc -O3 -march=native -ffp-contract=off, built on this host"
    except (subprocess.CalledProcessError, OSError):
        so = os.path.join(ROOT, "oracle", "libkktoracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        return "gcc -O3 -ffp-contract=off (portable build; native build failed)"


def cpu_baseline(pb, n_units):
    """The oracle (C restatement of the reference's QDLDL path, 1 thread, AMD ordering) timed on
    this box's host: `n_units` IPM-iteration units after one warm-up.  TEST INFRASTRUCTURE used
    only as the reported baseline, never as the thing measured."""
    import numpy as np
    build = native_oracle()
    from tests.oracle_bindings import make_oracle
    from cuclarabel_amd import _lib
    from tests.oracle_bindings import min_degree
    o0 = make_oracle(pb, perm=np.arange(pb.n + pb.m + 2 * sum(1 for c in pb.cones if c.kind == 2 and c.dim > 4)))
    K0 = o0.K()
    del o0
    rng = np.random.default_rng(0)
    rhs = [(rng.standard_normal(pb.n), rng.standard_normal(pb.m)) for _ in range(3)]

    def timed_units(o, units):
        times = []
        for it in range(units + 2):
            t0 = time.perf_counter()
            assert o.update_scaling(pb.s0, pb.z0)
            assert o.kktsolver_update()
            for rx, rz in rhs:
                o.kktsolver_setrhs(rx, rz)
                ok, _, _ = o.kktsolver_solve()
                assert ok
            times.append(time.perf_counter() - t0)
        times = sorted(times[2:])
        return times[len(times) // 2]

    # The reference orders with AMD (directldl_qdldl.jl:18-25).  The oracle's own ordering is a plain minimum-degree
    # (oracle/kkt_oracle.c, orc_min_degree: ~12 % more fill than AMD on this workload); the permutation that gives the
    # CPU side its best case is the AMD of the product's HOST-side symbolic analysis (hipkkt_symbolic_analyse,
    # ordering = AMD; no GPU involved).  `value` is quoted with the latter -- the faster CPU figure -- and the oracle
    # on its own ordering is timed beside it.
    perm, _ = _lib.symbolic_analyse(K0, ordering=_lib.ORDER_AMD)
    o = make_oracle(pb, perm=perm)
    med = timed_units(o, n_units)
    nnzL_amd = int(o.nnzL)
    del o
    own = None
    if n_units >= 4:
        o = make_oracle(pb, perm=min_degree(K0))
        med_own = timed_units(o, max(2, n_units // 3))
        own = dict(value=1.0 / med_own, ms_per_unit=med_own * 1e3, nnzL=int(o.nnzL),
                   ordering="oracle's own minimum degree (orc_min_degree)")
        del o
    return dict(value=1.0 / med, unit="KKT factorize+solve/s", cores=1, kind="port",
                sample=f"{n_units} timed units (+2 warm-ups) of the same workload, median; oracle/kkt_oracle.c "
                       f"({build}; scalar up-looking LDL' as QDLDL, nnzL={nnzL_amd}), permutation = AMD computed by the "
                       f"product's host-side symbolic analysis (hipkkt_symbolic_analyse; the reference orders with AMD too), "
                       f"host {host_cpu_model()}, threads=1 of {os.cpu_count()}",
                ms_per_unit=med * 1e3, own_ordering=own)


def run_configs(args):
    """All BASELINE.json configurations at full size on one MI355X (results table of BASELINE.md section 3), one JSON
    line each: the unit of work of SURVEY.md 8(d) timed on the GPU through the C ABI -- `gpu_ms_per_unit`: level B, update +
    three single solves on given right-hand sides (the figure of rounds 1-2); `system_ms_per_unit`: the same unit through
    the reduced-system layer in lazy mode, i.e. the way the headline issues it --, the last solve compared with
    the CPU oracle on the same K, b, and -- this is bench.py's cpu_baseline leg for the other configurations -- the
    oracle timed on this box's host (1 thread, AMD ordering) on a bounded sample.  cfg4 is the per-GPU share of the
    batch: 8 problems back to back on one stream ("4"), or stacked block-diagonally on one handle ("4b")."""
    import numpy as np
    import torch
    from cuclarabel_amd import _lib, problems
    from cuclarabel_amd.kktsolver import HipKKTSolver

    dev = torch.device("cuda", 0)
    makers = {"1": lambda: [problems.config1()], "2": lambda: [problems.config2()], "3": lambda: [problems.config3()],
              "4": lambda: [problems.config4(j=j) for j in range(8)], "5": lambda: [problems.config5()],
              # cfg2's long-range variant (1 % of A's entries re-drawn over all columns): a small-world KKT graph,
              # nnz(L) 1.4e8, a 14 154-row root, 1.2 TFLOP per factorisation -- no CPU leg (the scalar oracle needs ~half an
              # hour per unit); its check is the refinement loop's own residual test on the un-regularised K
              "2lr": lambda: [problems.config2(long_range_frac=0.01)],
              # the per-GPU share of cfg4 as ONE block-diagonal problem: all 8 in the same per-level launches
              "4b": lambda: [problems.block_diagonal([problems.config4(j=j) for j in range(8)])]}
    import tempfile
    for c in args.configs.split(","):
        # the library reports a bounded wait that expired (overlap mode / persistent kernel falling back) on stderr only:
        # capture this configuration's stderr and put the count into its row, so that a slow row explains itself
        sys.stderr.flush()
        err_copy = os.dup(2)
        err_file = tempfile.TemporaryFile(mode="w+b")
        os.dup2(err_file.fileno(), 2)
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
        # The same unit through the reduced-system layer the way the headline issues it (level C, lazy: kkt_update!, then
        # kkt_solve!(:affine) -- the constant and the affine right-hand side as one 2-column solve where the structure's
        # sweeps take two columns --, then kkt_solve!(:combined); three separate calls with one status read-back each)
        from cuclarabel_amd.kktsolver import HipKKTSystem
        systems = []
        for pb, ks in zip(pbs, sol):
            ks.profile_enable(False)
            system = HipKKTSystem(ks)
            system.init(pb.q, pb.b)
            system.set_lazy(True)
            dd = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            var = [dd(rng.standard_normal(pb.n)), dd(pb.s0), dd(pb.z0)]
            rhs2 = [[dd(rng.standard_normal(k)) for k in (pb.n, pb.m, pb.m)] for _ in range(2)]
            lhs = [torch.zeros(k, dtype=torch.float64, device=dev) for k in (pb.n, pb.m, pb.m)]
            P = lambda ts: [t.data_ptr() for t in ts]
            calls = system.prepared(P(lhs), [P(rhs2[0]), P(rhs2[1])], 0.3, -0.1, P(var), 1.1, 0.9)
            systems.append((system, calls, (var, rhs2, lhs)))

        def unit_sys(calls):
            update, solve_affine, solve_combined = calls
            assert update()
            assert solve_affine()[0] and solve_combined()[0]

        for _, calls, _ in systems:
            unit_sys(calls)
            unit_sys(calls)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for _, calls, _ in systems:
                unit_sys(calls)
        torch.cuda.synchronize(dev)
        dt_sys = (time.perf_counter() - t0) / (args.steps * len(sol))
        fallbacks = [sum(k.fallbacks[i] for k in sol) for i in (0, 1)]       # (overlap mode, persistent sweep kernel): the ABI's counters
        row = dict(config=c, problems_on_gpu=len(sol), N=info["N"], nnzK=info["nnzK"], nnzL=info["nnzL"],
                   nnzL_stored=info["nnzL_stored"], levels=info["nlevels"], max_front=info["max_front"],
                   factor_gflop=info["factor_flops"] / 1e9, setup_s=setup_s / len(sol),
                   gpu_ms_per_unit=dt * 1e3 / per_unit, gpu_units_per_s=per_unit / dt, problems_per_handle=per_unit,
                   system_ms_per_unit=dt_sys * 1e3 / per_unit, system_units_per_s=per_unit / dt_sys,
                   factor_ms=prof["factor_ms"] / max(prof["n_factor"], 1),
                   trisolve_ms=prof["trisolve_ms"] / max(prof["n_trisolve"], 1),
                   update_ms=prof["update_ms"] / max(prof["n_update"], 1),
                   ir_rounds_per_unit=prof["ir_iterations"] / args.steps,
                   factor_TFLOPs=info["factor_flops"] / (prof["factor_ms"] / max(prof["n_factor"], 1) * 1e-3) / 1e12)
        if not args.no_cpu_baseline and c != "2lr":
            native_oracle()
            from tests.oracle_bindings import make_oracle           # the cpu_baseline leg: checker and baseline only
            pb, ks, st = pbs[0], sol[0], state[0]
            o0 = make_oracle(pb, perm=np.arange(info["N"]))
            perm, _ = _lib.symbolic_analyse(o0.K(), ordering=_lib.ORDER_AMD)
            del o0
            o = make_oracle(pb, perm=perm)
            times = []
            for it in range(args.cfg_cpu_units + 1):
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
        sys.stderr.flush()
        os.dup2(err_copy, 2)
        os.close(err_copy)
        err_file.seek(0)
        err_text = err_file.read().decode(errors="replace")
        err_file.close()
        sys.stderr.write(err_text)
        row["fallbacks"] = fallbacks
        row["fallbacks_reported"] = err_text.count("gave up")
        print(json.dumps(row), flush=True)
        # Host memory released while the GPU works can stall its queues for tens of ms on this stack (a factorisation of
        # the NEXT configuration once read 10.6 instead of 5.6 ms): release this configuration's arrays now, well before
        # the next timed region.
        o = xo = zo = x = z = None
        systems = None
        del sol, state, pbs, o, xo, zo, x, z
        import gc
        gc.collect()
        torch.cuda.synchronize(dev)
        time.sleep(0.5)



def spawn_workers(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) through
    torch.distributed.run, BEFORE this process has touched the GPU or imported torch.  The parent only relays the
    children's output and exit code; rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def traffic_summary():
    """HBM traffic per phase from the separate `rocprofv3 --pmc` passes of this same command
    (scripts/profile_bench.sh -> profiles/*_traffic_summary.json).  A summary is used only when it says which
    kernel sources it was measured on and they are the sources of this tree: after any kernel or schedule change
    the old figure would be silently stale next to freshly measured times."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "cuclarabel_amd", "csrc", "*.[ch]*"))):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(open(f, "rb").read())
    src = h.hexdigest()[:16]
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic_summary.json")), reverse=True):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("csrc_sha16") == src:
            return t, os.path.basename(f), src
    return None, None, src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="iter", choices=["iter", "problems", "rhs"],
                    help="iter (default, the headline): cfg2, one instance per rank, weak scaling.  problems: cfg4's "
                         "batch of --problems independent SOCPs (n=10k) dealt to the ranks, block-diagonal handles of "
                         "--per-handle, records gathered over RCCL, strong scaling.  rhs: --nrhs right-hand sides "
                         "against cfg2's factor dealt to the ranks, one all-gather of the solutions, strong scaling "
                         "(SURVEY.md section 8e)")
    ap.add_argument("--n", type=int, default=None, help="primal dimension (default: 100000 for cfg2, 10000 for cfg4)")
    ap.add_argument("--problems", type=int, default=64)
    ap.add_argument("--per-handle", type=int, default=None,
                    help="--mode problems: problems stacked block-diagonally per handle (default: a third of the rank's share, at "
                         "least 8 -- measured on one GPU with 64 problems: 3 concurrent handles of 22 give 4350 problems/s, 8 of 8 "
                         "3500-3700, 2 of 32 4300, one of 64 3850)")
    ap.add_argument("--nrhs", type=int, default=512)
    ap.add_argument("--rhs-block", type=int, default=32,
                    help="--mode rhs with N > 1: columns per rank solved (and exchanged) at a time; block i's (x, z) "
                         "solutions are all-gathered while block i + 1 is being solved")
    ap.add_argument("--ordering", default="nd", choices=["nd", "amd"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-units", type=int, default=10)
    ap.add_argument("--sequential-solves", action="store_true",
                    help="three single-column solves per step instead of (constant, affine) as one 2-column solve + combined")
    ap.add_argument("--sync-status", action="store_true",
                    help="every update / solve call returns its own status (one read-back each) instead of one deferred "
                         "status query per step")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the ranks rendezvous over gloo and exercise the launch / barrier / gather plumbing "
                         "with an empty step (CPU test of the --gpus N path); the JSON line carries dry_run: true")
    ap.add_argument("--configs", default=None,
                    help="e.g. 1,2,3,4,4b,5 (and 2lr, cfg2 with long-range couplings): time every listed configuration on one GPU (one JSON line each) "
                         "instead of the headline run")
    ap.add_argument("--cfg-cpu-units", type=int, default=2, help="--configs: timed oracle units per configuration")
    ap.add_argument("--no-scale-modes", action="store_true",
                    help="iter mode: skip the short `problems` and `rhs` passes whose rows go into scale_modes")
    ap.add_argument("--sequential-handles", action="store_true",
                    help="--mode problems: drive the handles one after the other (rounds 1-3's figure) instead of concurrently")
    ap.add_argument("--concurrent", action="store_true",
                    help="--configs 4: one stream and host thread per handle (their latency-bound steps side by side); --mode problems does "
                         "so by default")
    args = ap.parse_args()
    if args.configs:
        return run_configs(args)
    if args.mode == "problems" and not args.sequential_handles:
        args.concurrent = True
    # (Concurrent handles need no setting: the library admits one overlapped factorisation per device at a time -- a handle
    #  that finds the device's claim busy factorises level by level that time -- and a handle without the persistent sweep
    #  kernel's claim takes the chained sweep kernels.  Until round 3 this line set HIPKKT_FACTOR_OVERLAP=0.)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_workers(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from cuclarabel_amd import _lib, problems
    from cuclarabel_amd.distributed import assign_problems, gather_columns, gather_records, reduce_max, shard_columns

    def barrier():
        if world > 1:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize(dev)

    def timed(step, steps=None, warmup=None):
        for _ in range(args.warmup if warmup is None else warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps if steps is None else steps):
            step()
        barrier()
        return reduce_max(time.perf_counter() - t0, device=dev)      # MAX over ranks (RCCL, 8 bytes)

    # Host memory that is unmapped while the GPU works can stall its queues for tens of ms (observed on this stack with
    # scripts that free large host arrays between solves): nothing large is allocated or freed inside the timed regions,
    # and the collector does not run there either.
    import gc
    gc.collect()
    gc.disable()

    def make_solver(pb):
        from cuclarabel_amd.kktsolver import HipKKTSolver
        st = _lib.default_settings(device=local_rank, ordering=_lib.ORDER_ND if args.ordering == "nd" else _lib.ORDER_AMD)
        ks = HipKKTSolver(pb.P, pb.A, pb.cones, settings=st)
        ks.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ks.set_deferred_status(not args.sync_status)
        return ks

    def resident(pb, rng):
        rhs_host = [(rng.standard_normal(pb.n), rng.standard_normal(pb.m)) for _ in range(3)]
        dd = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        return dict(s=dd(pb.s0), z=dd(pb.z0), rhs=[(dd(a), dd(b)) for a, b in rhs_host],
                    # the first two right-hand sides as one column-major pair (row j of the (2, n) tensor = column j)
                    rx2=dd(np.stack([rhs_host[0][0], rhs_host[1][0]])), rz2=dd(np.stack([rhs_host[0][1], rhs_host[1][1]])),
                    lx2=torch.zeros(2, pb.n, dtype=torch.float64, device=dev), lz2=torch.zeros(2, pb.m, dtype=torch.float64, device=dev),
                    lx=torch.zeros(pb.n, dtype=torch.float64, device=dev), lz=torch.zeros(pb.m, dtype=torch.float64, device=dev))

    def unit(ks, st, batched=None, may_repeat=False):
        """one IPM iteration's KKT work: 1 update + refactor, 3 solves with refinement.  The constant and the affine
        right-hand side do not depend on each other (/root/reference/src/kktsystem.jl:87-88 vs :170-171), so they go
        through the triangular sweeps TOGETHER as a 2-column solve; the combined right-hand side, which depends on the
        affine step, follows alone.  (--sequential-solves: three single-column solves.)  Deferred status (the default):
        the calls only enqueue -- the refinement loop's decisions are taken on the device -- and ONE status query per
        step says whether every call succeeded exactly as the synchronous calls would have; anything else invalidates
        the run.  --sync-status: every call returns its own status (one host round trip each)."""
        batched = (not args.sequential_solves) if batched is None else batched
        if not ks.kktsolver_update_from_sz_dev(st["s"].data_ptr(), st["z"].data_ptr()):
            raise RuntimeError("factorisation failed")
        single = st["rhs"][2:] if batched else st["rhs"]
        if batched:
            ok, _ = ks.kktsolver_solve_multi_dev(2, st["rx2"].data_ptr(), st["rz2"].data_ptr(), st["lx2"].data_ptr(), st["lz2"].data_ptr())
            if not ok:
                raise RuntimeError("solve failed")
        for rx, rz in single:
            ks.kktsolver_setrhs_dev(rx.data_ptr(), rz.data_ptr())
            if not ks.kktsolver_solve_dev(st["lx"].data_ptr(), st["lz"].data_ptr()):
                raise RuntimeError("solve failed")
        if not args.sync_status:
            rc = ks.deferred_status()
            if rc == 2 and may_repeat:
                # HIPKKT_REFINEMENT_INCOMPLETE: the step's results are void and the library has adapted (one more speculative
                # refinement round, or -- under a profiler that serialises kernels -- the factorisation's overlap mode
                # switched off): the caller repeats the step.  Only the warm-up may do that; in the timed region it is an error.
                return unit(ks, st, batched, False)
            if rc != 0:
                raise RuntimeError("deferred status %d: a factorisation / solve of this step failed or stopped refining early" % rc)

    def make_system(pb, rng, lazy=True, stream=None):
        """The reduced-system layer on the device (level C: DefaultKKTSystem, kktsystem.jl:21-215) with a synthetic
        iterate and right-hand sides resident in HBM."""
        from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem
        st = _lib.default_settings(device=local_rank, ordering=_lib.ORDER_ND if args.ordering == "nd" else _lib.ORDER_AMD)
        ks = HipKKTSolver(pb.P, pb.A, pb.cones, settings=st)
        ks.set_stream(stream.cuda_stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream)
        system = HipKKTSystem(ks)
        system.init(pb.q, pb.b)
        system.set_lazy(lazy)
        dd = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        var = [dd(rng.standard_normal(pb.n)), dd(pb.s0), dd(pb.z0)]
        rhs = [[dd(rng.standard_normal(k)) for k in (pb.n, pb.m, pb.m)] for _ in range(2)]      # affine, combined
        lhs = [torch.zeros(k, dtype=torch.float64, device=dev) for k in (pb.n, pb.m, pb.m)]
        P = lambda ts: [t.data_ptr() for t in ts]
        st = dict(var=P(var), rhs=[P(rhs[0]), P(rhs[1])], lhs=P(lhs), keep=(var, rhs, lhs), tau=1.1, kappa=0.9)
        st["calls"] = system.prepared(st["lhs"], st["rhs"], 0.3, -0.1, st["var"], st["tau"], st["kappa"])      # argument marshalling done once
        return ks, system, st

    def unit_c(system, st):
        """One interior-point iteration's reduced-system work as solver.jl:278-319 issues it: kkt_update!, then
        kkt_solve!(:affine), then kkt_solve!(:combined) -- three separate calls, each returning its own status (and
        (dtau, dkappa)) to the host.  In lazy mode the first leaves its constant-RHS solve to the second, which sends
        both right-hand sides through the sweeps together; 3 solves with refinement per step either way."""
        update, solve_affine, solve_combined = st["calls"]
        if not update():
            raise RuntimeError("kkt_update! failed")
        for solve in (solve_affine, solve_combined):
            ok, dtau, dkappa = solve()
            if not ok:
                raise RuntimeError("kkt_solve! failed")
        return dtau

    base = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic"}
    if dry:
        base["dry_run"] = True

    # ------------------------------------------------------------------ mode problems (cfg4)
    def run_problems(steps, warmup):
        """the batch of independent SOCPs dealt to the ranks.  Returns rank 0's result row."""
        n = (args.n if args.mode == "problems" else None) or 10_000
        concurrent = args.concurrent or (args.mode != "problems" and not args.sequential_handles)
        mine = assign_problems(args.problems, world, rank)            # independent problems: no data-path collective
        if args.per_handle is None:
            args.per_handle = max(8, -(-len(mine) // 3))
        groups = [mine[i:i + args.per_handle] for i in range(0, len(mine), args.per_handle)]
        handles = []
        if not dry:
            rng = np.random.default_rng(0)
            for g in groups:
                pb = problems.block_diagonal([problems.config4(j=j, n=n) for j in g])
                # --concurrent: every handle on a stream and a host thread of its own -- a handle's step is a chain of
                # narrow, latency-bound launches that leaves most of the device idle; several chains side by side fill it.
                # (One handle at a time may use the persistent sweep kernel on a device; the others sweep level by level.)
                stream = torch.cuda.Stream(device=dev) if concurrent else None
                ks, system, st = make_system(pb, rng, stream=stream)
                st["keep_stream"] = stream
                handles.append((ks, st, g, system))

        pool = None
        if concurrent and not dry and len(handles) > 1:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=len(handles))
            torch.cuda.synchronize(dev)

        def step():
            if pool is not None:
                futs = [pool.submit(unit_c, system, st) for ks, st, _, system in handles]
                for f in futs:
                    f.result()
                return
            for ks, st, _, system in handles:
                unit_c(system, st)

        elapsed = timed(step, steps, warmup)
        # per-problem records [index, status, refinement rounds of the last solve, N of its handle]: the one exchange
        recs = []
        for gi, g in enumerate(groups):
            for j in g:
                recs.append([j, 1.0, float(handles[gi][0].last_ir_iterations) if handles else 0.0,
                             float(handles[gi][0].info["N"]) if handles else 0.0])
        table = gather_records(recs, args.problems, 4, device=dev)
        fb_sum = [sum(h[0].fallbacks[i] for h in handles) for i in (0, 1)]
        deferrals = sum(int(h[0].profile()["overlap_deferrals"]) for h in handles)
        if pool is not None:
            pool.shutdown()
        handles.clear()                 # (the handles' device memory goes back before the next pass)
        if rank == 0:
            assert bool((table[:, 1] == 1.0).all()), "a problem is missing from the gathered records"
            out = dict(base, metric="KKT factorize+solve/sec (fp64) per IPM iter, batch of %d independent SOCPs n=%d" % (args.problems, n),
                       value=args.problems * steps / elapsed, unit="KKT factorize+solve/s",
                       ms_per_step=elapsed / steps * 1e3, scaling="strong", steps=steps, warmup=warmup,
                       config={"workload": "cfg4: %d independent SOCPs n=%d m=%d NN(%d)+%dxSOC(100), dealt round-robin to the ranks, "
                                           "%d per block-diagonal handle; per problem 1 update + 1 LDL' refactor + 3 solves with IR per step"
                                           % (args.problems, n, 2 * n, n, n // 100, args.per_handle),
                               "calls": "kkt_update! / kkt_solve!(:affine) / kkt_solve!(:combined) as three separate level-C calls, lazy constant-RHS solve",
                               "fallbacks": fb_sum, "overlap_deferrals": deferrals,
                               "problems_per_rank": len(mine), "handles_per_rank": len(groups),
                               "handles_driven": "concurrently, a stream and a host thread each" if pool is not None else "one after the other",
                               "parallelism": "independent problems per GPU, record all-gather over RCCL"})
            return out
        return None

    # ------------------------------------------------------------------ mode rhs (batched right-hand sides, cfg2's factor)
    def run_rhs(steps, warmup):
        """the columns of one factor dealt to the ranks.  Returns rank 0's result row."""
        n = (args.n if args.mode == "rhs" else None) or 100_000
        k = args.nrhs
        mine = shard_columns(k, world, rank)
        km = len(mine)
        pn = n if not dry else 64
        if not dry:
            pb = problems.config2(seed=1002, n=n)                     # the same K on every rank: the factor is replicated
            ks = make_solver(pb)
            ks.set_deferred_status(False)      # solve_multi reports per call (2..8 columns per rank would otherwise only enqueue)
            if not ks.kktsolver_update_from_sz(pb.s0, pb.z0):
                raise RuntimeError("factorisation failed")
            g = torch.Generator(device="cpu").manual_seed(7)
            RX = torch.randn(k, pb.n, dtype=torch.float64, generator=g)[mine].to(dev)       # row j = column j (contiguous)
            RZ = torch.randn(k, pb.m, dtype=torch.float64, generator=g)[mine].to(dev)
            LZ = torch.zeros(max(km, 1), pb.m, dtype=torch.float64, device=dev)
        pm = pb.m if not dry else 96
        LX = torch.zeros(max(km, 1), pn, dtype=torch.float64, device=dev)
        if dry:
            LZ = torch.zeros(max(km, 1), pm, dtype=torch.float64, device=dev)
        rounds = [0]
        # N > 1: the (x, z) solutions -- the solution of K [x; z] = [rx; rz] is both -- are exchanged in blocks of
        # --rhs-block columns per rank, block i on the links while block i + 1 is solved (BlockedColumnGather)
        from cuclarabel_amd.distributed import BlockedColumnGather
        gather = BlockedColumnGather(k, (pn, pm), args.rhs_block if world > 1 else max(km, 1), device=dev)

        def step():
            for q0, q1 in gather.blocks():
                kb = max(0, min(q1, km) - q0)
                if kb and not dry:
                    esz = 8
                    ok, ir = ks.kktsolver_solve_multi_dev(kb, RX.data_ptr() + q0 * pb.n * esz, RZ.data_ptr() + q0 * pb.m * esz,
                                                          LX.data_ptr() + q0 * pb.n * esz, LZ.data_ptr() + q0 * pb.m * esz)
                    if not ok:
                        raise RuntimeError("solve failed")
                    if (ir < 0).any():
                        raise RuntimeError("refinement-round counts missing (deferred status?)")
                    rounds[0] += int(ir.sum())
                if world > 1:
                    # (every block has rows of its own in LX / LZ, untouched until finish(): sent in place, no copy)
                    gather.post(q0, q1, [LX[q0:q1], LZ[q0:q1]], copy=False)
            if world > 1:
                gather.finish()

        elapsed = timed(step, steps, warmup)
        if rank == 0:
            out = dict(base, metric="KKT solves/sec (fp64, with refinement) against one factorisation, %d right-hand sides" % k,
                       value=k * steps / elapsed, unit="KKT solves/s", ms_per_step=elapsed / steps * 1e3, steps=steps, warmup=warmup,
                       scaling="strong",
                       config={"workload": "cfg2's factor (n=%d) replicated per rank; %d right-hand sides dealt round-robin, "
                                           "hipkkt_kkt_solve_multi_dev on each share, solutions all-gathered" % (n, k),
                               "columns_per_rank": km, "columns_per_block": args.rhs_block if world > 1 else km,
                               "parallelism": "RHS columns per GPU; (x, z) solutions all-gathered over RCCL in blocks of "
                                              "columns, block i's exchange overlapping block i + 1's solves"})
            if not dry:
                info = ks.info
                sweeps = 1.0 + rounds[0] / max(km * (steps + warmup), 1)
                B = 2 * info["nnzL"] * 12 + km * 6 * info["N"] * 8         # B_solve(k)
                gbs = B * sweeps / (elapsed / steps) / 1e9
                out["roofline"] = dict(kernel="multi-column tri-solve", bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s",
                                       frac=gbs / HBM_PEAK_GBS, traffic=None,
                                       note="B_solve(k) = 2 nnz(L) 12 + k 6 N 8 per sweep pair, k = %d columns on this rank, "
                                            "%.2f sweep pairs per column incl. refinement; whole call timed" % (km, sweeps))
            return out
        return None

    if args.mode in ("problems", "rhs"):
        out = run_problems(args.steps, args.warmup) if args.mode == "problems" else run_rhs(args.steps, args.warmup)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    # ------------------------------------------------------------------ mode iter (the headline)
    n = args.n or 100_000
    def scale_modes():
        """The two sharded axes of SURVEY.md 8(e) in the SAME ranks, behind the headline pass: (i) the batch of independent
        SOCPs (cfg4) dealt to the ranks, (ii) 512 right-hand sides against cfg2's factor dealt to the ranks with the
        blocked (x, z) exchange -- so that the one command the driver runs at N = 1, 2, 4, 8 measures both (the headline
        pass itself is one independent cfg2 per rank: linear by construction).  Short passes; --no-scale-modes skips them."""
        if args.no_scale_modes:
            return None
        sm = {}
        row = run_problems(max(3, min(args.steps, 8)), 2)
        if row is not None:
            sm["problems"] = {k: row[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "scaling", "config")}
        row = run_rhs(max(2, min(args.steps, 4)), 1)
        if row is not None:
            sm["rhs"] = {k: row[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "scaling", "config", "roofline") if k in row}
        return sm if rank == 0 else None

    if dry:
        elapsed = timed(lambda: None)
        sm = scale_modes()
        if rank == 0:
            print(json.dumps(dict(base, metric="KKT factorize+solve/sec (fp64) per IPM iter, 100k-var SOCP",
                                  value=world * args.steps / max(elapsed, 1e-9), unit="KKT factorize+solve/s",
                                  ms_per_step=elapsed / args.steps * 1e3, scaling="weak",
                                  config={"workload": "dry run (no GPU work)"}, scale_modes=sm)), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    pb = problems.config2(seed=1002 + rank, n=n)
    t0 = time.perf_counter()
    ks, system, stc = make_system(pb, np.random.default_rng(0), lazy=not args.sequential_solves)
    setup_s = time.perf_counter() - t0
    st = resident(pb, np.random.default_rng(0))
    # (set-up has freed large host arrays; let the unmapping settle before any timed GPU work: see the note at gc.disable)
    gc.collect()
    torch.cuda.synchronize(dev)
    time.sleep(0.5)

    # ---- the timed pass: level C, three separate calls per step, no instrumentation
    for _ in range(args.warmup):
        unit_c(system, stc)
    fb0 = ks.fallbacks
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        unit_c(system, stc)
    barrier()
    elapsed = reduce_max(time.perf_counter() - t0, device=dev)
    fb1 = ks.fallbacks
    if fb1 != fb0:
        raise RuntimeError("a fallback was taken inside the timed region (overlap, top) %s -> %s" % (fb0, fb1))
    # ---- the same step with the constant-RHS solve issued by kkt_update! itself (eager: three single-column solves)
    system.set_lazy(False)
    unit_c(system, stc)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        unit_c(system, stc)
    barrier()
    elapsed_seq = time.perf_counter() - t0
    # the per-phase breakdown comes from an instrumented pass over the same steps (hipEvents around every phase on the
    # kernels' stream: ~16 event records per step, each a marker packet the stream waits on -- instrumentation that does
    # not belong in the timed region), with single-column solves only: every phase launch is then one single-column
    # sweep / residual, which is what the roofline objects are quoted for.
    ks.profile_enable(True)
    ks.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        unit_c(system, stc)
    barrier()
    elapsed_profiled = time.perf_counter() - t0
    prof = ks.profile()
    ks.profile_enable(False)
    # ---- for continuity with rounds 1-2: the bare level-B sequence (update, three solves on given right-hand sides, no
    #      right-hand-side construction / step recovery), deferred status, constant + affine as an explicit 2-column call
    level_b = {}
    system.set_lazy(not args.sequential_solves)
    ks.set_deferred_status(not args.sync_status)
    # (--sequential-solves is the profiling configuration -- scripts/profile_bench.sh: every sweep of the run single-column,
    #  so that traffic per sweep pair is well defined -- and leaves the level-B passes out)
    for name, batched in (() if args.sequential_solves else (("batched_2col", True), ("three_single_solves", False))):
        unit(ks, st, batched, may_repeat=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            unit(ks, st, batched)
        barrier()
        level_b[name] = (time.perf_counter() - t0) / args.steps * 1e3
    ks.set_deferred_status(False)
    # ---- the drop-in figure: the same three level-C calls with HOST vectors, as integration/HipKKTExt.jl issues them for
    #      the reference's DefaultVariables (kktsystem.jl:135-143 takes host vectors): kkt_update! from the cones' scaling
    #      (w, eta, lambda: the Hs blocks and the sparse cones' u / v are formed on the device), the two kkt_solve! through
    #      the *_host entry points (the combined one re-using the affine one's variables), every array page-locked once
    #      as the glue does at construction.  PCIe-inclusive: never `value`.
    host_ms = {}
    if not args.sequential_solves:
        system.set_lazy(True)
        lam_h, psd_h = ks.scaling()
        w_h, eta_h = ks.scaling_w()
        hs_full = ks.get_Hs()
        rngh = np.random.default_rng(5)
        hv = dict(var=[rngh.standard_normal(pb.n), pb.s0.copy(), pb.z0.copy()],
                  rhs=[[rngh.standard_normal(k) for k in (pb.n, pb.m, pb.m)] for _ in range(2)],
                  lhs=[np.zeros(k) for k in (pb.n, pb.m, pb.m)],
                  scal=[np.ascontiguousarray(w_h), np.ascontiguousarray(eta_h), np.ascontiguousarray(lam_h), np.zeros(0), np.zeros(0)])
        pinned = [a for a in hv["var"] + hv["rhs"][0] + hv["rhs"][1] + hv["lhs"] + hv["scal"][:3] if a.size]
        n_pinned = sum(1 for a in pinned if _lib.host_register(a))
        calls_h = system.prepared_host(hv["lhs"], hv["rhs"], 0.3, -0.1, hv["var"], stc["tau"], stc["kappa"], hv["scal"])
        for tag, calls in (("lazy_reduced_upload_pinned", calls_h),):
            st_h = dict(calls=calls)
            unit_c(system, st_h)
            unit_c(system, st_h)
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                unit_c(system, st_h)
            barrier()
            host_ms[tag] = (time.perf_counter() - t0) / args.steps * 1e3
        for a in pinned:
            _lib.host_unregister(a)
        host_ms["arrays_page_locked"] = n_pinned
        del hs_full
    fb2 = ks.fallbacks
    sm = scale_modes()

    if rank == 0:
        info = ks.info
        B = algorithmic_bytes(info)
        phases = {}
        for name, ms_key, n_key, byt in (("factor", "factor_ms", "n_factor", B["factor"]),
                                         ("trisolve", "trisolve_ms", "n_trisolve", B["solve"]),
                                         ("residual", "residual_ms", "n_residual", B["spmv"]),
                                         ("update", "update_ms", "n_update", B["update"])):
            cnt = max(prof[n_key], 1)
            avg_ms = prof[ms_key] / cnt
            gbs = byt / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            phases[name] = dict(total_ms=prof[ms_key], launches=prof[n_key], avg_ms=avg_ms,
                                algorithmic_bytes=byt, achieved_GBs=gbs, frac=gbs / HBM_PEAK_GBS)
        # flop-side view of the factorisation (dense Schur / panel work on v_mfma_f64_16x16x4_f64)
        phases["factor"]["flops"] = info["factor_flops"]
        phases["factor"]["achieved_TFLOPs"] = info["factor_flops"] / (phases["factor"]["avg_ms"] * 1e-3) / 1e12
        phases["factor"]["frac_fp64_mfma_peak"] = phases["factor"]["achieved_TFLOPs"] / FP64_MFMA_PEAK_TF
        tsum, traffic_src, src_hash = (None, None, None)
        if pb.n == 100_000:
            tsum, traffic_src, src_hash = traffic_summary()

        def roofline_of(phase):
            d = phases[phase]
            traffic = None
            if tsum is not None:
                key = "trisolve_hbm_MB_per_solve" if phase == "trisolve" else "factor_hbm_MB_per_factorisation"
                traffic = tsum[key] * 1024.0 * 1024.0
            r = dict(kernel=phase, bound="hbm", achieved=d["achieved_GBs"], peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=d["frac"], traffic=traffic, traffic_source=traffic_src)
            if phase == "factor" and d["frac_fp64_mfma_peak"] > d["frac"]:
                # roofline.achieved(factor) = max(B_fact/t/8 TB/s, F_fact/t/78.6 TF)
                r.update(bound="mfma", achieved=d["achieved_TFLOPs"], peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                         frac=d["frac_fp64_mfma_peak"],
                         # the box's own v_mfma_f64_16x16x4_f64 rate (scripts/fp64_mfma_peak.hip, profiles/r02a_fp64_peak.json)
                         frac_vs_measured_peak=d["achieved_TFLOPs"] / FP64_MFMA_MEASURED_TF, measured_peak=FP64_MFMA_MEASURED_TF)
            r["note"] = ("phase = all kernel launches of one %s (a dependency chain over the elimination-tree levels, "
                         "so latency- rather than roofline-bound); algorithmic bytes/flops per SURVEY.md 8(d); "
                         "traffic = FETCH_SIZE+WRITE_SIZE bytes per launch of the phase from the committed PMC passes "
                         "measured on these kernel sources (null: no summary for csrc hash %s)" % (phase, src_hash))
            return r

        dominant = max(("factor", "trisolve"), key=lambda k: phases[k]["total_ms"])
        roofline = roofline_of(dominant)
        rt = roofline_of("trisolve")
        # the same time against the reference ordering's fill (AMD: what QDLDL would hold) -- the ND ordering's extra
        # fill is this build's own choice (tree height), so the fraction is quoted both ways
        if args.ordering == "nd" and pb.n == 100_000:
            try:
                _, amd = _lib.symbolic_analyse(ks.KKT(), ordering=_lib.ORDER_AMD)
                b_amd = 2 * amd["nnzL"] * 12 + 2 * (info["N"] + 1) * 4 + 6 * info["N"] * 8
                rt["frac_vs_amd_fill"] = b_amd / (phases["trisolve"]["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                rt["nnzL_amd"] = amd["nnzL"]
            except Exception as e:                                       # diagnostic only
                rt["frac_vs_amd_fill"] = None
                rt["amd_note"] = str(e)
        phase_sum = sum(phases[k]["total_ms"] for k in phases) / max(args.steps, 1)
        out = dict(base, **{
            "metric": "KKT factorize+solve/sec (fp64) per IPM iter, 100k-var SOCP",
            "value": world * args.steps / elapsed,
            "unit": "KKT factorize+solve/s",
            "ms_per_step": elapsed / args.steps * 1e3,
            "scaling": "weak",
            "config": {"workload": "cfg2: random sparse SOCP n=%d m=%d NN(%d)+%dxSOC(100), P=diag, A 4 nnz/row local "
                                   "window; 1 update + 1 LDL' refactor + 3 solves with IR per step" %
                                   (pb.n, pb.m, pb.n, pb.n // 100),
                       "N": info["N"], "nnzK": info["nnzK"], "nnzL": info["nnzL"], "nnzL_stored": info["nnzL_stored"],
                       "fill_ratio": info["nnzL"] / info["nnzK"], "etree_height": info["etree_height"],
                       "nsuper": info["nsuper"], "levels": info["nlevels"], "max_front": info["max_front"],
                       "factor_flops": info["factor_flops"], "ordering": args.ordering,
                       "ir_rounds_per_step": prof["ir_iterations"] / max(args.steps, 1),
                       "setup_s": setup_s, "parallelism": "independent problems per GPU",
                       "calls": "level C (hipkkt_kkt_system_*): kkt_update!, kkt_solve!(:affine), kkt_solve!(:combined) as "
                                "three separate calls per step (solver.jl:278-319), each returning its status and "
                                "(dtau, dkappa) to the host; right-hand-side construction and step recovery included; "
                                + ("eager constant-RHS solve" if args.sequential_solves else
                                   "lazy mode: kkt_update! leaves the constant-RHS solve to the affine kkt_solve! (one 2-column solve)"),
                       "csrc_sha16": traffic_summary()[2]},
            "roofline": roofline,
            "roofline_trisolve": rt,      # the north-star's named roofline target
            "phases": phases,
            "ms_per_step_sequential_solves": elapsed_seq / args.steps * 1e3,
            "solves": ("3 single-column solves per step" if args.sequential_solves else
                       "constant + affine right-hand sides share one 2-column solve (issued by the affine kkt_solve! in lazy "
                       "mode), combined alone (3 solves per step); ms_per_step_sequential_solves is the same step with "
                       "kkt_update! solving the constant right-hand side itself (three single-column solves)"),
            "fallbacks": {"overlap": fb2[0], "top": fb2[1], "in_timed_region": 0},
            # value updates (= factorisations) this process has run, for per-launch averages of profiler counters
            "updates_in_run": args.warmup + 3 * args.steps + 1 + (0 if args.sequential_solves else 2 * (args.steps + 1) + args.steps + 2),
            # rounds 1-2 quoted this: level B alone (no right-hand-side construction / step recovery), deferred status
            "level_B_ms_per_step": level_b,
            # The figure a caller with HOST vectors gets (the Julia glue: DefaultVariables are Vector{T}) -- the same three
            # lazy level-C calls through the *_host entry points, PCIe included; `value` / `ms_per_step` are the resident ones.
            "ms_per_step_host_vectors": host_ms.get("lazy_reduced_upload_pinned"),
            "host_vectors": dict(host_ms, calls="hipkkt_kkt_system_update_scaling (w, eta, lambda up; Hs, u, v formed on the device), "
                                                "hipkkt_kkt_system_solve_host x 2 (combined step re-uses the affine step's variables), "
                                                "arrays page-locked once (hipkkt_host_register)") if host_ms else None,
            # SURVEY.md 8(e)'s two sharded axes, measured in the same ranks behind the headline pass (None: --no-scale-modes)
            "scale_modes": sm,
            "ms_per_step_instrumented": elapsed_profiled / args.steps * 1e3,
            "ms_per_step_outside_phases": elapsed_profiled / args.steps * 1e3 - phase_sum,
        })
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pb, args.cpu_units)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
