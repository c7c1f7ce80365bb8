#!/usr/bin/env python3
"""One interior-point iteration's reduced-system work (kkt_update! + affine and combined kkt_solve!,
/root/reference/src/kktsystem.jl:62-215) on the BASELINE workload, two ways:

  level C  hipkkt_kkt_system_*: iterate, right-hand sides and steps stay in HBM; per solve two scalars
           come back;
  level B  the same algebra on the host (numpy, as ipm.py does it) around hipkkt_kkt_setrhs / hipkkt_kkt_solve
           with HOST vectors: every solve moves its right-hand side and solution over PCIe.

Usage: python scripts/bench_system.py [--n 100000] [--steps 10]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import numpy as np
    import scipy.sparse as sp
    import torch
    from cuclarabel_amd import ipm, problems
    from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem

    dev = torch.device("cuda", 0)
    pb = problems.config2(n=args.n)
    ks = HipKKTSolver(pb.P, pb.A, pb.cones)
    ks.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    system = HipKKTSystem(ks)
    system.init(pb.q, pb.b)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(pb.n)
    s, z = pb.s0, pb.z0
    tau, kappa = 1.1, 0.9
    rhs = [rng.standard_normal(pb.n), rng.standard_normal(pb.m), rng.standard_normal(pb.m)]        # x, s, z parts
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_var = [T(x), T(s), T(z)]
    d_rhs = [T(rhs[0]), T(rhs[1]), T(rhs[2])]
    d_lhs = [torch.zeros(pb.n, dtype=torch.float64, device=dev), torch.zeros(pb.m, dtype=torch.float64, device=dev),
             torch.zeros(pb.m, dtype=torch.float64, device=dev)]
    P = lambda ts: [t.data_ptr() for t in ts]

    def iteration_c():
        assert system.update_dev(d_var[1].data_ptr(), d_var[2].data_ptr())
        for affine in (True, False):
            ok, dtau, dkappa = system.solve_dev(P(d_lhs), P([d_rhs[0], d_var[1] if affine else d_rhs[1], d_rhs[2]]),
                                                0.3, -0.1, P(d_var), tau, kappa, affine)
            assert ok
        return dtau

    def iteration_c_batched():
        # kkt_update! and the affine kkt_solve! as one call: the constant and the affine right-hand side share one
        # 2-column solve (hipkkt_kkt_system_update_and_solve_affine); the combined step follows alone
        ok, dtau, dkappa = system.update_and_solve_affine_dev(P(d_lhs), P([d_rhs[0], d_rhs[2]]), 0.3, -0.1, P(d_var), tau, kappa)
        assert ok
        ok, dtau, dkappa = system.solve_dev(P(d_lhs), P(d_rhs), 0.3, -0.1, P(d_var), tau, kappa, False)
        assert ok
        return dtau

    # level B with host vectors: the algebra of kkt_solve! in numpy (ipm.py's kkt_solve), solves through the C ABI
    Pt = sp.triu(sp.csc_matrix(pb.P), format="csc")
    Pfull = (Pt + sp.triu(Pt, 1).T).tocsr()
    cones = ipm._make_cones(pb.cones)

    def each(fn, *vecs):
        out = np.empty(pb.m)
        for c in cones:
            out[c.rng] = fn(c, *[v[c.rng] for v in vecs])
        return out

    def ksolve(rx, rz):
        ks.kktsolver_setrhs(rx, rz)
        xo, zo = np.zeros(pb.n), np.zeros(pb.m)
        assert ks.kktsolver_solve(xo, zo)
        return xo, zo

    def iteration_b():
        assert ks.kktsolver_update_from_sz(s, z)
        for c in cones:
            c.update_scaling(s[c.rng].copy(), z[c.rng].copy())
        x2, z2 = ksolve(-pb.q, pb.b)
        for affine in (True, False):
            const = s.copy() if affine else each(lambda c, d, zz: c.ds_from_dz_offset(d, zz), rhs[1], z)
            x1, z1 = ksolve(rhs[0], const - rhs[2])
            xi = x / tau
            tnum = 0.3 + 0.1 / tau + pb.q @ x1 + pb.b @ z1 + 2 * (xi @ (Pfull @ x1))
            xm = xi - x2
            tden = kappa / tau - pb.q @ x2 - pb.b @ z2 + xm @ (Pfull @ xm) - x2 @ (Pfull @ x2)
            dtau = tnum / tden
            dx, dz = x1 + dtau * x2, z1 + dtau * z2
            ds = -(each(lambda c, v: c.mul_Hs(v), dz) + const)
        return dtau

    def iteration_c_lazy():
        # the reference's own call sequence (kkt_update!, kkt_solve! :affine, kkt_solve! :combined) with the handle in lazy
        # mode: the constant-RHS solve rides with the affine one, one read-back per call
        return iteration_c()

    # The calls the Julia glue makes (integration/HipKKTExt.jl, part C): DefaultVariables and the right-hand sides live on
    # the HOST, the cones' scaling is the caller's (computed once here, outside the timing: it is Clarabel's own CPU work).
    # Per iteration: the cone data up (Hs, sparse-SOC u / v / eta^2, w, eta, lambda), per solve 3 + 3 vectors up and 3 down.
    host_cones = ipm._make_cones(pb.cones)
    for c in host_cones:
        c.update_scaling(s[c.rng].copy(), z[c.rng].copy())
    cone_data = ipm.host_cone_data(host_cones)

    def iteration_c_host():
        assert system.update_cones(*cone_data)
        for affine in (True, False):
            ok, step = system.solve(rhs[0], s if affine else rhs[1], rhs[2], 0.3, -0.1, x, s, z, tau, kappa, affine)
            assert ok
        return step[3]

    # r04: what integration/HipKKTExt.jl does now -- kkt_update! from the scaling alone (w, eta, lambda: the Hs blocks and the
    # sparse cones' u / v are formed on the device), the combined kkt_solve! re-using the affine one's variables, every
    # vector page-locked once (hipkkt_host_register)
    from cuclarabel_amd import _lib
    hv = dict(var=[x.copy(), s.copy(), z.copy()], rhs=[[rhs[0].copy(), s.copy(), rhs[2].copy()], [rhs[0].copy(), rhs[1].copy(), rhs[2].copy()]],
              lhs=[np.zeros(pb.n), np.zeros(pb.m), np.zeros(pb.m)],
              scal=[np.ascontiguousarray(cone_data[4]), np.ascontiguousarray(cone_data[5]), np.ascontiguousarray(cone_data[6]), np.zeros(0), np.zeros(0)])
    pinned = [a for a in hv["var"] + hv["rhs"][0] + hv["rhs"][1] + hv["lhs"] + hv["scal"][:3] if a.size and _lib.host_register(a)]
    upd_h, aff_h, com_h = system.prepared_host(hv["lhs"], hv["rhs"], 0.3, -0.1, hv["var"], tau, kappa, hv["scal"])

    def iteration_c_host_reduced():
        assert upd_h()
        for solve in (aff_h, com_h):
            ok, dtau, dkappa = solve()
            assert ok
        return dtau

    out = {}
    for name, fn in (("level_C_device_resident", iteration_c), ("level_C_lazy_two_calls", iteration_c_lazy),
                     ("level_C_batched_affine", iteration_c_batched), ("level_C_lazy_host_vectors", iteration_c_host),
                     ("level_C_lazy_host_vectors_reduced_pinned", iteration_c_host_reduced),
                     ("level_B_host_vectors", iteration_b)):
        system.set_lazy(name in ("level_C_lazy_two_calls", "level_C_lazy_host_vectors", "level_C_lazy_host_vectors_reduced_pinned"))
        fn(); fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = fn()
        torch.cuda.synchronize(dev)
        out[name] = dict(ms_per_iteration=(time.perf_counter() - t0) / args.steps * 1e3, dtau=float(last))
    out["agreement_dtau"] = abs(out["level_C_device_resident"]["dtau"] - out["level_B_host_vectors"]["dtau"])
    out["agreement_dtau_batched"] = abs(out["level_C_batched_affine"]["dtau"] - out["level_B_host_vectors"]["dtau"])
    out["agreement_dtau_lazy"] = abs(out["level_C_lazy_two_calls"]["dtau"] - out["level_B_host_vectors"]["dtau"])
    out["agreement_dtau_host"] = abs(out["level_C_lazy_host_vectors"]["dtau"] - out["level_B_host_vectors"]["dtau"])
    out["agreement_dtau_host_reduced"] = abs(out["level_C_lazy_host_vectors_reduced_pinned"]["dtau"] - out["level_B_host_vectors"]["dtau"])
    print(json.dumps(dict(workload=f"cfg2 n={args.n}: kkt_update! + 2 x kkt_solve! (3 KKT solves with refinement)", **out)))


if __name__ == "__main__":
    main()
