#!/usr/bin/env python3
"""One problem through level B on the GPU, timed, and compared with the oracle: try_problem.py "<problems.maker(...)>" [--no-oracle]"""
import sys, time
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np
from cuclarabel_amd import problems
from cuclarabel_amd.kktsolver import HipKKTSolver
from tests.oracle_bindings import make_oracle
pb = eval("problems." + sys.argv[1])
t0 = time.time()
ks = HipKKTSolver(pb.P, pb.A, pb.cones)
print("setup %.2f s" % (time.time() - t0), {k: ks.info[k] for k in ("N", "nnzL", "nlevels", "max_front", "factor_flops")}, flush=True)
rng = np.random.default_rng(1)
rx, rz = rng.standard_normal(pb.n), rng.standard_normal(pb.m)
x, z = np.zeros(pb.n), np.zeros(pb.m)
for rep in range(2):
    t0 = time.time()
    assert ks.kktsolver_update_from_sz(pb.s0, pb.z0)
    ks.kktsolver_setrhs(rx, rz)
    assert ks.kktsolver_solve(x, z)
    print("unit (host vectors) %.2f ms, ir %d, fallbacks %s" % ((time.time() - t0) * 1e3, ks.last_ir_iterations, ks.fallbacks), flush=True)
ks.profile_enable(True); ks.profile_reset()
for rep in range(3):
    assert ks.kktsolver_update_from_sz(pb.s0, pb.z0)
    ks.kktsolver_setrhs(rx, rz); assert ks.kktsolver_solve(x, z)
p = ks.profile()
print("factor %.3f ms (%.2f TF), sweep pair %.3f ms" % (p["factor_ms"] / p["n_factor"], ks.info["factor_flops"] / (p["factor_ms"] / p["n_factor"] * 1e-3) / 1e12, p["trisolve_ms"] / p["n_trisolve"]), flush=True)
if "--no-oracle" not in sys.argv:
    o = make_oracle(pb, perm=ks.perm())
    t0 = time.time()
    assert o.update_scaling(pb.s0, pb.z0) and o.kktsolver_update()
    o.kktsolver_setrhs(rx, rz)
    ok, xo, zo = o.kktsolver_solve()
    print("oracle %.1f s, ir %d, rel err %.2e" % (time.time() - t0, o.last_ir_iters, max(np.abs(x - xo).max(), np.abs(z - zo).max()) / max(np.abs(xo).max(), np.abs(zo).max())), flush=True)
