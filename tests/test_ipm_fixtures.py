"""End-to-end known answers of the reference (test/OptTests/basic_*.jl, linear_solvers.jl) through
the IPM test driver: on the CPU with the oracle backend (pins the oracle), on the GPU with the HIP
backend (parity of the boundary, same tolerance as the reference's own backend test: atol 1e-3)."""
import numpy as np
import pytest

from cuclarabel_amd import ipm
from tests.golden.reference_fixtures import ALL
from tests.ipm_backends import OracleBackend

TOL = 1e-3          # linear_solvers.jl: tol


def _check(res, exp):
    assert res.status == exp["status"], (res.status, res.history[-1])
    if "x" in exp:
        assert np.linalg.norm(res.x - exp["x"]) < TOL
    if "obj" in exp:
        assert abs(res.obj_val - exp["obj"]) < TOL
        assert abs(res.obj_val_dual - exp["obj"]) < TOL


@pytest.mark.parametrize("name", sorted(ALL))
def test_reference_known_answers_with_oracle_backend(name):
    P, q, A, b, cones, exp = ALL[name]()
    res = ipm.solve(P, q, A, b, cones, OracleBackend(P, A, cones))
    _check(res, exp)
    assert res.iterations < 30


def test_driver_on_sparse_soc_lasso_like_problem():
    # SOC(dim > 4) -> sparse expansion end to end (the reference's only such test is socp-lasso.jl,
    # whose data depends on Julia's RNG; same structure, numpy data, checked against optimality conditions)
    rng = np.random.default_rng(12345)
    import scipy.sparse as sp
    from cuclarabel_amd.cones import NonnegativeConeT, SecondOrderConeT
    n, k = 8, 30
    B = rng.standard_normal((k, n)); y = rng.standard_normal(k)
    # min t  s.t. ||B x - y|| <= t, x >= -1 : variables (x, t)
    A = np.zeros((n + k + 1, n + 1)); b = np.zeros(n + k + 1)
    A[:n, :n] = -np.eye(n); b[:n] = 1.0
    A[n, n] = -1.0
    A[n + 1:, :n] = -B; b[n + 1:] = -y
    q = np.r_[np.zeros(n), 1.0]
    P = sp.csc_matrix((n + 1, n + 1))
    cones = [NonnegativeConeT(n), SecondOrderConeT(k + 1)]
    res = ipm.solve(P, q, sp.csc_matrix(A), b, cones, OracleBackend(P, sp.csc_matrix(A), cones))
    assert res.status == "SOLVED"
    x = res.x[:n]
    xl = np.linalg.lstsq(B, y, rcond=None)[0]
    if np.all(xl >= -1):
        assert abs(res.x[n] - np.linalg.norm(B @ xl - y)) < 1e-6
    assert abs(res.x[n] - np.linalg.norm(B @ x - y)) < 1e-6
    assert abs(res.obj_val - res.obj_val_dual) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ALL))
def test_reference_known_answers_with_hip_backend(name):
    P, q, A, b, cones, exp = ALL[name]()
    res = ipm.solve(P, q, A, b, cones, ipm.HipBackend(P, A, cones))
    _check(res, exp)
    # and it walks the same path as the oracle-backed run
    ref = ipm.solve(P, q, A, b, cones, OracleBackend(P, A, cones))
    assert res.iterations == ref.iterations
    if exp["status"] == "SOLVED":       # (an infeasibility certificate is a ray: on the singular KKT systems of those
        np.testing.assert_allclose(res.x, ref.x, atol=1e-7)       # cases its direction depends on the elimination order)


@pytest.mark.gpu
def test_ipm_on_cfg2_small_hip_vs_oracle():
    from cuclarabel_amd import problems
    pb = problems.config2(n=1000)
    res = ipm.solve(pb.P, pb.q, pb.A, pb.b, pb.cones, ipm.HipBackend(pb.P, pb.A, pb.cones))
    ref = ipm.solve(pb.P, pb.q, pb.A, pb.b, pb.cones, OracleBackend(pb.P, pb.A, pb.cones))
    # without equilibration this instance stalls at ~1e-8 primal residual on either backend
    # (INSUFFICIENT_PROGRESS -> ALMOST_SOLVED in info_post_process!); what is compared is the path:
    # (measured: both backends agree to 6+ digits through iteration 11, reach pres 3e-9 / dres 4e-8 there,
    #  then the primal residual grows on both and the exit status depends on round-off)
    for r in (res, ref):
        assert min(max(h["pres"], h["dres"]) for h in r.history) < 1e-7
    k = min(len(res.history), len(ref.history), 10)
    assert k == 10
    for a, b in zip(res.history[:k], ref.history[:k]):        # first ten iterations walk together
        assert abs(a["pcost"] - b["pcost"]) < 1e-6 * max(1, abs(b["pcost"]))
        assert abs(a["mu"] - b["mu"]) < 1e-6 * max(1e-12, abs(b["mu"]))


# ---- level C: the reduced-system layer (kktsystem.jl) on the device ----------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ALL))
def test_reference_known_answers_with_device_resident_kkt_system(name):
    """Same fixtures, but kkt_update! / kkt_solve_initial_point! / kkt_solve! run on the device
    (hipkkt_kkt_system_*): the driver only sees iterates and steps."""
    P, q, A, b, cones, exp = ALL[name]()
    res = ipm.solve(P, q, A, b, cones, ipm.HipSystemBackend(P, A, cones))
    _check(res, exp)
    ref = ipm.solve(P, q, A, b, cones, OracleBackend(P, A, cones))
    assert res.iterations == ref.iterations
    if exp["status"] == "SOLVED":
        np.testing.assert_allclose(res.x, ref.x, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ALL))
def test_reference_known_answers_with_batched_constant_and_affine_solves(name):
    """Level C with kkt_update! and the affine kkt_solve! as one call: the constant and the affine right-hand side
    go through the triangular sweeps together (one 2-column solve per iteration).  Same
    known answers, same iteration count, same iterates as the call-by-call sequence."""
    P, q, A, b, cones, exp = ALL[name]()
    res = ipm.solve(P, q, A, b, cones, ipm.HipSystemBackend(P, A, cones, batch_affine=True))
    _check(res, exp)
    ref = ipm.solve(P, q, A, b, cones, ipm.HipSystemBackend(P, A, cones))
    assert res.iterations == ref.iterations
    if exp["status"] == "SOLVED":
        np.testing.assert_allclose(res.x, ref.x, atol=1e-9)


@pytest.mark.gpu
def test_batched_update_and_affine_solve_equals_the_two_calls():
    """hipkkt_kkt_system_update_and_solve_affine against hipkkt_kkt_system_update + hipkkt_kkt_system_solve(:affine) on
    a problem with every cone kind: the step must agree to round-off (the 2-column sweep runs the same arithmetic per column)."""
    from cuclarabel_amd import problems
    from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem
    pb = problems.small_mixed(seed=43, psds=(2, 3, 6), socs=(3, 4, 6, 15))
    rng = np.random.default_rng(19)
    x = rng.standard_normal(pb.n)
    rhs_x, rhs_z = rng.standard_normal(pb.n), rng.standard_normal(pb.m)
    out = []
    for batched in (False, True):
        ks = HipKKTSolver(pb.P, pb.A, pb.cones)
        system = HipKKTSystem(ks)
        system.init(pb.q, pb.b)
        if batched:
            ok, step = system.update_and_solve_affine(rhs_x, rhs_z, 0.4, -0.2, x, pb.s0, pb.z0, 1.3, 0.7)
        else:
            assert system.update(pb.s0, pb.z0)
            ok, step = system.solve(rhs_x, pb.s0, rhs_z, 0.4, -0.2, x, pb.s0, pb.z0, 1.3, 0.7, True)
        assert ok
        out.append(step)
    for a, bb in zip(out[0], out[1]):
        np.testing.assert_allclose(a, bb, rtol=1e-12, atol=1e-12 * max(1.0, float(np.max(np.abs(a)))))


@pytest.mark.gpu
@pytest.mark.parametrize("affine", [True, False])
def test_device_kkt_solve_matches_host_algebra(affine):
    """kkt_solve! (kktsystem.jl:145-215) on the device against the same formulas in numpy with host cone
    operations and the oracle's K^{-1}: (dx, dz, ds, dtau, dkappa) to 1e-9."""
    import scipy.sparse as sp
    from cuclarabel_amd import problems
    from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem
    from tests.oracle_bindings import make_oracle
    pb = problems.small_mixed(seed=41, psds=(2, 3, 6), socs=(3, 4, 6, 15))   # zero, nonnegative, dense and sparse SOCs, PSD
    ks = HipKKTSolver(pb.P, pb.A, pb.cones)
    system = HipKKTSystem(ks)
    system.init(pb.q, pb.b)
    rng = np.random.default_rng(17)
    s, z = pb.s0, pb.z0
    x = rng.standard_normal(pb.n)
    tau, kappa = 1.3, 0.7
    assert system.update(s, z)
    rhs_x, rhs_z = rng.standard_normal(pb.n), rng.standard_normal(pb.m)
    rhs_s = s.copy() if affine else rng.standard_normal(pb.m)
    rhs_tau, rhs_kappa = 0.4, -0.2
    ok, (dx, dz, ds, dtau, dkappa) = system.solve(rhs_x, rhs_s, rhs_z, rhs_tau, rhs_kappa, x, s, z, tau, kappa, affine)
    assert ok
    # ---- the same on the host
    o = make_oracle(pb, perm=ks.perm())
    assert o.update_scaling(s, z) and o.kktsolver_update()
    cones = ipm._make_cones(pb.cones)
    for c in cones:
        assert c.update_scaling(s[c.rng].copy(), z[c.rng].copy())
    # a PSD cone's scaled space is fixed only up to the signs of the singular vectors: the device's and LAPACK's R
    # agree in R R' and lambda; the right-hand side rhs_s lives in the scaled space, so use the device's R on the host
    host_psd = [(c.R, c.lam) for c in cones if isinstance(c, ipm._PSD)]
    ipm.adopt_device_scaling(cones, ks.scaling()[1])
    for (Rh, lh), c in zip(host_psd, [c for c in cones if isinstance(c, ipm._PSD)]):
        np.testing.assert_allclose(c.lam, lh, rtol=1e-11)
        np.testing.assert_allclose(c.R @ c.R.T, Rh @ Rh.T, rtol=0, atol=1e-11 * np.abs(Rh @ Rh.T).max())
        assert np.abs(c.R @ c.Rinv - np.eye(c.k)).max() < 1e-11

    def each(fn, *vecs):
        out = np.empty(pb.m)
        for c in cones:
            out[c.rng] = fn(c, *[v[c.rng] for v in vecs])
        return out

    def ksolve(rx, rz):
        o.kktsolver_setrhs(rx, rz)
        ok_, xo, zo = o.kktsolver_solve()
        assert ok_
        return xo, zo

    Pt = sp.triu(sp.csc_matrix(pb.P), format="csc")
    Pfull = (Pt + sp.triu(Pt, 1).T).tocsr()
    x2, z2 = ksolve(-pb.q, pb.b)
    const = s.copy() if affine else each(lambda c, d, zz: c.ds_from_dz_offset(d, zz), rhs_s, z)
    x1, z1 = ksolve(rhs_x, const - rhs_z)
    xi = x / tau
    tnum = rhs_tau - rhs_kappa / tau + pb.q @ x1 + pb.b @ z1 + 2 * (xi @ (Pfull @ x1))
    xm = xi - x2
    tden = kappa / tau - pb.q @ x2 - pb.b @ z2 + xm @ (Pfull @ xm) - x2 @ (Pfull @ x2)
    dtau_h = tnum / tden
    dx_h, dz_h = x1 + dtau_h * x2, z1 + dtau_h * z2
    ds_h = -(each(lambda c, v: c.mul_Hs(v), dz_h) + const)
    dkappa_h = -(rhs_kappa + kappa * dtau_h) / tau
    assert abs(dtau - dtau_h) <= 1e-9 * max(1.0, abs(dtau_h))
    assert abs(dkappa - dkappa_h) <= 1e-9 * max(1.0, abs(dkappa_h))
    for a, bb in ((dx, dx_h), (dz, dz_h), (ds, ds_h)):
        assert np.abs(a - bb).max() <= 1e-9 * max(1.0, np.abs(bb).max())


# ---- level C through the reference's own two calls (lazy constant-RHS solve) ---------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ALL))
def test_reference_known_answers_with_lazy_constant_solve(name):
    """kkt_update! and kkt_solve!(:affine) as the TWO SEPARATE calls solver.jl:278-295 makes, with the handle in lazy
    mode (hipkkt_kkt_system_set_lazy): kkt_update! leaves the constant-RHS solve to the affine kkt_solve!, which sends
    both right-hand sides through the sweeps as one 2-column solve.  Same known answers, and the SAME iterates, iteration
    and refinement-round counts as the one-call form (batch_affine=True), with no fallback taken."""
    P, q, A, b, cones, exp = ALL[name]()
    lazy = ipm.HipSystemBackend(P, A, cones, lazy=True)
    res = ipm.solve(P, q, A, b, cones, lazy)
    _check(res, exp)
    ref = ipm.solve(P, q, A, b, cones, ipm.HipSystemBackend(P, A, cones, batch_affine=True))
    assert res.iterations == ref.iterations and res.kkt_ir_rounds == ref.kkt_ir_rounds
    np.testing.assert_allclose(res.x, ref.x, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(res.z, ref.z, rtol=1e-13, atol=1e-13)
    assert lazy.ks.fallbacks == (0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ALL))
def test_reference_known_answers_from_the_callers_cone_objects(name):
    """The route of the Julia glue's HipKKTSystem: kkt_update!(kktsystem, data, cones) hands over what the CALLER's
    cone objects hold (hipkkt_kkt_system_update_cones: get_Hs! blocks, sparse-SOC u / v / eta^2, w, eta, lambda, R,
    Rinv), lazy mode, host-resident iterates through the *_host entry points."""
    P, q, A, b, cones, exp = ALL[name]()
    res = ipm.solve(P, q, A, b, cones, ipm.HipSystemBackend(P, A, cones, lazy=True, host_cones=True))
    _check(res, exp)
    ref = ipm.solve(P, q, A, b, cones, OracleBackend(P, A, cones))
    assert res.iterations == ref.iterations
    if exp["status"] == "SOLVED":
        np.testing.assert_allclose(res.x, ref.x, atol=1e-7)


@pytest.mark.gpu
def test_lazy_mode_on_sparse_second_order_cones_walks_with_the_eager_calls():
    """cfg2-shaped problem (every second-order cone sparse-expanded): lazy two-call mode, from the device's own scaling
    and from the caller's cone objects, against the eager level-C calls -- same path for the first ten iterations."""
    from cuclarabel_amd import problems
    pb = problems.config2(n=1000)
    runs = [ipm.solve(pb.P, pb.q, pb.A, pb.b, pb.cones, ipm.HipSystemBackend(pb.P, pb.A, pb.cones, **kw))
            for kw in (dict(), dict(lazy=True), dict(lazy=True, host_cones=True), dict(lazy=True, staging="torch"))]
    k = min(min(len(r.history) for r in runs), 10)
    assert k == 10
    for r in runs[1:]:
        for a, b in zip(r.history[:k], runs[0].history[:k]):
            assert abs(a["pcost"] - b["pcost"]) < 1e-6 * max(1, abs(b["pcost"]))
            assert abs(a["mu"] - b["mu"]) < 1e-6 * max(1e-12, abs(b["mu"]))


@pytest.mark.gpu
def test_lazy_constant_solve_failure_and_flush():
    """(i) A numeric failure in the constant-RHS solve must end the iteration as it does in the reference
    (kktsystem.jl:62-92 returns false from kkt_update!, solver.jl:279-295 ANDs the two results): in lazy mode kkt_update!
    has returned true by then, so the failure has to come out of the affine kkt_solve!.  (ii) Any other consumer of
    (x2, z2) -- a :combined solve arriving first -- makes the pending solve run by itself: same step as the eager calls."""
    from cuclarabel_amd import problems
    from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem
    pb = problems.small_mixed(seed=43, psds=(2, 3, 6), socs=(3, 4, 6, 15))
    rng = np.random.default_rng(23)
    x = rng.standard_normal(pb.n)
    rhs_x, rhs_s, rhs_z = rng.standard_normal(pb.n), rng.standard_normal(pb.m), rng.standard_normal(pb.m)
    args = (0.4, -0.2, x, pb.s0, pb.z0, 1.3, 0.7)
    # (i) q with a non-finite entry: the constant right-hand side's residual is not finite
    qbad = pb.q.copy()
    qbad[0] = np.inf
    for lazy in (False, True):
        system = HipKKTSystem(HipKKTSolver(pb.P, pb.A, pb.cones))
        system.init(qbad, pb.b)
        system.set_lazy(lazy)
        ok_update = system.update(pb.s0, pb.z0)
        assert ok_update == lazy                       # eager: kkt_update! reports it; lazy: not yet
        if lazy:
            ok, step = system.solve(rhs_x, pb.s0, rhs_z, *args, True)
            assert not ok and step is None             # ... the affine kkt_solve! does: `a && b` is false either way
    # and through the driver: both modes stop with the same status on such data
    st = [ipm.solve(pb.P, qbad, pb.A, pb.b, pb.cones, ipm.HipSystemBackend(pb.P, pb.A, pb.cones, lazy=lz)).status for lz in (False, True)]
    assert st[0] == st[1] == ipm.NUMERICAL_ERROR, st
    # (ii) combined step first
    out = []
    for lazy in (False, True):
        system = HipKKTSystem(HipKKTSolver(pb.P, pb.A, pb.cones))
        system.init(pb.q, pb.b)
        system.set_lazy(lazy)
        assert system.update(pb.s0, pb.z0)
        ok, step = system.solve(rhs_x, rhs_s, rhs_z, *args, False)
        assert ok
        out.append(step)
    for a, bb in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, bb)


@pytest.mark.gpu
@pytest.mark.parametrize("maker,exact", [("problems.config2(n=3000)", True),
                                         ("problems.small_mixed(seed=43, psds=(2, 3, 6), socs=(3, 4, 6, 15))", False)])
def test_update_scaling_forms_the_kkt_values_on_the_device(maker, exact):
    """hipkkt_kkt_system_update_scaling takes the NT scaling alone (w, eta, lambda, R, Rinv) and forms get_Hs!'s blocks and
    the sparse second-order cones' u, v, eta^2 on the device -- half the bytes of hipkkt_kkt_system_update_cones over PCIe.
    Fed with the device's own scaling it must reproduce the K values of hipkkt_kkt_system_update BIT FOR BIT on
    elementwise and second-order cones (same arithmetic in the same order; PSD blocks: R R' is re-formed, round-off), in
    eager and in lazy mode, and the combined kkt_solve! that re-uses the affine one's uploaded variables
    (hipkkt_kkt_system_solve_host with NULL variables) must return what the call with variables returns."""
    from cuclarabel_amd import problems, _lib
    from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem
    pb = eval(maker)
    rng = np.random.default_rng(41)
    x = rng.standard_normal(pb.n)
    rhs_x, rhs_s, rhs_z = rng.standard_normal(pb.n), rng.standard_normal(pb.m), rng.standard_normal(pb.m)
    args = (0.4, -0.2, x, pb.s0, pb.z0, 1.3, 0.7)
    ka = HipKKTSolver(pb.P, pb.A, pb.cones)
    sa = HipKKTSystem(ka)
    sa.init(pb.q, pb.b)
    assert sa.update(pb.s0, pb.z0)
    Ka = ka.KKT().data.copy()
    ok, aff_a = sa.solve(rhs_x, pb.s0, rhs_z, *args, True)
    assert ok
    ok, com_a = sa.solve(rhs_x, rhs_s, rhs_z, *args, False)
    assert ok
    lam, psd = ka.scaling()
    w, eta = ka.scaling_w()
    R = np.concatenate([t[0].ravel(order="F") for t in psd]) if psd else np.zeros(0)
    Ri = np.concatenate([t[1].ravel(order="F") for t in psd]) if psd else np.zeros(0)
    for lazy in (False, True):
        kb = HipKKTSolver(pb.P, pb.A, pb.cones)
        sb = HipKKTSystem(kb)
        sb.init(pb.q, pb.b)
        sb.set_lazy(lazy)
        keep = [np.ascontiguousarray(v) for v in (w, eta, lam, R, Ri)]
        registered = [a for a in keep if a.size and _lib.host_register(a)]         # (page-locked as the glue keeps them)
        assert sb.update_scaling(*keep)
        ok, aff_b = sb.solve(rhs_x, pb.s0, rhs_z, *args, True)
        assert ok
        Kb = kb.KKT().data
        if exact:
            np.testing.assert_array_equal(Kb, Ka)
        else:
            np.testing.assert_allclose(Kb, Ka, rtol=1e-12, atol=1e-14 * np.abs(Ka).max())
        ok, com_b = sb.solve(rhs_x, rhs_s, rhs_z, *args, False, reuse_variables=True)
        assert ok
        for a in registered:
            assert _lib.host_unregister(a)
        # (the steps agree to refinement accuracy, not bit for bit: with a second handle alive in the process this handle's
        #  factorisation may be admitted to a different mode -- overlapped or level by level -- whose sums run in another order)
        for got, want in ((aff_b, aff_a), (com_b, com_a)):
            for g, v in zip(got, want):
                g, v = np.asarray(g, dtype=float), np.asarray(v, dtype=float)
                np.testing.assert_allclose(g, v, rtol=1e-9, atol=1e-10 * max(np.abs(v).max(), 1.0))


@pytest.mark.gpu
def test_lazy_update_then_startup_calls_read_the_update_status():
    """The reference's default-start sequence is kkt_update!, then kkt_solve_initial_point! (solver.jl:389-393).  In lazy
    mode kkt_update! only enqueues; its status record must be read by whatever comes next -- not only by kkt_solve! --
    or (a) a give-up of the overlap mode leaves a void factor under the initial point and (b) a stale failure word is
    folded into the first iteration's record (round-3 advisor).  (a): both bounded waits forced to expire, in a child
    process; the initial point and the first affine step must match the eager sequence of an undisturbed process.
    (b): an update on a non-interior point (reported as a failure), then a good one, then the initial point and a step."""
    import os
    import subprocess
    import sys
    from cuclarabel_amd import problems
    from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys
import numpy as np
sys.path.insert(0, {root!r})
from cuclarabel_amd import problems
from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem
pb = problems.config2(n=6000)
rng = np.random.default_rng(31)
x = rng.standard_normal(pb.n)
rhs_x, rhs_z = rng.standard_normal(pb.n), rng.standard_normal(pb.m)
args = (0.4, -0.2, x, pb.s0, pb.z0, 1.3, 0.7)
out = []
for lazy in (False, True):
    ks = HipKKTSolver(pb.P, pb.A, pb.cones)
    system = HipKKTSystem(ks)
    system.init(pb.q, pb.b)
    system.set_lazy(lazy)
    assert system.update(pb.s0, pb.z0)
    ok, x0, s0, z0 = system.solve_initial_point()
    assert ok
    ok, step = system.solve(rhs_x, pb.s0, rhs_z, *args, True)
    assert ok
    out.append((x0, s0, z0) + tuple(step))
    print("FALLBACKS", lazy, ks.fallbacks)
for a, b in zip(out[0], out[1]):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert np.abs(a - b).max() <= 1e-9 * max(np.abs(a).max(), 1e-300), np.abs(a - b).max()
print("STARTUP OK")
"""
    r = subprocess.run([sys.executable, "-c", script.format(root=root)],
                       env=dict(os.environ, HIPKKT_OV_TEST_LIMIT="0", HIPKKT_TOP_TEST_LIMIT="0"), cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "STARTUP OK" in r.stdout
    assert "gave up" in r.stderr, r.stderr              # the forced give-ups really happened
    # (b) a failed enqueued-only update must not haunt the calls after the next, good one
    pb = problems.small_mixed(seed=43, psds=(2, 3, 6), socs=(3, 4, 6, 15))
    rng = np.random.default_rng(37)
    x = rng.standard_normal(pb.n)
    rhs_x, rhs_z = rng.standard_normal(pb.n), rng.standard_normal(pb.m)
    args = (0.4, -0.2, x, pb.s0, pb.z0, 1.3, 0.7)
    ref = HipKKTSystem(HipKKTSolver(pb.P, pb.A, pb.cones))
    ref.init(pb.q, pb.b)
    assert ref.update(pb.s0, pb.z0)
    ok, x0, s0, z0 = ref.solve_initial_point()
    assert ok
    system = HipKKTSystem(HipKKTSolver(pb.P, pb.A, pb.cones))
    system.init(pb.q, pb.b)
    system.set_lazy(True)
    assert system.update(-pb.s0, pb.z0)                  # not interior: the failure sits in the unread record
    assert system.update(pb.s0, pb.z0)
    ok, x1, s1, z1 = system.solve_initial_point()
    assert ok
    for a, b in ((x0, x1), (s0, s1), (z0, z1)):
        assert np.abs(a - b).max() <= 1e-9 * max(np.abs(a).max(), 1e-300)
    ok, step = system.solve(rhs_x, pb.s0, rhs_z, *args, True)
    assert ok and step is not None


@pytest.mark.gpu
@pytest.mark.parametrize("affine", [True, False])
def test_system_update_cones_equals_the_device_scaling(affine):
    """hipkkt_kkt_system_update_cones (scaling handed over by the caller) against hipkkt_kkt_system_update (scaling
    computed on the device from (s, z)): fed with the device's own w, eta, lambda, R, Rinv and Hs blocks (and the
    oracle's sparse-SOC u, v, eta^2) the step must agree to round-off, on every cone kind."""
    from cuclarabel_amd import problems
    from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem
    from tests.oracle_bindings import make_oracle
    pb = problems.small_mixed(seed=43, psds=(2, 3, 6), socs=(3, 4, 6, 15))
    rng = np.random.default_rng(29)
    x = rng.standard_normal(pb.n)
    rhs_x, rhs_z = rng.standard_normal(pb.n), rng.standard_normal(pb.m)
    rhs_s = pb.s0.copy() if affine else rng.standard_normal(pb.m)
    args = (0.4, -0.2, x, pb.s0, pb.z0, 1.3, 0.7)
    ka = HipKKTSolver(pb.P, pb.A, pb.cones)
    sa = HipKKTSystem(ka)
    sa.init(pb.q, pb.b)
    assert sa.update(pb.s0, pb.z0)
    ok, step_a = sa.solve(rhs_x, rhs_s, rhs_z, *args, affine)
    assert ok
    lam, psd = ka.scaling()
    w, eta = ka.scaling_w()
    o = make_oracle(pb, perm=ka.perm())
    assert o.update_scaling(pb.s0, pb.z0)
    u, v, e2, _ = o.soc_sparse()
    R = np.concatenate([t[0].ravel(order="F") for t in psd]) if psd else np.zeros(0)
    Ri = np.concatenate([t[1].ravel(order="F") for t in psd]) if psd else np.zeros(0)
    for lazy in (False, True):
        kb = HipKKTSolver(pb.P, pb.A, pb.cones)
        sb = HipKKTSystem(kb)
        sb.init(pb.q, pb.b)
        sb.set_lazy(lazy)
        assert sb.update_cones(ka.get_Hs(), u, v, e2, w, eta, lam, R, Ri)
        ok, step_b = sb.solve(rhs_x, rhs_s, rhs_z, *args, affine)
        assert ok
        for a, bb in zip(step_a, step_b):
            np.testing.assert_allclose(a, bb, rtol=1e-9, atol=1e-9 * max(1.0, float(np.max(np.abs(a)))))


@pytest.mark.gpu
def test_speculative_refinement_depth_comes_down_again():
    """A solve enqueues as many refinement rounds ahead of its status read-back as the previous solves took; the
    reference's accept / stop rule (kktsolver_directldl.jl:397-449) runs on the device, so rounds beyond the ones the rule
    takes change nothing but cost a sweep pair each.  One solve that takes two rounds must not leave every later solve
    of the run with two enqueued: the depth comes down after eight status records in a row whose solves could have
    done with one fewer -- with bit-identical steps either way."""
    from cuclarabel_amd import problems
    from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem
    pb = problems.config2(n=4000)
    rng = np.random.default_rng(53)
    x = rng.standard_normal(pb.n)
    rhs_x, rhs_z, rhs_s = rng.standard_normal(pb.n), rng.standard_normal(pb.m), rng.standard_normal(pb.m)
    args = (0.4, -0.2, x, pb.s0, pb.z0, 1.3, 0.7)
    ks = HipKKTSolver(pb.P, pb.A, pb.cones)
    system = HipKKTSystem(ks)
    system.init(pb.q, pb.b)
    system.set_lazy(True)

    def step():
        assert system.update(pb.s0, pb.z0)
        ok, aff = system.solve(rhs_x, pb.s0, rhs_z, *args, True)
        assert ok
        ok, comb = system.solve(rhs_x, rhs_s, rhs_z, *args, False)
        assert ok
        return [np.asarray(v, dtype=float) for v in tuple(aff) + tuple(comb)]

    base = step()
    assert ks.last_ir_iterations <= 1
    depth0 = ks.speculative_rounds()
    assert depth0 <= 1
    assert ks.speculative_rounds(2) == 2
    deep = step()
    for a, b in zip(base, deep):
        assert np.array_equal(a, b)                      # the extra round is computed and not accepted
    for _ in range(5):
        step()                                           # two status records per step
    assert ks.speculative_rounds() == 1
    again = step()
    for a, b in zip(base, again):
        assert np.array_equal(a, b)
    assert ks.fallbacks == (0, 0)


@pytest.mark.gpu
def test_status_record_that_does_not_reach_the_host_is_noticed():
    """In lazy mode a kkt_solve!'s status record and (dtau, dkappa) are written into page-locked host memory by the
    call's last kernel, with a sequence number behind them.  If the number the host reads is not the one it passed --
    forced here for the first record of the handle -- the call is repeated with the synchronous sequence (its update
    included) and the handle copies its records from then on: same steps as an undisturbed process, no silent
    "all zeros = no error" record."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys
import numpy as np
sys.path.insert(0, {root!r})
from cuclarabel_amd import problems
from cuclarabel_amd.kktsolver import HipKKTSolver, HipKKTSystem
pb = problems.config2(n=3000)
rng = np.random.default_rng(61)
x = rng.standard_normal(pb.n)
rhs_x, rhs_z, rhs_s = rng.standard_normal(pb.n), rng.standard_normal(pb.m), rng.standard_normal(pb.m)
args = (0.4, -0.2, x, pb.s0, pb.z0, 1.3, 0.7)
ks = HipKKTSolver(pb.P, pb.A, pb.cones)
system = HipKKTSystem(ks)
system.init(pb.q, pb.b)
system.set_lazy(True)
out = []
for it in range(3):
    assert system.update(pb.s0, pb.z0)
    ok, aff = system.solve(rhs_x, pb.s0, rhs_z, *args, True)
    assert ok
    ok, comb = system.solve(rhs_x, rhs_s, rhs_z, *args, False)
    assert ok
    out.append([np.asarray(v, dtype=float) for v in tuple(aff) + tuple(comb)])
for o in out[1:]:
    for a, b in zip(out[0], o):
        assert np.array_equal(a, b)
np.save(sys.argv[1], np.concatenate([v.ravel() for v in out[0]]))
print("FALLBACKS", ks.fallbacks)
"""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        res = {}
        for tag, env in (("plain", {}), ("forced", {"HIPKKT_TEST_PUBLISH_FAIL": "1", "HIPKKT_VERBOSE": "1"})):
            path = os.path.join(tmp, tag + ".npy")
            r = subprocess.run([sys.executable, "-c", script.format(root=root), path], env=dict(os.environ, **env), cwd=root,
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stdout + r.stderr
            assert "FALLBACKS (0, 0)" in r.stdout
            assert ("not published" in r.stderr) == (tag == "forced"), r.stderr
            res[tag] = np.load(path)
        assert np.array_equal(res["plain"], res["forced"])
