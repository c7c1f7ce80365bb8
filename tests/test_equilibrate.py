"""Ruiz equilibration: the numpy restatement against the reference's own unit tests
(test/UnitTests/test_equilibration_bounds.jl) on CPU, and the device routine against the restatement on GPU."""
import numpy as np
import pytest
import scipy.sparse as sp

from cuclarabel_amd import problems
from cuclarabel_amd.cones import NonnegativeConeT
from tests.ref_equilibrate_numpy import equilibrate_ref


def reference_test_data():
    """test_equilibration_bounds.jl:6-20"""
    P = sp.csc_matrix(np.array([[4.0, 1.0], [1.0, 2.0]]))
    c = np.array([1.0, 1.0])
    A = np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]])
    l, u = np.array([1.0, 0.0, 0.0]), np.array([1.0, 0.7, 0.7])
    A = sp.csc_matrix(np.vstack([-A, A]))
    b = np.concatenate([-l, u])
    return P, c, A, b, [NonnegativeConeT(3), NonnegativeConeT(3)]


def _bounds_ok(d, e, lo=1e-4, hi=1e4):
    return d.min() >= lo and e.min() >= lo and d.max() <= hi and e.max() <= hi


def _variants():
    P, c, A, b, cones = reference_test_data()
    Pl = P.tolil(); Pl[0, 0] = 1e-15                       # "equilibrate lower bound" (:28-43)
    Au = A.tolil(); Au[0, 0] = 1e15                        # "equilibrate upper bound" (:45-61)
    Az = A.copy(); Az.data[:] = 0.0                        # "equilibrate zero rows" (:63-76)
    return {"lower": (Pl.tocsc(), c, A, b, cones), "upper": (P, c, Au.tocsc(), b, cones), "zero_rows": (P, c, Az, b, cones)}


@pytest.mark.parametrize("name", ["lower", "upper", "zero_rows"])
def test_restatement_meets_the_reference_unit_tests(name):
    P, c, A, b, cones = _variants()[name]
    _, _, _, _, d, e, _ = equilibrate_ref(P, c, A, b, cones)
    assert _bounds_ok(d, e)
    if name == "zero_rows":
        assert np.all(e == 1.0)


def test_restatement_scales_consistently():
    pb = problems.small_mixed(seed=3)
    Ps, qs, As, bs, d, e, c = equilibrate_ref(pb.P, pb.q, pb.A, pb.b, pb.cones)
    Pt = sp.triu(sp.csc_matrix(pb.P))
    np.testing.assert_allclose(Ps.toarray(), c * (np.diag(d) @ Pt.toarray() @ np.diag(d)), rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(As.toarray(), np.diag(e) @ pb.A.toarray() @ np.diag(d), rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(qs, c * d * pb.q, rtol=1e-12)
    np.testing.assert_allclose(bs, e * pb.b, rtol=1e-12)
    # cones without elementwise scaling carry one factor each
    off = 0
    for cone in pb.cones:
        if not isinstance(cone, NonnegativeConeT) and type(cone).__name__ != "ZeroConeT":
            assert np.ptp(e[off:off + cone.numel]) <= 1e-15 * e[off]
        off += cone.numel


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lower", "upper", "zero_rows"])
def test_device_equilibration_on_the_reference_unit_tests(name):
    from cuclarabel_amd.equilibrate import equilibrate
    P, c, A, b, cones = _variants()[name]
    Ps, qs, As, bs, eq = equilibrate(P, c, A, b, cones)
    assert _bounds_ok(eq.d, eq.e)
    if name == "zero_rows":
        assert np.all(eq.e == 1.0)
    Pr, qr, Ar, br, d, e, cc = equilibrate_ref(P, c, A, b, cones)
    np.testing.assert_allclose(eq.d, d, rtol=1e-13)
    np.testing.assert_allclose(eq.e, e, rtol=1e-13)
    assert eq.c == pytest.approx(cc, rel=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("maker", [lambda: problems.small_mixed(seed=3), lambda: problems.config1(),
                                   lambda: problems.config2(n=3000), lambda: problems.config5(n=300, npsd=5, psd_dim=6, nsoc=3, soc_dim=9)],
                         ids=["mixed", "cfg1", "cfg2_n3000", "cfg5_small"])
def test_device_equilibration_matches_restatement(maker):
    from cuclarabel_amd.equilibrate import equilibrate, rescale_A, rescale_P
    pb = maker()
    Ps, qs, As, bs, eq = equilibrate(pb.P, pb.q, pb.A, pb.b, pb.cones)
    Pr, qr, Ar, br, d, e, c = equilibrate_ref(pb.P, pb.q, pb.A, pb.b, pb.cones)
    # the only order-dependent quantity is the mean column norm of P (a sum): a few ulp
    np.testing.assert_allclose(eq.d, d, rtol=1e-12)
    np.testing.assert_allclose(eq.e, e, rtol=1e-12)
    assert eq.c == pytest.approx(c, rel=1e-12)
    np.testing.assert_array_equal(Ps.indices, Pr.indices)
    np.testing.assert_allclose(Ps.data, Pr.data, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(As.data, Ar.data, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(qs, qr, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(bs, br, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(eq.dinv * eq.d, 1.0, rtol=1e-15)
    # update_P! / update_A! re-scaling of fresh data reproduces the scaled matrices
    np.testing.assert_allclose(rescale_P(pb.P, eq).data, Ps.data, rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(rescale_A(pb.A, eq).data, As.data, rtol=1e-12, atol=1e-300)
    # no scaling requested: identity
    P0, q0, A0, b0, eq0 = equilibrate(pb.P, pb.q, pb.A, pb.b, pb.cones, max_iter=0)
    assert np.all(eq0.d == 1.0) and np.all(eq0.e == 1.0) and eq0.c == 1.0
    np.testing.assert_array_equal(A0.data, sp.csc_matrix(pb.A).data)


@pytest.mark.gpu
def test_equilibrated_problem_solves_to_the_same_point():
    """Solving the scaled problem and un-scaling (x = D x~, solution_post_process!) gives the solution of
    the original one: the basic QP of the reference's tests."""
    from cuclarabel_amd import ipm
    from cuclarabel_amd.equilibrate import equilibrate
    from tests.golden import reference_fixtures as fx
    P, q, A, b, cones, exp = fx.basic_qp()
    Ps, qs, As, bs, eq = equilibrate(P, q, A, b, cones)
    res = ipm.solve(Ps, qs, As, bs, cones, ipm.HipBackend(Ps, As, cones))
    assert res.status == "SOLVED"
    np.testing.assert_allclose(res.x * eq.d, exp["x"], atol=1e-3)
