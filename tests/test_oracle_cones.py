"""The reference's cone-algebra unit tests restated against the oracle: they pin the inputs of
the factorisation.
  test/UnitTests/test_coneops_secondordercone.jl:31-91
  test/UnitTests/test_coneops_psdtrianglecone.jl:112-251"""
import numpy as np
import pytest
import scipy.sparse as sp

from cuclarabel_amd.cones import NonnegativeConeT, SecondOrderConeT, PSDTriangleConeT, ZeroConeT
from cuclarabel_amd.problems import mat_to_svec, svec_to_mat
from tests.oracle_bindings import OracleKKT
from tests.ref_kkt_numpy import soc_nt, soc_W2, psd_W2


def _oracle(cones):
    m = sum(c.numel for c in cones)
    return OracleKKT(sp.identity(2, format="csc"), sp.csc_matrix(np.ones((m, 2))), cones)


def _soc_point(rng, n):
    t = rng.standard_normal(n - 1)
    return np.r_[np.linalg.norm(t) + rng.uniform(0.1, 2.0), t]


@pytest.mark.parametrize("n", [5, 12, 100])
def test_soc_sparse_expansion_identity(n):
    # eta^2 (D + u u' - v v') == eta^2 (2 w w' - J)   to 1e-14-ish (secondordercone.jl:60-66)
    rng = np.random.default_rng(242713)
    o = _oracle([SecondOrderConeT(n)])
    s, z = _soc_point(rng, n), _soc_point(rng, n)
    assert o.update_scaling(s, z)
    Hs = o.get_Hs()
    u, v, e2, d = o.soc_sparse()
    D = np.diag(Hs) / e2[0]
    assert D[0, 0] == pytest.approx(d[0], rel=1e-15) and np.allclose(np.diag(D)[1:], 1.0)
    W2 = e2[0] * (D + np.outer(u, u) - np.outer(v, v))
    eta, w = soc_nt(s, z)
    J = np.diag(np.r_[1.0, -np.ones(n - 1)])
    ref = eta ** 2 * (2 * np.outer(w, w) - J)
    np.testing.assert_allclose(W2, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    assert v[0] == 0.0                                  # coneops_socone.jl:143,148
    # W'W s... the NT point maps z to s: Hs z = s
    np.testing.assert_allclose(o.mul_Hs(z), s, rtol=1e-11)


@pytest.mark.parametrize("n", [2, 3, 4])
def test_soc_dense_block_for_small_cones(n):
    rng = np.random.default_rng(5 + n)
    o = _oracle([SecondOrderConeT(n)])
    s, z = _soc_point(rng, n), _soc_point(rng, n)
    assert o.update_scaling(s, z)
    Hs = o.get_Hs()
    assert len(Hs) == n * (n + 1) // 2                   # packed triu (compositecone_type.jl:136-138)
    M = np.zeros((n, n)); k = 0
    for col in range(n):
        for row in range(col + 1):
            M[row, col] = M[col, row] = Hs[k]; k += 1
    np.testing.assert_allclose(M, soc_W2(s, z), rtol=1e-12, atol=1e-13)
    assert o.p == 0


def test_soc_noninterior_point_is_reported():
    o = _oracle([SecondOrderConeT(6)])
    s = np.r_[1.0, 2.0, 0, 0, 0, 0]                      # outside the cone
    z = np.r_[3.0, 1.0, 0, 0, 0, 0]
    assert o.update_scaling(s, z) is False               # coneops_socone.jl:88


@pytest.mark.parametrize("k", [1, 3, 5, 8])
def test_psd_Hs_block_is_WtW(k):
    # get_Hs! packed block times v == W'(W v) to 1e-8 (psdtrianglecone.jl:213-251)
    rng = np.random.default_rng(242713)
    o = _oracle([PSDTriangleConeT(k)])
    G1, G2 = rng.standard_normal((k, k)), rng.standard_normal((k, k))
    S, Z = G1 @ G1.T + np.eye(k), G2 @ G2.T + np.eye(k)
    s, z = mat_to_svec(S), mat_to_svec(Z)
    assert o.update_scaling(s, z)
    t = k * (k + 1) // 2
    Hs = o.get_Hs()
    assert len(Hs) == t * (t + 1) // 2
    H = np.zeros((t, t)); q = 0
    for col in range(t):
        for row in range(col + 1):
            H[row, col] = H[col, row] = Hs[q]; q += 1
    np.testing.assert_allclose(H, psd_W2(s, z, k), rtol=1e-9, atol=1e-10)
    v = rng.standard_normal(t)
    np.testing.assert_allclose(H @ v, o.mul_Hs(v), rtol=1e-8, atol=1e-9)
    # NT identity: W'W z = s  (psdtrianglecone.jl:180-211)
    np.testing.assert_allclose(o.mul_Hs(z), s, rtol=1e-8, atol=1e-9)


def test_psd_not_positive_definite_is_reported():
    o = _oracle([PSDTriangleConeT(3)])
    S = np.diag([1.0, -1.0, 2.0])
    assert o.update_scaling(mat_to_svec(S), mat_to_svec(np.eye(3))) is False     # psd :101-103


def test_nn_and_zero_blocks():
    o = _oracle([ZeroConeT(2), NonnegativeConeT(3)])
    s = np.r_[0.0, 0.0, 1.0, 4.0, 9.0]; z = np.r_[0.0, 0.0, 4.0, 1.0, 1.0]
    assert o.update_scaling(s, z)
    Hs = o.get_Hs()
    np.testing.assert_allclose(Hs, [0, 0, 0.25, 4.0, 9.0], rtol=1e-15)           # w^2 = s/z; zero cone 0
    np.testing.assert_allclose(o.mul_Hs(np.ones(5)), [0, 0, 0.25, 4.0, 9.0], rtol=1e-15)


def test_driver_cone_objects_hand_over_what_the_oracle_computes():
    """cuclarabel_amd.ipm.host_cone_data (what the Julia glue reads from the reference's cone objects for kkt_update!:
    get_Hs! blocks, sparse second-order (u, v, eta^2), w, eta, lambda, R, Rinv) against the oracle's cones on a problem
    with every cone kind -- the data hipkkt_kkt_system_update_cones takes."""
    from cuclarabel_amd import ipm, problems
    from tests.oracle_bindings import make_oracle
    pb = problems.small_mixed(seed=43, psds=(2, 3, 6), socs=(3, 4, 6, 15))
    o = make_oracle(pb)
    assert o.update_scaling(pb.s0, pb.z0)
    cones = ipm._make_cones(pb.cones)
    for c in cones:
        assert c.update_scaling(pb.s0[c.rng].copy(), pb.z0[c.rng].copy())
    Hs, u, v, e2, w, eta, lam, R, Ri = ipm.host_cone_data(cones)
    Hs_o = o.get_Hs()
    np.testing.assert_allclose(Hs, Hs_o, rtol=1e-11, atol=1e-13 * np.abs(Hs_o).max())
    uo, vo, e2o, _ = o.soc_sparse()
    np.testing.assert_allclose(u, uo, rtol=1e-12)
    np.testing.assert_allclose(v, vo, rtol=1e-12)
    np.testing.assert_allclose(e2, e2o, rtol=1e-12)
    lam_o = o.cone_lambda()
    for c in cones:
        if isinstance(c, (ipm._NN, ipm._SOC)):
            np.testing.assert_allclose(lam[c.rng], lam_o[c.rng], rtol=1e-11, atol=1e-13)
        if isinstance(c, ipm._PSD):
            np.testing.assert_allclose(lam[c.off:c.off + c.k], lam_o[c.off:c.off + c.k], rtol=1e-11)
    # y = W'W x through the driver's cones equals the oracle's mul_Hs
    x = np.random.default_rng(3).standard_normal(pb.m)
    y = np.concatenate([c.mul_Hs(x[c.rng]) for c in cones])
    np.testing.assert_allclose(y, o.mul_Hs(x), rtol=0, atol=1e-11 * np.abs(y).max())
