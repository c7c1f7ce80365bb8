"""GPU parity on sparsity structures OUTSIDE the five BASELINE configurations (cuclarabel_amd.problems.ZOO).

The schedule's thresholds (which levels the persistent sweep kernel takes, when the factorisation's overlap mode is
admitted, panel widths, row slices, chained launches) were fitted on cfg1-cfg5, whose KKT graphs are banded, block-dense
or block-banded.  These cases give the same code mesh separators that grow with the subgraph (2-D / 3-D grids), hub
nodes, a forest of unequal trees, a tree with no top at all, one dense front, an LP with P = 0 and a third of the
pivots at -eps.  Same bar as tests/test_gpu_parity.py: HIP path through the C ABI against the CPU oracle on the same K
and b -- K values, solutions to 1e-9, the oracle's refinement-round counts -- and no fall-back taken.
"""
import numpy as np
import pytest

from cuclarabel_amd import problems

pytestmark = pytest.mark.gpu


def _hip():
    from cuclarabel_amd import _lib
    from cuclarabel_amd.kktsolver import HipKKTSolver
    assert _lib.lib().hipkkt_available() == 1, "no gfx950 device visible"
    return HipKKTSolver


def _pair(pb):
    from tests.oracle_bindings import make_oracle
    ks = _hip()(pb.P, pb.A, pb.cones)
    return ks, make_oracle(pb, perm=ks.perm())


def _rel(x, z, xo, zo):
    return max(np.abs(x - xo).max(), np.abs(z - zo).max()) / max(np.abs(xo).max(), np.abs(zo).max())


@pytest.mark.parametrize("name,maker", problems.ZOO, ids=[c[0] for c in problems.ZOO])
def test_structure_zoo_factor_and_solves_match_oracle(name, maker):
    pb = maker()
    ks, o = _pair(pb)
    rng = np.random.default_rng(77)
    s, z = pb.s0, pb.z0
    for rep in range(2):                                   # a factorisation, then a refactorisation in place
        assert ks.kktsolver_update_from_sz(s, z)
        assert o.update_scaling(s, z) and o.kktsolver_update()
        np.testing.assert_allclose(ks.get_Hs(), o.get_Hs(), rtol=1e-13, atol=0)
        # the sparse SOC columns u, v come from reductions over up to 4000 terms, summed in a different order on the
        # device: entries that nearly cancel differ by a few ulps of the LARGEST term, hence the absolute part
        Ko = o.K().data
        np.testing.assert_allclose(ks.KKT().data, Ko, rtol=1e-11, atol=1e-15 * np.abs(Ko).max())
        assert ks.diagonal_regularizer == pytest.approx(o.last_regularizer, rel=1e-15)
        for _ in range(2):
            rx, rz = rng.standard_normal(pb.n), rng.standard_normal(pb.m)
            ks.kktsolver_setrhs(rx, rz)
            o.kktsolver_setrhs(rx, rz)
            x, zz = np.zeros(pb.n), np.zeros(pb.m)
            assert ks.kktsolver_solve(x, zz)
            ok, xo, zo = o.kktsolver_solve()
            assert ok
            assert _rel(x, zz, xo, zo) < 1e-9, (rep, _rel(x, zz, xo, zo))
            # a borderline residual may fall on either side of the stop rule (different summation orders): at most
            # one round apart, as in the full-size configuration tests
            assert abs(ks.last_ir_iterations - o.last_ir_iters) <= 1, (ks.last_ir_iterations, o.last_ir_iters)
        # the next iterate: another strictly interior point (every cone's scaling changes)
        s = problems.interior_point(pb.cones, rng)
        z = problems.interior_point(pb.cones, rng)
    assert ks.fallbacks == (0, 0), ks.fallbacks


@pytest.mark.parametrize("name", ["grid3d_24", "powerlaw_20k", "forest_400", "diag_100k", "lp_transport"])
@pytest.mark.parametrize("k", [2, 5, 40])
def test_structure_zoo_many_columns(name, k):
    """The shared-sweep (k <= 8) and many-column (k > 8) solve paths on the same structures: every column against its
    own single solve and, sampled, against the oracle."""
    pb = dict(problems.ZOO)[name]()
    ks, o = _pair(pb)
    assert ks.kktsolver_update_from_sz(pb.s0, pb.z0)
    assert o.update_scaling(pb.s0, pb.z0) and o.kktsolver_update()
    rng = np.random.default_rng(900 + k)
    RX, RZ = rng.standard_normal((pb.n, k)), rng.standard_normal((pb.m, k))
    RX[:, 1] = 0.0; RZ[:, 1] = 0.0
    ok, LX, LZ, ir = ks.kktsolver_solve_multi(RX, RZ)
    assert ok and not LX[:, 1].any() and not LZ[:, 1].any() and ir[1] == 0
    for j in sorted({0, k // 2, k - 1} - {1}):
        ks.kktsolver_setrhs(RX[:, j], RZ[:, j])
        x, z = np.zeros(pb.n), np.zeros(pb.m)
        assert ks.kktsolver_solve(x, z)
        assert _rel(LX[:, j], LZ[:, j], x, z) < 1e-11, j
        o.kktsolver_setrhs(RX[:, j], RZ[:, j])
        oko, xo, zo = o.kktsolver_solve()
        assert oko and _rel(LX[:, j], LZ[:, j], xo, zo) < 1e-9, j
    assert ks.fallbacks == (0, 0), ks.fallbacks
