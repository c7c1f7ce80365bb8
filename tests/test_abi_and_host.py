"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/hipkkt.h
declares; the host-only symbolic analysis gives valid orderings whose structure statistics agree
with the oracle's own symbolic factorisation; host-side mirrors behave like the reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cuclarabel_amd import _lib, problems
from cuclarabel_amd.cones import (ZeroConeT, NonnegativeConeT, SecondOrderConeT, PSDTriangleConeT,
                                  cones_new_collapsed)
from tests.oracle_bindings import make_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "hipkkt.h")).read()
    declared = set(re.findall(r"\b(hipkkt_[a-z_A-Z0-9]+)\s*\(", header))
    declared -= {"hipkkt_ldl_s", "hipkkt_kkt_s"}
    assert len(declared) >= 35
    L = C.CDLL(_lib.SO_PATH)
    missing = [name for name in sorted(declared) if not hasattr(L, name)]
    assert not missing, f"libhipkkt.so lacks {missing}"
    # and the Python binding table covers the header exactly
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_library_loads_without_gpu_and_reports_unavailable_or_available():
    L = _lib.lib()
    assert L.hipkkt_version().startswith(b"hipkkt")
    assert L.hipkkt_available() in (0, 1)          # ldlsolver_is_available must not throw
    s = _lib.default_settings()
    assert s.static_regularization_constant == 1e-8
    assert s.static_regularization_proportional == np.finfo(float).eps ** 2      # settings.jl:119
    assert (s.dynamic_regularization_eps, s.dynamic_regularization_delta) == (1e-13, 2e-7)
    assert (s.iterative_refinement_reltol, s.iterative_refinement_abstol) == (1e-13, 1e-12)
    assert (s.iterative_refinement_max_iter, s.iterative_refinement_stop_ratio) == (10, 5.0)


def test_create_without_gpu_fails_loudly_not_silently():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cuclarabel_amd.kktsolver import HipKKTSolver
    pb = problems.config1(n=20, m=30, density=0.2)
    with pytest.raises(_lib.HipKKTError):
        HipKKTSolver(pb.P, pb.A, pb.cones)          # no CPU fallback


def test_malformed_csc_is_an_argument_error_not_a_crash():
    """A wrong index_base or a broken colptr / row index must come back as HIPKKT_ERR_ARG (-1) from the host-side
    assembly (before any device call, so this runs without a GPU), never as out-of-bounds host writes."""
    from cuclarabel_amd._lib import f64, i64, ptr
    L = _lib.lib()
    n, m = 3, 2
    Pp, Pi, Px = i64([0, 1, 2, 3]), i64([0, 1, 2]), f64([1.0, 1.0, 1.0])
    Ap, Ai, Ax = i64([0, 1, 2, 2]), i64([0, 1]), f64([1.0, 1.0])
    kinds, dims = np.array([1], dtype=np.int32), i64([2])

    def create(Pp=Pp, Pi=Pi, Ap=Ap, Ai=Ai, base=0):
        h = C.c_void_p()
        rc = L.hipkkt_kkt_create(C.byref(h), n, m, ptr(Pp), ptr(Pi), ptr(Px), ptr(Ap), ptr(Ai), ptr(Ax), 1,
                                 ptr(kinds), ptr(dims), None, base)
        if rc == 0:
            L.hipkkt_kkt_destroy(h)
        return rc, L.hipkkt_last_error().decode()

    for kw, what in ((dict(base=1), "index_base"),                           # 0-based arrays declared 1-based
                     (dict(Pp=i64([0, 2, 1, 3])), "non-decreasing"),
                     (dict(Pi=i64([0, -1, 2])), "row index out of range"),
                     (dict(Pi=i64([0, 1, 5])), "row index out of range"),
                     (dict(Ai=i64([0, 2])), "row index out of range"),
                     (dict(Ap=i64([0, 1, 0, 2])), "non-decreasing"),
                     (dict(Pi=i64([0, 0, 2]), Pp=i64([0, 1, 2, 3])), None)):  # fine: P(0,1) is upper triangular
        rc, msg = create(**kw)
        if what is None:
            assert rc in (0, -2), (rc, msg)           # valid input: success on a GPU box, HIP error without a device
        else:
            assert rc == -1 and what in msg, (kw, rc, msg)


@pytest.mark.parametrize("ordering", [_lib.ORDER_AMD, _lib.ORDER_ND, _lib.ORDER_NATURAL])
def test_symbolic_analysis_matches_oracle_symbolic(ordering):
    pb = problems.config2(n=3000)
    nsoc = sum(1 for c in pb.cones if isinstance(c, SecondOrderConeT) and c.dim > 4)
    o0 = make_oracle(pb, perm=np.arange(pb.n + pb.m + 2 * nsoc))
    K = o0.K()
    perm, info = _lib.symbolic_analyse(K, ordering=ordering, nd_leaf_size=300)
    assert sorted(perm.tolist()) == list(range(K.shape[0]))
    # the oracle's QDLDL-style etree/column counts on the same permutation give the same nnz(L)
    o = make_oracle(pb, perm=perm)
    assert o.nnzL == info["nnzL"]
    assert info["nnzL_stored"] >= info["nnzL"]
    assert info["N"] == K.shape[0] and info["nnzK"] == K.nnz
    assert info["nlevels"] >= 1 and info["nsuper"] >= 1 and info["max_front"] >= 1


def test_orderings_reduce_fill_and_nd_shortens_the_tree():
    pb = problems.config2(n=6000)
    nsoc = sum(1 for c in pb.cones if isinstance(c, SecondOrderConeT) and c.dim > 4)
    K = make_oracle(pb, perm=np.arange(pb.n + pb.m + 2 * nsoc)).K()
    _, nat = _lib.symbolic_analyse(K, ordering=_lib.ORDER_NATURAL)
    _, amd = _lib.symbolic_analyse(K, ordering=_lib.ORDER_AMD)
    _, nd = _lib.symbolic_analyse(K, ordering=_lib.ORDER_ND, nd_leaf_size=500)
    assert amd["nnzL"] < nat["nnzL"]
    assert nd["nnzL"] < nat["nnzL"]
    assert nd["etree_height"] < amd["etree_height"]       # what the GPU wants: bushy, short trees
    assert nd["nlevels"] < amd["nlevels"]


def test_dense_row_is_ordered_last_like_amd_dense():
    # cfg3-style A = [1'; -I]: the all-ones row has n entries >> 15 sqrt(N) and must go last
    # (amd_dense_scale = 1.5 at directldl_qdldl.jl:24)
    pb = problems.config3(nblocks=4, blk=400)
    o0 = make_oracle(pb, perm=np.arange(pb.n + pb.m))
    perm, _ = _lib.symbolic_analyse(o0.K(), ordering=_lib.ORDER_AMD)
    assert perm[-1] == pb.n       # the zero-cone row sits at KKT index n


def test_cones_new_collapsed_matches_reference_rules():
    # cone_api.jl:96-153
    c = cones_new_collapsed([NonnegativeConeT(3), NonnegativeConeT(3), SecondOrderConeT(3),
                             SecondOrderConeT(1), PSDTriangleConeT(1), NonnegativeConeT(2), ZeroConeT(0),
                             ZeroConeT(2)])
    assert [(type(x).__name__, x.dim) for x in c] == [("NonnegativeConeT", 6), ("SecondOrderConeT", 3),
                                                      ("NonnegativeConeT", 4), ("ZeroConeT", 2)]


def test_problem_generators_are_reproducible_and_interior():
    a, b = problems.config2(n=400), problems.config2(n=400)
    assert (a.A != b.A).nnz == 0 and np.array_equal(a.s0, b.s0) and np.array_equal(a.z0, b.z0)
    o = make_oracle(a)
    assert o.update_scaling(a.s0, a.z0)          # strictly interior iterate
    p5 = problems.config5(n=200, npsd=3, psd_dim=5, nsoc=2, soc_dim=8)
    assert make_oracle(p5).update_scaling(p5.s0, p5.z0)
    assert p5.m == 3 * 15 + 2 * 8


def test_schedule_height_does_not_regress():
    """Every tree level is a round of dependent launches on the GPU, so the height of the supernodal tree after
    amalgamation (tallest child first) and panel splitting (trapezoid capacity, row slices for tall fronts) is a
    first-order performance figure.  Pin it on two structures, with some slack for retuning."""
    for pb, max_levels, max_stored_ratio in ((problems.config2(n=20000), 20, 1.6),
                                             (problems.config3(nblocks=8, blk=300), 10, 1.1)):
        nsoc = sum(1 for c in pb.cones if isinstance(c, SecondOrderConeT) and c.dim > 4)
        K = make_oracle(pb, perm=np.arange(pb.n + pb.m + 2 * nsoc)).K()
        _, info = _lib.symbolic_analyse(K, ordering=_lib.ORDER_ND)
        assert info["nlevels"] <= max_levels, info
        assert info["nnzL_stored"] <= max_stored_ratio * info["nnzL"], info      # explicit zeros stay bounded
