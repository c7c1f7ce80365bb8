"""Known-answer fixtures typed in from the reference's own tests (data and expected values
only).  Each entry: (P, q, A, b, cones, expected) with cones as (kind, dim)."""
import numpy as np
import scipy.sparse as sp

from cuclarabel_amd.cones import (ZeroConeT, NonnegativeConeT, SecondOrderConeT, PSDTriangleConeT,
                                  cones_new_collapsed)


def basic_qp():
    # test/OptTests/basic_qp.jl:6-19 ; expected :70-73
    P = sp.csc_matrix(np.array([[4.0, 1.0], [1.0, 2.0]]))
    c = np.array([1.0, 1.0])
    A0 = np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]])
    l, u = np.array([1.0, 0.0, 0.0]), np.array([1.0, 0.7, 0.7])
    A = sp.csc_matrix(np.vstack([-A0, A0]))
    b = np.concatenate([-l, u])
    cones = cones_new_collapsed([NonnegativeConeT(3), NonnegativeConeT(3)])
    return P, c, A, b, cones, dict(status="SOLVED", x=np.array([0.3, 0.7]), obj=1.8800000298331538)


def basic_qp_dualinf():
    # basic_qp.jl:22-32 ; expected status :103-116
    P = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, 1.0]]))
    c = np.array([1.0, -1.0])
    A = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, 0.0]]))
    b = np.array([1.0, 1.0])
    return P, c, A, b, [NonnegativeConeT(2)], dict(status="DUAL_INFEASIBLE")


def basic_lp():
    # test/OptTests/basic_lp.jl:6-16 ; expected :32-35
    P = sp.csc_matrix((3, 3))
    A = sp.csc_matrix(np.vstack([np.eye(3), -np.eye(3)]) * 2.0)
    c = np.array([3.0, -2.0, 1.0])
    b = np.ones(6)
    cones = cones_new_collapsed([NonnegativeConeT(3), NonnegativeConeT(3)])
    return P, c, A, b, cones, dict(status="SOLVED", x=np.array([-0.5, 0.5, -0.5]), obj=-3.0)


def basic_socp():
    # test/OptTests/basic_socp.jl:6-30 ; expected :47-53
    P = np.array([[1.4652521089139698, 0.6137176286085666, -1.1527861771130112],
                  [0.6137176286085666, 2.219109946678485, -1.4400420548730628],
                  [-1.1527861771130112, -1.4400420548730628, 1.6014483534926371]])
    A1 = np.vstack([np.eye(3), -np.eye(3)]) * 2.0
    A = sp.csc_matrix(np.vstack([A1, np.eye(3)]))
    c = np.array([0.1, -2.0, 1.0])
    b = np.concatenate([np.ones(6), np.zeros(3)])
    cones = cones_new_collapsed([NonnegativeConeT(3), NonnegativeConeT(3), SecondOrderConeT(3)])
    return sp.csc_matrix(P), c, A, b, cones, dict(status="SOLVED", x=np.array([-0.5, 0.435603, -0.245459]),
                                                 obj=-8.4590e-01)


def eq_constrained(which):
    # test/OptTests/basic_eq_constrained.jl:14-63
    P = sp.identity(3, format="csc")
    if which == 1:
        c = np.zeros(3); A = np.array([[0.0, 1.0, 1.0], [0.0, 1.0, -1.0]]); b = np.array([2.0, 0.0])
        cones = [ZeroConeT(2)]; x = np.array([0.0, 1.0, 1.0])
    elif which == 2:
        c = np.array([1.0, 2.0, 3.0]); A = np.array([[1.0, 1.0, 1.0], [0.0, 1.0, -1.0]]); b = np.array([2.0, 0.0])
        cones = [ZeroConeT(2)]; x = np.array([10.0, 1.0, 1.0]) / 6
    else:   # redundant rows: K is singular without the static regularisation
        c = np.zeros(3); A0 = np.array([[0.0, 1.0, 1.0], [0.0, 1.0, -1.0]])
        A = np.vstack([A0, A0]); b = np.array([2.0, 0.0, 2.0, 0.0]); cones = [ZeroConeT(2), ZeroConeT(2)]
        x = np.array([0.0, 1.0, 1.0])
    return P, c, sp.csc_matrix(A), b, cones, dict(status="SOLVED", x=x)


def unconstrained(feasible=True):
    # test/OptTests/basic_unconstrained.jl:14-29 (feasible: x = -c) and :31-45 (dual infeasible): no constraints
    # at all -- A is 0 x 3, b and the cone list are empty
    P = sp.identity(3, format="csc")
    if feasible:
        c = np.array([1.0, 2.0, -3.0])
        exp = dict(status="SOLVED", x=-c)
    else:
        P = sp.csc_matrix(np.diag([0.0, 1.0, 1.0]))
        c = np.array([1.0, 0.0, 0.0])
        exp = dict(status="DUAL_INFEASIBLE")
    return P, c, sp.csc_matrix((0, 3)), np.zeros(0), [], exp


def basic_qp_univariate():
    # basic_qp.jl:42-57
    P = sp.identity(1, format="csc")
    return P, np.zeros(1), sp.identity(1, format="csc"), np.ones(1), [NonnegativeConeT(1)], \
        dict(status="SOLVED", x=np.zeros(1), obj=0.0)


def basic_qp_priminf():
    # basic_qp.jl:73-86: b[1] = b[4] = -1 (1-based)
    P, c, A, b, cones, _ = basic_qp()
    b = b.copy(); b[0] = -1.0; b[3] = -1.0
    return P, c, A, b, cones, dict(status="PRIMAL_INFEASIBLE")


def basic_qp_dualinf_nonqsd():
    # basic_qp.jl:99-113: one constraint row only -- the KKT matrix is not quasi-definite
    P, c, A, b, _, _ = basic_qp_dualinf()
    return P, c, sp.csc_matrix(A.toarray()[:1, :]), b[:1], [NonnegativeConeT(1)], dict(status="DUAL_INFEASIBLE")


def basic_lp_priminf():
    # basic_lp.jl:40-53
    P, c, A, b, cones, _ = basic_lp()
    b = b.copy(); b[0] = -1.0; b[3] = -1.0
    return P, c, A, b, cones, dict(status="PRIMAL_INFEASIBLE")


def basic_lp_dualinf(ill_conditioned=False):
    # basic_lp.jl:55-68 and :70-84
    P, c, A, b, cones, _ = basic_lp()
    A = A.toarray()
    if ill_conditioned:
        A[0, 0] = np.finfo(float).eps
        A[3, 0] = 0.0
    else:
        A[3, 0] = 1.0        # swap lower bound on first variable to redundant upper bound
    return P, np.array([1.0, 0.0, 0.0]), sp.csc_matrix(A), b, cones, dict(status="DUAL_INFEASIBLE")


def basic_socp_infeasible():
    # basic_socp.jl:71-83: b[7] = -10
    P, c, A, b, cones, _ = basic_socp()
    b = b.copy(); b[6] = -10.0
    return P, c, A, b, cones, dict(status="PRIMAL_INFEASIBLE")


def basic_socp_feasible_sparse():
    # basic_socp.jl:58-69: same data with the cone list [NN(3), NN(6)]; status only
    P, c, A, b, _, _ = basic_socp()
    return P, c, A, b, cones_new_collapsed([NonnegativeConeT(3), NonnegativeConeT(6)]), dict(status="SOLVED")


_SDP_REFSOL = np.array([-3.0729833267361095, 0.3696004167288786, -0.022226685581313674, 0.31441213129613066,
                        -0.026739700851545107, -0.016084530571308823])


def basic_sdp(extra_empty_cone=False):
    # test/OptTests/basic_sdp.jl:6-20 ; expected :37-48 (and the "empty SDP cone" variant :50-70);
    # the same data is the SDP leg of test/OptTests/linear_solvers.jl:51-69
    P = sp.identity(6, format="csc")
    A = sp.identity(6, format="csc")
    b = np.array([-3.0, 1.0, 4.0, 1.0, 2.0, 5.0])      # triu of some indefinite matrix
    cones = [PSDTriangleConeT(3)] + ([PSDTriangleConeT(0)] if extra_empty_cone else [])
    return P, np.zeros(6), A, b, cones_new_collapsed(cones), dict(status="SOLVED", x=_SDP_REFSOL, obj=4.840076866013861)


def basic_sdp_priminf():
    # basic_sdp.jl:72-87: adds a negative-definiteness constraint on x
    P, c, A, b, cones, _ = basic_sdp()
    A2 = sp.vstack([A, -A], format="csc")
    return P, c, A2, np.concatenate([b, np.zeros(6)]), cones + cones, dict(status="PRIMAL_INFEASIBLE")


def basic_sdp_1x1():
    # basic_sdp.jl:89-106: PSDTriangleConeT(1) is collapsed to a nonnegative cone (cone_api.jl:96-153)
    P = sp.identity(1, format="csc")
    return P, np.zeros(1), sp.identity(1, format="csc"), np.ones(1), cones_new_collapsed([PSDTriangleConeT(1)]), \
        dict(status="SOLVED", x=np.zeros(1), obj=0.0)


ALL = {
    "unconstrained": lambda: unconstrained(True),
    "unconstrained_dualinf": lambda: unconstrained(False),
    "basic_qp": basic_qp,
    "basic_qp_univariate": basic_qp_univariate,
    "basic_qp_priminf": basic_qp_priminf,
    "basic_qp_dualinf": basic_qp_dualinf,
    "basic_qp_dualinf_nonqsd": basic_qp_dualinf_nonqsd,
    "basic_lp": basic_lp,
    "basic_lp_priminf": basic_lp_priminf,
    "basic_lp_dualinf": basic_lp_dualinf,
    "basic_lp_dualinf_illcond": lambda: basic_lp_dualinf(True),
    "basic_socp": basic_socp,
    "basic_socp_infeasible": basic_socp_infeasible,
    "basic_socp_feasible_sparse": basic_socp_feasible_sparse,
    "basic_sdp": basic_sdp,
    "basic_sdp_empty_cone": lambda: basic_sdp(True),
    "basic_sdp_priminf": basic_sdp_priminf,
    "basic_sdp_1x1": basic_sdp_1x1,
    "eq_constrained_1": lambda: eq_constrained(1),
    "eq_constrained_2": lambda: eq_constrained(2),
    "eq_constrained_redundant": lambda: eq_constrained(3),
}
