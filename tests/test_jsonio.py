"""Problem-file I/O in the reference's JSON format (json.jl:118-156) -- SURVEY.md 8 f3."""
import json

import numpy as np
import pytest
import scipy.sparse as sp

from cuclarabel_amd import ipm, jsonio, problems
from cuclarabel_amd.cones import NonnegativeConeT, ZeroConeT, SecondOrderConeT, PSDTriangleConeT
from tests.ipm_backends import OracleBackend

# what Clarabel.save_to_file writes for the problem of the reference's own test (test/UnitTests/test_json.jl:4-12):
# 0-based colptr/rowval, P as its upper triangle, cones as {type name: dim}
REFERENCE_STYLE_FILE = {
    "settings": {"max_iter": 200, "time_limit": 1.7976931348623157e308, "verbose": True, "direct_solve_method": "qdldl"},
    "P": {"m": 2, "n": 2, "colptr": [0, 1, 3], "rowval": [0, 0, 1], "nzval": [4.0, 1.0, 2.0]},
    "q": [1.0, 1.0],
    "A": {"m": 3, "n": 2, "colptr": [0, 2, 4], "rowval": [0, 1, 0, 2], "nzval": [1.0, 1.0, 1.0, 1.0]},
    "b": [1.0, 1.0, 1.0],
    "cones": [{"NonnegativeConeT": 1}, {"ZeroConeT": 1}, {"NonnegativeConeT": 1}],
}


def test_load_reference_style_file(tmp_path):
    f = tmp_path / "p.json"
    f.write_text(json.dumps(REFERENCE_STYLE_FILE))
    P, q, A, b, cones, settings = jsonio.load_problem(str(f))
    np.testing.assert_array_equal(P.toarray(), [[4.0, 1.0], [0.0, 2.0]])
    np.testing.assert_array_equal(A.toarray(), [[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]])
    np.testing.assert_array_equal(q, [1.0, 1.0])
    np.testing.assert_array_equal(b, [1.0, 1.0, 1.0])
    assert cones == [NonnegativeConeT(1), ZeroConeT(1), NonnegativeConeT(1)]
    assert settings["time_limit"] == np.inf and settings["max_iter"] == 200     # floatmax -> Inf (json.jl:92-104)


def test_round_trip_solves_to_the_same_point(tmp_path):
    """test_json.jl:14-25: write, reload, solve both, same solution"""
    P = sp.csc_matrix(np.array([[4.0, 1.0], [1.0, 2.0]]))
    q = np.array([1.0, 1.0])
    A = sp.csc_matrix(np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]]))
    b = np.array([1.0, 1.0, 1.0])
    cones = [NonnegativeConeT(1), ZeroConeT(1), NonnegativeConeT(1)]
    f = tmp_path / "p.json"
    jsonio.save_problem(str(f), P, q, A, b, cones, settings={"max_iter": 50, "time_limit": np.inf})
    doc = json.loads(f.read_text())
    assert doc["P"]["colptr"] == [0, 1, 3] and doc["P"]["rowval"] == [0, 0, 1]          # 0-based, upper triangle
    assert doc["cones"] == [{"NonnegativeConeT": 1}, {"ZeroConeT": 1}, {"NonnegativeConeT": 1}]
    assert doc["settings"]["time_limit"] == jsonio._FLOATMAX
    P2, q2, A2, b2, cones2, _ = jsonio.load_problem(str(f))
    r1 = ipm.solve(P, q, A, b, cones, OracleBackend(P, A, cones))
    r2 = ipm.solve(P2, q2, A2, b2, cones2, OracleBackend(P2, A2, cones2))
    assert r1.status == r2.status == "SOLVED"
    np.testing.assert_allclose(r1.x, r2.x, atol=1e-10)


def test_all_supported_cone_kinds_and_sizes_survive(tmp_path):
    pb = problems.small_mixed(seed=5)
    f = tmp_path / "mixed.json"
    jsonio.save_problem(str(f), pb.P, pb.q, pb.A, pb.b, pb.cones)
    P, q, A, b, cones, _ = jsonio.load_problem(str(f))
    assert cones == list(pb.cones)
    assert any(isinstance(c, PSDTriangleConeT) for c in cones) and any(isinstance(c, SecondOrderConeT) for c in cones)
    np.testing.assert_array_equal(P.toarray(), sp.triu(pb.P).toarray())
    np.testing.assert_array_equal(A.toarray(), pb.A.toarray())
    np.testing.assert_array_equal(q, pb.q)
    np.testing.assert_array_equal(b, pb.b)


@pytest.mark.parametrize("mutate,msg", [
    (lambda d: d["cones"].append({"ExponentialConeT": []}), "unsupported cone"),
    (lambda d: d["cones"].pop(), "do not add up"),
    (lambda d: d["A"]["rowval"].__setitem__(0, 7), "out of range"),
    (lambda d: d["P"]["colptr"].__setitem__(2, 5), "malformed"),
])
def test_malformed_files_are_rejected(tmp_path, mutate, msg):
    d = json.loads(json.dumps(REFERENCE_STYLE_FILE))
    mutate(d)
    f = tmp_path / "bad.json"
    f.write_text(json.dumps(d))
    with pytest.raises(ValueError, match=msg):
        jsonio.load_problem(str(f))
