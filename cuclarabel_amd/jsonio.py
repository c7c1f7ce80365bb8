"""Problem files in the reference's JSON format.

`/root/reference/src/json.jl:118-156` writes a problem as
    {"settings": {...}, "P": csc, "q": [...], "A": csc, "b": [...], "cones": [{"NonnegativeConeT": dim}, ...]}
with csc = {"m", "n", "colptr", "rowval", "nzval"}, **0-based** indices (`json.jl:133-141`), P stored as its
upper triangle (what the solver holds internally, `json.jl:24-38`), and infinite settings replaced
by +-floatmax (`json.jl:80-104`).  `save_problem` / `load_problem` read and write exactly that, so
a problem dumped by `Clarabel.save_to_file` on a Julia machine can be replayed through the C ABI
here without Julia, and vice versa (`Clarabel.load_from_file`).

Cones: the four kinds the KKT path supports (`ZeroConeT`, `NonnegativeConeT`, `SecondOrderConeT`,
`PSDTriangleConeT`; a cone's value is its `dim`, `json.jl:143-156`).  Other cone types in a file
raise `ValueError` (they are outside this build's scope).
"""
import json
import sys

import numpy as np
import scipy.sparse as sp

from .cones import ZeroConeT, NonnegativeConeT, SecondOrderConeT, PSDTriangleConeT

_CONE_TYPES = {c.__name__: c for c in (ZeroConeT, NonnegativeConeT, SecondOrderConeT, PSDTriangleConeT)}
_FLOATMAX = sys.float_info.max


def _lower_csc(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    return {"m": int(M.shape[0]), "n": int(M.shape[1]), "colptr": [int(v) for v in M.indptr],
            "rowval": [int(v) for v in M.indices], "nzval": [float(v) for v in M.data]}


def _parse_csc(d):
    m, n = int(d["m"]), int(d["n"])
    colptr = np.asarray(d["colptr"], dtype=np.int64)
    rowval = np.asarray(d["rowval"], dtype=np.int64)
    nzval = np.asarray(d["nzval"], dtype=np.float64)
    if colptr.size != n + 1 or colptr[0] != 0 or colptr[-1] != rowval.size or rowval.size != nzval.size:
        raise ValueError("malformed CSC matrix in problem file")
    if rowval.size and (rowval.min() < 0 or rowval.max() >= m):
        raise ValueError("row index out of range in problem file")
    return sp.csc_matrix((nzval, rowval, colptr), shape=(m, n))


def _sanitize(settings):
    """json.jl:80-89: infinities cannot be serialised"""
    out = {}
    for k, v in settings.items():
        if isinstance(v, float) and np.isinf(v):
            v = np.sign(v) * _FLOATMAX
        out[k] = v
    return out


def _desanitize(settings):
    """json.jl:92-104"""
    out = {}
    for k, v in settings.items():
        if isinstance(v, float) and abs(v) == _FLOATMAX:
            v = np.sign(v) * np.inf
        out[k] = v
    return out


def save_problem(path, P, q, A, b, cones, settings=None):
    """`save_to_file` (json.jl:25-55).  P may be given full or as its upper triangle; the upper
    triangle is what is written."""
    Pt = sp.triu(sp.csc_matrix(P), format="csc")
    doc = {"settings": _sanitize(dict(settings or {})), "P": _lower_csc(Pt), "q": [float(v) for v in np.asarray(q)],
           "A": _lower_csc(A), "b": [float(v) for v in np.asarray(b)],
           "cones": [{type(c).__name__: int(c.dim)} for c in cones]}
    for c in cones:
        if type(c).__name__ not in _CONE_TYPES:
            raise ValueError(f"unsupported cone type {type(c).__name__}")
    with open(path, "w") as f:
        json.dump(doc, f)


def load_problem(path):
    """`load_from_file` (json.jl:66-88) up to the Solver construction: returns
    (P upper-triangular csc, q, A csc, b, cones, settings dict)."""
    with open(path) as f:
        doc = json.load(f)
    P = sp.triu(_parse_csc(doc["P"]), format="csc")
    A = _parse_csc(doc["A"])
    q = np.asarray(doc["q"], dtype=np.float64)
    b = np.asarray(doc["b"], dtype=np.float64)
    cones = []
    for entry in doc["cones"]:
        if len(entry) != 1:
            raise ValueError("malformed cone entry in problem file")
        (name, val), = entry.items()
        if name not in _CONE_TYPES:
            raise ValueError(f"unsupported cone type {name}")
        cones.append(_CONE_TYPES[name](int(val)))
    if P.shape != (q.size, q.size) or A.shape != (b.size, q.size):
        raise ValueError("inconsistent dimensions in problem file")
    if sum(c.numel for c in cones) != b.size:
        raise ValueError("cone dimensions do not add up to the number of constraint rows")
    return P, q, A, b, cones, _desanitize(dict(doc.get("settings") or {}))
