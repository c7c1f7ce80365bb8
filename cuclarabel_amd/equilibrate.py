"""Problem-data scaling on the device.

`equilibrate` = `data_equilibrate!` (`/root/reference/src/problemdata.jl:133-221`): Ruiz
equilibration of `[P A'; A 0]`, `q`, `b`; returns scaled copies and the scalings the solver keeps in
`data.equilibration` (`d`, `e`, `c`, and their inverses).  `rescale_P` / `rescale_A` /
`rescale_q` / `rescale_b` are what `update_P!/A!/q!/b!` apply to new data before it reaches the
KKT solver (`data_updating.jl:56-160,169-194`).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import check, f64, i64, ptr
from .cones import cone_kinds_dims


@dataclass
class Equilibration:          # types.jl: DefaultEquilibration
    d: np.ndarray
    dinv: np.ndarray
    e: np.ndarray
    einv: np.ndarray
    c: float


def equilibrate(P, q, A, b, cones, max_iter=10, min_scaling=1e-4, max_scaling=1e4, device=-1):
    """Returns (P_scaled triu csc, q_scaled, A_scaled csc, b_scaled, Equilibration)."""
    Pt = sp.triu(sp.csc_matrix(P), format="csc")
    Pt.sort_indices()
    Ac = sp.csc_matrix(A).copy()
    Ac.sort_indices()
    n, m = Pt.shape[0], Ac.shape[0]
    Px, Ax = f64(Pt.data).copy(), f64(Ac.data).copy()
    qs, bs = f64(q).copy(), f64(b).copy()
    kinds, dims = cone_kinds_dims(list(cones))
    d, e, c = np.ones(max(n, 1)), np.ones(max(m, 1)), np.ones(1)
    Pp, Pi, Ap, Ai = i64(Pt.indptr), i64(Pt.indices), i64(Ac.indptr), i64(Ac.indices)
    check(_lib.lib().hipkkt_equilibrate(n, m, ptr(Pp), ptr(Pi), ptr(Px), ptr(Ap), ptr(Ai), ptr(Ax), ptr(qs), ptr(bs),
                                        len(kinds), ptr(kinds), ptr(dims), int(max_iter), float(min_scaling),
                                        float(max_scaling), ptr(d), ptr(e), ptr(c), 0, int(device)),
          "hipkkt_equilibrate")
    d, e = d[:n], e[:m]
    Ps = sp.csc_matrix((Px, Pt.indices, Pt.indptr), shape=Pt.shape)
    As = sp.csc_matrix((Ax, Ac.indices, Ac.indptr), shape=Ac.shape)
    return Ps, qs, As, bs, Equilibration(d, 1.0 / d, e, 1.0 / e, float(c[0]))


def _scale_values(M, lscale, rscale, cscale, device=-1):
    M = sp.csc_matrix(M).copy()
    M.sort_indices()
    v = f64(M.data).copy()
    cp, ri = i64(M.indptr), i64(M.indices)
    L, R = f64(lscale), f64(rscale)
    check(_lib.lib().hipkkt_scale_matrix_values(M.shape[0], M.shape[1], ptr(cp), ptr(ri), ptr(v), ptr(L), ptr(R),
                                                float(cscale), 0, int(device)), "hipkkt_scale_matrix_values")
    return sp.csc_matrix((v, M.indices, M.indptr), shape=M.shape)


def rescale_P(P_new, eq, device=-1):
    """update_P! (data_updating.jl:56-80): c D P D on the upper triangle"""
    return _scale_values(sp.triu(sp.csc_matrix(P_new), format="csc"), eq.d, eq.d, eq.c, device)


def rescale_A(A_new, eq, device=-1):
    """update_A! (data_updating.jl:92-115): E A D"""
    return _scale_values(A_new, eq.e, eq.d, 1.0, device)


def rescale_q(q_new, eq):
    """update_q! (data_updating.jl:126-140)"""
    return np.asarray(q_new, float) * eq.d * eq.c


def rescale_b(b_new, eq):
    """update_b! (data_updating.jl:148-160)"""
    return np.asarray(b_new, float) * eq.e
