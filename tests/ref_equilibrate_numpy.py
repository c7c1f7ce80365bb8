"""TEST INFRASTRUCTURE: numpy restatement of the reference's Ruiz equilibration, used as the oracle for the
device routine hipkkt_equilibrate.  Follows /root/reference/src/problemdata.jl:133-242 (data_equilibrate!,
scale_data!), src/utils/mathutils.jl:129-244 (kkt_col_norms!, col_norms!, lrscale!, lscale!) and
rectify_equilibration! (cones/coneops_compositecone.jl:28-47, coneops_defaults.jl:32-44,
coneops_nncone.jl:8-17, coneops_zerocone.jl:16-25)."""
import numpy as np
import scipy.sparse as sp

from cuclarabel_amd.cones import NonnegativeConeT, ZeroConeT


def equilibrate_ref(P, q, A, b, cones, max_iter=10, smin=1e-4, smax=1e4):
    Pt = sp.triu(sp.csc_matrix(P), format="csc").copy()
    Pt.sort_indices()
    Ac = sp.csc_matrix(A).copy()
    Ac.sort_indices()
    n, m = Pt.shape[0], Ac.shape[0]
    q, b = np.array(q, float), np.array(b, float)
    d, e, c = np.ones(n), np.ones(m), 1.0
    prow, pcol = Pt.indices, np.repeat(np.arange(n), np.diff(Pt.indptr))
    arow, acol = Ac.indices, np.repeat(np.arange(n), np.diff(Ac.indptr))
    for _ in range(max_iter):
        dw, ew = np.zeros(n), np.zeros(m)
        np.maximum.at(dw, pcol, np.abs(Pt.data)); np.maximum.at(dw, prow, np.abs(Pt.data))     # col_norms_sym!
        np.maximum.at(dw, acol, np.abs(Ac.data))                                                 # col_norms_no_reset!
        np.maximum.at(ew, arow, np.abs(Ac.data))                                                 # row_norms!
        dw[dw == 0] = 1.0; ew[ew == 0] = 1.0
        dw, ew = 1.0 / np.sqrt(dw), 1.0 / np.sqrt(ew)
        dw = np.clip(dw, smin / d, smax / d); ew = np.clip(ew, smin / e, smax / e)
        Pt.data *= dw[prow] * dw[pcol]; Ac.data *= ew[arow] * dw[acol]; q *= dw; b *= ew       # scale_data!
        d *= dw; e *= ew
        cn = np.zeros(n)
        np.maximum.at(cn, pcol, np.abs(Pt.data))                                                 # col_norms!(dwork, P)
        mean_p = cn.mean() if n else 0.0
        nq = np.abs(q).max() if n else 0.0
        if mean_p != 0 and nq != 0:
            ctmp = float(np.clip(1.0 / max(nq, mean_p), smin / c, smax / c))
            Pt.data *= ctmp; q *= ctmp; c *= ctmp
    if max_iter > 0:
        delta = np.ones(m)
        off, changed = 0, False
        for cone in cones:
            k = cone.numel
            if not isinstance(cone, (NonnegativeConeT, ZeroConeT)):
                delta[off:off + k] = e[off:off + k].mean() / e[off:off + k]
                changed = True
            off += k
        if changed:
            Ac.data *= delta[arow]; b *= delta; e *= delta
    return Pt, q, Ac, b, d, e, c
