"""Independent dense numpy construction of the KKT matrix and of the cone scalings,
used to check the oracle (TEST INFRASTRUCTURE).  Written from the mathematical
definitions (NT scaling), not from the oracle's code:

    K = [ P   A'    0 ]        Hs = W'W per cone, SOC(dim>4) in its sparse form
        [ A  -Hs    V ]        Hs = eta^2 (D + u u' - v v'),  V = -eta^2 [v u],
        [ 0   V'    E ]        E = eta^2 diag(-1, +1)
"""
import numpy as np

from cuclarabel_amd.cones import (ZeroConeT, NonnegativeConeT, SecondOrderConeT,
                                  PSDTriangleConeT)
from cuclarabel_amd.problems import svec_to_mat


def soc_nt(s, z):
    """eta and normalised w for the second-order cone (NT point), textbook formulas."""
    J = np.ones(len(s)); J[1:] = -1
    sres = np.sqrt(s @ (J * s)); zres = np.sqrt(z @ (J * z))
    eta = np.sqrt(sres / zres)
    sb, zb = s / sres, z / zres
    gamma = np.sqrt((1 + sb @ zb) / 2)
    w = (sb + J * zb) / (2 * gamma)
    return eta, w


def soc_W2(s, z):
    eta, w = soc_nt(s, z)
    J = np.diag(np.r_[1.0, -np.ones(len(s) - 1)])
    return eta ** 2 * (2 * np.outer(w, w) - J)


def psd_W2(s, z, d):
    """W'W in svec coordinates for the PSD cone: (R R') (x)_s (R R'), built column by
    column by applying X -> (RR') X (RR') to the svec basis."""
    S, Z = svec_to_mat(s, d), svec_to_mat(z, d)
    # NT scaling point: W = R R' with R' Z R = R^{-1} S R^{-T} = Lambda
    Zh = _sqrtm(Z)
    M = Zh @ S @ Zh
    Wm = np.linalg.solve(Zh, _sqrtm(M)) @ np.linalg.inv(Zh)   # Z^{-1/2} (Z^{1/2} S Z^{1/2})^{1/2} Z^{-1/2}
    t = d * (d + 1) // 2
    H = np.zeros((t, t))
    from cuclarabel_amd.problems import mat_to_svec
    for k in range(t):
        e = np.zeros(t); e[k] = 1
        X = svec_to_mat(e, d)
        H[:, k] = mat_to_svec(Wm @ X @ Wm)
    return H


def _sqrtm(M):
    w, V = np.linalg.eigh((M + M.T) / 2)
    return (V * np.sqrt(w)) @ V.T


def dense_kkt_from_cones(pb, s, z):
    n, m = pb.n, pb.m
    nsparse = sum(1 for c in pb.cones if isinstance(c, SecondOrderConeT) and c.dim > 4)
    N = n + m + 2 * nsparse
    K = np.zeros((N, N))
    Pd = pb.P.toarray()
    K[:n, :n] = Pd + np.triu(Pd, 1).T
    Ad = pb.A.toarray()
    K[n:n + m, :n] = Ad
    K[:n, n:n + m] = Ad.T
    off = 0
    pcol = n + m
    for c in pb.cones:
        k = c.numel
        r = slice(n + off, n + off + k)
        sc, zc = s[off:off + k], z[off:off + k]
        if isinstance(c, ZeroConeT):
            pass
        elif isinstance(c, NonnegativeConeT):
            K[r, r] = -np.diag(sc / zc)
        elif isinstance(c, SecondOrderConeT):
            H = soc_W2(sc, zc)
            if c.dim <= 4:
                K[r, r] = -H
            else:
                eta, w = soc_nt(sc, zc)
                wsq = w @ w
                d = 0.5 / wsq
                u0 = np.sqrt(wsq - d); u1 = 2 * w[0] / u0
                v1 = np.sqrt(2 * (2 + 1 / wsq) / (2 * wsq - 1 / wsq))
                u = np.r_[u0, u1 * w[1:]]; v = np.r_[0.0, v1 * w[1:]]
                D = np.ones(k); D[0] = d
                # the identity the reference unit-tests (test_coneops_secondordercone.jl:60-66)
                np.testing.assert_allclose(eta ** 2 * (np.diag(D) + np.outer(u, u) - np.outer(v, v)), H,
                                           rtol=1e-10, atol=1e-12)
                K[r, r] = -eta ** 2 * np.diag(D)
                K[r, pcol] = K[pcol, r] = -eta ** 2 * v
                K[r, pcol + 1] = K[pcol + 1, r] = -eta ** 2 * u
                K[pcol, pcol] = -eta ** 2
                K[pcol + 1, pcol + 1] = eta ** 2
                pcol += 2
        elif isinstance(c, PSDTriangleConeT):
            K[r, r] = -psd_W2(sc, zc, c.dim)
        off += k
    return K
