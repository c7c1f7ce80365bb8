"""Synthetic (P, q, A, b, cones) generators for the five BASELINE.json configurations,
exactly as SURVEY.md section 8(d) specifies them (numpy `default_rng(1000 + cfg [+ j])`).

These are workload generators for tests and bench.py -- host-side numpy/scipy, not part of
the solver path.  Each returns a `Problem` with P as upper-triangular CSC, A as CSC, the
cone list, and a reproducible strictly-interior iterate (s0, z0) from which the KKT values
for timing are taken (so the benchmark needs no IPM loop).
"""
from dataclasses import dataclass, field
import numpy as np
import scipy.sparse as sp

from .cones import (ZeroConeT, NonnegativeConeT, SecondOrderConeT, PSDTriangleConeT,
                    total_numel)


@dataclass
class Problem:
    name: str
    P: sp.csc_matrix          # n x n, upper triangle only
    q: np.ndarray
    A: sp.csc_matrix          # m x n
    b: np.ndarray
    cones: list
    s0: np.ndarray
    z0: np.ndarray
    x0: np.ndarray
    meta: dict = field(default_factory=dict)

    @property
    def n(self):
        return self.P.shape[0]

    @property
    def m(self):
        return self.A.shape[0]


def _triu_csc(P):
    P = sp.triu(sp.csc_matrix(P), format="csc")
    P.sort_indices()
    P.sum_duplicates()
    return P


def _csc(A):
    A = sp.csc_matrix(A)
    A.sum_duplicates()
    A.sort_indices()
    return A


def interior_point(cones, rng):
    """Strictly interior point per cone: NN |N(0,1)|+0.1; SOC [||t||+1; t]; PSD svec(GG'+I);
    zero cone rows get 0 (they carry no scaling)."""
    out = np.zeros(total_numel(cones))
    off = 0
    for c in cones:
        k = c.numel
        if isinstance(c, NonnegativeConeT):
            out[off:off + k] = np.abs(rng.standard_normal(k)) + 0.1
        elif isinstance(c, SecondOrderConeT):
            t = rng.standard_normal(k - 1)
            out[off] = np.linalg.norm(t) + 1.0
            out[off + 1:off + k] = t
        elif isinstance(c, PSDTriangleConeT):
            d = c.dim
            G = rng.standard_normal((d, d))
            M = G @ G.T + np.eye(d)
            out[off:off + k] = mat_to_svec(M)
        off += k
    return out


def mat_to_svec(M):
    """Column-major upper triangle, off-diagonals scaled by sqrt(2)
    (`coneops_psdtrianglecone.jl:486-497`)."""
    d = M.shape[0]
    out = np.empty(d * (d + 1) // 2)
    idx = 0
    for col in range(d):
        for row in range(col + 1):
            out[idx] = M[row, col] if row == col else (M[row, col] + M[col, row]) / np.sqrt(2.0)
            idx += 1
    return out


def svec_to_mat(x, d):
    M = np.zeros((d, d))
    idx = 0
    for col in range(d):
        for row in range(col + 1):
            if row == col:
                M[row, col] = x[idx]
            else:
                M[row, col] = M[col, row] = x[idx] / np.sqrt(2.0)
            idx += 1
    return M


def _finish(name, P, A, cones, rng, q=None, b=None, meta=None, seed=None):
    n, m = P.shape[0], A.shape[0]
    x0 = rng.standard_normal(n)
    s0 = interior_point(cones, rng)
    if b is None:
        b = A @ x0 + s0
    if q is None:
        q = rng.standard_normal(n)
    zrng = np.random.default_rng((seed if seed is not None else 0) + 7_000_000)
    z0 = interior_point(cones, zrng)
    return Problem(name, _triu_csc(P), q, _csc(A), np.asarray(b, dtype=float), cones,
                   s0, z0, x0, meta or {})


def _local_window_A(m, n, nnz_per_row, rng, halfwidth=50, long_range_frac=0.0):
    rows = np.repeat(np.arange(m), nnz_per_row)
    home = (np.arange(m) * n) // m
    cols = np.repeat(home, nnz_per_row) + rng.integers(-halfwidth, halfwidth + 1, size=m * nnz_per_row)
    cols = np.clip(cols, 0, n - 1)
    if long_range_frac > 0:
        sel = rng.random(cols.shape[0]) < long_range_frac
        cols[sel] = rng.integers(0, n, size=int(sel.sum()))
    vals = rng.standard_normal(m * nnz_per_row)
    return sp.coo_matrix((vals, (rows, cols)), shape=(m, n))


def config1(seed=1001, n=500, m=1000, density=0.01):
    """cfg1: small random QP, NN(m); P = B'B + 0.01 I, B = sprandn(n,n,0.01); A = sprandn(m,n,0.01)."""
    rng = np.random.default_rng(seed)
    B = sp.random(n, n, density=density, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    P = (B.T @ B + 0.01 * sp.identity(n)).tocsc()
    A = sp.random(m, n, density=density, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    cones = [NonnegativeConeT(m)]
    return _finish("cfg1_qp", P, A, cones, rng, seed=seed)


def config2(seed=1002, n=100_000, soc_dim=100, long_range_frac=0.0, halfwidth=50):
    """cfg2: SOCP, m = 2n = NN(n) + (n/soc_dim) x SOC(soc_dim); P = diag(U(0.1,1));
    A has 4 nnz/row at columns home(i) + U{-50..50}."""
    rng = np.random.default_rng(seed)
    m = 2 * n
    nsoc = n // soc_dim
    P = sp.diags(rng.uniform(0.1, 1.0, size=n)).tocsc()
    A = _local_window_A(m, n, 4, rng, halfwidth=halfwidth, long_range_frac=long_range_frac)
    cones = [NonnegativeConeT(n)] + [SecondOrderConeT(soc_dim) for _ in range(nsoc)]
    return _finish(f"cfg2_socp_n{n}", P, A, cones, rng, seed=seed,
                   meta={"long_range_frac": long_range_frac})


def config3(seed=1003, nblocks=100, blk=500):
    """cfg3: portfolio-style QP, P = blockdiag of dense PSD blocks GG'/blk + 0.1 I;
    A = [1'; -I], b = [1; 0], cones Zero(1) + NN(n)."""
    rng = np.random.default_rng(seed)
    n = nblocks * blk
    blocks = []
    for _ in range(nblocks):
        G = rng.standard_normal((blk, blk))
        blocks.append(sp.csc_matrix(np.triu(G @ G.T / blk + 0.1 * np.eye(blk))))
    P = sp.block_diag(blocks, format="csc")
    A = sp.vstack([sp.csc_matrix(np.ones((1, n))), -sp.identity(n, format="csc")], format="csc")
    b = np.concatenate([[1.0], np.zeros(n)])
    cones = [ZeroConeT(1), NonnegativeConeT(n)]
    return _finish(f"cfg3_portfolio_n{n}", P, A, cones, rng, b=b, seed=seed)


def config4(j=0, seed=1004, n=10_000):
    """cfg4: element j of the batch of independent SOCPs (cfg2's generator at n=10k)."""
    pb = config2(seed=seed + j, n=n)
    pb.name = f"cfg4_socp_n{n}_j{j}"
    return pb


def config5(seed=1005, n=5000, npsd=200, psd_dim=20, nsoc=100, soc_dim=50):
    """cfg5: SDP, 200 x PSD(20) + 100 x SOC(50); A 3 nnz/row local window; P = 1e-3 I."""
    rng = np.random.default_rng(seed)
    cones = [PSDTriangleConeT(psd_dim) for _ in range(npsd)] + \
            [SecondOrderConeT(soc_dim) for _ in range(nsoc)]
    m = total_numel(cones)
    P = (1e-3 * sp.identity(n)).tocsc()
    A = _local_window_A(m, n, 3, rng)
    return _finish(f"cfg5_sdp_n{n}", P, A, cones, rng, seed=seed)


def config_unstructured(seed=1012, n=10_000):
    """Fully unstructured stress variant of cfg2 at n=10k only."""
    rng = np.random.default_rng(seed)
    m = 2 * n
    P = sp.diags(rng.uniform(0.1, 1.0, size=n)).tocsc()
    rows = np.repeat(np.arange(m), 4)
    cols = rng.integers(0, n, size=4 * m)
    A = sp.coo_matrix((rng.standard_normal(4 * m), (rows, cols)), shape=(m, n))
    cones = [NonnegativeConeT(n)] + [SecondOrderConeT(100) for _ in range(n // 100)]
    return _finish(f"cfg2u_socp_n{n}", P, A, cones, rng, seed=seed)


def small_mixed(seed=7, n=40, nn=30, socs=(3, 4, 5, 12), psds=(2, 3, 4), zero=3, density=0.15):
    """Small problem touching every cone kind and both SOC forms -- the parity workhorse."""
    rng = np.random.default_rng(seed)
    cones = []
    if zero:
        cones.append(ZeroConeT(zero))
    cones.append(NonnegativeConeT(nn))
    cones += [SecondOrderConeT(d) for d in socs]
    cones += [PSDTriangleConeT(d) for d in psds]
    m = total_numel(cones)
    B = sp.random(n, n, density=density, random_state=rng, data_rvs=rng.standard_normal)
    P = (B.T @ B).tocsc()
    A = sp.random(m, n, density=density, random_state=rng, data_rvs=rng.standard_normal).tocsc()
    return _finish(f"mixed_n{n}", P, A, cones, rng, seed=seed)


def block_diagonal(pbs, name=None):
    """Stack independent problems into ONE block-diagonal problem (P = blkdiag(P_j), A = blkdiag(A_j),
    cones concatenated): the "block-diagonal problem batch" of BASELINE.json's north_star / cfg4.  One
    KKT handle then factorises and solves all of them in the same per-level launches."""
    P = sp.block_diag([pb.P for pb in pbs], format="csc")
    A = sp.block_diag([pb.A for pb in pbs], format="csc")
    cat = lambda key: np.concatenate([getattr(pb, key) for pb in pbs])
    return Problem(name or f"blockdiag_{len(pbs)}x_{pbs[0].name}", _triu_csc(P), cat("q"), _csc(A), cat("b"),
                   [c for pb in pbs for c in pb.cones], cat("s0"), cat("z0"), cat("x0"),
                   dict(blocks=[(pb.n, pb.m) for pb in pbs]))


# ---------------------------------------------------------------------------------------------------------------
# Structure zoo: sparsity patterns OUTSIDE the five BASELINE configurations (whose KKT graphs are banded, block-
# dense or block-banded).  The schedule's thresholds were fitted on those five; these generators give the parity
# tests and scripts/bench_zoo.py elimination trees of other shapes -- mesh separators that grow with the subgraph,
# a root front that dominates, hubs, forests of unequal trees, trees with no top at all, LPs with P = 0.
# ---------------------------------------------------------------------------------------------------------------

def _laplacian(dims):
    """Graph Laplacian of a regular grid (5-point in 2-D, 7-point in 3-D) as the Kronecker sum of path Laplacians."""
    def path(k):
        d = np.full(k, 2.0)
        d[0] = d[-1] = 1.0
        return sp.diags([-np.ones(k - 1), d, -np.ones(k - 1)], [-1, 0, 1], format="csr")
    L = sp.csr_matrix((1, 1))
    for k in dims:
        L = sp.kron(L, sp.identity(k)) + sp.kron(sp.identity(L.shape[0]), path(k))
    return L.tocsc()


def zoo_grid(dims=(96, 96), seed=2001, soc_dim=8):
    """Mesh QP: P = grid Laplacian + 0.1 I (2-D or 3-D), bounds -x <= b as NN(n), and one SOC(soc_dim) per run of
    soc_dim - 1 consecutive variables.  Nested dissection of a mesh: separators of ~sqrt(n) (2-D) or ~n^(2/3) (3-D)
    nodes at the root, shrinking by sqrt(2) per level -- unlike cfg2's constant-width separators."""
    rng = np.random.default_rng(seed)
    n = int(np.prod(dims))
    P = _laplacian(dims) + 0.1 * sp.identity(n)
    nsoc = n // (soc_dim - 1)
    rows, cols, vals = [], [], []
    for j in range(nsoc):                              # SOC j: (t; x_run) with t a constant slack row (empty row of A)
        r0 = n + j * soc_dim
        run = np.arange(j * (soc_dim - 1), (j + 1) * (soc_dim - 1))
        rows.append(r0 + 1 + np.arange(soc_dim - 1))
        cols.append(run)
        vals.append(-np.ones(soc_dim - 1))
    A_soc = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows) - n, np.concatenate(cols))),
                          shape=(nsoc * soc_dim, n))
    A = sp.vstack([-sp.identity(n), A_soc], format="csc")
    cones = [NonnegativeConeT(n)] + [SecondOrderConeT(soc_dim) for _ in range(nsoc)]
    return _finish("zoo_grid_" + "x".join(map(str, dims)), P, A, cones, rng, seed=seed)


def zoo_chain(n=40_000, seed=2002):
    """Total-variation-like QP on a path: P tridiagonal, A = first differences (NN) -- every separator is ONE node,
    so the elimination tree is as deep and as thin as nested dissection can make it."""
    rng = np.random.default_rng(seed)
    P = sp.diags([-0.4 * np.ones(n - 1), np.ones(n) + rng.uniform(0, 1, n), -0.4 * np.ones(n - 1)], [-1, 0, 1])
    D = sp.diags([-np.ones(n - 1), np.ones(n - 1)], [0, 1], shape=(n - 1, n))
    A = sp.vstack([D, -D], format="csc")
    return _finish(f"zoo_chain_n{n}", P, A, [NonnegativeConeT(2 * (n - 1))], rng, seed=seed)


def zoo_arrow(n=20_000, dense_rows=6, seed=2003):
    """cfg2-like local couplings plus `dense_rows` equality rows that touch EVERY variable (budget / moment
    constraints): their multipliers are hubs of the KKT graph and must end up at the root."""
    rng = np.random.default_rng(seed)
    P = sp.diags(rng.uniform(0.1, 1.0, size=n))
    A_loc = _local_window_A(n, n, 3, rng, halfwidth=20)
    A_dense = sp.csc_matrix(rng.standard_normal((dense_rows, n)))
    A = sp.vstack([A_dense, A_loc], format="csc")
    return _finish(f"zoo_arrow_n{n}_k{dense_rows}", P, A, [ZeroConeT(dense_rows), NonnegativeConeT(n)], rng, seed=seed)


def zoo_powerlaw(n=8000, seed=2004):
    """Scale-free coupling: each row of A picks 3 columns with probability ~ 1/rank -- a few hub variables of
    degree in the thousands, a long tail of degree-one variables."""
    rng = np.random.default_rng(seed)
    m = 2 * n
    p = 1.0 / np.arange(1, n + 1)
    p /= p.sum()
    cols = rng.choice(n, size=3 * m, p=p)
    rows = np.repeat(np.arange(m), 3)
    A = sp.coo_matrix((rng.standard_normal(3 * m), (rows, cols)), shape=(m, n))
    P = sp.diags(rng.uniform(0.1, 1.0, size=n))
    cones = [NonnegativeConeT(n)] + [SecondOrderConeT(50) for _ in range(n // 50)]
    return _finish(f"zoo_powerlaw_n{n}", P, A, cones, rng, seed=seed)


def zoo_big_soc(n=12_000, soc_dim=4000, nsoc=3, seed=2005):
    """A few very large second-order cones (sparse expansion: two extra KKT columns of soc_dim entries each,
    `coneops_socone.jl:125-192`) over local couplings."""
    rng = np.random.default_rng(seed)
    m = nsoc * soc_dim
    P = sp.diags(rng.uniform(0.1, 1.0, size=n))
    A = _local_window_A(m, n, 3, rng, halfwidth=30)
    return _finish(f"zoo_bigsoc_{nsoc}x{soc_dim}", P, A, [SecondOrderConeT(soc_dim) for _ in range(nsoc)], rng, seed=seed)


def zoo_forest(seed=2006, ntiny=400):
    """One block-diagonal problem of very unequal parts: a cfg2 at n = 4000, a dense 150-variable QP and `ntiny`
    problems of 6 variables -- a forest whose trees end at every level."""
    parts = [config2(seed=seed, n=4000), small_mixed(seed=seed + 1, n=150, nn=200, socs=(20, 30), psds=(), zero=0, density=1.0)]
    parts += [small_mixed(seed=seed + 2 + j, n=6, nn=5, socs=(3,), psds=(), zero=0, density=0.5) for j in range(ntiny)]
    return block_diagonal(parts, name=f"zoo_forest_{ntiny}")


def zoo_diag(n=30_000, seed=2007):
    """Separable problem: P diagonal, A = -I, NN(n).  K permutes to n independent 2 x 2 blocks: every supernode is
    a root, the tree has no top for the persistent kernels to take."""
    rng = np.random.default_rng(seed)
    P = sp.diags(rng.uniform(0.1, 1.0, size=n))
    return _finish(f"zoo_diag_n{n}", P, -sp.identity(n), [NonnegativeConeT(n)], rng, seed=seed)


def zoo_dense(n=350, m=500, seed=2008):
    """Fully dense P and A: the whole KKT matrix is ONE front (a chain of panels), no tree at all."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n))
    P = sp.csc_matrix(G @ G.T / n + 0.1 * np.eye(n))
    A = sp.csc_matrix(rng.standard_normal((m, n)))
    return _finish(f"zoo_dense_n{n}", P, A, [NonnegativeConeT(m - 60), SecondOrderConeT(60)], rng, seed=seed)


def zoo_lp_transport(nsrc=120, ndst=150, seed=2009):
    """Transportation LP: P = 0 (the (1,1) block of K is the static regulariser alone), flow conservation as a
    Zero cone over a bipartite incidence matrix, x >= 0 as NN."""
    rng = np.random.default_rng(seed)
    n = nsrc * ndst
    i, j = np.divmod(np.arange(n), ndst)
    A_eq = sp.coo_matrix((np.ones(2 * n), (np.concatenate([i, nsrc + j]), np.concatenate([np.arange(n)] * 2))),
                         shape=(nsrc + ndst, n)).tocsr()[:-1]      # the last balance row is implied by the others
    A = sp.vstack([A_eq, -sp.identity(n)], format="csc")
    P = sp.csc_matrix((n, n))
    return _finish(f"zoo_lp_transport_{nsrc}x{ndst}", P, A, [ZeroConeT(nsrc + ndst - 1), NonnegativeConeT(n)], rng, seed=seed)


def zoo_equality_heavy(n=15_000, seed=2010):
    """Half as many equality rows as variables (Zero cone: -eps pivots on a third of the diagonal) over a banded P."""
    rng = np.random.default_rng(seed)
    meq = n // 2
    P = sp.diags([0.2 * np.ones(n - 2), np.ones(n) + rng.uniform(0, 1, n), 0.2 * np.ones(n - 2)], [-2, 0, 2])
    A = sp.vstack([_local_window_A(meq, n, 3, rng, halfwidth=10), -sp.identity(n)], format="csc")
    return _finish(f"zoo_equality_n{n}", P, A, [ZeroConeT(meq), NonnegativeConeT(n)], rng, seed=seed)


ZOO = [
    ("grid2d_160", lambda: zoo_grid((160, 160))),
    ("grid3d_24", lambda: zoo_grid((24, 24, 24), seed=2011)),
    ("chain_150k", lambda: zoo_chain(n=150_000)),
    ("arrow_6", lambda: zoo_arrow(n=60_000)),
    ("powerlaw_20k", lambda: zoo_powerlaw(n=20_000)),
    ("big_soc_4000", lambda: zoo_big_soc(n=30_000, soc_dim=4000, nsoc=8)),
    ("forest_400", lambda: zoo_forest()),
    ("diag_100k", lambda: zoo_diag(n=100_000)),
    ("dense_600", lambda: zoo_dense(n=600, m=900)),
    ("lp_transport", lambda: zoo_lp_transport(200, 260)),
    ("equality_heavy", lambda: zoo_equality_heavy(n=60_000)),
]

# the same shapes small enough for the CPU suite (oracle against scipy)
ZOO_SMALL = [
    ("grid2d_24", lambda: zoo_grid((24, 24))),
    ("grid3d_8", lambda: zoo_grid((8, 8, 8), seed=2011)),
    ("chain_2k", lambda: zoo_chain(n=2000)),
    ("arrow_3", lambda: zoo_arrow(n=1500, dense_rows=3)),
    ("powerlaw_1k", lambda: zoo_powerlaw(n=1000)),
    ("big_soc_600", lambda: zoo_big_soc(n=1500, soc_dim=600, nsoc=2)),
    ("forest_30", lambda: zoo_forest(ntiny=30)),
    ("diag_2k", lambda: zoo_diag(n=2000)),
    ("dense_80", lambda: zoo_dense(n=80, m=120)),
    ("lp_transport_small", lambda: zoo_lp_transport(20, 30)),
    ("equality_2k", lambda: zoo_equality_heavy(n=2000)),
]
