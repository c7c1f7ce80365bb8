"""Interior-point test driver over the KKT-solver boundary.

A host-side (numpy) restatement of the reference's IPM loop for Zero / Nonnegative /
SecondOrder / PSDTriangle cones, without presolve, equilibration or chordal decomposition:

    solve!                          /root/reference/src/solver.jl:189-380
    default start                   solver.jl:383-404, kktsystem.jl:95-132, variables.jl:196-237
    residuals, mu                   residuals.jl:1-37, variables.jl:1-10
    termination                     info.jl:1-120,225-330 (full tolerances, settings.jl:76-81)
    reduced KKT solves, dtau etc.   kktsystem.jl:62-215
    rhs construction                variables.jl:107-190
    step lengths                    variables.jl:13-45, coneops_nncone.jl:151-170, coneops_socone.jl:443-512
    NT scaling, W, lambda           coneops_nncone.jl:77-114, coneops_socone.jl:75-154,302-412
    PSD cone                        coneops_psdtrianglecone.jl:8-44 (margins, shift), :78-143 (scaling), :164-254,
                                    :299-466 (mul_Hs!, ds offsets, W / W^-1, Jordan product, step length)

It exists to drive a KKT backend through exactly the call sequence Clarabel uses
(`kktsolver_update!` -> constant-RHS solve -> affine solve -> combined solve, every iteration)
so that the reference's end-to-end known answers (test/OptTests/basic_*.jl) become parity
checks for the boundary.  The backend is anything with the `kktsolver_*` methods of
`cuclarabel_amd.kktsolver.HipKKTSolver`; cone scaling for the KKT update is done by the backend
(`update(s, z)`), the driver keeps its own numpy copy of the scaling for the step computations,
which stay on the host in the reference too (kktsystem.jl:135-215).
"""
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from .cones import ZeroConeT, NonnegativeConeT, SecondOrderConeT, PSDTriangleConeT

SOLVED, PRIMAL_INFEASIBLE, DUAL_INFEASIBLE = "SOLVED", "PRIMAL_INFEASIBLE", "DUAL_INFEASIBLE"
MAX_ITERATIONS, NUMERICAL_ERROR, INSUFFICIENT_PROGRESS, UNSOLVED = \
    "MAX_ITERATIONS", "NUMERICAL_ERROR", "INSUFFICIENT_PROGRESS", "UNSOLVED"
ALMOST_SOLVED = "ALMOST_SOLVED"


@dataclass
class IPMSettings:            # settings.jl:70-106 (code defaults)
    max_iter: int = 200
    max_step_fraction: float = 0.99
    tol_gap_abs: float = 1e-8
    tol_gap_rel: float = 1e-8
    tol_feas: float = 1e-8
    tol_infeas_abs: float = 1e-8
    tol_infeas_rel: float = 1e-8
    tol_ktratio: float = 1e-6
    min_terminate_step_length: float = 1e-4
    # reduced-accuracy tolerances (settings.jl:88-93), used by info_post_process! after an error exit
    reduced_tol_gap_abs: float = 5e-5
    reduced_tol_gap_rel: float = 5e-5
    reduced_tol_feas: float = 1e-4


@dataclass
class IPMResult:
    status: str
    x: np.ndarray
    z: np.ndarray
    s: np.ndarray
    obj_val: float
    obj_val_dual: float
    iterations: int
    kkt_ir_rounds: int = 0
    history: list = field(default_factory=list)


# ------------------------------------------------------------------------------------------
#  host cone operations (symmetric cones only)
# ------------------------------------------------------------------------------------------
class _Cone:
    def __init__(self, spec, off):
        self.spec, self.off, self.n = spec, off, spec.numel
        self.rng = slice(off, off + spec.numel)

    degree = 0


class _Zero(_Cone):
    degree = 0

    def margins(self, z):
        return np.finfo(float).max, 0.0

    def unit_shift(self, z, a, primal):
        if primal:
            z[:] = 0.0

    def update_scaling(self, s, z):
        return True

    def affine_ds(self, s):
        return np.zeros(self.n)

    def combined_ds_shift(self, dz, ds, sigma_mu):
        return np.zeros(self.n)

    def ds_from_dz_offset(self, ds, z):
        return np.zeros(self.n)

    def mul_Hs(self, x):
        return np.zeros(self.n)

    def get_Hs(self):                                    # coneops_zerocone.jl get_Hs!: zeros
        return np.zeros(self.n)

    def step_length(self, dz, ds, z, s, amax):
        return amax


class _NN(_Cone):
    @property
    def degree(self):
        return self.n

    def margins(self, z):
        a = z.min() if self.n else np.finfo(float).max
        return a, float(np.sum(z[z > 0]))

    def unit_shift(self, z, a, primal):
        z += a

    def update_scaling(self, s, z):
        self.lam = np.sqrt(s * z)
        self.w = np.sqrt(s / z)
        return bool(np.all(np.isfinite(self.w)))

    def affine_ds(self, s):
        return self.lam ** 2

    def combined_ds_shift(self, dz, ds, sigma_mu):
        return (ds / self.w) * (self.w * dz) - sigma_mu          # W^-1 ds o W dz - sigma mu e

    def ds_from_dz_offset(self, ds, z):
        return ds / z

    def mul_Hs(self, x):
        return self.w * (self.w * x)

    def get_Hs(self):                                    # coneops_nncone.jl get_Hs!: w.^2
        return self.w ** 2

    def step_length(self, dz, ds, z, s, amax):
        a = amax
        m = dz < 0
        if m.any():
            a = min(a, float(np.min(-z[m] / dz[m])))
        m = ds < 0
        if m.any():
            a = min(a, float(np.min(-s[m] / ds[m])))
        return a


def _soc_res(v):
    n1 = np.linalg.norm(v[1:])
    return (v[0] - n1) * (v[0] + n1)


class _SOC(_Cone):
    degree = 1

    def margins(self, z):
        a = z[0] - np.linalg.norm(z[1:])
        return a, max(0.0, a)

    def unit_shift(self, z, a, primal):
        z[0] += a

    def update_scaling(self, s, z):                      # coneops_socone.jl:75-123
        rs, rz = _soc_res(s), _soc_res(z)
        if not (rs > 0 and rz > 0):
            return False
        ss, zs = np.sqrt(rs), np.sqrt(rz)
        self.eta = np.sqrt(ss / zs)
        w = s / ss
        w[0] += z[0] / zs
        w[1:] -= z[1:] / zs
        rw = _soc_res(w)
        if not rw > 0:
            return False
        ws = np.sqrt(rw)
        w /= ws
        w[0] = np.sqrt(1 + w[1:] @ w[1:])
        self.w = w
        g = 0.5 * ws
        lam = np.empty(self.n)
        lam[0] = g
        lam[1:] = (((g + z[0] / zs) / ss) * s[1:] + ((g + s[0] / ss) / zs) * z[1:]) / (s[0] / ss + z[0] / zs + 2 * g)
        self.lam = lam * np.sqrt(ss * zs)
        return True

    def _W(self, x):                                     # mul_W!, :302-322
        zeta = self.w[1:] @ x[1:]
        c = x[0] + zeta / (1 + self.w[0])
        y = np.empty(self.n)
        y[0] = self.eta * (self.w[0] * x[0] + zeta)
        y[1:] = self.eta * (x[1:] + c * self.w[1:])
        return y

    def _Winv(self, x):                                  # mul_Winv!, :325-347
        zeta = self.w[1:] @ x[1:]
        c = -x[0] + zeta / (1 + self.w[0])
        y = np.empty(self.n)
        y[0] = (self.w[0] * x[0] - zeta) / self.eta
        y[1:] = (x[1:] + c * self.w[1:]) / self.eta
        return y

    @staticmethod
    def _circ(y, z):
        x = np.empty_like(y)
        x[0] = y @ z
        x[1:] = y[0] * z[1:] + z[0] * y[1:]
        return x

    def affine_ds(self, s):
        return self._circ(self.lam, self.lam)

    def combined_ds_shift(self, dz, ds, sigma_mu):
        out = self._circ(self._Winv(ds), self._W(dz))
        out[0] -= sigma_mu
        return out

    def ds_from_dz_offset(self, ds, z):                  # :243-268
        resz = _soc_res(z)
        l1 = self.lam[1:] @ ds[1:]
        w1 = self.w[1:] @ ds[1:]
        out = -z.copy()
        out[0] = z[0]
        out *= (self.lam[0] * ds[0] - l1) / resz
        out[0] += self.eta * w1
        out[1:] += self.eta * (ds[1:] + w1 / (1 + self.w[0]) * self.w[1:])
        return out / self.lam[0]

    def mul_Hs(self, x):
        c = 2 * (self.w @ x)
        y = x.copy()
        y[0] = -x[0]
        return (y + c * self.w) * self.eta ** 2

    @property
    def sparse(self):                                    # SOC_NO_EXPANSION_MAX_SIZE = 4 (cone_types.jl)
        return self.n > 4

    def sparse_data(self):                               # coneops_socone.jl:126-150: (u, v, d) of the rank-2 form of W'W
        w = self.w
        wsq = w[0] * w[0] + w[1:] @ w[1:]
        wsqinv = 1.0 / wsq
        d = wsqinv / 2
        u0 = np.sqrt(wsq - d)
        u1 = 2 * w[0] / u0
        v1 = np.sqrt(2 * (2 + wsqinv) / (2 * wsq - wsqinv))
        u, v = u1 * w, v1 * w
        u[0], v[0] = u0, 0.0
        return u, v, d

    def get_Hs(self):                                    # coneops_socone.jl:154-186
        e2 = self.eta ** 2
        if self.sparse:
            out = np.full(self.n, e2)
            out[0] *= self.sparse_data()[2]
            return out
        w = self.w
        blk = [(np.sqrt(2.0) * w[0] - 1.0) * (np.sqrt(2.0) * w[0] + 1.0)]
        for col in range(1, self.n):
            for row in range(col + 1):
                blk.append(2 * w[row] * w[col] + (1.0 if row == col else 0.0))
        return np.array(blk) * e2

    @staticmethod
    def _step(x, y, amax):                               # :443-512
        if x[0] >= 0 and y[0] < 0:
            amax = min(amax, -x[0] / y[0])
        a = _soc_res(y)
        b = 2 * (x[0] * y[0] - x[1:] @ y[1:])
        c = max(0.0, _soc_res(x))
        d = b * b - 4 * a * c
        if (a > 0 and b > 0) or d < 0:
            return amax
        if a == 0:
            return amax
        if c == 0:
            return amax if a >= 0 else 0.0
        t = (-b - np.sqrt(d)) if b >= 0 else (-b + np.sqrt(d))
        r1, r2 = (2 * c) / t, t / (2 * a)
        big = np.finfo(float).max
        r1 = big if r1 < 0 else r1
        r2 = big if r2 < 0 else r2
        return min(amax, r1, r2)

    def step_length(self, dz, ds, z, s, amax):
        return min(self._step(z, dz, amax), self._step(s, ds, amax))


def _svec_to_mat(x, k):                                  # coneops_psdtrianglecone.jl:469-483
    M = np.zeros((k, k))
    iu = np.triu_indices(k)
    # svec runs down the columns of the upper triangle: (row, col) with row <= col, column-major
    order = np.lexsort((iu[0], iu[1]))
    r, c = iu[0][order], iu[1][order]
    v = np.where(r == c, x, x / np.sqrt(2.0))
    M[r, c] = v
    M[c, r] = v
    return M


def _mat_to_svec(M):                                     # :486-497
    k = M.shape[0]
    iu = np.triu_indices(k)
    order = np.lexsort((iu[0], iu[1]))
    r, c = iu[0][order], iu[1][order]
    return np.where(r == c, M[r, c], (M[r, c] + M[c, r]) / np.sqrt(2.0))


class _PSD(_Cone):
    def __init__(self, spec, off):
        super().__init__(spec, off)
        self.k = spec.dim
        self.diag = np.array([j * (j + 1) // 2 + j for j in range(self.k)], dtype=int)   # triangular_index - 1

    @property
    def degree(self):
        return self.k

    def margins(self, z):                                # :8-27
        if self.n == 0:
            return np.finfo(float).max, 0.0
        e = np.linalg.eigvalsh(_svec_to_mat(z, self.k))
        return float(e.min()), float(e[e > 0].sum())

    def unit_shift(self, z, a, primal):                  # :30-44
        z[self.diag] += a

    def update_scaling(self, s, z):                      # :78-143
        if self.n == 0:
            return True
        try:
            L1 = np.linalg.cholesky(_svec_to_mat(s, self.k))
            L2 = np.linalg.cholesky(_svec_to_mat(z, self.k))
        except np.linalg.LinAlgError:
            return False
        U, sv, Vt = np.linalg.svd(L2.T @ L1)
        self.lam = sv
        isq = 1.0 / np.sqrt(sv)
        self.R = (L1 @ Vt.T) * isq[None, :]
        self.Rinv = isq[:, None] * (U.T @ L2.T)
        return True

    def _W(self, x):                                     # mul_W!(:N): R' X R
        return _mat_to_svec(self.R.T @ _svec_to_mat(x, self.k) @ self.R)

    def _Wt(self, x):                                    # mul_W!(:T): R X R'
        return _mat_to_svec(self.R @ _svec_to_mat(x, self.k) @ self.R.T)

    def _WinvT(self, x):                                 # mul_Winv!(:T): Rinv X Rinv'
        return _mat_to_svec(self.Rinv @ _svec_to_mat(x, self.k) @ self.Rinv.T)

    def affine_ds(self, s):                              # :189-204
        out = np.zeros(self.n)
        out[self.diag] = self.lam ** 2
        return out

    def combined_ds_shift(self, dz, ds, sigma_mu):       # coneops_symmetric_common.jl:2-36, circ_op! :361-382
        Y = _svec_to_mat(self._WinvT(ds), self.k)
        Z = _svec_to_mat(self._W(dz), self.k)
        out = _mat_to_svec((Y @ Z + Z @ Y) / 2)
        out[self.diag] -= sigma_mu
        return out

    def ds_from_dz_offset(self, ds, z):                  # :218-228, lambda_inv_circ_op! :335-353
        X = _svec_to_mat(ds, self.k)
        X = 2.0 * X / (self.lam[:, None] + self.lam[None, :])
        return self._Wt(_mat_to_svec(X))

    def mul_Hs(self, x):                                 # :164-187
        return self._Wt(self._W(x))

    def get_Hs(self):                                    # :146-162: packed upper triangle of (R R') (x)_s (R R')
        t = self.n
        M = np.empty((t, t))
        for e in range(t):
            unit = np.zeros(t)
            unit[e] = 1.0
            M[:, e] = self.mul_Hs(unit)
        return np.concatenate([M[:col + 1, col] for col in range(t)]) if t else np.zeros(0)

    def _step_component(self, d, amax):                  # :439-466
        if self.n == 0:
            return amax
        isq = 1.0 / np.sqrt(self.lam)
        M = _svec_to_mat(d, self.k) * isq[:, None] * isq[None, :]
        g = float(np.linalg.eigvalsh(M).min())
        return min(1.0 / -g, amax) if g < 0 else amax

    def step_length(self, dz, ds, z, s, amax):           # :230-254
        return min(self._step_component(self._W(dz), amax), self._step_component(self._WinvT(ds), amax))


def adopt_device_scaling(cones, dev_scaling):
    """Give the host PSD cone objects the (R, Rinv, lambda) triples `HipKKTSolver.scaling()` returns."""
    it = iter(dev_scaling)
    for c in cones:
        if isinstance(c, _PSD):
            c.R, c.Rinv, c.lam = next(it)


def host_cone_data(cones):
    """What the Julia glue reads from the reference's cone objects for kkt_update! (hipkkt_kkt_system_update_cones):
    get_Hs! blocks, the sparse second-order cones' (u, v, eta^2), and the NT scaling w (m), eta (per cone), lambda (m),
    R / Rinv of the PSD cones (column-major, concatenated).  After update_scaling on every cone."""
    m = cones[-1].off + cones[-1].n if cones else 0
    Hs, u, v, e2, R, Ri = [], [], [], [], [], []
    w, lam, eta = np.ones(m), np.zeros(m), np.ones(len(cones))
    for i, c in enumerate(cones):
        Hs.append(c.get_Hs())
        if isinstance(c, (_NN, _SOC)):
            w[c.rng] = c.w
            lam[c.rng] = c.lam
        if isinstance(c, _SOC):
            eta[i] = c.eta
            if c.sparse:
                uu, vv, _ = c.sparse_data()
                u.append(uu); v.append(vv); e2.append(c.eta ** 2)
        if isinstance(c, _PSD) and c.n:
            lam[c.off:c.off + c.k] = c.lam
            R.append(np.asarray(c.R).ravel(order="F")); Ri.append(np.asarray(c.Rinv).ravel(order="F"))
    cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0)
    return cat(Hs), cat(u), cat(v), np.array(e2), w, eta, lam, cat(R), cat(Ri)


def _make_cones(specs):
    out, off = [], 0
    for c in specs:
        if isinstance(c, ZeroConeT):
            out.append(_Zero(c, off))
        elif isinstance(c, NonnegativeConeT):
            out.append(_NN(c, off))
        elif isinstance(c, SecondOrderConeT):
            out.append(_SOC(c, off))
        elif isinstance(c, PSDTriangleConeT):
            out.append(_PSD(c, off))
        else:
            raise NotImplementedError("the IPM test driver covers Zero, Nonnegative, SecondOrder and PSDTriangle cones")
        off += c.numel
    return out


def identity_scaling_data(specs):
    """What get_Hs! and the sparse SOC data look like under set_identity_scaling!
    (coneops_*cone.jl set_identity_scaling!): Hsblocks, soc_u, soc_v, soc_eta2."""
    Hs, u, v, e2 = [], [], [], []
    for c in specs:
        if isinstance(c, ZeroConeT):
            Hs.append(np.zeros(c.dim))
        elif isinstance(c, NonnegativeConeT):
            Hs.append(np.ones(c.dim))
        elif isinstance(c, SecondOrderConeT):
            if c.dim > 4:
                d = np.ones(c.dim)
                d[0] = 0.5
                Hs.append(d)
                uu = np.zeros(c.dim)
                uu[0] = np.sqrt(0.5)
                u.append(uu)
                v.append(np.zeros(c.dim))
                e2.append(1.0)
            else:
                # packed triu of eta^2 (2 w w' - J) at w = e_1, eta = 1, with the reference's
                # cancellation-free first entry (coneops_socone.jl:168-186)
                blk = []
                for col in range(c.dim):
                    for row in range(col + 1):
                        blk.append(1.0 if row == col else 0.0)
                blk[0] = (np.sqrt(2.0) * 1.0 - 1.0) * (np.sqrt(2.0) * 1.0 + 1.0)
                Hs.append(np.array(blk))
        elif isinstance(c, PSDTriangleConeT):
            # Hs = I (coneops_psdtrianglecone.jl:65-75), packed upper triangle of the t x t identity
            t = c.numel
            blk = np.zeros(t * (t + 1) // 2)
            blk[[j * (j + 1) // 2 + j for j in range(t)]] = 1.0
            Hs.append(blk)
        else:
            raise NotImplementedError
    cat = lambda parts: np.concatenate(parts) if parts else np.zeros(0)
    return cat(Hs), cat(u), cat(v), np.array(e2)


# ------------------------------------------------------------------------------------------
#  the loop
# ------------------------------------------------------------------------------------------
def solve(P, q, A, b, cone_specs, backend, settings=None):
    """Clarabel.solve! restated.  `backend` must offer
         update_identity()            -> bool     (kkt_update! under set_identity_scaling!)
         update(s, z)                 -> bool     (kktsolver_update! after update_scaling!(s,z))
         kktsolver_setrhs(rx, rz); kktsolver_solve(x_out, z_out) -> bool
         last_ir_iterations
    """
    st = settings or IPMSettings()
    P = sp.csc_matrix(P)
    Pt = sp.triu(P, format="csc")
    Pfull = (Pt + sp.triu(Pt, 1).T).tocsr()
    A = sp.csr_matrix(A)
    At = A.T.tocsr()
    q, b = np.asarray(q, float), np.asarray(b, float)
    n, m = Pfull.shape[0], A.shape[0]
    cones = _make_cones(cone_specs)
    degree = sum(c.degree for c in cones)
    normq = np.abs(q).max() if n else 0.0
    normb = np.abs(b).max() if m else 0.0
    ir_total = 0

    def each(fn, *vecs):
        out = np.empty(m)
        for c in cones:
            out[c.rng] = fn(c, *[v[c.rng] for v in vecs])
        return out

    def ksolve(rx, rz, want_x=True, want_z=True):
        nonlocal ir_total
        backend.kktsolver_setrhs(rx, rz)
        xo, zo = np.zeros(n), np.zeros(m)
        ok = backend.kktsolver_solve(xo if want_x else None, zo if want_z else None)
        ir_total += backend.last_ir_iterations
        return ok, xo, zo

    # ---- default start (symmetric cones): solver.jl:383-404
    x = np.zeros(n); s = np.zeros(m); z = np.zeros(m)
    system = getattr(backend, "system", None)          # device-resident DefaultKKTSystem (level C), if the backend has one
    if system is not None:
        system.init(q, b)
    ok = backend.update_identity()
    if system is not None:
        ok_c = system.solve_constant_rhs()
        ok1, x, s, z = system.solve_initial_point()
        ok2 = True
        x2 = z2 = None
    else:
        ok_c, x2, z2 = ksolve(-q, b)                   # kkt_update! also solves the constant RHS
    if system is not None:
        pass
    elif Pt.nnz == 0:                                  # kktsystem.jl:101-120 (LP initialisation)
        ok1, x, s = ksolve(np.zeros(n), b)
        s = -s
        ok2, _, z = ksolve(-q, np.zeros(m), want_x=False)
    else:                                              # :121-129 (QP initialisation)
        ok1, x, z = ksolve(-q, b)
        s = -z.copy()
        ok2 = True

    def shift_to_interior(v, primal):                  # variables.jl:213-237
        mins, pos = np.finfo(float).max, 0.0
        for c in cones:
            a, bb = c.margins(v[c.rng])
            mins, pos = min(mins, a), pos + bb
        target = max(1.0, 0.1 * pos / max(degree, 1))
        shifts = []
        if mins <= 0:
            shifts = [-mins, target]
        elif mins < target:
            shifts = [target - mins]
        else:
            shifts = [0.0]
        for a in shifts:
            for c in cones:
                c.unit_shift(v[c.rng], a, primal)

    shift_to_interior(s, True)
    shift_to_interior(z, False)
    tau, kappa = 1.0, 1.0

    it, alpha, sigma = 0, 0.0, 1.0
    status = UNSOLVED
    prev = None
    hist = []
    prev_vars = None
    while True:
        # ---- residuals (residuals.jl:1-37)
        qx, bz, sz = q @ x, b @ z, s @ z
        Px = Pfull @ x
        xPx = x @ Px
        rx_inf = -(At @ z)
        rz_inf = A @ x + s
        rx = rx_inf - Px - q * tau
        rz = rz_inf - b * tau
        rtau = qx + bz + kappa + xPx / tau
        mu = (sz + tau * kappa) / (degree + 1)
        # ---- info_update! (info.jl:1-63), no equilibration
        tinv = 1.0 / tau
        cost_p = qx * tinv + xPx * tinv * tinv / 2
        cost_d = -bz * tinv - xPx * tinv * tinv / 2
        nx, nz, ns = np.linalg.norm(x), np.linalg.norm(z), np.linalg.norm(s)
        res_pinf = np.linalg.norm(rx_inf) / max(1.0, nz)
        res_dinf = max(np.linalg.norm(Px) / max(1.0, nx), np.linalg.norm(rz_inf) / max(1.0, nx + ns))
        nx, nz, ns = nx * tinv, nz * tinv, ns * tinv
        res_p = np.linalg.norm(rz) * tinv / max(1.0, normb + nx + ns)
        res_d = np.linalg.norm(rx) * tinv / max(1.0, normq + nx + nz)
        gap_abs = abs(cost_p - cost_d)
        gap_rel = gap_abs / max(1.0, min(abs(cost_p), abs(cost_d)))
        kt = kappa * tinv
        hist.append(dict(iter=it, pcost=cost_p, dcost=cost_d, gap=gap_abs, pres=res_p, dres=res_d, kt=kt, mu=mu,
                         step=alpha))
        # ---- termination (info.jl:65-120, 270-330)
        status = UNSOLVED
        if kt <= 1 and (gap_abs < st.tol_gap_abs or gap_rel < st.tol_gap_rel) and res_p < st.tol_feas and res_d < st.tol_feas:
            status = SOLVED
        elif kt > 1000.0 / st.tol_ktratio:
            if bz < -st.tol_infeas_abs and res_pinf < -st.tol_infeas_rel * bz:
                status = PRIMAL_INFEASIBLE
            elif qx < -st.tol_infeas_abs and res_dinf < -st.tol_infeas_rel * qx:
                status = DUAL_INFEASIBLE
        if status == UNSOLVED and it > 1 and prev is not None and (res_d > prev["res_d"] or res_p > prev["res_p"]):
            if kt < 100 * np.finfo(float).eps and (prev["gap_abs"] < st.tol_gap_abs or prev["gap_rel"] < st.tol_gap_rel):
                status = INSUFFICIENT_PROGRESS
            if kt < 1 and ((res_d > 100 * st.tol_feas and res_d > 100 * prev["res_d"]) or
                           (res_p > 100 * st.tol_feas and res_p > 100 * prev["res_p"])):
                status = INSUFFICIENT_PROGRESS
        if status == UNSOLVED and it == st.max_iter:
            status = MAX_ITERATIONS
        if status != UNSOLVED:
            if status == INSUFFICIENT_PROGRESS and prev_vars is not None:
                x, s, z, tau, kappa = prev_vars
            break
        # ---- scale cones, KKT update + constant-RHS solve (solver.jl:258-280, kktsystem.jl:62-92)
        if not all(c.update_scaling(s[c.rng].copy(), z[c.rng].copy()) for c in cones):
            status = NUMERICAL_ERROR
            break
        it += 1
        aff_step = None
        if system is not None and getattr(backend, "batch_affine", False):
            # kkt_update! and the affine kkt_solve! as ONE call: the constant and the affine right-hand side do not
            # depend on each other (kktsystem.jl:87-88 vs :170-171; the affine step does not read rhs.s, :157-158),
            # so their solves share every triangular sweep
            ok, aff_step = system.update_and_solve_affine(rx, rz, rtau, tau * kappa, x, s, z, tau, kappa)
            ir_total += backend.last_ir_iterations
            if ok and any(isinstance(c, _PSD) and c.n for c in cones):
                adopt_device_scaling(cones, backend.ks.scaling()[1])
        elif system is not None and getattr(backend, "host_cones", False):
            # kkt_update!(kktsystem, data, cones) as the Julia glue issues it: everything from the caller's cone objects
            ok = system.update_cones(*host_cone_data(cones))
        elif system is not None:
            ok = system.update(s, z)                   # kkt_update!: scaling, refactor, constant-RHS solve
            # The scaled space of a PSD cone is fixed only up to the signs of the singular vectors of L2'L1.  With the
            # reduced system on the device, ITS scaling is the one the right-hand sides must be expressed in (in the
            # reference one cone object serves both sides): adopt the device's R, Rinv, lambda.
            if ok and any(isinstance(c, _PSD) and c.n for c in cones):
                adopt_device_scaling(cones, backend.ks.scaling()[1])
        else:
            ok = backend.update(s, z)
            if ok:
                ok, x2, z2 = ksolve(-q, b)

        def kkt_solve(rhs_x, rhs_z, rhs_s, rhs_tau, rhs_kappa, affine, lhs_z_work=None):   # kktsystem.jl:135-215
            nonlocal ir_total
            if system is not None:
                okk, out = system.solve(rhs_x, rhs_s, rhs_z, rhs_tau, rhs_kappa, x, s, z, tau, kappa, affine)
                ir_total += backend.last_ir_iterations
                return okk, out
            if affine:
                const = s.copy()
            else:
                const = each(lambda c, ds, zz: c.ds_from_dz_offset(ds, zz), rhs_s, z)
            okk, x1, z1 = ksolve(rhs_x, const - rhs_z)
            if not okk:
                return False, None
            xi = x / tau
            tnum = rhs_tau - rhs_kappa / tau + q @ x1 + b @ z1 + 2 * (xi @ (Pfull @ x1))
            xm = xi - x2
            tden = kappa / tau - q @ x2 - b @ z2 + xm @ (Pfull @ xm) - x2 @ (Pfull @ x2)
            dtau = tnum / tden
            dx = x1 + dtau * x2
            dz = z1 + dtau * z2
            ds = -(each(lambda c, v: c.mul_Hs(v), dz) + const)
            dkappa = -(rhs_kappa + kappa * dtau) / tau
            return True, (dx, dz, ds, dtau, dkappa)

        def step_length(dz, ds, dtau, dkappa, combined):                                   # variables.jl:13-45
            at = -tau / dtau if dtau < 0 else np.finfo(float).max
            ak = -kappa / dkappa if dkappa < 0 else np.finfo(float).max
            a = min(at, ak, 1.0)
            for c in cones:
                a = min(a, c.step_length(dz[c.rng], ds[c.rng], z[c.rng], s[c.rng], a))
            return a * st.max_step_fraction if combined else a

        step = None
        if ok:
            aff_s = each(lambda c, v: c.affine_ds(v), s)
            if aff_step is not None:
                step = aff_step
            else:
                ok, step = kkt_solve(rx, rz, aff_s, rtau, tau * kappa, True)
        if ok:
            dx, dz, ds, dtau, dkappa = step
            alpha = step_length(dz, ds, dtau, dkappa, False)
            sigma = (1 - alpha) ** 3
            mcorr = 1.0 if it > 1 else alpha
            shift = each(lambda c, a_, b_: c.combined_ds_shift(a_, b_, sigma * mu), dz * mcorr, ds)
            rhs_s = aff_s + shift
            ok, step = kkt_solve((1 - sigma) * rx, (1 - sigma) * rz, rhs_s, (1 - sigma) * rtau,
                                 -sigma * mu + mcorr * dtau * dkappa + tau * kappa, False)
        if not ok:
            status = NUMERICAL_ERROR
            alpha = 0.0
            break
        dx, dz, ds, dtau, dkappa = step
        alpha = step_length(dz, ds, dtau, dkappa, True)
        if alpha <= max(0.0, st.min_terminate_step_length):
            status = INSUFFICIENT_PROGRESS
            alpha = 0.0
            break
        prev = dict(res_p=res_p, res_d=res_d, gap_abs=gap_abs, gap_rel=gap_rel)
        prev_vars = (x.copy(), s.copy(), z.copy(), tau, kappa)
        x = x + alpha * dx
        s = s + alpha * ds
        z = z + alpha * dz
        tau += alpha * dtau
        kappa += alpha * dkappa

    # ---- info_post_process! (info.jl:196-211): after an error / limit exit, accept an iterate that
    #      meets the reduced tolerances as ALMOST_SOLVED
    if status in (NUMERICAL_ERROR, INSUFFICIENT_PROGRESS, MAX_ITERATIONS):
        tinv = 1.0 / tau
        Px = Pfull @ x
        xPx = x @ Px
        cp = (q @ x) * tinv + xPx * tinv * tinv / 2
        cd = -(b @ z) * tinv - xPx * tinv * tinv / 2
        nx, nz, ns = np.linalg.norm(x) * tinv, np.linalg.norm(z) * tinv, np.linalg.norm(s) * tinv
        rp = np.linalg.norm(A @ x + s - b * tau) * tinv / max(1.0, normb + nx + ns)
        rd = np.linalg.norm(-(At @ z) - Px - q * tau) * tinv / max(1.0, normq + nx + nz)
        ga = abs(cp - cd)
        gr = ga / max(1.0, min(abs(cp), abs(cd)))
        if kappa * tinv <= 1 and (ga < st.reduced_tol_gap_abs or gr < st.reduced_tol_gap_rel) and \
                rp < st.reduced_tol_feas and rd < st.reduced_tol_feas:
            status = ALMOST_SOLVED
    # ---- solution_post_process!: unscale by tau (kappa for certificates)
    infeasible = status in (PRIMAL_INFEASIBLE, DUAL_INFEASIBLE)
    sc = 1.0 / (kappa if infeasible else tau)
    xo, zo, so = x * sc, z * sc, s * sc
    objp = q @ xo + 0.5 * xo @ (Pfull @ xo)
    objd = -b @ zo - 0.5 * xo @ (Pfull @ xo)
    if infeasible:
        objp = objd = float("nan")
    return IPMResult(status, xo, zo, so, objp, objd, it, ir_total, hist)


# ------------------------------------------------------------------------------------------
#  backends
# ------------------------------------------------------------------------------------------
class HipBackend:
    """The MI355X path: libhipkkt.so through HipKKTSolver (level B of the C ABI)."""

    def __init__(self, P, A, cone_specs, settings=None):
        from .kktsolver import HipKKTSolver
        self.ks = HipKKTSolver(P, A, cone_specs, settings=settings)
        self.specs = list(cone_specs)

    def update_identity(self):
        return self.ks.kktsolver_update(*identity_scaling_data(self.specs))

    def update(self, s, z):
        return self.ks.kktsolver_update_from_sz(s, z)

    def kktsolver_setrhs(self, rx, rz):
        self.ks.kktsolver_setrhs(rx, rz)

    def kktsolver_solve(self, x, z):
        return self.ks.kktsolver_solve(x, z)

    @property
    def last_ir_iterations(self):
        return self.ks.last_ir_iterations


class HipSystemBackend(HipBackend):
    """As HipBackend, but the reduced-system layer (kktsystem.jl) runs on the device too (level C of the
    C ABI): the driver hands over iterates and right-hand sides and gets the step back."""

    def __init__(self, P, A, cone_specs, settings=None, batch_affine=False, lazy=False, host_cones=False, staging="host"):
        super().__init__(P, A, cone_specs, settings=settings)
        from .kktsolver import HipKKTSystem
        self.system = HipKKTSystem(self.ks)
        self.system.staging = staging              # "host": the *_host entry points; "torch": device tensors + *_dev
        self.batch_affine = batch_affine           # kkt_update! + affine kkt_solve! as one 2-column solve, ONE call
        self.host_cones = host_cones               # kkt_update! from the driver's own cone objects (the Julia glue's route)
        if lazy:
            # the same pairing through the reference's TWO calls (solver.jl:278-295 untouched): kkt_update! leaves the
            # constant-RHS solve to the affine kkt_solve!, which sends both right-hand sides through the sweeps together
            self.system.set_lazy(True)
