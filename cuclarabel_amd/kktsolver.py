"""Host-side mirror of the reference's KKT solver objects over the C ABI.

`HipKKTSolver` has the methods of `DirectLDLKKTSolver <: AbstractKKTSolver`
(`/root/reference/src/kktsolvers/kktsolver_directldl.jl:46-92,197-386`, contract
`kktsolver_defaults.jl:2-48`) and `HipDirectLDLSolver` those of an
`AbstractDirectLDLSolver` backend (`direct-ldl/directldl_defaults.jl:1-72`, example
`directldl_qdldl.jl`).  Method names drop Julia's `!`; Bool returns keep the reference's
meaning (True = success, False = numeric failure; usage errors raise).
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import check, f64, i64, ptr
from .cones import cone_kinds_dims


class LinearSolverInfo:
    """types.jl:198-206"""

    def __init__(self, name, threads, direct, nnzA, nnzL):
        self.name, self.threads, self.direct, self.nnzA, self.nnzL = name, threads, direct, nnzA, nnzL

    def __repr__(self):
        return (f"LinearSolverInfo(name={self.name!r}, threads={self.threads}, direct={self.direct}, "
                f"nnzA={self.nnzA}, nnzL={self.nnzL})")


class HipKKTSolver:
    """DirectLDLKKTSolver on the MI355X: KKT assembly maps, value scatter, regularisation,
    numeric LDL', triangular solves and iterative refinement all run on the device."""

    def __init__(self, P, A, cones, m=None, n=None, settings=None):
        L = _lib.lib()
        P = sp.triu(sp.csc_matrix(P), format="csc")
        P.sort_indices()
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.n = P.shape[0] if n is None else n
        self.m = A.shape[0] if m is None else m
        if P.shape != (self.n, self.n) or A.shape != (self.m, self.n):
            raise ValueError("P must be n x n and A m x n")
        self.cones = list(cones)
        kinds, dims = cone_kinds_dims(self.cones)
        self.settings = settings if settings is not None else _lib.default_settings()
        self._h = C.c_void_p()
        Pp, Pi, Px = i64(P.indptr), i64(P.indices), f64(P.data)
        Ap, Ai, Ax = i64(A.indptr), i64(A.indices), f64(A.data)
        rc = L.hipkkt_kkt_create(C.byref(self._h), self.n, self.m, ptr(Pp), ptr(Pi), ptr(Px),
                                 ptr(Ap), ptr(Ai), ptr(Ax), len(self.cones), ptr(kinds), ptr(dims),
                                 C.byref(self.settings), 0)
        if not check(rc, "hipkkt_kkt_create"):
            raise _lib.HipKKTError("hipkkt_kkt_create reported a numeric failure")
        self._nnzP, self._nnzA = P.nnz, A.nnz
        info = _lib.Info()
        check(L.hipkkt_kkt_info(self._h, C.byref(info)), "hipkkt_kkt_info")
        self.info = info.as_dict()
        self.p = info.p
        self.N = info.N

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.lib().hipkkt_kkt_destroy(h)
            self._h = None

    # ---- AbstractKKTSolver interface (kktsolver_defaults.jl:2-48)
    def kktsolver_update(self, Hsblocks, soc_u=None, soc_v=None, soc_eta2=None):
        """kktsolver_update!(kktsolver, cones) with the cone data flattened the way the glue does:
        get_Hs! output plus each sparse second-order cone's (u, v, eta^2)."""
        Hs = f64(Hsblocks)
        u = f64(soc_u if soc_u is not None else [])
        v = f64(soc_v if soc_v is not None else [])
        e2 = f64(soc_eta2 if soc_eta2 is not None else [])
        if Hs.size != self.info["nHs"] or u.size != self.info["sparse_soc_len"] or \
                v.size != u.size or e2.size != self.info["nsparse_soc"]:
            raise ValueError("cone data has the wrong length")
        return check(_lib.lib().hipkkt_kkt_update_cones(self._h, ptr(Hs), ptr(u), ptr(v), ptr(e2)),
                     "hipkkt_kkt_update_cones")

    def kktsolver_update_from_sz(self, s, z):
        """Device-native variant: update_scaling! + get_Hs! run on the GPU from (s, z)."""
        s, z = f64(s), f64(z)
        if s.size != self.m or z.size != self.m:
            raise ValueError("s, z must have length m")
        return check(_lib.lib().hipkkt_kkt_update_from_sz(self._h, ptr(s), ptr(z)),
                     "hipkkt_kkt_update_from_sz")

    def kktsolver_update_from_sz_dev(self, d_s, d_z):
        return check(_lib.lib().hipkkt_kkt_update_from_sz_dev(self._h, C.c_void_p(d_s), C.c_void_p(d_z)),
                     "hipkkt_kkt_update_from_sz_dev")

    def kktsolver_setrhs(self, rhsx, rhsz):
        rx, rz = f64(rhsx), f64(rhsz)
        if rx.size != self.n or rz.size != self.m:
            raise ValueError("rhs has the wrong length")
        check(_lib.lib().hipkkt_kkt_setrhs(self._h, ptr(rx), ptr(rz)), "hipkkt_kkt_setrhs")

    def kktsolver_solve(self, lhsx=None, lhsz=None):
        """kktsolver_solve!(kktsolver, lhsx, lhsz): writes into the given arrays (either may be
        None, as in the reference) and returns is_success."""
        for a, k in ((lhsx, self.n), (lhsz, self.m)):
            if a is not None and (a.dtype != np.float64 or a.size != k or not a.flags.c_contiguous):
                raise ValueError("lhs arrays must be contiguous float64 of the right length")
        return check(_lib.lib().hipkkt_kkt_solve(self._h, ptr(lhsx), ptr(lhsz)), "hipkkt_kkt_solve")

    def kktsolver_setrhs_dev(self, d_rx, d_rz):
        check(_lib.lib().hipkkt_kkt_setrhs_dev(self._h, C.c_void_p(d_rx), C.c_void_p(d_rz)), "hipkkt_kkt_setrhs_dev")

    def kktsolver_solve_dev(self, d_lhsx, d_lhsz):
        return check(_lib.lib().hipkkt_kkt_solve_dev(self._h, C.c_void_p(d_lhsx) if d_lhsx else None,
                                                     C.c_void_p(d_lhsz) if d_lhsz else None),
                     "hipkkt_kkt_solve_dev")

    def set_deferred_status(self, on=True):
        """The *_dev entry points only enqueue; `deferred_status()` reports once for everything since the last query."""
        check(_lib.lib().hipkkt_kkt_set_deferred_status(self._h, int(on)), "hipkkt_kkt_set_deferred_status")

    def deferred_status(self):
        """0 = every call since the last query succeeded, 1 = numeric failure, 2 = some solve's refinement was cut short."""
        rc = _lib.lib().hipkkt_kkt_deferred_status(self._h)
        if rc < 0:
            check(rc, "hipkkt_kkt_deferred_status")
        return rc

    def kktsolver_solve_multi(self, rhsx, rhsz, want_x=True, want_z=True):
        """setrhs! + solve! for several right-hand sides against the current factorisation.
        rhsx: (n, k), rhsz: (m, k).  Returns (is_success, lhsx (n, k) | None, lhsz (m, k) | None,
        refinement rounds per column)."""
        RX = np.asfortranarray(rhsx, dtype=np.float64)
        RZ = np.asfortranarray(rhsz, dtype=np.float64)
        if RX.ndim != 2 or RZ.ndim != 2 or RX.shape[0] != self.n or RZ.shape[0] != self.m or RX.shape[1] != RZ.shape[1]:
            raise ValueError("rhsx must be (n, k) and rhsz (m, k)")
        k = RX.shape[1]
        LX = np.zeros((self.n, k), order="F") if want_x else None
        LZ = np.zeros((self.m, k), order="F") if want_z else None
        ir = np.zeros(max(k, 1), dtype=np.int64)
        ok = check(_lib.lib().hipkkt_kkt_solve_multi(self._h, k, ptr(RX), ptr(RZ), ptr(LX), ptr(LZ), ptr(ir)),
                   "hipkkt_kkt_solve_multi")
        return ok, LX, LZ, ir[:k]

    def kktsolver_solve_multi_dev(self, k, d_rx, d_rz, d_lhsx, d_lhsz):
        """Device-pointer variant (column-major, contiguous).  Returns (is_success, rounds per column)."""
        ir = np.zeros(max(k, 1), dtype=np.int64)
        ok = check(_lib.lib().hipkkt_kkt_solve_multi_dev(self._h, k, C.c_void_p(d_rx), C.c_void_p(d_rz),
                                                         C.c_void_p(d_lhsx) if d_lhsx else None,
                                                         C.c_void_p(d_lhsz) if d_lhsz else None, ptr(ir)),
                   "hipkkt_kkt_solve_multi_dev")
        return ok, ir[:k]

    def kktsolver_update_P(self, P):
        Px = f64(P.data if sp.issparse(P) else P)
        if Px.size != self._nnzP:
            raise ValueError("P.nzval has the wrong length")
        check(_lib.lib().hipkkt_kkt_update_P(self._h, ptr(Px)), "hipkkt_kkt_update_P")

    def kktsolver_update_A(self, A):
        Ax = f64(A.data if sp.issparse(A) else A)
        if Ax.size != self._nnzA:
            raise ValueError("A.nzval has the wrong length")
        check(_lib.lib().hipkkt_kkt_update_A(self._h, ptr(Ax)), "hipkkt_kkt_update_A")

    def kktsolver_linear_solver_info(self):
        return LinearSolverInfo("hipldl", 1, True, self.info["nnzK"], self.info["nnzL"])

    # ---- extras
    def mul_Hs(self, x):
        x = f64(x)
        y = np.zeros(self.m)
        check(_lib.lib().hipkkt_kkt_mul_Hs(self._h, ptr(y), ptr(x)), "hipkkt_kkt_mul_Hs")
        return y

    def get_Hs(self):
        out = np.zeros(max(self.info["nHs"], 1))
        check(_lib.lib().hipkkt_kkt_get_Hs(self._h, ptr(out)), "hipkkt_kkt_get_Hs")
        return out[:self.info["nHs"]]

    def KKT(self):
        """The assembled triu CSC KKT matrix with its current (un-regularised) values."""
        colptr = np.zeros(self.N + 1, dtype=np.int64)
        rowval = np.zeros(self.info["nnzK"], dtype=np.int64)
        nzval = np.zeros(self.info["nnzK"])
        L = _lib.lib()
        check(L.hipkkt_kkt_get_pattern(self._h, ptr(colptr), ptr(rowval)), "hipkkt_kkt_get_pattern")
        check(L.hipkkt_kkt_get_values(self._h, ptr(nzval)), "hipkkt_kkt_get_values")
        return sp.csc_matrix((nzval, rowval, colptr), shape=(self.N, self.N))

    def maps(self):
        i = self.info
        out = dict(P=np.zeros(self._nnzP, np.int64), A=np.zeros(self._nnzA, np.int64),
                   Hsblocks=np.zeros(i["nHs"], np.int64), diag_full=np.zeros(self.N, np.int64),
                   soc_u=np.zeros(i["sparse_soc_len"], np.int64), soc_v=np.zeros(i["sparse_soc_len"], np.int64),
                   soc_D=np.zeros(2 * i["nsparse_soc"], np.int64), dsigns=np.zeros(self.N, np.int64))
        check(_lib.lib().hipkkt_kkt_get_maps(self._h, *[ptr(out[k]) for k in
              ("P", "A", "Hsblocks", "diag_full", "soc_u", "soc_v", "soc_D", "dsigns")]), "hipkkt_kkt_get_maps")
        return out

    def perm(self):
        out = np.zeros(self.N, dtype=np.int64)
        check(_lib.lib().hipkkt_kkt_get_perm(self._h, ptr(out)), "hipkkt_kkt_get_perm")
        return out

    def scaling(self):
        """The NT scaling on the device: (lambda (m), [(R, Rinv, lam) per PSD cone])."""
        ks = [c.dim for c in self.cones if c.kind == 3]
        tot = sum(k * k for k in ks)
        lam, R, Ri = np.zeros(max(self.m, 1)), np.zeros(max(tot, 1)), np.zeros(max(tot, 1))
        check(_lib.lib().hipkkt_kkt_get_scaling(self._h, ptr(lam), ptr(R), ptr(Ri)), "hipkkt_kkt_get_scaling")
        out, o, off = [], 0, 0
        for c in self.cones:
            if c.kind == 3:
                k = c.dim
                out.append((R[o:o + k * k].reshape(k, k, order="F").copy(), Ri[o:o + k * k].reshape(k, k, order="F").copy(),
                            lam[off:off + k].copy()))
                o += k * k
            off += c.numel
        return lam[:self.m], out

    def scaling_w(self):
        """(w (m), eta (per cone)) of the device's NT scaling."""
        w, eta = np.zeros(max(self.m, 1)), np.zeros(max(len(self.cones), 1))
        check(_lib.lib().hipkkt_kkt_get_scaling_w(self._h, ptr(w), ptr(eta)), "hipkkt_kkt_get_scaling_w")
        return w[:self.m], eta[:len(self.cones)]

    @property
    def fallbacks(self):
        """(overlap-mode, persistent-sweep-kernel) fallbacks taken by this handle so far; expected (0, 0)."""
        p = self.profile()
        return int(p["overlap_fallbacks"]), int(p["top_fallbacks"])

    @property
    def diagonal_regularizer(self):
        return _lib.lib().hipkkt_kkt_last_regularizer(self._h)

    @property
    def last_ir_iterations(self):
        return int(_lib.lib().hipkkt_kkt_last_ir_iterations(self._h))

    def speculative_rounds(self, set=-1):
        """(tests) refinement rounds a solve enqueues ahead of its first status read-back; `set` >= 0 replaces it first"""
        return int(_lib.lib().hipkkt_kkt_speculative_rounds(self._h, int(set)))

    def set_stream(self, stream_ptr):
        check(_lib.lib().hipkkt_kkt_set_stream(self._h, C.c_void_p(stream_ptr)), "hipkkt_kkt_set_stream")

    def synchronize(self):
        check(_lib.lib().hipkkt_kkt_synchronize(self._h), "hipkkt_kkt_synchronize")

    def profile_enable(self, on=True):
        check(_lib.lib().hipkkt_kkt_profile_enable(self._h, int(on)), "hipkkt_kkt_profile_enable")

    def profile_reset(self):
        check(_lib.lib().hipkkt_kkt_profile_reset(self._h), "hipkkt_kkt_profile_reset")

    def profile(self):
        p = _lib.Profile()
        check(_lib.lib().hipkkt_kkt_profile_get(self._h, C.byref(p)), "hipkkt_kkt_profile_get")
        return p.as_dict()


class HipKKTSystem:
    """DefaultKKTSystem (`/root/reference/src/kktsystem.jl:21-215`) with its vectors in HBM (level C of
    the C ABI): `kkt_update!`, `kkt_solve_initial_point!`, `kkt_solve!`.  The `*_dev` methods take raw
    device pointers; the numpy-facing ones stage through torch tensors (plumbing for tests and the IPM
    test driver)."""

    def __init__(self, kktsolver):
        self.ks = kktsolver
        self._ready = False

    def init(self, q, b):
        q, b = f64(q), f64(b)
        if q.size != self.ks.n or b.size != self.ks.m:
            raise ValueError("q must have length n and b length m")
        check(_lib.lib().hipkkt_kkt_system_init(self.ks._h, ptr(q), ptr(b)), "hipkkt_kkt_system_init")
        self._ready = True

    # ---- device-pointer interface
    def update_dev(self, d_s, d_z):
        return check(_lib.lib().hipkkt_kkt_system_update(self.ks._h, C.c_void_p(d_s), C.c_void_p(d_z)),
                     "hipkkt_kkt_system_update")

    def solve_constant_rhs(self):
        return check(_lib.lib().hipkkt_kkt_system_solve_constant_rhs(self.ks._h), "hipkkt_kkt_system_solve_constant_rhs")

    def solve_initial_point_dev(self, d_x, d_s, d_z):
        return check(_lib.lib().hipkkt_kkt_system_solve_initial_point(self.ks._h, C.c_void_p(d_x), C.c_void_p(d_s),
                                                                       C.c_void_p(d_z)),
                     "hipkkt_kkt_system_solve_initial_point")

    def solve_dev(self, d_lhs, d_rhs, rhs_tau, rhs_kappa, d_var, var_tau, var_kappa, affine):
        """d_lhs, d_rhs, d_var: (x, s, z) triples of device pointers.  Returns (is_success, dtau, dkappa)."""
        tk = np.zeros(2)
        ok = check(_lib.lib().hipkkt_kkt_system_solve(
            self.ks._h, C.c_void_p(d_lhs[0]), C.c_void_p(d_lhs[1]), C.c_void_p(d_lhs[2]), ptr(tk),
            C.c_void_p(d_rhs[0]), C.c_void_p(d_rhs[1]), C.c_void_p(d_rhs[2]), float(rhs_tau), float(rhs_kappa),
            C.c_void_p(d_var[0]), C.c_void_p(d_var[1]), C.c_void_p(d_var[2]), float(var_tau), float(var_kappa),
            0 if affine else 1), "hipkkt_kkt_system_solve")
        return ok, tk[0], tk[1]

    def prepared(self, d_lhs, d_rhs, rhs_tau, rhs_kappa, d_var, var_tau, var_kappa):
        """(update, solve_affine, solve_combined) closures over pre-converted ctypes arguments for a caller that issues the
        same three calls on the same resident buffers every iteration (bench.py): the per-call argument marshalling of
        `update_dev` / `solve_dev` is host time during which the GPU idles.  d_rhs = (affine (x, s, z), combined (x, s, z))."""
        L = _lib.lib()
        h = self.ks._h
        tk = np.zeros(2)
        P = lambda v: C.c_void_p(v)
        upd_args = (h, P(d_var[1]), P(d_var[2]))
        sol_args = [(h, P(d_lhs[0]), P(d_lhs[1]), P(d_lhs[2]), ptr(tk), P(r[0]), P(r[1]), P(r[2]), C.c_double(rhs_tau),
                     C.c_double(rhs_kappa), P(d_var[0]), P(d_var[1]), P(d_var[2]), C.c_double(var_tau), C.c_double(var_kappa),
                     C.c_int(st)) for r, st in ((d_rhs[0], 0), (d_rhs[1], 1))]
        fu, fs = L.hipkkt_kkt_system_update, L.hipkkt_kkt_system_solve

        def update():
            return check(fu(*upd_args), "hipkkt_kkt_system_update")

        def solve(i):
            ok = check(fs(*sol_args[i]), "hipkkt_kkt_system_solve")
            return ok, tk[0], tk[1]
        return update, (lambda: solve(0)), (lambda: solve(1))

    def prepared_host(self, lhs, rhs, rhs_tau, rhs_kappa, var, var_tau, var_kappa, scaling, reduced=True, reuse_variables=True):
        """The same three closures for HOST-resident numpy vectors -- what integration/HipKKTExt.jl does per iteration:
        kkt_update! from the caller's cone scaling (reduced: hipkkt_kkt_system_update_scaling with (w, eta, lambda, R,
        Rinv); otherwise hipkkt_kkt_system_update_cones with the full nine arrays), then the two kkt_solve! through
        hipkkt_kkt_system_solve_host, the combined one re-using the affine one's variables.  lhs, var: (x, s, z) arrays;
        rhs = (affine (x, s, z), combined (x, s, z)); the arrays must stay alive (and may be page-locked: _lib.host_register)."""
        L = _lib.lib()
        h = self.ks._h
        tk = np.zeros(2)
        keep = (lhs, rhs, var, scaling, tk)
        fs = L.hipkkt_kkt_system_solve_host
        if reduced:
            w, eta, lam, R, Ri = scaling[-5:]
            fu, upd_args = L.hipkkt_kkt_system_update_scaling, (h, ptr(w), ptr(eta), ptr(lam), ptr(R), ptr(Ri))
        else:
            fu, upd_args = L.hipkkt_kkt_system_update_cones, (h,) + tuple(ptr(v) for v in scaling)
        sol_args = []
        for r, st in ((rhs[0], 0), (rhs[1], 1)):
            vv = (None, None, None) if (st == 1 and reuse_variables) else (ptr(var[0]), ptr(var[1]), ptr(var[2]))
            sol_args.append((h, ptr(lhs[0]), ptr(lhs[1]), ptr(lhs[2]), ptr(tk), ptr(r[0]), ptr(r[1]), ptr(r[2]), C.c_double(rhs_tau),
                             C.c_double(rhs_kappa)) + vv + (C.c_double(var_tau), C.c_double(var_kappa), C.c_int(st)))

        def update():
            return check(fu(*upd_args), "kkt_update! (host scaling)")

        def solve(i):
            ok = check(fs(*sol_args[i]), "hipkkt_kkt_system_solve_host")
            return ok, tk[0], tk[1]
        update._keep = keep
        return update, (lambda: solve(0)), (lambda: solve(1))

    # ---- lazy constant-RHS solve (hipkkt_kkt_system_set_lazy): kkt_update! + kkt_solve!(:affine) as two SEPARATE calls
    #      whose two solves still share one 2-column sweep
    def set_lazy(self, on=True):
        return check(_lib.lib().hipkkt_kkt_system_set_lazy(self.ks._h, int(on)), "hipkkt_kkt_system_set_lazy")

    def update_cones(self, Hsblocks, soc_u, soc_v, soc_eta2, w, eta, lam, psd_R=None, psd_Rinv=None):
        """kkt_update!(kktsystem, data, cones) from the CALLER's cone objects (what the Julia glue has): the data of
        kktsolver_update! plus the NT scaling (w, eta per cone, lambda, and R / Rinv of the PSD cones concatenated
        column-major) that kkt_solve!'s right-hand sides and step recovery use."""
        a = [f64(v if v is not None else []) for v in (Hsblocks, soc_u, soc_v, soc_eta2, w, eta, lam, psd_R, psd_Rinv)]
        i = self.ks.info
        if a[0].size != i["nHs"] or a[1].size != i["sparse_soc_len"] or a[2].size != a[1].size or \
                a[3].size != i["nsparse_soc"] or a[4].size != self.ks.m or a[6].size != self.ks.m or \
                a[5].size != len(self.ks.cones):
            raise ValueError("cone data has the wrong length")
        tot = sum(c.dim * c.dim for c in self.ks.cones if c.kind == 3)
        if a[7].size != tot or a[8].size != tot:
            raise ValueError("psd_R / psd_Rinv have the wrong length")
        return check(_lib.lib().hipkkt_kkt_system_update_cones(self.ks._h, *[ptr(v) for v in a]),
                     "hipkkt_kkt_system_update_cones")

    def update_scaling(self, w, eta, lam, psd_R=None, psd_Rinv=None):
        """kkt_update! from the NT scaling alone (hipkkt_kkt_system_update_scaling): the Hs blocks and the sparse
        second-order-cone vectors are formed on the device from (w, eta, R) instead of crossing PCIe."""
        a = [f64(v if v is not None else []) for v in (w, eta, lam, psd_R, psd_Rinv)]
        if a[0].size != self.ks.m or a[2].size != self.ks.m or a[1].size != len(self.ks.cones):
            raise ValueError("scaling data has the wrong length")
        tot = sum(c.dim * c.dim for c in self.ks.cones if c.kind == 3)
        if a[3].size != tot or a[4].size != tot:
            raise ValueError("psd_R / psd_Rinv have the wrong length")
        self._keep = a          # (registered or not, the arrays outlive the call)
        return check(_lib.lib().hipkkt_kkt_system_update_scaling(self.ks._h, *[ptr(v) for v in a]),
                     "hipkkt_kkt_system_update_scaling")

    # ---- numpy interface.  staging = "host": the *_host entry points of the C ABI (vectors staged by the library: what
    #      a caller with host-resident DefaultVariables uses); "torch": device tensors + the device-pointer entry points
    staging = "host"

    @property
    def _devstr(self):
        import torch
        d = self.ks.settings.device
        return f"cuda:{d if d >= 0 else torch.cuda.current_device()}"       # -1 = the current device

    def _dev(self, a):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(self._devstr)

    def update(self, s, z):
        if self.staging == "host":
            s, z = f64(s), f64(z)
            if s.size != self.ks.m or z.size != self.ks.m:
                raise ValueError("s, z must have length m")
            return check(_lib.lib().hipkkt_kkt_system_update_host(self.ks._h, ptr(s), ptr(z)), "hipkkt_kkt_system_update_host")
        ds, dz = self._dev(s), self._dev(z)
        return self.update_dev(ds.data_ptr(), dz.data_ptr())

    def update_and_solve_affine_dev(self, d_lhs, d_rhs, rhs_tau, rhs_kappa, d_var, tau, kappa):
        """The same on device pointers: d_lhs = (dx, ds, dz), d_rhs = (rhs.x, rhs.z), d_var = (x, s, z).
        Returns (is_success, dtau, dkappa)."""
        tk = np.zeros(2)
        ok = check(_lib.lib().hipkkt_kkt_system_update_and_solve_affine(
            self.ks._h, C.c_void_p(d_lhs[0]), C.c_void_p(d_lhs[1]), C.c_void_p(d_lhs[2]), ptr(tk),
            C.c_void_p(d_rhs[0]), C.c_void_p(d_rhs[1]), float(rhs_tau), float(rhs_kappa),
            C.c_void_p(d_var[0]), C.c_void_p(d_var[1]), C.c_void_p(d_var[2]), float(tau), float(kappa)),
            "hipkkt_kkt_system_update_and_solve_affine")
        return ok, float(tk[0]), float(tk[1])

    def update_and_solve_affine(self, rhs_x, rhs_z, rhs_tau, rhs_kappa, x, s, z, tau, kappa):
        """kkt_update! + kkt_solve!(:affine) in one call: the constant and the affine right-hand side share one
        2-column solve (hipkkt_kkt_system_update_and_solve_affine).  Returns (is_success, step | None)."""
        import torch
        dev = self._devstr
        n, m = self.ks.n, self.ks.m
        rhs = [self._dev(rhs_x), self._dev(rhs_z)]
        var = [self._dev(x), self._dev(s), self._dev(z)]
        lhs = [torch.zeros(max(k, 1), dtype=torch.float64, device=dev) for k in (n, m, m)]
        ok, dtau, dkappa = self.update_and_solve_affine_dev([t.data_ptr() for t in lhs], [t.data_ptr() for t in rhs],
                                                            rhs_tau, rhs_kappa, [t.data_ptr() for t in var], tau, kappa)
        if not ok:
            return False, None
        dx, ds, dz = lhs[0][:n].cpu().numpy(), lhs[1][:m].cpu().numpy(), lhs[2][:m].cpu().numpy()
        return True, (dx, dz, ds, dtau, dkappa)

    def solve_initial_point(self):
        n, m = self.ks.n, self.ks.m
        if self.staging == "host":
            x, s, z = np.zeros(max(n, 1)), np.zeros(max(m, 1)), np.zeros(max(m, 1))
            ok = check(_lib.lib().hipkkt_kkt_system_solve_initial_point_host(self.ks._h, ptr(x), ptr(s), ptr(z)),
                       "hipkkt_kkt_system_solve_initial_point_host")
            return ok, x[:n], s[:m], z[:m]
        import torch
        dev = self._devstr
        x = torch.zeros(max(n, 1), dtype=torch.float64, device=dev)
        s = torch.zeros(max(m, 1), dtype=torch.float64, device=dev)
        z = torch.zeros(max(m, 1), dtype=torch.float64, device=dev)
        ok = self.solve_initial_point_dev(x.data_ptr(), s.data_ptr(), z.data_ptr())
        return ok, x[:n].cpu().numpy(), s[:m].cpu().numpy(), z[:m].cpu().numpy()

    def solve(self, rhs_x, rhs_s, rhs_z, rhs_tau, rhs_kappa, x, s, z, tau, kappa, affine, reuse_variables=False):
        """reuse_variables (host staging): (x, s, z) are those of the previous call -- the combined step's are the affine
        step's (solver.jl:289-323) -- and are not sent again."""
        n, m = self.ks.n, self.ks.m
        if self.staging == "host":
            a = [f64(v) for v in (rhs_x, rhs_s, rhs_z, x, s, z)]
            for v, k in zip(a, (n, m, m, n, m, m)):
                if v.size != k:
                    raise ValueError("vector of the wrong length")
            lhs = [np.zeros(max(k, 1)) for k in (n, m, m)]
            tk = np.zeros(2)
            ok = check(_lib.lib().hipkkt_kkt_system_solve_host(
                self.ks._h, ptr(lhs[0]), ptr(lhs[1]), ptr(lhs[2]), ptr(tk), ptr(a[0]), ptr(a[1]), ptr(a[2]),
                float(rhs_tau), float(rhs_kappa), *([None, None, None] if reuse_variables else [ptr(a[3]), ptr(a[4]), ptr(a[5])]),
                float(tau), float(kappa), 0 if affine else 1), "hipkkt_kkt_system_solve_host")
            if not ok:
                return False, None
            return True, (lhs[0][:n], lhs[2][:m], lhs[1][:m], float(tk[0]), float(tk[1]))
        import torch
        dev = self._devstr
        rhs = [self._dev(rhs_x), self._dev(rhs_s), self._dev(rhs_z)]
        var = [self._dev(x), self._dev(s), self._dev(z)]
        lhs = [torch.zeros(max(k, 1), dtype=torch.float64, device=dev) for k in (n, m, m)]
        ok, dtau, dkappa = self.solve_dev([t.data_ptr() for t in lhs], [t.data_ptr() for t in rhs], rhs_tau, rhs_kappa,
                                          [t.data_ptr() for t in var], tau, kappa, affine)
        if not ok:
            return False, None
        dx, ds, dz = lhs[0][:n].cpu().numpy(), lhs[1][:m].cpu().numpy(), lhs[2][:m].cpu().numpy()
        return True, (dx, dz, ds, float(dtau), float(dkappa))


class HipDirectLDLSolver:
    """An AbstractDirectLDLSolver backend (`ldlsolver_constructor(::Val{:hipldl})`):
    constructor(KKT, Dsigns, settings), update_values!, scale_values!, refactor!, solve!."""

    matrix_shape = "triu"           # ldlsolver_matrix_shape

    @staticmethod
    def is_available():             # ldlsolver_is_available: must not throw
        try:
            return bool(_lib.lib().hipkkt_available())
        except Exception:
            return False

    def __init__(self, KKT, Dsigns, settings=None):
        L = _lib.lib()
        KKT = sp.csc_matrix(KKT)
        KKT.sort_indices()
        if sp.tril(KKT, -1).nnz:
            raise ValueError("KKT must be upper triangular (:triu)")
        self.N = KKT.shape[0]
        self.nnzK = KKT.nnz
        self.settings = settings if settings is not None else _lib.default_settings()
        self._h = C.c_void_p()
        cp, ri, nz, ds = i64(KKT.indptr), i64(KKT.indices), f64(KKT.data), i64(Dsigns)
        rc = L.hipkkt_ldl_create(C.byref(self._h), self.N, ptr(cp), ptr(ri), ptr(nz), ptr(ds),
                                 C.byref(self.settings), 0)
        if not check(rc, "hipkkt_ldl_create"):
            raise _lib.HipKKTError("hipkkt_ldl_create reported a numeric failure")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _lib.lib().hipkkt_ldl_destroy(h)
            self._h = None

    def update_values(self, index, values):
        idx, v = i64(index), f64(values)
        check(_lib.lib().hipkkt_ldl_update_values(self._h, ptr(idx), ptr(v), idx.size), "hipkkt_ldl_update_values")

    def scale_values(self, index, scale):
        idx = i64(index)
        check(_lib.lib().hipkkt_ldl_scale_values(self._h, ptr(idx), float(scale), idx.size), "hipkkt_ldl_scale_values")

    def refactor(self, K=None):
        return check(_lib.lib().hipkkt_ldl_refactor(self._h), "hipkkt_ldl_refactor")

    def solve(self, K, x, b):
        b = f64(b)
        if x.dtype != np.float64 or x.size != self.N or b.size != self.N:
            raise ValueError("x, b must be float64 of length N")
        check(_lib.lib().hipkkt_ldl_solve(self._h, ptr(x), ptr(b)), "hipkkt_ldl_solve")

    def solve_multi(self, K, X, B):
        """solve! on the columns of B (N, k) into X (N, k), both Fortran-ordered float64."""
        B = np.asfortranarray(B, dtype=np.float64)
        if X.dtype != np.float64 or X.shape != B.shape or B.shape[0] != self.N or not X.flags.f_contiguous:
            raise ValueError("X, B must be Fortran-ordered float64 (N, k)")
        check(_lib.lib().hipkkt_ldl_solve_multi(self._h, B.shape[1], ptr(X), ptr(B)), "hipkkt_ldl_solve_multi")

    def linear_solver_info(self):
        info = _lib.Info()
        check(_lib.lib().hipkkt_ldl_info(self._h, C.byref(info)), "hipkkt_ldl_info")
        return LinearSolverInfo("hipldl", 1, True, info.nnzK, info.nnzL)

    def perm(self):
        out = np.zeros(self.N, dtype=np.int64)
        check(_lib.lib().hipkkt_ldl_get_perm(self._h, ptr(out)), "hipkkt_ldl_get_perm")
        return out

    @property
    def fallbacks(self):
        out = np.zeros(2, dtype=np.int64)
        check(_lib.lib().hipkkt_ldl_fallbacks(self._h, ptr(out)), "hipkkt_ldl_fallbacks")
        return int(out[0]), int(out[1])
