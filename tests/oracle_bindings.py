"""ctypes bindings for oracle/libkktoracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's KKT path (oracle/kkt_oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KKT_ORACLE_SO: bench.py's cpu_baseline points this at the -march=native build it makes on the box it runs on
_SO = os.environ.get("KKT_ORACLE_SO") or os.path.join(os.path.dirname(_HERE), "oracle", "libkktoracle.so")

_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class Settings(C.Structure):
    _fields_ = [("static_reg_constant", C.c_double), ("static_reg_proportional", C.c_double),
                ("dynamic_reg_eps", C.c_double), ("dynamic_reg_delta", C.c_double),
                ("ir_reltol", C.c_double), ("ir_abstol", C.c_double),
                ("ir_stop_ratio", C.c_double), ("ir_max_iter", C.c_int),
                ("static_reg_enable", C.c_int), ("ir_enable", C.c_int)]


def _load():
    lib = C.CDLL(_SO)
    lib.orc_kkt_new.restype = C.c_void_p
    lib.orc_kkt_new.argtypes = [C.c_int64, C.c_int64, _i64p, _i64p, _f64p, _i64p, _i64p, _f64p,
                                C.c_int64, _i32p, _i64p, C.c_void_p, C.c_void_p]
    lib.orc_kkt_free.argtypes = [C.c_void_p]
    lib.orc_kkt_sizes.argtypes = [C.c_void_p, _i64p]
    for name in ("colptr", "rowval", "map_P", "map_A", "map_Hs", "map_diag_full", "map_soc_u",
                 "map_soc_v", "map_soc_D", "dsigns", "perm"):
        f = getattr(lib, "orc_kkt_" + name)
        f.restype = C.POINTER(C.c_int64)
        f.argtypes = [C.c_void_p]
    lib.orc_kkt_nzval.restype = C.POINTER(C.c_double)
    lib.orc_kkt_nzval.argtypes = [C.c_void_p]
    lib.orc_ldl_Dinv.restype = C.POINTER(C.c_double)
    lib.orc_ldl_Dinv.argtypes = [C.c_void_p]
    lib.orc_kkt_last_regularizer.restype = C.c_double
    lib.orc_kkt_last_regularizer.argtypes = [C.c_void_p]
    lib.orc_kkt_last_ir_iters.restype = C.c_int64
    lib.orc_kkt_last_ir_iters.argtypes = [C.c_void_p]
    lib.orc_kkt_num_dyn_regularized.restype = C.c_int64
    lib.orc_kkt_num_dyn_regularized.argtypes = [C.c_void_p]
    lib.orc_cones_update_scaling.restype = C.c_int
    lib.orc_cones_update_scaling.argtypes = [C.c_void_p, _f64p, _f64p]
    lib.orc_cones_set_identity_scaling.argtypes = [C.c_void_p]
    lib.orc_cones_get_Hs.argtypes = [C.c_void_p, _f64p]
    lib.orc_cones_mul_Hs.argtypes = [C.c_void_p, _f64p, _f64p]
    lib.orc_cones_soc_sparse.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, _f64p]
    lib.orc_cones_lambda.argtypes = [C.c_void_p, _f64p]
    lib.orc_cones_psd_scaling.argtypes = [C.c_void_p, _f64p, _f64p]
    lib.orc_kkt_update.restype = C.c_int
    lib.orc_kkt_update.argtypes = [C.c_void_p]
    lib.orc_kkt_update_values.restype = C.c_int
    lib.orc_kkt_update_values.argtypes = [C.c_void_p, _f64p, _f64p, _f64p, _f64p]
    lib.orc_kkt_update_P.argtypes = [C.c_void_p, _f64p]
    lib.orc_kkt_update_A.argtypes = [C.c_void_p, _f64p]
    lib.orc_kkt_setrhs.argtypes = [C.c_void_p, _f64p, _f64p]
    lib.orc_kkt_solve.restype = C.c_int
    lib.orc_kkt_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_ldl_solve.argtypes = [C.c_void_p, _f64p, _f64p]
    lib.orc_ldl_refactor.restype = C.c_int
    lib.orc_ldl_refactor.argtypes = [C.c_void_p]
    lib.orc_kkt_residual.restype = C.c_double
    lib.orc_kkt_residual.argtypes = [C.c_void_p, _f64p, _f64p, _f64p]
    lib.orc_min_degree.argtypes = [C.c_int64, _i64p, _i64p, _i64p]
    lib.orc_default_settings.argtypes = [C.c_void_p]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def default_settings(**kw):
    s = Settings()
    lib().orc_default_settings(C.byref(s))
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


class OracleKKT:
    """The oracle's DirectLDLKKTSolver: same method names as the reference's
    `kktsolver_*` generics (kktsolver_directldl.jl)."""

    def __init__(self, P, A, cones, perm=None, settings=None):
        from cuclarabel_amd.cones import cone_kinds_dims
        import scipy.sparse as sp
        P = sp.triu(sp.csc_matrix(P), format="csc"); P.sort_indices()
        A = sp.csc_matrix(A); A.sort_indices()
        self.n, self.m = P.shape[0], A.shape[0]
        self._nnz_P, self._nnz_A = P.nnz, A.nnz
        kinds, dims = cone_kinds_dims(cones)
        self.cones = list(cones)
        self._settings = settings or default_settings()
        permp = None
        if perm is not None:
            self._perm_in = np.ascontiguousarray(perm, dtype=np.int64)
            permp = self._perm_in.ctypes.data_as(C.c_void_p)
        self._h = lib().orc_kkt_new(
            self.n, self.m,
            P.indptr.astype(np.int64), P.indices.astype(np.int64), P.data.astype(np.float64),
            A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64),
            len(cones), kinds, dims, permp, C.byref(self._settings))
        if not self._h:
            raise ValueError("orc_kkt_new failed (cone dims do not sum to m?)")
        sz = np.zeros(9, dtype=np.int64)
        lib().orc_kkt_sizes(self._h, sz)
        (_, _, self.p, self.N, self.nnzK, self.nHs, self.nnzL, self.nsparse, self.sparse_len) = sz.tolist()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kkt_free(self._h)
            self._h = None

    # --- views of the assembled matrix and maps (copies)
    def K(self):
        import scipy.sparse as sp
        L = lib()
        colptr = _arr(L.orc_kkt_colptr(self._h), self.N + 1, np.int64)
        rowval = _arr(L.orc_kkt_rowval(self._h), self.nnzK, np.int64)
        nzval = _arr(L.orc_kkt_nzval(self._h), self.nnzK, np.float64)
        return sp.csc_matrix((nzval, rowval, colptr), shape=(self.N, self.N))

    def K_full(self):
        import scipy.sparse as sp
        U = self.K()
        return (U + sp.triu(U, 1).T).tocsc()

    def maps(self):
        L = lib()
        return dict(
            P=_arr(L.orc_kkt_map_P(self._h), self._nnzP(), np.int64),
            A=_arr(L.orc_kkt_map_A(self._h), self._nnzA(), np.int64),
            Hsblocks=_arr(L.orc_kkt_map_Hs(self._h), self.nHs, np.int64),
            diag_full=_arr(L.orc_kkt_map_diag_full(self._h), self.N, np.int64),
            soc_u=_arr(L.orc_kkt_map_soc_u(self._h), self.sparse_len, np.int64),
            soc_v=_arr(L.orc_kkt_map_soc_v(self._h), self.sparse_len, np.int64),
            soc_D=_arr(L.orc_kkt_map_soc_D(self._h), 2 * self.nsparse, np.int64))

    def _nnzP(self):
        return self._nnz_P

    def _nnzA(self):
        return self._nnz_A

    def dsigns(self):
        return _arr(lib().orc_kkt_dsigns(self._h), self.N, np.int64)

    def perm(self):
        return _arr(lib().orc_kkt_perm(self._h), self.N, np.int64)

    def Dinv(self):
        return _arr(lib().orc_ldl_Dinv(self._h), self.N, np.float64)

    # --- cones
    def update_scaling(self, s, z):
        return bool(lib().orc_cones_update_scaling(self._h, np.ascontiguousarray(s, dtype=np.float64),
                                                   np.ascontiguousarray(z, dtype=np.float64)))

    def set_identity_scaling(self):
        lib().orc_cones_set_identity_scaling(self._h)

    def get_Hs(self):
        out = np.zeros(max(self.nHs, 1))
        lib().orc_cones_get_Hs(self._h, out)
        return out[:self.nHs]

    def mul_Hs(self, x):
        y = np.zeros(max(self.m, 1))
        lib().orc_cones_mul_Hs(self._h, y, np.ascontiguousarray(x, dtype=np.float64))
        return y[:self.m]

    def soc_sparse(self):
        u = np.zeros(max(self.sparse_len, 1)); v = np.zeros(max(self.sparse_len, 1))
        e2 = np.zeros(max(self.nsparse, 1)); d = np.zeros(max(self.nsparse, 1))
        lib().orc_cones_soc_sparse(self._h, u, v, e2, d)
        return u[:self.sparse_len], v[:self.sparse_len], e2[:self.nsparse], d[:self.nsparse]

    def cone_lambda(self):
        out = np.zeros(max(self.m, 1))
        lib().orc_cones_lambda(self._h, out)
        return out[:self.m]

    def psd_scaling(self):
        """[(R, Rinv, lam)] per PSD cone, matrices k x k"""
        ks = [c.dim for c in self.cones if c.kind == 3]
        tot = sum(k * k for k in ks)
        R, Ri = np.zeros(max(tot, 1)), np.zeros(max(tot, 1))
        lib().orc_cones_psd_scaling(self._h, R, Ri)
        lam = self.cone_lambda()
        out, o, off = [], 0, 0
        for c in self.cones:
            if c.kind == 3:
                k = c.dim
                out.append((R[o:o + k * k].reshape(k, k, order="F").copy(), Ri[o:o + k * k].reshape(k, k, order="F").copy(),
                            lam[off:off + k].copy()))
                o += k * k
            off += c.numel
        return out

    # --- the kktsolver_* interface
    def kktsolver_update(self):
        return bool(lib().orc_kkt_update(self._h))

    def kktsolver_update_values(self, Hs, soc_u, soc_v, soc_eta2):
        pad = lambda a: np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64), [0.0]]))
        return bool(lib().orc_kkt_update_values(self._h, pad(Hs), pad(soc_u), pad(soc_v), pad(soc_eta2)))

    def kktsolver_update_P(self, Px):
        lib().orc_kkt_update_P(self._h, np.ascontiguousarray(Px, dtype=np.float64))

    def kktsolver_update_A(self, Ax):
        lib().orc_kkt_update_A(self._h, np.ascontiguousarray(Ax, dtype=np.float64))

    def kktsolver_setrhs(self, rhsx, rhsz):
        lib().orc_kkt_setrhs(self._h, np.ascontiguousarray(rhsx, dtype=np.float64),
                             np.ascontiguousarray(rhsz, dtype=np.float64))

    def kktsolver_solve(self, want_x=True, want_z=True):
        x = np.zeros(max(self.n, 1)); z = np.zeros(max(self.m, 1))
        ok = lib().orc_kkt_solve(self._h, x.ctypes.data_as(C.c_void_p) if want_x else None,
                                 z.ctypes.data_as(C.c_void_p) if want_z else None)
        return bool(ok), x[:self.n], z[:self.m]

    def ldl_solve(self, b):
        x = np.zeros(self.N)
        lib().orc_ldl_solve(self._h, x, np.ascontiguousarray(b, dtype=np.float64))
        return x

    def ldl_refactor(self):
        return bool(lib().orc_ldl_refactor(self._h))

    def residual(self, b, x):
        e = np.zeros(self.N)
        nrm = lib().orc_kkt_residual(self._h, e, np.ascontiguousarray(b, dtype=np.float64),
                                     np.ascontiguousarray(x, dtype=np.float64))
        return nrm, e

    @property
    def last_regularizer(self):
        return lib().orc_kkt_last_regularizer(self._h)

    @property
    def last_ir_iters(self):
        return int(lib().orc_kkt_last_ir_iters(self._h))

    @property
    def num_dyn_regularized(self):
        return int(lib().orc_kkt_num_dyn_regularized(self._h))


def make_oracle(pb, perm=None, settings=None):
    """OracleKKT for a cuclarabel_amd.problems.Problem"""
    return OracleKKT(pb.P, pb.A, pb.cones, perm=perm, settings=settings)


def min_degree(K_triu):
    import scipy.sparse as sp
    K = sp.csc_matrix(K_triu); K.sort_indices()
    N = K.shape[0]
    perm = np.zeros(N, dtype=np.int64)
    lib().orc_min_degree(N, K.indptr.astype(np.int64), K.indices.astype(np.int64), perm)
    return perm
