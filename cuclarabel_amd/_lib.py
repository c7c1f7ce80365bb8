"""ctypes loader for libhipkkt.so (the C ABI declared in include/hipkkt.h).

The HIP library is the product: there is no CPU fallback.  `lib()` raises if the shared
library has not been built in-tree (`python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (HIPKKT_LIB: another build of the same library, for A/B timing of kernel variants on one box)
SO_PATH = os.environ.get("HIPKKT_LIB") or os.path.join(_HERE, "libhipkkt.so")

OK, NUMERIC_FAILURE, REFINEMENT_INCOMPLETE = 0, 1, 2
ORDER_AMD, ORDER_ND, ORDER_NATURAL, ORDER_USER = 0, 1, 2, 3


class Settings(C.Structure):
    """hipkkt_settings -- the path-relevant fields of Clarabel.Settings (settings.jl:110-132)."""
    _fields_ = [
        ("static_regularization_constant", C.c_double),
        ("static_regularization_proportional", C.c_double),
        ("dynamic_regularization_eps", C.c_double),
        ("dynamic_regularization_delta", C.c_double),
        ("iterative_refinement_reltol", C.c_double),
        ("iterative_refinement_abstol", C.c_double),
        ("iterative_refinement_stop_ratio", C.c_double),
        ("iterative_refinement_max_iter", C.c_int32),
        ("static_regularization_enable", C.c_int32),
        ("iterative_refinement_enable", C.c_int32),
        ("ordering", C.c_int32),
        ("nd_leaf_size", C.c_int32),
        ("device", C.c_int32),
        ("user_perm", C.c_void_p),
        ("amd_dense_scale", C.c_double),
    ]


class Info(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n", "m", "p", "N", "nnzK", "nnzL", "nnzL_stored", "nsuper",
                                         "nlevels", "max_front", "etree_height", "nHs", "nsparse_soc",
                                         "sparse_soc_len")] + \
               [("factor_flops", C.c_double), ("front_bytes", C.c_double), ("update_bytes", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Profile(C.Structure):
    _fields_ = [("update_ms", C.c_double), ("factor_ms", C.c_double), ("trisolve_ms", C.c_double),
                ("residual_ms", C.c_double), ("other_ms", C.c_double),
                ("n_update", C.c_int64), ("n_factor", C.c_int64), ("n_trisolve", C.c_int64),
                ("n_residual", C.c_int64), ("ir_iterations", C.c_int64),
                ("dynamic_regularizations", C.c_int64),
                ("overlap_fallbacks", C.c_int64), ("top_fallbacks", C.c_int64), ("overlap_deferrals", C.c_int64), ("top_deferrals", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/hipkkt.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "hipkkt_available": (C.c_int, []),
    "hipkkt_last_error": (C.c_char_p, []),
    "hipkkt_default_settings": (None, [_P]),
    "hipkkt_version": (C.c_char_p, []),
    "hipkkt_symbolic_analyse": (C.c_int, [C.c_int64, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "hipkkt_ldl_create": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, C.c_int]),
    "hipkkt_ldl_destroy": (None, [_P]),
    "hipkkt_ldl_update_values": (C.c_int, [_P, _P, _P, C.c_int64]),
    "hipkkt_ldl_scale_values": (C.c_int, [_P, _P, C.c_double, C.c_int64]),
    "hipkkt_ldl_refactor": (C.c_int, [_P]),
    "hipkkt_ldl_solve": (C.c_int, [_P, _P, _P]),
    "hipkkt_ldl_solve_dev": (C.c_int, [_P, _P, _P]),
    "hipkkt_ldl_solve_multi": (C.c_int, [_P, C.c_int64, _P, _P]),
    "hipkkt_ldl_solve_multi_dev": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64]),
    "hipkkt_ldl_info": (C.c_int, [_P, _P]),
    "hipkkt_ldl_get_perm": (C.c_int, [_P, _P]),
    "hipkkt_ldl_fallbacks": (C.c_int, [_P, _P]),
    "hipkkt_kkt_create": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_int64, _P, _P, _P, C.c_int]),
    "hipkkt_kkt_destroy": (None, [_P]),
    "hipkkt_kkt_info": (C.c_int, [_P, _P]),
    "hipkkt_kkt_update_cones": (C.c_int, [_P, _P, _P, _P, _P]),
    "hipkkt_kkt_update_from_sz": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_update_from_sz_dev": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_update_P": (C.c_int, [_P, _P]),
    "hipkkt_kkt_update_A": (C.c_int, [_P, _P]),
    "hipkkt_kkt_setrhs": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_solve": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_setrhs_dev": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_solve_dev": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_set_deferred_status": (C.c_int, [_P, C.c_int]),
    "hipkkt_kkt_deferred_status": (C.c_int, [_P]),
    "hipkkt_kkt_solve_multi": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P]),
    "hipkkt_kkt_solve_multi_dev": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P]),
    "hipkkt_kkt_system_init": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_system_update": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_system_solve_constant_rhs": (C.c_int, [_P]),
    "hipkkt_kkt_system_solve_initial_point": (C.c_int, [_P, _P, _P, _P]),
    "hipkkt_kkt_system_solve": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double,
                                          _P, _P, _P, C.c_double, C.c_double, C.c_int]),
    "hipkkt_kkt_system_update_and_solve_affine": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double,
                                                            _P, _P, _P, C.c_double, C.c_double]),
    "hipkkt_kkt_system_set_lazy": (C.c_int, [_P, C.c_int]),
    "hipkkt_kkt_system_update_cones": (C.c_int, [_P] * 10),
    "hipkkt_kkt_system_update_scaling": (C.c_int, [_P] * 6),
    "hipkkt_selftest_handover": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "hipkkt_host_register": (C.c_int, [_P, C.c_int64]),
    "hipkkt_host_unregister": (C.c_int, [_P]),
    "hipkkt_kkt_system_update_host": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_system_solve_initial_point_host": (C.c_int, [_P, _P, _P, _P]),
    "hipkkt_kkt_system_solve_host": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double,
                                               _P, _P, _P, C.c_double, C.c_double, C.c_int]),
    "hipkkt_equilibrate": (C.c_int, [C.c_int64, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, _P, _P,
                                     C.c_int32, C.c_double, C.c_double, _P, _P, _P, C.c_int, C.c_int]),
    "hipkkt_scale_matrix_values": (C.c_int, [C.c_int64, C.c_int64, _P, _P, _P, _P, _P, C.c_double, C.c_int, C.c_int]),
    "hipkkt_kkt_mul_Hs": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_speculative_rounds": (C.c_int, [_P, C.c_int]),
    "hipkkt_kkt_get_pattern": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_get_values": (C.c_int, [_P, _P]),
    "hipkkt_kkt_get_maps": (C.c_int, [_P] * 9),
    "hipkkt_kkt_get_perm": (C.c_int, [_P, _P]),
    "hipkkt_kkt_get_Hs": (C.c_int, [_P, _P]),
    "hipkkt_kkt_get_scaling": (C.c_int, [_P, _P, _P, _P]),
    "hipkkt_kkt_get_scaling_w": (C.c_int, [_P, _P, _P]),
    "hipkkt_kkt_last_regularizer": (C.c_double, [_P]),
    "hipkkt_kkt_last_ir_iterations": (C.c_int64, [_P]),
    "hipkkt_kkt_set_stream": (C.c_int, [_P, _P]),
    "hipkkt_kkt_synchronize": (C.c_int, [_P]),
    "hipkkt_kkt_profile_enable": (C.c_int, [_P, C.c_int]),
    "hipkkt_kkt_profile_reset": (C.c_int, [_P]),
    "hipkkt_kkt_profile_get": (C.c_int, [_P, _P]),
}

_lib = None


class HipKKTError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise HipKKTError(
                f"{SO_PATH} is missing: the HIP extension is required (no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
        # PyTorch-ROCm ships its own copy of the HIP runtime.  Whichever copy a process loads first serves both;
        # torch fails to see the GPU ("No HIP GPUs are available") when the system copy got in first.  Callers
        # that also use torch for device memory / streams / RCCL therefore need torch's copy loaded before
        # ours: importing torch (not initialising it) is enough.  Without torch installed nothing happens.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)       # AttributeError if the ABI is incomplete
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    """0 -> True, >0 (numeric failure) -> False, <0 -> raise; the reference's Bool convention."""
    if rc == OK:
        return True
    if rc > 0:
        return False
    raise HipKKTError(f"{what} failed ({rc}): {lib().hipkkt_last_error().decode()}")


def default_settings(**kw):
    s = Settings()
    lib().hipkkt_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise TypeError(f"unknown setting {k}")
        setattr(s, k, v)
    return s


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def symbolic_analyse(K_triu, ordering=ORDER_ND, nd_leaf_size=0):
    """Host-only ordering + structure statistics of a triu CSC pattern -> (perm, info dict)."""
    import scipy.sparse as sp
    K = sp.csc_matrix(K_triu)
    K.sort_indices()
    N = K.shape[0]
    cp, ri = i64(K.indptr), i64(K.indices)
    perm = np.zeros(N, dtype=np.int64)
    info = Info()
    check(lib().hipkkt_symbolic_analyse(N, ptr(cp), ptr(ri), 0, ordering, nd_leaf_size, ptr(perm), C.byref(info)),
          "hipkkt_symbolic_analyse")
    return perm, info.as_dict()


def host_register(a):
    """Page-lock a numpy array the caller keeps (hipkkt_host_register): the *_host entry points then copy at the link's
    rate.  Returns True if the range could be registered."""
    return lib().hipkkt_host_register(ptr(a), int(a.nbytes)) == 0


def host_unregister(a):
    return lib().hipkkt_host_unregister(ptr(a)) == 0
