"""Backends for cuclarabel_amd.ipm.solve in the tests: the CPU oracle (checker) -- the HIP backend
lives in the package (cuclarabel_amd.ipm.HipBackend)."""
import numpy as np

from cuclarabel_amd.ipm import identity_scaling_data
from tests.oracle_bindings import OracleKKT


class OracleBackend:
    def __init__(self, P, A, cone_specs):
        self.o = OracleKKT(P, A, cone_specs)
        self.specs = list(cone_specs)

    def update_identity(self):
        self.o.set_identity_scaling()
        # the oracle's own identity scaling must agree with what the glue would hand over
        Hs, u, v, e2 = identity_scaling_data(self.specs)
        np.testing.assert_array_equal(self.o.get_Hs(), Hs)
        return self.o.kktsolver_update()

    def update(self, s, z):
        return self.o.update_scaling(s, z) and self.o.kktsolver_update()

    def kktsolver_setrhs(self, rx, rz):
        self.o.kktsolver_setrhs(rx, rz)

    def kktsolver_solve(self, x, z):
        ok, xo, zo = self.o.kktsolver_solve(x is not None, z is not None)
        if x is not None:
            x[:] = xo
        if z is not None:
            z[:] = zo
        return ok

    @property
    def last_ir_iterations(self):
        return self.o.last_ir_iters
