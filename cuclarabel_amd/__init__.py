"""cuclarabel_amd -- MI355X-native KKT linear-system solver for Clarabel-style IPMs.

Host-side mirror of the reference's KKT-solver interface
(`/root/reference/src/kktsolvers/kktsolver_defaults.jl:2-48`,
`direct-ldl/directldl_defaults.jl:1-72`) over the C ABI of `libhipkkt.so`
(`include/hipkkt.h`).  The HIP library is required: importing the solver classes
raises if it has not been built (`python -c "import __graft_entry__ as g; g.build()"`).
"""
from .cones import (ZeroConeT, NonnegativeConeT, SecondOrderConeT, PSDTriangleConeT,
                    cones_new_collapsed)

__all__ = ["ZeroConeT", "NonnegativeConeT", "SecondOrderConeT", "PSDTriangleConeT",
           "cones_new_collapsed"]
