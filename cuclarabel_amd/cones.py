"""Cone specification types, mirroring the reference's user-facing cone API
(`/root/reference/src/cones/cone_api.jl:18-55`): a problem's cone list is a sequence of
`ZeroConeT(dim)`, `NonnegativeConeT(dim)`, `SecondOrderConeT(dim)`, `PSDTriangleConeT(k)`.

`PSDTriangleConeT(k)` takes the matrix side length k; the cone then has k(k+1)/2 rows
(`cone_types.jl:171-186`).  Kind codes are the ones `include/hipkkt.h` uses.
"""
from dataclasses import dataclass

KIND_ZERO, KIND_NN, KIND_SOC, KIND_PSD = 0, 1, 2, 3

# cone_types.jl:101 -- second-order cones larger than this use the sparse expansion
SOC_NO_EXPANSION_MAX_SIZE = 4


@dataclass(frozen=True)
class _ConeT:
    dim: int

    @property
    def numel(self) -> int:
        return self.dim


class ZeroConeT(_ConeT):
    kind = KIND_ZERO


class NonnegativeConeT(_ConeT):
    kind = KIND_NN


class SecondOrderConeT(_ConeT):
    kind = KIND_SOC



class PSDTriangleConeT(_ConeT):
    kind = KIND_PSD

    @property
    def numel(self) -> int:
        return self.dim * (self.dim + 1) // 2


def cones_new_collapsed(cones):
    """Merge runs of nonnegative cones (and 1-D second-order / PSD cones, which are
    nonnegative cones) into one, and drop empty cones, as the reference does before the
    KKT system sees the list (`cone_api.jl:96-153`)."""
    def collapsible(c):
        return isinstance(c, NonnegativeConeT) or (
            isinstance(c, (SecondOrderConeT, PSDTriangleConeT)) and c.dim == 1)

    out = []
    run = None  # total dim of the nonnegative run being collapsed
    for c in cones:
        if c.numel == 0:
            continue
        if collapsible(c):
            run = (run or 0) + c.numel
            continue
        if run is not None:
            out.append(NonnegativeConeT(run))
            run = None
        out.append(c)
    if run is not None:
        out.append(NonnegativeConeT(run))
    return out


def cone_kinds_dims(cones):
    import numpy as np
    kinds = np.array([c.kind for c in cones], dtype=np.int32)
    dims = np.array([c.dim for c in cones], dtype=np.int64)
    return kinds, dims


def total_numel(cones) -> int:
    return sum(c.numel for c in cones)
