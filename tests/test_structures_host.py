"""CPU side of the structure zoo (cuclarabel_amd.problems.ZOO_SMALL: sparsity patterns outside the five BASELINE
configurations): the product's host-side ordering and symbolic analysis on each shape, the oracle on the product's
permutation against scipy, and the ordering's cost on graphs of very many components."""
import time

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from cuclarabel_amd import problems, _lib
from tests.oracle_bindings import make_oracle


@pytest.mark.parametrize("name,maker", problems.ZOO_SMALL, ids=[c[0] for c in problems.ZOO_SMALL])
@pytest.mark.parametrize("ordering", [_lib.ORDER_ND, _lib.ORDER_AMD])
def test_zoo_symbolic_and_oracle_against_scipy(name, maker, ordering):
    pb = maker()
    K = make_oracle(pb).K()
    perm, info = _lib.symbolic_analyse(K, ordering=ordering)
    assert sorted(perm.tolist()) == list(range(K.shape[0]))
    o = make_oracle(pb, perm=perm)
    assert o.nnzL == info["nnzL"] <= info["nnzL_stored"]        # QDLDL's column counts on the same permutation
    assert o.update_scaling(pb.s0, pb.z0) and o.kktsolver_update()
    assert o.num_dyn_regularized == 0
    rng = np.random.default_rng(3)
    rx, rz = rng.standard_normal(pb.n), rng.standard_normal(pb.m)
    o.kktsolver_setrhs(rx, rz)
    ok, x, z = o.kktsolver_solve()
    assert ok and 0 <= o.last_ir_iters <= 10
    b = np.concatenate([rx, rz, np.zeros(o.p)])
    # iterative refinement works on the un-regularised K (kktsolver_directldl.jl:283-291), which is non-singular for
    # every zoo problem (equality rows of full row rank), so that is the system scipy solves
    Kref = o.K_full().tocsc()
    tol = 1e-8
    xf = spla.splu(Kref).solve(b)
    scale = np.abs(xf).max()
    assert max(np.abs(x - xf[:pb.n]).max(), np.abs(z - xf[pb.n:pb.n + pb.m]).max()) / scale < tol


def test_nested_dissection_is_linear_in_the_number_of_components():
    """A separable problem: K is n independent 2 x 2 blocks.  Splitting components off one at a time made the ordering
    quadratic in their number (60 000 components: 7.7 s, 80 000: ~13 s); all components of a task are found in one
    pass now (80 000: ~0.1 s).  The bound asserted is far from both."""
    times = []
    for n in (40_000, 80_000):
        K = make_oracle(problems.zoo_diag(n=n)).K()
        t = time.perf_counter()
        perm, info = _lib.symbolic_analyse(K, ordering=_lib.ORDER_ND)
        times.append(time.perf_counter() - t)
        assert sorted(perm.tolist()) == list(range(2 * n))
        assert info["nnzL"] == n and info["nlevels"] == 1 and info["max_front"] == 2
    assert max(times) < 2.0, times


def test_block_diagonal_fill_is_the_sum_of_its_parts():
    """Components larger than a leaf are dissected on their own, in the order of their first nodes: the block-diagonal
    problem's nnz(L) is the sum of the parts' (what the one-at-a-time split gave; the permutations of cfg4b, the forest
    and the separable problem were compared bit for bit against the previous build when the split was rewritten)."""
    pbs = [problems.config2(seed=40 + j, n=300 + 100 * j) for j in range(4)]
    total = 0
    for pb in pbs:
        _, info = _lib.symbolic_analyse(make_oracle(pb).K(), ordering=_lib.ORDER_ND, nd_leaf_size=200)
        total += info["nnzL"]
    K = make_oracle(problems.block_diagonal(pbs)).K()
    perm, info = _lib.symbolic_analyse(K, ordering=_lib.ORDER_ND, nd_leaf_size=200)
    assert sorted(perm.tolist()) == list(range(K.shape[0]))
    assert info["nnzL"] == total


_BUNDLE_SCRIPT = r"""
import json, sys
sys.path.insert(0, {root!r})
from cuclarabel_amd import problems, _lib
from tests.oracle_bindings import make_oracle
out = {{}}
for name, mk in [("cfg2", lambda: problems.config2(n=6000)), ("cfg3", lambda: problems.config3(nblocks=6, blk=300)),
                 ("cfg5", lambda: problems.config5(n=600, npsd=12, psd_dim=12, nsoc=8, soc_dim=30)),
                 ("arrow", lambda: problems.zoo_arrow(n=8000)), ("dense", lambda: problems.zoo_dense(n=200, m=400)),
                 ("lp", lambda: problems.zoo_lp_transport(60, 500))]:
    pb = mk()
    perm, info = _lib.symbolic_analyse(make_oracle(pb).K())
    out[name] = dict(nsuper=info["nsuper"], nnzL=info["nnzL"], stored=info["nnzL_stored"], levels=info["nlevels"],
                     upd=info["update_bytes"], perm_ok=sorted(perm.tolist()) == list(range(len(perm))))
print("RESULT " + json.dumps(out))
"""


def test_sibling_bundles_only_where_children_are_very_many():
    """HIPKKT_BUNDLE_KIDS=0 (no bundles) against the default, host-side symbolic analysis only (the knobs are read once
    per process, hence two child processes).  The BASELINE generators' structures must not move at all; a dense row's
    singleton columns, the slack leaves under a dense A and a transportation LP's hub variables must end up with fewer
    supernodes, the same structural nnz(L), no additional level, and -- for the dense A -- a fraction of the update
    storage (one block per bundle instead of one per leaf)."""
    import json
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env in (("off", {"HIPKKT_BUNDLE_KIDS": "0"}), ("on", {})):
        e = {k: v for k, v in os.environ.items() if not k.startswith("HIPKKT_BUNDLE")}
        r = subprocess.run([sys.executable, "-c", _BUNDLE_SCRIPT.format(root=root)], env=dict(e, **env), cwd=root,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        res[tag] = json.loads(re.search(r"RESULT (.*)", r.stdout).group(1))
    for name in ("cfg2", "cfg3", "cfg5"):
        assert res["on"][name] == res["off"][name], name
    for name in ("arrow", "dense", "lp"):
        on, off = res["on"][name], res["off"][name]
        assert on["perm_ok"] and on["nnzL"] == off["nnzL"], name
        assert on["nsuper"] < (0.95 if name != "lp" else 1.0) * off["nsuper"], (name, on, off)
        assert on["levels"] <= off["levels"], (name, on, off)
    assert res["on"]["dense"]["upd"] < 0.2 * res["off"]["dense"]["upd"], res
