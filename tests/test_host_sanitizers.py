"""AddressSanitizer + UndefinedBehaviorSanitizer over the product's HOST code (csrc/kkt_assembly.cpp, csrc/ordering.cpp,
csrc/symbolic.cpp: the set-up path of hipkkt_kkt_create / hipkkt_symbolic_analyse).  The GPU pool offers no sanitizer, so
the kernels rest on the parity suite and the host code on this: tests/sanitize/host_driver.cpp is built from the same
sources with g++ -fsanitize=address,undefined, fed the small BASELINE configurations and the small structure zoo, and
checks the assembly maps, the permutation, the supernode partition and the level schedule for consistency on the way."""
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from cuclarabel_amd import problems
from cuclarabel_amd.cones import cone_kinds_dims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuclarabel_amd", "csrc")
SOURCES = [os.path.join(ROOT, "tests", "sanitize", "host_driver.cpp")] + [os.path.join(CSRC, f) for f in
                                                                          ("kkt_assembly.cpp", "symbolic.cpp", "ordering.cpp")]


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("sanitize") / "host_driver")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I", CSRC, "-I", os.path.join(ROOT, "include")] + SOURCES + ["-o", exe]
    subprocess.check_call(cmd)
    return exe


def _write(path, pb, leaf=0):
    P = sp.triu(sp.csc_matrix(pb.P), format="csc")
    P.sort_indices()
    A = sp.csc_matrix(pb.A)
    A.sort_indices()
    kinds, dims = cone_kinds_dims(pb.cones)
    with open(path, "wb") as f:
        np.array([pb.n, pb.m, len(pb.cones), P.nnz, A.nnz, leaf], dtype=np.int64).tofile(f)
        P.indptr.astype(np.int64).tofile(f)
        P.indices.astype(np.int64).tofile(f)
        P.data.astype(np.float64).tofile(f)
        A.indptr.astype(np.int64).tofile(f)
        A.indices.astype(np.int64).tofile(f)
        A.data.astype(np.float64).tofile(f)
        kinds.astype(np.int32).tofile(f)
        dims.astype(np.int64).tofile(f)


CASES = [("cfg1", problems.config1, 0),
         ("cfg2_n3000", lambda: problems.config2(n=3000), 300),
         ("cfg2_n8000_longrange", lambda: problems.config2(n=8000, long_range_frac=0.01), 0),
         ("cfg3_small", lambda: problems.config3(nblocks=4, blk=60), 0),
         ("cfg5_small", lambda: problems.config5(n=400, npsd=8, nsoc=6), 200)] + \
        [("zoo_" + name, maker, 0) for name, maker in problems.ZOO_SMALL]


def test_host_code_is_clean_under_asan_and_ubsan(driver, tmp_path):
    files = []
    for name, maker, leaf in CASES:
        path = str(tmp_path / (name + ".bin"))
        _write(path, maker(), leaf)
        files.append(path)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([driver] + files, capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    assert "HOST SANITIZER DRIVER OK" in p.stdout
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr and "LeakSanitizer" not in p.stderr, p.stderr[-4000:]
    assert p.stdout.count(" ordering ") >= 2 * len(CASES)


def test_the_driver_notices_a_bad_pattern(driver, tmp_path):
    """The assembly rejects out-of-range indices with an exception (no out-of-bounds write): the driver exits 1, cleanly."""
    pb = problems.config2(n=300)
    path = str(tmp_path / "bad.bin")
    _write(path, pb)
    raw = bytearray(open(path, "rb").read())
    P = sp.triu(sp.csc_matrix(pb.P), format="csc")
    off = 8 * 6 + 8 * (pb.n + 1) + 8 * P.nnz + 8 * P.nnz + 8 * (pb.n + 1)      # first row index of A
    raw[off:off + 8] = np.int64(pb.m + 5).tobytes()
    open(path, "wb").write(bytes(raw))
    p = subprocess.run([driver, path], capture_output=True, text=True, timeout=120)
    assert p.returncode == 1 and "host_driver:" in p.stderr, (p.returncode, p.stderr[-2000:])
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
