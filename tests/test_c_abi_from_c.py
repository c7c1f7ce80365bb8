"""The boundary is a C ABI: a plain C program (tests/c_abi/kkt_from_c.c -- no Python, no torch) builds against
include/hipkkt.h, links libhipkkt.so and drives levels A, B and C (lazy, host vectors) the way a ccall / cgo / JNI binding would."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "c_abi", "kkt_from_c")


def _build():
    libdir = os.path.join(ROOT, "cuclarabel_amd")
    assert os.path.exists(os.path.join(libdir, "libhipkkt.so")), "build libhipkkt.so first (__graft_entry__.build())"
    cmd = ["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_abi", "kkt_from_c.c"), "-o", EXE, "-L", libdir, "-lhipkkt",
           "-Wl,-rpath," + libdir, "-lm"]
    subprocess.check_call(cmd)


def test_header_compiles_and_links_from_plain_c():
    _build()
    # without a device the product says so and stops (no CPU fallback): exit code 2, "no gfx950 device"
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([EXE], capture_output=True, text=True)
        assert p.returncode == 2 and "no gfx950 device" in p.stderr


@pytest.mark.gpu
def test_levels_A_B_C_driven_from_plain_c():
    _build()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "C ABI OK" in p.stdout
