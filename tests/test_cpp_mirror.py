"""The reference's integration tests re-expressed against the C++ host mirror
(include/decaf377_amd.hpp, tests/cpp/reference_style.cpp).  The compile check runs on CPU;
running the binary needs the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "reference_style")


def _build():
    from decaf377_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as g
        g.build_native()
    libdir = os.path.dirname(_native.LIB_PATH)
    src = os.path.join(ROOT, "tests", "cpp", "reference_style.cpp")
    hdrs = [os.path.join(ROOT, "include", f) for f in ("decaf377_amd.hpp", "decaf377_amd.h")]
    if not os.path.exists(BIN) or any(os.path.getmtime(f) > os.path.getmtime(BIN) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", BIN,
                               "-L" + libdir, "-ldecaf377_amd", "-Wl,-rpath," + libdir,
                               "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"])
    return BIN


def test_cpp_mirror_compiles_and_links():
    assert os.path.exists(_build())


@pytest.mark.gpu
def test_cpp_reference_style_suite():
    r = subprocess.run([_build()], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CPP_MIRROR_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
