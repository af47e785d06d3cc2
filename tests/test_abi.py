"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/decaf377_amd.h declares; no compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "decaf377_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(d377_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def libpath():
    from decaf377_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as g
        g.build_native()
    return _native.LIB_PATH


def test_header_symbols_are_exported(libpath):
    from decaf377_amd import _native
    lib = ctypes.CDLL(libpath)
    decl = _declared()
    assert len(decl) >= 23
    for name in decl:
        assert hasattr(lib, name), "missing export " + name
    assert sorted(_native.EXPORTS) == decl


def test_binding_prototypes(libpath):
    from decaf377_amd import _native
    lib = _native.load()
    assert lib.d377_version().startswith(b"decaf377_amd")
    assert isinstance(lib.d377_device_count(), int)


def test_no_cpu_fallback_without_gpu(libpath):
    """Without a GPU, creating a context must fail loudly (never compute on the CPU)."""
    from decaf377_amd import _native, Context
    lib = _native.load()
    if lib.d377_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_native.NativeError):
        Context()


def test_ctx_options_are_validated_before_any_device_is_touched(libpath):
    """d377_ctx_create_ex: a bad option is D377_ERR_ARG whether or not a GPU is there (comb widths: 0, 18, 21, 23; comb_lazy
    0 or 1; size = sizeof(d377_ctx_opts)); a good one gets as far as the device check."""
    from decaf377_amd import _native
    lib = _native.load()
    h = ctypes.c_void_p()
    mk = lambda size, bits, lazy: _native.CtxOpts(size, bits, lazy)
    full = ctypes.sizeof(_native.CtxOpts)
    for opts in (mk(full, 7, 0), mk(full, 24, 0), mk(full, 23, 2), mk(4, 23, 0)):
        assert lib.d377_ctx_create_ex(None, 0, ctypes.byref(opts), ctypes.byref(h)) == -2, lib.d377_last_error()
        assert not h.value
    if lib.d377_device_count() == 0:
        good = mk(full, 18, 1)
        assert lib.d377_ctx_create_ex(None, 0, ctypes.byref(good), ctypes.byref(h)) == -3      # D377_ERR_NO_DEVICE
        with pytest.raises(_native.NativeError):
            from decaf377_amd import Context
            Context(comb_lazy=True)


def test_product_never_touches_oracle():
    """Nothing under decaf377_amd/ may import, link or name anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "decaf377_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".inc", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "d377o_" not in text and "libd377_oracle" not in text and "d377_model" not in text, f


def test_every_dev_export_is_named_in_a_gpu_test():
    """No `_dev` entry point without a `-m gpu` test that touches it: tests/test_gpu_parity.py's
    test_every_dev_export keys its table by export name (and asserts the table equals the export list
    on the GPU box); this is the same check without a GPU."""
    from decaf377_amd import _native
    text = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
    for name in _native.EXPORTS:
        if name.endswith("_dev"):
            assert '"%s"' % name in text, name


def _build_c99(libpath):
    import subprocess
    exe = os.path.join(ROOT, "tests", "c", "abi_c99")
    src = os.path.join(ROOT, "tests", "c", "abi_c99.c")
    libdir = os.path.dirname(libpath)
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
                           "-L" + libdir, "-ldecaf377_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib",
                           "-lamdhip64"])
    return exe


def test_header_is_plain_c99(libpath):
    """include/decaf377_amd.h compiles as C99 with -pedantic -Werror and links against the library: the boundary
    is a C ABI, not a C++ one."""
    import subprocess
    r = subprocess.run([_build_c99(libpath)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "C_ABI_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c99_client_runs(libpath):
    import subprocess
    r = subprocess.run([_build_c99(libpath), "--run"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C_ABI_RUN_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.environ.get("D377_CHECK_ARTEFACTS"), reason="release check (tools/round_artifacts.sh sets D377_CHECK_ARTEFACTS=1); "
                    "bench.py labels a stale record itself")
def test_pmc_record_matches_sources():
    """profiles/pmc_traffic.json (what bench.py replays as roofline.traffic and valu_insts_per_mac) was collected on
    exactly the kernels in the tree: the record carries a hash of decaf377_amd/csrc/* and this test recomputes it.  A
    release check, not a correctness test: it runs when the round's artefacts are collected; between collections
    bench.py compares the hash itself and says "stale" in roofline.traffic_source instead of replaying old counters."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("pmc_summarize", os.path.join(ROOT, "tools", "pmc_summarize.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert rec.get("_sources", {}).get("csrc_sha256") == m.csrc_sha256(), \
        "profiles/pmc_traffic.json predates the current kernels: run tools/collect_pmc.sh on the GPU box and copy it over"
