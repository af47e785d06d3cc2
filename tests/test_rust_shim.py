"""The Rust shim (rust/src/ffi.rs, rust/src/gpu.rs) cannot be compiled here (no cargo / rustc), so this
checks it textually against include/decaf377_amd.h: every declared function is bound with the same name,
arity and argument types; every `ffi::d377_*` call in gpu.rs names a bound function and passes the right
number of arguments; the constants match.  CPU only."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_TO_RUST = {
    "d377_ctx*": "*mut D377Ctx", "const d377_ctx*": "*const D377Ctx", "d377_ctx**": "*mut *mut D377Ctx",
    "const uint8_t*": "*const u8", "uint8_t*": "*mut u8", "const uint64_t*": "*const u64", "uint64_t*": "*mut u64",
    "const int*": "*const c_int", "int*": "*mut c_int", "int": "c_int", "size_t": "usize", "void*": "*mut c_void", "const void*": "*const c_void",
    "const char*": "*const c_char", "void": None, "int64_t": "i64", "int64_t*": "*mut i64",
    "const uint32_t**": "*mut *const u32", "const d377_ctx_opts*": "*const D377CtxOpts",
}


def test_ctx_opts_struct_matches_header():
    """d377_ctx_opts and its #[repr(C)] mirror: the same fields in the same order with the same C types."""
    htext = open(os.path.join(ROOT, "include", "decaf377_amd.h")).read()
    rtext = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    body = re.search(r"typedef struct d377_ctx_opts \{(.*?)\} d377_ctx_opts;", htext, re.S).group(1)
    cfields = [(t.strip(), n) for t, n in re.findall(r"\s*([a-z_0-9 ]+?)\s+([a-z_]+);", body)]
    rbody = re.search(r"#\[repr\(C\)\]\s*pub struct D377CtxOpts \{(.*?)\}", rtext, re.S).group(1)
    rfields = re.findall(r"pub ([a-z_]+): ([a-z_0-9]+),", rbody)
    assert [(C_TO_RUST[t], n) for t, n in cfields] == [(t, n) for n, t in rfields] and len(cfields) == 3



def header_prototypes():
    text = open(os.path.join(ROOT, "include", "decaf377_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(const\s+char\s*\*|int|void)\s+(d377_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        ret = re.sub(r"\s+", " ", m.group(1)).replace(" *", "*")
        args = []
        body = m.group(3).strip()
        if body and body != "void":
            for a in body.split(","):
                a = re.sub(r"\s+", " ", a).strip()
                if a.endswith("]"):                                   # uint64_t xyzt[16]
                    a = re.sub(r"\s*[A-Za-z_0-9]+\[\d+\]$", "*", a)
                else:
                    a = re.sub(r"\s*[A-Za-z_][A-Za-z_0-9]*$", "", a)
                args.append(a.replace(" *", "*"))
        protos[m.group(2)] = (ret, args)
    return protos


def rust_prototypes():
    text = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    block = text[text.index('extern "C" {'):]
    block = block[: block.index("\n}")]
    protos = {}
    for m in re.finditer(r"pub fn (d377_[a-z0-9_]+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", block):
        args = [a.split(":", 1)[1].strip() for a in m.group(2).split(",") if a.strip()]
        protos[m.group(1)] = (m.group(3).strip() if m.group(3) else None, args)
    return protos


def test_extern_block_matches_header():
    h, r = header_prototypes(), rust_prototypes()
    assert len(h) >= 55 and set(h) == set(r), set(h) ^ set(r)
    for name, (ret, args) in h.items():
        r_ret, r_args = r[name]
        assert C_TO_RUST[ret] == r_ret, name
        assert [C_TO_RUST[a] for a in args] == r_args, (name, args, r_args)


def test_constants_match_header():
    htext = open(os.path.join(ROOT, "include", "decaf377_amd.h")).read()
    rtext = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    consts = dict(re.findall(r"#define (D377_[A-Z0-9_]+) \(?(-?\d+)\)?", htext))
    assert len(consts) >= 20
    for k, v in consts.items():
        assert re.search(r"pub const %s: c_int = %s;" % (k, v), rtext), k


def _call_args(text, start):
    """number of top-level arguments of the call whose '(' is at text[start]"""
    depth, n, i, seen = 0, 0, start, False
    while True:
        c = text[i]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return n + (1 if seen else 0)
        elif c == "," and depth == 1:
            n += 1
            seen = False
        elif depth >= 1 and not c.isspace():
            seen = True
        i += 1


def test_wrappers_call_bound_functions_with_right_arity():
    text = open(os.path.join(ROOT, "rust", "src", "gpu.rs")).read()
    text = re.sub(r"//[^\n]*", "", text)
    h = header_prototypes()
    calls = [(m.group(1), _call_args(text, m.end() - 1)) for m in re.finditer(r"ffi::(d377_[a-z0-9_]+)\(", text)]
    assert len(calls) >= 44
    for name, nargs in calls:
        assert name in h, name
        assert nargs == len(h[name][1]), (name, nargs, len(h[name][1]))
    # the wrappers cover the crate's public surface on this path (src/lib.rs:8-31), not only a sample
    used = {n for n, _ in calls}
    for must in ("d377_batch_decompress", "d377_batch_compress", "d377_batch_roundtrip", "d377_batch_scalar_mul_base",
                 "d377_batch_scalar_mul_var", "d377_batch_encode_to_curve", "d377_batch_hash_to_curve", "d377_batch_add",
                 "d377_batch_double", "d377_batch_neg", "d377_batch_eq", "d377_batch_is_identity", "d377_batch_to_affine",
                 "d377_msm", "d377_msm_encoded", "d377_batch_sqrt_ratio_zeta_ex", "d377_batch_fq_op",
                 "d377_batch_fq_from_wide_bytes", "d377_batch_encode_to_curve_wide", "d377_batch_fq_from_bytes_checked",
                 "d377_batch_fq_to_bytes", "d377_batch_fr_from_le_bytes_mod_order", "d377_batch_fr_from_bytes_checked",
                 "d377_batch_scalar_mul_var_dev", "d377_msm_dev", "d377_batch_sharded_dev",
                 "d377_batch_scalar_mul_var_element", "d377_batch_scalar_mul_base_element", "d377_batch_compress_to_field",
                 "d377_batch_encode_to_curve_element", "d377_batch_hash_to_curve_element", "d377_batch_fr_op",
                 "d377_batch_fr_from_wide_bytes", "d377_batch_scalar_mul_var_element_dev", "d377_batch_fr_op_dev"):
        assert must in used, must
    # Projective::new_unchecked takes the crate's Fq (src/ark_curve/edwards.rs:21), never its private inner value
    assert ".0 .0" not in text and "new_unchecked(fq_limbs(" in text


def test_generated_ffi_is_current():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    protos = g.prototypes(open(os.path.join(ROOT, "include", "decaf377_amd.h")).read())
    text = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    for name, _, params in protos:
        assert "pub fn %s(" % name in text
