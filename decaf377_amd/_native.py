"""ctypes binding of the C ABI declared in include/decaf377_amd.h.

The shared library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950) as
decaf377_amd/lib/libdecaf377_amd.so.  There is no CPU fallback: if the library is missing
or no MI355X is visible, loading / context creation raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# D377_LIB lets a developer A/B another build of the same library (tools/ab_bench.sh)
LIB_PATH = os.environ.get("D377_LIB") or os.path.join(_HERE, "lib", "libdecaf377_amd.so")

# every symbol include/decaf377_amd.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "d377_version", "d377_device_count", "d377_last_error",
    "d377_ctx_create", "d377_ctx_destroy", "d377_ctx_num_devices", "d377_ctx_device_id",
    "d377_batch_sqrt_ratio_zeta", "d377_batch_decompress", "d377_batch_compress", "d377_batch_roundtrip",
    "d377_batch_scalar_mul_base", "d377_batch_scalar_mul_var", "d377_batch_encode_to_curve",
    "d377_batch_hash_to_curve", "d377_batch_add", "d377_batch_sub", "d377_batch_double", "d377_batch_eq",
    "d377_batch_add_dev", "d377_batch_sub_dev", "d377_batch_double_dev", "d377_batch_eq_dev",
    "d377_batch_fr_from_le_bytes_mod_order", "d377_batch_fr_from_bytes_checked",
    "d377_batch_fq_op", "d377_batch_fq_op_dev", "d377_batch_fq_from_bytes_checked", "d377_batch_fq_to_bytes",
    "d377_batch_neg", "d377_batch_is_identity", "d377_batch_neg_dev", "d377_batch_is_identity_dev",
    "d377_identity", "d377_generator",
    "d377_batch_fq_from_wide_bytes", "d377_batch_encode_to_curve_wide", "d377_batch_to_affine",
    "d377_batch_fq_from_wide_bytes_dev", "d377_batch_encode_to_curve_wide_dev", "d377_batch_to_affine_dev",
    "d377_msm", "d377_msm_encoded", "d377_msm_dev", "d377_msm_encoded_dev", "d377_sum_elements_dev",
    "d377_batch_sqrt_ratio_zeta_dev", "d377_batch_decompress_dev", "d377_batch_compress_dev",
    "d377_batch_roundtrip_dev", "d377_batch_scalar_mul_base_dev", "d377_batch_scalar_mul_var_dev",
    "d377_batch_encode_to_curve_dev", "d377_batch_hash_to_curve_dev",
    "d377_batch_sqrt_ratio_zeta_ex", "d377_batch_sqrt_ratio_zeta_ex_dev", "d377_batch_sharded_dev",
    "d377_ctx_invariant_failures", "d377_ctx_chunk_residency", "d377_ctx_set_tuning", "d377_ctx_get_tuning",
    "d377_ctx_health", "d377_ctx_reset_scratch", "d377_debug_poison_pool", "d377_ctx_peer_access",
    "d377_batch_scalar_mul_var_element", "d377_batch_scalar_mul_base_element", "d377_batch_compress_to_field",
    "d377_batch_encode_to_curve_element", "d377_batch_hash_to_curve_element",
    "d377_batch_scalar_mul_var_element_dev", "d377_batch_scalar_mul_base_element_dev", "d377_batch_compress_to_field_dev",
    "d377_batch_encode_to_curve_element_dev", "d377_batch_hash_to_curve_element_dev",
    "d377_batch_fr_op", "d377_batch_fr_op_dev", "d377_batch_fr_from_wide_bytes", "d377_batch_fr_from_wide_bytes_dev",
    "d377_batch_fq_from_bytes_checked_dev", "d377_batch_fq_to_bytes_dev", "d377_batch_fr_from_le_bytes_mod_order_dev",
    "d377_batch_fr_from_bytes_checked_dev", "d377_ctx_starved_counter_dev",
    "d377_ctx_create_ex", "d377_ctx_comb_info",
    "d377_batch_msm_small", "d377_batch_msm_small_encoded", "d377_batch_msm_small_dev", "d377_batch_msm_small_encoded_dev",
]


class CtxOpts(ctypes.Structure):
    """d377_ctx_opts (include/decaf377_amd.h)."""
    _fields_ = [("size", ctypes.c_size_t), ("comb_bits", ctypes.c_int), ("comb_lazy", ctypes.c_int)]


_lib = None


class NativeError(RuntimeError):
    pass


class StarvedError(NativeError):
    """D377_ERR_STARVED: workgroups of the call found no free lane set for 10 s and wrote no output (the call's
    outputs are invalid; Context.health / Context.reset_scratch)."""


ERR_STARVED = -5


def load():
    """Loads libdecaf377_amd.so; raises NativeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            "decaf377_amd: %s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (soname
    # libamdhip64.so.7, the same as /opt/rocm's).  If this library pulled in the system copy
    # first, a later `import torch` would map a second runtime that cannot open the GPU.
    # Importing torch first makes both share torch's copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.d377_version.restype = ctypes.c_char_p
    lib.d377_last_error.restype = ctypes.c_char_p
    lib.d377_device_count.restype = i32
    lib.d377_ctx_create.argtypes = [ctypes.POINTER(i32), i32, ctypes.POINTER(vp)]
    lib.d377_ctx_create_ex.argtypes = [ctypes.POINTER(i32), i32, ctypes.POINTER(CtxOpts), ctypes.POINTER(vp)]
    lib.d377_ctx_create_ex.restype = i32
    lib.d377_ctx_comb_info.argtypes = [vp, i32, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_uint64)]
    lib.d377_ctx_comb_info.restype = i32
    lib.d377_ctx_destroy.argtypes = [vp]
    lib.d377_ctx_destroy.restype = None
    lib.d377_ctx_num_devices.argtypes = [vp]
    lib.d377_ctx_device_id.argtypes = [vp, i32]
    lib.d377_ctx_peer_access.argtypes = [vp, i32, i32]
    lib.d377_ctx_peer_access.restype = i32
    lib.d377_ctx_invariant_failures.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_uint64)]
    lib.d377_ctx_invariant_failures.restype = i32
    lib.d377_ctx_chunk_residency.argtypes = [vp, i32, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    lib.d377_ctx_chunk_residency.restype = i32
    lib.d377_ctx_health.argtypes = [vp, i32, ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.d377_ctx_health.restype = i32
    lib.d377_ctx_reset_scratch.argtypes = [vp, i32, ctypes.POINTER(i32)]
    lib.d377_ctx_reset_scratch.restype = i32
    lib.d377_debug_poison_pool.argtypes = [vp, i32, i32]
    lib.d377_debug_poison_pool.restype = i32
    lib.d377_ctx_starved_counter_dev.argtypes = [vp, i32, ctypes.POINTER(vp)]
    lib.d377_ctx_starved_counter_dev.restype = i32
    lib.d377_ctx_set_tuning.argtypes = [vp, i32, ctypes.c_int64]
    lib.d377_ctx_set_tuning.restype = i32
    lib.d377_ctx_get_tuning.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_int64)]
    lib.d377_ctx_get_tuning.restype = i32
    host = {
        "d377_batch_sqrt_ratio_zeta": [vp, vp, vp, sz, vp, vp],
        "d377_batch_decompress": [vp, vp, sz, vp, vp],
        "d377_batch_compress": [vp, vp, sz, vp],
        "d377_batch_roundtrip": [vp, vp, sz, vp, vp],
        "d377_batch_scalar_mul_base": [vp, vp, sz, vp],
        "d377_batch_scalar_mul_var": [vp, vp, vp, sz, vp, vp],
        "d377_batch_encode_to_curve": [vp, vp, sz, vp],
        "d377_batch_hash_to_curve": [vp, vp, vp, sz, vp],
        "d377_batch_add": [vp, vp, vp, sz, vp],
        "d377_batch_sub": [vp, vp, vp, sz, vp],
        "d377_batch_double": [vp, vp, sz, vp],
        "d377_batch_eq": [vp, vp, vp, sz, vp],
        "d377_batch_neg": [vp, vp, sz, vp],
        "d377_batch_is_identity": [vp, vp, sz, vp],
        "d377_batch_scalar_mul_var_element": [vp, vp, vp, sz, vp],
        "d377_batch_scalar_mul_base_element": [vp, vp, sz, vp],
        "d377_batch_compress_to_field": [vp, vp, sz, vp],
        "d377_batch_encode_to_curve_element": [vp, vp, sz, vp],
        "d377_batch_hash_to_curve_element": [vp, vp, vp, sz, vp],
        "d377_batch_fq_from_bytes_checked": [vp, vp, sz, vp, vp],
        "d377_batch_fq_to_bytes": [vp, vp, sz, vp],
        "d377_batch_fr_from_le_bytes_mod_order": [vp, vp, sz, vp],
        "d377_batch_fr_from_bytes_checked": [vp, vp, sz, vp, vp],
    }
    for name, args in host.items():
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = i32
        dev = getattr(lib, name + "_dev")
        dev.argtypes = [vp, i32, vp] + args[1:]
        dev.restype = i32
    lib.d377_batch_fq_from_wide_bytes.argtypes = [vp, vp, sz, sz, vp]
    lib.d377_batch_encode_to_curve_wide.argtypes = [vp, vp, sz, sz, vp]
    lib.d377_batch_to_affine.argtypes = [vp, vp, sz, vp]
    lib.d377_batch_fr_from_wide_bytes.argtypes = [vp, vp, sz, sz, vp]
    lib.d377_batch_fr_from_wide_bytes_dev.argtypes = [vp, i32, vp, vp, sz, sz, vp]
    lib.d377_batch_fr_op.argtypes = [vp, i32, vp, vp, sz, vp, vp]
    lib.d377_batch_fr_op_dev.argtypes = [vp, i32, vp, i32, vp, vp, sz, vp, vp]
    for name in ("d377_batch_fr_from_wide_bytes", "d377_batch_fr_from_wide_bytes_dev", "d377_batch_fr_op", "d377_batch_fr_op_dev"):
        getattr(lib, name).restype = i32
    lib.d377_batch_fq_from_wide_bytes_dev.argtypes = [vp, i32, vp, vp, sz, sz, vp]
    lib.d377_batch_encode_to_curve_wide_dev.argtypes = [vp, i32, vp, vp, sz, sz, vp]
    lib.d377_batch_to_affine_dev.argtypes = [vp, i32, vp, vp, sz, vp]
    for name in ("d377_batch_fq_from_wide_bytes", "d377_batch_encode_to_curve_wide", "d377_batch_to_affine"):
        getattr(lib, name).restype = i32
        getattr(lib, name + "_dev").restype = i32
    lib.d377_batch_fr_from_le_bytes_mod_order.argtypes = [vp, vp, sz, vp]
    lib.d377_batch_fr_from_bytes_checked.argtypes = [vp, vp, sz, vp, vp]
    lib.d377_batch_fr_from_le_bytes_mod_order.restype = i32
    lib.d377_batch_fr_from_bytes_checked.restype = i32
    lib.d377_batch_fq_op.argtypes = [vp, i32, vp, vp, sz, vp, vp]
    lib.d377_batch_fq_op_dev.argtypes = [vp, i32, vp, i32, vp, vp, sz, vp, vp]
    lib.d377_batch_fq_from_bytes_checked.argtypes = [vp, vp, sz, vp, vp]
    lib.d377_batch_fq_to_bytes.argtypes = [vp, vp, sz, vp]
    for name in ("d377_batch_fq_op", "d377_batch_fq_op_dev", "d377_batch_fq_from_bytes_checked", "d377_batch_fq_to_bytes"):
        getattr(lib, name).restype = i32
    lib.d377_batch_sqrt_ratio_zeta_ex.argtypes = [vp, i32, vp, vp, sz, vp, vp]
    lib.d377_batch_sqrt_ratio_zeta_ex_dev.argtypes = [vp, i32, vp, i32, vp, vp, sz, vp, vp]
    lib.d377_batch_sharded_dev.argtypes = [vp, i32, vp, i32, vp, vp, sz, vp, vp]
    for name in ("d377_batch_sqrt_ratio_zeta_ex", "d377_batch_sqrt_ratio_zeta_ex_dev", "d377_batch_sharded_dev"):
        getattr(lib, name).restype = i32
    lib.d377_identity.argtypes = [vp]
    lib.d377_identity.restype = None
    lib.d377_generator.argtypes = [vp]
    lib.d377_generator.restype = None
    lib.d377_msm.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.d377_msm_encoded.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    lib.d377_msm_dev.argtypes = [vp, i32, vp, vp, vp, sz, vp, vp]
    lib.d377_msm_encoded_dev.argtypes = [vp, i32, vp, vp, vp, sz, vp, vp, vp]
    lib.d377_sum_elements_dev.argtypes = [vp, i32, vp, vp, sz, vp, vp]
    lib.d377_batch_msm_small.argtypes = [vp, vp, vp, sz, sz, vp, vp]
    lib.d377_batch_msm_small_encoded.argtypes = [vp, vp, vp, sz, sz, vp, vp, vp]
    lib.d377_batch_msm_small_dev.argtypes = [vp, i32, vp, vp, vp, sz, sz, vp, vp]
    lib.d377_batch_msm_small_encoded_dev.argtypes = [vp, i32, vp, vp, vp, sz, sz, vp, vp, vp]
    for name in ("d377_msm", "d377_msm_encoded", "d377_msm_dev", "d377_msm_encoded_dev", "d377_sum_elements_dev",
                 "d377_batch_msm_small", "d377_batch_msm_small_encoded", "d377_batch_msm_small_dev", "d377_batch_msm_small_encoded_dev"):
        getattr(lib, name).restype = i32
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise (StarvedError if rc == ERR_STARVED else NativeError)("decaf377_amd native call failed (%d): %s" % (rc, load().d377_last_error().decode()))
