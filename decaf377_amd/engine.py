"""Host-side mirror of the reference interface for the hot path, batch-shaped.

Reference (crate decaf377 v0.10.1)              here (one call = one batch)
  Encoding(pub [u8; 32])                          Encoding(array[n, 32] u8)
  Encoding::vartime_decompress()                  Encoding.vartime_decompress() -> (Element, status)
  Element::vartime_compress()                     Element.vartime_compress() -> Encoding
  Element::encode_to_curve(&Fq)                   Element.encode_to_curve(Fq) -> Encoding
  Element::hash_to_curve(&Fq, &Fq)                Element.hash_to_curve(Fq, Fq) -> Encoding
  Element::GENERATOR * Fr                         Element.generator_mul(Fr) -> Encoding
  element * Fr  (Encoding in, Encoding out)       Encoding.scalar_mul(Fr) -> (Encoding, status)
  Fq::sqrt_ratio_zeta(&num, &den)                 Fq.sqrt_ratio_zeta(num, den) -> (was_square, Fq)
  EncodingError::InvalidEncoding                  status byte 1 (0 = Ok) / EncodingError when raised

Arrays may be numpy (host path: the library stages through HBM) or torch CUDA tensors (device
path: no copies, launched on torch's current stream).  torch is plumbing only: device memory
and streams.  There is no CPU implementation behind these classes."""
import ctypes

import numpy as np

from . import _native


class EncodingError(ValueError):
    """src/error.rs:1-5. `InvalidEncoding` is reported per element as status 1."""


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _rows(a):
    if a.ndim < 1:
        raise ValueError("expected a [n, k] array of packed records")
    return int(a.shape[0])


# row layouts of the packed records (trailing shape, element kind): the native code reads and writes
# n * (that many bytes) unconditionally, so every array is checked against these before a call
ENC = ((32,), "u8")          # Encoding / Fq / Fr byte strings
ELEM = ((16,), "u64")        # Element: X, Y, Z, T as 4 Montgomery limbs each
FQM = ((4,), "u64")          # Fq as 4 Montgomery limbs
AFF = ((8,), "u64")          # affine x, y
FLAG = ((), "u8")            # one status / flag byte per element
SQRT_ROOT = {"ark": 0, "arkworks": 0, "min_curve": 1}
# D377_TUNE_* keys of d377_ctx_set_tuning (include/decaf377_amd.h, "Developer interface")
TUNE_KEYS = {"small_max": 0, "decompress_chunked_min": 1, "fb_wide": 2, "fb_k": 3, "affine_blocks_per_cu": 4, "msm_window": 5,
             "msm_seg": 6, "msm_small_max": 7, "msm_slices": 8, "msm_red": 9, "msm_skip": 10, "msm_chunked_sums": 11,
             "msm_enc_chunked_min": 12, "chunk_per_lane": 13, "msm_tiny_max": 14, "tiny_max": 15, "msm_sort_packed": 16}
SHARD_OPS = {"sqrt_ratio_zeta": 0, "decompress": 1, "compress": 2, "roundtrip": 3, "scalar_mul_base": 4,
             "scalar_mul_var": 5, "encode_to_curve": 6, "hash_to_curve": 7, "scalar_mul_var_element": 8,
             "scalar_mul_base_element": 9}


def _check(a, spec, n, what, device=None):
    """Raises ValueError unless `a` is a contiguous-able [n, *tail] array of the right element type
    (and, for torch tensors, lives on `device`)."""
    tail, kind = spec
    shape = tuple(int(x) for x in a.shape)
    if shape != (n,) + tail:
        raise ValueError("%s: expected shape %s, got %s" % (what, (n,) + tail, shape))
    if _is_torch(a):
        import torch
        ok = a.dtype == torch.uint8 if kind == "u8" else a.dtype in (torch.int64, torch.uint64)
        if device is not None and a.device != device:
            raise ValueError("%s: tensor on %s, expected %s" % (what, a.device, device))
    else:
        ok = a.dtype == np.uint8 if kind == "u8" else a.dtype in (np.uint64, np.int64)
    if not ok:
        raise ValueError("%s: expected %s elements, got %s" % (what, "uint8" if kind == "u8" else "uint64/int64", a.dtype))


class Context:
    """Owns the device tables and scratch for one or more GPUs (d377_ctx)."""

    def __init__(self, device_ids=None, comb_bits=0, comb_lazy=False):
        """comb_bits: width of the fixed-base comb, 0 = the library's default (23) or 18 / 21 / 23 (0.24 / 1.6 / 5.9 GB per
        device); comb_lazy: build it on the first fixed-base call instead of now (d377_ctx_create_ex) -- a context that
        never multiplies by the generator then holds 0.73 GB per device instead of 6.7."""
        self._lib = _native.load()
        self._h = ctypes.c_void_p()
        if device_ids is None:
            ids, n = None, 0
        else:
            ids, n = (ctypes.c_int * len(device_ids))(*device_ids), len(device_ids)
        if comb_bits == 0 and not comb_lazy:
            _native.check(self._lib.d377_ctx_create(ids, n, ctypes.byref(self._h)))
        else:
            opts = _native.CtxOpts(ctypes.sizeof(_native.CtxOpts), int(comb_bits), 1 if comb_lazy else 0)
            _native.check(self._lib.d377_ctx_create_ex(ids, n, ctypes.byref(opts), ctypes.byref(self._h)))
        self.device_ids = [self._lib.d377_ctx_device_id(self._h, i)
                           for i in range(self._lib.d377_ctx_num_devices(self._h))]

    def invariant_failures(self, dev=0):
        """(checks compiled in?, count): d377_ctx_invariant_failures -- the -DD377_CHECK_INVARIANTS build's
        counterpart of the reference's debug assertions (src/min_curve/element.rs:104-110)."""
        c = ctypes.c_uint64(0)
        rc = self._lib.d377_ctx_invariant_failures(self._h, dev, ctypes.byref(c))
        if rc < 0:
            _native.check(rc)
        return rc == 1, int(c.value)

    def comb_info(self, dev=0):
        """(width in bits, built yet?, table bytes) of the device's fixed-base comb: d377_ctx_comb_info."""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_uint64(0)
        _native.check(self._lib.d377_ctx_comb_info(self._h, dev, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return int(a.value), bool(b.value), int(c.value)

    def chunk_residency(self, dev=0):
        """(lane sets per CU, largest residency of a set-claiming kernel per CU, LDS padding in bytes):
        d377_ctx_chunk_residency -- what d377_ctx_create verified with the occupancy query."""
        a, b, c = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _native.check(self._lib.d377_ctx_chunk_residency(self._h, dev, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def peer_access(self, a, b):
        """d377_ctx_peer_access: 2 = distinct GPUs with peer access enabled, 1 = the same GPU listed twice, 0 = none."""
        return int(self._lib.d377_ctx_peer_access(self._h, a, b))

    def health(self, dev=0):
        """(lane sets claimed now, workgroups that waited > 0.25 s for a set, workgroups that gave up): d377_ctx_health."""
        a, b, c = ctypes.c_int(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
        _native.check(self._lib.d377_ctx_health(self._h, dev, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def reset_scratch(self, dev=0):
        """Frees lane sets leaked by a launch that died (d377_ctx_reset_scratch) -> how many it freed."""
        a = ctypes.c_int(0)
        _native.check(self._lib.d377_ctx_reset_scratch(self._h, dev, ctypes.byref(a)))
        return int(a.value)

    def starved_counter(self, dev=0):
        """The gave-up counter as a 1-element uint32 CUDA tensor that aliases the library's word
        (d377_ctx_starved_counter_dev): a `_dev` caller copies it on its stream before and after a call and compares
        once the stream has been synchronised -- a call whose kernels starved has left output records unwritten."""
        import torch
        p = ctypes.c_void_p()
        _native.check(self._lib.d377_ctx_starved_counter_dev(self._h, dev, ctypes.byref(p)))

        class _Word:
            __cuda_array_interface__ = {"shape": (1,), "typestr": "<i4", "data": (int(p.value), False), "version": 2}
        return torch.as_tensor(_Word(), device=torch.device("cuda", self.device_ids[dev]))

    def _debug_poison_pool(self, dev=0, sets=-1):
        _native.check(self._lib.d377_debug_poison_pool(self._h, dev, sets))

    def set_tuning(self, key, value=None):
        """d377_ctx_set_tuning: developer override of one launch rule (TUNE_KEYS); value None restores the built-in rule."""
        _native.check(self._lib.d377_ctx_set_tuning(self._h, TUNE_KEYS[key], -1 if value is None else int(value)))

    def get_tuning(self, key):
        v = ctypes.c_int64(0)
        _native.check(self._lib.d377_ctx_get_tuning(self._h, TUNE_KEYS[key], ctypes.byref(v)))
        return None if v.value < 0 else int(v.value)

    def tuning(self, **kv):
        """Context manager: the given overrides for the calls made inside, the previous values afterwards."""
        ctx = self

        class _Scope:
            def __enter__(self):
                self.old = {k: ctx.get_tuning(k) for k in kv}
                for k, v in kv.items():
                    ctx.set_tuning(k, v)
                return ctx

            def __exit__(self, *exc):
                for k, v in self.old.items():
                    ctx.set_tuning(k, v)
                return False

        return _Scope()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.d377_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ---------------------------------------------------------------------------
    def _dev_index(self, dev):
        if dev.type != "cuda":
            raise _native.NativeError("torch tensors must live on the GPU (there is no CPU path)")
        if dev.index not in self.device_ids:
            raise _native.NativeError("tensor on cuda:%s but context owns %s" % (dev.index, self.device_ids))
        return self.device_ids.index(dev.index)

    def _run(self, name, ins, in_specs, out_specs, outs=None, pre=(), sharded_op=None, in_slots=None):
        """ins: input arrays (specs in `in_specs`); out_specs: row layouts of the outputs.  Host or
        device path by the type of the first input; `pre` = extra integer arguments that precede the
        buffers in the C signature; `outs` lets a caller reuse output arrays (checked like inputs).
        sharded_op: run through d377_batch_sharded_dev (HBM-resident batch split over the context's GPUs).
        in_slots: input-pointer parameters of the C signature when some are unused by this call (passed NULL)."""
        n = _rows(ins[0])
        np_dt = {"u8": np.uint8, "u64": np.uint64}
        if _is_torch(ins[0]):
            import torch
            dev = ins[0].device
            di = self._dev_index(dev)
            for a, spec in zip(ins, in_specs):
                _check(a, spec, n, name + " input", dev)
            tdt = {"u8": torch.uint8, "u64": torch.int64}
            ins = [t.contiguous() for t in ins]
            if outs is None:
                outs = [torch.empty((n,) + tail, dtype=tdt[k], device=dev) for tail, k in out_specs]
            else:
                if len(outs) != len(out_specs):
                    raise ValueError("%s: expected %d output arrays" % (name, len(out_specs)))
                for a, spec in zip(outs, out_specs):
                    _check(a, spec, n, name + " output", dev)
                    if not a.is_contiguous():
                        raise ValueError("%s output must be contiguous" % name)
            stream = torch.cuda.current_stream(dev).cuda_stream
            bufs = [ctypes.c_void_p(t.data_ptr()) for t in ins]
            obufs = [ctypes.c_void_p(t.data_ptr()) for t in outs]
            if sharded_op is not None:
                bufs += [None] * (2 - len(bufs))
                obufs += [None] * (2 - len(obufs))
                _native.check(self._lib.d377_batch_sharded_dev(self._h, di, ctypes.c_void_p(stream), sharded_op,
                                                               bufs[0], bufs[1], ctypes.c_size_t(n), obufs[0], obufs[1]))
                return outs
            bufs += [None] * ((in_slots or len(bufs)) - len(bufs))
            args = [self._h, di, ctypes.c_void_p(stream)] + list(pre) + bufs + [ctypes.c_size_t(n)] + obufs
            _native.check(getattr(self._lib, name + "_dev")(*args))
            return outs
        if sharded_op is not None:
            raise ValueError("the sharded path takes torch CUDA tensors (host arrays are sharded by the host path)")
        ins = [np.ascontiguousarray(a) for a in ins]
        for a, spec in zip(ins, in_specs):
            _check(a, spec, n, name + " input")
        if outs is None:
            # fresh arrays of this size are mmap'ed anew on every call and fault their pages in during the copy back;
            # a caller that cares passes `outs` it allocated once (tools/host_path_bench.py: 2^20 round trips 7-9 ms vs 4.8 ms)
            outs = [np.empty((n,) + tail, dtype=np_dt[k]) for tail, k in out_specs]
        else:
            if len(outs) != len(out_specs):
                raise ValueError("%s: expected %d output arrays" % (name, len(out_specs)))
            for a, spec in zip(outs, out_specs):
                _check(a, spec, n, name + " output")
                if not a.flags["C_CONTIGUOUS"]:
                    raise ValueError("%s output must be contiguous" % name)
        args = [self._h] + list(pre) + [a.ctypes.data_as(ctypes.c_void_p) for a in ins]
        args += [None] * ((in_slots or len(ins)) - len(ins)) + [ctypes.c_size_t(n)]
        args += [a.ctypes.data_as(ctypes.c_void_p) for a in outs]
        _native.check(getattr(self._lib, name)(*args))
        return outs

    # -- the batch operations (C ABI names) ---------------------------------------------------
    def sqrt_ratio_zeta(self, num32, den32, outs=None, root="ark"):
        """root: "ark" (Sarkar tables, src/ark_curve/invsqrt.rs:75-166) or "min_curve" (Tonelli-Shanks
        seeded with 11^m, src/min_curve/invsqrt.rs:11-95): same flags, root negated about half the time."""
        return self._run("d377_batch_sqrt_ratio_zeta_ex", [num32, den32], [ENC, ENC], [ENC, FLAG], outs,
                         pre=(ctypes.c_int(SQRT_ROOT[root]),))

    def decompress(self, enc32, outs=None):
        return self._run("d377_batch_decompress", [enc32], [ENC], [ELEM, FLAG], outs)

    def compress(self, xyzt, outs=None):
        return self._run("d377_batch_compress", [xyzt], [ELEM], [ENC], outs)[0]

    def roundtrip(self, enc32, outs=None):
        return self._run("d377_batch_roundtrip", [enc32], [ENC], [ENC, FLAG], outs)

    def scalar_mul_base(self, scalar32, outs=None):
        return self._run("d377_batch_scalar_mul_base", [scalar32], [ENC], [ENC], outs)[0]

    def scalar_mul_var(self, enc32, scalar32, outs=None):
        return self._run("d377_batch_scalar_mul_var", [enc32, scalar32], [ENC, ENC], [ENC, FLAG], outs)

    def encode_to_curve(self, fq32, outs=None):
        return self._run("d377_batch_encode_to_curve", [fq32], [ENC], [ENC], outs)[0]

    def hash_to_curve(self, r1_32, r2_32, outs=None):
        return self._run("d377_batch_hash_to_curve", [r1_32, r2_32], [ENC, ENC], [ENC], outs)[0]

    # The reference's own signatures: Elements in and out, no encoding step (src/min_curve/ops.rs:89-95,
    # src/min_curve/element.rs:163-181, 190-244).  A scalar multiplication returns some extended representative
    # of the reference's group element: compare with eq() or through compress().
    def scalar_mul_var_element(self, p_xyzt, scalar32, outs=None):
        return self._run("d377_batch_scalar_mul_var_element", [p_xyzt, scalar32], [ELEM, ENC], [ELEM], outs)[0]

    def scalar_mul_base_element(self, scalar32, outs=None):
        return self._run("d377_batch_scalar_mul_base_element", [scalar32], [ENC], [ELEM], outs)[0]

    def compress_to_field(self, p_xyzt, outs=None):
        """Element::vartime_compress_to_field -> [n, 4] u64 Montgomery limbs of s."""
        return self._run("d377_batch_compress_to_field", [p_xyzt], [ELEM], [FQM], outs)[0]

    def encode_to_curve_element(self, fq32, outs=None):
        return self._run("d377_batch_encode_to_curve_element", [fq32], [ENC], [ELEM], outs)[0]

    def hash_to_curve_element(self, r1_32, r2_32, outs=None):
        return self._run("d377_batch_hash_to_curve_element", [r1_32, r2_32], [ENC, ENC], [ELEM], outs)[0]

    def add(self, p_xyzt, q_xyzt, outs=None):
        return self._run("d377_batch_add", [p_xyzt, q_xyzt], [ELEM, ELEM], [ELEM], outs)[0]

    def sub(self, p_xyzt, q_xyzt, outs=None):
        """Element - Element = self + other.neg() (src/min_curve/ops.rs:43-87)."""
        return self._run("d377_batch_sub", [p_xyzt, q_xyzt], [ELEM, ELEM], [ELEM], outs)[0]

    def double(self, p_xyzt, outs=None):
        return self._run("d377_batch_double", [p_xyzt], [ELEM], [ELEM], outs)[0]

    def eq(self, p_xyzt, q_xyzt, outs=None):
        return self._run("d377_batch_eq", [p_xyzt, q_xyzt], [ELEM, ELEM], [FLAG], outs)[0]

    def neg(self, p_xyzt, outs=None):
        return self._run("d377_batch_neg", [p_xyzt], [ELEM], [ELEM], outs)[0]

    def is_identity(self, p_xyzt, outs=None):
        return self._run("d377_batch_is_identity", [p_xyzt], [ELEM], [FLAG], outs)[0]

    def sharded(self, op, in0, in1=None, outs=None):
        """d377_batch_sharded_dev: an HBM-resident batch (torch CUDA tensors on one device of this context)
        split into contiguous slices over all the context's GPUs by peer copies over xGMI.  op is one of
        SHARD_OPS; inputs and outputs are those of the method of the same name."""
        specs = {"sqrt_ratio_zeta": ([ENC, ENC], [ENC, FLAG]), "decompress": ([ENC], [ELEM, FLAG]),
                 "compress": ([ELEM], [ENC]), "roundtrip": ([ENC], [ENC, FLAG]), "scalar_mul_base": ([ENC], [ENC]),
                 "scalar_mul_var": ([ENC, ENC], [ENC, FLAG]), "encode_to_curve": ([ENC], [ENC]),
                 "hash_to_curve": ([ENC, ENC], [ENC]), "scalar_mul_var_element": ([ELEM, ENC], [ELEM]),
                 "scalar_mul_base_element": ([ENC], [ELEM])}[op]
        ins = [in0] if in1 is None else [in0, in1]
        if len(ins) != len(specs[0]):
            raise ValueError("%s takes %d input arrays" % (op, len(specs[0])))
        return self._run("d377_batch_sharded_dev:" + op, ins, specs[0], specs[1], outs, sharded_op=SHARD_OPS[op])

    def identity(self):
        """Element::IDENTITY as one [16] u64 record (src/min_curve/element.rs:53-58)."""
        out = np.zeros(16, np.uint64)
        self._lib.d377_identity(out.ctypes.data_as(ctypes.c_void_p))
        return out

    def generator(self):
        """Element::GENERATOR as one [16] u64 record (src/min_curve/element.rs:61-81)."""
        out = np.zeros(16, np.uint64)
        self._lib.d377_generator(out.ctypes.data_as(ctypes.c_void_p))
        return out


    # -- Fr byte handling ---------------------------------------------------------------------------
    def fr_from_le_bytes_mod_order(self, bytes32, outs=None):
        """Fr::from_le_bytes_mod_order on [n, 32] byte strings -> canonical [n, 32] (src/fields/fr.rs:82-94)."""
        return self._run("d377_batch_fr_from_le_bytes_mod_order", [bytes32], [ENC], [ENC], outs)[0]

    def fr_from_bytes_checked(self, bytes32, outs=None):
        """Fr::from_bytes_checked (src/fields/fr.rs:100-107) -> ([n, 32], status[n]); a rejected record is all-zero."""
        return self._run("d377_batch_fr_from_bytes_checked", [bytes32], [ENC], [ENC, FLAG], outs)

    def fr_op(self, op, a32, b32=None, outs=None):
        """Fr add/sub/mul (binary) and square/neg/inverse (unary) on [n, 32] little-endian scalars
        (src/fields/fr/u64/wrapper.rs:76-108): inputs reduced mod r, outputs canonical -> (out [n, 32], status [n]);
        status 1 (and a zero record) only for inverse(0).  numpy arrays or torch CUDA tensors."""
        code = self.FQ_OPS[op]
        binary = code <= 2
        if binary != (b32 is not None):
            raise ValueError("fr_op %s takes %s" % (op, "two operands" if binary else "one operand"))
        ins = [a32, b32] if binary else [a32]
        return self._run("d377_batch_fr_op", ins, [ENC] * len(ins), [ENC, FLAG], outs, pre=(ctypes.c_int(code),), in_slots=2)

    def fr_from_wide_bytes(self, data):
        """Fr::from_le_bytes_mod_order on [n, 48|64] byte strings -> canonical [n, 32] (src/fields/fr.rs:82-94)."""
        return self._wide("d377_batch_fr_from_wide_bytes", data)

    # -- Fq on in-memory elements (4 Montgomery u64 limbs) ------------------------------------------
    FQ_OPS = {"add": 0, "sub": 1, "mul": 2, "square": 3, "neg": 4, "inverse": 5}

    def fq_op(self, op, a, b=None):
        """Fq add/sub/mul (binary) and square/neg/inverse (unary) on [n, 4] u64 Montgomery records
        (src/fields/fq/u64/wrapper.rs:99-132) -> (out [n, 4], status [n]); status 1 only for inverse(0).
        numpy arrays (host path) or torch CUDA tensors (device path)."""
        code = self.FQ_OPS[op]
        binary = code <= 2
        if binary and b is None:
            raise ValueError("fq_op %s takes two operands" % op)
        if _is_torch(a):
            import torch
            n = _rows(a)
            dev = a.device
            di = self._dev_index(dev)
            _check(a, FQM, n, "fq_op input", dev)
            if binary:
                _check(b, FQM, n, "fq_op input", dev)
            a = a.contiguous()
            bb = b.contiguous() if binary else None
            out = torch.empty((n, 4), dtype=torch.int64, device=dev)
            st = torch.zeros((max(n, 1),), dtype=torch.uint8, device=dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _native.check(self._lib.d377_batch_fq_op_dev(self._h, di, stream, code, ctypes.c_void_p(a.data_ptr()),
                                                         ctypes.c_void_p(bb.data_ptr()) if binary else None,
                                                         ctypes.c_size_t(n), ctypes.c_void_p(out.data_ptr()),
                                                         ctypes.c_void_p(st.data_ptr())))
            return out, st[:n]
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        n = a.shape[0]
        out = np.empty((n, 4), np.uint64)
        st = np.zeros(max(n, 1), np.uint8)
        p = lambda x: x.ctypes.data_as(ctypes.c_void_p)
        bb = None
        if binary:
            bb = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
            if bb.shape[0] != n:
                raise ValueError("fq_op: operands have %d and %d rows" % (n, bb.shape[0]))
        _native.check(self._lib.d377_batch_fq_op(self._h, code, p(a), p(bb) if bb is not None else None,
                                                 ctypes.c_size_t(n), p(out), p(st)))
        return out, st[:n]

    def fq_from_bytes_checked(self, bytes32):
        """Fq::from_bytes_checked (src/fields/fq.rs:108-115) -> ([n, 4] u64, status[n])."""
        return self._run("d377_batch_fq_from_bytes_checked", [bytes32], [ENC], [FQM, FLAG])

    def fq_to_bytes(self, a):
        """Fq::to_bytes_le on [n, 4] u64 Montgomery records -> [n, 32] u8."""
        return self._run("d377_batch_fq_to_bytes", [a], [FQM], [ENC])[0]

    # -- wide byte strings and affine normalisation -----------------------------------------------
    def _wide(self, name, data):
        if data.ndim != 2 or int(data.shape[1]) not in (48, 64):
            raise ValueError("%s: expected [n, 48] or [n, 64] byte strings, got shape %s" % (name, tuple(data.shape)))
        n, length = int(data.shape[0]), int(data.shape[1])
        _check(data, ((length,), "u8"), n, name + " input")
        if _is_torch(data):
            import torch
            dev = data.device
            self._dev_index(dev)
            out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _native.check(getattr(self._lib, name + "_dev")(self._h, self.device_ids.index(dev.index), stream,
                                                            ctypes.c_void_p(data.contiguous().data_ptr()),
                                                            ctypes.c_size_t(length), ctypes.c_size_t(n),
                                                            ctypes.c_void_p(out.data_ptr())))
            return out
        data = np.ascontiguousarray(data, dtype=np.uint8)
        out = np.zeros((n, 32), np.uint8)
        _native.check(getattr(self._lib, name)(self._h, data.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(length),
                                               ctypes.c_size_t(n), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def fq_from_wide_bytes(self, data):
        """Fq::from_le_bytes_mod_order on [n, 48|64] byte strings -> canonical [n, 32]."""
        return self._wide("d377_batch_fq_from_wide_bytes", data)

    def encode_to_curve_wide(self, data):
        """encode_to_curve(Fq::from_le_bytes_mod_order(bytes)) for [n, 48|64] byte strings."""
        return self._wide("d377_batch_encode_to_curve_wide", data)

    def to_affine(self, xyzt, outs=None):
        """CurveGroup::normalize_batch: [n, 16] Elements -> [n, 8] (x, y Montgomery limbs)."""
        return self._run("d377_batch_to_affine", [xyzt], [ELEM], [AFF], outs)[0]

    # -- multi-scalar multiplication -------------------------------------------------------------
    def msm_small(self, points, scalar32, m, outs=None, elements=False):
        """n independent multiscalar sums of m terms each (d377_batch_msm_small[_encoded]): Element::vartime_multiscalar_mul
        (src/ark_curve/element/projective.rs:99-117) in the shape of the reference's own test, a 3-term sum per case
        (tests/operations.rs:44-60), many at once.  points: [n * m, 16] u64 Elements or [n * m, 32] u8 Encodings, scalar32:
        [n * m, 32], term-major within a sum; 1 <= m <= 8.  Returns enc [n, 32] for Elements, (enc, status [n * m]) for
        Encodings (an invalid Encoding is reported and left out of its sum); with elements=True the sums also come back as
        Element records, what the reference's function returns: (enc, xyzt [n, 16]) / (enc, xyzt, status), in d377_msm's
        order.  outs: the same tuple of preallocated arrays."""
        m = int(m)
        if points.ndim != 2 or int(points.shape[1]) not in (16, 32):
            raise ValueError("msm_small: points must be [n * m, 16] Elements or [n * m, 32] Encodings")
        encoded = int(points.shape[1]) == 32
        terms = _rows(points)
        if m < 1 or terms % m:
            raise ValueError("msm_small: the number of points must be a multiple of m")
        n = terms // m
        _check(points, ENC if encoded else ELEM, terms, "msm_small points")
        _check(scalar32, ENC, terms, "msm_small scalars", points.device if _is_torch(points) else None)
        name = "d377_batch_msm_small_encoded" if encoded else "d377_batch_msm_small"
        specs = [(ENC, n)] + ([(ELEM, n)] if elements else []) + ([(FLAG, terms)] if encoded else [])
        null = ctypes.c_void_p(None)
        if _is_torch(points):
            import torch
            dev = points.device
            di = self._dev_index(dev)
            pts, sc = points.contiguous(), scalar32.contiguous()
            if outs is None:
                outs = [torch.empty((rows,) + tail, dtype=torch.uint8 if k == "u8" else torch.int64, device=dev) for (tail, k), rows in specs]
            if len(outs) != len(specs):
                raise ValueError("msm_small: %d output arrays expected" % len(specs))
            for a, (spec, rows) in zip(outs, specs):
                _check(a, spec, rows, "msm_small output", dev)
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            p = lambda t: ctypes.c_void_p(t.data_ptr())
            ptrs = [p(o) for o in outs]
            if not elements:
                ptrs.insert(1, null)
            _native.check(getattr(self._lib, name + "_dev")(self._h, di, stream, p(pts), p(sc), ctypes.c_size_t(m), ctypes.c_size_t(n), *ptrs))
            return tuple(outs) if len(outs) > 1 else outs[0]
        pts, sc = np.ascontiguousarray(points), np.ascontiguousarray(scalar32)
        if outs is None:
            outs = [np.zeros((rows,) + tail, np.uint8 if k == "u8" else np.uint64) for (tail, k), rows in specs]
        if len(outs) != len(specs):
            raise ValueError("msm_small: %d output arrays expected" % len(specs))
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        ptrs = [p(o) for o in outs]
        if not elements:
            ptrs.insert(1, null)
        _native.check(getattr(self._lib, name)(self._h, p(pts), p(sc), ctypes.c_size_t(m), ctypes.c_size_t(n), *ptrs))
        return tuple(outs) if len(outs) > 1 else outs[0]

    def msm(self, points, scalar32, encoded=None):
        """Element::vartime_multiscalar_mul (src/ark_curve/element/projective.rs:99-117).
        points: [n, 16] u64 Elements or [n, 32] u8 Encodings (detected by the row width).
        Returns (enc[32] u8, xyzt[16] u64, status[n] or None)."""
        n = _rows(points)
        if points.ndim != 2 or int(points.shape[1]) not in (16, 32):
            raise ValueError("msm: points must be [n, 16] Elements or [n, 32] Encodings")
        if encoded is None:
            encoded = int(points.shape[1]) == 32
        # the reference zips scalars with points (projective.rs:99-117); a length mismatch is a caller bug here
        _check(points, ENC if encoded else ELEM, n, "msm points")
        _check(scalar32, ENC, n, "msm scalars", points.device if _is_torch(points) else None)
        if _is_torch(points):
            import torch
            dev = points.device
            self._dev_index(dev)
            pts, sc = points.contiguous(), scalar32.contiguous()
            enc = torch.empty((32,), dtype=torch.uint8, device=dev)
            xyzt = torch.empty((16,), dtype=torch.int64, device=dev)
            st = torch.empty((max(n, 1),), dtype=torch.uint8, device=dev) if encoded else None
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            di = self.device_ids.index(dev.index)
            p = lambda t: ctypes.c_void_p(t.data_ptr())
            if encoded:
                _native.check(self._lib.d377_msm_encoded_dev(self._h, di, stream, p(pts), p(sc), ctypes.c_size_t(n),
                                                             p(enc), p(xyzt), p(st)))
                return enc, xyzt, st[:n]
            _native.check(self._lib.d377_msm_dev(self._h, di, stream, p(pts), p(sc), ctypes.c_size_t(n), p(enc), p(xyzt)))
            return enc, xyzt, None
        pts = np.ascontiguousarray(points)
        sc = np.ascontiguousarray(scalar32)
        enc = np.zeros(32, np.uint8)
        xyzt = np.zeros(16, np.uint64)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        if encoded:
            st = np.zeros(max(n, 1), np.uint8)
            _native.check(self._lib.d377_msm_encoded(self._h, p(pts), p(sc), ctypes.c_size_t(n), p(enc), p(xyzt), p(st)))
            return enc, xyzt, st[:n]
        _native.check(self._lib.d377_msm(self._h, p(pts), p(sc), ctypes.c_size_t(n), p(enc), p(xyzt)))
        return enc, xyzt, None

    def sum_elements(self, xyzt):
        """Sum of m Element records held in HBM -> (enc[32], xyzt[16]); used to combine the
        per-rank partial sums of a sharded MSM."""
        import torch
        m = _rows(xyzt)
        _check(xyzt, ELEM, m, "sum_elements input")
        dev = xyzt.device
        self._dev_index(dev)
        enc = torch.empty((32,), dtype=torch.uint8, device=dev)
        out = torch.empty((16,), dtype=torch.int64, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _native.check(self._lib.d377_sum_elements_dev(self._h, self.device_ids.index(dev.index), stream,
                                                      ctypes.c_void_p(xyzt.contiguous().data_ptr()), ctypes.c_size_t(m),
                                                      ctypes.c_void_p(enc.data_ptr()), ctypes.c_void_p(out.data_ptr())))
        return enc, out


_default = None


def default_context():
    global _default
    if _default is None:
        _default = Context()
    return _default


class _Bytes32:
    """A batch of 32-byte little-endian records, shape [n, 32] uint8."""

    def __init__(self, data, ctx=None):
        if not _is_torch(data):
            data = np.ascontiguousarray(np.asarray(data, dtype=np.uint8)).reshape(-1, 32)
        self.data = data
        self.ctx = ctx

    def _ctx(self):
        return self.ctx or default_context()

    def __len__(self):
        return int(self.data.shape[0])

    def to_bytes(self):
        d = self.data.cpu().numpy() if _is_torch(self.data) else self.data
        return [bytes(r) for r in d]


class Fq(_Bytes32):
    """Batch of base-field elements given as 32 bytes, interpreted mod q like
    Fq::from_le_bytes_mod_order (src/fields/fq.rs:90-102)."""

    @staticmethod
    def sqrt_ratio_zeta(num, den, root="ark"):
        """Fq::sqrt_ratio_zeta (src/ark_curve/invsqrt.rs:75-166; root="min_curve": the min_curve backend's
        non_arkworks_sqrt_ratio_zeta, src/min_curve/invsqrt.rs:73-95) -> (was_square[n], Fq roots)."""
        r, ws = num._ctx().sqrt_ratio_zeta(num.data, den.data, root=root)
        return ws, Fq(r, num.ctx)


class Fr(_Bytes32):
    """Batch of scalars given as 32 bytes, interpreted mod r like Fr::from_le_bytes_mod_order
    (src/fields/fr.rs:82-94)."""

    def _op(self, op, other=None):
        out, st = self._ctx().fr_op(op, self.data, None if other is None else other.data)
        return Fr(out, self.ctx), st

    def __add__(self, other):
        return self._op("add", other)[0]

    def __sub__(self, other):
        return self._op("sub", other)[0]

    def __mul__(self, other):
        """Fr * Fr, or Fr * Element = Element * Fr (src/min_curve/ops.rs:89-95)."""
        if isinstance(other, Element):
            return other * self
        return self._op("mul", other)[0]

    def __neg__(self):
        return self._op("neg")[0]

    def square(self):
        return self._op("square")[0]

    def inverse(self):
        """Fr::inverse (src/fields/fr/u64/wrapper.rs:80-86) -> (Fr, status[n]); status 1 = None (zero input)."""
        return self._op("inverse")

    @staticmethod
    def from_le_bytes_mod_order(data, ctx=None):
        """Fr::from_le_bytes_mod_order for [n, 32], [n, 48] or [n, 64] byte strings (src/fields/fr.rs:82-94)."""
        c = ctx or default_context()
        width = int(data.shape[1])
        return Fr(c.fr_from_le_bytes_mod_order(data) if width == 32 else c.fr_from_wide_bytes(data), ctx)


class Encoding(_Bytes32):
    """Batch of `Encoding([u8; 32])` (src/ark_curve/encoding.rs:14-15)."""

    def vartime_decompress(self):
        """-> (Element, status[n]); status 1 = EncodingError::InvalidEncoding (encoding.rs:32-83)."""
        xyzt, st = self._ctx().decompress(self.data)
        return Element(xyzt, self.ctx), st

    def roundtrip(self):
        out, st = self._ctx().roundtrip(self.data)
        return Encoding(out, self.ctx), st

    def scalar_mul(self, scalars):
        """decompress(self) * scalars, compressed (src/min_curve/ops.rs:89-95)."""
        out, st = self._ctx().scalar_mul_var(self.data, scalars.data)
        return Encoding(out, self.ctx), st

    def unwrap_decompress(self):
        el, st = self.vartime_decompress()
        bad = st.cpu().numpy() if _is_torch(st) else st
        if bad.any():
            raise EncodingError("InvalidEncoding at index %d" % int(np.nonzero(bad)[0][0]))
        return el


class Element:
    """Batch of group elements in the reference's in-memory form: X, Y, Z, T as 4 Montgomery
    u64 limbs each (shape [n, 16]); src/min_curve/element.rs:31-38."""

    def __init__(self, xyzt, ctx=None):
        if not _is_torch(xyzt):
            xyzt = np.ascontiguousarray(np.asarray(xyzt, dtype=np.uint64)).reshape(-1, 16)
        self.data = xyzt
        self.ctx = ctx

    def _ctx(self):
        return self.ctx or default_context()

    def __len__(self):
        return int(self.data.shape[0])

    def __add__(self, other):
        """Element + Element (src/min_curve/element.rs:291-322)."""
        return Element(self._ctx().add(self.data, other.data), self.ctx)

    def __neg__(self):
        """-Element (src/min_curve/element.rs:324-332)."""
        return Element(self._ctx().neg(self.data), self.ctx)

    def is_identity(self):
        """Element::is_identity (src/min_curve/element.rs:113-117) -> u8[n]."""
        return self._ctx().is_identity(self.data)

    def double(self):
        """Element::double (src/min_curve/element.rs:119-136)."""
        return Element(self._ctx().double(self.data), self.ctx)

    def eq(self, other):
        """PartialEq for Element: x1*y2 == x2*y1 (src/min_curve/element.rs:334-340) -> u8[n]."""
        return self._ctx().eq(self.data, other.data)

    def vartime_compress(self):
        """Element::vartime_compress (src/ark_curve/encoding.rs:116-128)."""
        return Encoding(self._ctx().compress(self.data), self.ctx)

    def vartime_compress_to_field(self):
        """Element::vartime_compress_to_field (src/min_curve/element.rs:163-181) -> [n, 4] u64 Montgomery limbs."""
        return self._ctx().compress_to_field(self.data)

    def __mul__(self, scalars):
        """Element * Fr -> Element (src/min_curve/ops.rs:89-95)."""
        return Element(self._ctx().scalar_mul_var_element(self.data, scalars.data), self.ctx)

    @staticmethod
    def encode_to_curve_element(r):
        """Element::encode_to_curve (src/min_curve/element.rs:242-244) as an Element."""
        return Element(r._ctx().encode_to_curve_element(r.data), r.ctx)

    @staticmethod
    def hash_to_curve_element(r1, r2):
        """Element::hash_to_curve (src/min_curve/element.rs:235-240) as an Element."""
        return Element(r1._ctx().hash_to_curve_element(r1.data, r2.data), r1.ctx)

    @staticmethod
    def generator_mul_element(scalars):
        """Element::GENERATOR * scalars as Elements."""
        return Element(scalars._ctx().scalar_mul_base_element(scalars.data), scalars.ctx)

    @staticmethod
    def encode_to_curve(r):
        """Element::encode_to_curve (src/ark_curve/elligator.rs:74-76), returned compressed."""
        return Encoding(r._ctx().encode_to_curve(r.data), r.ctx)

    @staticmethod
    def hash_to_curve(r1, r2):
        """Element::hash_to_curve (src/ark_curve/elligator.rs:67-71), returned compressed."""
        return Encoding(r1._ctx().hash_to_curve(r1.data, r2.data), r1.ctx)

    @staticmethod
    def vartime_multiscalar_mul(scalars, points):
        """Element::vartime_multiscalar_mul(scalars, points) -> Encoding of the sum
        (src/ark_curve/element/projective.rs:99-117); `points` is an Element or Encoding batch."""
        enc, _, _ = points._ctx().msm(points.data, scalars.data)
        return Encoding(enc.reshape(1, 32) if not _is_torch(enc) else enc.reshape(1, 32), points.ctx)

    @staticmethod
    def generator_mul(scalars):
        """Element::GENERATOR * scalars, returned compressed."""
        return Encoding(scalars._ctx().scalar_mul_base(scalars.data), scalars.ctx)
