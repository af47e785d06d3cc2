"""ctypes front-end for the CPU oracle (oracle/libd377_oracle.so). Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libd377_oracle.so")


def build_oracle():
    """Builds the checker if it is missing or older than its source.  Several ranks of a multi-GPU bench may get here at
    once: the check and the build happen under a file lock, so one of them builds and the others find it done."""
    import fcntl
    src = os.path.join(ORACLE_DIR, "d377_oracle.c")
    with open(os.path.join(ORACLE_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
                subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


NATIVE_LIB = os.path.join(ORACLE_DIR, "libd377_oracle_native.so")


def build_native():
    """The -march=native copy for bench.py's cpu_baseline, built on THIS host.  -> path, or None without a compiler."""
    import fcntl
    try:
        with open(os.path.join(ORACLE_DIR, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-C", ORACLE_DIR, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return NATIVE_LIB
    except Exception:
        return None


def prebuild(native=False):
    """Everything Oracle(...) would have to build, built now: callers that open the GPU later (bench.py) call this first and
    then construct Oracle(build=False), which only loads what exists and never starts a program."""
    build_oracle()
    if native:
        build_native()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def as_u8(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.uint8)).reshape(-1, 32)
    if n is not None:
        assert a.shape[0] == n
    return a


class Oracle:
    OPS = {"roundtrip": 0, "scalar_mul_base": 1, "scalar_mul_var": 2, "encode_to_curve": 3, "sqrt_ratio_zeta": 4,
           "sqrt_ratio_zeta_min_curve": 5}

    def __init__(self, native=False, build=True):
        """native=True: a copy built on THIS host with -march=native (bench.py's cpu_baseline on the GPU box);
        falls back to the portable build when the compiler is not there.  build=False: load what prebuild() left, start
        no program (for processes that have opened the GPU)."""
        path = build_oracle() if build else LIB
        self.flags = "-O3 -march=x86-64-v3"
        if native:
            npath = build_native() if build else (NATIVE_LIB if os.path.exists(NATIVE_LIB) else None)
            if npath:
                path = npath
                self.flags = "-O3 -march=native (built on this host)"
        self.lib = ctypes.CDLL(path)
        self.lib.d377o_init()
        self.lib.d377o_run_threads.restype = ctypes.c_int

    def _n(self, n):
        return ctypes.c_size_t(n)

    def sqrt_ratio_zeta(self, num, den):
        num, den = as_u8(num), as_u8(den)
        n = num.shape[0]
        root = np.zeros((n, 32), np.uint8)
        ws = np.zeros(n, np.uint8)
        self.lib.d377o_sqrt_ratio_zeta(_p(num), _p(den), self._n(n), _p(root), _p(ws))
        return root, ws

    def sqrt_ratio_zeta_min_curve(self, num, den):
        """The min_curve backend's root (constant-time Tonelli-Shanks, src/min_curve/invsqrt.rs:73-95)."""
        num, den = as_u8(num), as_u8(den)
        n = num.shape[0]
        root = np.zeros((n, 32), np.uint8)
        ws = np.zeros(n, np.uint8)
        self.lib.d377o_sqrt_ratio_zeta_min_curve(_p(num), _p(den), self._n(n), _p(root), _p(ws))
        return root, ws

    def decompress(self, enc):
        enc = as_u8(enc)
        n = enc.shape[0]
        xyzt = np.zeros((n, 16), np.uint64)
        st = np.zeros(n, np.uint8)
        self.lib.d377o_decompress(_p(enc), self._n(n), _p(xyzt), _p(st))
        return xyzt, st

    def compress(self, xyzt):
        xyzt = np.ascontiguousarray(xyzt, dtype=np.uint64).reshape(-1, 16)
        n = xyzt.shape[0]
        out = np.zeros((n, 32), np.uint8)
        self.lib.d377o_compress(_p(xyzt), self._n(n), _p(out))
        return out

    def roundtrip(self, enc):
        enc = as_u8(enc)
        n = enc.shape[0]
        out = np.zeros((n, 32), np.uint8)
        st = np.zeros(n, np.uint8)
        self.lib.d377o_roundtrip(_p(enc), self._n(n), _p(out), _p(st))
        return out, st

    def scalar_mul_base(self, k):
        k = as_u8(k)
        n = k.shape[0]
        out = np.zeros((n, 32), np.uint8)
        self.lib.d377o_scalar_mul_base(_p(k), self._n(n), _p(out))
        return out

    def scalar_mul_var(self, enc, k):
        enc, k = as_u8(enc), as_u8(k)
        n = enc.shape[0]
        out = np.zeros((n, 32), np.uint8)
        st = np.zeros(n, np.uint8)
        self.lib.d377o_scalar_mul_var(_p(enc), _p(k), self._n(n), _p(out), _p(st))
        return out, st

    def encode_to_curve(self, r0):
        r0 = as_u8(r0)
        n = r0.shape[0]
        out = np.zeros((n, 32), np.uint8)
        self.lib.d377o_encode_to_curve(_p(r0), self._n(n), _p(out))
        return out

    def elligator_map_xyzt(self, r0):
        r0 = as_u8(r0)
        n = r0.shape[0]
        out = np.zeros((n, 16), np.uint64)
        self.lib.d377o_elligator_map_xyzt(_p(r0), self._n(n), _p(out))
        return out

    def hash_to_curve(self, r1, r2):
        r1, r2 = as_u8(r1), as_u8(r2)
        n = r1.shape[0]
        out = np.zeros((n, 32), np.uint8)
        self.lib.d377o_hash_to_curve(_p(r1), _p(r2), self._n(n), _p(out))
        return out

    def add_xyzt(self, p, q):
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 16)
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, 16)
        out = np.zeros_like(p)
        self.lib.d377o_add_xyzt(_p(p), _p(q), self._n(p.shape[0]), _p(out))
        return out

    def sub_xyzt(self, p, q):
        """Element - Element = self + other.neg() (src/min_curve/ops.rs:43-49)."""
        return self.add_xyzt(p, self.neg_xyzt(q))

    def double_xyzt(self, p):
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 16)
        out = np.zeros_like(p)
        self.lib.d377o_double_xyzt(_p(p), self._n(p.shape[0]), _p(out))
        return out

    def scalar_mul_xyzt(self, p, k):
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 16)
        k = as_u8(k)
        out = np.zeros_like(p)
        self.lib.d377o_scalar_mul_xyzt(_p(p), _p(k), self._n(p.shape[0]), _p(out))
        return out

    def eq_xyzt(self, p, q):
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 16)
        q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, 16)
        eq = np.zeros(p.shape[0], np.uint8)
        self.lib.d377o_eq_xyzt(_p(p), _p(q), self._n(p.shape[0]), _p(eq))
        return eq

    def neg_xyzt(self, p):
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 16)
        out = np.zeros_like(p)
        self.lib.d377o_neg_xyzt(_p(p), self._n(p.shape[0]), _p(out))
        return out

    def is_identity(self, p):
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 16)
        out = np.zeros(p.shape[0], np.uint8)
        self.lib.d377o_is_identity(_p(p), self._n(p.shape[0]), _p(out))
        return out

    def identity_xyzt(self):
        out = np.zeros(16, np.uint64)
        self.lib.d377o_identity_xyzt(_p(out))
        return out

    def fq_op(self, op, a, b=None):
        """op: 0 add, 1 sub, 2 mul, 3 square, 4 neg, 5 inverse -> (Montgomery limbs, status)."""
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        bb = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4) if b is not None else None
        out = np.zeros_like(a)
        st = np.zeros(a.shape[0], np.uint8)
        self.lib.d377o_fq_op(ctypes.c_int(op), _p(a), _p(bb) if bb is not None else None, self._n(a.shape[0]), _p(out), _p(st))
        return out, st

    def fr_op(self, op, a, b=None):
        """op codes as fq_op, on [n, 32] little-endian scalars (any bytes: reduced mod r) -> (canonical bytes, status)."""
        a = as_u8(a)
        bb = as_u8(b) if b is not None else None
        out = np.zeros_like(a)
        st = np.zeros(a.shape[0], np.uint8)
        self.lib.d377o_fr_op(ctypes.c_int(op), _p(a), _p(bb) if bb is not None else None, self._n(a.shape[0]), _p(out), _p(st))
        return out, st

    def fr_from_wide_bytes(self, data):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n, length = data.shape
        out = np.zeros((n, 32), np.uint8)
        self.lib.d377o_fr_from_wide_bytes(_p(data), ctypes.c_size_t(length), self._n(n), _p(out))
        return out

    def compress_to_field(self, p):
        p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 16)
        out = np.zeros((p.shape[0], 4), np.uint64)
        self.lib.d377o_compress_to_field(_p(p), self._n(p.shape[0]), _p(out))
        return out

    def hash_to_curve_xyzt(self, r1, r2):
        r1, r2 = as_u8(r1), as_u8(r2)
        out = np.zeros((r1.shape[0], 16), np.uint64)
        self.lib.d377o_hash_to_curve_xyzt(_p(r1), _p(r2), self._n(r1.shape[0]), _p(out))
        return out

    def msm(self, xyzt, k, threads=8):
        """Reference fold; slices are folded on `threads` python-side chunks and summed (the sum is
        commutative and the encoding canonical, so chunking does not change the result)."""
        xyzt = np.ascontiguousarray(xyzt, dtype=np.uint64).reshape(-1, 16)
        k = as_u8(k)
        n = xyzt.shape[0]
        enc = np.zeros(32, np.uint8)
        out = np.zeros(16, np.uint64)
        if n <= 256 or threads <= 1:
            self.lib.d377o_msm(_p(xyzt), _p(k), self._n(n), _p(enc), _p(out))
            return enc, out
        from concurrent.futures import ThreadPoolExecutor
        bounds = np.linspace(0, n, threads + 1).astype(int)

        def part(j):
            lo, hi = int(bounds[j]), int(bounds[j + 1])
            e = np.zeros(32, np.uint8)
            o = np.zeros(16, np.uint64)
            x, kk = np.ascontiguousarray(xyzt[lo:hi]), np.ascontiguousarray(k[lo:hi])
            self.lib.d377o_msm(_p(x), _p(kk), self._n(hi - lo), _p(e), _p(o))
            return o
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(part, range(threads)))      # ctypes releases the GIL
        acc = parts[0].reshape(1, 16)
        for q in parts[1:]:
            acc = self.add_xyzt(acc, q.reshape(1, 16))
        return self.compress(acc)[0], acc[0]

    def fq_from_wide_bytes(self, b, length):
        b = np.ascontiguousarray(np.asarray(b, dtype=np.uint8)).reshape(-1, length)
        out = np.zeros((b.shape[0], 32), np.uint8)
        self.lib.d377o_fq_from_wide_bytes(_p(b), self._n(length), self._n(b.shape[0]), _p(out))
        return out

    def encode_to_curve_wide(self, b, length):
        b = np.ascontiguousarray(np.asarray(b, dtype=np.uint8)).reshape(-1, length)
        out = np.zeros((b.shape[0], 32), np.uint8)
        self.lib.d377o_encode_to_curve_wide(_p(b), self._n(length), self._n(b.shape[0]), _p(out))
        return out

    def to_affine(self, xyzt):
        xyzt = np.ascontiguousarray(xyzt, dtype=np.uint64).reshape(-1, 16)
        out = np.zeros((xyzt.shape[0], 8), np.uint64)
        self.lib.d377o_to_affine(_p(xyzt), self._n(xyzt.shape[0]), _p(out))
        return out

    def generator_xyzt(self):
        out = np.zeros(16, np.uint64)
        self.lib.d377o_generator_xyzt(_p(out))
        return out

    def fq_from_bytes_mod_order(self, b):
        b = as_u8(b)
        out = np.zeros((b.shape[0], 4), np.uint64)
        self.lib.d377o_fq_from_bytes_mod_order(_p(b), self._n(b.shape[0]), _p(out))
        return out

    def fq_from_bytes_checked(self, b):
        b = as_u8(b)
        out = np.zeros((b.shape[0], 4), np.uint64)
        st = np.zeros(b.shape[0], np.uint8)
        self.lib.d377o_fq_from_bytes_checked(_p(b), self._n(b.shape[0]), _p(out), _p(st))
        return out, st

    def fq_to_bytes(self, mont):
        mont = np.ascontiguousarray(mont, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((mont.shape[0], 32), np.uint8)
        self.lib.d377o_fq_to_bytes(_p(mont), self._n(mont.shape[0]), _p(out))
        return out

    def fq_mul_mont(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros_like(a)
        self.lib.d377o_fq_mul_mont(_p(a), _p(b), self._n(a.shape[0]), _p(out))
        return out

    def fr_from_bytes_mod_order(self, b):
        b = as_u8(b)
        out = np.zeros((b.shape[0], 32), np.uint8)
        self.lib.d377o_fr_from_bytes_mod_order(_p(b), self._n(b.shape[0]), _p(out))
        return out

    def fr_from_bytes_checked(self, b):
        b = as_u8(b)
        st = np.zeros(b.shape[0], np.uint8)
        self.lib.d377o_fr_from_bytes_checked(_p(b), self._n(b.shape[0]), _p(st))
        return st

    def run_threads(self, op, a, b, threads):
        """Timed-baseline driver: contiguous slices over `threads` pthreads."""
        a = as_u8(a)
        n = a.shape[0]
        bb = as_u8(b) if b is not None else None
        out = np.zeros((n, 32), np.uint8)
        st = np.zeros(n, np.uint8)
        used = self.lib.d377o_run_threads(self.OPS[op], _p(a), _p(bb) if bb is not None else None,
                                          self._n(n), _p(out), _p(st), threads)
        assert used > 0
        return out, st, used
