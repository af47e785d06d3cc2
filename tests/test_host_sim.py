"""CPU check of the per-lane device functions (decaf377_amd/csrc/fq29.hpp, curve.hpp).

The headers are compiled for the host with g++ into a test-only library under
tests/host_sim/ (never loaded by the package) and compared with the oracle.  This
covers the 29-bit-limb arithmetic, the lazy-reduction bounds, the Sarkar table
construction and the curve formulas before any GPU run; the `-m gpu` tests then check
the same functions as compiled by hipcc for gfx950."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

R_ORDER = 2111115437357092606062206234695386632838870926408408195193685246394721360383

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM_DIR = os.path.join(ROOT, "tests", "host_sim")
CSRC = os.path.join(ROOT, "decaf377_amd", "csrc")
Q = 725501752471715841 | 6461107452199829505 << 64 | 6968279316240510977 << 128 | 1345280370688173398 << 192


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def sim():
    lib = os.path.join(SIM_DIR, "libd377_sim.so")
    srcs = [os.path.join(SIM_DIR, "sim.cpp")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        # -DD377_FB_BITS=12: the product's 23-bit fixed-base comb has 46 M entries, each an inversion -- far too slow to build on a CPU
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DD377_FB_BITS=12", "-I" + CSRC,
                               os.path.join(SIM_DIR, "sim.cpp"), "-o", lib])
    L = ctypes.CDLL(lib)
    L.sim_init.restype = ctypes.c_int
    assert L.sim_init() == 0, "s_lookup perfect hash has collisions"
    return L


def n_(n):
    return ctypes.c_size_t(n)


def test_constants(sim):
    sub = np.zeros(9, np.uint32)
    sub_nc = np.zeros(9, np.uint32)
    ql = np.zeros(9, np.uint32)
    sim.sim_consts(_p(sub), _p(sub_nc), _p(ql))
    assert sum(int(v) << (29 * i) for i, v in enumerate(ql)) == Q
    assert sum(int(v) << (29 * i) for i, v in enumerate(sub)) == 32 * Q
    assert all((1 << 30) + 64 <= int(v) < (1 << 31) for v in sub[:8])
    assert sum(int(v) << (29 * i) for i, v in enumerate(sub_nc)) == 16 * Q
    assert all((1 << 29) + 8 <= int(v) < (1 << 30) + 8 for v in sub_nc[:8])


def test_generated_multiplier_streams():
    """fe_asm.inc is what tools/gen_fe_asm.py writes (no hand edits), and the instruction counts are the
    ones DESIGN.md quotes: 196 / 168 VALU instructions for 153 / 117 MACs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_fe_asm", os.path.join(ROOT, "tools", "gen_fe_asm.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    text = open(os.path.join(CSRC, "fe_asm.inc")).read()
    want = {("mul", False): (196, 153), ("mul", True): (205, 153), ("sqr", False): (168, 117),
            ("sqr", True): (177, 117), ("sqr2x", False): (169, 117)}
    for (kind, strict), (n, nmac) in want.items():
        body, got_mac, got_n, _ = g.gen(kind, strict)
        assert (got_n, got_mac) == (n, nmac)
        assert '"%s"' % body in text


def test_field_ops_match_oracle(sim, oracle):
    rng = np.random.default_rng(11)
    n = 4096
    raw = rng.integers(0, 256, (2, n, 32), dtype=np.uint8)
    # edge values: 0, 1, q-1, 2^256-1 (reduced), q+1
    edges = [0, 1, Q - 1, (1 << 256) - 1, Q + 1, 2, Q - 2, 1 << 255]
    for j, e in enumerate(edges):
        raw[0, j] = np.frombuffer(int(e).to_bytes(32, "little"), np.uint8)
        raw[1, -1 - j] = np.frombuffer(int(e).to_bytes(32, "little"), np.uint8)
    a = oracle.fq_from_bytes_mod_order(raw[0])
    b = oracle.fq_from_bytes_mod_order(raw[1])
    # bytes -> Montgomery-256 through the 29-bit path
    a_sim = np.zeros((n, 4), np.uint64)
    sim.sim_fq_from_bytes(_p(np.ascontiguousarray(raw[0]).view(np.uint32)), n_(n), _p(a_sim))
    assert (a_sim == a).all()
    out = np.zeros((n, 4), np.uint64)
    sim.sim_fq_mul(_p(a), _p(b), n_(n), _p(out))
    assert (out == oracle.fq_mul_mont(a, b)).all()
    sim.sim_fq_sqr(_p(a), n_(n), _p(out))
    assert (out == oracle.fq_mul_mont(a, a)).all()
    # add / sub against big-int arithmetic on canonical values
    ab = oracle.fq_to_bytes(a)
    bb = oracle.fq_to_bytes(b)
    ai = [int.from_bytes(bytes(x), "little") for x in ab]
    bi = [int.from_bytes(bytes(x), "little") for x in bb]
    w = np.zeros((n, 32), np.uint8)
    sim.sim_fq_add(_p(a), _p(b), n_(n), _p(out))
    sim.sim_fq_to_bytes(_p(out), n_(n), _p(w))
    assert [int.from_bytes(bytes(x), "little") for x in w] == [(x + y) % Q for x, y in zip(ai, bi)]
    sim.sim_fq_sub(_p(a), _p(b), n_(n), _p(out))
    sim.sim_fq_to_bytes(_p(out), n_(n), _p(w))
    assert [int.from_bytes(bytes(x), "little") for x in w] == [(x - y) % Q for x, y in zip(ai, bi)]


def test_sqrt_matches_oracle(sim, oracle, vectors):
    rng = np.random.default_rng(12)
    n = 512
    num = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    den = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    v = vectors["sqrt_ratio_zeta"]
    for i, c in enumerate(v):
        num[i] = np.frombuffer(bytes.fromhex(c["num"]), np.uint8)
        den[i] = np.frombuffer(bytes.fromhex(c["den"]), np.uint8)
    root = np.zeros((n, 32), np.uint8)
    ws = np.zeros(n, np.uint8)
    sim.sim_sqrt_ratio_zeta(_p(num), _p(den), n_(n), _p(root), _p(ws))
    r0, w0 = oracle.sqrt_ratio_zeta(num, den)
    assert (ws == w0).all() and (root == r0).all()


def test_group_ops_match_oracle(sim, oracle, vectors, kats):
    rng = np.random.default_rng(13)
    n = 256
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    enc = np.zeros((n, 32), np.uint8)
    xyzt = np.zeros((n, 16), np.uint64)
    sim.sim_encode_to_curve(_p(r0), n_(n), _p(enc), _p(xyzt))
    assert (enc == oracle.encode_to_curve(r0)).all()
    assert (xyzt == oracle.elligator_map_xyzt(r0)).all()
    # decompress: valid + raw + golden decompress set
    raw = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    raw[:, 31] &= 0x3F
    dv = np.array([list(bytes.fromhex(c["enc"])) for c in vectors["decompress"]], dtype=np.uint8)
    allenc = np.concatenate([enc, raw, dv])
    m = allenc.shape[0]
    x2 = np.zeros((m, 16), np.uint64)
    st = np.zeros(m, np.uint8)
    sim.sim_decompress(_p(allenc), n_(m), _p(x2), _p(st))
    xo, so = oracle.decompress(allenc)
    assert (st == so).all() and (x2 == xo).all()
    out = np.zeros((m, 32), np.uint8)
    sim.sim_roundtrip(_p(allenc), n_(m), _p(out), _p(st))
    oo, so = oracle.roundtrip(allenc)
    assert (st == so).all() and (out == oo).all()
    # compress of non-affine points (z != 1): doubles of the Elligator images
    dbl = oracle.double_xyzt(xyzt)
    c1 = np.zeros((n, 32), np.uint8)
    sim.sim_compress(_p(dbl), n_(n), _p(c1))
    assert (c1 == oracle.compress(dbl)).all()
    # identity and basepoint multiples
    hexes = kats["basepoint_multiples"]["hex"]
    ks = np.zeros((16, 32), np.uint8)
    ks[:, 0] = np.arange(16)
    o = np.zeros((16, 32), np.uint8)
    sim.sim_scalar_mul_base(_p(ks), n_(16), _p(o))
    assert [bytes(x).hex() for x in o] == hexes


def test_scalar_mul_matches_oracle(sim, oracle, vectors):
    v = vectors["scalar_mul_var"]
    pts = np.array([list(bytes.fromhex(c["point"])) for c in v], dtype=np.uint8)
    ks = np.array([list(bytes.fromhex(c["scalar"])) for c in v], dtype=np.uint8)
    n = len(v)
    out = np.zeros((n, 32), np.uint8)
    st = np.zeros(n, np.uint8)
    sim.sim_scalar_mul_var(_p(pts), _p(ks), n_(n), _p(out), _p(st))
    assert [bytes(x).hex() for x in out] == [c["enc"] for c in v]
    assert list(st) == [c["status"] for c in v]
    v = vectors["scalar_mul_base"]
    ks = np.array([list(bytes.fromhex(c["scalar"])) for c in v], dtype=np.uint8)
    out = np.zeros((len(v), 32), np.uint8)
    sim.sim_scalar_mul_base(_p(ks), n_(len(v)), _p(out))
    assert [bytes(x).hex() for x in out] == [c["enc"] for c in v]
    # random set against the oracle
    rng = np.random.default_rng(14)
    n = 96
    enc = oracle.encode_to_curve(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    out = np.zeros((n, 32), np.uint8)
    st = np.zeros(n, np.uint8)
    sim.sim_scalar_mul_var(_p(enc), _p(k), n_(n), _p(out), _p(st))
    oo, so = oracle.scalar_mul_var(enc, k)
    assert (out == oo).all() and (st == so).all()
    sim.sim_scalar_mul_base(_p(k), n_(n), _p(out))
    assert (out == oracle.scalar_mul_base(k)).all()
    red = np.zeros((n, 32), np.uint8)
    sim.sim_fr_reduce(_p(k), n_(n), _p(red))
    assert (red == oracle.fr_from_bytes_mod_order(k)).all()


def test_sqrt_free_compression_edges(sim, oracle):
    """The batched, square-root-free compression of the scalar-multiplication and Elligator kernels (curve.hpp,
    dcb_finish; reference: src/ark_curve/encoding.rs:91-128 via [k]P = [2]([k/2]P)) against the oracle and against
    the generic compressor, on the cases where its state degenerates: results equal to the identity (k = 0, k = r,
    the identity as base point: X = 0, where the reference's sqrt_ratio_zeta(1, 0) gives the all-zero encoding),
    invalid encodings in the middle of a round (their lanes must not poison the shared inversion), odd and even
    scalars (k/2 mod r takes both branches), and batch sizes around the round size of 32."""
    rng = np.random.default_rng(33)
    for n in (1, 2, 31, 32, 33, 64, 70):
        enc = oracle.encode_to_curve(rng.integers(0, 256, (n, 32), dtype=np.uint8))
        k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        k[0] = 0                                        # [0]P = identity
        if n > 2:
            k[1] = 0; k[1, 0] = 1                       # [1]P: k/2 = (1 + r)/2
            k[2] = 0; k[2, 0] = 2
            enc[n // 2] = 0                             # the identity as base point
        if n > 40:
            enc[35] = 0xFF                              # invalid (top bits), inside the second round
            enc[36] = 0; enc[36, 0] = 1                 # invalid (s negative)
            enc[37] = 0; enc[37, 0] = 4                 # most small s are not on the curve
        out = np.zeros((n, 32), np.uint8); st = np.zeros(n, np.uint8)
        out2 = np.zeros((n, 32), np.uint8); st2 = np.zeros(n, np.uint8)
        sim.sim_scalar_mul_var(_p(enc), _p(k), n_(n), _p(out), _p(st))
        sim.sim_scalar_mul_var_sqrt(_p(enc), _p(k), n_(n), _p(out2), _p(st2))
        oo, so = oracle.scalar_mul_var(enc, k)
        assert (st == so).all() and (st2 == so).all()
        assert (out == oo).all() and (out2 == oo).all()
        assert not out[0].any() and (n <= 2 or not out[n // 2].any())
        sim.sim_scalar_mul_base(_p(k), n_(n), _p(out))
        assert (out == oracle.scalar_mul_base(k)).all()
        r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        r0[0] = 0                                       # Elligator of 0
        if n > 2:
            r0[1] = 0; r0[1, 0] = 1
        sim.sim_encode_to_curve(_p(r0), n_(n), _p(out), None)
        sim.sim_encode_to_curve_sqrt(_p(r0), n_(n), _p(out2))
        assert (out == out2).all() and (out == oracle.encode_to_curve(r0)).all()
    # k / 2 mod r on its own: 2 * half == k (mod r)
    k = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    half = np.zeros_like(k)
    sim.sim_fr_half(_p(k), n_(64), _p(half))
    R = 2111115437357092606062206234695386632838870926408408195193685246394721360383
    for a, h in zip(k, half):
        kv, hv = int.from_bytes(bytes(a), "little") % R, int.from_bytes(bytes(h), "little")
        assert hv < R and (2 * hv - kv) % R == 0


def test_inverse_assisted_square_roots(sim, oracle):
    """The kernels hand every square root the inverse of its denominator (one divsteps inversion per lane per round,
    curve.hpp dcb_invert_slot) instead of letting it build den^(2^47-1) itself (src/ark_curve/invsqrt.rs:88-94).  The two
    forms must agree byte for byte -- raw sqrt_ratio_zeta roots and flags included -- with each other and with the
    oracle, on random pairs and on the early-out pairs (0, 1), (1, 0), (0, 0) placed inside and at the end of a round."""
    rng = np.random.default_rng(72)
    for n in (1, 15, 16, 17, 50):
        num = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        den = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        den[0] = 0                                          # den = 0: (false, 0)
        if n > 3:
            num[2] = 0                                      # num = 0: (true, 0)
            num[3] = 0; den[3] = 0
            den[n - 1] = 0
        r1, w1 = np.zeros((n, 32), np.uint8), np.zeros(n, np.uint8)
        r2, w2 = np.zeros((n, 32), np.uint8), np.zeros(n, np.uint8)
        sim.sim_sqrt_ratio_zeta(_p(num), _p(den), n_(n), _p(r1), _p(w1))
        sim.sim_sqrt_ratio_zeta_plain(_p(num), _p(den), n_(n), _p(r2), _p(w2))
        ro, wo = oracle.sqrt_ratio_zeta(num, den)
        assert (r1 == r2).all() and (w1 == w2).all() and (r1 == ro).all() and (w1 == wo).all()
        # compress with and without the inverse, identity and 2-torsion representative included
        P = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))
        P[0] = oracle.identity_xyzt()
        e1, e2 = np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8)
        sim.sim_compress_assisted(_p(P), n_(n), _p(e1))
        sim.sim_compress(_p(P), n_(n), _p(e2))
        assert (e1 == e2).all() and (e1 == oracle.compress(P)).all() and not e1[0].any()
        # hash_to_curve: two assisted square roots and the generic compressor
        a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        a[0] = 0
        h = np.zeros((n, 32), np.uint8)
        sim.sim_hash_to_curve(_p(a), _p(b), n_(n), _p(h))
        assert (h == oracle.hash_to_curve(a, b)).all()
        # decompression of raw strings (mostly invalid) and of valid encodings
        raw = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        raw[:, 31] &= 0x1F
        enc = np.concatenate([raw, oracle.compress(P)])
        x = np.zeros((2 * n, 16), np.uint64); st = np.zeros(2 * n, np.uint8)
        sim.sim_decompress_assisted(_p(enc), n_(2 * n), _p(x), _p(st))
        xo, so = oracle.decompress(enc)
        assert (x == xo).all() and (st == so).all()
        o1, s1 = np.zeros((2 * n, 32), np.uint8), np.zeros(2 * n, np.uint8)
        sim.sim_roundtrip_assisted(_p(enc), n_(2 * n), _p(o1), _p(s1))
        o2, s2 = oracle.roundtrip(enc)
        assert (o1 == o2).all() and (s1 == s2).all()
        # ... and with the compressor's denominators inverted together as well (k_roundtrip_chunked)
        o3, s3 = np.zeros((2 * n, 32), np.uint8), np.zeros(2 * n, np.uint8)
        sim.sim_roundtrip_chunked(_p(enc), n_(2 * n), _p(o3), _p(s3))
        assert (o3 == o2).all() and (s3 == s2).all()


def test_divsteps_inversion(sim, oracle):
    """fe_invert (inv30.hpp: 20 x 30 constant-time divsteps on signed 30-bit limbs) against big integers, against the
    x^(q-2) ladder and the square-root chain, and against the oracle's Fq inverse
    (src/fields/fq/u64/wrapper.rs:104-112): random values and 0, 1, 2, q-1, q-2, powers of two, values around 2^30k."""
    Q = 8444461749428370424248824938781546531375899335154063827935233455917409239041
    rng = np.random.default_rng(71)
    vals = [0, 1, 2, 3, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2] + [1 << k for k in range(0, 253, 7)]
    vals += [(1 << (30 * k)) - 1 for k in range(1, 9)] + [(1 << (30 * k)) + 1 for k in range(1, 9)]
    vals += [(1 << (29 * k)) - 1 for k in range(1, 9)]
    vals += [int.from_bytes(bytes(rng.integers(0, 256, 32, dtype=np.uint8)), "little") % Q for _ in range(400)]
    n = len(vals)
    # plain integers on 29-bit limbs
    x = np.array([[(v >> (29 * i)) & ((1 << 29) - 1) for i in range(9)] for v in vals], dtype=np.uint32)
    y = np.zeros_like(x)
    sim.sim_modinv_limbs29(_p(x), n_(n), _p(y))
    for v, row in zip(vals, y):
        got = sum(int(l) << (29 * i) for i, l in enumerate(row))
        assert got == (pow(v, -1, Q) if v else 0), v
    # field elements (Montgomery-256 words in and out)
    a = np.array([list(((v << 256) % Q).to_bytes(32, "little")) for v in vals], dtype=np.uint8).view(np.uint64).reshape(n, 4)
    g, l, c = np.zeros_like(a), np.zeros_like(a), np.zeros_like(a)
    sim.sim_invert(_p(a), n_(n), _p(g), _p(l), _p(c))
    assert (g == l).all() and (g == c).all()
    want, st = oracle.fq_op(5, a)
    assert (g == want).all()
    for v, row in zip(vals, g):
        got = int.from_bytes(row.tobytes(), "little") * pow(1 << 256, -1, Q) % Q
        assert got == (pow(v, -1, Q) if v else 0)


def test_fr_arithmetic_matches_oracle(sim, oracle):
    """The scalar-field arithmetic of curve.hpp (word-level Montgomery, what k_fr_op runs) against the oracle's
    bit-serial restatement, on random and edge operands (0, 1, r - 1, r, 2^256 - 1)."""
    rng = np.random.default_rng(15)
    n = 300
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    edges = [0, 1, 2, R_ORDER - 1, R_ORDER, R_ORDER + 1, (1 << 256) - 1, 1 << 255]
    for i, v in enumerate(edges):
        a[i] = np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
        b[len(edges) - 1 - i] = np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
    out = np.zeros((n, 32), np.uint8)
    st = np.zeros(n, np.uint8)
    for op in range(6):
        sim.sim_fr_op(ctypes.c_int(op), _p(a), _p(b), n_(n), _p(out), _p(st))
        oo, so = oracle.fr_op(op, a, b if op <= 2 else None)
        assert (out == oo).all() and (st == so).all(), op
    for length in (48, 64):
        d = rng.integers(0, 256, (n, length), dtype=np.uint8)
        d[0] = 255
        d[1] = 0
        sim.sim_fr_from_wide(_p(d), ctypes.c_int(length), n_(n), _p(out))
        assert (out == oracle.fr_from_wide_bytes(d)).all()


def test_conversion_free_forms_match_oracle(sim, oracle):
    """k_add / k_double / k_eq / k_neg / k_fq_op read Montgomery-256 records without converting them (the words,
    taken as limbs, are the value scaled by 2^-5) and fold the scale into the product that writes the result:
    the words they store must still be the reference formulas' (oracle add / double / neg / eq, Fq ops), identity
    and zero coordinates included; a non-canonical word string is reported, not mis-negated."""
    rng = np.random.default_rng(16)
    n = 64
    P = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    Q2 = oracle.double_xyzt(oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
    P[0] = oracle.identity_xyzt()
    Q2[1] = oracle.identity_xyzt()
    Q2[2] = P[2]
    Q2[3] = oracle.add_xyzt(P[3:4], P[3:4])[0]
    Q2[4] = oracle.neg_xyzt(P[4:5])[0]
    z = lambda *shape: np.zeros(shape, np.uint64)
    s_, d_, ng = z(n, 16), z(n, 16), z(n, 16)
    fm, fs, fa, fb, fn = z(n, 4), z(n, 4), z(n, 4), z(n, 4), z(n, 4)
    eq, nok, fok = np.zeros(n, np.uint8), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    sim.sim_raw_forms(_p(P), _p(Q2), n_(n), _p(s_), _p(d_), _p(ng), _p(eq), _p(nok), _p(fm), _p(fs), _p(fa), _p(fb), _p(fn), _p(fok))
    assert (s_ == oracle.add_xyzt(P, Q2)).all()
    df = z(n, 16)
    sim.sim_raw_ge_sub(_p(P), _p(Q2), n_(n), _p(df))              # Element - Element = self + other.neg(), ops.rs:43-49
    assert (df == oracle.add_xyzt(P, oracle.neg_xyzt(Q2))).all()
    assert (d_ == oracle.double_xyzt(P)).all()
    assert (ng == oracle.neg_xyzt(P)).all() and nok.all() and fok.all()
    assert (eq == oracle.eq_xyzt(P, Q2)).all() and eq[2] == 1 and eq[0] == 0
    # the field ops take the X words of the two records (rows of 16 u64: X is the first four)
    a, b = np.ascontiguousarray(P[:, :4]), np.ascontiguousarray(Q2[:, :4])
    for got, op in ((fa, 0), (fb, 1), (fm, 2), (fs, 3), (fn, 4)):
        assert (got == oracle.fq_op(op, a, b if op <= 2 else None)[0]).all(), op
    # a non-canonical X (q + 5 as words): the word-level forms decline
    bad = P.copy()
    q = 8444461749428370424248824938781546531375899335154063827935233455917409239041
    bad[0, :4] = np.frombuffer(int(q + 5).to_bytes(32, "little"), np.uint64)
    sim.sim_raw_forms(_p(bad), _p(Q2), n_(1), _p(s_), _p(d_), _p(ng), _p(eq), _p(nok), _p(fm), _p(fs), _p(fa), _p(fb), _p(fn), _p(fok))
    assert nok[0] == 0 and fok[0] == 0


def _val(limbs):
    return sum(int(v) << (29 * i) for i, v in enumerate(limbs))


def test_limb_bounds_adversarial(sim):
    """fq29.hpp's representation contract at its edges: every limb at the documented maximum
    (lazy = 2^30 + 16 on both sides; carried + 2^30 against a carried operand), so that each 64-bit
    column accumulator is pushed as close to 2^64 as the contract allows -- with the 32-bit Montgomery
    digit of the relaxed multiplier and with the 29-bit digit of the strict one.  Checked against
    exact big-integer arithmetic: result = a*b / 2^261 mod q, product limbs, value < a*b/2^261 + 8q
    (relaxed) or + q (strict)."""
    R = 1 << 261
    Rinv = pow(R, -1, Q)
    lazy = (1 << 30) + 16
    carried = (1 << 29) + 8
    wide = (1 << 30) + (1 << 29) + 16        # fe_sub_nc output: carried + a 2^30 offset digit
    top = (1 << 24)                         # limb 8 of a value around 13 q
    rng = np.random.default_rng(15)
    cases = []
    for la, lb in [(lazy, lazy), (wide, carried), (carried, wide), (lazy, carried), (carried, carried)]:
        cases.append(([la] * 8 + [top], [lb] * 8 + [top]))
        cases.append(([la] * 8 + [0], [lb] * 8 + [0]))
        for _ in range(200):                # random mixtures of extreme and random limbs
            a = [la if rng.random() < 0.7 else int(rng.integers(0, la + 1)) for _ in range(8)] + [int(rng.integers(0, top))]
            b = [lb if rng.random() < 0.7 else int(rng.integers(0, lb + 1)) for _ in range(8)] + [int(rng.integers(0, top))]
            cases.append((a, b))
    a = np.array([c[0] for c in cases], dtype=np.uint32)
    b = np.array([c[1] for c in cases], dtype=np.uint32)
    n = a.shape[0]
    out = np.zeros((n, 9), np.uint32)
    for mode, slack in ((0, 8), (1, 1)):
        sim.sim_raw_mul(mode, _p(a), _p(b), n_(n), _p(out))
        for i in range(n):
            va, vb, vr = _val(a[i]), _val(b[i]), _val(out[i])
            assert vr % Q == va * vb * Rinv % Q, (mode, i)
            assert vr < va * vb // R + slack * Q + 1
            assert all(int(x) < (1 << 29) for x in out[i][:8])
    for mode, slack, scale in ((2, 8, 1), (3, 1, 1), (4, 8, 2)):
        sel = [i for i in range(n) if max(int(x) for x in a[i][:8]) <= (lazy if scale == 1 else carried)]
        aa = np.ascontiguousarray(a[sel])
        oo = np.zeros((len(sel), 9), np.uint32)
        sim.sim_raw_mul(mode, _p(aa), _p(aa), n_(len(sel)), _p(oo))
        for i in range(len(sel)):
            va, vr = _val(aa[i]), _val(oo[i])
            assert vr % Q == scale * va * va * Rinv % Q, (mode, i)
            assert vr < scale * va * va // R + slack * Q + 1
            assert all(int(x) < (1 << 29) for x in oo[i][:8])
    # subtraction: minuend lazy, subtrahend lazy with value < 31q -> exact value a - b + 32q, carried limbs
    sub_a = np.array([[lazy] * 8 + [top]] * 4 + [[0] * 9] * 4, dtype=np.uint32)
    sub_b = np.array([[lazy] * 8 + [0], [0] * 9, [carried] * 8 + [1 << 21], [lazy] * 8 + [(1 << 24)]] * 2, dtype=np.uint32)
    out = np.zeros((8, 9), np.uint32)
    sim.sim_raw_sub(0, _p(sub_a), _p(sub_b), n_(8), _p(out))
    for i in range(8):
        assert _val(sub_b[i]) < 31 * Q
        assert _val(out[i]) == _val(sub_a[i]) - _val(sub_b[i]) + 32 * Q
        assert all(int(x) < (1 << 29) + 8 for x in out[i][:8])
    # no-carry subtraction: carried subtrahend, result = a - b + 16q limb by limb, limbs < a's + 2^30 + 8
    nc_a = np.array([[carried] * 8 + [top], [0] * 9, [carried] * 8 + [0], [5] * 9], dtype=np.uint32)
    nc_b = np.array([[carried] * 8 + [1 << 24], [carried] * 8 + [1 << 24], [0] * 9, [carried] * 8 + [7]], dtype=np.uint32)
    out = np.zeros((4, 9), np.uint32)
    sim.sim_raw_sub(1, _p(nc_a), _p(nc_b), n_(4), _p(out))
    for i in range(4):
        assert _val(out[i]) == _val(nc_a[i]) - _val(nc_b[i]) + 16 * Q
        assert all(int(x) < int(y) + (1 << 30) + 8 for x, y in zip(out[i][:8], nc_a[i][:8]))
    # canonicalisation of the representatives of zero and of small multiples of q (relaxed products reach 8q+)
    reps = []
    for k in range(0, 12):
        v = k * Q
        reps.append([(v >> (29 * i)) & ((1 << 29) - 1) for i in range(8)] + [v >> 232])
    reps = np.array(reps, dtype=np.uint32)
    out = np.zeros((12, 9), np.uint32)
    sim.sim_raw_canon(_p(reps), n_(12), _p(out))
    assert not out.any()


def test_static_bounds(oracle):
    """The same headers built with -DD377_BOUNDS: every field element carries its worst-case limb and
    value bounds (over all inputs, not the ones of this run) and every primitive asserts its
    precondition -- column sums below 2^64, subtrahend digits below the offset digits, values that are
    compared with q below 2q, hash keys below q + 2^248.  One pass through every curve function
    therefore proves the bounds of every call site.  Runs in a child process: a violation aborts."""
    import sys
    lib = os.path.join(SIM_DIR, "libd377_sim_bounds.so")
    srcs = [os.path.join(SIM_DIR, "sim.cpp")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    if not os.path.exists(lib) or any(os.path.getmtime(s_) > os.path.getmtime(lib) for s_ in srcs):
        # (-DD377_FB_BITS=8: the small fixed-base comb, so that the unoptimised build's table construction stays short;
        #  the additions it feeds are the same code)
        subprocess.check_call(["g++", "-O0", "-g", "-std=c++17", "-fPIC", "-shared", "-DD377_BOUNDS", "-DD377_FB_BITS=8", "-I" + CSRC,
                               os.path.join(SIM_DIR, "sim.cpp"), "-o", lib])
    code = r"""
import ctypes, sys, numpy as np
L = ctypes.CDLL(sys.argv[1])
L.sim_init.restype = ctypes.c_int
assert L.sim_init() == 0
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
n_ = ctypes.c_size_t
rng = np.random.default_rng(1)
n = 8
r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8); k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
enc = np.zeros((n, 32), np.uint8); xyzt = np.zeros((n, 16), np.uint64); out = np.zeros((n, 32), np.uint8)
st = np.zeros(n, np.uint8); x2 = np.zeros((n, 16), np.uint64); a = np.zeros((n, 16), np.uint64); b = np.zeros((n, 16), np.uint64)
L.sim_encode_to_curve(p(r0), n_(n), p(enc), p(xyzt))
L.sim_decompress(p(enc), n_(n), p(x2), p(st))
L.sim_roundtrip(p(k), n_(n), p(out), p(st))
L.sim_compress(p(xyzt), n_(n), p(out))
L.sim_sqrt_ratio_zeta(p(r0), p(k), n_(n), p(out), p(st))
L.sim_scalar_mul_var(p(enc), p(k), n_(n), p(out), p(st))
L.sim_scalar_mul_base(p(k), n_(n), p(out))
L.sim_double_variants(p(xyzt), n_(n), p(x2), p(a), p(b))
L.sim_group_misc(p(xyzt), p(x2), n_(n), p(a), p(b))
L.sim_quad_forms(p(xyzt), p(x2), n_(n), p(a), p(b), p(np.zeros((n, 16), np.uint64)), p(np.zeros((n, 16), np.uint64)))
for lazy in (0, 1):
    L.sim_row_records(p(xyzt), n_(n), lazy, p(a), p(b), p(out)); L.sim_row_records(p(x2), n_(n), lazy, p(a), p(b), p(out))
L.sim_tiny4(p(r0), n_(n), p(np.zeros((n, 32), np.uint8)), p(out), p(a), p(st)); L.sim_tiny4(p(np.zeros((n, 32), np.uint8)), n_(4), p(np.zeros((n, 32), np.uint8)), p(out), p(a), p(st))
L.sim_to_affine_raw(p(xyzt), n_(n), p(np.zeros((n, 8), np.uint64))); L.sim_to_affine_raw(p(np.full((n, 16), 0xFFFFFFFFFFFFFFFF, np.uint64)), n_(n), p(np.zeros((n, 8), np.uint64)))
L.sim_decompress(p(enc), n_(n), p(x2), p(st)); L.sim_msm_bucket(p(x2), p(np.array([0, 1, 1, 0, 0, 1, 0, 1], np.uint8)), n_(n), p(a)); L.sim_msm_bucket(p(x2), p(np.ones(n, np.uint8)), n_(2), p(a)); L.sim_msm_bucket(p(x2), p(np.zeros(n, np.uint8)), n_(3), p(a))
f = [np.zeros((n, 4), np.uint64) for _ in range(5)]
fl = [np.zeros(n, np.uint8) for _ in range(3)]
full = np.full((n, 16), 0xFFFFFFFFFFFFFFFF, np.uint64)      # every word string the conversion-free forms may be handed
for u, v in ((xyzt, x2), (full, full)):
    L.sim_raw_forms(p(u), p(v), n_(n), p(a), p(b), p(np.zeros((n, 16), np.uint64)), p(fl[0]), p(fl[1]), p(f[0]), p(f[1]), p(f[2]), p(f[3]), p(f[4]), p(fl[2]))
L.sim_raw_ge_sub(p(xyzt), p(x2), n_(n), p(a)); L.sim_raw_ge_sub(p(full), p(full), n_(n), p(a))
L.sim_hash_to_curve_quartic(p(r0), p(k), n_(n), p(out), 0, None, None); L.sim_hash_to_curve_quartic(p(r0), p(k), n_(n), p(out), 1, None, None)
L.sim_hash_to_curve(p(r0), p(k), n_(n), p(out)); L.sim_compress_assisted(p(xyzt), n_(n), p(out)); L.sim_sqrt_ratio_zeta_plain(p(r0), p(k), n_(n), p(out), p(st))
L.sim_decompress_assisted(p(enc), n_(n), p(x2), p(st)); L.sim_roundtrip_assisted(p(k), n_(n), p(out), p(st))
L.sim_roundtrip_chunked(p(enc), n_(n), p(out), p(st)); L.sim_roundtrip_chunked(p(k), n_(n), p(out), p(st))
L.sim_scalar_mul_var_sqrt(p(enc), p(k), n_(n), p(out), p(st)); L.sim_encode_to_curve_sqrt(p(r0), n_(n), p(out))
for m in (1, 3, 8):                                         # the Straus chain of d377_batch_msm_small (straus.hpp): 8 // m sums of m terms
    L.sim_batch_msm(p(xyzt), p(k), ctypes.c_int(m), n_(n // m), p(out))
L.sim_batch_msm(p(np.zeros((n, 16), np.uint64)), p(k), ctypes.c_int(2), n_(2), p(out))   # records with Z = 0
w = np.zeros((n, 4), np.uint64)
L.sim_fq_mul(p(xyzt), p(x2), n_(n), p(w)); L.sim_fq_sub(p(xyzt), p(x2), n_(n), p(w)); L.sim_fq_add(p(xyzt), p(x2), n_(n), p(w))
print("BOUNDS_OK")
"""
    r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "BOUNDS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_batch_msm_straus_chain_matches_oracle_fold(sim, oracle):
    """d377_batch_msm_small's lane kernel on the host (straus.hpp: the device's own chain -- m tables of cached 0 .. 8 P, the
    digit words, 252 shared doublings -- under the kernels' round structure): n sums of m terms against the oracle's fold of its
    own products (src/ark_curve/element/projective.rs:99-117; tests/operations.rs:44-60), m = 1 .. 8, with scalars 0, 1,
    r - 1, r, 2^256 - 1, the identity, projective Elements and a record with Z = 0 (the identity by contract) mixed in."""
    rng = np.random.default_rng(8100)
    R = 2111115437357092606062206234695386632838870926408408195193685246394721360383
    ident = oracle.decompress(np.zeros((1, 32), np.uint8))[0][0]
    for m in (1, 2, 3, 5, 8):
        n = 11
        terms = n * m
        P = oracle.elligator_map_xyzt(rng.integers(0, 256, (terms, 32), dtype=np.uint8))
        k = rng.integers(0, 256, (terms, 32), dtype=np.uint8)
        for i, v in enumerate([0, 1, R - 1, R, (1 << 256) - 1]):
            k[(3 * i) % terms] = np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
        P[terms - 1] = ident
        if terms >= 4:
            P[2:4] = oracle.scalar_mul_xyzt(P[2:4], k[0:2])                      # projective representatives (Z != 1)
        Pz = P.copy()
        Pref = P.copy()
        if terms >= 6:
            Pz[5] = 0                                                            # Z = 0
            Pref[5] = ident
        prod = oracle.scalar_mul_xyzt(Pref, k)
        acc = prod[0::m].copy()
        for j in range(1, m):
            acc = oracle.add_xyzt(acc, prod[j::m])
        want = oracle.compress(acc)
        out = np.zeros((n, 32), np.uint8)
        sim.sim_batch_msm(_p(np.ascontiguousarray(Pz)), _p(k), ctypes.c_int(m), n_(n), _p(out))
        assert (out == want).all(), (m, np.nonzero((out != want).any(axis=1))[0])


def test_msm_bucket_chain_matches_oracle(sim, oracle):
    """The per-lane chain of the MSM's bucket sums (msm.hip k_msm_spans / k_msm_reduce): cached affine records, the
    first point of a run converted instead of added to the identity, mixed additions, full additions of partial sums --
    the same group element as the oracle's fold, for every sign pattern of a short run."""
    rng = np.random.default_rng(29)
    for n in (2, 3, 5, 8, 33):
        for rep in range(6):
            enc = oracle.encode_to_curve(rng.integers(0, 256, (n, 32), dtype=np.uint8))
            P, st = oracle.decompress(enc)
            negs = rng.integers(0, 2, n).astype(np.uint8) if rep else np.zeros(n, np.uint8)
            out = np.zeros((1, 16), np.uint64)
            sim.sim_msm_bucket(_p(P), _p(negs), n_(n), _p(out))
            signed = [oracle.neg_xyzt(P[i:i + 1]) if negs[i] else P[i:i + 1] for i in range(n)]
            a = signed[0]
            for i in range(1, n // 2):
                a = oracle.add_xyzt(a, signed[i])
            b = signed[n // 2]
            for i in range(n // 2 + 1, n):
                b = oracle.add_xyzt(b, signed[i])
            want = oracle.add_xyzt(oracle.add_xyzt(a, b), oracle.add_xyzt(a, a))
            assert bytes(oracle.compress(out)[0]) == bytes(oracle.compress(want)[0]), (n, rep)


def test_hash_to_curve_on_the_quartic(sim, oracle):
    """hash_to_curve with the sum formed on the Jacobi quartic and encoded without a square root (curve.hpp
    ge_dcb_from_jacobi_sum) gives the reference's bytes (src/ark_curve/elligator.rs:67-71, oracle): random pairs, r2 = r1
    (a doubling), r2 = -r1, zeros; the route for exceptional pairs forced for every element; and constructed pairs
    (s, t), (+-1/s, +-t/s^2) that really hit the exceptional case, against the big-integer model."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("d377_model", os.path.join(ROOT, "oracle", "d377_model.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    rng = np.random.default_rng(31)
    n = 96
    r1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    r2 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    r2[0] = r1[0]
    neg = (M.Q - M.fq_from_le_bytes_mod_order(bytes(r1[1]))) % M.Q
    r2[1] = np.frombuffer(neg.to_bytes(32, "little"), np.uint8)
    r1[2] = 0
    r1[3] = 0; r2[3] = 0
    want = oracle.hash_to_curve(r1, r2)
    for force in (0, 1):
        out = np.zeros((n, 32), np.uint8)
        exc = np.zeros(n, np.uint8)
        sim.sim_hash_to_curve_quartic(_p(r1), _p(r2), n_(n), _p(out), force, None, _p(exc))
        assert (out == want).all(), force
        assert not exc.any()
    # constructed exceptional pairs: (s, t) and (e1 / s, e2 t / s^2) are both on the quartic and s1 s2 = e1 = +-1
    Q, D = M.Q, M.COEFF_D

    def phi(s, t):
        E, F, G = 2 * s % Q, (1 - s * s) % Q, (1 + s * s) % Q
        return (E * t % Q, F * G % Q, F * t % Q, E * G % Q)

    m = 16
    pairs = np.zeros((m, 4, 32), np.uint8)
    expect = []
    for i in range(m):
        r = M.fq_from_le_bytes_mod_order(bytes(r2[8 + i]))
        rr = M.ZETA * r % Q * r % Q
        # (s, t) of the map of r, by the model's own elligator (affine x, y -> not needed: rebuild s, t from its steps)
        A = M.COEFF_A
        den = (D * rr - (D - A)) % Q * (((D - A) * rr - D) % Q) % Q
        num = (rr + 1) * (A - 2 * D) % Q
        iss, isri = M.sqrt_ratio_zeta(1, num * den % Q)
        sgn, tw = (1, 1) if iss else (Q - 1, r % Q)
        isri = isri * tw % Q
        s = isri * num % Q
        t = ((-sgn) * isri % Q * s % Q * (rr - 1) % Q * pow((A - 2 * D) % Q, 2, Q) - 1) % Q
        if M.is_negative(s) == iss:
            s = (-s) % Q
        e1, e2 = (1, Q - 1)[i & 1], (1, Q - 1)[(i >> 1) & 1]
        si = pow(s, Q - 2, Q)
        s2, t2 = e1 * si % Q, e2 * t % Q * si % Q * si % Q
        for c, v in enumerate((s, t, s2, t2)):
            pairs[i, c] = np.frombuffer(v.to_bytes(32, "little"), np.uint8)
        expect.append(bytes(M.compress(M.pt_add(phi(s, t), phi(s2, t2)))))
    out = np.zeros((m, 32), np.uint8)
    exc = np.zeros(m, np.uint8)
    sim.sim_hash_to_curve_quartic(_p(r1[:m].copy()), _p(r2[:m].copy()), n_(m), _p(out), 0, _p(pairs), _p(exc))
    assert exc.all()
    assert [bytes(o) for o in out] == expect


def test_doubling_variants_agree(sim, oracle):
    """ge_double (reference formulas) equals the oracle limb for limb; ge_double_fast (sign-folded)
    gives the same group element; ge_double_neg (the scalar-multiplication loops' doubling) gives its
    negative, coordinate for coordinate (-X, Y, Z, -T up to the common projective factor)."""
    rng = np.random.default_rng(16)
    n = 512
    P = oracle.double_xyzt(oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
    ref = np.zeros((n, 16), np.uint64)
    fast = np.zeros((n, 16), np.uint64)
    negd = np.zeros((n, 16), np.uint64)
    sim.sim_double_variants(_p(P), n_(n), _p(ref), _p(fast), _p(negd))
    assert (ref == oracle.double_xyzt(P)).all()
    assert oracle.eq_xyzt(fast, ref).all()
    assert (oracle.compress(fast) == oracle.compress(ref)).all()
    back = oracle.neg_xyzt(negd)
    assert oracle.eq_xyzt(back, ref).all() and (oracle.compress(back) == oracle.compress(ref)).all()
    assert oracle.is_identity(oracle.add_xyzt(negd, ref)).all()


def test_four_lane_forms_agree(sim, oracle):
    """The doubling and the addition as the quads of lanes run them (quad_ops.hpp: one linear combination per lane between
    the two rounds of products), lanes emulated on the host with the device's own per-lane arithmetic: [4]P, [4]P + Q,
    [4]P - Q and a doubling-then-subtraction step of the chains, against the oracle."""
    rng = np.random.default_rng(17)
    n = 256
    P = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    Q = oracle.double_xyzt(oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
    P[0] = oracle.identity_xyzt()
    Q[1] = oracle.identity_xyzt()
    Q[2] = oracle.double_xyzt(oracle.double_xyzt(P[2:3]))[0]          # [4]P - Q = identity
    out = [np.zeros((n, 16), np.uint64) for _ in range(4)]
    sim.sim_quad_forms(_p(P), _p(Q), n_(n), *[_p(o) for o in out])
    p4 = oracle.double_xyzt(oracle.double_xyzt(P))
    s1 = oracle.add_xyzt(p4, Q)
    want = [p4, s1, oracle.add_xyzt(p4, oracle.neg_xyzt(Q)), oracle.add_xyzt(oracle.double_xyzt(s1), oracle.neg_xyzt(Q))]
    for got, w in zip(out, want):
        assert oracle.eq_xyzt(got, w).all() and (oracle.compress(got) == oracle.compress(w)).all()
    assert oracle.is_identity(out[2][2:3]).all()


def test_row_records_round_trip(sim, oracle):
    """Whole elements in and out of the lane-spread form's records (curve.hpp fe_to_limbs28 / fe_from_limbs28: what
    k_msm_final, k_msm_tiny and k_scalar_mul_var_tiny do at either end of their chains): the point that comes back is the
    point that went in, also when the record's limbs are lazily reduced (above 2^28, as the rows leave them); its doubling
    and the square-root-free encoding of that doubling are the oracle's."""
    rng = np.random.default_rng(41)
    n = 48
    P = oracle.double_xyzt(oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
    P[0] = oracle.identity_xyzt()
    for lazy in (0, 1):
        back, dbl = np.zeros((n, 16), np.uint64), np.zeros((n, 16), np.uint64)
        enc = np.zeros((n, 32), np.uint8)
        sim.sim_row_records(_p(P), n_(n), lazy, _p(back), _p(dbl), _p(enc))
        assert (back == P).all()                                  # canonical limbs both ways: the same records
        want = oracle.double_xyzt(P)
        assert oracle.eq_xyzt(dbl, want).all()
        assert (enc == oracle.compress(want)).all()


def test_four_per_wave_whole_element_steps(sim, oracle):
    """What the four-elements-per-wave kernels (d377.hip k_*_tiny) do in whole-element code: the square root's powers handed
    over as values (curve.hpp GivenPowers) to the Elligator map, the decompression and the generic compressor, and the
    encodings of four elements from one shared inversion (tiny4_encode / dcb_encode_one) -- all equal to the oracle's."""
    rng = np.random.default_rng(43)
    n = 64
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    r0[0] = 0
    r0[5] = r0[4]                                            # equal elements inside a quad
    enc_a, enc_b = np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8)
    xyzt, st = np.zeros((n, 16), np.uint64), np.zeros(n, np.uint8)
    sim.sim_tiny4(_p(r0), n_(n), _p(enc_a), _p(enc_b), _p(xyzt), _p(st))
    want = oracle.encode_to_curve(r0)
    assert (enc_a == want).all() and (enc_b == want).all()
    o_xyzt, o_st = oracle.decompress(want)
    assert (st == 0).all() and (o_st == 0).all() and (xyzt == o_xyzt).all()


def test_to_affine_on_raw_records(sim, oracle):
    """normalize_batch as k_to_affine computes it -- Montgomery's trick on the records as they lie in memory, the power of
    two of the skipped conversions folded into the lane's one inverse -- against the oracle's x / z, y / z; a record with
    z = 0 (in its canonical and in a non-canonical spelling: q itself) gives a zero record and leaves the others intact;
    a non-canonical but non-zero z (z + q) gives the same affine point as z."""
    rng = np.random.default_rng(18)
    n = 64
    P = oracle.double_xyzt(oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
    want = oracle.to_affine(P)
    got = np.zeros((n, 8), np.uint64)
    sim.sim_to_affine_raw(_p(P), n_(n), _p(got))
    assert (got == want).all()
    def words(v):
        return np.frombuffer(int(v).to_bytes(32, "little"), np.uint64)
    bad = P.copy()
    bad[3, 8:12] = 0
    bad[7, 8:12] = words(Q)                                   # zero mod q, spelled q
    z9 = int.from_bytes(P[9, 8:12].tobytes(), "little")
    bad[9, 8:12] = words(z9 + Q)                              # the same z, not canonical
    sim.sim_to_affine_raw(_p(bad), n_(n), _p(got))
    keep = np.ones(n, bool)
    keep[[3, 7]] = False
    assert (got[keep] == want[keep]).all() and not got[3].any() and not got[7].any()


def test_msm_windows_digits_and_span_plan(sim):
    """msm_plan.hpp, the code the MSM kernels run, on the host: (1) for every window width c = 4 .. 16 the windows -- the first
    `nwide` c bits wide, the rest c - 1 -- tile the 252 scalar bits, and the signed digits of a scalar recompose to
    k / 2 mod r (the MSM sums with k / 2 and doubles at the end) with |digit| <= 2^(width - 1) and an unwrapped,
    non-negative top digit inside the buckets of its width, for random, zero, tiny, near-r and all-ones scalars;
    (2) the span plan: with every lane taking L consecutive sorted entries, a bucket is touched by exactly the lanes the
    closed form names, and slot = lane + (non-empty buckets before) is strictly increasing along the entries -- no two
    partial sums share a slot."""
    L = sim
    rng = np.random.default_rng(55)
    n = 400
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k[0] = 0
    k[1, :] = 0; k[1, 0] = 1
    k[2] = np.frombuffer((R_ORDER - 1).to_bytes(32, "little"), np.uint8)
    k[3] = np.frombuffer(R_ORDER.to_bytes(32, "little"), np.uint8)
    k[4] = 255
    k[5, 8:] = 0                                                   # a 64-bit scalar
    kw = np.ascontiguousarray(k).view(np.uint32).reshape(n, 8)
    for c in range(4, 17):
        shape = np.zeros(2, np.int32)
        dig = np.zeros((n, 64), np.int32)
        L.sim_msm_digits(kw.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), c, shape.ctypes.data_as(ctypes.c_void_p), dig.ctypes.data_as(ctypes.c_void_p))
        W, nwide = int(shape[0]), int(shape[1])
        widths = [c if w < nwide else c - 1 for w in range(W)]
        assert W == -(-252 // c) and sum(widths) == 252 and W <= 63, (c, W, nwide)
        first = [sum(widths[:w]) for w in range(W)]
        for i in range(n):
            kv = int.from_bytes(bytes(k[i]), "little") % R_ORDER
            half = kv // 2 if kv % 2 == 0 else (kv + R_ORDER) // 2
            d = [int(x) for x in dig[i, :W]]
            assert dig[i, 63] == 0 and sum(dv << first[w] for w, dv in enumerate(d)) == half, (c, i)
            assert all(abs(dv) <= 1 << (widths[w] - 1) for w, dv in enumerate(d)), (c, i)
            assert 0 <= d[W - 1] <= 1 << (widths[W - 1] - 1), (c, i)
    # (2) the span plan on random bucket sizes (many empty, some huge), several L
    for trial in range(40):
        nb = int(rng.integers(1, 300))
        size = rng.integers(0, 40, nb).astype(np.uint32)
        size[rng.random(nb) < 0.3] = 0
        if trial % 5 == 0:
            size[int(rng.integers(0, nb))] = 5000
        o = np.concatenate(([0], np.cumsum(size)[:-1])).astype(np.uint32)
        Ls = int(rng.integers(1, 70))
        fl = np.zeros(nb, np.uint32); pc = np.zeros(nb, np.uint32)
        L.sim_span_plan(o.ctypes.data_as(ctypes.c_void_p), size.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nb), Ls,
                        fl.ctypes.data_as(ctypes.c_void_p), pc.ctypes.data_as(ctypes.c_void_p))
        bucket_of = np.repeat(np.arange(nb), size)                 # entry -> bucket
        lane_of = np.arange(bucket_of.size) // Ls                  # entry -> lane
        ne = np.concatenate(([0], np.cumsum(size != 0)[:-1]))      # non-empty buckets before b
        slots = []
        for b in range(nb):
            lanes = np.unique(lane_of[bucket_of == b])
            assert pc[b] == lanes.size and (lanes.size == 0 or (fl[b] == lanes[0] and lanes[-1] == lanes[0] + lanes.size - 1)), (trial, b)
            slots += [int(l) + int(ne[b]) for l in lanes]
        assert all(a < b_ for a, b_ in zip(slots, slots[1:])), trial


def test_bench_mac_counts():
    """bench.py's KERNEL_OPS (field products / squarings per element, the numerator of roofline_valu) are the
    counts the instrumented host build of the same headers reports for one element of each operation."""
    import importlib.util
    import sys
    lib = os.path.join(SIM_DIR, "libd377_sim_bounds.so")
    if not os.path.exists(lib):
        pytest.skip("bounds build not present (test_static_bounds builds it)")
    code = r"""
import ctypes, sys, numpy as np
L = ctypes.CDLL(sys.argv[1]); L.sim_init.restype = ctypes.c_int; assert L.sim_init() == 0
p = lambda a: a.ctypes.data_as(ctypes.c_void_p); n_ = ctypes.c_size_t
rng = np.random.default_rng(1)
m = ctypes.c_ulong(); s = ctypes.c_ulong()
def run(n, what):
    # n = the elements a lane finishes with one inversion in the benchmark's launch: 2^22 / (512 blocks x 256 lanes) = 32 for
    # the variable-base workload (a full round), 2^20 / 131072 = 8 for the fixed-base extra
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8); k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    enc = np.zeros((n, 32), np.uint8); out = np.zeros((n, 32), np.uint8); st = np.zeros(n, np.uint8)
    L.sim_encode_to_curve(p(r0), n_(n), p(enc), None); L.sim_op_counts(ctypes.byref(m), ctypes.byref(s))
    if what == "scalar_mul_var": L.sim_scalar_mul_var(p(enc), p(k), n_(n), p(out), p(st))
    if what == "roundtrip": L.sim_roundtrip(p(enc), n_(n), p(out), p(st))
    if what == "scalar_mul_base_w8": L.sim_scalar_mul_base(p(k), n_(n), p(out))
    if what == "sqrt_ratio_zeta": L.sim_sqrt_ratio_zeta(p(r0), p(k), n_(n), p(out), p(st))
    if what == "encode_to_curve": L.sim_encode_to_curve(p(r0), n_(n), p(enc), None)
    if what == "hash_to_curve": L.sim_hash_to_curve_quartic(p(r0), p(k), n_(n), p(out), 0, None, None)
    if what == "decompress": L.sim_decompress(p(enc), n_(n), p(np.zeros((n, 16), np.uint64)), p(st))
    if what == "decompress_chunked": L.sim_decompress_assisted(p(enc), n_(n), p(np.zeros((n, 16), np.uint64)), p(st))
    if what == "roundtrip_chunked": L.sim_roundtrip_chunked(p(enc), n_(n), p(out), p(st))
    if what == "compress_chunked":
        x = np.zeros((n, 16), np.uint64); L.sim_decompress(p(enc), n_(n), p(x), p(st)); L.sim_op_counts(ctypes.byref(m), ctypes.byref(s))
        L.sim_compress_assisted(p(x), n_(n), p(out))
    if what == "compress":
        x = np.zeros((n, 16), np.uint64); L.sim_decompress(p(enc), n_(n), p(x), p(st)); L.sim_op_counts(ctypes.byref(m), ctypes.byref(s))
        L.sim_compress(p(x), n_(n), p(out))
    L.sim_op_counts(ctypes.byref(m), ctypes.byref(s)); print(what, m.value / n, s.value / n)
run(32, "scalar_mul_var"); run(8, "roundtrip"); run(8, "scalar_mul_base_w8"); run(8, "sqrt_ratio_zeta")
run(8, "encode_to_curve"); run(8, "hash_to_curve"); run(8, "decompress"); run(8, "compress"); run(8, "decompress_chunked")
run(8, "compress_chunked"); run(8, "roundtrip_chunked")
"""
    r = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = {l.split()[0]: (float(l.split()[1]), float(l.split()[2])) for l in r.stdout.splitlines() if l.strip()}
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for name in ("scalar_mul_var", "roundtrip", "sqrt_ratio_zeta", "encode_to_curve", "hash_to_curve", "decompress", "compress", "decompress_chunked",
                 "compress_chunked", "roundtrip_chunked"):
        assert got[name] == b.KERNEL_OPS[name], (name, got[name])
    # the bounds build uses the 8-bit comb (32 mixed additions of 7 products); the product build's 23-bit comb has 11
    m8, s8 = got["scalar_mul_base_w8"]
    assert (m8 - 21 * 7, s8) == b.KERNEL_OPS["scalar_mul_base"]
    assert b.KERNEL_MACS["scalar_mul_var"] == 1668.5 * 153 + 1009.0 * 117 + 2 * 20 * 90 / 8.0
