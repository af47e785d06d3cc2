"""Parity tests proper: the HIP kernels, called through the C ABI (include/decaf377_amd.h),
against the oracle on the same seeded inputs, against the committed golden fixtures, and --
at BASELINE.json's full sizes -- through size-independent properties.  Bit-exact everywhere:
this path is integer arithmetic.  Needs a real MI355X: run with `-m gpu`."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Q = 725501752471715841 | 6461107452199829505 << 64 | 6968279316240510977 << 128 | 1345280370688173398 << 192
R_ORDER = (13356249993388743167 | 5950279507993463550 << 64 | 10965441865914903552 << 128
           | 336320092672043349 << 192)


@pytest.fixture(scope="module")
def ctx():
    import decaf377_amd as d
    c = d.Context([0])
    yield c
    c.close()


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def hx(rows):
    return [bytes(r).hex() for r in np.asarray(rows, dtype=np.uint8).reshape(-1, 32)]


def frombytes(lst):
    return np.array([list(bytes.fromhex(h)) if isinstance(h, str) else list(h) for h in lst], dtype=np.uint8)


def ibytes(v):
    return np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)


def test_native_library_is_loaded(ctx):
    """The .so the driver looks for must actually be mapped into this process."""
    with open("/proc/self/maps") as f:
        assert "libdecaf377_amd.so" in f.read()


# --- reference KATs through the GPU ----------------------------------------------------------
def test_basepoint_multiples(ctx, kats):
    """tests/encoding.rs:55-95 of the reference."""
    hexes = kats["basepoint_multiples"]["hex"]
    enc = frombytes(hexes)
    xyzt, st = ctx.decompress(enc)
    assert not st.any()
    assert hx(ctx.compress(xyzt)) == hexes
    out, st = ctx.roundtrip(enc)
    assert not st.any() and hx(out) == hexes
    ks = np.zeros((16, 32), np.uint8)
    ks[:, 0] = np.arange(16)
    assert hx(ctx.scalar_mul_base(ks)) == hexes
    g = np.tile(frombytes([kats["generator"]["hex"]]), (16, 1))
    out, st = ctx.scalar_mul_var(g, ks)
    assert not st.any() and hx(out) == hexes


def test_identity_and_generator(ctx, kats):
    """tests/encoding.rs:20-52."""
    xyzt, st = ctx.decompress(frombytes([kats["identity"]["hex"], kats["generator"]["hex"]]))
    assert not st.any()
    one = [0x7d1c7ffffffffff3, 0x7257f50f6ffffff2, 0x16d81575512c0fee, 0x0d4bda322bbb9a9d]   # Fq::ONE
    assert [int(v) for v in xyzt[0]] == [0] * 4 + one + one + [0] * 4
    assert [int(v) for v in xyzt[1, 0:4]] == kats["generator"]["x_mont"]
    assert [int(v) for v in xyzt[1, 4:8]] == kats["generator"]["y_mont"]
    assert [int(v) for v in xyzt[1, 8:12]] == one
    assert [int(v) for v in xyzt[1, 12:16]] == kats["generator"]["t_mont"]
    firsts = np.zeros((255, 32), np.uint8)
    firsts[:, 0] = np.arange(1, 256)
    _, st = ctx.decompress(firsts)
    assert int(np.nonzero(st == 0)[0][0]) + 1 == kats["generator"]["min_first_byte"]


def test_elligator_kats(ctx, kats, oracle):
    """src/ark_curve/elligator.rs:86-207 -- the 8 inputs; encodings equal the oracle's, whose
    affine (x, y) are pinned to the reference's decimals in tests/test_oracle.py."""
    inputs = np.array(kats["elligator"]["inputs"], dtype=np.uint8)
    enc = ctx.encode_to_curve(inputs)
    assert (enc == oracle.encode_to_curve(inputs)).all()
    xyzt, st = ctx.decompress(enc)
    assert not st.any()
    assert oracle.eq_xyzt(xyzt, oracle.elligator_map_xyzt(inputs)).all()


def test_sqrt_edge_cases(ctx, kats):
    """src/ark_curve/invsqrt.rs:204-211 and proptest-regressions/invsqrt.txt:7."""
    num = np.stack([ibytes(0), ibytes(1), ibytes(0), ibytes(1 << 248), ibytes(Q), ibytes(1)])
    den = np.stack([ibytes(1), ibytes(0), ibytes(0), ibytes(1 << 248), ibytes(5), ibytes(Q)])
    root, ws = ctx.sqrt_ratio_zeta(num, den)
    assert list(ws[:3]) == [1, 0, 1] and not root[:3].any()
    r = int.from_bytes(bytes(root[3]), "little")
    assert ws[3] == 1 and r * r % Q == 1
    assert ws[4] == 1 and not root[4].any()      # num = q = 0 mod q
    assert ws[5] == 0 and not root[5].any()      # den = q = 0 mod q


def test_regression_seeds(ctx, kats):
    seeds = np.array(kats["regression_seeds"]["encoding_bytes"], dtype=np.uint8)
    out, st = ctx.roundtrip(seeds)
    assert list(st) == [0, 0, 1]
    assert (out[:2] == seeds[:2]).all() and not out[2].any()


def test_groth16_gadget_regression_inputs_gpu(ctx, oracle, kats):
    """tests/groth16_gadgets.proptest-regressions:7-15: the reference's shrunk gadget inputs (4 Element encodings, 3 Fr,
    2 Fq, one scalar byte array) through every hot-path operation they are values for, against the oracle (which
    tests/test_oracle.py pins on the same inputs against the big-integer model): decompress -> compress, each point x each
    scalar (Encoding and Element forms), GENERATOR x each scalar, the pair's sum, encode_to_curve / hash_to_curve and
    sqrt_ratio_zeta of the field elements, and the 16 products as one multiscalar sum."""
    from _kat_inputs import groth16_regression_inputs
    g = groth16_regression_inputs(kats)
    pts, ks, fq = g["points"], g["scalars"], g["fq"]
    out, st = ctx.roundtrip(pts)
    assert not st.any() and (out == pts).all()
    xyzt, st = ctx.decompress(pts)
    assert not st.any() and (xyzt == oracle.decompress(pts)[0]).all()
    assert (ctx.compress(xyzt) == pts).all()
    P_ = np.repeat(pts, 4, axis=0)
    K_ = np.tile(ks, (4, 1))
    o_out, o_st = oracle.scalar_mul_var(P_, K_)
    out, st = ctx.scalar_mul_var(P_, K_)
    assert (st == o_st).all() and (out == o_out).all()
    X_ = np.repeat(xyzt, 4, axis=0)
    assert (ctx.compress(ctx.scalar_mul_var_element(X_, K_)) == o_out).all()
    assert (ctx.scalar_mul_base(ks) == oracle.scalar_mul_base(ks)).all()
    assert (ctx.compress(ctx.add(xyzt[2:3], xyzt[3:4])) == oracle.compress(oracle.add_xyzt(xyzt[2:3], xyzt[3:4]))).all()
    assert (ctx.encode_to_curve(fq) == oracle.encode_to_curve(fq)).all()
    assert (ctx.hash_to_curve(fq[0:1], fq[1:2]) == oracle.hash_to_curve(fq[0:1], fq[1:2])).all()
    one = ibytes(1)
    num = np.stack([fq[0], fq[1], fq[0], fq[1], one, one])
    den = np.stack([fq[1], fq[0], one, one, fq[0], fq[1]])
    for conv, orc_fn in (("ark", oracle.sqrt_ratio_zeta), ("min_curve", oracle.sqrt_ratio_zeta_min_curve)):
        root, ws = ctx.sqrt_ratio_zeta(num, den, root=conv)
        o_root, o_ws = orc_fn(num, den)
        assert (ws == o_ws).all() and (root == o_root).all(), conv
    # sum_i k_i P_i over the 16 pairs, both input forms, against the oracle's fold of its own products
    acc = oracle.decompress(o_out[0:1])[0]
    for i in range(1, 16):
        acc = oracle.add_xyzt(acc, oracle.decompress(o_out[i:i + 1])[0])
    want = oracle.compress(acc)[0]
    assert (np.asarray(ctx.msm(P_, K_)[0]).reshape(32) == want).all()
    assert (np.asarray(ctx.msm(X_, K_)[0]).reshape(32) == want).all()


def test_lazy_comb_context_and_comb_widths(oracle, torch_mod):
    """d377_ctx_create_ex (include/decaf377_amd.h): in the reference Element::GENERATOR is a constant
    (src/min_curve/element.rs:61-81) and costs nothing until it is used.  A context created with comb_lazy holds under
    1 GB of HBM (hipMemGetInfo before and after); operations that need no comb run on it; its first fixed-base call builds
    the 5.9 GB table and is bit-exact; the 18- and 21-bit combs give the same bytes through every fixed-base route
    (wave, quad, lane: narrow and wide launch; Encoding and Element forms)."""
    import decaf377_amd as d
    torch = torch_mod
    rng = np.random.default_rng(606)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    c = d.Context([0], comb_lazy=True)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 1_000_000_000, free0 - free1
    bits, built, nbytes = c.comb_info()
    assert bits == 23 and not built and nbytes == 11 * (2**22 + 1) * 128
    r0 = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (300, 32), dtype=np.uint8)
    enc = c.encode_to_curve(r0)
    out, st = c.scalar_mul_var(enc, k)
    o_out, o_st = oracle.scalar_mul_var(oracle.encode_to_curve(r0), k)
    assert (out == o_out).all() and (st == o_st).all()
    assert not c.comb_info()[1]                                  # still no table
    want = oracle.scalar_mul_base(k)
    assert (c.scalar_mul_base(k) == want).all()                  # builds it
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info()
    assert c.comb_info()[1] and free1 - free2 >= nbytes * 9 // 10
    kd = torch.from_numpy(k).cuda()
    assert (c.scalar_mul_base(kd).cpu().numpy() == want).all()    # the `_dev` route on the built table
    c.close()
    torch.cuda.synchronize()
    # the narrower combs, eager and lazy, every route: 100 (a wave per scalar), 5 000 (a quad), 70 000 (lanes, narrow launch),
    # 1.25 x 2^20 (lanes, wide launch: sample against the oracle, the rest against the default context)
    sizes = (100, 5000, 70000)
    ks = rng.integers(0, 256, (70000, 32), dtype=np.uint8)
    ks[0] = 0
    ks[1] = ibytes(R_ORDER - 1)
    ks[2] = ibytes(R_ORDER)
    ks[3] = 255
    want = oracle.run_threads("scalar_mul_base", ks, None, len(os.sched_getaffinity(0)))[0]
    big = torch.randint(0, 256, (5 * 2**18, 32), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
    ref = None
    for bits, lazy in ((23, False), (18, False), (21, True)):
        c = d.Context([0], comb_bits=bits, comb_lazy=lazy)
        assert c.comb_info()[:2] == (bits, not lazy)
        for n in sizes:
            assert (c.scalar_mul_base(ks[:n]) == want[:n]).all(), (bits, n)
            assert (c.compress(c.scalar_mul_base_element(ks[:n])) == want[:n]).all(), (bits, n)
        assert c.comb_info()[1]
        got = c.scalar_mul_base(big)
        if ref is None:
            ref = got
            assert (got[:4096].cpu().numpy() == oracle.scalar_mul_base(big[:4096].cpu().numpy())).all()
        else:
            assert bool((got == ref).all()), bits
        c.close()
        torch.cuda.synchronize()


# --- committed model vectors -------------------------------------------------------------------
def test_vectors(ctx, vectors):
    v = vectors["sqrt_ratio_zeta"]
    root, ws = ctx.sqrt_ratio_zeta(frombytes([c["num"] for c in v]), frombytes([c["den"] for c in v]))
    assert hx(root) == [c["root"] for c in v] and list(ws) == [c["was_square"] for c in v]
    v = vectors["encode_to_curve"]
    assert hx(ctx.encode_to_curve(frombytes([c["r0"] for c in v]))) == [c["enc"] for c in v]
    v = vectors["decompress"]
    enc = frombytes([c["enc"] for c in v])
    xyzt, st = ctx.decompress(enc)
    out, st2 = ctx.roundtrip(enc)
    assert list(st) == [c["status"] for c in v] == list(st2)
    for i, c in enumerate(v):
        if c["status"] == 0:
            assert [[int(x) for x in xyzt[i, 4 * j:4 * j + 4]] for j in range(4)] == c["xyzt_mont"]
            assert bytes(out[i]).hex() == c["enc"]
        else:
            assert not xyzt[i].any() and not out[i].any()
    v = vectors["scalar_mul_base"]
    assert hx(ctx.scalar_mul_base(frombytes([c["scalar"] for c in v]))) == [c["enc"] for c in v]
    v = vectors["scalar_mul_var"]
    out, st = ctx.scalar_mul_var(frombytes([c["point"] for c in v]), frombytes([c["scalar"] for c in v]))
    assert hx(out) == [c["enc"] for c in v] and list(st) == [c["status"] for c in v]
    v = vectors["hash_to_curve"]
    assert hx(ctx.hash_to_curve(frombytes([c["r1"] for c in v]), frombytes([c["r2"] for c in v]))) == \
        [c["enc"] for c in v]


# --- seeded random parity against the oracle ---------------------------------------------------
def test_sqrt_random_2_16(ctx, oracle):
    """BASELINE config 1 size (2^16 pairs), edge pairs first; every (flag, root) vs the oracle."""
    rng = np.random.default_rng(666)
    n = 1 << 16
    num = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    den = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for i, (u, v) in enumerate([(0, 1), (1, 0), (0, 0), (1, 1), (1 << 248, 1 << 248)]):
        num[i], den[i] = ibytes(u), ibytes(v)
    root, ws = ctx.sqrt_ratio_zeta(num, den)
    o_root, o_ws, _ = oracle.run_threads("sqrt_ratio_zeta", num, den, 8)
    assert (ws == o_ws).all() and (root == o_root).all()


def test_raw_root_pinned_by_tonelli_shanks_zeta_seed(ctx, oracle):
    """Raw sqrt_ratio_zeta root VALUES against a definition that shares nothing with the Sarkar text the kernel and the
    oracle both follow: Tonelli-Shanks seeded with zeta^m on num/den (oracle/d377_model.py sqrt_ratio_zeta_ts_zeta;
    src/min_curve/constants.rs:10-15, loop of src/min_curve/invsqrt.rs:36-54).  2^12 seeded pairs, edge pairs first;
    the D377_SQRT_ROOT_ARK output must equal it bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import d377_model as m
    from test_oracle import _root_pin_pairs
    n = 1 << 12
    num, den = _root_pin_pairs(n)
    root, ws = ctx.sqrt_ratio_zeta(num, den, root="ark")          # d377_batch_sqrt_ratio_zeta_ex, D377_SQRT_ROOT_ARK
    for i in range(n):
        fl, r = m.sqrt_ratio_zeta_ts_zeta(m.fq_from_le_bytes_mod_order(bytes(num[i])), m.fq_from_le_bytes_mod_order(bytes(den[i])))
        assert (bool(ws[i]), int.from_bytes(bytes(root[i]), "little")) == (fl, r), i


def test_roundtrip_random_raw(ctx, oracle):
    """2^16 raw strings (mostly invalid) + valid encodings: status and zero-output equal the oracle."""
    rng = np.random.default_rng(667)
    n = 1 << 16
    raw = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    raw[: n // 2, 31] &= 0x1F
    out, st = ctx.roundtrip(raw)
    o_out, o_st, _ = oracle.run_threads("roundtrip", raw, None, 8)
    assert (st == o_st).all() and (out == o_out).all()
    assert 1000 < (st == 0).sum() < n - 1000
    xyzt, st2 = ctx.decompress(raw[:4096])
    o_xyzt, o_st2 = oracle.decompress(raw[:4096])
    assert (st2 == o_st2).all() and (xyzt == o_xyzt).all()
    ok = st2 == 0
    assert (ctx.compress(xyzt[ok]) == raw[:4096][ok]).all()


def test_encode_and_scalar_mul_random(ctx, oracle):
    rng = np.random.default_rng(668)
    n = 1 << 13
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for i, v in enumerate([0, 1, R_ORDER - 1, R_ORDER, R_ORDER + 1, (1 << 256) - 1]):
        k[i] = ibytes(v)
    enc = ctx.encode_to_curve(r0)
    o_enc, _, _ = oracle.run_threads("encode_to_curve", r0, None, 8)
    assert (enc == o_enc).all()
    out, st = ctx.scalar_mul_var(enc, k)
    o_out, o_st, _ = oracle.run_threads("scalar_mul_var", enc, k, 8)
    assert (st == o_st).all() and (out == o_out).all()
    fb = ctx.scalar_mul_base(k)
    o_fb, _, _ = oracle.run_threads("scalar_mul_base", k, None, 8)
    assert (fb == o_fb).all()
    h = ctx.hash_to_curve(r0[: n // 2], r0[n // 2:])
    assert (h[:512] == oracle.hash_to_curve(r0[:512], r0[n // 2: n // 2 + 512])).all()


def test_ragged_and_empty(ctx, oracle):
    """n = 0, 1, 63, 64, 65, 257 (partial waves and blocks)."""
    rng = np.random.default_rng(669)
    for n in (0, 1, 63, 64, 65, 257):
        r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        enc = ctx.encode_to_curve(r0)
        assert enc.shape == (n, 32)
        if n:
            assert (enc == oracle.encode_to_curve(r0)).all()
            out, st = ctx.scalar_mul_var(enc, k)
            o_out, o_st = oracle.scalar_mul_var(enc, k)
            assert (out == o_out).all() and (st == o_st).all()
            rt, st = ctx.roundtrip(enc)
            assert (rt == enc).all() and not st.any()


# --- device-pointer path and full-size properties ------------------------------------------------
def test_device_path_matches_host_path(ctx, torch_mod):
    torch = torch_mod
    rng = np.random.default_rng(670)
    n = 5000
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    dev = torch.device("cuda:0")
    enc_d = ctx.encode_to_curve(torch.from_numpy(r0).to(dev))
    out_d, st_d = ctx.scalar_mul_var(enc_d, torch.from_numpy(k).to(dev))
    xyzt_d, st2_d = ctx.decompress(enc_d)
    cmp_d = ctx.compress(xyzt_d)
    torch.cuda.synchronize()
    enc_h = ctx.encode_to_curve(r0)
    out_h, st_h = ctx.scalar_mul_var(enc_h, k)
    assert (enc_d.cpu().numpy() == enc_h).all()
    assert (out_d.cpu().numpy() == out_h).all() and (st_d.cpu().numpy() == st_h).all()
    assert (cmp_d.cpu().numpy() == enc_h).all() and not st2_d.cpu().numpy().any()


def test_full_size_roundtrip_2_20(ctx, torch_mod, oracle):
    """BASELINE config 2: 2^20 valid encodings decompress->compress to themselves (status 0);
    a seeded sample is also compared with the oracle."""
    torch = torch_mod
    n = 1 << 20
    g = torch.Generator(device="cpu").manual_seed(666)
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
    dev = torch.device("cuda:0")
    enc = ctx.encode_to_curve(r0.to(dev))
    out, st = ctx.roundtrip(enc)
    torch.cuda.synchronize()
    assert torch.equal(out, enc) and int(st.sum().item()) == 0
    idx = np.arange(0, n, 4099)
    assert (enc[idx].cpu().numpy() == oracle.encode_to_curve(r0.numpy()[idx])).all()


def test_full_size_scalar_mul_algebra(ctx, torch_mod, oracle):
    """tests/operations.rs:19-43 at 2^18 elements: aP + bP = (a+b)P and b(aP) = (ab)P as encodings,
    using only GPU operations plus host big-int scalar arithmetic; then a sample vs the oracle."""
    torch = torch_mod
    n = 1 << 18
    rng = np.random.default_rng(671)
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    P = ctx.encode_to_curve(r0)
    aP, st = ctx.scalar_mul_var(P, a)
    assert not st.any()
    baP, st = ctx.scalar_mul_var(aP, b)
    # (ab)P on a sample (host big-int products are slow in pure Python for 2^18)
    m = 2048
    ai = [int.from_bytes(bytes(x), "little") % R_ORDER for x in a[:m]]
    bi = [int.from_bytes(bytes(x), "little") % R_ORDER for x in b[:m]]
    ab = np.array([list((x * y % R_ORDER).to_bytes(32, "little")) for x, y in zip(ai, bi)], dtype=np.uint8)
    apb = np.array([list(((x + y) % R_ORDER).to_bytes(32, "little")) for x, y in zip(ai, bi)], dtype=np.uint8)
    abP, _ = ctx.scalar_mul_var(P[:m], ab)
    assert (abP == baP[:m]).all()
    # aP + bP == (a+b)P: add through the oracle's group law on decompressed points
    bP, _ = ctx.scalar_mul_var(P[:m], b[:m])
    x1, _ = ctx.decompress(aP[:m])
    x2, _ = ctx.decompress(bP)
    s = ctx.compress(oracle.add_xyzt(x1, x2))
    apbP, _ = ctx.scalar_mul_var(P[:m], apb)
    assert (s == apbP).all()
    idx = np.arange(0, n, 1031)
    o_out, _ = oracle.scalar_mul_var(P[idx], a[idx])
    assert (aP[idx] == o_out).all()


def test_bad_arguments(ctx):
    import decaf377_amd as d
    with pytest.raises(d.NativeError):
        d.Context([99])


def test_add_double_eq(ctx, oracle, kats):
    """Element + Element, double, == through the GPU: the reference's running-sum check
    (tests/encoding.rs:79-94: accumulator += basepoint; assert_eq!(accumulator, point)),
    and bit-exact extended coordinates vs the oracle's restatement of the same formulas."""
    hexes = kats["basepoint_multiples"]["hex"]
    pts, st = ctx.decompress(frombytes(hexes))
    gen = pts[1:2]
    acc = pts[0:1]
    for i in range(16):
        assert ctx.eq(acc, pts[i:i + 1])[0] == 1
        assert hx(ctx.compress(acc)) == [hexes[i]]
        acc = ctx.add(acc, gen)
    rng = np.random.default_rng(672)
    n = 3000
    P = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    Qp = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    s = ctx.add(P, Qp)
    assert (s == oracle.add_xyzt(P, Qp)).all()
    d2 = ctx.double(P)
    assert (d2 == oracle.double_xyzt(P)).all()
    assert ctx.eq(d2, ctx.add(P, P)).all()
    assert not ctx.eq(P, Qp).any()
    assert (ctx.eq(s, ctx.add(Qp, P)) == 1).all()
    # Element - Element = self + other.neg() (src/min_curve/ops.rs:43-87): the reference's coordinates, and (P + Q) - Q = P
    df = ctx.sub(P, Qp)
    assert (df == oracle.sub_xyzt(P, Qp)).all()
    assert ctx.eq(ctx.sub(s, Qp), P).all() and ctx.is_identity(ctx.sub(P, P)).all()
    ident = np.tile(oracle.identity_xyzt(), (4, 1))
    assert (ctx.sub(ident, P[:4]) == oracle.sub_xyzt(ident, P[:4])).all() and ctx.eq(ctx.sub(P[:4], ident), P[:4]).all()


def test_full_size_fixed_base_2_20(ctx, torch_mod, oracle):
    """BASELINE config 3: 2^20 fixed-base mults.  Full-size check: the comb kernel must agree with
    the variable-base kernel run on the generator's encoding (two independent code paths); a
    seeded sample and the edge scalars 0, 1, r-1 are compared with the oracle."""
    torch = torch_mod
    n = 1 << 20
    g = torch.Generator(device="cpu").manual_seed(673)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
    for i, v in enumerate([0, 1, R_ORDER - 1]):
        k[i] = torch.from_numpy(ibytes(v).copy())
    dev = torch.device("cuda:0")
    kd = k.to(dev)
    fb = ctx.scalar_mul_base(kd)
    gen = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    gen[:, 0] = 8
    vb, st = ctx.scalar_mul_var(gen, kd)
    torch.cuda.synchronize()
    assert torch.equal(fb, vb) and int(st.sum().item()) == 0
    idx = np.concatenate([np.arange(3), np.arange(3, n, 8191)])
    assert (fb[idx].cpu().numpy() == oracle.scalar_mul_base(k.numpy()[idx])).all()


def test_fixed_base_launch_shapes_agree(ctx, torch_mod, oracle):
    """The fixed-base kernel is launched narrow (2 workgroups per CU, 8 elements per shared inversion) below 2^21
    elements and wide (3 per CU, 16 per inversion) from there.  Every shape -- forced through the developer overrides at
    a ragged size, and the default on both sides of the threshold -- gives the same bytes, a sample of them the oracle's,
    and the wide launch (where the third lane set of the scratch areas is used) also in place."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(2291)
    n = 16 * 196608 // 5 + 77
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    want = ctx.scalar_mul_base(k)
    idx = np.unique(np.concatenate([np.arange(40), np.arange(n - 300, n), np.arange(17, n, n // 61)]))
    assert (want[torch.from_numpy(idx).to(dev)].cpu().numpy() == oracle.scalar_mul_base(k[torch.from_numpy(idx).to(dev)].cpu().numpy())).all()
    for sets, kk in ((3, 16), (3, 8), (2, 16), (3, 5), (2, 1)):
        with ctx.tuning(fb_wide=int(sets == 3), fb_k=kk):
            assert torch.equal(ctx.scalar_mul_base(k), want), (sets, kk)
            k2 = k.clone()
            ctx.scalar_mul_base(k2, outs=[k2])
            assert torch.equal(k2, want), (sets, kk)
    assert ctx.get_tuning("fb_wide") is None and ctx.get_tuning("fb_k") is None
    # the default above the threshold against the narrow launch forced at the same size
    n = (3 << 20) + 4099
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    wide = ctx.scalar_mul_base(k)
    with ctx.tuning(fb_wide=0, fb_k=8):
        assert torch.equal(ctx.scalar_mul_base(k), wide)
    ti = torch.from_numpy(np.arange(5, n, n // 53)).to(dev)
    assert (wide[ti].cpu().numpy() == oracle.scalar_mul_base(k[ti].cpu().numpy())).all()


def test_batched_inverse_decoding_from_2_21(ctx, torch_mod, oracle):
    """From 2^21 elements `decompress` and the decoding pass of the Encoding-input MSM run in chunks with the square roots'
    denominators inverted together (k_decompress_chunked, k_msm_prepare_enc_chunked).  Same bytes and statuses as the
    one-lane-per-element kernels forced at the same size (developer overrides), invalid encodings of every kind sprinkled
    in, a sample equal to the oracle's; ragged size, so that the last chunk is partial."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(9107)
    n = (1 << 21) + 4321
    enc = ctx.encode_to_curve(torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g))
    enc[7::1013, 31] |= 0x40                                      # top bits set
    enc[11::2003, 0] |= 1                                         # negative s
    enc[13::4001] = torch.randint(0, 256, enc[13::4001].shape, dtype=torch.uint8, device=dev, generator=g)   # raw strings
    enc[17::5003] = 0                                             # the identity
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    P, st = ctx.decompress(enc)
    m = ctx.msm(enc, k)
    rt, rst = ctx.roundtrip(enc)
    ce = ctx.compress(P)
    with ctx.tuning(decompress_chunked_min=1 << 40, msm_enc_chunked_min=1 << 40):
        P0, st0 = ctx.decompress(enc)
        m0 = ctx.msm(enc, k)
        rt0, rst0 = ctx.roundtrip(enc)
        ce0 = ctx.compress(P)
    assert torch.equal(P, P0) and torch.equal(st, st0) and 0 < int(st.sum().item()) < n // 100
    # compress and the round trip (two batched inversions per chunk) in chunks against the wide grid; in place too
    assert torch.equal(rt, rt0) and torch.equal(rst, rst0) and torch.equal(rst, st) and torch.equal(ce, ce0)
    assert torch.equal(rt[st == 0], enc[st == 0]) and not rt[st != 0].any()
    e2 = enc.clone()
    ctx.roundtrip(e2, outs=[e2, rst0])
    assert torch.equal(e2, rt)
    assert bytes(m[0]) == bytes(m0[0]) and torch.equal(torch.as_tensor(m[2]), torch.as_tensor(m0[2])) and torch.equal(torch.as_tensor(m[2]), st)
    idx = np.unique(np.concatenate([np.arange(40), np.arange(7, n, 1013)[:20], np.arange(13, n, 4001)[:20], np.arange(n - 40, n)]))
    ti = torch.from_numpy(idx).to(dev)
    o_P, o_st = oracle.decompress(enc[ti].cpu().numpy())
    assert (P[ti].cpu().numpy().view(np.uint64) == o_P).all() and (st[ti].cpu().numpy() == o_st).all()
    o_rt, o_rst = oracle.roundtrip(enc[ti].cpu().numpy())
    assert (rt[ti].cpu().numpy() == o_rt).all() and (rst[ti].cpu().numpy() == o_rst).all()
    ok_i = ti[st[ti] == 0]
    assert (ce[ok_i].cpu().numpy() == oracle.compress(P[ok_i].cpu().numpy().view(np.uint64))).all()
    # the sum itself: against the MSM of the decoded Elements with the invalid ones left out (zero scalars)
    k2 = k.clone()
    k2[st != 0] = 0
    Pz = P.clone()
    ident = torch.from_numpy(oracle.identity_xyzt().view(np.int64)).to(dev)
    Pz[st != 0] = ident
    assert bytes(ctx.msm(Pz, k2)[0]) == bytes(m[0])
    # the chunked forms start at 2 (decompress, compress, round trip: d377.hip CODEC_CHUNKED_MIN) and 3 (the MSM's decoding pass:
    # DCB_ASSIST_MIN) elements per resident lane (2 workgroups x 256 lanes per CU): ragged sizes from there up, where the
    # rounds are dealt out unevenly (DcbScratch::extra), against the wide grid
    lanes = torch.cuda.get_device_properties(0).multi_processor_count * 2 * 256
    for n2 in (2 * lanes, 2 * lanes + 256 * 3 + 5, 3 * lanes, 3 * lanes + 77, 5 * lanes - 3, 8 * lanes + 256 * 5 + 1, 9 * lanes + 13):
        P, st = ctx.decompress(enc[:n2])
        m = ctx.msm(enc[:n2], k[:n2])
        rt, rst = ctx.roundtrip(enc[:n2])
        ce = ctx.compress(P)
        with ctx.tuning(decompress_chunked_min=1 << 40, msm_enc_chunked_min=1 << 40):
            P0, st0 = ctx.decompress(enc[:n2])
            m0 = ctx.msm(enc[:n2], k[:n2])
            rt0, rst0 = ctx.roundtrip(enc[:n2])
            ce0 = ctx.compress(P)
        assert torch.equal(P, P0) and torch.equal(st, st0), n2
        assert torch.equal(rt, rt0) and torch.equal(rst, rst0) and torch.equal(ce, ce0), n2
        assert bytes(m[0]) == bytes(m0[0]) and torch.equal(torch.as_tensor(m[2]), st), n2


def test_full_size_hash_to_curve_two_routes_2_20(ctx, torch_mod, oracle):
    """hash_to_curve at BASELINE size by two routes that share no formula after the maps: the kernel's (the two points
    added on the Jacobi quartic, encoded without a square root) against the reference's own statement
    (src/ark_curve/elligator.rs:67-71) spelt out with the Element entry points -- encode_to_curve_element twice, Edwards
    addition, generic compression with its square root.  2^20 pairs, every byte; a sample against the oracle; and the
    Element-returning form."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    n = 1 << 20
    g = torch.Generator(device=dev).manual_seed(9311)
    r1 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    r2 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    r2[:1000] = r1[:1000]                                # doublings on the quartic
    r1[1000:1100] = 0                                    # the map of 0
    r2[1100:1200] = 0
    r1[1200:1300] = 0; r2[1200:1300] = 0
    h = ctx.hash_to_curve(r1, r2)
    want = ctx.compress(ctx.add(ctx.encode_to_curve_element(r1), ctx.encode_to_curve_element(r2)))
    assert torch.equal(h, want)
    assert torch.equal(ctx.compress(ctx.hash_to_curve_element(r1, r2)), want)
    idx = np.unique(np.concatenate([np.arange(0, 1400, 7), np.arange(17, n, n // 301)]))
    ti = torch.from_numpy(idx).to(dev)
    assert (h[ti].cpu().numpy() == oracle.hash_to_curve(r1[ti].cpu().numpy(), r2[ti].cpu().numpy())).all()


def test_batched_compressor_rounds_and_lane_tails(ctx, torch_mod, oracle):
    """The square-root-free compressor (curve.hpp `dcb_finish`; reference: src/ark_curve/encoding.rs:91-128) and the
    batched inversions work in chunks: one workgroup takes per_lane x 256 consecutive elements, per_lane = ceil(n /
    resident lanes) capped at 16 (dcb.hpp DCB_K_LONG; d377.hip `launch`; resident lanes = 2 workgroups x 256 lanes per CU
    = 131 072 on 256 CUs), and a lane inverts once for the per_lane elements it holds.  Sizes around every per_lane transition
    (k x 131 072 +- 1 for k = 1..8, 12, 16, 17: the last workgroup's lanes then hold per_lane, per_lane - 1 or 0 elements), sizes
    that leave whole waves of the last chunk empty, and sizes several grid generations long must give the oracle's
    bytes on the first and last records and a spread in between -- for the variable-base, the fixed-base and the
    Elligator kernels, with invalid encodings and identity results sprinkled in, and in place."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    lanes = torch.cuda.get_device_properties(0).multi_processor_count * 2 * 256
    g = torch.Generator(device=dev).manual_seed(9077)
    sizes = [lanes - 5, lanes + 7, 31 * lanes + 3, 32 * lanes, 32 * lanes + 4099]
    for kk in list(range(1, 9)) + [12, 16, 17]:
        sizes += [kk * lanes - 1, kk * lanes, kk * lanes + 1]
    sizes += [8 * lanes + 256 * 8 + 1, 255, 257, 64, 1]
    for n in sorted(set(sizes)):
        r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        enc = ctx.encode_to_curve(r0)
        idx = np.unique(np.concatenate([np.arange(min(40, n)), np.arange(max(0, n - 300), n), np.arange(17, n, max(1, n // 97))]))
        assert idx.min() >= 0 and idx.max() < n
        ti = torch.from_numpy(idx).to(dev)
        assert (enc[ti].cpu().numpy() == oracle.encode_to_curve(r0[ti].cpu().numpy())).all(), n
        raw = enc.clone()
        raw[ti[::7]] = 0xFF                             # invalid encodings (top bits set) inside the checked set
        k[ti[3::11]] = 0                                # [0]P = identity: the all-zero encoding
        out, st = ctx.scalar_mul_var(raw, k)
        o_out, o_st = oracle.scalar_mul_var(raw[ti].cpu().numpy(), k[ti].cpu().numpy())
        assert (out[ti].cpu().numpy() == o_out).all() and (st[ti].cpu().numpy() == o_st).all(), n
        assert n < 64 or (o_st.any() and not o_st.all())
        fb = ctx.scalar_mul_base(k)
        assert (fb[ti].cpu().numpy() == oracle.scalar_mul_base(k[ti].cpu().numpy())).all(), n
        # in place: the output records double as the parking places of the prefix products
        k2 = k.clone()
        ctx.scalar_mul_base(k2, outs=[k2])
        assert torch.equal(k2, fb), n


def test_ragged_chunks_match_uniform_chunks(ctx, torch_mod, oracle):
    """Batches of up to DCB_K rounds per resident workgroup run in one generation with the rounds dealt out evenly: the
    first `rounds % places` workgroups take one round more than the others (d377.hip `chunks_of`, DcbScratch::extra);
    larger ones in the fewest generations that hold them, dealt out the same way,
    and a workgroup decides by its own count whether it shares an inversion between its square roots.  Every chunked
    operation gives the same bytes as with uniform chunks of 1, 2, 3 and 8 elements per lane forced through the tuning
    call, at sizes that leave 1 .. places - 1 workgroups with the extra round (and a partial last round), and a
    sample of them is the oracle's."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    places = torch.cuda.get_device_properties(0).multi_processor_count * 2
    g = torch.Generator(device=dev).manual_seed(515)
    # ... and beyond 8 rounds per place chunks of up to 16 per lane (dcb.hpp DCB_K_LONG), dealt the same way over the fewest
    # generations that hold them (host_state.hpp deal_chunks): 10 rounds per place and a few (one generation of 10, three
    # chunks of 11), 17 and a third (two generations of 8 and 9)
    sizes = [places * 256 + 1, places * 256 + 256 * 97 - 13, 2 * places * 256 + 255, 3 * places * 256 - 256 - 1,
             (7 * places + 1) * 256 + 5, (3 * places + places // 2) * 256,
             (10 * places + 3) * 256 + 7, (17 * places + places // 3) * 256 - 1]
    for n in sizes:
        r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        enc = ctx.encode_to_curve(r0)
        enc[5::1009, 31] |= 0x80
        want = {
            "sqrt": ctx.sqrt_ratio_zeta(r0, k),
            "encode": (enc,),
            "hash": (ctx.hash_to_curve(r0, k),),
            "var": ctx.scalar_mul_var(enc, k),
            "base": (ctx.scalar_mul_base(k),),
            "encode_el": (ctx.encode_to_curve_element(r0),),
        }
        for per_lane in (1, 2, 3, 8):
            with ctx.tuning(chunk_per_lane=per_lane):
                got = {
                    "sqrt": ctx.sqrt_ratio_zeta(r0, k),
                    "encode": (ctx.encode_to_curve(r0),),
                    "hash": (ctx.hash_to_curve(r0, k),),
                    "var": ctx.scalar_mul_var(enc, k),
                    "base": (ctx.scalar_mul_base(k),),
                    "encode_el": (ctx.encode_to_curve_element(r0),),
                }
            got["encode"][0][5::1009, 31] |= 0x80
            for name in want:
                for a, b in zip(want[name], got[name]):
                    assert torch.equal(a, b), (name, n, per_lane)
        idx = np.unique(np.concatenate([np.arange(40), np.arange(n - 300, n), np.arange(17, n, n // 61)]))
        ti = torch.from_numpy(idx).to(dev)
        o_root, o_sq = oracle.sqrt_ratio_zeta(r0[ti].cpu().numpy(), k[ti].cpu().numpy())
        assert (want["sqrt"][0][ti].cpu().numpy() == o_root).all() and (want["sqrt"][1][ti].cpu().numpy() == o_sq).all(), n
        o_out, o_st = oracle.scalar_mul_var(enc[ti].cpu().numpy(), k[ti].cpu().numpy())
        assert (want["var"][0][ti].cpu().numpy() == o_out).all() and (want["var"][1][ti].cpu().numpy() == o_st).all(), n
        assert (want["hash"][0][ti].cpu().numpy() == oracle.hash_to_curve(r0[ti].cpu().numpy(), k[ti].cpu().numpy())).all(), n
        assert (want["base"][0][ti].cpu().numpy() == oracle.scalar_mul_base(k[ti].cpu().numpy())).all(), n


def test_workgroups_walk_several_chunks_beyond_the_grid_cap(ctx, torch_mod, oracle):
    """Beyond 64 chunks per CU (2^26 elements on 256 CUs) the grid stops growing and a workgroup walks several chunks of 16
    elements per lane, drawing a new ticket for each (dcb.hpp dcb_rounds; host_state.hpp deal_chunks): the largest shape a
    call can take.  Same bytes as the two halves as calls of their own (dealt out evenly, one chunk per workgroup), and a
    sample is the oracle's."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n = cus * 64 * 16 * 256 + 256 * 3 + 5                 # one chunk more than the cap, the last one ragged
    g = torch.Generator(device=dev).manual_seed(2525)
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    out = ctx.encode_to_curve(r0)
    h = n // 2 + 11
    assert torch.equal(out[:h], ctx.encode_to_curve(r0[:h])) and torch.equal(out[h:], ctx.encode_to_curve(r0[h:]))
    idx = np.unique(np.concatenate([np.arange(300), np.arange(n - 1500, n), np.arange(cus * 64 * 16 * 256 - 300, cus * 64 * 16 * 256 + 300),
                                    np.arange(31, n, n // 97)]))
    ti = torch.from_numpy(idx).to(dev)
    assert (out[ti].cpu().numpy() == oracle.encode_to_curve(r0[ti].cpu().numpy())).all()
    del r0, out
    torch.cuda.empty_cache()


def test_lane_set_pool_health_and_reset(torch_mod, oracle):
    """The lane-set pool has a way back (dcb.hpp, d377_ctx_health / d377_ctx_reset_scratch).  A context of its own: every
    set is marked as claimed by nobody (the debug hook: what a launch that died mid-kernel leaves behind), a chunked
    operation is enqueued and starves -- its workgroups count themselves as having waited long, the health call sees
    them without waiting for the kernel --, the reset frees exactly the leaked sets while the kernel is still waiting,
    the kernel then runs to completion with the oracle's bytes, and the pool is empty afterwards."""
    import time
    import decaf377_amd as d
    torch = torch_mod
    dev = torch.device("cuda:0")
    c = d.Context([0])
    try:
        sets_total = c.chunk_residency()[0] * torch.cuda.get_device_properties(0).multi_processor_count
        assert c.health() == (0, 0, 0) and c.reset_scratch() == 0
        g = torch.Generator(device=dev).manual_seed(77)
        n = 70000
        r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
        want = c.encode_to_curve(r0)
        torch.cuda.synchronize()
        c._debug_poison_pool()
        assert c.health()[0] == sets_total
        out = torch.zeros_like(want)
        c.encode_to_curve(r0, outs=[out])                      # enqueued; every workgroup waits for a set
        t0 = time.time()
        while c.health()[1] == 0:
            assert time.time() - t0 < 8.0, "no workgroup reported a long wait"
            time.sleep(0.05)
        h = c.health()
        assert h[2] == 0, (h, time.time() - t0)                # nobody has given up yet (10 s)
        freed = c.reset_scratch()                              # waits 1.5 s, sees the same tickets, frees them, waits for the kernel
        assert freed == sets_total, (freed, sets_total)
        torch.cuda.synchronize()
        assert torch.equal(out, want)
        claimed, waited, gave_up = c.health()
        assert claimed == 0 and waited >= 1 and gave_up == 0
        idx = np.arange(0, n, n // 50)
        assert (out[torch.from_numpy(idx).to(dev)].cpu().numpy() == oracle.encode_to_curve(r0[torch.from_numpy(idx).to(dev)].cpu().numpy())).all()
        # a partial leak: the operation still completes by itself (free sets remain), and the reset then finds the leaked ones
        c._debug_poison_pool(sets=37)
        assert torch.equal(c.encode_to_curve(r0), want)
        torch.cuda.synchronize()
        assert c.health()[0] == 37 and c.reset_scratch() == 37 and c.health()[0] == 0
        assert torch.equal(c.scalar_mul_base(r0), c.scalar_mul_base(r0.clone()))
    finally:
        c.close()


@pytest.mark.gpu
def test_starved_call_is_an_error_not_a_silent_partial_output(torch_mod, oracle):
    """A workgroup that finds no lane set for 10 s leaves without writing its records (dcb.hpp).  That must never read as
    success -- the reference's fallible operations always return a Result (src/ark_curve/encoding.rs:34-60): a
    HOST-POINTER call on a poisoned pool FAILS with D377_ERR_STARVED (StarvedError), the MSM of encodings likewise; a
    `_dev` caller sees the gave-up counter move through the word d377_ctx_starved_counter_dev hands out; after
    d377_ctx_reset_scratch the same calls give the oracle's bytes."""
    import time
    import decaf377_amd as d
    torch = torch_mod
    dev = torch.device("cuda:0")
    c = d.Context([0])
    try:
        rng = np.random.default_rng(9107)
        n = 70000                                       # above the quad kernels: the chunked route (claims lane sets), one generation of workgroups
        r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        want = c.encode_to_curve(r0)                    # host-pointer call on a healthy context
        word = c.starved_counter()
        assert word.dtype == torch.int32 and int(word.item()) == 0
        c._debug_poison_pool()
        t0 = time.time()
        with pytest.raises(d.StarvedError) as ei:
            c.encode_to_curve(r0)
        dt = time.time() - t0
        assert 9.0 < dt < 40.0, dt                      # every workgroup waited its 10 s (they wait side by side)
        assert "lane set" in str(ei.value)
        claimed, waited, gave_up = c.health()
        assert gave_up >= 1 and int(word.item()) == gave_up
        # the `_dev` form cannot fail at the call (it only enqueues): its caller watches the counter on the stream
        r0_d = torch.from_numpy(r0).to(dev)
        before = word.clone()
        out = c.encode_to_curve(r0_d)
        after = word.clone()                            # torch's current stream: ordered behind the kernel
        torch.cuda.synchronize()
        assert int(after.item()) > int(before.item())
        assert c.reset_scratch() == claimed
        assert (c.encode_to_curve(r0) == want).all()
        assert torch.equal(c.encode_to_curve(r0_d), torch.from_numpy(want).to(dev))
        idx = np.arange(0, n, n // 64)
        assert (want[idx] == oracle.encode_to_curve(r0[idx])).all()
        assert c.health()[0] == 0
    finally:
        c.close()


def test_small_batch_quad_kernel_matches_lane_kernel(ctx, oracle, torch_mod):
    """Batches that cannot fill the chip with one lane per element (up to 7 or 8 x 16 quads per CU) run one element per QUAD of
    lanes (d377.hip k_scalar_mul_var_small, quad_ops.hpp).  Same bytes as the one-lane-per-element kernel (forced with
    the small_max tuning key) at sizes around every edge of the quad kernel's grid -- invalid encodings, zero and extreme
    scalars included -- and as the oracle; likewise the Element form (records in, records out)."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4401)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    one_gen, el_max, enc_max = cus * 4 * 16, cus * 7 * 16, cus * 8 * 16       # one wave per SIMD; the two forms' thresholds
    if True:
        for n in (1, 3, 16, 17, 1000, 4 * cus, 4 * cus + 1, one_gen - 1, one_gen + 1, el_max, el_max + 1, enc_max, enc_max + 1):
            enc = oracle.encode_to_curve(rng.integers(0, 256, (min(n, 2048), 32), dtype=np.uint8))
            enc = np.tile(enc, ((n + enc.shape[0] - 1) // enc.shape[0], 1))[:n].copy()
            k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            enc[::13, 31] |= 0x80                            # invalid encodings
            for i, v in enumerate([0, 1, 2, R_ORDER - 1, R_ORDER, R_ORDER + 1, (1 << 256) - 1][:n]):
                k[i] = ibytes(v)
            te, tk = torch.from_numpy(enc).to(dev), torch.from_numpy(k).to(dev)
            with ctx.tuning(tiny_max=0):                      # quads
                out_q, st_q = ctx.scalar_mul_var(te, tk)
            with ctx.tuning(small_max=0):
                out_l, st_l = ctx.scalar_mul_var(te, tk)
            assert torch.equal(out_q, out_l) and torch.equal(st_q, st_l), n
            if n <= 4 * cus + 1:                              # one wave per element (lane-spread arithmetic), also one past its size
                with ctx.tuning(tiny_max=10**6):
                    out_w, st_w = ctx.scalar_mul_var(te, tk)
                assert torch.equal(out_w, out_l) and torch.equal(st_w, st_l), n
            out_d, st_d = ctx.scalar_mul_var(te, tk)          # whatever the size picks
            assert torch.equal(out_d, out_l) and torch.equal(st_d, st_l), n
            sel = np.unique(np.concatenate([np.arange(min(n, 24)), np.arange(max(0, n - 24), n)]))
            o_out, o_st = oracle.scalar_mul_var(enc[sel], k[sel])
            assert (out_q.cpu().numpy()[sel] == o_out).all() and (st_q.cpu().numpy()[sel] == o_st).all(), n
            # Element form: compare the group elements (encodings of the returned representatives)
            valid = torch.from_numpy(oracle.encode_to_curve(rng.integers(0, 256, (min(n, 512), 32), dtype=np.uint8))).to(dev)
            valid = valid.repeat((n + valid.shape[0] - 1) // valid.shape[0], 1)[:n].contiguous()
            P, _ = ctx.decompress(valid)
            with ctx.tuning(tiny_max=0):
                e_q = ctx.compress(ctx.scalar_mul_var_element(P, tk))
            with ctx.tuning(small_max=0):
                e_l = ctx.compress(ctx.scalar_mul_var_element(P, tk))
            assert torch.equal(e_q, e_l), n
            if n <= 4 * cus + 1:
                with ctx.tuning(tiny_max=10**6):
                    assert torch.equal(ctx.compress(ctx.scalar_mul_var_element(P, tk)), e_l), n
            assert torch.equal(ctx.compress(ctx.scalar_mul_var_element(P, tk)), e_l), n
            want, _ = ctx.scalar_mul_var(valid, tk)
            assert torch.equal(e_q, want), n


def test_tiny_batch_fixed_base_wave_kernel(ctx, oracle, torch_mod):
    """Up to one scalar per SIMD the fixed-base multiplication gives every scalar a wave (d377.hip k_scalar_mul_base_tiny: comb
    entries as row records, lane-spread additions, the wave's inversion), up to 64 per CU a quad of lanes
    (k_scalar_mul_base_small: sixteen encodings per inversion).  Same bytes as the lane-per-element kernel (small_max = 0) and
    as the oracle at sizes around every threshold, zero and extreme scalars included; the Element form likewise (as group
    elements)."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4402)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for n in (1, 2, 15, 16, 17, 1000, 4 * cus, 4 * cus + 1, 64 * cus - 1, 64 * cus + 1, 112 * cus + 1, 128 * cus + 1):
        k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        for i, v in enumerate([0, 1, 2, R_ORDER - 1, R_ORDER, R_ORDER + 1, (1 << 256) - 1, (1 << 18) - 1, 1 << 17, (1 << 17) + 1][:n]):
            k[i] = ibytes(v)
        tk = torch.from_numpy(k).to(dev)
        with ctx.tuning(small_max=0):                         # one lane per scalar
            out_l = ctx.scalar_mul_base(tk)
            el_l = ctx.compress(ctx.scalar_mul_base_element(tk))
        with ctx.tuning(tiny_max=0, small_max=10**6):         # one quad of lanes per scalar (k_scalar_mul_base_small)
            out_q = ctx.scalar_mul_base(tk)
            el_q = ctx.compress(ctx.scalar_mul_base_element(tk))
        assert torch.equal(out_q, out_l) and torch.equal(el_q, el_l), n
        if n <= 4 * cus + 1:
            with ctx.tuning(tiny_max=10**6):                  # one wave per scalar
                out_w = ctx.scalar_mul_base(tk)
                el_w = ctx.compress(ctx.scalar_mul_base_element(tk))
        else:
            out_w, el_w = out_q, el_q
        out_d = ctx.scalar_mul_base(tk)                       # whatever the size picks
        assert torch.equal(out_w, out_l) and torch.equal(out_d, out_l) and torch.equal(el_w, el_l) and torch.equal(el_l, out_l), n
        sel = np.unique(np.concatenate([np.arange(min(n, 24)), np.arange(max(0, n - 24), n)]))
        assert (out_w.cpu().numpy()[sel] == oracle.scalar_mul_base(k[sel])).all(), n


def test_tiny_batch_sqrt_family_four_per_wave(ctx, oracle, torch_mod):
    """Up to four elements per SIMD the square-root family -- sqrt_ratio_zeta (either root), decompress, compress, the round
    trip, encode_to_curve, hash_to_curve -- runs four elements per wave: the power chains of the square roots on the rows of
    the wave, one inversion per wave for the encodings (d377.hip k_*_tiny; hash_to_curve up to half the size: two pairs per wave,
    both maps of a pair in one pass over the rows).  Same bytes as the lane-per-element kernels
    (tiny_max = 0) at sizes around every edge (ragged last wave, one past the threshold), invalid encodings and zero
    numerators / denominators included, and as the oracle."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4403)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for n in (1, 2, 3, 4, 5, 7, 17, 1000, 8 * cus - 1, 8 * cus, 8 * cus + 1, 16 * cus - 1, 16 * cus, 16 * cus + 1):
        r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        r1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        r0[0] = 0                                            # zero numerator / Elligator of 0
        if n > 2:
            r1[2] = 0                                        # zero denominator
        if n > 3:
            r0[3] = 0; r1[3] = 0
        enc = oracle.encode_to_curve(rng.integers(0, 256, (min(n, 1024), 32), dtype=np.uint8))
        enc = np.tile(enc, ((n + enc.shape[0] - 1) // enc.shape[0], 1))[:n].copy()
        enc[::5] = rng.integers(0, 256, (len(enc[::5]), 32), dtype=np.uint8)      # mostly invalid
        if n > 1:
            enc[1] = 0                                       # the identity
        t0, t1, te = (torch.from_numpy(a).to(dev) for a in (r0, r1, enc))
        res = {}
        for mode, kv in (("lane", dict(tiny_max=0)), ("wave", dict(tiny_max=10**6)), ("default", {})):
            with ctx.tuning(**kv):
                P, st = ctx.decompress(te)
                valid = st == 0
                res[mode] = [ctx.sqrt_ratio_zeta(t0, t1), ctx.sqrt_ratio_zeta(t0, t1, root="min_curve"), (P, st), ctx.roundtrip(te),
                             ctx.encode_to_curve(t0), ctx.hash_to_curve(t0, t1), ctx.compress(P[valid].contiguous()) if bool(valid.any()) else None]
        flat = lambda r: [x for item in r if item is not None for x in (item if isinstance(item, (tuple, list)) else (item,))]
        for mode in ("wave", "default"):
            assert all(torch.equal(a, b) for a, b in zip(flat(res[mode]), flat(res["lane"]))), (n, mode)
        sel = np.unique(np.concatenate([np.arange(min(n, 40)), np.arange(max(0, n - 40), n)]))
        w = res["wave"]
        o_root, o_sq = oracle.sqrt_ratio_zeta(r0[sel], r1[sel])
        assert (w[0][0].cpu().numpy()[sel] == o_root).all() and (w[0][1].cpu().numpy()[sel] == o_sq).all(), n
        o_root, o_sq = oracle.sqrt_ratio_zeta_min_curve(r0[sel], r1[sel])
        assert (w[1][0].cpu().numpy()[sel] == o_root).all() and (w[1][1].cpu().numpy()[sel] == o_sq).all(), n
        o_P, o_st = oracle.decompress(enc[sel])
        assert (w[2][1].cpu().numpy()[sel] == o_st).all() and (w[2][0].cpu().numpy().view(np.uint64)[sel] == o_P).all(), n
        o_rt, o_rst = oracle.roundtrip(enc[sel])
        assert (w[3][0].cpu().numpy()[sel] == o_rt).all() and (w[3][1].cpu().numpy()[sel] == o_rst).all(), n
        assert (w[4].cpu().numpy()[sel] == oracle.encode_to_curve(r0[sel])).all(), n
        assert (w[5].cpu().numpy()[sel] == oracle.hash_to_curve(r0[sel], r1[sel])).all(), n


def test_hash_to_curve_exceptional_pairs_in_the_product_build(ctx, oracle, torch_mod):
    """The exceptional case of the Jacobi quartic's addition law (s1 s2 = +-1), on real inputs in the PRODUCT binary:
    tests/golden/hash_exceptional_pairs.json holds constructed pairs (r1, r2) that hit it.  Every route of hash_to_curve --
    two pairs per wave, four per wave, one lane per pair in chunks (a wave takes the fallback branch when one of its lanes
    needs it) -- and the Element form give the oracle's bytes, for the pairs alone, swapped, and scattered through batches of
    ordinary pairs."""
    import json
    torch = torch_mod
    dev = torch.device("cuda:0")
    pairs = json.load(open(os.path.join(ROOT, "tests", "golden", "hash_exceptional_pairs.json")))["pairs"]
    e1 = np.array([list(bytes.fromhex(p["r1"])) for p in pairs], np.uint8)
    e2 = np.array([list(bytes.fromhex(p["r2"])) for p in pairs], np.uint8)
    want_e = np.array([list(bytes.fromhex(p["encoding"])) for p in pairs], np.uint8)
    rng = np.random.default_rng(4404)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    # ... the last two in chunks of 5-6 and of 12-13 pairs per lane (the redo loop after a chunk's outputs: d377.hip k_hash_to_curve)
    for n in (len(pairs), 8 * cus - 3, 16 * cus - 3, 70001, 5 * 512 * cus + 3, 12 * 512 * cus + 77):
        r1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        r2 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        pos = np.arange(len(pairs)) if n == len(pairs) else rng.choice(n, len(pairs), replace=False)
        r1[pos], r2[pos] = e1, e2
        half = pos[: len(pos) // 2]
        r1[half], r2[half] = e2[: len(half)], e1[: len(half)]          # swapped
        t1, t2 = torch.from_numpy(r1).to(dev), torch.from_numpy(r2).to(dev)
        outs = []
        for kv in (dict(tiny_max=0), dict(tiny_max=10**6), {}):
            with ctx.tuning(**kv):
                outs.append(ctx.hash_to_curve(t1, t2).cpu().numpy())
                outs.append(ctx.compress(ctx.hash_to_curve_element(t1, t2)).cpu().numpy())
        assert all((o == outs[0]).all() for o in outs), n
        assert (outs[0][pos] == want_e).all(), n
        sel = np.unique(np.concatenate([pos, rng.integers(0, n, 40)]))
        assert (outs[0][sel] == oracle.hash_to_curve(r1[sel], r2[sel])).all(), n


def test_chunk_residency_is_checked(ctx):
    """The lane-set pool of the scratch areas assumes at most `sets` resident workgroups per CU of every kernel that
    claims a set.  d377_ctx_create verifies that with the occupancy query (and pads the launch's LDS where registers
    alone would admit more) instead of trusting what the compiler happened to allocate; the numbers are reported."""
    sets, blocks, pad = ctx.chunk_residency()
    assert sets == 3 and 1 <= blocks <= sets          # 3: the fixed-base kernel; every other chunked kernel is held to 2
    assert pad in (0, (160 * 1024) // 3 + 1024, (160 * 1024) // 4 + 1024)


def test_graph_replay_overlaps_eager_calls(ctx, oracle, torch_mod):
    """A captured graph takes no part in the event hand-over of the scratch areas, so a replay may run while an eager
    call from another stream uses the same lane-set areas.  Every workgroup claims a free set atomically and nothing
    resets the pool between launches, so both must come out right (before: the eager call's pool reset could hand
    two workgroups the same window tables).  Also: an MSM workspace that a graph has seen is retired, not freed, when
    a later call outgrows it, and the old graph still replays correctly afterwards."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(8123)
    n = 1 << 17
    r0 = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev)
    k = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev)
    enc = ctx.encode_to_curve(r0)
    want_out, want_st = ctx.scalar_mul_var(enc, k)
    want_fb = ctx.scalar_mul_base(k)
    m = 3000
    want_msm = ctx.msm(enc[:m], k[:m])[0].clone()
    torch.cuda.synchronize()
    sidx = np.arange(0, n, n // 64)
    o_out, o_st = oracle.scalar_mul_var(enc[sidx].cpu().numpy(), k[sidx].cpu().numpy())
    assert (want_out[sidx].cpu().numpy() == o_out).all()
    g_out = torch.empty_like(want_out); g_st = torch.empty_like(want_st)
    side = torch.cuda.Stream(device=dev)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        ctx.scalar_mul_var(enc, k, outs=[g_out, g_st])
        g_msm = ctx.msm(enc[:m], k[:m])[0]
    for rep in range(3):
        g_out.zero_(); g_st.fill_(9)
        torch.cuda.synchronize()
        gr.replay()                                     # on torch's current stream ...
        with torch.cuda.stream(side):                   # ... while eager calls on another stream use the same areas
            e_fb = ctx.scalar_mul_base(k)
            e_out, e_st = ctx.scalar_mul_var(enc, k)
        torch.cuda.synchronize()
        assert torch.equal(g_out, want_out) and torch.equal(g_st, want_st), rep
        assert torch.equal(e_out, want_out) and torch.equal(e_st, want_st) and torch.equal(e_fb, want_fb), rep
        assert torch.equal(g_msm, want_msm)
    # outgrow the MSM workspace the graph has seen, then replay the old graph
    big = 1 << 19
    rb = torch.from_numpy(rng.integers(0, 256, (big, 32), dtype=np.uint8)).to(dev)
    ctx.msm(ctx.encode_to_curve(rb), rb)
    torch.cuda.synchronize()
    g_out.zero_()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(g_out, want_out) and torch.equal(g_msm, want_msm)


def test_soak_tool_small():
    """tools/soak.py (the large-sample parity run whose full-size output is profiles/r03_soak.txt) at 2^14: every
    operation on seeded random inputs with invalid encodings, identity points, zero operands and the scalars
    0, 1, 2, 3, r-1, r, r+1, (r+-1)/2, 2^251-1, 2^256-1 mixed in, and a train of small calls of awkward sizes (the
    small-batch kernels), byte for byte against the oracle."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "14", "7"], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    legs = [l for l in r.stdout.splitlines() if " n = " in l]
    assert len(legs) >= 9 and all("bit-exact" in l for l in legs) and "MISMATCH" not in r.stdout, r.stdout   # every leg the tool has (9 today)
    assert any(l.startswith("msm_small") for l in legs) and any(l.startswith("hash_to_curve") for l in legs)


def test_full_size_var_base_2_22(ctx, torch_mod, oracle):
    """BASELINE config 4 size on one GPU: 2^22 (point, scalar) pairs.  Property at full size:
    [k]P computed with scalars k and k + r (same class mod r, different bytes) must give identical
    encodings, every status 0; a seeded sample is compared with the oracle."""
    torch = torch_mod
    n = 1 << 22
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(674)
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k[:, 31] &= 0x03                                  # k < 2^250 < r, so k + r < 2^252 fits 32 bytes
    P = ctx.encode_to_curve(r0)
    out, st = ctx.scalar_mul_var(P, k)
    # k + r on the host for a slice (pure byte arithmetic), full batch would take minutes in Python
    m = 1 << 14
    kh = k[:m].cpu().numpy()
    kr = np.array([list((int.from_bytes(bytes(x), "little") + R_ORDER).to_bytes(32, "little")) for x in kh],
                  dtype=np.uint8)
    out2, st2 = ctx.scalar_mul_var(P[:m], torch.from_numpy(kr).to(dev))
    torch.cuda.synchronize()
    assert int(st.sum().item()) == 0 and int(st2.sum().item()) == 0
    assert torch.equal(out[:m], out2)
    # no two distinct inputs collapse: outputs of distinct (P, k) are distinct encodings w.h.p.
    idx = np.arange(0, n, 16411)
    o_out, o_st = oracle.scalar_mul_var(P[idx].cpu().numpy(), k[idx].cpu().numpy())
    assert (out[idx].cpu().numpy() == o_out).all() and not o_st.any()


def test_wide_bytes_and_affine(ctx, oracle):
    """Fq::from_le_bytes_mod_order on 48/64-byte strings (src/fields/fq.rs:90-102),
    fused into encode_to_curve, and CurveGroup::normalize_batch (src/ark_curve/element.rs:74-81)."""
    import decaf377_amd as d
    rng = np.random.default_rng(675)
    n = 4099
    for length in (48, 64):
        raw = rng.integers(0, 256, (n, length), dtype=np.uint8)
        raw[0] = 0
        raw[1] = 0xFF
        raw[2, :32] = 0
        raw[3, 32:] = 0
        fq = ctx.fq_from_wide_bytes(raw)
        assert (fq == oracle.fq_from_wide_bytes(raw, length)).all()
        ints = [int.from_bytes(bytes(r), "little") % Q for r in raw[:64]]
        assert [int.from_bytes(bytes(r), "little") for r in fq[:64]] == ints
        enc = ctx.encode_to_curve_wide(raw)
        assert (enc == oracle.encode_to_curve_wide(raw, length)).all()
        assert (enc == ctx.encode_to_curve(fq)).all()
    with pytest.raises(ValueError):                   # the host mirror refuses other widths before the call ...
        ctx.fq_from_wide_bytes(rng.integers(0, 256, (4, 40), dtype=np.uint8))
    import ctypes                                     # ... and so does the C ABI itself
    bad = np.zeros((4, 40), np.uint8)
    out = np.zeros((4, 32), np.uint8)
    assert ctx._lib.d377_batch_fq_from_wide_bytes(ctx._h, bad.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(40),
                                                  ctypes.c_size_t(4), out.ctypes.data_as(ctypes.c_void_p)) == -2
    P = oracle.double_xyzt(oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
    xy = ctx.to_affine(P)
    assert (xy == oracle.to_affine(P)).all()
    # batched inversion (Montgomery's trick per lane): sizes around the lanes-per-launch boundary, and
    # records with z = 0 (no inverse: zero output) that must not poison their lane's other elements
    for m in (1, 2, 255, 257, 70000, (1 << 19) + 11):
        big = np.tile(P, (m // n + 1, 1))[:m].copy()
        big[::1237, 8:12] = 0
        want = oracle.to_affine(big[: min(m, 5000)])
        got = ctx.to_affine(big)
        assert (got[: want.shape[0]] == want).all()
        assert not got[::1237].any()
        assert (got[-3:] == oracle.to_affine(big[-3:])).all()


def test_api_contract_edges(ctx, torch_mod, oracle):
    """Boundary behaviour of the C ABI: misaligned device records are refused, two contexts on one
    device are independent, a ragged large batch grid-strides correctly, outputs never alias state."""
    import decaf377_amd as d
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(676)
    # misaligned device pointer -> D377_ERR_ARG, not a fault
    buf = torch.zeros(32 * 64 + 8, dtype=torch.uint8, device=dev)
    mis = buf[8:8 + 32 * 64].view(64, 32)
    assert mis.data_ptr() % 16 == 8
    with pytest.raises(d.NativeError):
        ctx.encode_to_curve(mis)
    # two contexts, interleaved calls, same answers
    ctx2 = d.Context([0])
    r0 = rng.integers(0, 256, (2048, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (2048, 32), dtype=np.uint8)
    e1 = ctx.encode_to_curve(r0)
    e2 = ctx2.encode_to_curve(r0)
    o1, _ = ctx.scalar_mul_var(e1, k)
    o2, _ = ctx2.scalar_mul_var(e2, k)
    o1b, _ = ctx.scalar_mul_var(e1, k)
    assert (e1 == e2).all() and (o1 == o2).all() and (o1 == o1b).all()
    ctx2.close()
    # ragged large batch on the device path: 2^21 + 3 elements
    n = (1 << 21) + 3
    g = torch.Generator(device=dev).manual_seed(677)
    r = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    enc = ctx.encode_to_curve(r)
    rt, st = ctx.roundtrip(enc)
    torch.cuda.synchronize()
    assert torch.equal(rt, enc) and int(st.sum().item()) == 0
    tail = slice(n - 5, n)
    assert (enc[tail].cpu().numpy() == oracle.encode_to_curve(r[tail].cpu().numpy())).all()


def test_neg_identity_generator(ctx, oracle, kats):
    """neg, is_identity, IDENTITY, GENERATOR (src/min_curve/element.rs:53-117,324-332)."""
    gen, ident = ctx.generator(), ctx.identity()
    assert (gen == oracle.generator_xyzt()).all()
    assert [int(v) for v in gen[0:4]] == kats["generator"]["x_mont"]
    assert bytes(ctx.compress(ident.reshape(1, 16))[0]) == bytes(32)
    assert bytes(ctx.compress(gen.reshape(1, 16))[0]).hex() == kats["generator"]["hex"]
    rng = np.random.default_rng(678)
    n = 1500
    P = oracle.double_xyzt(oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
    P[0], P[1] = ident, gen
    N = ctx.neg(P)
    s = ctx.add(P, N)                                  # P + (-P) = identity
    assert ctx.is_identity(s).all()
    assert ctx.is_identity(P)[0] == 1 and not ctx.is_identity(P)[1:].any()
    assert (ctx.compress(s) == 0).all()
    # -P has x and t negated, y and z unchanged (exact limbs)
    assert (N[:, 4:12] == P[:, 4:12]).all()
    assert ctx.eq(ctx.neg(N), P).all()


def test_multi_device_context_slicing(ctx, oracle):
    """A context that owns several devices slices the batch into contiguous shards.
    With one physical GPU the same device is listed twice / three times: that exercises the whole
    multi-device host path (per-device tables, streams, slices, MSM partial-sum combine; the small sums sliced by SUMS)."""
    import decaf377_amd as d
    rng = np.random.default_rng(679)
    for ids in ([0, 0], [0, 0, 0]):
        c = d.Context(ids)
        assert c.device_ids == ids
        for n in (1, 2, 5, 1001, 4096):
            r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            enc = c.encode_to_curve(r0)
            assert (enc == oracle.encode_to_curve(r0)).all()
            raw = enc.copy()
            raw[::7, 31] |= 0x80                      # some invalid encodings, spread over all shards
            out, st = c.scalar_mul_var(raw, k)
            o_out, o_st = oracle.scalar_mul_var(raw, k)
            assert (out == o_out).all() and (st == o_st).all()
            xyzt, st = c.decompress(enc)
            e, _, _ = c.msm(xyzt, k)
            assert bytes(e) == bytes(oracle.msm(xyzt, k)[0])
            e2, _, st2 = c.msm(raw, k)
            keep = o_st == 0
            assert (st2 == o_st).all()
            assert bytes(e2) == bytes(oracle.msm(xyzt[keep], k[keep])[0])
            for m in (2, 3):                          # d377_batch_msm_small: whole sums per device, outputs and statuses in place
                t = m * (n // m)
                if t == 0:
                    continue
                es, xs = c.msm_small(xyzt[:t], k[:t], m, elements=True)
                es1 = ctx.msm_small(xyzt[:t], k[:t], m)       # the single-device context (tests/test_msm.py pins it to the oracle)
                assert (es == es1).all() and (c.compress(xs) == es1).all(), (ids, n, m)
                ee, st3 = c.msm_small(raw[:t], k[:t], m)
                ee1, st1 = ctx.msm_small(raw[:t], k[:t], m)
                assert (ee == ee1).all() and (st3 == st1).all() and (st3 == o_st[:t]).all(), (ids, n, m)
        c.close()
    # a lazy comb on every device of a context: each builds its own on the first fixed-base slice it is handed
    c = d.Context([0, 0], comb_bits=18, comb_lazy=True)
    assert [c.comb_info(i)[1] for i in (0, 1)] == [False, False]
    k = rng.integers(0, 256, (1001, 32), dtype=np.uint8)
    assert (c.scalar_mul_base(k) == oracle.scalar_mul_base(k)).all()
    assert [c.comb_info(i)[1] for i in (0, 1)] == [True, True] and c.comb_info(1)[0] == 18
    c.close()


def test_field_regression_seeds_gpu(ctx, oracle, kats):
    """The same proptest regression seeds through the kernels."""
    rs = kats["regression_seeds"]
    z = np.zeros((3, rs["fq_wide_zero_len"]), np.uint8)
    assert not ctx.fq_from_wide_bytes(z).any()
    ks = np.zeros((2, 32), np.uint8)
    ks[1] = np.frombuffer((1 << rs["fr_values_pow2"][0]).to_bytes(32, "little"), np.uint8)
    out = ctx.scalar_mul_base(ks)
    assert (out == oracle.scalar_mul_base(ks)).all() and not out[0].any()
    g = np.tile(frombytes([kats["generator"]["hex"]]), (2, 1))
    out2, st = ctx.scalar_mul_var(g, ks)
    assert (out2 == out).all() and not st.any()


def test_fq_field_ops(ctx, oracle, kats):
    """SURVEY 8a rows a1-a3 through the C ABI: Fq add/sub/mul/square/neg/inverse and byte I/O on the
    reference's in-memory form, against the oracle and big-integer arithmetic; the reference's own
    examples (src/fields/fq/arkworks.rs:603-673)."""
    rng = np.random.default_rng(680)
    n = 4096
    raw = rng.integers(0, 256, (2, n, 32), dtype=np.uint8)
    for j, e in enumerate([0, 1, Q - 1, Q, Q + 1, (1 << 256) - 1, 2, 1 << 192]):
        raw[0, j] = ibytes(e)
        raw[1, n - 1 - j] = ibytes(e)
    a = oracle.fq_from_bytes_mod_order(raw[0])
    b = oracle.fq_from_bytes_mod_order(raw[1])
    ai = [int.from_bytes(bytes(x), "little") % Q for x in raw[0]]
    bi = [int.from_bytes(bytes(x), "little") % Q for x in raw[1]]
    val = lambda recs: [int.from_bytes(bytes(x), "little") for x in ctx.fq_to_bytes(recs)]
    assert val(a) == ai
    out, _ = ctx.fq_op("mul", a, b)
    assert (out == oracle.fq_mul_mont(a, b)).all()
    out, _ = ctx.fq_op("square", a)
    assert (out == oracle.fq_mul_mont(a, a)).all()
    assert val(ctx.fq_op("add", a, b)[0]) == [(x + y) % Q for x, y in zip(ai, bi)]
    assert val(ctx.fq_op("sub", a, b)[0]) == [(x - y) % Q for x, y in zip(ai, bi)]
    assert val(ctx.fq_op("neg", a)[0]) == [(-x) % Q for x in ai]
    inv, st = ctx.fq_op("inverse", a)
    assert list(st) == [1 if x == 0 else 0 for x in ai]
    assert val(inv) == [pow(x, -1, Q) if x else 0 for x in ai]
    prod, _ = ctx.fq_op("mul", a, inv)
    assert val(prod) == [1 if x else 0 for x in ai]
    # from_bytes_checked: canonical strings only (src/fields/fq.rs:108-115, :149-152)
    recs, st = ctx.fq_from_bytes_checked(raw[0])
    want_bad = [int.from_bytes(bytes(x), "little") >= Q for x in raw[0]]
    assert list(st) == [int(v) for v in want_bad]
    o_recs, o_st = oracle.fq_from_bytes_checked(raw[0])
    assert (st == o_st).all() and (recs == o_recs).all()
    # (-1)^2 == 1, (p + 1) reduces to 1
    m1 = oracle.fq_from_bytes_mod_order(ibytes(Q - 1).reshape(1, 32))
    assert val(ctx.fq_op("square", m1)[0]) == [1]
    p1 = np.array(kats["fq_examples"]["p_plus_1_bytes"], dtype=np.uint8).reshape(1, 32)
    assert val(oracle.fq_from_bytes_mod_order(p1)) == [1]


def test_pipelined_host_path(ctx, oracle):
    """Host-pointer calls with >= 2^19 records are pipelined in 2^18-record chunks (copy stream +
    compute stream, double-buffered staging): results must equal the unpipelined device path,
    including a ragged tail and per-element status."""
    import torch
    rng = np.random.default_rng(681)
    n = (1 << 19) + (1 << 18) + 77                     # 3 full chunks boundary + ragged tail
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    enc = ctx.encode_to_curve(r0)                      # pipelined host path
    dev = torch.device("cuda:0")
    enc_d = ctx.encode_to_curve(torch.from_numpy(r0).to(dev))
    assert (enc == enc_d.cpu().numpy()).all()
    raw = enc.copy()
    raw[::1000, 31] |= 0x40                            # invalid encodings scattered over all chunks
    out, st = ctx.scalar_mul_var(raw, k)
    out_d, st_d = ctx.scalar_mul_var(torch.from_numpy(raw).to(dev), torch.from_numpy(k).to(dev))
    assert (out == out_d.cpu().numpy()).all() and (st == st_d.cpu().numpy()).all()
    assert st.sum() == len(range(0, n, 1000))
    idx = np.concatenate([np.arange(0, n, 30011), [n - 1, (1 << 18) - 1, 1 << 18, (1 << 19) - 1, 1 << 19]])
    o_out, o_st = oracle.scalar_mul_var(raw[idx], k[idx])
    assert (out[idx] == o_out).all() and (st[idx] == o_st).all()
    xyzt, st = ctx.decompress(enc)                     # 128-byte output records through the pipeline
    assert not st.any()
    assert (ctx.compress(xyzt) == enc).all()


def test_fr_bytes(ctx, oracle, kats):
    """Fr byte handling (src/fields/fr.rs:82-107, examples fr/arkworks.rs:590-660)."""
    rng = np.random.default_rng(682)
    n = 5000
    raw = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for j, e in enumerate([0, 1, R_ORDER - 1, R_ORDER, R_ORDER + 1, (1 << 256) - 1, 32 * R_ORDER, 53 * R_ORDER + 5, 1 << 192]):
        raw[j] = ibytes(e)
    red = ctx.fr_from_le_bytes_mod_order(raw)
    assert (red == oracle.fr_from_bytes_mod_order(raw)).all()
    assert [int.from_bytes(bytes(x), "little") for x in red[:64]] == [int.from_bytes(bytes(x), "little") % R_ORDER for x in raw[:64]]
    out, st = ctx.fr_from_bytes_checked(raw)
    assert (st == oracle.fr_from_bytes_checked(raw)).all()
    ok = st == 0
    assert (out[ok] == raw[ok]).all() and not out[~ok].any()
    p1 = np.array(kats["fr_examples"]["p_plus_1_bytes"], dtype=np.uint8).reshape(1, 32)
    assert bytes(ctx.fr_from_le_bytes_mod_order(p1)[0]) == (1).to_bytes(32, "little")


# --- round 2: oracle parity for a9 / a3, the min_curve root (a4'), every _dev export, streams, devices ---
def test_neg_is_identity_eq_inverse_vs_oracle(ctx, oracle):
    """SURVEY 8a rows a9 and a3 against the oracle's restatements, bit for bit: Element neg
    (src/min_curve/element.rs:324-332), is_identity (:113-117), PartialEq (:334-340), Fq::inverse
    (src/fields/fq/u64/wrapper.rs:104-112)."""
    rng = np.random.default_rng(701)
    n = 3000
    P = oracle.double_xyzt(oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
    Qp = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    P[0] = oracle.identity_xyzt()
    P[1] = oracle.generator_xyzt()
    assert (ctx.neg(P) == oracle.neg_xyzt(P)).all()
    mixed = P.copy()
    mixed[::5] = oracle.add_xyzt(P[::5], oracle.neg_xyzt(P[::5]))       # some identities in non-canonical coordinates
    assert (ctx.is_identity(mixed) == oracle.is_identity(mixed)).all() and ctx.is_identity(mixed)[::5].all()
    # equality: same representative, another representative of the same element, negation, unrelated
    other = oracle.add_xyzt(P, np.tile(oracle.identity_xyzt(), (n, 1)))
    assert (other != P).any()
    pairs_l = np.concatenate([P, P, P, P])
    pairs_r = np.concatenate([P, other, oracle.neg_xyzt(P), Qp])
    got = ctx.eq(pairs_l, pairs_r)
    want = oracle.eq_xyzt(pairs_l, pairs_r)
    assert (got == want).all()
    assert got[:2 * n].all() and not got[3 * n:].any()
    a = oracle.fq_from_bytes_mod_order(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    a[0] = 0
    for name, code in (("add", 0), ("sub", 1), ("mul", 2), ("square", 3), ("neg", 4), ("inverse", 5)):
        b = a[::-1].copy()
        out, st = ctx.fq_op(name, a, b if code <= 2 else None)
        o_out, o_st = oracle.fq_op(code, a, b if code <= 2 else None)
        assert (out == o_out).all() and (st == o_st).all(), name


def test_min_curve_root_2_16(ctx, oracle):
    """SURVEY 8a row a4': `Fq::non_arkworks_sqrt_ratio_zeta` (src/min_curve/invsqrt.rs:11-95, the root the
    min_curve backend returns) on 2^16 random pairs plus the edge pairs, bit-exact against the oracle's
    constant-time Tonelli-Shanks restatement; flags equal the arkworks root's.  The convention only exists
    on the raw square-root entry point: the group-level kernels take no such argument, so their outputs
    cannot depend on it (and both roots give the same encodings: test_oracle.py)."""
    rng = np.random.default_rng(702)
    n = 1 << 16
    num = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    den = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    edges = [(0, 1), (1, 0), (0, 0), (1, 1), (1 << 248, 1 << 248), (Q, 5), (1, Q), (Q - 1, 1), (1, Q - 1)]
    for j, (u, v) in enumerate(edges):
        num[j], den[j] = ibytes(u), ibytes(v)
    root, ws = ctx.sqrt_ratio_zeta(num, den, root="min_curve")
    o_root, o_ws, _ = oracle.run_threads("sqrt_ratio_zeta_min_curve", num, den, os.cpu_count() or 4)
    assert (ws == o_ws).all()
    assert (root == o_root).all()
    root_a, ws_a = ctx.sqrt_ratio_zeta(num, den)
    assert (ws_a == ws).all()
    differs = (root_a != root).any(axis=1)
    assert 0.4 < differs.mean() < 0.6
    ra = [int.from_bytes(bytes(x), "little") for x in root_a[:512]]
    rm = [int.from_bytes(bytes(x), "little") for x in root[:512]]
    assert all(y in (x, (Q - x) % Q) for x, y in zip(ra, rm))
    import decaf377_amd as d
    with pytest.raises(KeyError):
        ctx.sqrt_ratio_zeta(num[:4], den[:4], root="other")


def _fq_ints(oracle, mont):
    """[n, 4] Montgomery limbs -> python ints"""
    return [int.from_bytes(bytes(r), "little") for r in oracle.fq_to_bytes(np.ascontiguousarray(mont).reshape(-1, 4))]


def test_element_form_operations(ctx, oracle):
    """The reference's own signatures (Elements in and out): `Element * Fr` (src/min_curve/ops.rs:89-95),
    `GENERATOR * Fr`, `vartime_compress_to_field` (element.rs:163-181), `encode_to_curve` / `hash_to_curve` as
    Elements (element.rs:190-244).  Scalar multiples are compared as group elements (encoding, decaf equality) and
    checked to be valid extended points; the maps and compress_to_field are compared limb for limb."""
    rng = np.random.default_rng(801)
    n = 1 << 12
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    r1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    for i, v in enumerate([0, 1, 2, R_ORDER - 1, R_ORDER, R_ORDER + 1, (1 << 256) - 1]):
        k[i] = ibytes(v)
    P = oracle.elligator_map_xyzt(r0)                               # projective, Z != 1
    P[40] = oracle.identity_xyzt()
    P[41] = oracle.add_xyzt(P[42:43], oracle.neg_xyzt(P[42:43]))[0]  # the identity with Z != 1
    out = ctx.scalar_mul_var_element(P, k)
    ref = oracle.scalar_mul_xyzt(P, k)
    assert (ctx.compress(out) == oracle.compress(ref)).all()
    assert oracle.eq_xyzt(out, ref).all()
    assert list(np.nonzero(ctx.is_identity(out))[0]) == list(np.nonzero(oracle.is_identity(ref))[0])
    assert ctx.is_identity(out)[[0, 4, 40, 41]].all()
    m = 64                                                           # a valid extended point: on the curve, XY = ZT
    c = _fq_ints(oracle, out[:m].reshape(-1, 4))
    for i in range(m):
        X, Y, Z, T = c[4 * i:4 * i + 4]
        assert (X * Y - Z * T) % Q == 0 and (-X * X + Y * Y - Z * Z - 3021 * T * T) % Q == 0 and Z % Q != 0
    # in place, and the Encoding-form kernel agrees
    buf = P.copy()
    ctx.scalar_mul_var_element(buf, k, outs=[buf])
    assert (buf == out).all()
    enc_out, st = ctx.scalar_mul_var(oracle.compress(P), k)
    assert not st.any() and (enc_out == ctx.compress(out)).all()
    # fixed base
    gb = ctx.scalar_mul_base_element(k)
    assert (ctx.compress(gb) == oracle.scalar_mul_base(k)).all()
    assert oracle.eq_xyzt(gb, oracle.scalar_mul_xyzt(np.tile(oracle.generator_xyzt(), (n, 1)), k)).all()
    # compress_to_field: the Fq whose bytes are the encoding
    f = ctx.compress_to_field(P)
    assert (f == oracle.compress_to_field(P)).all()
    assert (ctx.fq_to_bytes(f) == oracle.compress(P)).all()
    # the maps, as Elements: the coordinates the reference formulas give
    assert (ctx.encode_to_curve_element(r0) == oracle.elligator_map_xyzt(r0)).all()
    assert (ctx.hash_to_curve_element(r0, r1) == oracle.hash_to_curve_xyzt(r0, r1)).all()


def test_element_scalar_mul_full_size(ctx, oracle, torch_mod):
    """2^20 Elements x scalars on device tensors: same group elements as the Encoding-form kernel (the one the
    bench measures and the other tests pin to the oracle), a sample against the oracle, and (a + b) P = aP + bP."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    n = 1 << 20
    g = torch.Generator(device=dev).manual_seed(802)
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k2 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    P = ctx.double(ctx.encode_to_curve_element(r0))
    enc = ctx.compress(P)
    out = ctx.scalar_mul_var_element(P, k)
    ref, st = ctx.scalar_mul_var(enc, k)
    torch.cuda.synchronize()
    assert int(st.sum().item()) == 0 and torch.equal(ctx.compress(out), ref)
    idx = np.arange(0, n, n // 128)
    Ph, kh = P.cpu().numpy().view(np.uint64)[idx], k.cpu().numpy()[idx]
    assert (ref.cpu().numpy()[idx] == oracle.compress(oracle.scalar_mul_xyzt(Ph, kh))).all()
    ksum, _ = ctx.fr_op("add", k, k2)
    lhs = ctx.scalar_mul_var_element(P, ksum)
    rhs = ctx.add(out, ctx.scalar_mul_var_element(P, k2))
    torch.cuda.synchronize()
    assert bool(ctx.eq(lhs, rhs).all().item())


def test_fr_arithmetic(ctx, oracle, torch_mod):
    """Fr add / sub / mul / square / neg / inverse (src/fields/fr/u64/wrapper.rs:76-108) and the wide reduction
    (src/fields/fr.rs:82-94) on 32-byte scalars: host and device paths against the oracle, edge values included."""
    torch = torch_mod
    rng = np.random.default_rng(803)
    n = 1 << 13
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    edges = [0, 1, 2, R_ORDER - 1, R_ORDER, R_ORDER + 1, (1 << 256) - 1, 1 << 255]
    for i, v in enumerate(edges):
        a[i] = ibytes(v)
        b[len(edges) - 1 - i] = ibytes(v)
    t = lambda x: torch.from_numpy(x).to("cuda:0")
    for code, op in enumerate(["add", "sub", "mul", "square", "neg", "inverse"]):
        m = n if op != "inverse" else 1 << 10                      # the oracle's inversion is slow
        args = (a[:m], b[:m]) if code <= 2 else (a[:m],)
        want, wst = oracle.fr_op(code, *args)
        got, st = ctx.fr_op(op, *args)
        assert (got == want).all() and (st == wst).all(), op
        gd, sd = ctx.fr_op(op, *[t(x) for x in args])
        assert (gd.cpu().numpy() == want).all() and (sd.cpu().numpy() == wst).all(), op
    assert wst[0] == 1 and not want[0].any()                        # inverse(0): None
    ai = [int.from_bytes(bytes(x), "little") % R_ORDER for x in a[:256]]
    inv, _ = ctx.fr_op("inverse", a[:256])
    assert all(x == 0 or x * int.from_bytes(bytes(y), "little") % R_ORDER == 1 for x, y in zip(ai, inv))
    for length in (48, 64):
        d = rng.integers(0, 256, (n, length), dtype=np.uint8)
        d[0] = 255
        d[1] = 0
        want = oracle.fr_from_wide_bytes(d)
        assert (ctx.fr_from_wide_bytes(d) == want).all()
        assert (ctx.fr_from_wide_bytes(t(d)).cpu().numpy() == want).all()
        assert [int.from_bytes(bytes(x), "little") for x in want[:64]] == [int.from_bytes(bytes(x), "little") % R_ORDER for x in d[:64]]
    with pytest.raises(ValueError):
        ctx.fr_op("add", a)
    with pytest.raises(ValueError):
        ctx.fr_from_wide_bytes(a)


def _dev_cases(ctx, oracle, torch, n):
    """name of the _dev export -> (callable on device tensors, callable on host arrays)."""
    rng = np.random.default_rng(703)
    dev = torch.device("cuda:0")
    u8 = lambda m, w=32: rng.integers(0, 256, (m, w), dtype=np.uint8)
    r0, r1, k = u8(n), u8(n), u8(n)
    enc = oracle.encode_to_curve(r0)
    raw = enc.copy()
    raw[::9, 31] |= 0x40
    P = oracle.elligator_map_xyzt(r0)
    Qp = oracle.double_xyzt(oracle.elligator_map_xyzt(r1))
    a = oracle.fq_from_bytes_mod_order(r0)
    b = oracle.fq_from_bytes_mod_order(r1)
    a[0] = 0
    w48, w64 = u8(n, 48), u8(n, 64)
    t = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x).to(dev)
    cases = {
        "d377_batch_sqrt_ratio_zeta_dev": lambda f: f.sqrt_ratio_zeta,
        "d377_batch_sqrt_ratio_zeta_ex_dev": lambda f: (lambda x, y: f.sqrt_ratio_zeta(x, y, root="min_curve")),
        "d377_batch_decompress_dev": lambda f: f.decompress,
        "d377_batch_compress_dev": lambda f: f.compress,
        "d377_batch_roundtrip_dev": lambda f: f.roundtrip,
        "d377_batch_scalar_mul_base_dev": lambda f: f.scalar_mul_base,
        "d377_batch_scalar_mul_var_dev": lambda f: f.scalar_mul_var,
        "d377_batch_encode_to_curve_dev": lambda f: f.encode_to_curve,
        "d377_batch_hash_to_curve_dev": lambda f: f.hash_to_curve,
        "d377_batch_add_dev": lambda f: f.add,
        "d377_batch_sub_dev": lambda f: f.sub,
        "d377_batch_double_dev": lambda f: f.double,
        "d377_batch_eq_dev": lambda f: f.eq,
        "d377_batch_neg_dev": lambda f: f.neg,
        "d377_batch_is_identity_dev": lambda f: f.is_identity,
        "d377_batch_to_affine_dev": lambda f: f.to_affine,
        "d377_batch_fq_from_wide_bytes_dev": lambda f: f.fq_from_wide_bytes,
        "d377_batch_encode_to_curve_wide_dev": lambda f: f.encode_to_curve_wide,
        "d377_batch_fq_op_dev": lambda f: f.fq_op,
        "d377_batch_scalar_mul_var_element_dev": lambda f: f.scalar_mul_var_element,
        "d377_batch_scalar_mul_base_element_dev": lambda f: f.scalar_mul_base_element,
        "d377_batch_compress_to_field_dev": lambda f: f.compress_to_field,
        "d377_batch_encode_to_curve_element_dev": lambda f: f.encode_to_curve_element,
        "d377_batch_hash_to_curve_element_dev": lambda f: f.hash_to_curve_element,
        "d377_batch_fr_op_dev": lambda f: f.fr_op,
        "d377_batch_fr_from_wide_bytes_dev": lambda f: f.fr_from_wide_bytes,
        "d377_batch_fq_from_bytes_checked_dev": lambda f: f.fq_from_bytes_checked,
        "d377_batch_fq_to_bytes_dev": lambda f: f.fq_to_bytes,
        "d377_batch_fr_from_le_bytes_mod_order_dev": lambda f: f.fr_from_le_bytes_mod_order,
        "d377_batch_fr_from_bytes_checked_dev": lambda f: f.fr_from_bytes_checked,
        "d377_msm_dev": lambda f: f.msm,
        "d377_msm_encoded_dev": lambda f: f.msm,
        "d377_batch_msm_small_dev": lambda f: (lambda p, k_, m: f.msm_small(p, k_, m, elements=True)),   # Encodings and Element records
        "d377_batch_msm_small_encoded_dev": lambda f: (lambda p, k_, m: f.msm_small(p, k_, m, elements=m == 5)),
        "d377_sum_elements_dev": None,
        "d377_ctx_starved_counter_dev": None,          # test_starved_call_is_an_error_not_a_silent_partial_output
        "d377_batch_sharded_dev": None,
    }
    args = {
        "d377_batch_sqrt_ratio_zeta_dev": [(r0, k)], "d377_batch_sqrt_ratio_zeta_ex_dev": [(r0, k)],
        "d377_batch_decompress_dev": [(raw,)], "d377_batch_compress_dev": [(Qp,)], "d377_batch_roundtrip_dev": [(raw,)],
        "d377_batch_scalar_mul_base_dev": [(k,)], "d377_batch_scalar_mul_var_dev": [(raw, k)],
        "d377_batch_encode_to_curve_dev": [(r0,)], "d377_batch_hash_to_curve_dev": [(r0, r1)],
        "d377_batch_add_dev": [(P, Qp)], "d377_batch_sub_dev": [(P, Qp), (Qp, Qp)], "d377_batch_double_dev": [(Qp,)], "d377_batch_eq_dev": [(P, Qp), (Qp, Qp)],
        "d377_batch_neg_dev": [(Qp,)], "d377_batch_is_identity_dev": [(oracle.add_xyzt(P, oracle.neg_xyzt(P)),), (P,)],
        "d377_batch_to_affine_dev": [(Qp,)], "d377_batch_fq_from_wide_bytes_dev": [(w48,), (w64,)],
        "d377_batch_encode_to_curve_wide_dev": [(w48,), (w64,)],
        "d377_batch_fq_op_dev": [("add", a, b), ("sub", a, b), ("mul", a, b), ("square", a), ("neg", a), ("inverse", a)],
        "d377_msm_dev": [(Qp, k)], "d377_msm_encoded_dev": [(raw, k)],
        "d377_batch_msm_small_dev": [(Qp, k, 5), (Qp, k, 2)],                      # 500 sums: a wave per sum; 1 250: a lane per sum
        "d377_batch_msm_small_encoded_dev": [(raw, k, 5), (raw, k, 2)],
        "d377_batch_scalar_mul_var_element_dev": [(Qp, k)], "d377_batch_scalar_mul_base_element_dev": [(k,)],
        "d377_batch_compress_to_field_dev": [(Qp,)], "d377_batch_encode_to_curve_element_dev": [(r0,)],
        "d377_batch_hash_to_curve_element_dev": [(r0, r1)],
        "d377_batch_fr_op_dev": [("add", r0, k), ("sub", r0, k), ("mul", r0, k), ("square", r0), ("neg", r0), ("inverse", r0)],
        "d377_batch_fr_from_wide_bytes_dev": [(w48,), (w64,)],
        "d377_batch_fq_from_bytes_checked_dev": [(k,)], "d377_batch_fq_to_bytes_dev": [(a,)],
        "d377_batch_fr_from_le_bytes_mod_order_dev": [(k,)], "d377_batch_fr_from_bytes_checked_dev": [(k,)],
    }
    return cases, args, t, (P, Qp, raw, k, r0)


def _same(x, y):
    import torch
    if x is None or y is None:
        return x is None and y is None
    xs = x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    ys = y.cpu().numpy() if isinstance(y, torch.Tensor) else np.asarray(y)
    return xs.shape == ys.shape and (xs.view(np.uint8) == ys.view(np.uint8)).all()


def test_every_dev_export(ctx, oracle, torch_mod):
    """Drives EVERY `_dev` export of include/decaf377_amd.h on device tensors and compares it, byte for
    byte, with the host-pointer form of the same entry point (which the other tests pin to the oracle);
    MSM results (not unique as coordinates) are compared as encodings."""
    import decaf377_amd._native as nat
    torch = torch_mod
    n = 2500
    cases, args, t, (P, Qp, raw, k, r0) = _dev_cases(ctx, oracle, torch, n)
    dev_exports = {e for e in nat.EXPORTS if e.endswith("_dev")}
    assert dev_exports == set(cases), dev_exports ^ set(cases)
    for name, getter in cases.items():
        if getter is None:
            continue
        f = getter(ctx)
        for a in args[name]:
            host = f(*a)
            devr = f(*[t(x) if isinstance(x, np.ndarray) else x for x in a])
            torch.cuda.synchronize()
            host = host if isinstance(host, (tuple, list)) else (host,)
            devr = devr if isinstance(devr, (tuple, list)) else (devr,)
            if name.startswith("d377_msm"):
                assert _same(host[0], devr[0]) and _same(host[2], devr[2]), name      # encoding and statuses
                assert oracle.eq_xyzt(host[1].reshape(1, 16), devr[1].cpu().numpy().view(np.uint64).reshape(1, 16)).all()
            else:
                assert len(host) == len(devr) and all(_same(h, g) for h, g in zip(host, devr)), name
    # sum of Element records held in HBM
    enc, xyzt = ctx.sum_elements(t(Qp[:37]))
    acc = Qp[0:1]
    for i in range(1, 37):
        acc = oracle.add_xyzt(acc, Qp[i:i + 1])
    assert bytes(enc.cpu().numpy()) == bytes(oracle.compress(acc)[0])
    assert oracle.eq_xyzt(xyzt.cpu().numpy().view(np.uint64).reshape(1, 16), acc).all()


def test_sharded_device_path(oracle, torch_mod):
    """d377_batch_sharded_dev: an HBM-resident batch split over the context's devices by peer copies (SURVEY
    8e at the C ABI).  One physical GPU listed two and three times exercises slicing, staging, the event
    ordering and the gather; results equal the single-device entry points, ragged sizes included."""
    import decaf377_amd as d
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(704)
    t = lambda x: torch.from_numpy(x).to(dev)
    for ids in ([0, 0], [0, 0, 0]):
        c = d.Context(ids)
        for n in (1, 17, 1000, 40001):
            r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            r1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            enc = c.sharded("encode_to_curve", t(r0))[0]
            assert _same(enc, c.encode_to_curve(r0))
            raw = enc.clone()
            raw[::7, 31] |= 0x80
            out, st = c.sharded("scalar_mul_var", raw, t(k))
            h_out, h_st = c.scalar_mul_var(raw.cpu().numpy(), k)
            assert _same(out, h_out) and _same(st, h_st)
            rt, st = c.sharded("roundtrip", raw)
            h_rt, h_st = c.roundtrip(raw.cpu().numpy())
            assert _same(rt, h_rt) and _same(st, h_st)
            xyzt, st = c.sharded("decompress", enc)
            assert _same(c.sharded("compress", xyzt)[0], enc)
            assert _same(c.sharded("scalar_mul_base", t(k))[0], c.scalar_mul_base(k))
            assert _same(c.sharded("hash_to_curve", t(r0), t(r1))[0], c.hash_to_curve(r0, r1))
            root, ws = c.sharded("sqrt_ratio_zeta", t(r0), t(r1))
            h_root, h_ws = c.sqrt_ratio_zeta(r0, r1)
            assert _same(root, h_root) and _same(ws, h_ws)
            el = c.sharded("scalar_mul_var_element", xyzt, t(k))[0]
            assert _same(el, c.scalar_mul_var_element(xyzt.cpu().numpy().view(np.uint64), k))
            assert _same(c.sharded("scalar_mul_base_element", t(k))[0], c.scalar_mul_base_element(k))
        c.close()


def test_concurrent_streams_share_scratch(ctx, oracle, torch_mod):
    """Two torch streams issue variable-base batches and MSMs at the same time.  The per-device window tables
    and the MSM workspace are handed from launch to launch by events (include/decaf377_amd.h, "Threads and
    streams"), so the results must equal the one-at-a-time results; before that hand-over existed the two
    launches overwrote each other's tables."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(705)
    n = 1 << 15
    sets = []
    for _ in range(2):
        enc = torch.from_numpy(oracle.encode_to_curve(rng.integers(0, 256, (4096, 32), dtype=np.uint8))).to(dev)
        enc = enc.repeat(n // 4096, 1).contiguous()
        k = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev)
        sets.append((enc, k))
    want = [ctx.scalar_mul_var(e, k) for e, k in sets]
    want_msm = [ctx.msm(e, k) for e, k in sets]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    for rep in range(3):
        got, got_msm = [None, None], [None, None]
        for j in (0, 1):
            with torch.cuda.stream(streams[j]):
                got[j] = ctx.scalar_mul_var(*sets[j])
                got_msm[j] = ctx.msm(*sets[j])
        # and the host-pointer path (the context's private stream) while those are in flight
        h_out, h_st = ctx.scalar_mul_var(sets[0][0][:2048].cpu().numpy(), sets[0][1][:2048].cpu().numpy())
        torch.cuda.synchronize()
        for j in (0, 1):
            assert torch.equal(got[j][0], want[j][0]) and torch.equal(got[j][1], want[j][1])
            assert torch.equal(got_msm[j][0], want_msm[j][0])
        assert _same(h_out, want[0][0][:2048]) and _same(h_st, want[0][1][:2048])
    assert (want[0][0][:256].cpu().numpy() == oracle.scalar_mul_var(sets[0][0][:256].cpu().numpy(),
                                                                    sets[0][1][:256].cpu().numpy())[0]).all()


def test_multi_device_host_path_is_concurrent(oracle):
    """The host-pointer path drives each device of a multi-device context from its own host thread.  With
    D377_DEBUG_DEVICE_DELAY_MS every per-device worker first sleeps: three devices (one GPU listed three
    times) must take about one delay, not three -- the serial loop this replaced took the sum."""
    import time
    import decaf377_amd as d
    rng = np.random.default_rng(706)
    n = 3000
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    c = d.Context([0, 0, 0])
    enc = c.encode_to_curve(r0)                      # warm-up: buffers, code objects
    c.scalar_mul_var(enc, k)
    xyzt, _ = c.decompress(enc)
    c.msm(xyzt, k)
    delay = 0.4
    os.environ["D377_DEBUG_DEVICE_DELAY_MS"] = str(int(delay * 1000))
    try:
        t0 = time.perf_counter()
        out, st = c.scalar_mul_var(enc, k)
        t1 = time.perf_counter()
        e, _, _ = c.msm(xyzt, k)
        t2 = time.perf_counter()
    finally:
        del os.environ["D377_DEBUG_DEVICE_DELAY_MS"]
    assert delay <= t1 - t0 < 2 * delay, t1 - t0
    assert delay <= t2 - t1 < 2 * delay, t2 - t1
    o_out, o_st = oracle.scalar_mul_var(enc, k)
    assert (out == o_out).all() and (st == o_st).all()
    assert bytes(e) == bytes(oracle.msm(xyzt, k)[0])
    c.close()


def test_engine_rejects_malformed_arrays(ctx, torch_mod):
    """The native code reads n * record_size bytes unconditionally; the host mirror must refuse anything
    else before the call: short second operands, wrong row widths, wrong dtypes, short outputs."""
    torch = torch_mod
    rng = np.random.default_rng(707)
    enc = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    xyzt = np.zeros((64, 16), np.uint64)
    with pytest.raises(ValueError):
        ctx.scalar_mul_var(enc, k[:10])
    with pytest.raises(ValueError):
        ctx.roundtrip(enc[:, :16])
    with pytest.raises(ValueError):
        ctx.roundtrip(enc.astype(np.int32))
    with pytest.raises(ValueError):
        ctx.compress(enc)
    with pytest.raises(ValueError):
        ctx.msm(xyzt, k[:5])
    with pytest.raises(ValueError):
        ctx.msm(enc[:, :20], k)
    with pytest.raises(ValueError):
        ctx.roundtrip(enc, outs=[np.zeros((10, 32), np.uint8), np.zeros(64, np.uint8)])
    with pytest.raises(ValueError):
        ctx.fq_op("mul", xyzt[:, :4], xyzt[:5, :4])
    with pytest.raises(ValueError):
        ctx.fq_from_wide_bytes(np.zeros((4, 40), np.uint8))
    dev = torch.device("cuda:0")
    with pytest.raises(ValueError):
        ctx.scalar_mul_var(torch.from_numpy(enc).to(dev), torch.from_numpy(k))       # second operand on the CPU
    with pytest.raises(ValueError):
        ctx.add(torch.zeros((8, 16), dtype=torch.int64, device=dev), torch.zeros((8, 15), dtype=torch.int64, device=dev))


def test_check_invariants_build(oracle):
    """The -DD377_CHECK_INVARIANTS build (libdecaf377_amd_check.so): the curve equation, T Z = X Y and Z != 0
    are re-checked on the device after decompression, the Elligator map, the scalar-multiplication loops and on
    the way into compression -- the reference's debug assertions (src/min_curve/element.rs:104-110,
    src/ark_curve/on_curve.rs:14-39; its CI profile keeps them on).  Valid and invalid inputs through every
    group-level kernel leave the counter at 0 with results equal to the oracle's; records that are not curve
    points, pushed into compress, are counted."""
    import subprocess
    lib = os.path.join(ROOT, "decaf377_amd", "lib", "libdecaf377_amd_check.so")
    assert os.path.exists(lib), "build() makes it next to the product library"
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import decaf377_amd as d
from _oracle import Oracle
orc = Oracle(); ctx = d.Context([0])
on, cnt = ctx.invariant_failures()
assert on and cnt == 0
rng = np.random.default_rng(901); n = 6000
r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8); r1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
enc = ctx.encode_to_curve(r0)
assert (enc == orc.encode_to_curve(r0)).all()
raw = enc.copy(); raw[::5] = rng.integers(0, 256, (len(raw[::5]), 32), dtype=np.uint8)
out, st = ctx.scalar_mul_var(raw, k); o_out, o_st = orc.scalar_mul_var(raw, k)
assert (out == o_out).all() and (st == o_st).all() and st.any()
rt, st = ctx.roundtrip(raw); o_rt, o_st = orc.roundtrip(raw)
assert (rt == o_rt).all() and (st == o_st).all()
assert (ctx.hash_to_curve(r0, r1) == orc.hash_to_curve(r0, r1)).all()     # this build routes every fourth pair through the
                                                                           # exceptional case of the quartic's addition law
assert (ctx.scalar_mul_base(k) == orc.scalar_mul_base(k)).all()
xyzt, st = ctx.decompress(enc)
assert (ctx.compress(ctx.double(ctx.add(xyzt, xyzt[::-1].copy()))) == orc.compress(orc.double_xyzt(orc.add_xyzt(xyzt, xyzt[::-1].copy())))).all()
e, _, _ = ctx.msm(enc[:3000], k[:3000]); assert bytes(e) == bytes(orc.msm(xyzt[:3000], k[:3000])[0])
m = 1000                                                    # the smallest batches' kernels (four elements / one scalar per wave)
assert (ctx.hash_to_curve(r0[:m], r1[:m]) == orc.hash_to_curve(r0[:m], r1[:m])).all()   # exceptional route included (two pairs per wave)
assert (ctx.hash_to_curve(r0[:3 * m], r1[:3 * m]) == orc.hash_to_curve(r0[:3 * m], r1[:3 * m])).all()   # ... and four per wave
assert (ctx.encode_to_curve(r0[:m]) == orc.encode_to_curve(r0[:m])).all()
rt, st = ctx.roundtrip(raw[:m]); o_rt, o_st = orc.roundtrip(raw[:m])
assert (rt == o_rt).all() and (st == o_st).all()
assert (ctx.scalar_mul_base(k[:m]) == orc.scalar_mul_base(k[:m])).all()
assert (ctx.compress(xyzt[:m]) == enc[:m]).all()
assert ctx.invariant_failures() == (True, 0)
bad = xyzt[:100].copy(); bad[:, 0] ^= np.uint64(2)          # x limb flipped: not on the curve any more
ctx.compress(bad)
on, cnt = ctx.invariant_failures()
assert on and cnt == 100, cnt
# Element records whose T is not X Y / Z would sum differently on the MSM's two routes (the quads run on the record's T, the
# buckets rebuild it from X, Y, Z): both routes count them here
tbad = xyzt[:300].copy(); tbad[7, 12] ^= np.uint64(4); tbad[200, 13] ^= np.uint64(1)
with ctx.tuning(msm_small_max=0):
    ctx.msm(tbad, k[:300])
assert ctx.invariant_failures() == (True, 102)
with ctx.tuning(msm_small_max=1000000):
    ctx.msm(tbad, k[:300])
assert ctx.invariant_failures() == (True, 104)
ctx.msm(xyzt[:300], k[:300])
assert ctx.invariant_failures() == (True, 104)
print("INVARIANTS_OK")
"""
    env = dict(os.environ, D377_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "INVARIANTS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    import decaf377_amd as d
    c = d.Context([0])
    assert c.invariant_failures() == (False, 0)         # the product build compiles the checks out
    c.close()


def test_bench_modes_and_rccl_single_rank():
    """bench.py's strong-scaling and from-root modes on the GPU.  (1) One rank under torch.distributed.run with
    the real `nccl` backend (= RCCL): process-group creation on the device, the scatter of HBM input tensors,
    the gather of outputs and the MAX all-reduce of the timings all go through RCCL -- with one rank they
    move nothing between GPUs, but it is the code path the driver's N > 1 runs take, executed on hardware.
    (2) Two ranks sharing GPU 0 over gloo (the collectives stage through the host): strong scaling halves the
    per-rank shard and the from-root line reports its collective share."""
    import json
    import socket
    import subprocess
    def port():
        s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--master-addr", "127.0.0.1"]
    common = ["--steps", "2", "--warmup", "1", "--log2n", "16", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(base + ["--nproc-per-node", "1", "--master-port", str(port()), os.path.join(ROOT, "bench.py"), "--gpus", "1",
                               "--scaling", "strong", "--from-root"] + common, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["scaling"] == "strong" and line["config"]["mode"] == "from-root"
    assert line["config"]["elements_total"] == 1 << 16 and line["collective_ms"] > 0 and line["value"] > 0
    r = subprocess.run(base + ["--nproc-per-node", "2", "--master-port", str(port()), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                               "--scaling", "strong", "--from-root", "--backend", "gloo", "--same-device"] + common,
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["elements_total"] == 1 << 16 and line["config"]["elements_per_gpu"] == 1 << 15
    assert line["collective_ms"] > 0


def test_bench_line_contract():
    """The one JSON line of `python bench.py` (N = 1, small size so that the CPU baseline leg is short): every key of the
    driver's contract, the `roofline` and `cpu_baseline` objects, this round's `ranks_seen` / `parity_sample_ok`, and a
    `roofline_valu` for every extra that is one of BASELINE.json's configurations."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log2n", "16"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d_ = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "ranks_seen", "parity_sample_ok", "extra", "roofline_valu"):
        assert k in d_, k
    assert d_["n_gpus"] == 1 and d_["steps"] == 2 and d_["higher_is_better"] is True and d_["vs_baseline"] is None
    assert d_["dtype"] == "u32" and d_["data"] == "synthetic" and "workload" in d_["config"] and "model" not in d_["config"]
    assert d_["ranks_seen"] == 1 and d_["parity_sample_ok"] is True
    rf = d_["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf
    cb = d_["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["matches_gpu_output"] is True and "sample" in cb
    for name in ("roundtrip", "scalar_mul_base", "encode_to_curve", "sqrt_ratio_zeta", "decompress", "compress", "hash_to_curve"):
        assert 0 < d_["extra"][name]["roofline_valu"]["frac"] < 1, name
    assert abs(d_["value"] - (1 << 16) * 2 / (d_["ms_per_step"] * 2e-3)) / d_["value"] < 1e-6
    small = d_["extra"]["small_batch_2^12_ms_per_call"]
    assert set(small) == {"msm", "msm_encoded", "scalar_mul_var", "scalar_mul_var_element", "scalar_mul_base"} and all(0 < v < 5 for v in small.values())
    tiny = d_["extra"]["tiny_batch_2^8_ms_per_call"]
    assert set(tiny) == {"msm", "msm_encoded", "scalar_mul_var", "scalar_mul_var_element", "scalar_mul_base", "sqrt_ratio_zeta", "decompress",
                         "compress", "encode_to_curve", "hash_to_curve"} and all(0 < v < 5 for v in tiny.values())
    assert set(d_["extra"]["msm_mid_ms_per_call"]) == {"2^16"}


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with NO launcher: the parent starts the two ranks itself before anything touches the
    GPU and relays rank 0's line.  Both ranks share GPU 0 and rendezvous over gloo (this box has one GPU); on a node
    with N GPUs the same command without --same-device / --backend runs one rank per GPU over RCCL.  The line says how
    many ranks the process group had and that every rank's outputs matched the oracle on its own sample."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo",
                        "--steps", "2", "--warmup", "1", "--log2n", "16"],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads(lines[0])
    ctx_msg = lines[0][:4000]                      # (a failed check below shows the line it was made on)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "weak", ctx_msg
    assert line["parity_sample_ok"] is True and line["parity_sample_per_rank"] == 256, ctx_msg
    assert line["config"]["elements_total"] == 2 << 16 and line["value"] > 0, ctx_msg
    # one invocation carries everything a multi-GPU node can give: the weak headline, BASELINE configs[3] as written
    # (2^log2n in total, and from rank 0 with the scatter / gather timed), configs[4] at its total size, the CPU leg
    ex = line["extra"]
    strong, root, ell = ex["strong_2^16_total"], ex["from_root"], ex["encode_to_curve_2^20_total"]
    assert strong["elements_total"] == 1 << 16 and strong["elements_per_rank"] == [1 << 15, 1 << 15], ctx_msg
    assert len(strong["kernel_ms_per_rank"]) == 2 and all(v > 0 for v in strong["kernel_ms_per_rank"]) and strong["value"] > 0, ctx_msg
    assert root["elements_total"] == 1 << 16 and root["collective_ms"] > 0 and root["parity_sample_ok"] and strong["parity_sample_ok"], ctx_msg
    assert root["ms_per_step"] >= root["collective_ms"] and len(root["kernel_ms_per_rank"]) == 2, ctx_msg
    assert ell["elements_total"] == 1 << 16 and len(ell["kernel_ms_per_rank"]) == 2 and ell["parity_sample_ok"] and ell["value"] > 0, ctx_msg
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["matches_gpu_output"] is True and cb["value"] > 0 and cb["cores"] >= 1, ctx_msg
    assert line["roofline"]["kernel"] == "k_scalar_mul_var" and "encodes_per_sec" in ex, ctx_msg


def test_multigpu_selftest_tool():
    """tools/multigpu_selftest.py on whatever this box has (one GPU: every multi-device path runs with GPU 0 listed
    twice and the RCCL leg with one rank; on a multi-GPU node it uses distinct devices and peer copies)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multigpu_selftest.py"), "--log2n", "14"], cwd=ROOT,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "MULTIGPU_SELFTEST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    # the log says what it did not cover, and --require-distinct turns that into a failure: a one-GPU log cannot pass for
    # coverage of the peer copies / cross-device events of d377_batch_sharded_dev and d377_ctx_create
    r2 = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=600)
    if int(r2.stdout.strip().splitlines()[-1]) < 2:
        assert "distinct_device_paths=NOT covered" in r.stdout and "peer_pairs_enabled=0" in r.stdout
        r3 = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multigpu_selftest.py"), "--log2n", "14", "--require-distinct"],
                            cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r3.returncode == 3 and "MULTIGPU_SELFTEST_OK" not in r3.stdout, r3.stdout[-2000:]
        r4 = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multigpu_worker.py"), "--log2n", "10", "--devices", "0,0",
                             "--require-distinct"], cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r4.returncode == 3 and "REQUIRE_DISTINCT_FAILED" in r4.stdout, r4.stdout[-2000:]
    else:
        assert "distinct_device_paths=covered" in r.stdout


def test_rccl_sharding_single_rank():
    """decaf377_amd/sharding.py on HBM tensors over the real `nccl` backend (RCCL): scatter / gather of records,
    the all-gather of MSM partial sums and the timing all-reduce, one rank (the builder's boxes have one GPU; the
    same worker runs unchanged on N ranks)."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_gpu.py")], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_OK world=1" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_structured_invalid_encodings_2_18(ctx, oracle):
    """Decompression's four rejection rules (src/ark_curve/encoding.rs:34-60: high bits set, s >= q, s negative,
    u2*u1^2 not a square) and the valid path, 2^18 records built to hit each rule on purpose rather than by
    chance: valid encodings, their negations q - s, s + q (non-canonical), valid with each of the top three bits
    set, even field elements that are not on the curve, values around q and 2^253.  Status and all four
    coordinates equal the oracle's; scalar multiplication and the round trip zero exactly the rejected records."""
    rng = np.random.default_rng(910)
    base = oracle.encode_to_curve(rng.integers(0, 256, (1 << 15, 32), dtype=np.uint8))
    ints = [int.from_bytes(bytes(x), "little") for x in base[:4096]]
    def pack(vals):
        return np.array([list(int(v % (1 << 256)).to_bytes(32, "little")) for v in vals], dtype=np.uint8)
    groups = [base]
    groups.append(pack([(Q - v) % Q for v in ints]))                     # negative s (odd) or zero
    groups.append(pack([v + Q for v in ints]))                            # same residue, non-canonical
    for bit in (253, 254, 255):
        g = base[:4096].copy()
        g[:, 31] |= np.uint8(1 << (bit - 248))
        groups.append(g)
    even = rng.integers(0, 256, (1 << 15, 32), dtype=np.uint8)
    even[:, 0] &= 0xFE
    even[:, 31] &= 0x0F                                                   # < 2^252 < q, non-negative: ~half are not squares
    groups.append(even)
    groups.append(pack([Q - 2, Q - 1, Q, Q + 1, Q + 2, (1 << 253) - 2, (1 << 253), (1 << 256) - 2, 0, 2, 4, 6]))
    enc = np.concatenate(groups)
    reps = (1 << 18) // enc.shape[0] + 1
    enc = np.tile(enc, (reps, 1))[: 1 << 18]
    enc = enc[rng.permutation(enc.shape[0])]                              # valid and invalid lanes mixed inside every wave
    n = enc.shape[0]
    xyzt, st = ctx.decompress(enc)
    o_out, o_st, _ = oracle.run_threads("roundtrip", enc, None, os.cpu_count() or 4)
    assert (st == o_st).all()
    assert 0.2 < st.mean() < 0.8
    rt, st2 = ctx.roundtrip(enc)
    assert (st2 == o_st).all() and (rt == o_out).all()
    assert not rt[st2 == 1].any() and not xyzt[st == 1].any()
    sample = np.concatenate([np.nonzero(st == 0)[0][:3000], np.nonzero(st == 1)[0][:3000]])
    o_xyzt, o_s = oracle.decompress(enc[sample])
    assert (xyzt[sample] == o_xyzt).all() and (st[sample] == o_s).all()
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    out, st3 = ctx.scalar_mul_var(enc, k)
    o3, o3s, _ = oracle.run_threads("scalar_mul_var", enc[:20000], k[:20000], os.cpu_count() or 4)
    assert (st3 == o_st).all() and (out[:20000] == o3).all() and not out[st3 == 1].any()


def test_dev_calls_capture_into_a_hip_graph(ctx, oracle, torch_mod):
    """Launch-bound use (many small batches): the `_dev` entry points enqueue kernels only -- no host
    synchronisation, no allocation once the workspaces have grown -- so a sequence of them captures into a
    hipGraph (here through torch.cuda.graph, which captures torch's current stream) and replays with new inputs
    written into the same buffers."""
    torch = torch_mod
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(920)
    n = 512
    r0 = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev)
    k = torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).to(dev)
    enc = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    st = torch.empty((n,), dtype=torch.uint8, device=dev)
    # the small sums on both of their kernels: 128 four-term sums (a wave per sum), 5 000 two-term sums (a lane per sum: table scratch)
    n2 = 10000
    r2 = torch.from_numpy(rng.integers(0, 256, (n2, 32), dtype=np.uint8)).to(dev)
    k2 = torch.from_numpy(rng.integers(0, 256, (n2, 32), dtype=np.uint8)).to(dev)
    enc2 = torch.empty((n2, 32), dtype=torch.uint8, device=dev)
    s4, s4x, s4st = (torch.empty(sh, dtype=dt, device=dev) for sh, dt in (((n // 4, 32), torch.uint8), ((n // 4, 16), torch.int64), ((n,), torch.uint8)))
    s2, s2st = torch.empty((n2 // 2, 32), dtype=torch.uint8, device=dev), torch.empty((n2,), dtype=torch.uint8, device=dev)
    ctx.encode_to_curve(r0, outs=[enc]); ctx.scalar_mul_var(enc, k, outs=[out, st]); ctx.msm(enc, k)      # warm up: grow every workspace
    ctx.encode_to_curve(r2, outs=[enc2]); ctx.msm_small(enc2, k2, 2, outs=[s2, s2st])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ctx.encode_to_curve(r0, outs=[enc])
        ctx.scalar_mul_var(enc, k, outs=[out, st])
        m_enc, _, m_st = ctx.msm(enc, k)
        ctx.msm_small(enc, k, 4, outs=[s4, s4x, s4st], elements=True)
        ctx.encode_to_curve(r2, outs=[enc2])
        ctx.msm_small(enc2, k2, 2, outs=[s2, s2st])
    for rep in range(3):
        r0.copy_(torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
        k.copy_(torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)))
        g.replay()
        torch.cuda.synchronize()
        e_h = oracle.encode_to_curve(r0.cpu().numpy())
        assert (enc.cpu().numpy() == e_h).all()
        o_out, o_st = oracle.scalar_mul_var(e_h, k.cpu().numpy())
        assert (out.cpu().numpy() == o_out).all() and not st.cpu().numpy().any()
        xyzt, _ = oracle.decompress(e_h)
        assert bytes(m_enc.cpu().numpy()) == bytes(oracle.msm(xyzt, k.cpu().numpy())[0])
        r2.copy_(torch.from_numpy(rng.integers(0, 256, (n2, 32), dtype=np.uint8)))
        k2.copy_(torch.from_numpy(rng.integers(0, 256, (n2, 32), dtype=np.uint8)))
        g.replay()
        torch.cuda.synchronize()
        got4, got4x, got2 = s4.clone(), s4x.clone(), s2.clone()
        e4, e4st = ctx.msm_small(enc, k, 4)                                        # eager calls on what the replay left in enc / enc2
        e2, e2st = ctx.msm_small(enc2, k2, 2)
        assert torch.equal(got4, e4) and torch.equal(ctx.compress(got4x), e4) and torch.equal(got2, e2)
        assert not s4st.any() and not s2st.any() and not e4st.any() and not e2st.any()
        assert (enc2.cpu().numpy()[:64] == oracle.encode_to_curve(r2.cpu().numpy()[:64])).all()
