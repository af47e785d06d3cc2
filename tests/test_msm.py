"""vartime_multiscalar_mul: oracle-level checks on CPU, Pippenger kernels on GPU."""
import os

import numpy as np
import pytest

R_ORDER = (13356249993388743167 | 5950279507993463550 << 64 | 10965441865914903552 << 128
           | 336320092672043349 << 192)


def test_oracle_msm_matches_property(oracle):
    """tests/operations.rs:44-60: (a*P) + (b*Q) + (c*R) == vartime_multiscalar_mul([a,b,c],[P,Q,R])."""
    rng = np.random.default_rng(700)
    for n in (0, 1, 3, 17):
        P = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8)) if n else np.zeros((0, 16), np.uint64)
        k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        enc, _ = oracle.msm(P, k, threads=1)
        acc = oracle.decompress(np.zeros((1, 32), np.uint8))[0]
        for i in range(n):
            acc = oracle.add_xyzt(acc, oracle.scalar_mul_xyzt(P[i:i + 1], k[i:i + 1]))
        assert bytes(enc) == bytes(oracle.compress(acc)[0])
    P = oracle.elligator_map_xyzt(rng.integers(0, 256, (600, 32), dtype=np.uint8))
    k = rng.integers(0, 256, (600, 32), dtype=np.uint8)
    assert bytes(oracle.msm(P, k, threads=1)[0]) == bytes(oracle.msm(P, k, threads=4)[0])


@pytest.fixture(scope="module")
def ctx():
    import decaf377_amd as d
    c = d.Context([0])
    yield c
    c.close()


@pytest.mark.gpu
def test_four_lane_group_operations_selftest():
    """tests/cpp/gq_selftest.hip: the cooperative (four lanes per point) doubling and addition of the MSM's Horner
    chains against the single-lane formulas, one at a time and inside loops, compiled and run on the GPU box."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "gq_selftest")
    src = os.path.join(root, "tests", "cpp", "gq_selftest.hip")
    deps = [src] + [os.path.join(root, "decaf377_amd", "csrc", f) for f in os.listdir(os.path.join(root, "decaf377_amd", "csrc"))]
    if not os.path.exists(exe) or any(os.path.getmtime(d_) > os.path.getmtime(exe) for d_ in deps):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "--offload-arch=gfx950", "-std=c++17", src, "-o", exe],
                              timeout=900)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "GQ_SELFTEST_OK" in r.stdout, r.stdout + r.stderr



@pytest.fixture(params=["buckets", "quads", "waves"])
def route(request, ctx):
    """The three routes of the MSM on the same inputs: the bucket method (every size; forced here for the small ones too),
    the one-quad-per-point kernel that batches up to 64 per CU take by default, and the one-wave-per-point kernel
    (lane-spread arithmetic, row_ops.hpp) of batches up to 4 per CU -- each forced here over all the test sizes, also the
    ones where its grid no longer fits the chip at once and the partial sums take several trips."""
    with ctx.tuning(msm_small_max=0 if request.param == "buckets" else 1000000, msm_tiny_max=1000000 if request.param == "waves" else 0):
        yield request.param

@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 2, 7, 15, 16, 17, 64, 257, 1000, 2049, 5000])
def test_msm_matches_oracle(ctx, oracle, n, route):
    rng = np.random.default_rng(701 + n)
    r0 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if n >= 7:
        for i, v in enumerate([0, 1, R_ORDER - 1, R_ORDER, (1 << 256) - 1]):
            k[i] = np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
    P = oracle.elligator_map_xyzt(r0) if n else np.zeros((0, 16), np.uint64)
    enc, xyzt, _ = ctx.msm(P, k)
    o_enc, _ = oracle.msm(P, k)
    assert bytes(enc) == bytes(o_enc)
    assert bytes(ctx.compress(xyzt.reshape(1, 16))[0]) == bytes(o_enc)
    # the Encoding-input form, with two invalid encodings mixed in
    if n >= 7:
        encs = oracle.compress(P)
        encs[3] = 0xFF
        encs[5, 31] = 0x80
        e2, _, st = ctx.msm(encs, k)
        keep = np.ones(n, bool)
        keep[[3, 5]] = False
        assert list(np.nonzero(st)[0]) == [3, 5]
        assert bytes(e2) == bytes(oracle.msm(P[keep], k[keep])[0])


@pytest.mark.gpu
def test_msm_affine_and_projective_inputs_mixed(ctx, oracle, route):
    """Element inputs are normalised to affine records with one inversion per lane, which a wave skips when
    every Z it sees is the canonical 1.  Waves of affine-only points (decompress output), waves of projective
    ones and waves holding both, plus identities in both forms, give the oracle's sum; a record with Z = 0 (not
    a valid Element) counts as the identity."""
    rng = np.random.default_rng(704)
    n = 1000
    proj = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))          # Z != 1
    aff, st = oracle.decompress(oracle.compress(proj))                                       # same points, Z = 1
    assert not st.any() and not np.array_equal(proj[:, 8:12], aff[:, 8:12])
    P = proj.copy()
    P[:256] = aff[:256]                       # four waves that skip the inversion
    P[320:1000:2] = aff[320:1000:2]           # mixed waves
    ident = oracle.identity_xyzt()
    P[300] = ident                                                                           # identity, Z = 1
    P[301] = oracle.add_xyzt(proj[301:302], oracle.neg_xyzt(proj[301:302]))[0]               # identity, Z != 1
    P[10] = ident
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    enc, xyzt, _ = ctx.msm(P, k)
    assert bytes(enc) == bytes(oracle.msm(P, k)[0])
    bad = P.copy()
    bad[500, 8:12] = 0
    keep = np.ones(n, bool)
    keep[500] = False
    assert bytes(ctx.msm(bad, k)[0]) == bytes(oracle.msm(P[keep], k[keep])[0])


@pytest.mark.gpu
@pytest.mark.parametrize("n", [700, 5000, 70000])
def test_msm_equal_and_few_distinct_scalars(ctx, oracle, n, route):
    """Scalars that are all equal (coefficients 1, a common weight) or take a few values put most points of a window
    into one bucket: the bucket reduction has to stay a tree (k_msm_reduce levels), and the sum has to stay right.
    n = 70000 crosses into the third level."""
    rng = np.random.default_rng(705 + n)
    P = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    one = np.zeros((n, 32), np.uint8)
    one[:, 0] = 1
    total = P[0:1]
    for i in range(1, min(n, 5000)):
        total = oracle.add_xyzt(total, P[i:i + 1])
    if n <= 5000:
        assert bytes(ctx.msm(P, one)[0]) == bytes(oracle.compress(total)[0])                  # the plain sum
    k1 = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    same = np.repeat(k1, n, axis=0)
    want = ctx.scalar_mul_var(ctx.msm(P, one)[0].reshape(1, 32), k1)[0][0]                     # k * sum P_i
    assert bytes(ctx.msm(P, same)[0]) == bytes(want)
    few = np.repeat(rng.integers(0, 256, (3, 32), dtype=np.uint8), (n + 2) // 3, axis=0)[:n]
    if n <= 5000:
        assert bytes(ctx.msm(P, few)[0]) == bytes(oracle.msm(P, few)[0])
    else:
        parts = [ctx.msm(P[i:i + 10000], few[i:i + 10000])[1].reshape(1, 16) for i in range(0, n, 10000)]
        acc = parts[0]
        for q in parts[1:]:
            acc = ctx.add(acc, q)
        assert bytes(ctx.msm(P, few)[0]) == bytes(ctx.compress(acc)[0])


@pytest.mark.gpu
def test_msm_on_elements_from_every_producer(ctx, oracle):
    """The small-batch route reads an Element's T (each quad runs the extended-coordinate chain on the record as given); the
    bucket route rebuilds it from X, Y, Z.  Elements as every Element-producing entry point of this library leaves them --
    Z = 1 or not, whatever representative the schedule produced -- must give the same sum on both routes, and that sum must
    be the one of their encodings (the Encoding-input form, and the oracle on a prefix)."""
    rng = np.random.default_rng(707)
    n = 1500
    k1, k2, kk = (rng.integers(0, 256, (n, 32), dtype=np.uint8) for _ in range(3))
    base = ctx.scalar_mul_base_element(k1)
    prod = {
        "scalar_mul_base_element": base,
        "scalar_mul_var_element": ctx.scalar_mul_var_element(base, k2),
        "encode_to_curve_element": ctx.encode_to_curve_element(k1),
        "hash_to_curve_element": ctx.hash_to_curve_element(k1, k2),
        "decompress": ctx.decompress(ctx.scalar_mul_base(k2))[0],
    }
    prod["add"] = ctx.add(prod["scalar_mul_var_element"], prod["encode_to_curve_element"])
    prod["sub"] = ctx.sub(prod["hash_to_curve_element"], base)
    prod["double"] = ctx.double(prod["add"])
    prod["neg"] = ctx.neg(prod["double"])
    for name, P in prod.items():
        with ctx.tuning(msm_small_max=1000000):
            quads = ctx.msm(P, kk)
        with ctx.tuning(msm_small_max=0):
            buckets = ctx.msm(P, kk)
            assert bytes(quads[0]) == bytes(buckets[0]), name
            assert bytes(ctx.compress(quads[1].reshape(1, 16))[0]) == bytes(quads[0]), name
            encs = ctx.compress(P)
            assert bytes(ctx.msm(encs, kk)[0]) == bytes(quads[0]), name
        with ctx.tuning(msm_small_max=1000000):
            assert bytes(ctx.msm(encs, kk)[0]) == bytes(quads[0]), name
            assert bytes(ctx.msm(P[:40], kk[:40])[0]) == bytes(oracle.msm(np.asarray(P[:40]), kk[:40])[0]), name


@pytest.mark.gpu
@pytest.mark.parametrize("window", [4, 5, 6, 7, 9, 12, 13, 14, 15, 16, 17, 18])
def test_msm_every_window_width(ctx, oracle, window):
    """Same inputs through different bucket widths (developer override) give the same bytes.  17 and 18 bits: int32 digits,
    the counting pass in 2 / 4 parts of the bucket range, 513 / 1 025 super-buckets, a middle level in the tree of bit-sums."""
    c = ctx
    with ctx.tuning(msm_window=window, msm_small_max=0):       # 3000 points would not reach the buckets otherwise
        rng = np.random.default_rng(702)
        n = 3000
        P = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))
        k = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        k[0] = np.frombuffer((R_ORDER - 1).to_bytes(32, "little"), np.uint8)
        k[1] = np.frombuffer(int("7" * 62, 16).to_bytes(32, "little"), np.uint8)
        k[2] = np.frombuffer(int("8" * 62, 16).to_bytes(32, "little"), np.uint8)
        enc, _, _ = c.msm(P, k)
        assert bytes(enc) == bytes(oracle.msm(P, k)[0])


@pytest.mark.gpu
def test_msm_sort_forms_and_span_lengths_agree(ctx, oracle):
    """The counting sort's level-1 entries packed into one word (up to 2^24 points) or as a word and a byte (beyond), and the
    span sums with a forced number of entries per lane (1, 3, 8, 57, 128: spans that end inside, at and across bucket and window
    boundaries; the built-in rule is entries / resident lanes), with random, equal and short scalars: same bytes as the
    oracle's fold."""
    rng = np.random.default_rng(7051)
    n = 20000
    P = oracle.elligator_map_xyzt(rng.integers(0, 256, (n, 32), dtype=np.uint8))
    ks = {"random": rng.integers(0, 256, (n, 32), dtype=np.uint8)}
    ks["equal"] = np.tile(ks["random"][:1], (n, 1))
    short = ks["random"].copy()
    short[:, 8:] = 0                                              # 64-bit scalars: the upper windows hold nothing
    ks["short"] = short
    few = np.zeros((n, 32), np.uint8)
    few[:, 0] = rng.integers(0, 3, n)                             # scalars 0, 1, 2: three buckets of one window, most entries dropped
    ks["few"] = few
    for name, k in ks.items():
        want = bytes(oracle.msm(P, k, threads=8)[0])
        with ctx.tuning(msm_small_max=0):
            assert bytes(ctx.msm(P, k)[0]) == want, name
            with ctx.tuning(msm_sort_packed=0):
                assert bytes(ctx.msm(P, k)[0]) == want, name
            for L in (1, 3, 8, 57, 128):
                with ctx.tuning(msm_seg=L):
                    assert bytes(ctx.msm(P, k)[0]) == want, (name, L)
            with ctx.tuning(msm_window=16, msm_seg=5):
                assert bytes(ctx.msm(P, k)[0]) == want, name
            for wide in (17, 18):                                 # the wide windows' own kernels, Elements and Encodings, both sort forms
                with ctx.tuning(msm_window=wide):
                    assert bytes(ctx.msm(P, k)[0]) == want, (name, wide)
                    e2, _, st2 = ctx.msm(oracle.compress(P), k)
                    assert bytes(e2) == want and not st2.any(), (name, wide)
                    with ctx.tuning(msm_sort_packed=0, msm_seg=3):
                        assert bytes(ctx.msm(P, k)[0]) == want, (name, wide)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [20, 22, 23, 24])
def test_msm_full_size_properties(ctx, oracle, log_n):
    """2^20, 2^22 (the advertised size), 2^23 and 2^24 points on the device path (14-bit windows below 3 x 2^20 points, 16-bit
    from there; above 2^24 points the sort's level-1 entries are no longer packed: not reached here): (1) all points equal P ->
    [sum k_i] P; (2) linearity: MSM(A u B) == MSM(A) + MSM(B); (3) a 2^14 prefix against the oracle fold."""
    import torch
    dev = torch.device("cuda:0")
    n = 1 << log_n
    g = torch.Generator(device=dev).manual_seed(703)
    r0 = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    encs = ctx.encode_to_curve(r0)
    P, st = ctx.decompress(encs)
    full, fx, _ = ctx.msm(P, k)
    h = n // 3
    a, ax, _ = ctx.msm(P[:h], k[:h])
    b, bx, _ = ctx.msm(P[h:], k[h:])
    s = ctx.compress(ctx.add(ax.reshape(1, 16), bx.reshape(1, 16)))
    torch.cuda.synchronize()
    assert torch.equal(s[0], full)
    # Encoding input gives the same sum
    full2, _, st2 = ctx.msm(encs, k)
    assert torch.equal(full2, full) and int(st2.sum().item()) == 0
    # all points equal
    m = 1 << 16
    same = P[7:8].expand(m, 16).contiguous()
    e1, _, _ = ctx.msm(same, k[:m])
    ksum = sum(int.from_bytes(bytes(x), "little") % R_ORDER for x in k[:m].cpu().numpy()) % R_ORDER
    kb = torch.from_numpy(np.frombuffer(ksum.to_bytes(32, "little"), np.uint8).copy()).reshape(1, 32).to(dev)
    e2, _ = ctx.scalar_mul_var(encs[7:8], kb)
    assert torch.equal(e1, e2[0])
    # prefix against the oracle
    q = 1 << 14
    e3, _, _ = ctx.msm(P[:q], k[:q])
    assert bytes(e3.cpu().numpy()) == bytes(oracle.msm(P[:q].cpu().numpy().view(np.uint64), k[:q].cpu().numpy(), threads=16)[0])


# ---- many small sums at once: d377_batch_msm_small (batch_msm.hip) -----------------------------------------------------------
def _fold_sums(oracle, P, k, m, threads=8):
    """The reference's fold (src/ark_curve/element/projective.rs:99-117) per sum, on the oracle: Elements P [n m, 16], scalars
    k [n m, 32] -> encodings [n, 32].  The products through the threaded scalar multiplication (on Encodings), the sums on
    Element records."""
    enc = oracle.compress(P)
    prod_enc, st, _ = oracle.run_threads("scalar_mul_var", enc, k, threads)
    assert not st.any()
    prod = oracle.decompress(prod_enc)[0]
    acc = prod[0::m].copy()
    for j in range(1, m):
        acc = oracle.add_xyzt(acc, prod[j::m])
    return oracle.compress(acc)


def _check_element_records(oracle, xyzt, want_enc, sample=48):
    """The Element form of the sums (what the reference's function returns): every record encodes to the sum's Encoding, and
    a sample is checked as the crate checks an Element (src/min_curve/element.rs:84-110): limbs below q, on the curve,
    T Z = X Y."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import d377_model as mod
    xyzt = np.ascontiguousarray(xyzt).view(np.uint64).reshape(-1, 16)
    assert (oracle.compress(xyzt) == want_enc).all()
    for i in np.unique(np.linspace(0, len(xyzt) - 1, min(sample, len(xyzt))).astype(int)) if len(xyzt) else []:
        raw = [sum(int(l) << (64 * j) for j, l in enumerate(xyzt[i, 4 * c:4 * c + 4])) for c in range(4)]
        assert all(v < mod.Q for v in raw), i
        x, y, z, t = (mod.from_mont_limbs([int(l) for l in xyzt[i, 4 * c:4 * c + 4]]) for c in range(4))
        assert z != 0 and mod.pt_on_curve((x, y, z, t)), i


@pytest.fixture(params=["default", "lanes", "waves"])
def small_route(request, ctx):
    """d377_batch_msm_small's two kernels on the same inputs: a wave per sum (batches up to one sum per SIMD by default) and a
    lane per sum (beyond) -- each also forced over all the test sizes."""
    if request.param == "default":
        yield request.param
    else:
        with ctx.tuning(tiny_max=0 if request.param == "lanes" else 1 << 20):
            yield request.param


@pytest.mark.gpu
@pytest.mark.parametrize("m", [1, 2, 3, 5, 8])
def test_batch_msm_small_matches_oracle_fold(ctx, oracle, m, small_route):
    """tests/operations.rs:44-60 at scale: n independent m-term sums, Element and Encoding inputs, bit-exact against the
    oracle's fold of its own products.  Mixed in: scalars 0, 1, r - 1, r, 2^256 - 1; the identity; projective Elements (Z != 1:
    outputs of a scalar multiplication); a record with Z = 0 (the identity by contract); invalid Encodings (reported, left out)."""
    rng = np.random.default_rng(7100 + m)
    for n in (1, 3, 256, 1025, 3000):
        terms = n * m
        P = oracle.elligator_map_xyzt(rng.integers(0, 256, (terms, 32), dtype=np.uint8))
        k = rng.integers(0, 256, (terms, 32), dtype=np.uint8)
        for i, v in enumerate([0, 1, R_ORDER - 1, R_ORDER, (1 << 256) - 1]):
            if i < terms:
                k[(i * 7) % terms] = np.frombuffer(int(v).to_bytes(32, "little"), np.uint8)
        if terms >= 8:
            P[5] = oracle.decompress(np.zeros((1, 32), np.uint8))[0][0]            # the identity as a term
            P[6:8] = oracle.scalar_mul_xyzt(P[6:8], k[0:2])                         # projective representatives
        want = _fold_sums(oracle, P, k, m)
        got = ctx.msm_small(P, k, m)
        assert got.shape == (n, 32) and (got == want).all(), (m, n, small_route, np.nonzero((got != want).any(axis=1))[0][:8])
        got2, got_x = ctx.msm_small(P, k, m, elements=True)                        # the sums as Elements too
        assert (got2 == want).all() and got_x.shape == (n, 16)
        _check_element_records(oracle, got_x, want)
        # Encoding input: the same sums; then with invalid encodings, which drop out of their sums
        encs = oracle.compress(P)
        got_e, st = ctx.msm_small(encs, k, m)
        assert not st.any() and (got_e == want).all(), (m, n, small_route)
        if terms >= 8:
            bad = encs.copy()
            bad[3] = 0xFF
            bad[terms - 1, 31] |= 0x80
            ident = oracle.decompress(np.zeros((1, 32), np.uint8))[0][0]
            Pref = P.copy()
            Pref[3] = ident
            Pref[terms - 1] = ident
            want_b = _fold_sums(oracle, Pref, k, m)
            Pz = Pref.copy()
            Pz[3] = 0                                                               # Z = 0: no group element, counts as the identity
            got_b, got_bx, st = ctx.msm_small(bad, k, m, elements=True)
            assert list(np.nonzero(st)[0]) == [3, terms - 1] and (got_b == want_b).all(), (m, n, small_route)
            _check_element_records(oracle, got_bx, want_b)
            assert (ctx.msm_small(Pz, k, m) == want_b).all(), (m, n, small_route)


@pytest.mark.gpu
def test_batch_msm_small_device_path_and_errors(ctx, oracle):
    """Device tensors (the `_dev` entry points), in order on a stream; argument errors are errors."""
    import torch
    import decaf377_amd._native as nat
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7200)
    m, n = 3, 2000
    P = oracle.elligator_map_xyzt(rng.integers(0, 256, (n * m, 32), dtype=np.uint8))
    k = rng.integers(0, 256, (n * m, 32), dtype=np.uint8)
    want = _fold_sums(oracle, P, k, m)
    Pd = torch.from_numpy(P.view(np.int64)).to(dev)
    kd = torch.from_numpy(k).to(dev)
    out = ctx.msm_small(Pd, kd, m)
    assert (out.cpu().numpy() == want).all()
    enc_d = ctx.compress(Pd)
    out_e, st = ctx.msm_small(enc_d, kd, m)
    assert (out_e.cpu().numpy() == want).all() and not st.cpu().numpy().any()
    out2, out_x = ctx.msm_small(Pd, kd, m, elements=True)
    out3, out_ex, st = ctx.msm_small(enc_d, kd, m, elements=True)
    assert torch.equal(out2, out) and torch.equal(out3, out) and not st.cpu().numpy().any()
    assert torch.equal(ctx.compress(out_x), out) and bool(ctx.eq(out_x, out_ex).all())
    _check_element_records(oracle, out_x.cpu().numpy(), want)
    with pytest.raises(ValueError):
        ctx.msm_small(Pd, kd, m, outs=[out], elements=True)                         # two output arrays expected
    with pytest.raises(nat.NativeError):
        ctx.msm_small(P[: 9 * 4], k[: 9 * 4], 9)                                    # more than D377_BATCH_MSM_MAX_TERMS terms
    with pytest.raises(ValueError):
        ctx.msm_small(P[:10], k[:10], 3)                                            # not a whole number of sums
    assert ctx.msm_small(P[:0], k[:0], 3).shape == (0, 32)


@pytest.mark.gpu
@pytest.mark.parametrize("m,log_n", [(3, 20), (2, 18), (8, 17)])
def test_batch_msm_small_full_size(ctx, oracle, m, log_n):
    """2^20 three-term sums (and 2^18 pairs, 2^17 eight-term sums) on the device path: equal to the composition the call
    replaces -- m scalar-multiplication batches on Elements, m - 1 addition batches, a compression -- on every sum, and to the
    oracle's fold on a 2 048-sum sample; the Encoding form gives the same bytes."""
    import torch
    dev = torch.device("cuda:0")
    n = 1 << log_n
    g = torch.Generator(device=dev).manual_seed(7300 + m)
    r0 = torch.randint(0, 256, (n * m, 32), dtype=torch.uint8, device=dev, generator=g)
    k = torch.randint(0, 256, (n * m, 32), dtype=torch.uint8, device=dev, generator=g)
    encs = ctx.encode_to_curve(r0)
    P, _ = ctx.decompress(encs)
    out = ctx.msm_small(P, k, m)
    prod = ctx.scalar_mul_var_element(P, k)
    acc = prod[0::m].contiguous()
    for j in range(1, m):
        acc = ctx.add(acc, prod[j::m].contiguous())
    ref = ctx.compress(acc)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    out_e, out_ex, st = ctx.msm_small(encs, k, m, elements=True)
    assert torch.equal(out_e, ref) and int(st.sum().item()) == 0
    assert torch.equal(ctx.compress(out_ex), ref) and bool(ctx.eq(out_ex, acc).all())   # the Element form: the same group elements
    q = 2048
    want = _fold_sums(oracle, P[: q * m].cpu().numpy().view(np.uint64), k[: q * m].cpu().numpy(), m, threads=16)
    assert (out[:q].cpu().numpy() == want).all()
