"""Big-integer restatement of the decaf377 hot path (TEST INFRASTRUCTURE ONLY).

This is a second, independent statement of the reference algorithms, written
with Python integers so that every step can be read against the Rust source.
It exists to (1) pin the C oracle (`d377_oracle.c`) and (2) regenerate the
fixtures under `tests/golden/`.  Nothing in the shipped package imports it;
only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may.

Every function cites the reference file:line (relative to /root/reference) it
follows.  Values are plain (non-Montgomery) integers mod q unless noted.
"""

# --- fields ----------------------------------------------------------------
# src/fields/fq.rs:29-34 (MODULUS_LIMBS, little-endian u64)
Q = (
    725501752471715841
    | 6461107452199829505 << 64
    | 6968279316240510977 << 128
    | 1345280370688173398 << 192
)
# src/fields/fr.rs:29-34
R_ORDER = (
    13356249993388743167
    | 5950279507993463550 << 64
    | 10965441865914903552 << 128
    | 336320092672043349 << 192
)
MONT_R = 1 << 256  # ark-ff MontBackend<4>: R = 2^256


def from_mont_limbs(limbs):
    """`Fq::from_montgomery_limbs` (src/fields/fq/u64/wrapper.rs:82-85)."""
    v = sum(l << (64 * i) for i, l in enumerate(limbs))
    return v * pow(MONT_R, -1, Q) % Q


def to_mont_limbs(x):
    v = x * MONT_R % Q
    return [(v >> (64 * i)) & (2**64 - 1) for i in range(4)]


# src/min_curve/constants.rs:3-39 (stored there as Montgomery limbs)
ZETA = from_mont_limbs(
    [5947794125541564500, 11292571455564096885, 11814268415718120036, 155746270000486182]
)
COEFF_A = from_mont_limbs(
    [10157024534604021774, 16668528035959406606, 5322190058819395602, 387181115924875961]
)
COEFF_D = from_mont_limbs(
    [15008245758212136496, 17341409599856531410, 648869460136961410, 719771289660577536]
)
COEFF_K = from_mont_limbs(
    [10844245690243005535, 9774967673803681700, 12776203677742963460, 94262208632981673]
)
# src/min_curve/element.rs:61-81 (GENERATOR), src/ark_curve/constants.rs:61-79
B_X = from_mont_limbs(
    [5825153684096051627, 16988948339439369204, 186539475124256708, 1230075515893193738]
)
B_Y = from_mont_limbs(
    [9786171649960077610, 13527783345193426398, 10983305067350511165, 1251302644532346138]
)
B_T = from_mont_limbs(
    [7466800842436274004, 14314110021432015475, 14108125795146788134, 1305086759679105397]
)

# src/ark_curve/constants.rs:30-58
SQRT_N = 47
SQRT_M = 60001509534603559531609739528203892656505753216962260608619555
SQRT_M_MINUS_ONE_DIV_TWO = 30000754767301779765804869764101946328252876608481130304309777
ZETA_TO_ONE_MINUS_M_DIV_TWO = (
    6762755396584113496485389421189479608933826763106393667349575256979972066439
)
SQRT_G = pow(ZETA, SQRT_M, Q)
SQRT_W = 8


def fq_from_le_bytes_mod_order(b):
    """src/fields/fq.rs:90-102 for a 32-byte input: plain reduction mod q."""
    assert len(b) == 32
    return int.from_bytes(b, "little") % Q


def fq_from_bytes_checked(b):
    """src/fields/fq.rs:108-115: None unless the bytes are already < q."""
    v = int.from_bytes(b, "little")
    return v if v < Q else None


def fq_to_bytes(x):
    return int(x % Q).to_bytes(32, "little")


def fr_from_le_bytes_mod_order(b):
    """src/fields/fr.rs:82-94 for a 32-byte input."""
    return int.from_bytes(b, "little") % R_ORDER


def fr_from_bytes_checked(b):
    v = int.from_bytes(b, "little")
    return v if v < R_ORDER else None


def is_negative(x):
    """src/sign.rs:19-23: low bit of the canonical value."""
    return (x % Q) & 1 == 1


def fq_abs(x):
    return (-x) % Q if is_negative(x) else x % Q


# --- sqrt_ratio_zeta (Sarkar 2020 tables) ----------------------------------
class _SqrtTables:
    """src/ark_curve/invsqrt.rs:14-64."""

    def __init__(self):
        self.s_lookup = {}
        for nu in range(256):
            g_pow = pow(SQRT_G, nu << (SQRT_N - SQRT_W), Q)
            self.s_lookup[pow(g_pow, -1, Q)] = nu
        self.gtab = {p: [pow(SQRT_G, nu << p, Q) for nu in range(256)] for p in (0, 8, 16, 24, 32, 40)}
        self.nonsquare_lookup = [1, ZETA_TO_ONE_MINUS_M_DIV_TWO]


_TABLES = None


def sqrt_tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = _SqrtTables()
    return _TABLES


def sqrt_ratio_zeta(num, den):
    """src/ark_curve/invsqrt.rs:75-166, statement by statement."""
    T = sqrt_tables()
    num %= Q
    den %= Q
    if num == 0:
        return True, 0
    if den == 0:
        return False, 0
    s = pow(den, (1 << SQRT_N) - 1, Q)
    t = s * s % Q * den % Q
    w = pow(num * t % Q, SQRT_M_MINUS_ONE_DIV_TWO, Q) * s % Q
    v = w * den % Q
    uv = w * num % Q
    x5 = uv * v % Q
    x4 = pow(x5, 1 << 8, Q)
    x3 = pow(x4, 1 << 8, Q)
    x2 = pow(x3, 1 << 8, Q)
    x1 = pow(x2, 1 << 8, Q)
    x0 = pow(x1, 1 << 7, Q)
    g = T.gtab
    q0p = T.s_lookup[x0]
    t = q0p
    a1 = x1 * g[32][t & 0xFF] % Q
    t += T.s_lookup[a1] << 7
    a2 = x2 * g[24][t & 0xFF] % Q * g[32][(t >> 8) & 0xFF] % Q
    t += T.s_lookup[a2] << 15
    a3 = x3 * g[16][t & 0xFF] % Q * g[24][(t >> 8) & 0xFF] % Q * g[32][(t >> 16) & 0xFF] % Q
    t += T.s_lookup[a3] << 23
    a4 = (
        x4 * g[8][t & 0xFF] % Q * g[16][(t >> 8) & 0xFF] % Q * g[24][(t >> 16) & 0xFF] % Q
        * g[32][(t >> 24) & 0xFF] % Q
    )
    t += T.s_lookup[a4] << 31
    a5 = (
        x5 * g[0][t & 0xFF] % Q * g[8][(t >> 8) & 0xFF] % Q * g[16][(t >> 16) & 0xFF] % Q
        * g[24][(t >> 24) & 0xFF] % Q * g[32][(t >> 32) & 0xFF] % Q
    )
    t += T.s_lookup[a5] << 39
    t = (t + 1) >> 1
    res = (
        uv * T.nonsquare_lookup[q0p & 1] % Q
        * g[0][t & 0xFF] % Q * g[8][(t >> 8) & 0xFF] % Q * g[16][(t >> 16) & 0xFF] % Q
        * g[24][(t >> 24) & 0xFF] % Q * g[32][(t >> 32) & 0xFF] % Q * g[40][(t >> 40) & 0xFF] % Q
    )
    return (q0p & 1) == 0, res


# --- sqrt_ratio_zeta, min_curve backend (constant-time Tonelli-Shanks) ------
# src/fields/fq.rs:62-67: QUADRATIC_NON_RESIDUE_TO_TRACE (Montgomery limbs as written there) = 11^m
QNR_TO_TRACE = from_mont_limbs(
    [4340692304772210610, 11102725085307959083, 15540458298643990566, 944526744080888988]
)


def _our_sqrt(x):
    """src/min_curve/invsqrt.rs:11-57 (c1 = 47, c3 = (m-1)/2, c5 = QNR_TO_TRACE)."""
    z = pow(x, SQRT_M_MINUS_ONE_DIV_TWO, Q)
    t = z * z % Q * x % Q
    z = z * x % Q
    b = t
    c = QNR_TO_TRACE
    for i in range(SQRT_N, 1, -1):
        for _ in range(1, i - 1):
            b = b * b % Q
        if b != 1:
            z = z * c % Q
        c = c * c % Q
        if b != 1:
            t = t * c % Q
        b = t
    return z


def sqrt_ratio_zeta_min_curve(num, den):
    """src/min_curve/invsqrt.rs:73-95 (`non_arkworks_sqrt_ratio_zeta`)."""
    num %= Q
    den %= Q
    if num == 0:
        return True, 0
    if den == 0:
        return False, 0
    x = num * pow(den, -1, Q) % Q
    if pow(x, (Q - 1) // 2, Q) == 1:
        return True, _our_sqrt(x)
    return False, _our_sqrt(ZETA * x % Q)



# --- the raw root, pinned independently of the Sarkar text ------------------
# src/min_curve/constants.rs:10-15: ZETA_TO_TRACE (Montgomery limbs as written there) = zeta^m; the reference never uses it.
ZETA_TO_TRACE = from_mont_limbs(
    [6282505393754313363, 14378628227555923904, 9804873068900332207, 302335131180501866]
)


def _ts_sqrt(x, seed):
    """Tonelli-Shanks in the loop shape of src/min_curve/invsqrt.rs:11-57 with the seed (c5) a parameter."""
    z = pow(x, SQRT_M_MINUS_ONE_DIV_TWO, Q)
    t = z * z % Q * x % Q
    z = z * x % Q
    b = t
    c = seed
    for i in range(SQRT_N, 1, -1):
        for _ in range(1, i - 1):
            b = b * b % Q
        if b != 1:
            z = z * c % Q
        c = c * c % Q
        if b != 1:
            t = t * c % Q
        b = t
    return z


def sqrt_ratio_zeta_ts_zeta(num, den):
    """The arkworks backend's root WITHOUT the Sarkar tables: Tonelli-Shanks seeded with zeta^m on num/den (or on
    zeta num/den for a non-square), by way of a modular inverse and a Legendre symbol -- nothing of
    src/ark_curve/invsqrt.rs:75-166 is used.  Theorem: write x = u^2 with x^m = g^(-2k) in the
    2-Sylow subgroup, g = zeta^m of order 2^47; Tonelli-Shanks with seed g walks the bits of k from the bottom and
    returns x^((m+1)/2) g^k, and so does the Sarkar method, which reads the same k eight bits at a time out of its
    tables (t = 2k there, halved at invsqrt.rs:148).  The two differ from the min_curve backend's root (seed 11^m) by
    a sign on about half of all inputs.  tests/test_oracle.py::test_raw_root_is_tonelli_shanks_with_zeta_seed and
    tests/test_gpu_parity.py::test_raw_root_pinned_by_tonelli_shanks_zeta_seed hold the Sarkar statement above, the C
    oracle and the GPU's D377_SQRT_ROOT_ARK output to this function."""
    num %= Q
    den %= Q
    if num == 0:
        return True, 0
    if den == 0:
        return False, 0
    x = num * pow(den, -1, Q) % Q
    if pow(x, (Q - 1) // 2, Q) == 1:
        return True, _ts_sqrt(x, ZETA_TO_TRACE)
    return False, _ts_sqrt(ZETA * x % Q, ZETA_TO_TRACE)


# --- group (extended twisted Edwards, a=-1, d=3021) ------------------------
IDENTITY = (0, 1, 1, 0)  # src/min_curve/element.rs:53-58 (x, y, z, t)
GENERATOR = (B_X, B_Y, 1, B_T)


def pt_add(p, q_):
    """src/min_curve/element.rs:291-322."""
    x1, y1, z1, t1 = p
    x2, y2, z2, t2 = q_
    a = (y1 - x1) * (y2 - x2) % Q
    b = (y1 + x1) * (y2 + x2) % Q
    c = COEFF_K * t1 % Q * t2 % Q
    d = (z1 + z1) * z2 % Q
    e, f, g, h = (b - a) % Q, (d - c) % Q, (d + c) % Q, (b + a) % Q
    return (e * f % Q, g * h % Q, f * g % Q, e * h % Q)


def pt_double(p):
    """src/min_curve/element.rs:119-136."""
    x, y, z, _ = p
    a = x * x % Q
    b = y * y % Q
    c = 2 * z * z % Q
    d = (-a) % Q
    e = ((x + y) * (x + y) - a - b) % Q
    g = (d + b) % Q
    f = (g - c) % Q
    h = (d - b) % Q
    return (e * f % Q, g * h % Q, f * g % Q, e * h % Q)


def pt_neg(p):
    """src/min_curve/element.rs:324-332."""
    x, y, z, t = p
    return ((-x) % Q, y, z, (-t) % Q)


def pt_eq(p, q_):
    """src/min_curve/element.rs:334-340 / src/ark_curve/element/projective.rs:65-70."""
    return p[0] * q_[1] % Q == q_[0] * p[1] % Q


def pt_on_curve(p):
    """src/min_curve/element.rs:84-98 (new_checked) plus the x*y = t*z invariant."""
    x, y, z, t = p
    return (y * y + COEFF_A * x * x) % Q == (z * z + COEFF_D * t * t) % Q and x * y % Q == t * z % Q


def scalar_mul(p, k):
    """src/min_curve/element.rs:138-157: LSB-first over the 256 bits of 4 LE u64 limbs."""
    acc = IDENTITY
    ins = p
    for i in range(256):
        if (k >> i) & 1:
            acc = pt_add(acc, ins)
        ins = pt_double(ins)
    return acc


def compress_to_field(p):
    """src/ark_curve/encoding.rs:91-114 == src/min_curve/element.rs:163-181."""
    x, y, z, t = p
    a_minus_d = (COEFF_A - COEFF_D) % Q
    u1 = (x + t) * (x - t) % Q
    _, v = sqrt_ratio_zeta(1, u1 * a_minus_d % Q * x % Q * x % Q)
    u2 = fq_abs(v * u1 % Q)
    u3 = (u2 * z - t) % Q
    return fq_abs(a_minus_d * v % Q * u3 % Q * x % Q)


def compress(p):
    """src/ark_curve/encoding.rs:116-128."""
    b = bytearray(fq_to_bytes(compress_to_field(p)))
    b[31] &= 0b00011111
    return bytes(b)


def decompress(enc):
    """src/ark_curve/encoding.rs:32-83 == src/min_curve/element.rs:248-288.
    Returns the extended point or None for InvalidEncoding."""
    assert len(enc) == 32
    if enc[31] >> 5 != 0:
        return None
    s = fq_from_bytes_checked(enc)
    if s is None or is_negative(s):
        return None
    ss = s * s % Q
    u1 = (1 - ss) % Q
    u2 = (u1 * u1 - 4 * COEFF_D * ss) % Q
    was_square, v = sqrt_ratio_zeta(1, u2 * u1 % Q * u1 % Q)
    if not was_square:
        return None
    two_s_u1 = 2 * s * u1 % Q
    if is_negative(two_s_u1 * v % Q):
        v = (-v) % Q
    x = two_s_u1 * v % Q * v % Q * u2 % Q
    y = (1 + ss) * v % Q * u1 % Q
    return (x, y, 1, x * y % Q)


def elligator_map(r0):
    """src/ark_curve/elligator.rs:15-62 == src/min_curve/element.rs:190-230."""
    A, D = COEFF_A, COEFF_D
    r = ZETA * r0 % Q * r0 % Q
    den = (D * r - (D - A)) % Q * (((D - A) * r - D) % Q) % Q
    num = (r + 1) * (A - 2 * D) % Q
    x = num * den % Q
    iss, isri = sqrt_ratio_zeta(1, x)
    if iss:
        sgn, twiddle = 1, 1
    else:
        sgn, twiddle = Q - 1, r0 % Q
    isri = isri * twiddle % Q
    s = isri * num % Q
    t = ((-sgn) * isri % Q * s % Q * (r - 1) % Q * pow((A - 2 * D) % Q, 2, Q) - 1) % Q
    if is_negative(s) == iss:
        s = (-s) % Q
    E = 2 * s % Q
    F = (1 + A * s * s) % Q
    G = (1 - A * s * s) % Q
    H = t
    return (E * H % Q, F * G % Q, F * H % Q, E * G % Q)


def encode_to_curve(r0):
    """src/ark_curve/elligator.rs:74-76."""
    return elligator_map(r0)


def hash_to_curve(r1, r2):
    """src/ark_curve/elligator.rs:67-71."""
    return pt_add(elligator_map(r1), elligator_map(r2))


def affine(p):
    zi = pow(p[2], -1, Q)
    return p[0] * zi % Q, p[1] * zi % Q
