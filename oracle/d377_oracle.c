/*
 * d377_oracle.c -- CPU restatement of the decaf377 hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load the library built from it.  The shipped
 * package (decaf377_amd/) never links, imports or calls it.
 *
 * What it restates (paths relative to /root/reference, crate decaf377 v0.10.1):
 *   - Fq arithmetic of the default "u64" backend: ark-ff 0.4 MontBackend<4>
 *     (third-party, un-vendored, `ark-ff ^0.4`, no lockfile) -- 4 x u64 Montgomery
 *     limbs, R = 2^256, fully reduced outputs.  Call sites:
 *     src/fields/fq/u64/wrapper.rs:45-132.
 *   - Fq::sqrt_ratio_zeta, Sarkar-2020 table method: src/ark_curve/invsqrt.rs:14-166,
 *     constants src/ark_curve/constants.rs:20-58.
 *   - Fq::non_arkworks_sqrt_ratio_zeta, the min_curve backend's constant-time Tonelli-Shanks
 *     (seed 11^m): src/min_curve/invsqrt.rs:11-95, src/fields/fq.rs:62-67.
 *   - Encoding::vartime_decompress: src/ark_curve/encoding.rs:32-83
 *     (== src/min_curve/element.rs:248-288).
 *   - Element::vartime_compress(_to_field): src/ark_curve/encoding.rs:91-128.
 *   - Element::elligator_map / encode_to_curve / hash_to_curve:
 *     src/ark_curve/elligator.rs:15-76 (== src/min_curve/element.rs:190-244).
 *   - Element add / double / neg / eq / is_identity: src/min_curve/element.rs:113-136,291-340.
 *   - Element * Fr: src/min_curve/ops.rs:89-95 -> element.rs:138-157
 *     (LSB-first, 256 iterations of conditional add + double).
 *   - Fr / Fq byte handling: src/fields/fq.rs:90-115, src/fields/fr.rs:82-107.
 *
 * Parity pinning: checked by tests/test_oracle.py against every golden vector the
 * reference's tests hold for this path (16 basepoint multiples, generator/identity,
 * 8 Elligator KATs, sqrt edge cases, Fq/Fr byte examples, proptest regression seeds)
 * and against the independent big-integer model oracle/d377_model.py.
 * Raw sqrt_ratio_zeta root VALUES are pinned by no reference vector (the reference tests
 * res^2 only, invsqrt.rs:182-202); they are pinned by a theorem instead: the Sarkar root
 * is Tonelli-Shanks seeded with zeta^m (ZETA_TO_TRACE, src/min_curve/constants.rs:10-15)
 * applied to num/den -- an algorithm that shares nothing with invsqrt.rs:75-166
 * (oracle/d377_model.py sqrt_ratio_zeta_ts_zeta).  This file, the big-integer Sarkar
 * statement and the GPU's D377_SQRT_ROOT_ARK output are held to it on 2^12 seeded pairs:
 * tests/test_oracle.py::test_raw_root_is_tonelli_shanks_with_zeta_seed,
 * tests/test_gpu_parity.py::test_raw_root_pinned_by_tonelli_shanks_zeta_seed.
 *
 * Build: make -C oracle   (gcc -O3 -march=x86-64-v3 -fPIC -shared)
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fq;      /* Montgomery form, R = 2^256, value < q */

/* src/fields/fq.rs:29-34 */
static const uint64_t Q[4] = {725501752471715841ULL, 6461107452199829505ULL,
                              6968279316240510977ULL, 1345280370688173398ULL};
/* -q^-1 mod 2^64 (q = 1 mod 2^47, so the low 47 bits are all ones) */
#define Q_INV_NEG 0x0a117fffffffffffULL
/* src/fields/fr.rs:29-34 */
static const uint64_t R_ORDER[4] = {13356249993388743167ULL, 5950279507993463550ULL,
                                    10965441865914903552ULL, 336320092672043349ULL};

/* R mod q = Fq::ONE; R^2 mod q (derived, checked in tests against the model) */
static const fq FQ_ONE = {{0x7d1c7ffffffffff3ULL, 0x7257f50f6ffffff2ULL,
                           0x16d81575512c0feeULL, 0x0d4bda322bbb9a9dULL}};
static const fq FQ_R2 = {{0x25d577bab861857bULL, 0xcc2c27b58860591fULL,
                          0xa7cc008fe5dc8593ULL, 0x011fdae7eff1c939ULL}};
static const fq FQ_ZERO = {{0, 0, 0, 0}};

/* src/min_curve/constants.rs:3-39 (Montgomery limbs as written there) */
static const fq ZETA = {{5947794125541564500ULL, 11292571455564096885ULL,
                         11814268415718120036ULL, 155746270000486182ULL}};
static const fq COEFF_A = {{10157024534604021774ULL, 16668528035959406606ULL,
                            5322190058819395602ULL, 387181115924875961ULL}};
static const fq COEFF_D = {{15008245758212136496ULL, 17341409599856531410ULL,
                            648869460136961410ULL, 719771289660577536ULL}};
static const fq COEFF_K = {{10844245690243005535ULL, 9774967673803681700ULL,
                            12776203677742963460ULL, 94262208632981673ULL}};
/* src/min_curve/element.rs:61-81 */
static const fq B_X = {{5825153684096051627ULL, 16988948339439369204ULL,
                        186539475124256708ULL, 1230075515893193738ULL}};
static const fq B_Y = {{9786171649960077610ULL, 13527783345193426398ULL,
                        10983305067350511165ULL, 1251302644532346138ULL}};
static const fq B_T = {{7466800842436274004ULL, 14314110021432015475ULL,
                        14108125795146788134ULL, 1305086759679105397ULL}};

/* ---------------------------------------------------------------- Fq --- */
static inline int geq_q(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > Q[i]) return 1;
        if (a[i] < Q[i]) return 0;
    }
    return 1;
}
static inline void sub_q(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - Q[i] - br;
        a[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static inline fq fq_add(fq a, fq b) {
    fq r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    /* q < 2^253 so no carry out of 256 bits */
    if (geq_q(r.l)) sub_q(r.l);
    return r;
}
static inline fq fq_sub(fq a, fq b) {
    fq r; u128 br = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - br;
        r.l[i] = (uint64_t)d; br = (d >> 64) & 1;
    }
    if (br) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] + Q[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
static inline fq fq_neg(fq a) { return fq_sub(FQ_ZERO, a); }
static inline int fq_is_zero(fq a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline int fq_eq(fq a, fq b) {
    return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3];
}
/* CIOS Montgomery multiplication, result fully reduced (ark-ff MontBackend::mul_assign contract) */
static inline fq fq_mul(fq a, fq b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c; c >>= 64;
        }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * Q_INV_NEG;
        c = (u128)m * Q[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * Q[j] + t[j];
            t[j - 1] = (uint64_t)c; c >>= 64;
        }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fq r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq_q(r.l)) sub_q(r.l);
    return r;
}
static inline fq fq_square(fq a) { return fq_mul(a, a); }
/* generic MSB-first square-and-multiply, like ark-ff Field::pow over u64 limbs */
static fq fq_pow(fq a, const uint64_t *e, int nlimbs) {
    fq r = FQ_ONE; int started = 0;
    for (int i = nlimbs * 64 - 1; i >= 0; --i) {
        if (started) r = fq_square(r);
        if ((e[i / 64] >> (i % 64)) & 1) { r = started ? fq_mul(r, a) : a; started = 1; }
    }
    return r;
}
static fq fq_inverse(fq a) {   /* a^(q-2); only used for table construction */
    uint64_t e[4] = {Q[0] - 2, Q[1], Q[2], Q[3]};
    return fq_pow(a, e, 4);
}
/* canonical (non-Montgomery) limbs */
static inline void fq_to_canonical(fq a, uint64_t out[4]) {
    fq one = {{1, 0, 0, 0}};
    fq r = fq_mul(a, one);
    memcpy(out, r.l, 32);
}
static inline fq fq_from_canonical(const uint64_t in[4]) { /* in < q */
    fq a; memcpy(a.l, in, 32);
    return fq_mul(a, FQ_R2);
}
static inline void load_le(const uint8_t *b, uint64_t out[4]) {
    for (int i = 0; i < 4; ++i) {
        uint64_t v = 0;
        for (int j = 7; j >= 0; --j) v = (v << 8) | b[8 * i + j];
        out[i] = v;
    }
}
static inline void store_le(const uint64_t in[4], uint8_t *b) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(in[i] >> (8 * j));
}
/* src/fields/fq.rs:90-102 for exactly 32 input bytes: value mod q (value < 2^256 < 14q) */
static fq fq_from_le_bytes_mod_order(const uint8_t *b) {
    uint64_t v[4]; load_le(b, v);
    while (geq_q(v)) sub_q(v);
    return fq_from_canonical(v);
}
/* src/fields/fq.rs:108-115: returns 0 and sets *out when canonical, 1 otherwise */
static int fq_from_bytes_checked(const uint8_t *b, fq *out) {
    uint64_t v[4]; load_le(b, v);
    if (geq_q(v)) return 1;
    *out = fq_from_canonical(v);
    return 0;
}
static inline void fq_to_bytes(fq a, uint8_t *b) { uint64_t c[4]; fq_to_canonical(a, c); store_le(c, b); }
/* src/sign.rs:19-23 */
static inline int fq_is_negative(fq a) { uint64_t c[4]; fq_to_canonical(a, c); return (int)(c[0] & 1); }
static inline fq fq_abs(fq a) { return fq_is_negative(a) ? fq_neg(a) : a; }
static inline fq fq_from_u64(uint64_t x) { uint64_t c[4] = {x, 0, 0, 0}; return fq_from_canonical(c); }

/* ---------------------------------------------------------------- Fr --- */
static inline int geq_r(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] > R_ORDER[i]) return 1;
        if (a[i] < R_ORDER[i]) return 0;
    }
    return 1;
}
/* src/fields/fr.rs:82-94 (32 bytes) followed by to_le_limbs (fr/u64/wrapper.rs:43-59) */
static void fr_from_le_bytes_mod_order(const uint8_t *b, uint64_t out[4]) {
    load_le(b, out);
    while (geq_r(out)) {
        u128 br = 0;
        for (int i = 0; i < 4; ++i) {
            u128 d = (u128)out[i] - R_ORDER[i] - br;
            out[i] = (uint64_t)d; br = (d >> 64) & 1;
        }
    }
}

/* Fr arithmetic on canonical values (src/fields/fr/u64/wrapper.rs:76-108 wraps ark-ff's Fp256; only the values
 * matter here, so this is schoolbook multiplication and a bit-serial reduction -- nothing shared with the GPU's
 * word-level Montgomery form). */
static inline void fr_sub_r(uint64_t a[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - R_ORDER[i] - br; a[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
static void fr_add(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {   /* a, b < r < 2^251 */
    u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a[i] + b[i]; out[i] = (uint64_t)c; c >>= 64; }
    if (geq_r(out)) fr_sub_r(out);
}
static void fr_neg(const uint64_t a[4], uint64_t out[4]) {
    if ((a[0] | a[1] | a[2] | a[3]) == 0) { memset(out, 0, 32); return; }
    u128 br = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)R_ORDER[i] - a[i] - br; out[i] = (uint64_t)d; br = (d >> 64) & 1; }
}
static void fr_sub(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    uint64_t nb[4]; fr_neg(b, nb); fr_add(a, nb, out);
}
static void fr_reduce_bits(const uint64_t *p, int nlimbs, uint64_t out[4]) {     /* p mod r, one bit at a time */
    uint64_t acc[4] = {0, 0, 0, 0};
    for (int bit = 64 * nlimbs - 1; bit >= 0; --bit) {
        acc[3] = (acc[3] << 1) | (acc[2] >> 63); acc[2] = (acc[2] << 1) | (acc[1] >> 63);
        acc[1] = (acc[1] << 1) | (acc[0] >> 63); acc[0] = (acc[0] << 1) | ((p[bit >> 6] >> (bit & 63)) & 1);
        if (geq_r(acc)) fr_sub_r(acc);                                        /* 2 acc + 1 <= 2r - 1 */
    }
    memcpy(out, acc, 32);
}
static void fr_mul(const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    uint64_t p[8] = {0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)a[i] * b[j] + p[i + j]; p[i + j] = (uint64_t)c; c >>= 64; }
        p[i + 4] = (uint64_t)c;
    }
    fr_reduce_bits(p, 8, out);
}
static int fr_inverse(const uint64_t a[4], uint64_t out[4]) {                  /* wrapper.rs:80-86: None for zero */
    if ((a[0] | a[1] | a[2] | a[3]) == 0) { memset(out, 0, 32); return 0; }
    uint64_t e[4] = {R_ORDER[0] - 2, R_ORDER[1], R_ORDER[2], R_ORDER[3]};      /* r is odd and > 2: no borrow */
    uint64_t acc[4] = {1, 0, 0, 0};
    for (int bit = 255; bit >= 0; --bit) {
        fr_mul(acc, acc, acc);
        if ((e[bit >> 6] >> (bit & 63)) & 1) fr_mul(acc, a, acc);
    }
    memcpy(out, acc, 32);
    return 1;
}
/* Fr::from_le_bytes_mod_order for any length, src/fields/fr.rs:82-94: 32-byte chunks, folded from the most
 * significant one with acc * FIELD_SIZE_POWER_OF_TWO + chunk (fr.rs:75-80 = 2^256 mod r) */
static void fr_from_le_bytes_mod_order_any(const uint8_t *b, size_t len, uint64_t out[4]) {
    static const uint64_t two256[5] = {0, 0, 0, 0, 1};
    uint64_t pow2[4], acc[4] = {0, 0, 0, 0};
    fr_reduce_bits(two256, 5, pow2);
    size_t nchunks = (len + 31) / 32;
    for (size_t c = nchunks; c-- > 0;) {
        uint8_t padded[32] = {0};
        size_t l = (c * 32 + 32 <= len) ? 32 : len - c * 32;
        memcpy(padded, b + c * 32, l);
        uint64_t x[4], t[4];
        fr_from_le_bytes_mod_order(padded, x);
        fr_mul(acc, pow2, t);
        fr_add(t, x, acc);
    }
    memcpy(out, acc, 32);
}

/* -------------------------------------------------- sqrt_ratio_zeta --- */
/* src/ark_curve/constants.rs:30-58 */
#define SQRT_N 47
static const uint64_t M_LIMBS[4] = {17149038877957297187ULL, 11113960768935211860ULL,
                                    14608890324369326440ULL, 9558ULL};             /* fq.rs:45-50 TRACE */
static const uint64_t M_MINUS_ONE_DIV_TWO[4] = {8574519438978648593ULL, 5556980384467605930ULL,
                                                7304445162184663220ULL, 4779ULL};  /* fq.rs:52-57 */
typedef struct {
    fq s_key[256];          /* s_lookup keys: (g^(nu*2^(n-w)))^-1, value = index nu */
    fq gtab[6][256];        /* g0,g8,g16,g24,g32,g40 */
    fq nonsquare_lookup[2];
    uint16_t hash[1024];    /* open-addressed index into s_key (0 = empty, else nu+1) */
} sqrt_tables;
static sqrt_tables TAB;
static pthread_once_t tab_once = PTHREAD_ONCE_INIT;

static inline unsigned s_hash(fq k) { return (unsigned)((k.l[0] * 0x9E3779B97F4A7C15ULL) >> 54); }
static void tables_init(void) {   /* src/ark_curve/invsqrt.rs:25-63 */
    fq g = fq_pow(ZETA, M_LIMBS, 4);                 /* G = ZETA^M, constants.rs:53-55 */
    for (uint64_t nu = 0; nu < 256; ++nu) {
        uint64_t e[1] = {nu << (SQRT_N - 8)};
        TAB.s_key[nu] = fq_inverse(fq_pow(g, e, 1));
    }
    static const int powers[6] = {0, 8, 16, 24, 32, 40};
    for (int p = 0; p < 6; ++p)
        for (uint64_t nu = 0; nu < 256; ++nu) {
            uint64_t e[1] = {nu << powers[p]};
            TAB.gtab[p][nu] = fq_pow(g, e, 1);
        }
    TAB.nonsquare_lookup[0] = FQ_ONE;
    /* ZETA^((1-M)/2) = (ZETA^((M-1)/2))^-1, constants.rs:46-51 */
    TAB.nonsquare_lookup[1] = fq_inverse(fq_pow(ZETA, M_MINUS_ONE_DIV_TWO, 4));
    memset(TAB.hash, 0, sizeof TAB.hash);
    for (unsigned nu = 0; nu < 256; ++nu) {
        unsigned h = s_hash(TAB.s_key[nu]);
        while (TAB.hash[h]) h = (h + 1) & 1023;
        TAB.hash[h] = (uint16_t)(nu + 1);
    }
}
static uint64_t s_lookup(fq k) {
    unsigned h = s_hash(k);
    while (TAB.hash[h]) {
        unsigned nu = TAB.hash[h] - 1u;
        if (fq_eq(TAB.s_key[nu], k)) return nu;
        h = (h + 1) & 1023;
    }
    abort();   /* the reference would panic on a missing HashMap key; unreachable for field inputs */
}
static fq sqr_n(fq x, int n) { while (n--) x = fq_square(x); return x; }

/* src/ark_curve/invsqrt.rs:75-166 */
static int fq_sqrt_ratio_zeta(fq num, fq den, fq *res) {
    pthread_once(&tab_once, tables_init);
    if (fq_is_zero(num)) { *res = num; return 1; }
    if (fq_is_zero(den)) { *res = den; return 0; }
    const uint64_t s_exp[1] = {(1ULL << SQRT_N) - 1};
    fq s = fq_pow(den, s_exp, 1);
    fq t_ = fq_mul(fq_square(s), den);
    fq w = fq_mul(fq_pow(fq_mul(num, t_), M_MINUS_ONE_DIV_TWO, 4), s);
    fq v = fq_mul(w, den);
    fq uv = fq_mul(w, num);
    fq x5 = fq_mul(uv, v);
    fq x4 = sqr_n(x5, 8), x3 = sqr_n(x4, 8), x2 = sqr_n(x3, 8), x1 = sqr_n(x2, 8), x0 = sqr_n(x1, 7);
    fq (*g)[256] = TAB.gtab;   /* g[0]=g0, g[1]=g8, g[2]=g16, g[3]=g24, g[4]=g32, g[5]=g40 */
    uint64_t q0p = s_lookup(x0);
    uint64_t t = q0p;
    fq a1 = fq_mul(x1, g[4][t & 0xFF]);
    t += s_lookup(a1) << 7;
    fq a2 = fq_mul(fq_mul(x2, g[3][t & 0xFF]), g[4][(t >> 8) & 0xFF]);
    t += s_lookup(a2) << 15;
    fq a3 = fq_mul(fq_mul(fq_mul(x3, g[2][t & 0xFF]), g[3][(t >> 8) & 0xFF]), g[4][(t >> 16) & 0xFF]);
    t += s_lookup(a3) << 23;
    fq a4 = fq_mul(fq_mul(fq_mul(fq_mul(x4, g[1][t & 0xFF]), g[2][(t >> 8) & 0xFF]),
                          g[3][(t >> 16) & 0xFF]), g[4][(t >> 24) & 0xFF]);
    t += s_lookup(a4) << 31;
    fq a5 = fq_mul(fq_mul(fq_mul(fq_mul(fq_mul(x5, g[0][t & 0xFF]), g[1][(t >> 8) & 0xFF]),
                                 g[2][(t >> 16) & 0xFF]), g[3][(t >> 24) & 0xFF]), g[4][(t >> 32) & 0xFF]);
    t += s_lookup(a5) << 39;
    t = (t + 1) >> 1;
    fq r = fq_mul(uv, TAB.nonsquare_lookup[q0p & 1]);
    r = fq_mul(r, g[0][t & 0xFF]);
    r = fq_mul(r, g[1][(t >> 8) & 0xFF]);
    r = fq_mul(r, g[2][(t >> 16) & 0xFF]);
    r = fq_mul(r, g[3][(t >> 24) & 0xFF]);
    r = fq_mul(r, g[4][(t >> 32) & 0xFF]);
    r = fq_mul(r, g[5][(t >> 40) & 0xFF]);
    *res = r;
    return (q0p & 1) == 0;
}

/* ------------------------------------ min_curve backend: Tonelli-Shanks root --- */
/* src/fields/fq.rs:62-67 QUADRATIC_NON_RESIDUE_TO_TRACE (= 11^m, Montgomery limbs as written there) */
static const fq QNR_TO_TRACE = {{4340692304772210610ULL, 11102725085307959083ULL,
                                 15540458298643990566ULL, 944526744080888988ULL}};
/* src/fields/fq.rs MODULUS_MINUS_ONE_DIV_TWO_LIMBS = (q - 1) / 2 */
static void q_minus_one_div_two(uint64_t e[4]) {
    uint64_t t[4] = {Q[0] - 1, Q[1], Q[2], Q[3]};
    for (int i = 0; i < 4; ++i) e[i] = (t[i] >> 1) | (i < 3 ? t[i + 1] << 63 : 0);
}
/* src/min_curve/invsqrt.rs:59-71 pow_le_limbs: LSB-first square-and-multiply */
static fq fq_pow_le_limbs(fq a, const uint64_t *limbs, int n) {
    fq acc = FQ_ONE, insert = a;
    for (int l = 0; l < n; ++l)
        for (int i = 0; i < 64; ++i) {
            if ((limbs[l] >> i) & 1) acc = fq_mul(acc, insert);
            insert = fq_mul(insert, insert);
        }
    return acc;
}
/* src/min_curve/invsqrt.rs:11-57 our_sqrt: constant-time Tonelli-Shanks of
 * draft-irtf-cfrg-hash-to-curve appendix, c1 = 47, c3 = (m-1)/2, c5 = 11^m */
static fq fq_our_sqrt(fq x) {
    fq z = fq_pow_le_limbs(x, M_MINUS_ONE_DIV_TWO, 4);      /* step 1 */
    fq t = fq_mul(fq_mul(z, z), x);                         /* step 2 */
    z = fq_mul(z, x);                                       /* step 3 */
    fq b = t;                                               /* step 4 */
    fq c = QNR_TO_TRACE;                                    /* step 5 */
    for (int i = SQRT_N; i >= 2; --i) {                     /* step 6 */
        for (int j = 1; j <= i - 2; ++j) b = fq_mul(b, b);  /* steps 7-8 */
        int b_ne_one = !fq_eq(b, FQ_ONE);
        if (b_ne_one) z = fq_mul(z, c);                     /* step 9 (CMOV) */
        c = fq_mul(c, c);                                   /* step 10 */
        if (b_ne_one) t = fq_mul(t, c);                     /* step 11 (CMOV) */
        b = t;                                              /* step 12 */
    }
    return z;
}
/* src/min_curve/invsqrt.rs:73-95 non_arkworks_sqrt_ratio_zeta */
static int fq_sqrt_ratio_zeta_min_curve(fq num, fq den, fq *res) {
    if (fq_is_zero(num)) { *res = num; return 1; }
    if (fq_is_zero(den)) { *res = den; return 0; }
    fq x = fq_mul(num, fq_inverse(den));                    /* num / den: Div = mul by inverse (fq/ops.rs:200-249) */
    uint64_t e[4]; q_minus_one_div_two(e);
    fq symbol = fq_pow_le_limbs(x, e, 4);
    if (fq_eq(symbol, FQ_ONE)) { *res = fq_our_sqrt(x); return 1; }
    *res = fq_our_sqrt(fq_mul(ZETA, x));
    return 0;
}

/* -------------------------------------------------------------- group --- */
typedef struct { fq x, y, z, t; } element;

static element el_identity(void) { element e = {FQ_ZERO, FQ_ONE, FQ_ONE, FQ_ZERO}; return e; }
static element el_generator(void) { element e = {B_X, B_Y, FQ_ONE, B_T}; return e; }

/* src/min_curve/element.rs:291-322 */
static element el_add(element p, element q) {
    fq a = fq_mul(fq_sub(p.y, p.x), fq_sub(q.y, q.x));
    fq b = fq_mul(fq_add(p.y, p.x), fq_add(q.y, q.x));
    fq c = fq_mul(fq_mul(COEFF_K, p.t), q.t);
    fq d = fq_mul(fq_add(p.z, p.z), q.z);
    fq e = fq_sub(b, a), f = fq_sub(d, c), g = fq_add(d, c), h = fq_add(b, a);
    element r = {fq_mul(e, f), fq_mul(g, h), fq_mul(f, g), fq_mul(e, h)};
    return r;
}
/* src/min_curve/element.rs:119-136 */
static element el_double(element p) {
    fq a = fq_square(p.x), b = fq_square(p.y), c = fq_square(p.z);
    c = fq_add(c, c);
    fq d = fq_neg(a);
    fq e = fq_sub(fq_sub(fq_square(fq_add(p.x, p.y)), a), b);
    fq g = fq_add(d, b), f = fq_sub(g, c), h = fq_sub(d, b);
    element r = {fq_mul(e, f), fq_mul(g, h), fq_mul(f, g), fq_mul(e, h)};
    return r;
}
/* src/min_curve/element.rs:138-157, vartime variant */
static element el_scalar_mul(element p, const uint64_t k[4]) {
    element acc = el_identity(), ins = p;
    for (int l = 0; l < 4; ++l)
        for (int i = 0; i < 64; ++i) {
            if ((k[l] >> i) & 1) acc = el_add(acc, ins);
            ins = el_double(ins);
        }
    return acc;
}
/* src/ark_curve/encoding.rs:91-114 */
static fq el_compress_to_field(element p) {
    fq a_minus_d = fq_sub(COEFF_A, COEFF_D);
    fq u1 = fq_mul(fq_add(p.x, p.t), fq_sub(p.x, p.t));
    fq v;
    (void)fq_sqrt_ratio_zeta(FQ_ONE, fq_mul(fq_mul(u1, a_minus_d), fq_square(p.x)), &v);
    fq u2 = fq_abs(fq_mul(v, u1));
    fq u3 = fq_sub(fq_mul(u2, p.z), p.t);
    return fq_abs(fq_mul(fq_mul(fq_mul(a_minus_d, v), u3), p.x));
}
/* src/ark_curve/encoding.rs:116-128 */
static void el_compress(element p, uint8_t out[32]) {
    fq_to_bytes(el_compress_to_field(p), out);
    out[31] &= 0x1f;
}
/* src/ark_curve/encoding.rs:32-83; returns 0 ok, 1 InvalidEncoding */
static int el_decompress(const uint8_t enc[32], element *out) {
    if (enc[31] >> 5) return 1;
    fq s;
    if (fq_from_bytes_checked(enc, &s)) return 1;
    if (fq_is_negative(s)) return 1;
    fq d4 = fq_mul(COEFF_D, fq_from_u64(4));
    fq ss = fq_square(s);
    fq u1 = fq_sub(FQ_ONE, ss);
    fq u2 = fq_sub(fq_square(u1), fq_mul(d4, ss));
    fq v;
    if (!fq_sqrt_ratio_zeta(FQ_ONE, fq_mul(u2, fq_square(u1)), &v)) return 1;
    fq two = fq_add(FQ_ONE, FQ_ONE);
    fq two_s_u1 = fq_mul(fq_mul(two, s), u1);
    if (fq_is_negative(fq_mul(two_s_u1, v))) v = fq_neg(v);
    out->x = fq_mul(fq_mul(two_s_u1, fq_square(v)), u2);
    out->y = fq_mul(fq_mul(fq_add(FQ_ONE, ss), v), u1);
    out->z = FQ_ONE;
    out->t = fq_mul(out->x, out->y);
    return 0;
}
/* src/ark_curve/elligator.rs:15-62 */
static element el_elligator_map(fq r0) {
    fq A = COEFF_A, D = COEFF_D, one = FQ_ONE, two = fq_add(FQ_ONE, FQ_ONE);
    fq r = fq_mul(ZETA, fq_square(r0));
    fq den = fq_mul(fq_sub(fq_mul(D, r), fq_sub(D, A)), fq_sub(fq_mul(fq_sub(D, A), r), D));
    fq a_2d = fq_sub(A, fq_mul(two, D));
    fq num = fq_mul(fq_add(r, one), a_2d);
    fq x = fq_mul(num, den);
    fq isri;
    int iss = fq_sqrt_ratio_zeta(one, x, &isri);
    fq sgn, twiddle;
    if (iss) { sgn = one; twiddle = one; } else { sgn = fq_neg(one); twiddle = r0; }
    isri = fq_mul(isri, twiddle);
    fq s = fq_mul(isri, num);
    fq t = fq_sub(fq_mul(fq_mul(fq_mul(fq_mul(fq_neg(sgn), isri), s), fq_sub(r, one)), fq_square(a_2d)), one);
    if (fq_is_negative(s) == iss) s = fq_neg(s);
    fq E = fq_mul(two, s);
    fq F = fq_add(one, fq_mul(A, fq_square(s)));
    fq G = fq_sub(one, fq_mul(A, fq_square(s)));
    fq H = t;
    element e = {fq_mul(E, H), fq_mul(F, G), fq_mul(F, H), fq_mul(E, G)};   /* (x, y, z, t) */
    return e;
}

/* ------------------------------------------------- batch entry points --- */
/* Same per-element contracts as include/decaf377_amd.h, plain loops on the CPU. */
#define API __attribute__((visibility("default")))

API void d377o_init(void) { pthread_once(&tab_once, tables_init); }

API void d377o_fq_mul_mont(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; ++i) {
        fq x, y; memcpy(x.l, a + 4 * i, 32); memcpy(y.l, b + 4 * i, 32);
        fq r = fq_mul(x, y); memcpy(out + 4 * i, r.l, 32);
    }
}
/* bytes (mod order) -> Montgomery limbs, and back to canonical bytes */
API void d377o_fq_from_bytes_mod_order(const uint8_t *in, size_t n, uint64_t *mont) {
    for (size_t i = 0; i < n; ++i) { fq r = fq_from_le_bytes_mod_order(in + 32 * i); memcpy(mont + 4 * i, r.l, 32); }
}
API void d377o_fq_to_bytes(const uint64_t *mont, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) { fq a; memcpy(a.l, mont + 4 * i, 32); fq_to_bytes(a, out + 32 * i); }
}
API void d377o_fq_from_bytes_checked(const uint8_t *in, size_t n, uint64_t *mont, uint8_t *status) {
    for (size_t i = 0; i < n; ++i) {
        fq r = FQ_ZERO; status[i] = (uint8_t)fq_from_bytes_checked(in + 32 * i, &r);
        memcpy(mont + 4 * i, r.l, 32);
    }
}
API void d377o_fr_from_bytes_mod_order(const uint8_t *in, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) { uint64_t k[4]; fr_from_le_bytes_mod_order(in + 32 * i, k); store_le(k, out + 32 * i); }
}
API void d377o_fr_from_bytes_checked(const uint8_t *in, size_t n, uint8_t *status) {
    for (size_t i = 0; i < n; ++i) { uint64_t k[4]; load_le(in + 32 * i, k); status[i] = (uint8_t)geq_r(k); }
}
API void d377o_sqrt_ratio_zeta(const uint8_t *num32, const uint8_t *den32, size_t n,
                               uint8_t *root32, uint8_t *was_square) {
    for (size_t i = 0; i < n; ++i) {
        fq r;
        int ws = fq_sqrt_ratio_zeta(fq_from_le_bytes_mod_order(num32 + 32 * i),
                                    fq_from_le_bytes_mod_order(den32 + 32 * i), &r);
        fq_to_bytes(r, root32 + 32 * i);
        was_square[i] = (uint8_t)ws;
    }
}
static void el_store(element e, uint64_t *xyzt) {
    memcpy(xyzt, e.x.l, 32); memcpy(xyzt + 4, e.y.l, 32); memcpy(xyzt + 8, e.z.l, 32); memcpy(xyzt + 12, e.t.l, 32);
}
static element el_load(const uint64_t *xyzt) {
    element e;
    memcpy(e.x.l, xyzt, 32); memcpy(e.y.l, xyzt + 4, 32); memcpy(e.z.l, xyzt + 8, 32); memcpy(e.t.l, xyzt + 12, 32);
    return e;
}
API void d377o_decompress(const uint8_t *enc32, size_t n, uint64_t *xyzt, uint8_t *status) {
    for (size_t i = 0; i < n; ++i) {
        element e; int st = el_decompress(enc32 + 32 * i, &e);
        status[i] = (uint8_t)st;
        if (st) memset(xyzt + 16 * i, 0, 128); else el_store(e, xyzt + 16 * i);
    }
}
API void d377o_compress(const uint64_t *xyzt, size_t n, uint8_t *enc32) {
    for (size_t i = 0; i < n; ++i) el_compress(el_load(xyzt + 16 * i), enc32 + 32 * i);
}
API void d377o_roundtrip(const uint8_t *enc32, size_t n, uint8_t *out32, uint8_t *status) {
    for (size_t i = 0; i < n; ++i) {
        element e; int st = el_decompress(enc32 + 32 * i, &e);
        status[i] = (uint8_t)st;
        if (st) memset(out32 + 32 * i, 0, 32); else el_compress(e, out32 + 32 * i);
    }
}
API void d377o_scalar_mul_base(const uint8_t *scalar32, size_t n, uint8_t *out32) {
    for (size_t i = 0; i < n; ++i) {
        uint64_t k[4]; fr_from_le_bytes_mod_order(scalar32 + 32 * i, k);
        el_compress(el_scalar_mul(el_generator(), k), out32 + 32 * i);
    }
}
API void d377o_scalar_mul_var(const uint8_t *enc32, const uint8_t *scalar32, size_t n,
                              uint8_t *out32, uint8_t *status) {
    for (size_t i = 0; i < n; ++i) {
        element e; int st = el_decompress(enc32 + 32 * i, &e);
        status[i] = (uint8_t)st;
        if (st) { memset(out32 + 32 * i, 0, 32); continue; }
        uint64_t k[4]; fr_from_le_bytes_mod_order(scalar32 + 32 * i, k);
        el_compress(el_scalar_mul(e, k), out32 + 32 * i);
    }
}
API void d377o_encode_to_curve(const uint8_t *fq32, size_t n, uint8_t *out32) {
    for (size_t i = 0; i < n; ++i)
        el_compress(el_elligator_map(fq_from_le_bytes_mod_order(fq32 + 32 * i)), out32 + 32 * i);
}
/* extended coordinates of the Elligator image (for the affine (x,y) KATs) */
API void d377o_elligator_map_xyzt(const uint8_t *fq32, size_t n, uint64_t *xyzt) {
    for (size_t i = 0; i < n; ++i)
        el_store(el_elligator_map(fq_from_le_bytes_mod_order(fq32 + 32 * i)), xyzt + 16 * i);
}
/* src/ark_curve/elligator.rs:67-71 */
API void d377o_hash_to_curve(const uint8_t *r1, const uint8_t *r2, size_t n, uint8_t *out32) {
    for (size_t i = 0; i < n; ++i)
        el_compress(el_add(el_elligator_map(fq_from_le_bytes_mod_order(r1 + 32 * i)),
                           el_elligator_map(fq_from_le_bytes_mod_order(r2 + 32 * i))), out32 + 32 * i);
}
/* group law helpers on extended Montgomery coordinates (tests/operations.rs properties) */
API void d377o_add_xyzt(const uint64_t *p, const uint64_t *q, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; ++i) el_store(el_add(el_load(p + 16 * i), el_load(q + 16 * i)), out + 16 * i);
}
API void d377o_double_xyzt(const uint64_t *p, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; ++i) el_store(el_double(el_load(p + 16 * i)), out + 16 * i);
}
API void d377o_scalar_mul_xyzt(const uint64_t *p, const uint8_t *scalar32, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; ++i) {
        uint64_t k[4]; fr_from_le_bytes_mod_order(scalar32 + 32 * i, k);
        el_store(el_scalar_mul(el_load(p + 16 * i), k), out + 16 * i);
    }
}
/* decaf equality, src/ark_curve/element/projective.rs:65-70 */
API void d377o_eq_xyzt(const uint64_t *p, const uint64_t *q, size_t n, uint8_t *eq) {
    for (size_t i = 0; i < n; ++i) {
        element a = el_load(p + 16 * i), b = el_load(q + 16 * i);
        eq[i] = (uint8_t)fq_eq(fq_mul(a.x, b.y), fq_mul(a.y, b.x));
    }
}
API void d377o_generator_xyzt(uint64_t *xyzt) { el_store(el_generator(), xyzt); }
API void d377o_identity_xyzt(uint64_t *xyzt) { el_store(el_identity(), xyzt); }
/* src/min_curve/element.rs:324-332 */
API void d377o_neg_xyzt(const uint64_t *p, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; ++i) {
        element e = el_load(p + 16 * i);
        e.x = fq_neg(e.x); e.t = fq_neg(e.t);
        el_store(e, out + 16 * i);
    }
}
/* src/min_curve/element.rs:113-117 */
API void d377o_is_identity(const uint64_t *p, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)fq_is_zero(el_load(p + 16 * i).x);
}
/* Fq operations on Montgomery limbs (src/fields/fq/u64/wrapper.rs:99-132); op codes as in
 * include/decaf377_amd.h: 0 add, 1 sub, 2 mul, 3 square, 4 neg, 5 inverse (status 1 and a zero
 * record for the inverse of zero: wrapper.rs:104-112 returns None) */
API void d377o_fq_op(int op, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out, uint8_t *status) {
    for (size_t i = 0; i < n; ++i) {
        fq x, y = FQ_ZERO, r; memcpy(x.l, a + 4 * i, 32);
        if (op <= 2) memcpy(y.l, b + 4 * i, 32);
        uint8_t st = 0;
        switch (op) {
        case 0: r = fq_add(x, y); break;
        case 1: r = fq_sub(x, y); break;
        case 2: r = fq_mul(x, y); break;
        case 3: r = fq_square(x); break;
        case 4: r = fq_neg(x); break;
        default: if (fq_is_zero(x)) { r = FQ_ZERO; st = 1; } else r = fq_inverse(x); break;
        }
        memcpy(out + 4 * i, r.l, 32);
        if (status) status[i] = st;
    }
}
/* Fr operations on 32-byte little-endian scalars (inputs reduced mod r, outputs canonical), same op codes */
API void d377o_fr_op(int op, const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out, uint8_t *status) {
    for (size_t i = 0; i < n; ++i) {
        uint64_t x[4], y[4] = {0, 0, 0, 0}, r[4];
        fr_from_le_bytes_mod_order(a + 32 * i, x);
        if (op <= 2) fr_from_le_bytes_mod_order(b + 32 * i, y);
        uint8_t st = 0;
        switch (op) {
        case 0: fr_add(x, y, r); break;
        case 1: fr_sub(x, y, r); break;
        case 2: fr_mul(x, y, r); break;
        case 3: fr_mul(x, x, r); break;
        case 4: fr_neg(x, r); break;
        default: st = (uint8_t)!fr_inverse(x, r); break;
        }
        store_le(r, out + 32 * i);
        if (status) status[i] = st;
    }
}
API void d377o_fr_from_wide_bytes(const uint8_t *in, size_t len, size_t n, uint8_t *out32) {
    for (size_t i = 0; i < n; ++i) { uint64_t r[4]; fr_from_le_bytes_mod_order_any(in + len * i, len, r); store_le(r, out32 + 32 * i); }
}
/* Element::vartime_compress_to_field (src/min_curve/element.rs:163-181) as Montgomery limbs */
API void d377o_compress_to_field(const uint64_t *xyzt, size_t n, uint64_t *mont) {
    for (size_t i = 0; i < n; ++i) { fq s = el_compress_to_field(el_load(xyzt + 16 * i)); memcpy(mont + 4 * i, s.l, 32); }
}
/* Element::hash_to_curve (src/min_curve/element.rs:235-240) as an Element */
API void d377o_hash_to_curve_xyzt(const uint8_t *r1, const uint8_t *r2, size_t n, uint64_t *xyzt) {
    for (size_t i = 0; i < n; ++i)
        el_store(el_add(el_elligator_map(fq_from_le_bytes_mod_order(r1 + 32 * i)),
                        el_elligator_map(fq_from_le_bytes_mod_order(r2 + 32 * i))), xyzt + 16 * i);
}
/* the min_curve backend's root (src/min_curve/invsqrt.rs:73-95) */
API void d377o_sqrt_ratio_zeta_min_curve(const uint8_t *num32, const uint8_t *den32, size_t n,
                                         uint8_t *root32, uint8_t *was_square) {
    for (size_t i = 0; i < n; ++i) {
        fq r;
        int ws = fq_sqrt_ratio_zeta_min_curve(fq_from_le_bytes_mod_order(num32 + 32 * i),
                                              fq_from_le_bytes_mod_order(den32 + 32 * i), &r);
        fq_to_bytes(r, root32 + 32 * i);
        was_square[i] = (uint8_t)ws;
    }
}

/* Fq::from_le_bytes_mod_order for any length (src/fields/fq.rs:90-102): 32-byte chunks, folded
 * from the most significant one with acc * FIELD_SIZE_POWER_OF_TWO + chunk (fq.rs:83-88 = 2^256 mod q). */
static const fq FQ_FIELD_SIZE_POWER_OF_TWO = {{2726216793283724667ULL, 14712177743343147295ULL,
                                               12091039717619697043ULL, 81024008013859129ULL}};
static fq fq_from_le_bytes_mod_order_any(const uint8_t *b, size_t len) {
    fq acc = FQ_ZERO;
    size_t nchunks = (len + 31) / 32;
    for (size_t c = nchunks; c-- > 0;) {
        uint8_t padded[32] = {0};
        size_t l = (c * 32 + 32 <= len) ? 32 : len - c * 32;
        memcpy(padded, b + c * 32, l);
        acc = fq_add(fq_mul(acc, FQ_FIELD_SIZE_POWER_OF_TWO), fq_from_le_bytes_mod_order(padded));
    }
    return acc;
}
API void d377o_fq_from_wide_bytes(const uint8_t *in, size_t len, size_t n, uint8_t *out32) {
    for (size_t i = 0; i < n; ++i) fq_to_bytes(fq_from_le_bytes_mod_order_any(in + len * i, len), out32 + 32 * i);
}
API void d377o_encode_to_curve_wide(const uint8_t *in, size_t len, size_t n, uint8_t *out32) {
    for (size_t i = 0; i < n; ++i)
        el_compress(el_elligator_map(fq_from_le_bytes_mod_order_any(in + len * i, len)), out32 + 32 * i);
}
/* CurveGroup::normalize_batch / into_affine (src/ark_curve/element.rs:74-85): (x/z, y/z) as Montgomery limbs */
API void d377o_to_affine(const uint64_t *xyzt, size_t n, uint64_t *xy) {
    for (size_t i = 0; i < n; ++i) {
        element e = el_load(xyzt + 16 * i);
        fq zi = fq_inverse(e.z);
        fq x = fq_mul(e.x, zi), y = fq_mul(e.y, zi);
        memcpy(xy + 8 * i, x.l, 32); memcpy(xy + 8 * i + 4, y.l, 32);
    }
}

/* Element::vartime_multiscalar_mul, src/ark_curve/element/projective.rs:99-117:
 * fold(Element::default(), |acc, (scalar, point)| acc + scalar * point), result compressed. */
API void d377o_msm(const uint64_t *xyzt, const uint8_t *scalar32, size_t n, uint8_t *enc32_out, uint64_t *xyzt_out) {
    element acc = el_identity();
    for (size_t i = 0; i < n; ++i) {
        uint64_t k[4]; fr_from_le_bytes_mod_order(scalar32 + 32 * i, k);
        acc = el_add(acc, el_scalar_mul(el_load(xyzt + 16 * i), k));
    }
    el_compress(acc, enc32_out);
    if (xyzt_out) el_store(acc, xyzt_out);
}

/* ---- threaded driver for the timed CPU baseline (contiguous slices) ---- */
typedef struct { int op; const uint8_t *a, *b; uint8_t *out, *status; size_t n; } job;
static void *job_run(void *p) {
    job *j = (job *)p;
    switch (j->op) {
    case 0: d377o_roundtrip(j->a, j->n, j->out, j->status); break;
    case 1: d377o_scalar_mul_base(j->a, j->n, j->out); break;
    case 2: d377o_scalar_mul_var(j->a, j->b, j->n, j->out, j->status); break;
    case 3: d377o_encode_to_curve(j->a, j->n, j->out); break;
    case 4: d377o_sqrt_ratio_zeta(j->a, j->b, j->n, j->out, j->status); break;
    case 5: d377o_sqrt_ratio_zeta_min_curve(j->a, j->b, j->n, j->out, j->status); break;
    }
    return NULL;
}
/* op: 0 roundtrip, 1 scalar_mul_base, 2 scalar_mul_var, 3 encode_to_curve, 4 sqrt_ratio_zeta, 5 the min_curve root */
API int d377o_run_threads(int op, const uint8_t *a, const uint8_t *b, size_t n,
                          uint8_t *out, uint8_t *status, int threads) {
    d377o_init();
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; job jobs[256];
    size_t per = (n + (size_t)threads - 1) / (size_t)threads;
    int used = 0;
    for (int t = 0; t < threads; ++t) {
        size_t lo = per * (size_t)t; if (lo >= n) break;
        size_t cnt = (lo + per <= n) ? per : n - lo;
        job jb = {op, a + 32 * lo, b ? b + 32 * lo : NULL, out + 32 * lo, status ? status + lo : NULL, cnt};
        jobs[t] = jb;
        if (pthread_create(&th[t], NULL, job_run, &jobs[t])) return -1;
        ++used;
    }
    for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
    return used;
}
