// curve.hpp -- per-lane decaf377 operations on top of fq29.hpp.
//
// Each function is the work ONE lane does for ONE group element; the kernels in
// d377.hip / msm.hip only add the batch indexing and the 32-byte record loads/stores.
// Reference semantics (file:line relative to the reference crate):
//   sqrt_ratio_zeta      src/ark_curve/invsqrt.rs:75-166 (Sarkar tables, same root)
//   decompress           src/ark_curve/encoding.rs:32-83
//   compress             src/ark_curve/encoding.rs:91-128
//   elligator_map        src/ark_curve/elligator.rs:15-62
//   ge_add / ge_double   src/min_curve/element.rs:291-322 / 119-136
//   scalar multiplication: same group element as src/min_curve/element.rs:138-157, computed
//   with signed 4-bit windows (encodings are canonical per group element, so the result
//   bytes are identical).
#pragma once
#include "fq29.hpp"

namespace d377 {

#include "constants.inc"

// ---- device tables (built once per context by the init kernels) -----------------------
// gtab: 6 tables (g^(nu * 2^{0,8,16,24,32,40})), 256 entries each, 12 words per entry
//       (9 limbs + 3 pad so an entry is three 16-byte loads).
// s_lookup: perfect hash of the 256 elements g^-(nu * 2^39) (both tight representations)
//       -> nu.  4096 one-byte slots.
constexpr int GT_STRIDE = 12;
constexpr int S_HASH_BITS = 12;
struct SqrtTables {
  const uint32_t* gtab;      // [6][256][GT_STRIDE]
  const uint8_t* s_lookup;   // [1 << S_HASH_BITS]
  uint32_t* inv_fail;        // -DD377_CHECK_INVARIANTS builds: device counter of violated invariants (else unused)
};

#include "s_hash.inc"        // defines D377_S_HASH_K (searched offline, verified at init)

D377_HD uint32_t s_hash_raw(const fe& x) {     // table construction: explicit representations
  return ((x.l[0] ^ (x.l[1] << 3)) * D377_S_HASH_K) >> (32 - S_HASH_BITS);
}
D377_HD uint32_t s_hash(const fe& x) {         // lookups: the key must be one of x, x + q with x < 2^248
  D377_B(bound_require(x.vq < 1.0535, "s_lookup key must be below q + 2^248"));
  return s_hash_raw(x);
}

D377_HD fe gt_load(const SqrtTables& T, int table, uint32_t idx) {
  const uint32_t* p = T.gtab + ((size_t)table * 256 + idx) * GT_STRIDE;
  fe r;
#if defined(__HIP_DEVICE_COMPILE__)
  const uint4* p4 = reinterpret_cast<const uint4*>(p);
  uint4 a = p4[0], b = p4[1], c = p4[2];
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x;
#else
  for (int i = 0; i < NL; ++i) r.l[i] = p[i];
#endif
  fe_assume_carried(r, 1.1);      // the init kernels store strict products: value < 1.02q
  return r;
}

// ---- fixed exponentiations --------------------------------------------------------------
// Two squarings per trip: the hand-written squarer's outputs may not overlap its inputs, so a one-squaring loop copies
// the nine limbs back every trip (18 v_mov per 168 instructions); with two the second squaring lands in the first
// one's input registers and the copies disappear.
D377_HD fe fe_sqr_n(fe x, int n) {
#pragma unroll 1
  for (int i = 0; i + 1 < n; i += 2) x = fe_sqr(fe_sqr(x));
  if (n & 1) x = fe_sqr(x);
  return x;
}

// x^(2^47 - 1): 46 S + 9 M   (e_k = x^(2^k - 1); e_2k = e_k^(2^k) e_k; e_2k+1 = e_2k^2 x)
D377_HD fe fe_pow_2_47_m1(const fe& x) {
  fe e2 = fe_mul(fe_sqr(x), x);
  fe e4 = fe_mul(fe_sqr_n(e2, 2), e2);
  fe e5 = fe_mul(fe_sqr(e4), x);
  fe e10 = fe_mul(fe_sqr_n(e5, 5), e5);
  fe e11 = fe_mul(fe_sqr(e10), x);
  fe e22 = fe_mul(fe_sqr_n(e11, 11), e11);
  fe e23 = fe_mul(fe_sqr(e22), x);
  fe e46 = fe_mul(fe_sqr_n(e23, 23), e23);
  return fe_mul(fe_sqr(e46), x);
}

// x^((m-1)/2), m = (q-1)/2^47: sliding window (w = 4) over the fixed exponent, the schedule
// is the table D377_POW_CHAIN (wave-uniform, so no divergence): w=4: 201 S + 36 M + (1 S + 7 M table); w=3: see constants.inc.
// The 8 odd powers live in `PT` (LDS on the GPU, one column per lane: conflict-free), which
// keeps 72 VGPRs free for a second wave per SIMD.
#ifndef D377_POW_W
#define D377_POW_W 4                    // window of the fixed exponentiation: 3 or 4
#endif
#if D377_POW_W == 3
#define D377_POW_CHAIN POW_M12_CHAIN_W3
#define D377_POW_LEN D377_POW_M12_W3_LEN
#define D377_POW_TRAIL D377_POW_M12_W3_TRAILING_SQ
#else
#define D377_POW_CHAIN POW_M12_CHAIN_W4
#define D377_POW_LEN D377_POW_M12_W4_LEN
#define D377_POW_TRAIL D377_POW_M12_W4_TRAILING_SQ
#endif
constexpr int POW_TAB = 1 << (D377_POW_W - 1);   // odd powers x^1 .. x^(2^w - 1)

struct RegPowTab {                      // plain registers / stack (host simulation, init kernels)
  fe t[8];
  D377_HD void put(int j, const fe& v) { t[j] = v; }
  D377_HD fe get(int j) const {
    fe f;
    switch (j) {
      case 0: f = t[0]; break;
      case 1: f = t[1]; break;
      case 2: f = t[2]; break;
      case 3: f = t[3]; break;
      case 4: f = t[4]; break;
      case 5: f = t[5]; break;
      case 6: f = t[6]; break;
      default: f = t[7]; break;
    }
    return f;
  }
};

template <class PT>
D377_HD fe fe_pow_m12(const fe& x, PT& pt) {
  fe x2 = fe_sqr(x);
  fe cur = x;
  pt.put(0, cur);
#pragma unroll 1
  for (int j = 1; j < POW_TAB; ++j) {
    cur = fe_mul(cur, x2);
    pt.put(j, cur);
  }
  fe acc = pt.get((int)(D377_POW_CHAIN[0] & 15u) >> 1);
#pragma unroll 1
  for (int i = 1; i < D377_POW_LEN; ++i) {
    const uint32_t e = D377_POW_CHAIN[i];
    acc = fe_sqr_n(acc, (int)(e >> 4));
    acc = fe_mul(acc, pt.get((int)(e & 15u) >> 1));
  }
  return fe_sqr_n(acc, D377_POW_TRAIL);
}

// zero test for a strict product (product limbs, value < 2q): 0 or q
D377_HD bool fe_strict_is_zero(const fe& a) {
  D377_B(bound_require(a.vq < 2.0, "fe_strict_is_zero needs a value below 2q"));
  uint32_t o = 0, d = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) { o |= a.l[i]; d |= a.l[i] ^ QL[i]; }
  return o == 0 || d == 0;
}

// ---- sqrt_ratio_zeta ----------------------------------------------------------------------
// src/ark_curve/invsqrt.rs:75-166.  `den` (and `num`) must be strict products (value < 2q: the zero
// tests compare with 0 and q).  The values that go through s_lookup are strict products of strict
// products, so that each has at most the two representations x and x + q (x < 2^248) the table holds.
// NUM_IS_ONE: the callers on the group path always pass num = 1 (encoding.rs:57,100;
// elligator.rs:26); the generic form is used by the raw batch entry point.
// min_curve_root: return the root the min_curve backend's constant-time Tonelli-Shanks returns
// (src/min_curve/invsqrt.rs:11-95, seed 11^m) instead of the arkworks backend's Sarkar root: the two
// differ by a sign that is a function of the discrete log t the table phase has already found
// (constants.inc, D377_TS_U), so no second square root is computed.
// inv_den: 1/den when the caller has it (the kernels invert the denominators of a lane's whole round at once,
// dcb_invert_slot below); den = 0 may come with any inv_den.  invsqrt.rs:88-94 builds v = z^((m-1)/2) and
// uv = z^((m+1)/2), z = num/den, WITHOUT an inversion -- s = den^(2^47-1), t = s^2 den, w = (num t)^((m-1)/2) s,
// whose exponent of den is -(m+1)/2 modulo q - 1 -- at the price of the 46 S + 9 M chain for s; with 1/den at hand
// the same two field values are z^((m-1)/2) and its product with z, and everything after them is unchanged.
// The table phase and everything after it (invsqrt.rs:97-166), from v = z^((m-1)/2) and uv = z^((m+1)/2): shared by
// fe_sqrt_ratio_zeta below and by the callers that raise z to those powers elsewhere (the one-wave-per-element kernels:
// row_ops.hpp runs the 300 products of the power chains in the lane-spread form).
D377_HD bool fe_sqrt_tail(const SqrtTables& T, const fe& v, const fe& uv, bool den_zero, bool num_zero, fe* res, bool min_curve_root);

// A "power table" that holds the two powers themselves: the four-elements-per-wave kernels (d377.hip) raise z to them on
// the rows of the wave and pass the results where the other callers pass the table of the exponentiation.
struct GivenPowers {
  fe v, uv;                             // z^((m-1)/2), z^((m+1)/2), z = num / den
  D377_HD void put(int, const fe&) {}
  D377_HD fe get(int) const { return fe_zero(); }
};
template <class PT> struct pt_has_powers { static constexpr bool value = false; };
template <> struct pt_has_powers<GivenPowers> { static constexpr bool value = true; };

template <bool NUM_IS_ONE, class PT>
D377_HD bool fe_sqrt_ratio_zeta(const SqrtTables& T, PT& pt, const fe& num, const fe& den, fe* res,
                                bool min_curve_root = false, const fe* inv_den = nullptr, bool use_inv = true) {
  const bool den_zero = fe_strict_is_zero(den);
  bool num_zero = false;
  if (!NUM_IS_ONE) num_zero = fe_strict_is_zero(num);

  fe v, uv;
  // use_inv: a launch-uniform switch for callers that always hold a (possibly meaningless) inverse: a pointer that is
  // null on one path and the address of a local on the other would force that local into scratch memory
  if constexpr (pt_has_powers<PT>::value) {
    v = pt.v;
    uv = pt.uv;
  } else if (inv_den != nullptr && use_inv) {
    const fe z = NUM_IS_ONE ? *inv_den : fe_mul(num, *inv_den);
    v = fe_pow_m12(z, pt);
    uv = fe_mul(v, z);
  } else {
    fe s = fe_pow_2_47_m1(den);                       // invsqrt.rs:88-89
    fe t_ = fe_mul(fe_sqr(s), den);                   // :90
    fe w = NUM_IS_ONE ? fe_mul(fe_pow_m12(t_, pt), s)     // :91
                      : fe_mul(fe_pow_m12(fe_mul(num, t_), pt), s);
    v = fe_mul(w, den);                               // :93
    uv = NUM_IS_ONE ? w : fe_mul(w, num);             // :94
  }
  return fe_sqrt_tail(T, v, uv, den_zero, num_zero, res, min_curve_root);
}
D377_HD bool fe_sqrt_tail(const SqrtTables& T, const fe& v, const fe& uv, bool den_zero, bool num_zero, fe* res, bool min_curve_root) {
  fe x5 = fe_mul(uv, v);                            // :97
  fe x4 = fe_sqr_n(x5, 8);                          // :101-107
  fe x3 = fe_sqr_n(x4, 8);
  fe x2 = fe_sqr_n(x3, 8);
  fe x1 = fe_sqr_n(x2, 8);
  fe x0 = fe_sqr_strict(fe_sqr_strict(fe_sqr_n(x1, 5)));   // :110  (x1^(2^7))

  const uint64_t q0p = T.s_lookup[s_hash(x0)];      // :113
  uint64_t t = q0p;
  fe a1 = fe_mul_strict(x1, gt_load(T, 4, (uint32_t)(t & 0xFF)));                // :117-119
  t += (uint64_t)T.s_lookup[s_hash(a1)] << 7;
  fe a2 = fe_mul_strict(fe_mul(x2, gt_load(T, 3, (uint32_t)(t & 0xFF))),
                        gt_load(T, 4, (uint32_t)((t >> 8) & 0xFF)));             // :122-126
  t += (uint64_t)T.s_lookup[s_hash(a2)] << 15;
  fe a3 = fe_mul_strict(fe_mul(fe_mul(x3, gt_load(T, 2, (uint32_t)(t & 0xFF))),
                               gt_load(T, 3, (uint32_t)((t >> 8) & 0xFF))),
                        gt_load(T, 4, (uint32_t)((t >> 16) & 0xFF)));            // :129-134
  t += (uint64_t)T.s_lookup[s_hash(a3)] << 23;
  fe a4 = fe_mul_strict(fe_mul(fe_mul(fe_mul(x4, gt_load(T, 1, (uint32_t)(t & 0xFF))),
                                      gt_load(T, 2, (uint32_t)((t >> 8) & 0xFF))),
                               gt_load(T, 3, (uint32_t)((t >> 16) & 0xFF))),
                        gt_load(T, 4, (uint32_t)((t >> 24) & 0xFF)));            // :137-143
  t += (uint64_t)T.s_lookup[s_hash(a4)] << 31;
  fe a5 = fe_mul_strict(fe_mul(fe_mul(fe_mul(fe_mul(x5, gt_load(T, 0, (uint32_t)(t & 0xFF))),
                                             gt_load(T, 1, (uint32_t)((t >> 8) & 0xFF))),
                                      gt_load(T, 2, (uint32_t)((t >> 16) & 0xFF))),
                               gt_load(T, 3, (uint32_t)((t >> 24) & 0xFF))),
                        gt_load(T, 4, (uint32_t)((t >> 32) & 0xFF)));            // :146-153
  t += (uint64_t)T.s_lookup[s_hash(a5)] << 39;

  // sign of the Tonelli-Shanks (11^m) root relative to this one: bit 46 of u * (u^-1 * (t >> 1) mod 2^46)
  const uint64_t e2 = (D377_TS_U_INV * (t >> 1)) & ((1ull << 46) - 1);
  const bool flip = min_curve_root && (((D377_TS_U * e2) >> 46) & 1ull) != 0;
  t = (t + 1) >> 1;                                                              // :155
  const bool nonsq = (q0p & 1) != 0;
  fe r = fe_select(nonsq, fe_mul(uv, fe_const(FE_NONSQUARE)), uv);               // :156-157
  r = fe_mul(r, gt_load(T, 0, (uint32_t)(t & 0xFF)));
  r = fe_mul(r, gt_load(T, 1, (uint32_t)((t >> 8) & 0xFF)));
  r = fe_mul(r, gt_load(T, 2, (uint32_t)((t >> 16) & 0xFF)));
  r = fe_mul(r, gt_load(T, 3, (uint32_t)((t >> 24) & 0xFF)));
  r = fe_mul(r, gt_load(T, 4, (uint32_t)((t >> 32) & 0xFF)));
  r = fe_mul(r, gt_load(T, 5, (uint32_t)((t >> 40) & 0xFF)));                    // :158-163

  if (min_curve_root) r = fe_select(flip, fe_neg(r), r);
  bool was_square = !nonsq;
  // early-outs of invsqrt.rs:81-86, applied as selects so the wave stays converged
  if (den_zero) { r = fe_zero(); was_square = false; }
  if (num_zero) { r = fe_zero(); was_square = true; }
  *res = r;
  return was_square;
}

// ---- byte / word conversions -------------------------------------------------------------
// 32 LE bytes (as 8 words, any 256-bit value) -> Montgomery-261, reduced mod q on the way
// (Fq::from_le_bytes_mod_order, src/fields/fq.rs:90-102, for 32-byte inputs).
D377_HD fe fe_from_words_mod_order(const uint32_t w[8]) {
  return fe_mul(fe_from_words(w), fe_const(FE_R2));
}
D377_HD fe fe_from_words_mod_order_strict(const uint32_t w[8]) {   // value < 2q (sqrt_ratio_zeta operands)
  return fe_mul_strict(fe_from_words(w), fe_const(FE_R2));
}
// Fq::from_le_bytes_mod_order for 33..64 input bytes (src/fields/fq.rs:90-102): two 32-byte
// chunks, value = lo + 2^256 * hi (hi zero-padded).  The result is a lazy sum of two products.
D377_HD fe fe_from_wide_words(const uint32_t lo[8], const uint32_t hi[8]) {
  return fe_add(fe_mul(fe_from_words(lo), fe_const(FE_R2)), fe_mul(fe_from_words(hi), fe_const(FE_R2_SHIFT256)));
}
// A whole element <-> the 10 x 28-bit limbs of the lane-spread form (row_ops.hpp: plain residues, one 16-word record per
// element, words 10..15 zero).  Out: the canonical value.  In: lazily reduced limbs (each below 2^29, value below 2^273:
// what row_mul leaves), back as a PRODUCT -- tight limbs, value < 9q, what every consumer of a whole element (doubling,
// cached forms, the encoder) is specified for.  (The sum of the two wide-reduction products by itself is lazy with a value
// up to 18q and was handed on like that at first: the doubling of such a point overflowed a column for some inputs.)
D377_HD void fe_to_limbs28(const fe& x, uint32_t rec16[16]) {
  uint32_t w[8];
  fe_to_words(fe_canon(x), w);
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const int bit = 28 * k, lo = bit >> 5, sh = bit & 31;
    uint32_t v = w[lo] >> sh;
    if (sh + 28 > 32 && lo + 1 < 8) v |= w[lo + 1] << (32 - sh);
    rec16[k] = v & 0x0FFFFFFFu;
  }
#pragma unroll
  for (int k = 10; k < 16; ++k) rec16[k] = 0;
}
D377_HD fe fe_from_limbs28(const uint32_t rec16[16]) {
  uint32_t w[9];
  uint64_t acc = 0;                                                // the integer, 32 bits at a time
  int have = 0, wi = 0;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    acc += (uint64_t)rec16[k] << have;                             // limbs below 2^29: the carry rides in acc
    have += 28;
    if (have >= 32) { w[wi++] = (uint32_t)acc; acc >>= 32; have -= 32; }
  }
  w[wi] = (uint32_t)acc;                                           // wi == 8: bits 256 .. (value < 2^273)
  const uint32_t hi[8] = {w[8], 0, 0, 0, 0, 0, 0, 0};
  return fe_mul(fe_from_wide_words(w, hi), fe_const(FE_ONE));      // (lo R + hi 2^256 R), then times R / R
}
// Montgomery-261 -> canonical 32 bytes (Fq::to_bytes_le)
D377_HD void fe_to_bytes_words(const fe& a, uint32_t w[8]) { fe_to_words(fe_canon(a), w); }

// full reduction of a tight value < 2q to [0, q)
D377_HD fe fe_reduce_once(const fe& a) {
  D377_B(bound_require(a.vq < 2.0, "fe_reduce_once needs a value below 2q"));
  fe d;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    uint32_t t = a.l[i] - QL[i] - borrow;
    borrow = t >> 31;
    d.l[i] = (i < NL - 1) ? (t & MASK29) : t;
  }
#if defined(D377_BOUNDS)
  for (int i = 0; i < NL; ++i) d.ub[i] = a.ub[i];
  d.vq = 1.0;
#endif
  fe r = fe_select(borrow != 0, a, d);
  D377_B(r.vq = 1.0);
  return r;
}
// q - c for canonical c != 0 (plain integers, tight limbs)
D377_HD fe fe_canon_negate(const fe& c) {
  fe d;
  uint32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    uint32_t t = QL[i] - c.l[i] - borrow;
    borrow = t >> 31;
    d.l[i] = (i < NL - 1) ? (t & MASK29) : t;
  }
#if defined(D377_BOUNDS)
  for (int i = 0; i < NL - 1; ++i) d.ub[i] = MASK29;
  d.ub[NL - 1] = QL[NL - 1];
  d.vq = 1.0;
#endif
  return d;
}

}  // namespace d377
#include "inv30.hpp"
namespace d377 {

// 1/x for a Montgomery-261 element in any lazy / carried / product form (0 -> 0, as x^(q-2) gives): the plain
// integer inverse of the residue x R by divsteps (inv30.hpp), times R^3 / R.  ~19 600 instructions.
D377_HD fe fe_invert(const fe& x) {
  const fe c = fe_reduce_once(fe_mul_strict(x, fe_const(FE_ONE)));       // the residue x R in [0, q), tight limbs
  fe y;
  modinv_limbs29(c.l, y.l);
#if defined(D377_BOUNDS)
  for (int i = 0; i < NL - 1; ++i) y.ub[i] = MASK29;
  y.ub[NL - 1] = QL[NL - 1];
  y.vq = 1.0;
#endif
  return fe_mul(y, fe_const(FE_R3));
}

// the reference's external element layout: 4 x u64 Montgomery limbs, R = 2^256
// (Fq::from_montgomery_limbs, src/fields/fq/u64/wrapper.rs:82-85), as 8 x u32 words.
D377_HD void fe_to_mont256_words(const fe& a, uint32_t w[8]) {
  fe_to_words(fe_reduce_once(fe_mul_strict(a, fe_const(FE_TO_MONT256))), w);
}
D377_HD fe fe_from_mont256_words(const uint32_t w[8]) {
  return fe_mul(fe_from_words(w), fe_const(FE_FROM_MONT256));
}

// ---- records used without conversion ---------------------------------------------------------
// The eight words of x * 2^256 (a reference Fq in memory), taken as limbs, are the Montgomery-261 form of x * 2^-5.
// An Element whose four coordinates are all scaled by one factor is the same projective point, and the addition /
// doubling formulas are homogeneous (degree 2 in each operand), so the element-wise kernels skip the four (eight)
// conversion products on the way in and fold the accumulated power of two into the constant of the product that
// writes the record: the words they store are those of the reference formulas on the unscaled coordinates.
D377_HD void fe_scaled_to_mont256_words(const fe& a, const uint32_t (&c)[NL], uint32_t w[8]) {
  fe_to_words(fe_reduce_once(fe_mul_strict(a, fe_const(c))), w);
}
// q - x on canonical words (x = 0 stays 0); false if x was not canonical (the caller converts the long way then)
D377_HD bool fq_neg_words(const uint32_t x[8], uint32_t out[8]) {
  uint32_t nz = 0;
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    nz |= x[i];
    const uint64_t t = (uint64_t)FQ_MODULUS_W_LIT[i] - x[i] - borrow;
    out[i] = (uint32_t)t;
    borrow = (t >> 63) & 1u;
  }
  if (nz == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = 0;
  }
  return borrow == 0 && !(nz != 0 && words_geq(x, FQ_MODULUS_W_LIT));
}
// (a + b) mod q and (a - b) mod q on canonical words; false if an operand was not canonical
D377_HD bool fq_addsub_words(const uint32_t a[8], const uint32_t b[8], bool sub, uint32_t out[8]) {
  const bool ok = !words_geq(a, FQ_MODULUS_W_LIT) && !words_geq(b, FQ_MODULUS_W_LIT);
  uint32_t bb[8];
  if (sub) (void)fq_neg_words(b, bb);
#pragma unroll
  for (int i = 0; i < 8; ++i) if (!sub) bb[i] = b[i];
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c += (uint64_t)a[i] + bb[i]; out[i] = (uint32_t)c; c >>= 32; }   // < 2q < 2^254
  if (words_geq(out, FQ_MODULUS_W_LIT)) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint64_t t = (uint64_t)out[i] - FQ_MODULUS_W_LIT[i] - borrow;
      out[i] = (uint32_t)t;
      borrow = (t >> 63) & 1u;
    }
  }
  return ok;
}

// src/sign.rs:11-17
D377_HD fe fe_abs(const fe& a) { return fe_select(fe_is_negative(a), fe_neg(a), a); }

// ---- group ----------------------------------------------------------------------------------
struct ge { fe x, y, z, t; };   // extended twisted Edwards, a = -1, d = 3021; x*y = z*t

D377_HD ge ge_identity() {
  ge r;
  r.x = fe_zero(); r.y = fe_const(FE_ONE); r.z = fe_const(FE_ONE); r.t = fe_zero();
  return r;
}
D377_HD ge ge_generator() {
  ge r;
  r.x = fe_const(FE_BX); r.y = fe_const(FE_BY); r.z = fe_const(FE_ONE); r.t = fe_const(FE_BT);
  return r;
}

// src/min_curve/element.rs:291-322 (unified, complete: a = -1 is a square, d is not)
D377_HD ge ge_add(const ge& p, const ge& q) {
  fe a = fe_mul(fe_sub(p.y, p.x), fe_sub(q.y, q.x));
  fe b = fe_mul(fe_add(p.y, p.x), fe_add(q.y, q.x));
  fe c = fe_mul(fe_mul(fe_const(FE_K), p.t), q.t);
  fe d = fe_mul(fe_dbl(p.z), q.z);
  fe e = fe_sub(b, a), f = fe_sub(d, c), g = fe_add(d, c), h = fe_add(b, a);
  ge r;
  r.x = fe_mul(e, f); r.y = fe_mul(g, h); r.z = fe_mul(f, g); r.t = fe_mul(e, h);
  return r;
}

// p - q = p + (-q), -q = (-x, y, z, -t) (src/min_curve/ops.rs:43-49, element.rs:324-332): the addition above with
// y2 -+ x2 exchanged and the sign of C folded into F and G -- the same field values, no negation computed
D377_HD ge ge_sub_pts(const ge& p, const ge& q) {
  fe a = fe_mul(fe_sub(p.y, p.x), fe_add(q.y, q.x));
  fe b = fe_mul(fe_add(p.y, p.x), fe_sub(q.y, q.x));
  fe c = fe_mul(fe_mul(fe_const(FE_K), p.t), q.t);
  fe d = fe_mul(fe_dbl(p.z), q.z);
  fe e = fe_sub(b, a), f = fe_add(d, c), g = fe_sub(d, c), h = fe_add(b, a);
  ge r;
  r.x = fe_mul(e, f); r.y = fe_mul(g, h); r.z = fe_mul(f, g); r.t = fe_mul(e, h);
  return r;
}

// src/min_curve/element.rs:119-136
D377_HD ge ge_double(const ge& p) {
  fe a = fe_sqr(p.x), b = fe_sqr(p.y);
  fe c = fe_dbl(fe_sqr(p.z));
  fe ab = fe_add(a, b);
  fe e = fe_sub(fe_sqr(fe_add(p.x, p.y)), ab);
  fe g = fe_sub(b, a);          // d + b with d = -a
  fe f = fe_sub(g, c);
  fe h = fe_neg(ab);            // d - b
  ge r;
  r.x = fe_mul(e, f); r.y = fe_mul(g, h); r.z = fe_mul(f, g); r.t = fe_mul(e, h);
  return r;
}

D377_HD ge ge_neg(const ge& p) {   // src/min_curve/element.rs:324-332
  ge r = p;
  r.x = fe_neg(p.x); r.t = fe_neg(p.t);
  return r;
}
// Elements read without conversion (see "records used without conversion" above): coordinates scaled by 2^-5
D377_HD ge ge_from_raw_words(const uint32_t w[32]) {
  ge g;
  g.x = fe_from_words(w); g.y = fe_from_words(w + 8); g.z = fe_from_words(w + 16); g.t = fe_from_words(w + 24);
  return g;
}
// The results of ge_add / ge_double on raw operands are scaled by 2^-20; the reference's words come out of the formulas'
// last stage: X3 = E F, Y3 = G H, Z3 = F G, T3 = E H each hold exactly one of E, G, so scaling those two carries the
// power of two into all four coordinates -- two products, not one per coordinate.
D377_HD void ge_raw_efgh_to_words(const fe& e, const fe& f, const fe& g, const fe& h, uint32_t w[32]) {
  // (strict: es, gs below ~1.3 q keep the four products below 2q for the single conditional subtraction)
  const fe es = fe_mul_strict(e, fe_const(FE_RAW4_TO_MONT256)), gs = fe_mul_strict(g, fe_const(FE_RAW4_TO_MONT256));
  fe_to_words(fe_reduce_once(fe_mul_strict(es, f)), w);
  fe_to_words(fe_reduce_once(fe_mul_strict(gs, h)), w + 8);
  fe_to_words(fe_reduce_once(fe_mul_strict(gs, f)), w + 16);
  fe_to_words(fe_reduce_once(fe_mul_strict(es, h)), w + 24);
}
// P + Q, or P - Q (ge_add / ge_sub_pts above, one instruction stream for both), raw records in, the reference's words out
D377_HD void ge_add_raw_words(const uint32_t pw[32], const uint32_t qw[32], bool negate, uint32_t w[32]) {
  const ge p = ge_from_raw_words(pw), q = ge_from_raw_words(qw);
  const fe qm = fe_sub(q.y, q.x), qp = fe_carry(fe_add(q.y, q.x));
  const fe a = fe_mul(fe_sub(p.y, p.x), fe_select(negate, qp, qm));
  const fe b = fe_mul(fe_add(p.y, p.x), fe_select(negate, qm, qp));
  const fe c = fe_mul(fe_mul(fe_const(FE_K), p.t), q.t);
  const fe d = fe_mul(fe_dbl(p.z), q.z);
  const fe dmc = fe_sub(d, c), dpc = fe_carry(fe_add(d, c));
  ge_raw_efgh_to_words(fe_sub(b, a), fe_select(negate, dpc, dmc), fe_select(negate, dmc, dpc), fe_add(b, a), w);
}
// [2]P (ge_double above) likewise
D377_HD void ge_double_raw_words(const uint32_t pw[32], uint32_t w[32]) {
  const ge p = ge_from_raw_words(pw);
  const fe a = fe_sqr(p.x), b = fe_sqr(p.y);
  const fe c = fe_dbl(fe_sqr(p.z));
  const fe ab = fe_add(a, b);
  const fe e = fe_sub(fe_sqr(fe_add(p.x, p.y)), ab);
  const fe g = fe_sub(b, a);
  ge_raw_efgh_to_words(e, fe_sub(g, c), g, fe_neg(ab), w);
}
// decaf equality x1 * y2 == x2 * y1 (src/min_curve/element.rs:334-340): both sides carry the same scale
D377_HD bool ge_eq_raw_words(const uint32_t p[32], const uint32_t q[32]) {
  return fe_eq(fe_mul(fe_from_words(p), fe_from_words(q + 8)), fe_mul(fe_from_words(q), fe_from_words(p + 8)));
}
// -P on the record itself: X and T replaced by q - X, q - T.  False if a word string was not canonical.
D377_HD bool ge_neg_words(const uint32_t p[32], uint32_t out[32]) {
  bool ok = fq_neg_words(p, out);
#pragma unroll
  for (int i = 8; i < 24; ++i) out[i] = p[i];
  ok = fq_neg_words(p + 24, out + 24) && ok;
  return ok && !words_geq(p + 8, FQ_MODULUS_W_LIT) && !words_geq(p + 16, FQ_MODULUS_W_LIT);
}
// ---- normalize_batch (to_affine) on raw records ------------------------------------------------------------------
// Montgomery's trick on the z's as they lie in memory (each scaled by 2^-5, see above): the prefix products collect a
// power of two per factor, and it cancels again in  1 / z_k = inverse_k * prefix_(k-1)  up to one 2^5 -- which, with the
// change of radix, is a single constant multiplied into the lane's inverse once (FE_TO_MONT256).  Per element: one product
// on the way up, four on the way down (1/z_k, the next inverse, x / z, y / z), no conversion of x, y or z.
// A coordinate that is zero mod q (any of its 256-bit spellings) is no group element's z: flagged, kept out of the product.
D377_HD fe affine_raw_z(const uint32_t zw[8], bool* zero) {
  uint32_t nz = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) nz |= zw[k];
  bool z0 = nz == 0;
  if (!z0 && words_geq(zw, FQ_MODULUS_W_LIT)) z0 = fe_is_zero(fe_from_mont256_words(zw));   // a non-canonical string: the long way
  *zero = z0;
  return fe_select(z0, fe_const(FE_ONE), fe_from_words(zw));
}
// x / z and y / z as the reference's words; zi = the scaled 1 / z of this element
D377_HD void affine_raw_finish(const fe& zi, const uint32_t xw[8], const uint32_t yw[8], bool zero, uint32_t out[16]) {
  fe_to_words(fe_reduce_once(fe_mul_strict(fe_from_words(xw), zi)), out);
  fe_to_words(fe_reduce_once(fe_mul_strict(fe_from_words(yw), zi)), out + 8);
  if (zero) {
#pragma unroll
    for (int k = 0; k < 16; ++k) out[k] = 0;
  }
}

D377_HD ge ge_select(bool c, const ge& a, const ge& b) {
  ge r;
  r.x = fe_select(c, a.x, b.x); r.y = fe_select(c, a.y, b.y);
  r.z = fe_select(c, a.z, b.z); r.t = fe_select(c, a.t, b.t);
  return r;
}

// ---- debug invariants ------------------------------------------------------------------------
// The reference re-checks the curve equation whenever it builds an Element under debug assertions
// (`Element::new`, src/min_curve/element.rs:84-110; the arkworks path asserts `is_on_curve()`,
// src/ark_curve/encoding.rs:75-78, on_curve.rs:14-39) and its CI profile keeps them on.  A build with
// -DD377_CHECK_INVARIANTS does the same on the device after decompression, the Elligator map and the
// scalar-multiplication loops: -X^2 + Y^2 = Z^2 + d T^2, T Z = X Y, Z != 0 (the [2r]P = 0 test of
// on_curve.rs:26-35 costs a scalar multiplication per element and is left to the parity tests).
// A violation bumps a counter in device memory (d377_ctx_invariant_failures); nothing traps.
D377_HD bool ge_invariants_hold(const ge& p) {
  const fe lhs = fe_sub(fe_sqr(p.y), fe_sqr(p.x));
  const fe rhs = fe_add(fe_sqr(p.z), fe_mul(fe_const(FE_D), fe_sqr(p.t)));
  const bool on_curve = fe_eq(lhs, rhs);
  const bool segre = fe_eq(fe_mul(p.t, p.z), fe_mul(p.x, p.y));
  return on_curve && segre && !fe_is_zero(p.z);
}
#if defined(D377_CHECK_INVARIANTS) && defined(__HIP_DEVICE_COMPILE__)
#define D377_INVARIANT(T, point, enabled)                                            \
  do { if ((enabled) && !ge_invariants_hold(point)) atomicAdd((T).inv_fail, 1u); } while (0)
#else
#define D377_INVARIANT(T, point, enabled) do { } while (0)
#endif

// ---- encoding ------------------------------------------------------------------------------
// Encoding::vartime_decompress, src/ark_curve/encoding.rs:32-83.  Returns status
// (0 ok, 1 InvalidEncoding); on failure *out is unspecified (callers write zeros).
// the argument of decompression's square root, u2 u1^2 (encoding.rs:50-57), as a strict product
D377_HD fe ge_decompress_den(const uint32_t w[8]) {
  fe s = fe_mul(fe_from_words(w), fe_const(FE_R2));
  fe ss = fe_sqr(s);
  fe u1sq = fe_sqr_strict(fe_sub(fe_const(FE_ONE), ss));
  return fe_mul_strict(fe_sub(u1sq, fe_mul(fe_const(FE_4D), ss)), u1sq);
}
// SQRT: (den, &v) -> was_square, the square root of 1 / den (encoding.rs:57)
template <class SQRT>
D377_HD uint32_t ge_decompress_with(const SqrtTables& T, const uint32_t w[8], ge* out, SQRT sqrt_of) {
  uint32_t bad = (w[7] >> 29) != 0;                       // top three bits, encoding.rs:34
  bad |= (uint32_t)words_geq(w, FQ_MODULUS_W_LIT);        // canonical, :43-44
  bad |= (w[0] & 1u);                                     // s negative, :45
  fe s = fe_mul(fe_from_words(w), fe_const(FE_R2));
  fe ss = fe_sqr(s);                                      // :50
  fe u1 = fe_sub(fe_const(FE_ONE), ss);                   // :51
  fe u1sq = fe_sqr_strict(u1);                            // strict: keeps den below 2q for the zero test
  fe u2 = fe_sub(u1sq, fe_mul(fe_const(FE_4D), ss));      // :54
  fe v;
  const bool was_square = sqrt_of(fe_mul_strict(u2, u1sq), &v);                                                     // :57
  bad |= (uint32_t)!was_square;                           // :58-60
  fe two_s_u1 = fe_mul(fe_dbl(s), u1);                    // :63
  if (fe_is_negative(fe_mul(two_s_u1, v))) v = fe_neg(v); // :64-67
  out->x = fe_mul(fe_mul(two_s_u1, fe_sqr(v)), u2);       // :70
  out->y = fe_mul(fe_mul(fe_add(fe_const(FE_ONE), ss), v), u1);   // :71
  out->z = fe_const(FE_ONE);
  out->t = fe_mul(out->x, out->y);
  D377_INVARIANT(T, *out, bad == 0);                      // encoding.rs:75-78 / element.rs:104-110
  return bad;
}
template <class PT>
D377_HD uint32_t ge_decompress(const SqrtTables& T, PT& pt, const uint32_t w[8], ge* out, const fe* inv_den = nullptr) {
  return ge_decompress_with(T, w, out, [&](const fe& den, fe* v) {
    return fe_sqrt_ratio_zeta<true>(T, pt, fe_zero(), den, v, false, inv_den);
  });
}
// the same with v = (1/den)^((m-1)/2), uv = (1/den)^((m+1)/2) raised elsewhere (see fe_sqrt_tail)
D377_HD uint32_t ge_decompress_from_powers(const SqrtTables& T, const uint32_t w[8], const fe& pv, const fe& puv, ge* out) {
  return ge_decompress_with(T, w, out, [&](const fe& den, fe* v) {
    return fe_sqrt_tail(T, pv, puv, fe_strict_is_zero(den), false, v, false);
  });
}

// Element::vartime_compress, src/ark_curve/encoding.rs:91-128 -> canonical words of s
// the argument of compression's square root, u1 (a - d) X^2 (encoding.rs:97-101), as a strict product
D377_HD fe ge_compress_den(const ge& p) {
  fe u1 = fe_mul(fe_add(p.x, p.t), fe_sub(p.x, p.t));
  return fe_mul_strict(fe_mul(u1, fe_const(FE_A_MINUS_D)), fe_sqr(p.x));
}
template <class PT>
D377_HD void ge_compress(const SqrtTables& T, PT& pt, const ge& p, uint32_t w[8], bool is_element = true,
                         const fe* inv_den = nullptr) {
  // every Element the reference can hold satisfies the invariants (Element::new); lanes that carry the
  // leftovers of a failed decompression pass is_element = false
  D377_INVARIANT(T, p, is_element);
  const fe a_minus_d = fe_const(FE_A_MINUS_D);
  fe u1 = fe_mul(fe_add(p.x, p.t), fe_sub(p.x, p.t));                       // :97
  fe v;
  (void)fe_sqrt_ratio_zeta<true>(T, pt, fe_zero(), fe_mul_strict(fe_mul(u1, a_minus_d), fe_sqr(p.x)), &v, false, inv_den);  // :101
  fe u2 = fe_abs(fe_mul(v, u1));                                            // :104
  fe u3 = fe_sub(fe_mul(u2, p.z), p.t);                                     // :107
  fe s = fe_mul(fe_mul(fe_mul(a_minus_d, v), u3), p.x);                     // :110
  fe c = fe_canon(s);
  const bool neg = (c.l[0] & 1u) != 0;                                      // .abs()
  c = fe_select(neg, fe_canon_negate(c), c);
  fe_to_words(c, w);                                                        // top bits are 0: s < q < 2^253
}

// Element::elligator_map, src/ark_curve/elligator.rs:15-62.  r0 in Montgomery-261.
// First half: the point (s, t) of the Jacobi quartic t^2 = (1 + a s^2)^2 - 4 d s^2 (elligator.rs:20-45).
// the argument of the Elligator map's square root, num * den (elligator.rs:20-26), as a strict product
D377_HD fe ge_elligator_den(const fe& r0) {
  const fe one = fe_const(FE_ONE), dma = fe_const(FE_D_MINUS_A), dd = fe_const(FE_D);
  fe r = fe_mul(fe_const(FE_ZETA), fe_sqr(r0));
  fe den = fe_mul(fe_sub(fe_mul(dd, r), dma), fe_sub(fe_mul(dma, r), dd));
  return fe_mul_strict(fe_mul(fe_add(r, one), fe_const(FE_A_MINUS_2D)), den);
}
template <class PT>
D377_HD void ge_elligator_st(const SqrtTables& T, PT& pt, const fe& r0, fe* s_out, fe* t_out, const fe* inv_den = nullptr,
                             bool use_inv = true) {
  const fe one = fe_const(FE_ONE), dma = fe_const(FE_D_MINUS_A), dd = fe_const(FE_D);
  fe r = fe_mul(fe_const(FE_ZETA), fe_sqr(r0));                                   // :20
  fe den = fe_mul(fe_sub(fe_mul(dd, r), dma), fe_sub(fe_mul(dma, r), dd));        // :22
  fe num = fe_mul(fe_add(r, one), fe_const(FE_A_MINUS_2D));                       // :23
  fe isri;
  const bool iss = fe_sqrt_ratio_zeta<true>(T, pt, fe_zero(), fe_mul_strict(num, den), &isri, false, inv_den, use_inv);   // :25-26
  isri = fe_select(iss, isri, fe_mul(isri, r0));                                  // twiddle, :28-38
  fe s = fe_mul(isri, num);                                                       // :40
  fe p = fe_mul(fe_mul(fe_mul(isri, s), fe_sub(r, one)), fe_const(FE_A_MINUS_2D_SQ));
  *t_out = fe_sub(fe_select(iss, fe_neg(p), p), one);                             // :41  (-sgn * ... - 1)
  if (fe_is_negative(s) == iss) s = fe_neg(s);                                    // :43-45
  *s_out = s;
}
// Second half: the isogeny to the Edwards curve (elligator.rs:48-59)
D377_HD ge ge_from_jacobi_st(const fe& s, const fe& t) {
  const fe one = fe_const(FE_ONE);
  fe e = fe_dbl(s);                                                               // :48
  fe ss = fe_sqr(s);
  fe f = fe_sub(one, ss);                                                         // 1 + a s^2, a = -1
  fe g = fe_add(one, ss);                                                         // 1 - a s^2
  ge o;
  o.x = fe_mul(e, t); o.y = fe_mul(f, g); o.z = fe_mul(f, t); o.t = fe_mul(e, g); // :52-54
  return o;
}
template <class PT>
D377_HD ge ge_elligator_map(const SqrtTables& T, PT& pt, const fe& r0, const fe* inv_den = nullptr, bool use_inv = true) {
  fe s, t;
  ge_elligator_st(T, pt, r0, &s, &t, inv_den, use_inv);
  ge o = ge_from_jacobi_st(s, t);
  D377_INVARIANT(T, o, true);                                                     // :56-59
  return o;
}

// ---- scalars -------------------------------------------------------------------------------
// Fr::from_le_bytes_mod_order for 32 bytes (src/fields/fr.rs:82-94): k mod r, r ~ 2^250.2,
// so at most 2^256/r < 54 subtractions: do it by conditional subtraction of 32r..r.
D377_HD void fr_reduce_words(uint32_t k[8]) {
#pragma unroll
  for (int sh = 5; sh >= 0; --sh) {
    uint32_t d[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // (r << sh) word i
      uint32_t lo = FR_ORDER_W_LIT[i] << sh;
      if (sh > 0 && i > 0) lo |= FR_ORDER_W_LIT[i - 1] >> (32 - sh);
      uint64_t t = (uint64_t)k[i] - lo - borrow;
      d[i] = (uint32_t)t;
      borrow = (t >> 63) & 1u;
    }
    // r << 5 still fits in 256 bits (r < 2^251)
    if (borrow == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) k[i] = d[i];
    }
  }
  // after subtracting multiples 32r..r once each the value may still be >= r only if it
  // started >= 63r, impossible for a 256-bit value (2^256 / r < 54)
}

// ---- scalar-field arithmetic on canonical values (eight 32-bit words, < r) --------------------
// Fr add / sub / mul / neg / square / inverse: src/fields/fr/u64/wrapper.rs:76-108 (-> ark-ff MontBackend).
// Scalars cross the boundary as canonical 32-byte strings (Fr::to_bytes_le, :63-70), so these take and return
// canonical words; the Montgomery form (R = 2^256, word-level CIOS) is internal to a product.
D377_HD void fr_const_words(const uint32_t (&c)[8], uint32_t out[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = c[i];
}
D377_HD void fr_cond_sub_r(uint32_t a[8], uint32_t top) {           // a + top * 2^256 < 2r  ->  a mod r
  uint32_t d[8];
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t t = (uint64_t)a[i] - FR_ORDER_W_LIT[i] - borrow;
    d[i] = (uint32_t)t;
    borrow = (t >> 63) & 1u;
  }
  if (top != 0 || borrow == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = d[i];
  }
}
D377_HD void fr_addmod(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c += (uint64_t)a[i] + b[i]; out[i] = (uint32_t)c; c >>= 32; }
  fr_cond_sub_r(out, (uint32_t)c);                                   // a, b < r < 2^251: no carry out, kept for clarity
}
D377_HD void fr_submod(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t t = (uint64_t)a[i] - b[i] - borrow;
    out[i] = (uint32_t)t;
    borrow = (t >> 63) & 1u;
  }
  if (borrow) {                                                      // a < b: add r back
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { c += (uint64_t)out[i] + FR_ORDER_W_LIT[i]; out[i] = (uint32_t)c; c >>= 32; }
  }
}
// a * b / 2^256 mod r for a, b < r: CIOS, one word of b per round; the running value stays below 2r
D377_HD void fr_montmul(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c += (uint64_t)a[j] * b[i] + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (uint32_t)c;
    t[9] = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * FR_NINV32;
    c = ((uint64_t)m * FR_ORDER_W_LIT[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      c += (uint64_t)m * FR_ORDER_W_LIT[j] + t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (uint32_t)c;
    t[8] = t[9] + (uint32_t)(c >> 32);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = t[i];
  fr_cond_sub_r(out, t[8]);
}
D377_HD void fr_mulmod(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
  uint32_t t[8], r2[8];
  fr_const_words(FR_R2_W_LIT, r2);
  fr_montmul(a, b, t);                                               // ab / R
  fr_montmul(t, r2, out);                                            // ab
}
// a^(r-2); returns false (and zero) for a = 0, as Fr::inverse returns None (wrapper.rs:80-86)
D377_HD bool fr_invmod(const uint32_t a[8], uint32_t out[8]) {
  uint32_t x[8], acc[8], c[8];
  fr_const_words(FR_R2_W_LIT, c);
  fr_montmul(a, c, x);                                               // a R
  fr_const_words(FR_R1_W_LIT, acc);                                  // 1 R
#pragma unroll 1
  for (int i = 250; i >= 0; --i) {                                   // r - 2 < 2^251
    fr_montmul(acc, acc, acc);
    if ((FR_ORDER_MINUS_2_W_LIT[i >> 5] >> (i & 31)) & 1u) fr_montmul(acc, x, acc);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = i == 0 ? 1u : 0u;
  fr_montmul(acc, c, out);                                           // out of Montgomery form
  uint32_t nz = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) nz |= a[i];
  return nz != 0;
}
// Fr::from_le_bytes_mod_order for 33..64 bytes (src/fields/fr.rs:82-94): lo + hi * 2^256 mod r
D377_HD void fr_from_wide_words(const uint32_t lo_in[8], const uint32_t hi_in[8], uint32_t out[8]) {
  uint32_t lo[8], hi[8], r2[8], t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { lo[i] = lo_in[i]; hi[i] = hi_in[i]; }
  fr_reduce_words(lo);
  fr_reduce_words(hi);
  fr_const_words(FR_R2_W_LIT, r2);
  fr_montmul(hi, r2, t);                                             // hi * 2^256
  fr_addmod(lo, t, out);
}

// signed radix-16 recoding of k < r < 2^251: k = sum d_i 16^i, d_i in [-8, 8), i = 0..63.
// Packed as 64 nibbles (two's complement 4-bit) in 8 words.
D377_HD void fr_recode_signed16(const uint32_t k[8], uint32_t digits[8]) {
  uint32_t carry = 0;
#pragma unroll
  for (int wi = 0; wi < 8; ++wi) {
    uint32_t outw = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t d = ((k[wi] >> (4 * j)) & 15u) + carry;
      carry = (d >= 8u) ? 1u : 0u;          // d in 0..16; >= 8 means digit d-16 and carry
      outw |= (d & 15u) << (4 * j);
    }
    digits[wi] = outw;
  }
  // k < 2^251 -> nibble 62 is <= 7 + carry, nibble 63 is the final carry (0 or 1): no overflow
}
D377_HD int fr_digit(const uint32_t digits[8], int i) {   // signed value of nibble i
  uint32_t n = (digits[i >> 3] >> (4 * (i & 7))) & 15u;
  return (int)(n ^ 8u) - 8;
}

// ---- fast group formulas for the scalar-multiplication loops -------------------------------
// Same group law as ge_add / ge_double above (src/min_curve/element.rs:291-322, 119-136), with
// the usual savings: T is only produced when the next operation consumes it, table entries are
// kept in "cached" form (Y+X, Y-X, 2Z, 2d*T), and the doubling is sign-folded so it needs two
// offset subtractions instead of four (with G' = A-B, H' = A+B, F' = G'+C, E' = H'-S:
// X3 = E'F', Y3 = G'H', Z3 = F'G', T3 = E'H').
struct gec { fe ypx, ymx, z2, kt; };      // cached extended point

// ---- per-lane arithmetic of the four-lane group operations (quad_ops.hpp) -------------------------
// Between its two rounds of products a lane of the quad forms ONE linear combination of the round-one products -- the
// value its role contributes to round two -- and the second round's operands are quad permutations of those values (no
// lane computes what another lane uses).  The arithmetic lives here, host-compilable, so that the bounds build walks
// it (tests/host_sim: sim_quad_forms); the cross-lane moves are quad_ops.hpp's.
// doubling: the first round gives A = X^2, B = Y^2, Z^2 and T Z (= X Y: the quad holds T, so lane 3 multiplies its own
// coordinate like the others).  role 0: G' = A - B, 1: H' = A + B, 2: F' = A - B + 2 Z^2, 3: E = 2 T Z.
// u = (A, A, A, TZ), v = (B, B, B, TZ), c = (*, *, Z^2, *)
D377_HD fe gq_double_own(int role, const fe& u, const fe& v, const fe& c) {
  const fe t2 = fe_select((role & 1) != 0, v, fe_neg_nc(v));
  const fe t3 = fe_select(role == 2, c, fe_zero());
  return fe_carry(fe_add(fe_add(u, t2), fe_dbl(t3)));
}
// addition, first round's own operand: role 0: Y - X, 1: Y + X, 2: T, 3: 2Z.   u = (Y, Y, T, Z), v = (X, X, *, Z)
D377_HD fe gq_add_in_own(int role, const fe& u, const fe& v) {
  const fe t2 = fe_select(role == 2, fe_zero(), fe_select(role == 0, fe_neg_nc(v), v));
  return fe_carry(fe_add(u, t2));
}
// addition, between the rounds: u - v or u + v.   role 0: E = b - a, 1: H = b + a, 2: F = d -+ c, 3: G = d +- c
D377_HD fe gq_add_own(bool sub, const fe& u, const fe& v) { return fe_carry(fe_add(u, fe_select(sub, fe_neg_nc(v), v))); }



D377_HD ge ge_double_fast(const ge& p, bool with_t) {
  fe a = fe_sqr(p.x), b = fe_sqr(p.y);
  fe c = fe_sqr2x(p.z);                   // 2 Z^2 for the price of Z^2
  fe s_ = fe_sqr(fe_add(p.x, p.y));
  fe h = fe_add(a, b);                    // H' lazy
  fe e = fe_sub(h, s_);                   // E' = A + B - (X+Y)^2, carried
  fe g = fe_sub(a, b);                    // G' carried
  fe f = fe_add(g, c);                    // F' = G' + 2Z^2, lazy
  ge r;
  r.x = fe_mul(e, f); r.y = fe_mul(g, h); r.z = fe_mul(f, g);
  r.t = p.t;
  if (with_t) r.t = fe_mul(e, h);
  return r;
}

// -[2]P: the doubling of the scalar-multiplication loops.  E = 2XY is taken as the product
// X * (2Y) with its true sign while F', G', H' are the sign-folded (negated) forms, so the
// result is (-X3, Y3, Z3, -T3) = -[2]P -- and the (X+Y)^2 squaring with its offset subtraction
// and carry pass (222 instructions) becomes one addition and one product (205).  An even number
// of these in a row (the 4 per window) restores the sign.
D377_HD ge ge_double_neg(const ge& p, bool with_t) {
  fe a = fe_sqr(p.x), b = fe_sqr(p.y);
  fe c = fe_sqr2x(p.z);
  fe e = fe_mul(p.x, fe_dbl(p.y));        // 2XY
  fe h = fe_add(a, b);                    // lazy
  fe g = fe_sub(a, b);                    // carried
  fe f = fe_add(g, c);                    // lazy
  ge r;
  r.x = fe_mul(e, f); r.y = fe_mul(g, h); r.z = fe_mul(f, g);
  r.t = p.t;
  if (with_t) r.t = fe_mul(e, h);
  return r;
}

// one lane running alone (the MSM's Horner tail): same formulas; the hand-written multiplier has no
// pad instructions, so a lone wave issues it as fast as it can issue anything (~4.6 cycles each)
D377_HD ge ge_double_latency(const ge& p) { return ge_double_fast(p, true); }

D377_HD gec ge_to_cached(const ge& p) {
  gec c;
  c.ypx = fe_carry(fe_add(p.y, p.x));     // carried, like ymx: a negative digit swaps the two
  c.ymx = fe_sub(p.y, p.x);               // carried
  c.z2 = fe_dbl(p.z);                     // lazy
  c.kt = fe_mul(fe_const(FE_K), p.t);
  return c;
}

// p + (neg ? -q : q).  The caller has already swapped q.ypx / q.ymx for a negative digit (the
// table loader does it by address); the sign of 2dT is applied here by swapping F and G.
// p's coordinates are products; q.ymx is carried, so Y - X needs no carry pass before it.
D377_HD ge ge_add_cached(const ge& p, const gec& q, bool neg, bool with_t) {
  fe a = fe_mul(fe_sub_nc(p.y, p.x), q.ymx);
  fe b = fe_mul(fe_add(p.y, p.x), q.ypx);
  fe c = fe_mul(p.t, q.kt);
  fe d = fe_mul(p.z, q.z2);
  fe e = fe_sub(b, a), h = fe_add(b, a);
  fe dmc = fe_sub(d, c), dpc = fe_add(d, c);
  fe f = fe_select(neg, dpc, dmc), g = fe_select(neg, dmc, dpc);
  ge r;
  r.x = fe_mul(e, f); r.y = fe_mul(g, h); r.z = fe_mul(f, g);
  r.t = p.t;
  if (with_t) r.t = fe_mul(e, h);
  return r;
}

// affine cached entry (Z = 1): fixed-base table, 6 M (+1 for T)
struct gea { fe ypx, ymx, kt; };
D377_HD ge ge_add_affine(const ge& p, const gea& q, bool neg, bool with_t) {
  fe a = fe_mul(fe_sub_nc(p.y, p.x), q.ymx);
  fe b = fe_mul(fe_add(p.y, p.x), q.ypx);
  fe c = fe_mul(p.t, q.kt);
  fe d = fe_dbl(p.z);                     // lazy
  fe e = fe_sub(b, a), h = fe_add(b, a);
  fe dmc = fe_sub(d, c), dpc = fe_carry(fe_add(d, c));
  fe f = fe_select(neg, dpc, dmc), g = fe_select(neg, dmc, dpc);
  ge r;
  r.x = fe_mul(e, f); r.y = fe_mul(g, h); r.z = fe_mul(f, g);
  r.t = p.t;
  if (with_t) r.t = fe_mul(e, h);
  return r;
}

// ---- cached affine records (the MSM's point records, the fixed-base comb) -----------------------
// affine (x, y) -> cached affine; the identity (0, 1) for a point that contributes nothing
D377_HD gea gea_from_affine(const fe& x, const fe& y) {
  gea c;
  c.ypx = fe_carry(fe_add(y, x));
  c.ymx = fe_sub(y, x);
  c.kt = fe_mul(fe_mul(fe_const(FE_K), x), y);
  return c;
}
// The point of a cached affine record in extended coordinates with Z = 2: X = 2x = (y+x) - (y-x), Y = 2y, T = X Y / Z =
// 2xy = (2/K) * (K x y): the first point of a run costs 4 products instead of a 7-product addition to the identity.
// The record's y-x is a carried difference (value up to 41q, beyond what a subtrahend may be: fq29.hpp), so it is
// brought down by a product with 1 first, and X, Y leave as products like the coordinates every group formula expects.
// (swap / neg: the record of -P has y+x and y-x exchanged by the loader, which negates X; T follows.)
D377_HD ge ge_from_cached_affine(const gea& q, bool neg) {
  const fe one = fe_const(FE_ONE);
  ge r;
  r.x = fe_mul(fe_sub(q.ypx, fe_mul(q.ymx, one)), one);
  r.y = fe_mul(fe_add(q.ypx, q.ymx), one);
  r.z = fe_const(FE_TWO);
  const fe t = fe_mul(q.kt, fe_const(FE_2_OVER_K));
  r.t = fe_select(neg, fe_neg(t), t);
  return r;
}

// ---- scalar multiplication ----------------------------------------------------------------
// [k]P with signed 4-bit windows, MSB first: 63 x (4 doublings + 1 cached addition).
// `Tab` holds the per-lane table of cached 0..8 * P (global scratch on the GPU): it provides
// store(j, gec) and load(j, swap) -> gec, where swap exchanges ypx / ymx (negative digit).
// want_t: whether the caller uses T of the result (the square-root-free compressor does not).
template <class Tab>
D377_HD ge ge_scalar_mul_w4(const ge& p, const uint32_t digits[8], Tab& tab, bool want_t = true) {
  {
    gec id;
    id.ypx = fe_const(FE_ONE); id.ymx = fe_const(FE_ONE); id.z2 = fe_dbl(fe_const(FE_ONE)); id.kt = fe_zero();
    tab.store(0, id);
  }
  const gec pc = ge_to_cached(p);
  tab.store(1, pc);
  ge acc = ge_double_fast(p, true);
  tab.store(2, ge_to_cached(acc));
#pragma unroll 1
  for (int j = 3; j <= 8; ++j) {
    acc = ge_add_cached(acc, pc, false, true);
    tab.store(j, ge_to_cached(acc));
  }
  int d = fr_digit(digits, 63);                 // 0 or 1
  ge r = ge_select(d != 0, p, ge_identity());
#pragma unroll 1
  for (int i = 62; i >= 0; --i) {
    // fetch this window's table entry first: its ~1-2 us of memory latency hides under the
    // four doublings instead of stalling the addition
    d = fr_digit(digits, i);
    const bool neg = d < 0;
    const gec e = tab.load(neg ? -d : d, neg);
#pragma unroll 1
    for (int j = 0; j < 4; ++j) r = ge_double_neg(r, j == 3);   // (-2)^4 = 16
    r = ge_add_cached(r, e, neg, want_t && i == 0);   // only the last T can have a reader
  }
  return r;
}

// signed radix-256 recoding of k < 2^251: k = sum d_i 256^i, d_i in [-128, 128), i = 0..31
D377_HD void fr_recode_signed256(const uint32_t k[8], int digits[32]) {
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    uint32_t d = ((k[i >> 2] >> (8 * (i & 3))) & 255u) + carry;
    carry = (d >= 128u) ? 1u : 0u;
    digits[i] = (int)d - (int)(carry << 8);
  }
  // byte 31 of k is <= 7 (k < 2^251), so the last digit is <= 8 and there is no final carry
}

// [k]B from the shared table FB[i][j] = affine cached j * 2^(FB_BITS i) * B (i < FB_WINDOWS, j <= 2^(FB_BITS-1)):
// FB_WINDOWS mixed additions, no doublings.  FTab::load(i, j, swap) -> gea.  FB_BITS: 23 (11 windows x 4 194 305 entries,
// 5.9 GB in HBM: the default -- 2 % of the device's 288 GB), 21 (12 x 1 048 577, 1.6 GB), 18 (14 x 131 073, 235 MB), 16 (16 x
// 32 769, 67 MB: inside the Infinity Cache), 14, 12 or 8 (32 windows, 528 KB); the static_asserts below say what a width has
// to satisfy.  The kernel is instruction-bound and its gathers are covered (profiles/r05_ab_fixed_base_raw_gather.txt), so
// every addition a wider comb saves is time saved, cache or no cache: 12 / 14 / 18 bits 7.8 / 8.6 / 11.2e8 /s (rounds 2-3);
// 16 / 18: 18 ahead by 7-8 % (r05_fixed_base_ab.txt); 18 / 20 / 21 / 23 bits at 2^20: 953 / 942 / 920 / 864 us, at 2^22:
// 3579 / 3530 / 3370 / 3156 us (r05_ab_fixed_base_wide.txt).  The next step down in windows (10) would be 26 bits: 43 GB.
// The host simulation builds its tables with 12 or 8.
#ifndef D377_FB_BITS
#define D377_FB_BITS 23
#endif
// The comb's shape for a width of BITS.  The library instantiates its fixed-base kernels for FB_BITS (the build's default,
// what d377_ctx_create takes) and for 18 / 21 / 23, and a context picks one of them at run time (d377_ctx_create_ex:
// include/decaf377_amd.h); the host simulation builds its tables with 12 or 8.
// The top digit must take the recoding's carry without one of its own.  Scalars are < r and r >> 228 = 0x4aad95, so the
// top window's value is at most R_TOP >> (its first bit - 228); that + 1 has to stay below 2^(BITS - 1).  True for
// the widths that tile the 252 bits (18, 14, 12) and for ragged tops (8: 32 windows; 16: 16 windows, 11 bits in the last).
constexpr unsigned long long FB_R_TOP = 0x4aad95ull;                        // r >> 228
template <int BITS>
struct FbShape {
  static constexpr int bits = BITS;
  static constexpr int windows = (252 + BITS - 1) / BITS;
  static constexpr int entries = (1 << (BITS - 1)) + 1;
  static constexpr int top_bit = BITS * (windows - 1);
  static_assert(BITS >= 4 && BITS <= 23 && top_bit >= 228 && BITS * windows >= 252, "comb width");
  static_assert((FB_R_TOP >> (top_bit - 228)) + 1 < (1ull << (BITS - 1)), "the top digit of a scalar below r must not carry out");
  static_assert(windows >= 2, "the first window is peeled off the loop");
};
constexpr int FB_BITS = D377_FB_BITS;
constexpr int FB_WINDOWS = FbShape<FB_BITS>::windows;
constexpr int FB_ENTRIES = FbShape<FB_BITS>::entries;
// signed digit i of k (BITS wide), with the running carry of the recoding
template <int BITS = FB_BITS>
D377_HD int fb_digit(const uint32_t k[8], int i, uint32_t& carry) {
  const int bit = BITS * i, wi = bit >> 5, sh = bit & 31;
  uint32_t v = k[wi] >> sh;
  if (sh + BITS > 32 && wi + 1 < 8) v |= k[wi + 1] << (32 - sh);
  const uint32_t dd = (v & ((1u << BITS) - 1u)) + carry;
  carry = (dd >= (1u << (BITS - 1))) ? 1u : 0u;        // k < r < 2^251: the top digit never carries out
  return (int)dd - (int)(carry << BITS);
}
// want_t: whether the caller reads T of the result.  The entry of window i + 1 is fetched before the addition of
// window i: the table lives in L2 / Infinity Cache, and one mixed addition is only ~1 500 instructions.  (Two entries
// in flight instead of one: 1.07-1.11e9/s against 1.09-1.12e9/s at 2^20 and 2^22, same box -- the gathers are covered.)
template <int BITS = FB_BITS, class FTab>
D377_HD ge ge_scalar_mul_base_w8(const uint32_t k[8], const FTab& ftab, bool want_t = true) {
  constexpr int W = FbShape<BITS>::windows;
  uint32_t carry = 0;
  int d = fb_digit<BITS>(k, 0, carry);
  bool neg = d < 0;
  gea e = ftab.load(0, neg ? -d : d, neg);
  // window 0: the sum starts from the record itself (ge_from_cached_affine: 4 products) instead of a 7-product addition to
  // the identity -- 85 M + 3 S per scalar where the plain loop took 88 M + 3 S
  ge r;
  {
    const gea cur = e;
    const bool neg_cur = neg;
    d = fb_digit<BITS>(k, 1, carry);
    neg = d < 0;
    e = ftab.load(1, neg ? -d : d, neg);
    r = ge_from_cached_affine(cur, neg_cur);
  }
#pragma unroll 1
  for (int i = 1; i < W; ++i) {
    const gea cur = e;
    const bool neg_cur = neg;
    if (i + 1 < W) {
      d = fb_digit<BITS>(k, i + 1, carry);
      neg = d < 0;
      e = ftab.load(i + 1, neg ? -d : d, neg);
    }
    r = ge_add_affine(r, cur, neg_cur, want_t || i + 1 < W);
  }
  return r;
}

// generic x^e for a 256-bit exponent given as 8 words (init kernels only: inversion by q - 2)
D377_HD fe fe_pow_words(const fe& x, const uint32_t (&e)[8]) {
  fe r = fe_const(FE_ONE);
#pragma unroll 1
  for (int i = 255; i >= 0; --i) {
    r = fe_sqr(r);
    if ((e[i >> 5] >> (i & 31)) & 1u) r = fe_mul(r, x);
  }
  return r;
}
// x^(q-2), the plain ladder: kept as the cross-check of fe_invert (tests/host_sim)
D377_HD fe fe_invert_pow(const fe& x) {
  // q - 2: q's low word is 0x00000001, so the subtraction borrows from the next word
  const uint32_t ee[8] = {0xFFFFFFFFu, FQ_MODULUS_W_LIT[1] - 1u, FQ_MODULUS_W_LIT[2], FQ_MODULUS_W_LIT[3],
                          FQ_MODULUS_W_LIT[4], FQ_MODULUS_W_LIT[5], FQ_MODULUS_W_LIT[6], FQ_MODULUS_W_LIT[7]};
  return fe_pow_words(x, ee);
}

// x^(q-2) from the two fixed exponentiations of the square root: q - 2 = 2^47 (m - 1) + (2^47 - 1), so
// 1/x = (x^((m-1)/2))^(2^48) * x^(2^47 - 1): 296 S + 53 M against 256 S + ~128 M for the plain ladder.
// (Superseded by the divsteps inversion fe_invert, ~3x cheaper; kept as a second cross-check.)
template <class PT>
D377_HD fe fe_invert_chain(const fe& x, PT& pt) {
  return fe_mul(fe_sqr_n(fe_pow_m12(x, pt), 48), fe_pow_2_47_m1(x));
}

// k / 2 mod r for canonical k < r (r odd): (k + r) / 2 when k is odd
D377_HD void fr_half_words(uint32_t k[8]) {
  const uint32_t odd = k[0] & 1u;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c += (uint64_t)k[i] + (odd ? FR_ORDER_W_LIT[i] : 0u); k[i] = (uint32_t)c; c >>= 32; }   // < 2r < 2^252
#pragma unroll
  for (int i = 0; i < 8; ++i) k[i] = (k[i] >> 1) | (i < 7 ? (k[i + 1] << 31) : 0u);
}

// ---- compression without a square root, batched ------------------------------------------------
// Element::vartime_compress (src/ark_curve/encoding.rs:91-128) takes v = 1/sqrt(u1 (a-d) X^2), u1 = X^2 - T^2.
// When the point is known as an image of the decaf isogeny,
//     P = (E F : G H : F G : E H)   with   F^2 - H^2 = -(1 + d) E^2,
// that argument is the perfect square ((1 + d) E^3 F)^2 and the steps of encoding.rs:97-110 collapse to
//     s = | (sigma G - H) / E |,   sigma = +1 if E / F is non-negative, -1 otherwise
// (u2 = |v u1| = |E / F|, u3 = u2 Z - T = E (sigma G - H), s = |(a-d) v u3 X|; the two |.| absorb the sign of
// the root, so either root gives the reference's bytes).  Both producers on the hot path have that shape:
//   * a doubling P = [2]R, R = (X0 : Y0 : Z0):  E = 2 X0 Y0, G = Y0^2 - X0^2, H = -(X0^2 + Y0^2), F = G - 2 Z0^2,
//     so s = |Y0 / X0| or |X0 / Y0| -- a scalar multiplication [k]P is computed as [2]([k/2 mod r]P);
//   * the Elligator map (elligator.rs:48-54): (E, F, G, H) = (2s, t, 1 - s^2, 1 + s^2), so s_out = |s| or |1/s|.
// What is left is ONE inversion per element, and inversions batch (Montgomery's trick) where square roots do
// not: a lane collects the states of the DCB_K elements of its chunk and inverts once for all of them.
// State of one element: enc = | (is_negative(w / p) ? n1 : n0) / p |; p = 0 (the identity's X = 0, where the
// reference's sqrt_ratio_zeta(1, 0) returns 0 and the encoding is 0) and failed lanes are stored as
// p = 1, n0 = n1 = 0.  All four values are strict products (tight limbs, < 2q) and travel as 32-byte records.
#ifndef D377_DCB_K
#define D377_DCB_K 8
#endif
constexpr int DCB_K = D377_DCB_K;       // elements per lane per inversion in one generation of workgroups (measured under the arbiter's own
                                        // order: 4 / 8 / 16 -> 60.9 / 61.0 / 62.0 ms per 2^22 var-base, 8 best for the 2^20 operations);
                                        // longer launches: dcb.hpp DCB_K_LONG
#if defined(D377_CHECK_INVARIANTS)
constexpr bool DCB_WANT_T = true;       // the debug assertions re-check T Z = X Y on the half point
#else
constexpr bool DCB_WANT_T = false;      // the state of a doubling does not read T
#endif
struct dcb_state { fe p, w, n0, n1; };

D377_HD dcb_state dcb_neutral() {
  dcb_state st;
  st.p = fe_const(FE_ONE); st.w = fe_zero(); st.n0 = fe_zero(); st.n1 = fe_zero();
  return st;
}
D377_HD dcb_state dcb_guard_zero(dcb_state st, bool dead) {
  const bool z = dead || fe_strict_is_zero(st.p);
  const dcb_state nt = dcb_neutral();
  st.p = fe_select(z, nt.p, st.p); st.w = fe_select(z, nt.w, st.w);
  st.n0 = fe_select(z, nt.n0, st.n0); st.n1 = fe_select(z, nt.n1, st.n1);
  return st;
}
// state of [2]R (T of R is not used)
D377_HD dcb_state ge_dcb_from_half(const ge& r, bool dead) {
  const fe a = fe_sqr(r.x), b = fe_sqr(r.y), c = fe_sqr2x(r.z);
  const fe f = fe_sub(b, fe_add(a, c));                 // F = Y0^2 - X0^2 - 2 Z0^2
  const fe q = fe_mul(r.x, r.y);                        // E / 2
  dcb_state st;
  st.p = fe_mul_strict(q, f);                           // E F / 2
  st.w = fe_mul_strict(q, fe_dbl(q));                   // w / p = 2 q / F = E / F
  st.n0 = fe_mul_strict(b, f);                          // n0 / p = Y0^2 / (X0 Y0)
  st.n1 = fe_mul_strict(a, f);                          // n1 / p = X0^2 / (X0 Y0)
  return dcb_guard_zero(st, dead);
}
// state of the Elligator point of (s, t)
D377_HD dcb_state ge_dcb_from_jacobi_st(const fe& s_in, const fe& t) {
  const fe s = fe_mul_strict(s_in, fe_const(FE_ONE));   // same value below 1.1 q: keeps p below 2q for the zero test
  dcb_state st;
  st.p = fe_mul_strict(s, t);                           // E F / 2
  st.w = fe_mul_strict(s, fe_dbl(s));                   // w / p = 2 s / t
  st.n0 = fe_mul_strict(s, st.p);                       // n0 / p = s
  st.n1 = fe_mul_strict(t, fe_const(FE_ONE));           // n1 / p = 1 / s
  return dcb_guard_zero(st, false);
}
// State of the SUM of two Elligator points, from their Jacobi-quartic preimages (hash_to_curve, elligator.rs:67-71).
// The decaf isogeny J -> E, (s, t) -> (2s t : (1 - s^2)(1 + s^2) : (1 - s^2) t : 2s (1 + s^2)), is a homomorphism, so
// map(r1) + map(r2) = image of (s1, t1) + (s2, t2) ADDED ON THE QUARTIC  J: t^2 = s^4 - 2 delta s^2 + 1, delta = 1 + 2d
// (a = -1), and a point with a known preimage encodes without a square root (above): the reference's third square root
// of hash_to_curve -- the compression of a sum of two Edwards points -- disappears.  Addition law of a Jacobi quartic in
// this form, kept projective (s3 = X / Z, t3 = Y / Z^2):
//     X = s1 t2 + t1 s2,   Z = 1 - (s1 s2)^2,   Y = (1 + (s1 s2)^2)(t1 t2 - 2 delta s1 s2) + 2 s1 s2 (s1^2 + s2^2);
// with s = X / Z, t = Y / Z^2 the state (p, w, n0, n1) of ge_dcb_from_jacobi_st, cleared of denominators by Z^3, is
//     p = X Y Z,  w = 2 (X Z)^2,  n0 = X^2 Y,  n1 = Y Z^2          (w / p = 2s / t, n0 / p = s, n1 / p = 1 / s).
// X = 0 or Y = 0 is the identity class (encoding 0, as the p = 0 rule gives).  Z = 0 (s1 s2 = +-1) is the one exceptional
// case of the law -- the sum has no finite (s, t) and X, Y may vanish with it -- reported through *exceptional: the caller
// takes the reference's route (Edwards addition, compression with its square root) for such a pair.  Checked against the
// big-integer model for random pairs, doubling, opposite points and constructed exceptional pairs
// (tests/test_host_sim.py::test_hash_to_curve_on_the_quartic).  8 M + 3 S for the sum, 5 M + 3 S for the state.
D377_HD dcb_state ge_dcb_from_jacobi_sum(const fe& s1, const fe& t1, const fe& s2, const fe& t2, bool* exceptional) {
  const fe one = fe_const(FE_ONE);
  const fe xx = fe_mul(s1, s2);
  const fe xx2 = fe_sqr_strict(xx);                                 // strict: Z = 1 - xx2 feeds a zero test
  const fe x = fe_mul(fe_add(fe_mul(s1, t2), fe_mul(t1, s2)), one); // X (a product again: it is squared and multiplied below)
  const fe z = fe_mul_strict(fe_sub(one, xx2), one);                // Z, value below 2q
  *exceptional = fe_strict_is_zero(z);
  const fe inner = fe_sub(fe_mul(t1, t2), fe_mul(fe_const(FE_2_PLUS_4D), xx));
  const fe outer = fe_mul(fe_dbl(xx), fe_add(fe_sqr(s1), fe_sqr(s2)));
  const fe y = fe_mul(fe_add(fe_mul(fe_add(one, xx2), inner), outer), one);   // Y
  const fe xz = fe_mul(x, z), x2 = fe_sqr(x), z2 = fe_sqr(z);
  dcb_state st;
  st.p = fe_mul_strict(xz, y);
  st.w = fe_mul_strict(xz, fe_dbl(xz));
  st.n0 = fe_mul_strict(x2, y);
  st.n1 = fe_mul_strict(z2, y);
  return dcb_guard_zero(st, false);
}
// a finished encoding (canonical words of s) as a state: p = 1, w = 0 (non-negative: n0 is taken), n0 = s
D377_HD dcb_state dcb_from_encoding_words(const uint32_t w[8]) {
  dcb_state st = dcb_neutral();
  st.n0 = fe_mul_strict(fe_from_words(w), fe_const(FE_R2));
  return st;
}
// IO: where a lane keeps the states of its current round and where results go --
//   get(slot, j, w) / put(slot, j, w): 32-byte record `slot` (0 p, 1 w, 2 n0, 3 n1) of the round's j-th element;
//   park(j, w) / parked(j, w): a 32-byte place per element that is free until its result is written (the output
//   record itself); emit(j, w): the element's encoding.
template <class IO>
D377_HD void dcb_put(IO& io, int j, const dcb_state& st) {
  uint32_t w[8];
  fe_to_words(st.p, w); io.put(0, j, w);
  fe_to_words(st.w, w); io.put(1, j, w);
  fe_to_words(st.n0, w); io.put(2, j, w);
  fe_to_words(st.n1, w); io.put(3, j, w);
}
// The denominators of a round's square roots, inverted together (see fe_sqrt_ratio_zeta, inv_den): records `slot`
// hold strict products x_j (a zero is stored as 1: its lane takes the den = 0 exit of the square root whatever the
// inverse says) and are replaced by 1 / x_j; records `tmp` hold the exclusive prefix products meanwhile -- not the
// output records, which an in-place caller still needs as inputs at this point.
constexpr int DCB_TMP_SLOT = 4;
template <class IO>
D377_HD void dcb_put_den(IO& io, int slot, int j, const fe& den) {
  uint32_t w[8];
  fe_to_words(fe_select(fe_strict_is_zero(den), fe_const(FE_ONE), den), w);
  io.put(slot, j, w);
}
template <class IO>
D377_HD fe dcb_get_inv(const IO& io, int slot, int j) {
  uint32_t w[8];
  io.get(slot, j, w);
  return fe_from_words(w);
}
// `invert`: fe -> fe, the inversion of the lane's product.  A lane without elements (cnt = 0) still takes part with the
// product 1: the device passes an inversion that the whole wave does together (row_ops.hpp fe_invert_lanes).
template <class IO, class INV>
D377_HD void dcb_invert_slot_with(IO& io, int slot, int cnt, INV invert) {
  uint32_t w[8];
  fe c = fe_const(FE_ONE);
#pragma unroll 1
  for (int j = 0; j < cnt; ++j) {
    fe_to_words(c, w);
    io.put(DCB_TMP_SLOT, j, w);
    io.get(slot, j, w);
    c = fe_mul_strict(c, fe_from_words(w));
  }
  fe inv = invert(c);
#pragma unroll 1
  for (int j = cnt - 1; j >= 0; --j) {
    io.get(DCB_TMP_SLOT, j, w);
    const fe inv_j = fe_mul_strict(inv, fe_from_words(w));
    io.get(slot, j, w);
    inv = fe_mul(inv, fe_from_words(w));
    fe_to_words(inv_j, w);
    io.put(slot, j, w);
  }
}
template <class IO>
D377_HD void dcb_invert_slot(IO& io, int slot, int cnt) {
  if (cnt <= 0) return;
  dcb_invert_slot_with(io, slot, cnt, [](const fe& c) { return fe_invert(c); });
}

// (Fetching the next element's records ahead of the current one's work, here and in the round driver, was measured
// and bought nothing: the waits are already covered.)
// `invert`: fe -> fe, the inversion of the one product (fe_invert in a lane; the one-wave kernels pass the whole wave's,
// row_ops.hpp fe_invert_wave)
template <class IO, class INV>
D377_HD void dcb_finish_with(IO& io, int cnt, INV invert) {       // (cnt = 0: the lane only takes part in `invert`)
  uint32_t w[8];
  fe c = fe_const(FE_ONE);
#pragma unroll 1
  for (int j = 0; j < cnt; ++j) {                       // prefix products, parked in the output records
    fe_to_words(c, w);
    io.park(j, w);
    io.get(0, j, w);
    c = fe_mul_strict(c, fe_from_words(w));
  }
  fe inv = invert(c);
#pragma unroll 1
  for (int j = cnt - 1; j >= 0; --j) {
    io.parked(j, w);
    const fe inv_j = fe_mul(inv, fe_from_words(w));     // 1 / p_j
    io.get(0, j, w);
    inv = fe_mul(inv, fe_from_words(w));
    io.get(1, j, w);
    const bool neg = fe_is_negative(fe_mul(fe_from_words(w), inv_j));     // sign of E / F (encoding.rs:104)
    io.get(neg ? 3 : 2, j, w);
    fe s = fe_canon(fe_mul(fe_from_words(w), inv_j));
    s = fe_select((s.l[0] & 1u) != 0, fe_canon_negate(s), s);             // .abs(), encoding.rs:110
    fe_to_words(s, w);
    io.emit(j, w);
  }
}
template <class PT, class IO>
D377_HD void dcb_finish(PT& pt, IO& io, int cnt) {
  (void)pt;
  if (cnt <= 0) return;
  dcb_finish_with(io, cnt, [](const fe& c) { return fe_invert(c); });
}
// one element's encoding from its state and 1 / p (what the loop above does per element, on values instead of records)
D377_HD void dcb_encode_one(const dcb_state& st, const fe& inv_p, uint32_t w[8]) {
  const bool neg = fe_is_negative(fe_mul(st.w, inv_p));                   // sign of E / F (encoding.rs:104)
  fe s = fe_canon(fe_mul(fe_select(neg, st.n1, st.n0), inv_p));
  s = fe_select((s.l[0] & 1u) != 0, fe_canon_negate(s), s);               // .abs(), encoding.rs:110
  fe_to_words(s, w);
}

}  // namespace d377
