// fq29.hpp -- Fq (BLS12-377 scalar field, the decaf377 base field) for CDNA4 lanes.
//
// One field element = 9 limbs of 29 bits in 9 VGPRs, Montgomery form with R = 2^261.
// Why this shape (measured on MI355X, profiles/r01_valu_rates_microbench.txt):
//   * v_mad_u64_u32 issues at the same rate as every other 3-operand VALU op (~4.8 cyc per
//     wave-instruction), while each carry instruction (v_addc_co_u32) costs as much as a MAC.
//     With 29-bit limbs a whole product column (<= 9 a_i*b_j + 8 m_i*q_j terms < 2^64)
//     accumulates in ONE 64-bit register pair by chained v_mad_u64_u32 with no carry
//     instruction at all; saturated 32-bit limbs would need one v_addc per MAC.
//   * q = 1 (mod 2^47), so -q^-1 = -1 (mod 2^29): the Montgomery digit is m = -t (mod 2^29),
//     no multiplication, and q's limb 0 is 1.
//   * 9*29 = 261 bits leaves 8 spare bits above q (253 bits): products of values up to
//     ~16q come out < 2q with no final conditional subtraction, and additions are lazy.
//
// Semantics follow the reference's Fq (src/fields/fq/u64/wrapper.rs:99-132: add, sub, mul,
// square, neg; src/fields/fq.rs:90-115: byte I/O; src/sign.rs:19-23: sign) -- identical
// field values, different internal representation.  Conversions to the reference's
// 4 x u64 Montgomery (R = 2^256) limbs are at the API boundary (fe_to_mont256 / fe_from_mont256).
//
// Representation contract
//   "tight":  limbs 0..7 < 2^29 + 16, limb 8 small; value < 2^257.
//   "lazy":   limbs < 2^30 + 32 (sum of two tights).
//   fe_mul / fe_sqr accept lazy operands (9 * 2^60.1 + 8 * 2^58 + carry < 2^64) and return
//   tight limbs with value < a*b/2^261 + q  (< 1.01q for operands < 2q, < 2q for
//   operands up to 20q).
//   fe_add is lazy (no carry).  fe_sub adds 8q in a borrow-proof digit form and then runs one
//   carry pass, so its result is tight (value < a + 8q).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define D377_HD __device__ __forceinline__
#define D377_CONST static __device__ __constant__ const
#else
#define D377_HD inline __attribute__((always_inline))
#define D377_CONST static const
#endif

namespace d377 {

constexpr int NL = 9;
constexpr int RB = 29;
constexpr uint32_t MASK29 = (1u << RB) - 1u;

struct fe { uint32_t l[NL]; };

// q in radix 2^29 as literals: the compiler keeps them in SGPRs (one s_mov each).
constexpr uint32_t QL[NL] = {0x00000001u, 0x108c0000u, 0x00000042u, 0x14edfda0u, 0x1b00159au,
                             0x068f2e1bu, 0x155982d1u, 0x0bd34594u, 0x0012ab65u};
// 8q with limbs 0..7 in [2^30 + 64, 2^31): a + C - b is borrow-free limb by limb for lazy b
// (tools/gen_constants.py sub_offset; tests check it equals 8q).
constexpr uint32_t SUB8Q[NL] = {0x60000008u, 0x445ffffdu, 0x40000212u, 0x476fecfeu, 0x5800acd3u,
                                0x547970dcu, 0x4acc1687u, 0x5e9a2ca3u, 0x00955b28u};

// One column = ONE strictly sequential chain of v_mad_u64_u32 into a single 64-bit pair.  Left
// alone, hipcc splits a column into several partial sums and merges them with v_lshl_add_u64
// (25 extra VOP3 ops per squaring, 20 more live VGPRs); the empty asm pins the accumulator so the
// chain stays sequential.  Measured (tools/field_bench.hip, MI355X, 2 / 4 / 8 waves per SIMD):
// squaring 820 / 772 / 756 -> 776 / 745 / 717 cycles, multiplication 926 / 889 / 873 -> 893 / 867 / 839.
#if defined(__HIP_DEVICE_COMPILE__)
#define D377_PIN(x) asm volatile("" : "+v"(x))
#else
#define D377_PIN(x) ((void)0)
#endif
D377_HD uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
  uint64_t r = (uint64_t)a * b + c;
  D377_PIN(r);
  return r;
}

// Montgomery product a*b/2^261 mod q, column-wise (product scanning) with the reduction
// interleaved: column k gets sum a_i*b_{k-i} + sum m_i*q_{k-i}, one 64-bit accumulator.
D377_HD fe fe_mul(const fe& a, const fe& b) {
  uint64_t acc = 0;
  uint32_t m[NL];
  fe r;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc = mad64(m[i], QL[k - i], acc);
    m[k] = (0u - (uint32_t)acc) & MASK29;   // -q^-1 = -1 mod 2^29
    acc = mad64(m[k], 1u, acc);             // m_k * q_0, q_0 = 1: low 29 bits become zero
    acc >>= RB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - (NL - 1); i < NL; ++i) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
    for (int i = k - (NL - 1); i < NL; ++i) acc = mad64(m[i], QL[k - i], acc);
    r.l[k - NL] = (uint32_t)acc & MASK29;
    acc >>= RB;
  }
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// Montgomery square: 45 limb products instead of 81 (off-diagonal terms use 2*a_i).
D377_HD fe fe_sqr(const fe& a) {
  uint64_t acc = 0;
  uint32_t m[NL], a2[NL];
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) a2[i] = a.l[i] << 1;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc = mad64(a2[i], a.l[k - i], acc);
    if ((k & 1) == 0) acc = mad64(a.l[k / 2], a.l[k / 2], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc = mad64(m[i], QL[k - i], acc);
    m[k] = (0u - (uint32_t)acc) & MASK29;
    acc = mad64(m[k], 1u, acc);
    acc >>= RB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - (NL - 1); 2 * i < k; ++i) acc = mad64(a2[i], a.l[k - i], acc);
    if ((k & 1) == 0) acc = mad64(a.l[k / 2], a.l[k / 2], acc);
#pragma unroll
    for (int i = k - (NL - 1); i < NL; ++i) acc = mad64(m[i], QL[k - i], acc);
    r.l[k - NL] = (uint32_t)acc & MASK29;
    acc >>= RB;
  }
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// Two independent squarings with their column chains interleaved MAC by MAC: each chain's
// dependent v_mad_u64_u32 then has the other chain's MAC between itself and its predecessor, so
// a wave does not stall on its own accumulator (tools/field_bench.hip: 744 vs 795 cycles per
// squaring at 2 waves per SIMD, 823 vs 1217 for a lone wave).  Used where one lane runs alone
// (the MSM's Horner tail); in the throughput kernels it made no measurable difference.
D377_HD void fe_sqr2(const fe& a, const fe& b, fe& ra, fe& rb) {
  uint64_t acc = 0, bcc = 0;
  uint32_t m[NL], n[NL], a2[NL], b2[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) { a2[i] = a.l[i] << 1; b2[i] = b.l[i] << 1; }
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) { acc = mad64(a2[i], a.l[k - i], acc); bcc = mad64(b2[i], b.l[k - i], bcc); }
    if ((k & 1) == 0) { acc = mad64(a.l[k / 2], a.l[k / 2], acc); bcc = mad64(b.l[k / 2], b.l[k / 2], bcc); }
#pragma unroll
    for (int i = 0; i < k; ++i) { acc = mad64(m[i], QL[k - i], acc); bcc = mad64(n[i], QL[k - i], bcc); }
    m[k] = (0u - (uint32_t)acc) & MASK29; n[k] = (0u - (uint32_t)bcc) & MASK29;
    acc = mad64(m[k], 1u, acc); bcc = mad64(n[k], 1u, bcc);
    acc >>= RB; bcc >>= RB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - (NL - 1); 2 * i < k; ++i) { acc = mad64(a2[i], a.l[k - i], acc); bcc = mad64(b2[i], b.l[k - i], bcc); }
    if ((k & 1) == 0) { acc = mad64(a.l[k / 2], a.l[k / 2], acc); bcc = mad64(b.l[k / 2], b.l[k / 2], bcc); }
#pragma unroll
    for (int i = k - (NL - 1); i < NL; ++i) { acc = mad64(m[i], QL[k - i], acc); bcc = mad64(n[i], QL[k - i], bcc); }
    ra.l[k - NL] = (uint32_t)acc & MASK29; rb.l[k - NL] = (uint32_t)bcc & MASK29;
    acc >>= RB; bcc >>= RB;
  }
  ra.l[NL - 1] = (uint32_t)acc; rb.l[NL - 1] = (uint32_t)bcc;
}

// Two independent products a*b and c*d, chains interleaved (see fe_sqr2).
D377_HD void fe_mul2(const fe& a, const fe& b, const fe& c, const fe& d, fe& rab, fe& rcd) {
  uint64_t acc = 0, bcc = 0;
  uint32_t m[NL], n[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc = mad64(a.l[i], b.l[k - i], acc); bcc = mad64(c.l[i], d.l[k - i], bcc); }
#pragma unroll
    for (int i = 0; i < k; ++i) { acc = mad64(m[i], QL[k - i], acc); bcc = mad64(n[i], QL[k - i], bcc); }
    m[k] = (0u - (uint32_t)acc) & MASK29; n[k] = (0u - (uint32_t)bcc) & MASK29;
    acc = mad64(m[k], 1u, acc); bcc = mad64(n[k], 1u, bcc);
    acc >>= RB; bcc >>= RB;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - (NL - 1); i < NL; ++i) { acc = mad64(a.l[i], b.l[k - i], acc); bcc = mad64(c.l[i], d.l[k - i], bcc); }
#pragma unroll
    for (int i = k - (NL - 1); i < NL; ++i) { acc = mad64(m[i], QL[k - i], acc); bcc = mad64(n[i], QL[k - i], bcc); }
    rab.l[k - NL] = (uint32_t)acc & MASK29; rcd.l[k - NL] = (uint32_t)bcc & MASK29;
    acc >>= RB; bcc >>= RB;
  }
  rab.l[NL - 1] = (uint32_t)acc; rcd.l[NL - 1] = (uint32_t)bcc;
}

// lazy add: no carry propagation (operands tight -> result lazy)
D377_HD fe fe_add(const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = a.l[i] + b.l[i];
  return r;
}

// one parallel carry pass: limbs 0..7 back below 2^29 + 8, value unchanged
D377_HD fe fe_carry(const fe& a) {
  fe r;
  r.l[0] = a.l[0] & MASK29;
#pragma unroll
  for (int i = 1; i < NL - 1; ++i) r.l[i] = (a.l[i] & MASK29) + (a.l[i - 1] >> RB);
  r.l[NL - 1] = a.l[NL - 1] + (a.l[NL - 2] >> RB);
  return r;
}

// a - b + 8q, tight result.  b may be lazy (limbs < 2^30), b < 8q.
D377_HD fe fe_sub(const fe& a, const fe& b) {
  fe t;
#pragma unroll
  for (int i = 0; i < NL; ++i) t.l[i] = a.l[i] + SUB8Q[i] - b.l[i];
  return fe_carry(t);
}

D377_HD fe fe_neg(const fe& a) {
  fe t;
#pragma unroll
  for (int i = 0; i < NL; ++i) t.l[i] = SUB8Q[i] - a.l[i];
  return fe_carry(t);
}

D377_HD fe fe_dbl(const fe& a) { return fe_add(a, a); }

D377_HD fe fe_select(bool c, const fe& a, const fe& b) {   // c ? a : b
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}

D377_HD fe fe_const(const uint32_t (&c)[NL]) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = c[i];
  return r;
}

D377_HD fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = 0;
  return r;
}

// Plain (non-Montgomery) canonical value in [0, q) as tight limbs: x*R -> x.
// Montgomery reduction alone (multiplication by the integer 1); its result is in [0, q].
D377_HD fe fe_canon(const fe& a) {
  fe one = fe_zero();
  one.l[0] = 1;
  fe r = fe_mul(a, one);
  uint32_t diff = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) diff |= r.l[i] ^ QL[i];
  if (diff == 0) r = fe_zero();     // the value q itself represents 0
  return r;
}

D377_HD bool fe_canon_is_zero(const fe& c) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) o |= c.l[i];
  return o == 0;
}
D377_HD bool fe_is_zero(const fe& a) { return fe_canon_is_zero(fe_canon(a)); }
// src/sign.rs:19-23: "negative" = low bit of the canonical value
D377_HD bool fe_is_negative(const fe& a) { return (fe_canon(a).l[0] & 1u) != 0; }
D377_HD bool fe_eq(const fe& a, const fe& b) { return fe_is_zero(fe_sub(a, b)); }

// 8 little-endian 32-bit words (a 256-bit integer) <-> 9 x 29-bit limbs
D377_HD fe fe_from_words(const uint32_t w[8]) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int bit = RB * i, lo = bit >> 5, sh = bit & 31;
    uint32_t v = w[lo] >> sh;
    if (sh + RB > 32 && lo + 1 < 8) v |= w[lo + 1] << (32 - sh);
    r.l[i] = v & MASK29;
  }
  return r;
}
D377_HD void fe_to_words(const fe& c, uint32_t w[8]) {   // c must be canonical (< 2^256)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int bit = 32 * j, lo = bit / RB, sh = bit % RB;     // word j starts inside limb lo
    uint32_t v = c.l[lo] >> sh;
    if (lo + 1 < NL) v |= c.l[lo + 1] << (RB - sh);
    if (2 * RB - sh < 32 && lo + 2 < NL) v |= c.l[lo + 2] << (2 * RB - sh);
    w[j] = v;
  }
}

// canonical integer comparison against q (words, little-endian): true if w >= q
D377_HD bool words_geq(const uint32_t w[8], const uint32_t (&mod)[8]) {
  bool gt = false, lt = false;
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    gt = gt || (!lt && w[i] > mod[i]);
    lt = lt || (!gt && w[i] < mod[i]);
  }
  return !lt;
}

}  // namespace d377
