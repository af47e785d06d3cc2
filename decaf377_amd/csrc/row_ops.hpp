// row_ops.hpp -- Fq arithmetic spread over the lanes of a DPP row (device only), for the chains nothing else runs beside.
//
// The tail of every multi-scalar multiplication is ONE dependency chain: ~250 doublings of a single point (msm.hip,
// k_msm_final), whatever the batch size.  One lane per element (fq29.hpp) makes that chain 196 + 168 instructions per
// product pair and leaves 63 lanes of the wave idle; four lanes per point (quad_ops.hpp) run the four products of a
// round side by side, 521 instructions per doubling.  Here a field element lies ACROSS lanes -- limb j in lane j of a
// 16-lane row -- so that the 100 limb products of one field product are 10 per lane, and the four rows of a wave
// carry the four products of a round: ~100 instructions per product on the critical path instead of 196.
//
// Representation: 10 limbs of 28 bits (lanes 0..9 of the row; lanes 10..15 hold 0), PLAIN residues mod q -- no
// Montgomery factor: the reduction is a fold of the product's high columns with the precomputed residues
// FOLD[m] = 2^(28 (10 + m)) mod q, which has no serial digit chain (a Montgomery reduction produces its digits one
// after the other: nine dependent steps, each a cross-lane broadcast).  Values are lazily reduced:
//   tight   output of row_mul: limbs 0..8 < 2^28 + 2^12, limb 9 < 2^20 + 2^12     (value < 2^272.01)
//   lazy    sums and differences of a few tight values: limbs 0..8 < 2^30.25, limb 9 < 2^24
//   row_mul takes tight or lazy operands (every accumulator stays below 2^64: tools/row_model.py walks the worst case
//   and runs the same steps on random operands against a * b mod q).
// Semantics: the reference's Fq product / sum / difference (src/fields/fq/u64/wrapper.rs:99-132) on residues.
//
// Why 10 x 28 and not the 9 x 29 of fq29.hpp: 9 x 29 = 261 bits leave 8 bits above q, and the unreduced low half of a
// product overhangs its top limb by ~35 bits, so every product would need three or four folds to come back under
// 2^261; 280 bits leave 27: the fold of the high columns, one of the 36-bit overhang and one of the top limb's upper ten
// bits (2^272 mod q) bring a product back under 2^272, which leaves the lazy sums 4 bits in every limb and 11 in the top one.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve.hpp"

namespace d377 {
namespace row {

#include "row_constants.inc"

constexpr int RL = 10, RW = 28;
constexpr uint32_t M28 = (1u << RW) - 1u;

// DPP moves inside a 16-lane row.  Lanes that would read outside the row get 0 (bound_ctrl).
template <int N> __device__ __forceinline__ uint32_t bcast(uint32_t v) {       // every lane <- lane N of its row
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + N, 0xF, 0xF, true);
}
template <int N> __device__ __forceinline__ uint32_t shr(uint32_t v) {         // lane j <- lane j - N (0 for j < N)
  if (N == 0) return v;
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xF, 0xF, true);
}
template <int N> __device__ __forceinline__ uint32_t shl(uint32_t v) {         // lane j <- lane j + N (0 beyond the row)
  if (N == 0) return v;
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xF, 0xF, true);
}

// per-lane constants, loaded once per kernel
struct RowK {
  uint32_t fold[11];     // limb j of FOLD[m]
  uint32_t f272;         // limb j of 2^272 mod q
  uint32_t keep2;        // all ones for lanes 0..8, 2^20 - 1 for lane 9, 0 beyond
  uint32_t keep;         // final carry passes: M28 for lanes 0..8, all ones for lane 9 (the top limb is not split), 0 beyond
  uint32_t cmask;        // all ones for lanes 0..8 (they hand a carry up), 0 for lane 9 and beyond
  uint32_t valid;        // all ones for lanes 0..9
  uint32_t sub_tight, sub_lazy, one, k2d;
};
__device__ __forceinline__ RowK row_consts() {
  const int j = threadIdx.x & 15;
  RowK K;
#pragma unroll
  for (int m = 0; m < 11; ++m) K.fold[m] = ROW_FOLD[j][m];
  K.f272 = ROW_F272[j];
  K.keep2 = j < 9 ? 0xFFFFFFFFu : (j == 9 ? 0xFFFFFu : 0u);
  K.keep = j < 9 ? M28 : (j == 9 ? 0xFFFFFFFFu : 0u);
  K.cmask = j < 9 ? 0xFFFFFFFFu : 0u;
  K.valid = j < 10 ? 0xFFFFFFFFu : 0u;
  K.sub_tight = ROW_SUB_TIGHT[j]; K.sub_lazy = ROW_SUB_LAZY[j]; K.one = ROW_ONE[j]; K.k2d = ROW_2D[j];
  return K;
}

// x = lo + 2^28 mid + 2^56 top  ->  lo_j + mid_(j-1) + top_(j-2): limbs < 2^29 + 2^8, the value unchanged
__device__ __forceinline__ uint32_t split3(uint64_t x) {
  const uint32_t lo = (uint32_t)x & M28, mid = (uint32_t)(x >> RW) & M28, top = (uint32_t)(x >> (2 * RW));
  return lo + shr<1>(mid) + shr<2>(top);
}

// The cross-lane operands of the whole product are fetched first, each into a register of its own, and multiplied
// afterwards: a DPP move whose destination was written by one of the two instructions before it costs wait states
// (the destination is also its tied "old" operand), which is what register reuse inside a fetch-multiply-fetch sequence gives.
template <int I> struct Fetch {
  static __device__ __forceinline__ void run(uint32_t a, uint32_t b, uint32_t (&ai)[RL], uint32_t (&bl)[RL]) {
    ai[I] = bcast<I>(a);
    bl[I] = shr<I>(b);                                   // lane j: b_(j-i), so that a_i b_(j-i) belongs to column j
    Fetch<I + 1>::run(a, b, ai, bl);
  }
};
template <> struct Fetch<RL> { static __device__ __forceinline__ void run(uint32_t, uint32_t, uint32_t (&)[RL], uint32_t (&)[RL]) {} };
template <int M> struct FoldFetch {
  static __device__ __forceinline__ void run(uint32_t h, uint32_t (&hm)[11]) { hm[M] = bcast<M>(h); FoldFetch<M + 1>::run(h, hm); }
};
template <> struct FoldFetch<11> { static __device__ __forceinline__ void run(uint32_t, uint32_t (&)[11]) {} };

// acc + a * k for a multiplicand k that does not change inside a loop (the fold constants).  Written as the instruction itself:
// from `acc + (uint64_t)a * k` the compiler hoists the widening of k out of the loop, and when the widened value then
// reaches the loop through a join of two paths it no longer knows that the upper half is zero and multiplies 64 x 32 bits
// -- two multiply-adds and two moves where one multiply-add does (seen in k_msm_tiny: 82 instead of 54 per doubling).
__device__ __forceinline__ uint64_t mad_k(uint32_t a, uint32_t k, uint64_t acc) {
  uint64_t r, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(carry) : "v"(a), "v"(k), "v"(acc));
  return r;
}

// a * b mod q (lazily reduced: tight).  a, b: tight or lazy, lanes 10..15 zero.
__device__ __forceinline__ uint32_t row_mul(uint32_t a, uint32_t b, const RowK& K) {
  // The 19 columns of the product.  The row has 16 lanes, so lane j of L takes column j for ALL j = 0..15 (b's lanes
  // 10..15 are zero: shr<i>(b) feeds lanes 10..15 exactly the terms of columns 10..15), and only columns 16, 17, 18 need
  // a second accumulator: H, lanes 6..8, from the steps i = 7, 8, 9.
  const int j = threadIdx.x & 15;
  uint32_t ai[RL], bl[RL];
  Fetch<0>::run(a, b, ai, bl);
  const uint32_t bh7 = shl<3>(b), bh8 = shl<2>(b), bh9 = shl<1>(b);    // lane j: b_(j+10-i)
  uint64_t L = 0;
#pragma unroll
  for (int i = 0; i < RL; ++i) L += (uint64_t)ai[i] * bl[i];
  const uint64_t H = (uint64_t)ai[7] * bh7 + (uint64_t)ai[8] * bh8 + (uint64_t)ai[9] * bh9;   // lanes 6..8 count (lanes 0..5: strays)
  // the high half, columns 10..18, as lanes 0..8
  // (the moves stand outside the selects: a DPP move inside a conditional arm runs with the other lanes switched off, and a
  // switched-off source lane reads as zero)
  const uint32_t slo = shl<10>((uint32_t)L), shi = shl<10>((uint32_t)(L >> 32));
  const uint32_t ulo = j < 6 ? slo : (uint32_t)H, uhi = j < 6 ? shi : (uint32_t)(H >> 32);
  L = j < RL ? L : 0;
  const uint32_t h = split3(((uint64_t)uhi << 32) | ulo);   // ... as limbs 0..10 of 2^280 x (...)
  uint32_t hm[11];
  FoldFetch<0>::run(h, hm);
#pragma unroll
  for (int m = 0; m < 11; ++m) L = mad_k(hm[m], K.fold[m], L);   // L + sum_m h_m FOLD[m]: < 2^64 per lane
  const uint32_t n = split3(L);                          // limbs 0..11; 10 and 11 overhang the representation
  uint64_t R = (uint64_t)(n & K.valid);
  R = mad_k(bcast<10>(n), K.fold[0], R);
  R = mad_k(bcast<11>(n), K.fold[1], R);               // < 2^56: the value is below 2^283 now
  const uint32_t f = ((uint32_t)R & K.keep) + shr<1>((uint32_t)(R >> RW) & K.cmask);     // < 2^29.5, limb 9 whole
  const uint32_t t = bcast<9>(f) >> 20;                  // what the top limb holds above 2^272: ten bits
  const uint64_t R3 = mad_k(t, K.f272, (uint64_t)(f & K.keep2));
  return ((uint32_t)R3 & K.keep) + shr<1>((uint32_t)(R3 >> RW) & K.cmask);
}
__device__ __forceinline__ uint32_t row_add(uint32_t a, uint32_t b) { return a + b; }
// a - b for a TIGHT b (limbs 0..8 <= 2^28 + 8, limb 9 < 2^31)
__device__ __forceinline__ uint32_t row_sub(uint32_t a, uint32_t b, const RowK& K) { return a + K.sub_tight - b; }
// one carry pass: lazy -> tight-ish (limbs 0..8 <= 2^28 + 15), the value unchanged
__device__ __forceinline__ uint32_t row_carry(uint32_t f, const RowK& K) { return (f & K.keep) + shr<1>((f >> RW) & K.cmask); }


// ---- group elements on the four rows of a wave ---------------------------------------------------------------------
// Row r of the wave holds coordinate r of the point (X, Y, Z, T), limb j in lane j of the row; the four products of a
// round of the doubling / addition formulas run on the four rows.  Between the rounds the rows trade values with
// v_permlane16_swap / v_permlane32_swap (gfx950):
//   swap16(a, b) -> lo = (a0, b0, a2, b2), hi = (a1, b1, a3, b3)      (the rows of each result, from rows of a and b)
//   swap32(a, b) -> lo = (a0, a1, b0, b1), hi = (a2, a3, b2, b3)
struct RowPair { uint32_t lo, hi; };
__device__ __forceinline__ RowPair swap16(uint32_t a, uint32_t b) {
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  return RowPair{r[0], r[1]};
}
__device__ __forceinline__ RowPair swap32(uint32_t a, uint32_t b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  return RowPair{r[0], r[1]};
}

// Which row a lane belongs to, as masks: selections between per-row values are bitfield inserts (one VALU instruction,
// no condition code, no branch for the compiler to build out of a conditional expression).
struct RowSel {
  int r, j;
  uint32_t m0, m1, m2, m3;     // all ones in row 0 / 1 / 2 / 3
  __device__ __forceinline__ static uint32_t pick(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
};
__device__ __forceinline__ RowSel row_sel() {
  RowSel S;
  S.r = (threadIdx.x >> 4) & 3; S.j = threadIdx.x & 15;
  S.m0 = S.r == 0 ? ~0u : 0u; S.m1 = S.r == 1 ? ~0u : 0u; S.m2 = S.r == 2 ? ~0u : 0u; S.m3 = S.r == 3 ? ~0u : 0u;
  return S;
}

// -[2]P with the formulas of ge_double_neg / gq_double_neg (curve.hpp, quad_ops.hpp; the reference's doubling,
// src/min_curve/element.rs:119-136, with E = 2XY taken as 2TZ and F, G, H sign-folded): v tight -> tight.
__device__ __forceinline__ uint32_t rq_double_neg(uint32_t v, const RowSel& S, const RowK& K) {
  const RowPair e = swap16(v, v);                                  // (X, X, Z, Z), (Y, Y, T, T)
  const uint32_t m1 = row_mul(v, RowSel::pick(S.m3, e.lo, v), K);  // A = X^2, B = Y^2, Z^2, T Z
  const RowPair q = swap16(m1, m1);                                // (A, A, ZZ, ZZ), (B, B, TZ, TZ)
  const RowPair s = swap32(m1, m1);                                // (A, B, A, B), ...
  const RowPair t = swap32(q.hi, q.hi);                            // (B, B, B, B), ...
  const uint32_t pa = RowSel::pick(S.m0, m1, RowSel::pick(S.m1, q.lo, s.lo));   // A in rows 0, 1, 2
  const uint32_t diff = pa + K.sub_tight - t.lo, sum = pa + t.lo;  // A - B, A + B
  // G' = A - B, H' = A + B, F' = G' + 2 Z^2, E = 2 T Z
  const uint32_t w = (RowSel::pick(S.m1, sum, diff) & ~S.m3) + ((m1 + m1) & (S.m2 | S.m3));
  const RowPair a1 = swap16(w, w);                                 // (G', G', F', F'), (H', H', E, E)
  const RowPair a2 = swap32(w, w);                                 // (G', H', G', H'), (F', E, F', E)
  const RowPair a3 = swap32(a1.hi, a1.hi);                         // (H', H', H', H'), (E, E, E, E)
  const uint32_t opa = RowSel::pick(S.m0, a3.hi, RowSel::pick(S.m1, a1.lo, w));   // E, G', F', E
  const uint32_t opb = RowSel::pick(S.m0, a2.hi, RowSel::pick(S.m1, w, a2.lo));   // F', H', G', H'
  return row_mul(opa, opb, K);                                     // E F', G' H', F' G', E H'
}
// P + Q or P - Q (neg_q, wave-uniform) with Q cached as four rows of 16 words -- slot 0 Y - X, 1 Y + X, 2 2dT, 3 Z, plain
// residues with limbs below 2^28 -- as gq_add (src/min_curve/element.rs:291-322): v tight -> tight.
__device__ __forceinline__ uint32_t rq_add(uint32_t v, const uint32_t* qrec, const RowSel& S, bool neg_q, const RowK& K) {
  const RowPair e = swap16(v, v);                                  // (X, X, Z, Z), (Y, Y, T, T)
  // Y - X, Y + X, T, 2Z
  const uint32_t opa = RowSel::pick(S.m0, e.hi + K.sub_tight - v, RowSel::pick(S.m1, v + e.lo, RowSel::pick(S.m2, e.hi, e.lo + e.lo)));
  const int slot = (S.r < 2 && neg_q) ? (S.r ^ 1) : S.r;           // -Q: Y - X and Y + X change places
  const uint32_t m1 = row_mul(opa, qrec[slot * 16 + S.j], K);      // a, b, c, d
  const RowPair q = swap16(m1, m1);                                // (a, a, c, c), (b, b, d, d)
  const uint32_t odd = S.m1 | S.m3;
  const uint32_t u = RowSel::pick(odd, m1, q.hi), w_ = RowSel::pick(odd, q.lo, m1);   // (b, b, d, d), (a, a, c, c)
  const uint32_t subm = S.m0 | (neg_q ? S.m3 : S.m2);              // E = b - a, H = b + a, F = d -+ c, G = d +- c
  const uint32_t w = RowSel::pick(subm, u + K.sub_tight - w_, u + w_);
  const RowPair a1 = swap16(w, w);                                 // (E, E, F, F), (H, H, G, G)
  const RowPair a2 = swap32(w, w);                                 // (E, H, E, H), (F, G, F, G)
  const RowPair a3 = swap32(a1.lo, a1.lo);                         // (E, E, E, E), (F, F, F, F)
  const uint32_t opx = RowSel::pick(S.m1, a2.hi, RowSel::pick(S.m3, a3.lo, w));   // E, G, F, E
  const uint32_t opy = RowSel::pick(S.m0, a3.hi, RowSel::pick(S.m1, w, RowSel::pick(S.m2, a1.hi, a2.lo)));   // F, H, G, H
  return row_mul(opx, opy, K);                                     // E F, G H, F G, E H
}

// The cached slot a row multiplies by, from a point in row form: row 0 Y - X, 1 Y + X, 2 2dT, 3 Z (what rq_add reads of
// its other operand).  Lazy limbs (a difference, a sum): multiplicands only.
__device__ __forceinline__ uint32_t rq_cached_slot(uint32_t v, const RowSel& S, const RowK& K) {
  const RowPair e = swap16(v, v);                                  // (X, X, Z, Z), (Y, Y, T, T)
  const uint32_t kt = row_mul(e.hi, K.k2d, K);                     // row 2: 2d T (the other rows' products are not used)
  return RowSel::pick(S.m0, e.hi + K.sub_tight - v, RowSel::pick(S.m1, v + e.lo, RowSel::pick(S.m2, kt, e.lo)));
}
// [k]P for ONE point per wave: signed 4-bit windows (fr_recode_signed16), most significant first -- 63 x (4 sign-folded
// doublings, 1 addition) over a table of the cached slots of 0 .. 8 times P, 9 x 64 words of LDS that belong to this wave
// (gq_scalar_mul_w4 of quad_ops.hpp with a wave where that has a quad).  v1: P in row form; every lane of the wave calls this.
constexpr int RQ_TAB_ENTRIES = 9;
// the table of one point: entry j = the cached slots of [j]P, RQ_TAB_ENTRIES x 64 words (lane t: word t of every entry)
__device__ __forceinline__ void rq_build_table(uint32_t v1, uint32_t* tab, const RowSel& S, const RowK& K) {
  const int t = S.r * 16 + S.j;
  const uint32_t one_or_zero = S.j == 0 ? 1u : 0u;
  tab[t] = S.r == 2 ? 0u : one_or_zero;                            // the identity's cached slots: 1, 1, 0, 1
  tab[64 + t] = rq_cached_slot(v1, S, K);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  uint32_t acc = v1;
#pragma unroll 1
  for (int j = 2; j < RQ_TAB_ENTRIES; ++j) {
    acc = rq_add(acc, tab + 64, S, false, K);                      // [j]P = [j-1]P + P (the unified addition also doubles)
    tab[j * 64 + t] = rq_cached_slot(acc, S, K);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the identity in row form: X = 0, Y = 1, Z = 1, T = 0
__device__ __forceinline__ uint32_t rq_identity(const RowSel& S) { return ((S.r == 1 || S.r == 2) && S.j == 0) ? 1u : 0u; }
__device__ __forceinline__ uint32_t rq_scalar_mul_w4(uint32_t v1, const uint32_t dg[8], uint32_t* tab, const RowSel& S, const RowK& K) {
  rq_build_table(v1, tab, S, K);
  int d = fr_digit(dg, 63);                                        // 0 or 1
  uint32_t v = d != 0 ? v1 : rq_identity(S);
#pragma unroll 1
  for (int i = 62; i >= 0; --i) {
#pragma unroll 1
    for (int k = 0; k < 4; ++k) v = rq_double_neg(v, S, K);        // four sign-folded doublings keep the sign
    d = fr_digit(dg, i);
    const bool neg = d < 0;
    v = rq_add(v, tab + (neg ? -d : d) * 64, S, neg, K);
  }
  return v;
}

// ---- the power chains of the square root (curve.hpp: fe_pow_2_47_m1, fe_pow_m12) in the lane-spread form ---------------
// A square root is ~300 dependent products before its table phase; with one element per wave (or per row) they run here at
// ~0.2 us each instead of ~0.33.  The table phase stays whole-element code (fe_sqrt_tail: canonical values, hash keys).
__device__ __forceinline__ uint32_t row_sqr_n(uint32_t x, int n, const RowK& K) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) x = row_mul(x, x, K);
  return x;
}
// x^(2^47 - 1): 46 S + 9 M, the chain of fe_pow_2_47_m1
__device__ __forceinline__ uint32_t row_pow_2_47_m1(uint32_t x, const RowK& K) {
  const uint32_t e2 = row_mul(row_mul(x, x, K), x, K);
  const uint32_t e4 = row_mul(row_sqr_n(e2, 2, K), e2, K);
  const uint32_t e5 = row_mul(row_mul(e4, e4, K), x, K);
  const uint32_t e10 = row_mul(row_sqr_n(e5, 5, K), e5, K);
  const uint32_t e11 = row_mul(row_mul(e10, e10, K), x, K);
  const uint32_t e22 = row_mul(row_sqr_n(e11, 11, K), e11, K);
  const uint32_t e23 = row_mul(row_mul(e22, e22, K), x, K);
  const uint32_t e46 = row_mul(row_sqr_n(e23, 23, K), e23, K);
  return row_mul(row_mul(e46, e46, K), x, K);
}
// x^((m-1)/2) by the fixed sliding-window schedule D377_POW_CHAIN; the POW_TAB odd powers in `tab` (POW_TAB x 64 words of
// LDS that belong to this wave; lane t uses word t of every entry)
__device__ __forceinline__ uint32_t row_pow_m12(uint32_t x, uint32_t* tab, int t, const RowK& K) {
  const uint32_t x2 = row_mul(x, x, K);
  uint32_t cur = x;
  tab[t] = cur;
#pragma unroll 1
  for (int j = 1; j < POW_TAB; ++j) {
    cur = row_mul(cur, x2, K);
    tab[j * 64 + t] = cur;
  }
  uint32_t acc = tab[((int)(D377_POW_CHAIN[0] & 15u) >> 1) * 64 + t];     // (a lane reads back only words it wrote itself)
#pragma unroll 1
  for (int i = 1; i < D377_POW_LEN; ++i) {
    const uint32_t e = D377_POW_CHAIN[i];
    acc = row_sqr_n(acc, (int)(e >> 4), K);
    acc = row_mul(acc, tab[((int)(e & 15u) >> 1) * 64 + t], K);
  }
  return row_sqr_n(acc, D377_POW_TRAIL, K);
}
// v = (1 / den)^((m-1)/2) and uv = (1 / den)^((m+1)/2) without an inversion, as invsqrt.rs:88-94 builds them for num = 1:
// s = den^(2^47 - 1), t = s^2 den, w = t^((m-1)/2) s, v = w den, uv = w
struct RowPowers { uint32_t v, uv; };
__device__ __forceinline__ RowPowers row_sqrt_powers(uint32_t den, uint32_t* tab, int t, const RowK& K) {
  const uint32_t s = row_pow_2_47_m1(den, K);
  const uint32_t t_ = row_mul(row_mul(s, s, K), den, K);
  const uint32_t w = row_mul(row_pow_m12(t_, tab, t, K), s, K);
  return RowPowers{row_mul(w, den, K), w};
}
// the same for a ratio num / den (invsqrt.rs:91-94): w = (num t)^((m-1)/2) s, v = w den, uv = w num
__device__ __forceinline__ RowPowers row_sqrt_powers_num(uint32_t num, uint32_t den, uint32_t* tab, int t, const RowK& K) {
  const uint32_t s = row_pow_2_47_m1(den, K);
  const uint32_t t_ = row_mul(row_mul(s, s, K), den, K);
  const uint32_t w = row_mul(row_pow_m12(row_mul(num, t_, K), tab, t, K), s, K);
  return RowPowers{row_mul(w, den, K), row_mul(w, num, K)};
}

// ---- one inversion by the whole wave ------------------------------------------------------------------------------
// The encoders at the end of the one-wave chains (k_msm_final, k_msm_small_sum, the one-wave-per-element kernels) invert ONE
// value while 63 lanes have nothing else to do, and that value is the same in every lane -- so nothing diverges whatever the
// code branches on.  fe_invert (inv30.hpp) is written for 64 different values per wave: 20 rounds of ~880 instructions, 570 of
// them the 30 branch-free divsteps on the low words and ~310 the round's 2 x 2 matrix applied to the nine limbs of f, g, d, e.
// Here
//   * the four numbers lie across the four rows of the wave (row 0 f, 1 g, 2 d, 3 e; limb j in lane j, lanes 9..15 zero): the
//     matrix is three multiply-adds per lane and two lazy carry passes by DPP, ~30 instructions.  Limbs stay signed and only
//     nearly normalised between rounds -- lanes 0..7 in (-4, 2^30 + 5), lane 8 the signed top -- which the next round's
//     products (|u| + |v| <= 2^30: below 2^61.6) and the low-word extractions (x & (2^30 - 1)) take as they are; the result
//     is carried once, at the end;
//   * the divsteps are the variable-time form (scalar code: the low words are wave-uniform): runs of zero bits of g in one
//     step, and up to eight bits of g cancelled at a time by the multiple w = -g / f mod 2^k of f that the next k divsteps
//     would add one by one -- the same sequence of divsteps as the plain delta = 1 iteration (Bernstein-Yang 2019; the batched
//     form is the one libsecp256k1's modinv32 "var" uses), ~6 trips per 30 divsteps instead of 30.  At most 724 divsteps
//     bring g to zero for inputs below 2^256: 25 rounds; the rounds after g = 0 pass in ~60 instructions each (f, g, d keep
//     their values).
// 31 us -> 11 us on a lone wave (tools/row_invert_check.py).  An inverse is a field value: same result as fe_invert.
// Every lane of the wave must call this; the value inverted is LANE 0's x, the result comes back in every lane.
__device__ __forceinline__ int32_t divsteps_30_var(int32_t eta, uint32_t f0, uint32_t g0, trans30* t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
  int i = 30;
  for (;;) {
    const int zeros = __builtin_ctz(g | (0xFFFFFFFFu << i));       // g even: halve it (at most the i steps that are left)
    g >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
    if (i == 0) break;
    if (eta < 0) {                                                   // delta > 0 and g odd: swap
      eta = -eta;
      const uint32_t tf = f, tu = u, tv = v;
      f = g; g = 0u - tf; u = q; q = 0u - tu; v = r; r = 0u - tv;
    }
    const int limit = eta + 1 > i ? i : eta + 1;                     // the next `limit` steps do not swap
    const uint32_t mask = (0xFFFFFFFFu >> (32 - limit)) & 255u;
    uint32_t fi = f;                                                 // 1 / f mod 2^12 (f odd: f * f = 1 mod 8)
    fi *= 2u - f * fi;
    fi *= 2u - f * fi;
    const uint32_t w = (0u - g * fi) & mask;                         // g + w f = 0 mod 2^limit
    g += f * w; q += u * w; r += v * w;
  }
  t->u = (int32_t)u; t->v = (int32_t)v; t->q = (int32_t)q; t->r = (int32_t)r;
  return eta;
}
__device__ __forceinline__ int32_t pick9(const int32_t (&v)[9], int j) {
  int32_t r = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) r = j == k ? v[k] : r;
  return r;
}
template <bool DBG>
__device__ __forceinline__ fe fe_invert_wave_impl(const fe& x_own, uint32_t* dbg) {
  const fe c_own = fe_reduce_once(fe_mul_strict(x_own, fe_const(FE_ONE)));     // the residue x R in [0, q), as fe_invert takes it
  uint32_t cl[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) cl[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)c_own.l[i]);
  const s30 g_in = s30_from_limbs29(cl);
  const int lane = threadIdx.x & 63, r = lane >> 4, j = lane & 15;
  const bool odd = (r & 1) != 0, de = r >= 2, top = j == 8;
  const int32_t qj = pick9(FQ_MODULUS_S30, j);
  int32_t v = r == 0 ? qj : (r == 1 ? pick9(g_in.v, j) : ((r == 3 && j == 0) ? 1 : 0));
  int32_t eta = -1;                                                  // -delta
#pragma unroll 1
  for (int it = 0; it < 25; ++it) {
    const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane(v, 0) & (uint32_t)M30, g0 = (uint32_t)__builtin_amdgcn_readlane(v, 16) & (uint32_t)M30;
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane(v, 32), e0 = (uint32_t)__builtin_amdgcn_readlane(v, 48);
    trans30 m;
    eta = divsteps_30_var(eta, f0, g0, &m);
    // the multiples of q that make the new low limbs of d and e vanish (q^-1 mod 2^30 = 1): md, me in (-2^30, 0].  update_de_30
    // first lifts a negative d or e by q, which keeps them in (-2q, q); the sign of a lazily carried number cannot be read off
    // its top limb (top limb -1 over a large limb below is a small POSITIVE number: lifting that one walks d out of range,
    // which a batch of structured residues found), and it is not needed -- without the lift |d|, |e| grow by at most q per
    // round, (-(round + 2) q, q], 2^17 in the top limb after 25 rounds; the one reduction at the end allows for it.
    const int32_t md = -(int32_t)(((uint32_t)m.u * d0 + (uint32_t)m.v * e0) & (uint32_t)M30);
    const int32_t me = -(int32_t)(((uint32_t)m.q * d0 + (uint32_t)m.r * e0) & (uint32_t)M30);
    const RowPair p = swap16((uint32_t)v, (uint32_t)v);                          // (f, f, d, d), (g, g, e, e)
    const int32_t a = odd ? m.q : m.u, b = odd ? m.r : m.v, mm = de ? (odd ? me : md) : 0;
    const int64_t c = mac_i64_i32(a, (int32_t)p.lo, mac_i64_i32(b, (int32_t)p.hi, mac_i64_i32(mm, qj, 0)));
    // c = lo + 2^30 mid + 2^60 tp; the division by 2^30 moves lo one lane down (lane 0's is zero by construction)
    const uint32_t lo = (uint32_t)c & (uint32_t)M30;
    const int64_t h = c >> 30;
    const uint32_t mid = top ? (uint32_t)h : ((uint32_t)h & (uint32_t)M30);     // the top limb keeps its sign
    const int32_t tp = top ? 0 : (int32_t)(h >> 30);
    const uint32_t t1 = shl<1>(lo) + mid;                                       // lanes 0..7: below 2^31
    const uint32_t keep = top ? t1 : (t1 & (uint32_t)M30);
    const uint32_t up = top ? 0u : ((t1 >> 30) + (uint32_t)tp);
    v = (int32_t)(keep + shr<1>(up));
    if (DBG) {                                                              // (tools/row_proto.hip: the rounds against the Python model)
      dbg[it * 80 + lane] = (uint32_t)v;
      if (lane == 0) {
        dbg[it * 80 + 64] = (uint32_t)eta; dbg[it * 80 + 65] = (uint32_t)m.u; dbg[it * 80 + 66] = (uint32_t)m.v; dbg[it * 80 + 67] = (uint32_t)m.q;
        dbg[it * 80 + 68] = (uint32_t)m.r; dbg[it * 80 + 69] = (uint32_t)md; dbg[it * 80 + 70] = (uint32_t)me; dbg[it * 80 + 71] = f0; dbg[it * 80 + 72] = g0;
      }
    }
  }
  const uint32_t f0 = (uint32_t)__builtin_amdgcn_readlane(v, 0);                // f = +-1 (+-q for x = 0, where d = 0): bit 1 tells which
  // +-d + 32 q, carried: a non-negative integer below 60 q < 2^259 that is +-d mod q; the product with R^3 / R reduces it
  const bool negate = (f0 & 2u) != 0;
  s30 d;
  int64_t carry = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int32_t di = __builtin_amdgcn_readlane(v, 32 + i);
    const int64_t t = (int64_t)(negate ? -di : di) + 32ll * FQ_MODULUS_S30[i] + carry;
    d.v[i] = i < 8 ? (int32_t)(t & M30) : (int32_t)t;
    carry = t >> 30;
  }
  fe y;
  s30_to_limbs29(d, y.l);
  return fe_mul(y, fe_const(FE_R3));
}
__device__ __forceinline__ fe fe_invert_wave(const fe& x_own) { return fe_invert_wave_impl<false>(x_own, nullptr); }

// 1 / c for EVERY lane of a full wave out of ONE inversion: the lanes' values are multiplied up a butterfly (six exchange
// steps: after step l a lane holds the product of its block of 2^(l+1) lanes, and keeps the sibling block's product), the
// total -- the same residue in every lane -- is inverted by the wave together (fe_invert_wave: 17 us on a lone wave against
// 31 us for the lane's own divsteps, and the 64 lanes used to run 64 of those side by side), and on the way back
// 1 / (my block) = 1 / (parent block) x (sibling block).  12 products, 54 exchanged words and one wave inversion per lane
// instead of ~26 000 instructions of divsteps: the chunked kernels pay one such inversion per lane per 8 (fixed base: 16)
// elements and per batched-inversion pass, 4-10 % of their time.  Values must be non-zero (the callers store 1 for a
// zero); every lane of the wave must be here.
__device__ __forceinline__ fe fe_shfl_xor(const fe& x, int mask) {
  fe r = x;
#pragma unroll
  for (int k = 0; k < NL; ++k) r.l[k] = (uint32_t)__shfl_xor((int)x.l[k], mask, 64);
  return r;
}
__device__ __forceinline__ fe fe_invert_lanes(const fe& c) {
  fe sib[6];
  fe p = c;
#pragma unroll
  for (int l = 0; l < 6; ++l) {
    sib[l] = fe_shfl_xor(p, 1 << l);
    p = fe_mul(p, sib[l]);
  }
  fe inv = fe_invert_wave(p);                          // (lane 0's product; every lane's is the same residue)
#pragma unroll
  for (int l = 5; l >= 0; --l) inv = fe_mul(inv, sib[l]);
  return inv;
}

// ---- between the two forms (whole field elements in a lane, Montgomery 9 x 29 <-> plain 10 x 28 across a row) --------
// one lane writes an element as a row record (16 words, canonical value, limbs 10..15 zero) / reads a tight record back as
// a product: curve.hpp fe_to_limbs28 / fe_from_limbs28 (whole-element code, so the host build checks its bounds)
__device__ __forceinline__ void row_store_from_fe(uint32_t* rec16, const fe& x) { fe_to_limbs28(x, rec16); }
__device__ __forceinline__ fe row_load_to_fe(const uint32_t* rec16) { return fe_from_limbs28(rec16); }

// A point as row records: four rows of 16 words (X, Y, Z, T, or the cached form Y - X, Y + X, 2dT, Z), written by ONE lane
// from whole coordinates; and back, every lane the whole point.
constexpr int RQ_WORDS = 64;
__device__ __forceinline__ void rq_store_point(uint32_t* rec, const ge& p) {
  row_store_from_fe(rec, p.x); row_store_from_fe(rec + 16, p.y);
  row_store_from_fe(rec + 32, p.z); row_store_from_fe(rec + 48, p.t);
}
__device__ __forceinline__ void rq_store_cached(uint32_t* rec, const ge& p) {
  row_store_from_fe(rec, fe_sub(p.y, p.x)); row_store_from_fe(rec + 16, fe_add(p.y, p.x));
  row_store_from_fe(rec + 32, fe_mul(fe_const(FE_K), p.t)); row_store_from_fe(rec + 48, p.z);
}
__device__ __forceinline__ ge rq_load_point(const uint32_t* rec) {
  ge g;
  g.x = row_load_to_fe(rec); g.y = row_load_to_fe(rec + 16);
  g.z = row_load_to_fe(rec + 32); g.t = row_load_to_fe(rec + 48);
  return g;
}

}  // namespace row
}  // namespace d377
