// quad_ops.hpp -- group operations on four lanes (device only): the MSM's Horner chains (msm.hip) and the small-batch
// scalar multiplication (d377.hip) run one group element per quad of lanes instead of one per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve.hpp"

namespace d377 {

// The Horner chains (over the bit-sums of a window, over the windows) are dependency chains whatever the batch size:
// 252 doublings in the tail.  A doubling is two rounds of four independent field products (X^2, Y^2, Z^2, TZ, then
// EF, GH, FG, EH) and so is an addition, so four lanes each take one product per round -- the same instruction stream
// on different operands, no divergence.  The point lives DISTRIBUTED over the quad: lane r holds coordinate r (X, Y, Z,
// T), which is exactly what lane r's second product produces, and a round's operands are fetched with DPP quad_perm
// moves (one VALU instruction per limb, no LDS round trip).  (Round 2 kept the whole point in every lane and picked
// operands with selects: ~660 instructions per doubling, 392 of them the two products; this form is ~510, and an
// addition ~600.)
// The other operand of an addition comes from memory in CACHED form -- (Y-X, Y+X, 2dT, Z), made once per point, in
// parallel, before the chain starts -- so lane r just loads the slot it multiplies by; subtracting a point swaps two
// slots and two sums, which is how the chains absorb the sign of the sign-folded doubling (-[2]P, curve.hpp
// ge_double_neg) instead of negating after every step.
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ fe fe_quad_perm(const fe& v) {       // lane r of every quad <- lane P_r
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i)
    r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true);
  // Keep the moves as moves: hipcc's DPP combiner folds them into the additions and subtractions that consume them
  // (v_add_u32_dpp / v_subrev_u32_dpp whose destination is also their second source), and some of those folded
  // subtractions came back computed on the lane's OWN value instead of the permuted one (measured: limbs 0 and 1 of
  // A - B in gq_double_neg, lanes 0, 2, 3, once the doubling sat in a loop; tests/cpp/gq_selftest.hip).
#pragma unroll
  for (int i = 0; i < NL; ++i) asm("" : "+v"(r.l[i]));
  return r;
}
__device__ __forceinline__ fe fe_pick(int role, const fe& a, const fe& b, const fe& c, const fe& d) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const uint32_t lo = (role & 1) ? b.l[i] : a.l[i], hi = (role & 1) ? d.l[i] : c.l[i];
    r.l[i] = (role & 2) ? hi : lo;
  }
  return r;
}
// -[2]P (the formulas of ge_double_neg; a lane forms only the value it contributes to the second round: curve.hpp
// gq_double_own).  E = 2XY is taken as 2TZ: every lane's first product has its own coordinate as one operand.
__device__ __forceinline__ fe gq_double_neg(const fe& v, int role) {
  const fe m1 = fe_mul(v, fe_quad_perm<0, 1, 2, 2>(v));         // A = X^2, B = Y^2, Z^2, T Z
  const fe w = gq_double_own(role, fe_quad_perm<0, 0, 0, 3>(m1), fe_quad_perm<1, 1, 1, 3>(m1), fe_quad_perm<2, 2, 2, 2>(m1));   // G', H', F', E
  return fe_mul(fe_quad_perm<3, 0, 2, 3>(w), fe_quad_perm<2, 1, 0, 1>(w));   // E F', G' H', F' G', E H'
}
// A point of a chain's other operands in cached form, four 9-word slots in LDS: slot 0 Y-X, 1 Y+X (both carried), 2 2dT, 3 Z
constexpr int GQ_WORDS = 4 * NL;
__device__ __forceinline__ void gq_store_cached(uint32_t* rec, const ge& p) {
  const fe ymx = fe_sub(p.y, p.x), ypx = fe_carry(fe_add(p.y, p.x)), kt = fe_mul(fe_const(FE_K), p.t);
#pragma unroll
  for (int i = 0; i < NL; ++i) { rec[i] = ymx.l[i]; rec[NL + i] = ypx.l[i]; rec[2 * NL + i] = kt.l[i]; rec[3 * NL + i] = p.z.l[i]; }
}
// P + Q, or P - Q with neg_q (wave-uniform): src/min_curve/element.rs:291-322 with Q cached, as ge_add_cached
// the slot of Q's cached form lane `role` multiplies by: -Q has Y-X and Y+X change places
__device__ __forceinline__ int gq_add_slot(int role, bool neg_q) { return (role < 2 && neg_q) ? (role ^ 1) : role; }
// ... with the lane's slot of Q already in hand (opb), wherever it came from
__device__ __forceinline__ fe gq_add_with(const fe& v, const fe& opb, int role, bool neg_q) {
  // lane 0: (Yp - Xp)(Yq - Xq), lane 1: (Yp + Xp)(Yq + Xq), lane 2: Tp * 2dTq, lane 3: 2Zp * Zq
  const fe opa = gq_add_in_own(role, fe_quad_perm<1, 1, 3, 2>(v), fe_quad_perm<0, 0, 3, 2>(v));
  const fe m1 = fe_mul(opa, opb);                               // a, b, c, d
  // E = b - a, H = b + a, F = d - c, G = d + c; the sign of 2dT (-Q) makes F and G change places
  const bool sub = (role == 0) | ((role >= 2) & ((role == 2) != neg_q));   // (bitwise: no branches on a lane's role)
  const fe w = gq_add_own(sub, fe_quad_perm<1, 1, 3, 3>(m1), fe_quad_perm<0, 0, 2, 2>(m1));   // E, H, F, G
  return fe_mul(fe_quad_perm<0, 3, 2, 0>(w), fe_quad_perm<2, 1, 3, 1>(w));   // E F, G H, F G, E H
}
__device__ __forceinline__ fe gq_add(const fe& v, const uint32_t* qrec, int role, bool neg_q) {
  const int slot = gq_add_slot(role, neg_q);
  fe opb;
#pragma unroll
  for (int i = 0; i < NL; ++i) opb.l[i] = qrec[slot * NL + i];
  return gq_add_with(v, opb, role, neg_q);
}
// a whole point (every lane the same copy) -> its distributed form, and back
__device__ __forceinline__ fe gq_from_ge(const ge& p, int role) { return fe_pick(role, p.x, p.y, p.z, p.t); }
__device__ __forceinline__ ge gq_to_ge(const fe& v) {
  ge r;
  r.x = fe_quad_perm<0, 0, 0, 0>(v); r.y = fe_quad_perm<1, 1, 1, 1>(v); r.z = fe_quad_perm<2, 2, 2, 2>(v); r.t = fe_quad_perm<3, 3, 3, 3>(v);
  return r;
}

// The cached slot this lane multiplies by, straight from a distributed point (lane 0: Y-X, 1: Y+X, 2: 2dT, 3: Z): what
// gq_store_cached computes from a whole point, without gathering the point first.  Every lane runs the three forms and
// keeps its own.
__device__ __forceinline__ fe gq_cached_slot(const fe& v, int role) {
  const fe x = fe_quad_perm<0, 0, 3, 2>(v);                     // X, X, T, Z
  const fe y = fe_quad_perm<1, 1, 1, 1>(v);
  return fe_pick(role, fe_sub(y, x), fe_carry(fe_add(y, x)), fe_mul(fe_const(FE_K), x), x);
}

// [k]P on a quad: signed 4-bit windows (fr_recode_signed16), most significant first -- 63 x (4 sign-folded doublings,
// 1 addition) over a table of the cached slots of 0 .. 8 times P, GQ_TAB_ENTRIES x GQ_WORDS words of LDS that belong to
// this quad.  `v1` is P in distributed form; every thread of the workgroup calls this together (two barriers).
constexpr int GQ_TAB_ENTRIES = 9;
__device__ __forceinline__ fe gq_scalar_mul_w4(const fe& v1, const uint32_t dg[8], uint32_t* qtab, int role) {
  // table: entry j = the cached slots of [j]P, each lane the slot it will multiply by
  const fe id_slot = fe_pick(role, fe_const(FE_ONE), fe_const(FE_ONE), fe_zero(), fe_const(FE_ONE));
  const fe s1 = gq_cached_slot(v1, role);
#pragma unroll
  for (int i = 0; i < NL; ++i) { qtab[role * NL + i] = id_slot.l[i]; qtab[GQ_WORDS + role * NL + i] = s1.l[i]; }
  __syncthreads();
  fe acc = v1;
#pragma unroll 1
  for (int j = 2; j < GQ_TAB_ENTRIES; ++j) {
    acc = gq_add(acc, qtab + GQ_WORDS, role, false);           // [j]P = [j-1]P + P (the unified addition also doubles)
    const fe sj = gq_cached_slot(acc, role);
#pragma unroll
    for (int i = 0; i < NL; ++i) qtab[j * GQ_WORDS + role * NL + i] = sj.l[i];
  }
  __syncthreads();
  int d = fr_digit(dg, 63);                                    // 0 or 1
  fe v = fe_select(d != 0, v1, gq_from_ge(ge_identity(), role));
#pragma unroll 1
  for (int i = 62; i >= 0; --i) {
#pragma unroll 1
    for (int j = 0; j < 4; ++j) v = gq_double_neg(v, role);    // four sign-folded doublings keep the sign
    d = fr_digit(dg, i);
    const bool neg = d < 0;
    v = gq_add(v, qtab + (neg ? -d : d) * GQ_WORDS, role, neg);
  }
  return v;
}

}  // namespace d377
