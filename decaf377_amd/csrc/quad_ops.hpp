// quad_ops.hpp -- group operations on four lanes (device only): the MSM's Horner chains (msm.hip) and the small-batch
// scalar multiplication (d377.hip) run one group element per quad of lanes instead of one per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve.hpp"

namespace d377 {

// The Horner chains (over the bit-sums of a window, over the windows) are dependency chains whatever the batch size:
// 252 doublings in the tail.  A doubling is two rounds of four independent field products (X^2, Y^2, 2Z^2, 2XY, then
// EF, GH, FG, EH) and so is an addition, so four lanes each take one product per round -- the same instruction stream
// on different operands, no divergence.  The point lives DISTRIBUTED over the quad: lane r holds coordinate r (X, Y, Z,
// T), which is exactly what lane r's second product produces, and a round's operands are fetched with DPP quad_perm
// moves (one VALU instruction per limb, no LDS round trip).  (Round 2 kept the whole point in every lane and picked
// operands with selects: ~660 instructions per doubling, 392 of them the two products; this form is ~540.)
// The other operand of an addition comes from memory in CACHED form -- (Y-X, Y+X, 2dT, Z), made once per point, in
// parallel, before the chain starts -- so lane r just loads the slot it multiplies by; subtracting a point swaps two
// slots and two sums, which is how the chains absorb the sign of the sign-folded doubling (-[2]P, curve.hpp
// ge_double_neg) instead of negating after every step.
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ fe fe_quad_perm(const fe& v) {       // lane r of every quad <- lane P_r
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i)
    r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, false);
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
// -[2]P (same formulas and operand forms as ge_double_neg, whose bounds the host build checks)
__device__ __forceinline__ fe gq_double_neg(const fe& v, int role) {
  const fe opa = fe_quad_perm<0, 1, 2, 0>(v);                   // X, Y, Z, X
  fe opb = fe_quad_perm<0, 1, 2, 1>(v);                         // X, Y, 2Z, 2Y
  const uint32_t sh = (uint32_t)role >> 1;
#pragma unroll
  for (int i = 0; i < NL; ++i) opb.l[i] <<= sh;
  const fe m1 = fe_mul(opa, opb);                               // A = X^2, B = Y^2, C = 2Z^2, E = 2XY
  const fe a = fe_quad_perm<0, 0, 0, 0>(m1), b = fe_quad_perm<1, 1, 1, 1>(m1);
  const fe h = fe_add(a, b), g = fe_sub(a, b);                  // H' = A + B (lazy), G' = A - B (carried)
  const fe f = fe_add(g, fe_quad_perm<2, 2, 2, 2>(m1));         // F' = G' + C (lazy)
  const fe e = fe_quad_perm<3, 3, 3, 3>(m1);
  return fe_mul(fe_pick(role, e, g, f, e), fe_pick(role, f, h, g, h));   // E F', G' H', F' G', E H'
}
// A point of a chain's other operands in cached form, four 9-word slots in LDS: slot 0 Y-X, 1 Y+X (both carried), 2 2dT, 3 Z
constexpr int GQ_WORDS = 4 * NL;
__device__ __forceinline__ void gq_store_cached(uint32_t* rec, const ge& p) {
  const fe ymx = fe_sub(p.y, p.x), ypx = fe_carry(fe_add(p.y, p.x)), kt = fe_mul(fe_const(FE_K), p.t);
#pragma unroll
  for (int i = 0; i < NL; ++i) { rec[i] = ymx.l[i]; rec[NL + i] = ypx.l[i]; rec[2 * NL + i] = kt.l[i]; rec[3 * NL + i] = p.z.l[i]; }
}
// P + Q, or P - Q with neg_q (wave-uniform): src/min_curve/element.rs:291-322 with Q cached, as ge_add_cached
__device__ __forceinline__ fe gq_add(const fe& v, const uint32_t* qrec, int role, bool neg_q) {
  // lane 0: (Yp - Xp)(Yq - Xq), lane 1: (Yp + Xp)(Yq + Xq), lane 2: Tp * 2dTq, lane 3: 2Zp * Zq
  const fe x = fe_quad_perm<0, 0, 3, 2>(v);                     // X, X, T, Z
  const fe y = fe_quad_perm<1, 1, 1, 1>(v);
  const fe opa = fe_pick(role, fe_sub(y, x), fe_add(y, x), x, fe_add(x, x));
  const int slot = (role < 2 && neg_q) ? (role ^ 1) : role;     // -Q: Y-X and Y+X change places
  fe opb;
#pragma unroll
  for (int i = 0; i < NL; ++i) opb.l[i] = qrec[slot * NL + i];
  const fe m1 = fe_mul(opa, opb);                               // a, b, c, d
  const fe a = fe_quad_perm<0, 0, 0, 0>(m1), b = fe_quad_perm<1, 1, 1, 1>(m1);
  const fe c = fe_quad_perm<2, 2, 2, 2>(m1), d = fe_quad_perm<3, 3, 3, 3>(m1);
  const fe e = fe_sub(b, a), h = fe_add(b, a);
  const fe dmc = fe_sub(d, c), dpc = fe_carry(fe_add(d, c));
  const fe f = fe_select(neg_q, dpc, dmc), g = fe_select(neg_q, dmc, dpc);      // the sign of 2dT: F and G change places
  return fe_mul(fe_pick(role, e, g, f, e), fe_pick(role, f, h, g, h));   // E F, G H, F G, E H
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

}  // namespace d377
