// inv30.hpp -- 1/x mod q by divsteps ("safegcd", Bernstein-Yang 2019), for the batched inversions.
//
// The reference inverts with ark-ff's `Field::inverse` (src/fields/fq/u64/wrapper.rs:104-112; its u32 backend uses
// the same divsteps family, src/fields/fq/u32/wrapper.rs:129-207); an inverse is a field value, so any correct
// algorithm gives the reference's result.  x^(q-2) costs 296 S + 53 M = ~63 000 VALU instructions per lane
// (curve.hpp, fe_invert_chain); this is ~29 000 (26 000 measured in issue slots): 20 rounds of 30 constant-time "half-delta" divsteps on the
// low words (a 2x2 transition matrix with entries below 2^30), each followed by one matrix application to
// (f, g) and one to (d, e) modulo q, on nine signed 30-bit limbs.  600 divsteps cover every modulus below 2^256
// (the bound for this variant is 590).  Same instruction sequence in every lane: no divergence.
// Limb form: value = sum v[i] 2^(30 i), v[0..7] in [0, 2^30), v[8] signed.
#pragma once
#include "fq29.hpp"

namespace d377 {

struct s30 { int32_t v[9]; };
struct trans30 { int32_t u, v, q, r; };
constexpr int32_t M30 = (int32_t)(0xFFFFFFFFu >> 2);

// c + a * b on signed 32-bit factors: one v_mad_i64_i32.  (Left to itself hipcc recognises the sign-extended form in a
// quarter of the places below and expands the rest into 64 x 64-bit products: 3 unsigned MACs, 2 v_mul_lo and moves each.)
D377_HD int64_t mac_i64_i32(int32_t a, int32_t b, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  int64_t r;
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c) : "vcc");
  return r;
#else
  return c + (int64_t)a * b;
#endif
}

// 30 divsteps on the low words.  zeta = -(delta + 1/2).  Returns the new zeta; 2^30 [f', g'] = t [f, g].
D377_HD int32_t divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, trans30* t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
#pragma unroll
  for (int i = 0; i < 30; ++i) {
    uint32_t c1 = (uint32_t)(zeta >> 31);            // delta > 0
    const uint32_t c2 = 0u - (g & 1u);               // g odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;   // (f, u, v) or their negatives
    g += x & c2; q += y & c2; r += z & c2;
    c1 &= c2;                                        // swap: delta > 0 and g odd
    zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;       // -zeta - 2 or zeta - 1
    f += g & c1; u += q & c1; v += r & c1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  t->u = (int32_t)u; t->v = (int32_t)v; t->q = (int32_t)q; t->r = (int32_t)r;
  return zeta;
}

// (f, g) <- t (f, g) / 2^30 (exact)
D377_HD void update_fg_30(s30* f, s30* g, const trans30& t) {
  int64_t cf = mac_i64_i32(t.v, g->v[0], mac_i64_i32(t.u, f->v[0], 0));
  int64_t cg = mac_i64_i32(t.r, g->v[0], mac_i64_i32(t.q, f->v[0], 0));
  cf >>= 30; cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    const int32_t fi = f->v[i], gi = g->v[i];
    cf = mac_i64_i32(t.v, gi, mac_i64_i32(t.u, fi, cf));
    cg = mac_i64_i32(t.r, gi, mac_i64_i32(t.q, fi, cg));
    f->v[i - 1] = (int32_t)cf & M30; cf >>= 30;
    g->v[i - 1] = (int32_t)cg & M30; cg >>= 30;
  }
  f->v[8] = (int32_t)cf; g->v[8] = (int32_t)cg;
}

// (d, e) <- t (d, e) / 2^30 mod q, both kept in (-2q, q)
D377_HD void update_de_30(s30* d, s30* e, const trans30& t) {
  const int32_t sd = d->v[8] >> 31, se = e->v[8] >> 31;
  int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
  int64_t cd = mac_i64_i32(t.v, e->v[0], mac_i64_i32(t.u, d->v[0], 0));
  int64_t ce = mac_i64_i32(t.r, e->v[0], mac_i64_i32(t.q, d->v[0], 0));
  md -= (int32_t)(((uint32_t)cd + (uint32_t)md) & (uint32_t)M30);     // q^-1 mod 2^30 = 1
  me -= (int32_t)(((uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
  cd = mac_i64_i32(FQ_MODULUS_S30[0], md, cd);
  ce = mac_i64_i32(FQ_MODULUS_S30[0], me, ce);
  cd >>= 30; ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    const int32_t di = d->v[i], ei = e->v[i];
    cd = mac_i64_i32(FQ_MODULUS_S30[i], md, mac_i64_i32(t.v, ei, mac_i64_i32(t.u, di, cd)));
    ce = mac_i64_i32(FQ_MODULUS_S30[i], me, mac_i64_i32(t.r, ei, mac_i64_i32(t.q, di, ce)));
    d->v[i - 1] = (int32_t)cd & M30; cd >>= 30;
    e->v[i - 1] = (int32_t)ce & M30; ce >>= 30;
  }
  d->v[8] = (int32_t)cd; e->v[8] = (int32_t)ce;
}

// r in (-2q, q) -> [0, q), negated first if sign < 0
D377_HD void normalize_30(s30* r, int32_t sign) {
  int32_t v[9];
  int32_t cond_add = r->v[8] >> 31;
  const int32_t cond_negate = sign >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) v[i] = ((r->v[i] + (FQ_MODULUS_S30[i] & cond_add)) ^ cond_negate) - cond_negate;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i + 1] += v[i] >> 30; v[i] &= M30; }
  cond_add = v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) v[i] += FQ_MODULUS_S30[i] & cond_add;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i + 1] += v[i] >> 30; v[i] &= M30; }
#pragma unroll
  for (int i = 0; i < 9; ++i) r->v[i] = v[i];
}

// integer in [0, q), nine tight 29-bit limbs <-> nine 30-bit limbs
D377_HD s30 s30_from_limbs29(const uint32_t l[NL]) {
  s30 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int bit = 30 * i, lo = bit / RB, sh = bit % RB;       // 30-bit limb i starts inside 29-bit limb lo
    uint32_t v = lo < NL ? l[lo] >> sh : 0u;
    if (lo + 1 < NL) v |= l[lo + 1] << (RB - sh);
    if (2 * RB - sh < 30 && lo + 2 < NL) v |= l[lo + 2] << (2 * RB - sh);
    r.v[i] = (int32_t)(v & (uint32_t)M30);
  }
  return r;
}
D377_HD void s30_to_limbs29(const s30& a, uint32_t l[NL]) {
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int bit = RB * i, lo = bit / 30, sh = bit % 30;
    uint32_t v = (uint32_t)a.v[lo] >> sh;
    if (lo + 1 < 9) v |= (uint32_t)a.v[lo + 1] << (30 - sh);
    l[i] = v & MASK29;
  }
}

// y = x^-1 mod q for the integer x in [0, q) (0 -> 0), limbs29 in and out
D377_HD void modinv_limbs29(const uint32_t x[NL], uint32_t y[NL]) {
  s30 d, e, f, g = s30_from_limbs29(x);
#pragma unroll
  for (int i = 0; i < 9; ++i) { d.v[i] = 0; e.v[i] = 0; f.v[i] = FQ_MODULUS_S30[i]; }
  e.v[0] = 1;
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < 20; ++it) {
    trans30 t;
    zeta = divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], &t);
    update_de_30(&d, &e, t);
    update_fg_30(&f, &g, t);
  }
  normalize_30(&d, f.v[8]);          // g = 0, f = +-1 (or +-q for x = 0, where d = 0)
  s30_to_limbs29(d, y);
}

}  // namespace d377
