// Test-only host build of the per-lane device functions (decaf377_amd/csrc/*.hpp), so the
// limb arithmetic and the curve formulas can be checked against the oracle on a machine with
// no GPU.  NOT part of the product: nothing in decaf377_amd/ loads this; it is compiled by
// tests/test_host_sim.py with g++ and exists only under tests/.
#include <cstdint>
#include <cstring>
#include <vector>
#include "curve.hpp"
#include "msm_plan.hpp"
#include "straus.hpp"
using namespace d377;

static std::vector<uint32_t> g_gtab(6 * 256 * GT_STRIDE);
static std::vector<uint8_t> g_slook(1u << S_HASH_BITS);
static std::vector<uint32_t> g_fbase((size_t)FB_WINDOWS * FB_ENTRIES * 27);
static SqrtTables g_T;
static int g_collisions = -1;

static fe fe_pow_u64(fe x, uint64_t e) {   // Montgomery x^e, e > 0
  fe r = x; int top = 63; while (!((e >> top) & 1)) --top;
  for (int i = top - 1; i >= 0; --i) { r = fe_sqr(r); if ((e >> i) & 1) r = fe_mul(r, x); }
  return r;
}
static fe full_norm(fe a) {   // sequential carry, value < 2^261
  uint32_t c = 0; fe r;
  for (int i = 0; i < NL; ++i) { uint32_t t = a.l[i] + c; if (i < NL - 1) { r.l[i] = t & MASK29; c = t >> 29; } else r.l[i] = t; }
  return r;
}
struct HostTab {
  gec e[9];
  void store(int j, const gec& g) { e[j] = g; }
  gec load(int j, bool swap) const { gec c = e[j]; if (swap) { fe t = c.ypx; c.ypx = c.ymx; c.ymx = t; } return c; }
};
struct HostFTab {
  const uint32_t* p;
  gea load(int i, int j, bool swap) const {
    gea g; const uint32_t* q = p + ((size_t)i * FB_ENTRIES + j) * 27;
    for (int k = 0; k < 9; ++k) { g.ypx.l[k] = q[(swap ? 9 : 0) + k]; g.ymx.l[k] = q[(swap ? 0 : 9) + k]; g.kt.l[k] = q[18 + k]; }
    fe_assume_carried(g.ypx, 26.0); fe_assume_carried(g.ymx, 26.0); fe_assume_carried(g.kt, 9.0);
    return g;
  }
};

// the square-root-free compression of the scalar-multiplication and Elligator kernels (curve.hpp, "compression
// without a square root"): one host "lane" walks the batch in rounds of DCB_K elements, as a device lane does
struct HostDcbIO {
  uint32_t st[5][DCB_K][8];
  uint32_t* out;
  size_t base;
  void put(int s, int j, const uint32_t* w) { memcpy(st[s][j], w, 32); }
  void get(int s, int j, uint32_t* w) const { memcpy(w, st[s][j], 32); }
  void park(int j, const uint32_t* w) { memcpy(out + 8 * (base + j), w, 32); }
  void parked(int j, uint32_t* w) const { memcpy(w, out + 8 * (base + j), 32); }
  void emit(int j, const uint32_t* w) { memcpy(out + 8 * (base + j), w, 32); }
};
// the kernels' round structure (d377.hip, dcb_rounds): phase 0 leaves the denominators of the round's square roots in
// records 0 .. NINV-1, they are inverted together, phase 1 does the element's work with the inverses at hand, and the
// square-root-free compressor finishes the round when the operation ends in an encoding it can produce
template <int NINV, class P0, class P1>
static void dcb_rounds(size_t n, uint32_t* out, bool finish, P0 phase0, P1 phase1) {
  for (size_t base = 0; base < n; base += DCB_K) {
    HostDcbIO io; io.out = out; io.base = base;
    const int cnt = (int)((n - base) < (size_t)DCB_K ? (n - base) : (size_t)DCB_K);
    for (int j = 0; j < cnt; ++j) phase0(io, base + j, j);
    for (int sl = 0; sl < NINV; ++sl) dcb_invert_slot(io, sl, cnt);
    for (int j = 0; j < cnt; ++j) phase1(io, base + j, j);
    if (finish) { RegPowTab pt; dcb_finish(pt, io, cnt); }
  }
}

// The four-lane group operations (quad_ops.hpp) with the lanes emulated: a quad is an array of four values, a DPP
// quad_perm an index permutation; the per-lane arithmetic is the device's own (curve.hpp gq_*_own), so the bounds build
// proves the operand bounds of both rounds of products, chained (products feed the next operation).
namespace {
struct Quad { fe l[4]; };
template <int P0, int P1, int P2, int P3> Quad qperm(const Quad& v) { return Quad{{v.l[P0], v.l[P1], v.l[P2], v.l[P3]}}; }
Quad q_double_neg(const Quad& v) {
  const Quad opb = qperm<0, 1, 2, 2>(v);
  Quad m1, w, r;
  for (int role = 0; role < 4; ++role) m1.l[role] = fe_mul(v.l[role], opb.l[role]);
  const Quad u = qperm<0, 0, 0, 3>(m1), b = qperm<1, 1, 1, 3>(m1), c = qperm<2, 2, 2, 2>(m1);
  for (int role = 0; role < 4; ++role) w.l[role] = gq_double_own(role, u.l[role], b.l[role], c.l[role]);
  const Quad oa = qperm<3, 0, 2, 3>(w), ob = qperm<2, 1, 0, 1>(w);
  for (int role = 0; role < 4; ++role) r.l[role] = fe_mul(oa.l[role], ob.l[role]);
  return r;
}
Quad q_cached(const ge& p) {                                 // gq_store_cached's record: Y-X, Y+X (both carried), 2dT, Z
  return Quad{{fe_sub(p.y, p.x), fe_carry(fe_add(p.y, p.x)), fe_mul(fe_const(FE_K), p.t), p.z}};
}
Quad q_add(const Quad& v, const Quad& qrec, bool neg_q) {
  const Quad u0 = qperm<1, 1, 3, 2>(v), v0 = qperm<0, 0, 3, 2>(v);
  Quad m1, w, r;
  for (int role = 0; role < 4; ++role) {
    const int slot = (role < 2 && neg_q) ? (role ^ 1) : role;
    m1.l[role] = fe_mul(gq_add_in_own(role, u0.l[role], v0.l[role]), qrec.l[slot]);
  }
  const Quad u = qperm<1, 1, 3, 3>(m1), b = qperm<0, 0, 2, 2>(m1);
  for (int role = 0; role < 4; ++role) {
    const bool sub = role == 0 || (role >= 2 && ((role == 2) != neg_q));
    w.l[role] = gq_add_own(sub, u.l[role], b.l[role]);
  }
  const Quad oa = qperm<0, 3, 2, 0>(w), ob = qperm<2, 1, 3, 1>(w);
  for (int role = 0; role < 4; ++role) r.l[role] = fe_mul(oa.l[role], ob.l[role]);
  return r;
}
ge q_to_ge(const Quad& v) { ge g; g.x = v.l[0]; g.y = v.l[1]; g.z = v.l[2]; g.t = v.l[3]; return g; }
}  // namespace

extern "C" {
int sim_init() {
  // same construction the init kernels perform on the device
  fe g = fe_const(FE_SQRT_G), ginv = fe_const(FE_SQRT_G_INV), one = fe_const(FE_ONE);
  static const int pw[6] = {0, 8, 16, 24, 32, 40};
  for (int t = 0; t < 6; ++t) {
    fe base = fe_sqr_n(g, pw[t]);
    for (int nu = 0; nu < 256; ++nu) {
      fe v = nu == 0 ? one : fe_pow_u64(base, (uint64_t)nu);
      v = fe_mul_strict(v, one);
      for (int i = 0; i < NL; ++i) g_gtab[((size_t)t * 256 + nu) * GT_STRIDE + i] = v.l[i];
    }
  }
  std::vector<int> owner(1u << S_HASH_BITS, -1);
  int coll = 0;
  fe b39 = fe_sqr_n(ginv, 39);
  for (int nu = 0; nu < 256; ++nu) {
    fe v = nu == 0 ? one : fe_pow_u64(b39, (uint64_t)nu);
    fe c = fe_reduce_once(fe_mul_strict(v, one));  // canonical Montgomery value in [0, q)
    fe cq = full_norm(fe_add(c, fe_const(Q_LIMBS)));   // the other tight representation
    const int nrep = (c.l[NL - 1] < (1u << 16)) ? 2 : 1;   // x + q only when x < 2^248
    for (int rep = 0; rep < nrep; ++rep) {
      fe k = rep ? cq : c;
      uint32_t h = s_hash_raw(k);
      if (owner[h] >= 0 && owner[h] != nu) ++coll;
      owner[h] = nu; g_slook[h] = (uint8_t)nu;
    }
  }
  g_collisions = coll;
  g_T.gtab = g_gtab.data(); g_T.s_lookup = g_slook.data();
  // fixed-base comb: affine cached j * 256^i * B (the k_init_fbase construction)
  ge pi = ge_generator();
  for (int i = 0; i < FB_WINDOWS; ++i) {
    ge acc = ge_identity();
    for (int j = 0; j < FB_ENTRIES; ++j) {
      const fe zi = fe_invert(acc.z);
      const fe x = fe_mul(acc.x, zi), y = fe_mul(acc.y, zi);
      const fe ypx = fe_carry(fe_add(y, x)), ymx = fe_sub(y, x), kt = fe_mul(fe_mul(fe_const(FE_K), x), y);
      uint32_t* q = g_fbase.data() + ((size_t)i * FB_ENTRIES + j) * 27;
      for (int k = 0; k < 9; ++k) { q[k] = ypx.l[k]; q[9 + k] = ymx.l[k]; q[18 + k] = kt.l[k]; }
      acc = ge_add(acc, pi);
    }
    for (int j = 0; j < FB_BITS; ++j) pi = ge_double(pi);
  }
  return coll;
}
// field ops on Montgomery-256 words (8 x u32 per element), through the internal form
void sim_fq_mul(const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) fe_to_mont256_words(fe_mul(fe_from_mont256_words(a + 8 * i), fe_from_mont256_words(b + 8 * i)), out + 8 * i);
}
void sim_fq_sqr(const uint32_t* a, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) fe_to_mont256_words(fe_sqr(fe_from_mont256_words(a + 8 * i)), out + 8 * i);
}
void sim_fq_add(const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) fe_to_mont256_words(fe_add(fe_from_mont256_words(a + 8 * i), fe_from_mont256_words(b + 8 * i)), out + 8 * i);
}
void sim_fq_sub(const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) fe_to_mont256_words(fe_sub(fe_from_mont256_words(a + 8 * i), fe_from_mont256_words(b + 8 * i)), out + 8 * i);
}
void sim_fq_from_bytes(const uint32_t* w, size_t n, uint32_t* mont256) {
  for (size_t i = 0; i < n; ++i) fe_to_mont256_words(fe_from_words_mod_order(w + 8 * i), mont256 + 8 * i);
}
void sim_fq_to_bytes(const uint32_t* mont256, size_t n, uint32_t* w) {
  for (size_t i = 0; i < n; ++i) fe_to_bytes_words(fe_from_mont256_words(mont256 + 8 * i), w + 8 * i);
}
// raw limb-level entry points (9 x u32 per element) for the bound stress tests
static fe raw_load(const uint32_t* p) {      // bounds = the actual limbs (these tests feed the extremes themselves)
  fe x; memcpy(x.l, p, 36);
#if defined(D377_BOUNDS)
  for (int i = 0; i < NL; ++i) x.ub[i] = x.l[i];
  x.vq = ((double)x.l[NL - 1] + 2.0) / Q_TOP;
#endif
  return x;
}
// mode: 0 fe_mul, 1 fe_mul_strict, 2 fe_sqr, 3 fe_sqr_strict, 4 fe_sqr2x
void sim_raw_mul(int mode, const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) {
    fe x = raw_load(a + 9 * i), y = raw_load(b + 9 * i), r;
    switch (mode) {
      case 0: r = fe_mul(x, y); break;
      case 1: r = fe_mul_strict(x, y); break;
      case 2: r = fe_sqr(x); break;
      case 3: r = fe_sqr_strict(x); break;
      default: r = fe_sqr2x(x); break;
    }
    memcpy(out + 9 * i, r.l, 36);
  }
}
void sim_raw_sub(int nc, const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) { fe x = raw_load(a + 9 * i), y = raw_load(b + 9 * i); fe r = nc ? fe_sub_nc(x, y) : fe_sub(x, y); memcpy(out + 9 * i, r.l, 36); }
}
void sim_raw_canon(const uint32_t* a, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) { fe x = raw_load(a + 9 * i); fe r = fe_canon(x); memcpy(out + 9 * i, r.l, 36); }
}
void sim_consts(uint32_t* sub16q, uint32_t* sub16q_nc, uint32_t* ql) { for (int i = 0; i < NL; ++i) { sub16q[i] = SUB32Q[i]; sub16q_nc[i] = SUB16Q_NC[i]; ql[i] = QL[i]; } }

void sim_sqrt_ratio_zeta(const uint32_t* num, const uint32_t* den, size_t n, uint32_t* root, uint8_t* ws) {
  dcb_rounds<1>(n, root, false,
    [&](HostDcbIO& io, size_t i, int j) { dcb_put_den(io, 0, j, fe_from_words_mod_order_strict(den + 8 * i)); },
    [&](HostDcbIO& io, size_t i, int j) {
      const fe inv = dcb_get_inv(io, 0, j);
      RegPowTab pt; fe r; bool w = fe_sqrt_ratio_zeta<false>(g_T, pt, fe_from_words_mod_order_strict(num + 8 * i), fe_from_words_mod_order_strict(den + 8 * i), &r, false, &inv);
      fe_to_bytes_words(r, root + 8 * i); ws[i] = w;
    });
}
// the reference's own inversion-free form (invsqrt.rs:88-94), element by element
void sim_sqrt_ratio_zeta_plain(const uint32_t* num, const uint32_t* den, size_t n, uint32_t* root, uint8_t* ws) {
  for (size_t i = 0; i < n; ++i) {
    RegPowTab pt; fe r; bool w = fe_sqrt_ratio_zeta<false>(g_T, pt, fe_from_words_mod_order_strict(num + 8 * i), fe_from_words_mod_order_strict(den + 8 * i), &r);
    fe_to_bytes_words(r, root + 8 * i); ws[i] = w;
  }
}
static void ge_store256(const ge& g, uint32_t* o) {
  fe_to_mont256_words(g.x, o); fe_to_mont256_words(g.y, o + 8); fe_to_mont256_words(g.z, o + 16); fe_to_mont256_words(g.t, o + 24);
}
static ge ge_load256(const uint32_t* o) {
  ge g; g.x = fe_from_mont256_words(o); g.y = fe_from_mont256_words(o + 8); g.z = fe_from_mont256_words(o + 16); g.t = fe_from_mont256_words(o + 24);
  return g;
}
// ge_double (reference form), ge_double_fast and ge_double_latency on Montgomery-256 records
void sim_double_variants(const uint32_t* xyzt, size_t n, uint32_t* ref, uint32_t* fast, uint32_t* negd) {
  for (size_t i = 0; i < n; ++i) {
    ge g = ge_load256(xyzt + 32 * i);
    ge_store256(ge_double(g), ref + 32 * i);
    ge_store256(ge_double_fast(g, true), fast + 32 * i);
    ge_store256(ge_double_neg(g, true), negd + 32 * i);      // -[2]P
  }
}
// dbl4 = [4]P (two sign-folded doublings), sum = [4]P + Q, diff = [4]P - Q, chain = [2]([4]P + Q) - Q on the four-lane forms
void sim_quad_forms(const uint32_t* p, const uint32_t* q, size_t n, uint32_t* dbl4, uint32_t* sum, uint32_t* diff, uint32_t* chain) {
  for (size_t i = 0; i < n; ++i) {
    const ge a = ge_load256(p + 32 * i), b = ge_load256(q + 32 * i);
    const Quad rec = q_cached(b);
    const Quad d = q_double_neg(q_double_neg(Quad{{a.x, a.y, a.z, a.t}}));
    ge_store256(q_to_ge(d), dbl4 + 32 * i);
    const Quad s1 = q_add(d, rec, false);
    ge_store256(q_to_ge(s1), sum + 32 * i);
    ge_store256(q_to_ge(q_add(d, rec, true)), diff + 32 * i);
    // -(2 s1) - (-Q) = -(2 s1 - Q): the chains' way of absorbing the doubling's sign
    ge_store256(ge_neg(q_to_ge(q_add(q_double_neg(s1), rec, false))), chain + 32 * i);
    (void)q_add(q_add(s1, rec, true), q_cached(q_to_ge(s1)), false);      // sums of sums (the trees), a sum as the cached operand
  }
}
// The lane-spread kernels' way in and out (msm.hip k_msm_final / k_msm_wsum_window / k_msm_tiny, d377.hip
// k_scalar_mul_var_tiny): a point to four records of 10 x 28-bit limbs and back, and then everything those kernels do with
// a point that came back -- the doubling of the final result, its negation, the encoder's state, the cached forms for the
// lane and quad chains, the table phase of a square root whose powers came back the same way.  `slack` is added to every
// limb of the records first (the rows hand back lazily reduced limbs; 0 <= slack < 2^12), with the value it adds removed
// from limb 0's neighbour so that the element stays the same: limbs (l0 + s, l1 - ... ) is not expressible in general, so
// the record is re-split instead: limb k gives up 2^28 to limb k-1 wherever it can.
void sim_row_records(const uint32_t* p, size_t n, int lazy, uint32_t* back, uint32_t* dbl, uint32_t* enc) {
  for (size_t i = 0; i < n; ++i) {
    const ge a = ge_load256(p + 32 * i);
    uint32_t rec[4][16];
    fe_to_limbs28(a.x, rec[0]); fe_to_limbs28(a.y, rec[1]); fe_to_limbs28(a.z, rec[2]); fe_to_limbs28(a.t, rec[3]);
    if (lazy)                                                    // the same integers with limbs above 2^28: borrow downwards
      for (int r = 0; r < 4; ++r)
        for (int k = 9; k >= 1; --k)
          if (rec[r][k] > 0) { rec[r][k] -= 1; rec[r][k - 1] += 1u << 28; }
    ge g;
    g.x = fe_from_limbs28(rec[0]); g.y = fe_from_limbs28(rec[1]); g.z = fe_from_limbs28(rec[2]); g.t = fe_from_limbs28(rec[3]);
    ge_store256(g, back + 32 * i);
    ge_store256(ge_double(g), dbl + 32 * i);
    (void)ge_neg(g);
    const Quad rec4 = q_cached(g);                               // gq_store_cached's record
    (void)q_add(q_double_neg(Quad{{g.x, g.y, g.z, g.t}}), rec4, true);
    uint32_t cached[16];                                         // rq_store_cached: lazy sums go through fe_canon
    fe_to_limbs28(fe_sub(g.y, g.x), cached); fe_to_limbs28(fe_add(g.y, g.x), cached); fe_to_limbs28(fe_mul(fe_const(FE_K), g.t), cached);
    (void)ge_add(g, ge_double(g));                               // partial results summed by k_msm_small_sum / k_msm_combine
    HostDcbIO io;                                                // msm_emit_doubled: the encoding of [2]g without a square root
    io.out = enc; io.base = i;
    dcb_put(io, 0, ge_dcb_from_half(g, false));
    RegPowTab pt;
    dcb_finish(pt, io, 1);
    fe root;
    (void)fe_sqrt_tail(g_T, g.x, g.y, false, false, &root, false);   // powers that came back from the rows into the table phase
  }
}
// The whole-element steps of the four-elements-per-wave kernels (d377.hip k_*_tiny): the square root's powers arrive as VALUES
// (GivenPowers: raised on the rows of the wave, back through a 28-bit-limb record) instead of being raised here, and the
// encodings of four elements share one inversion (tiny4_encode: the product of the four p's by exchanges inside the quad).
// enc_a: encode_to_curve with the powers handed over, each element finished by itself (dcb_finish); enc_b: the same
// states through the four-way product and dcb_encode_one; xyzt: decompress(enc_a) and status, powers handed over again.
void sim_tiny4(const uint32_t* r0, size_t n, uint32_t* enc_a, uint32_t* enc_b, uint32_t* xyzt, uint8_t* status) {
  auto powers_of = [](const fe& den) {                              // invsqrt.rs:88-94 for num = 1, as row_sqrt_powers computes them
    RegPowTab pt;
    const fe s = fe_pow_2_47_m1(den), t_ = fe_mul(fe_sqr(s), den), w = fe_mul(fe_pow_m12(t_, pt), s);
    uint32_t rec[16];
    GivenPowers g;
    fe_to_limbs28(fe_mul(w, den), rec); g.v = fe_from_limbs28(rec);
    fe_to_limbs28(w, rec); g.uv = fe_from_limbs28(rec);
    return g;
  };
  for (size_t i = 0; i + 4 <= n; i += 4) {
    dcb_state st[4];
    for (int j = 0; j < 4; ++j) {
      const fe r = fe_from_words_mod_order(r0 + 8 * (i + j));
      GivenPowers gp = powers_of(ge_elligator_den(r));
      fe s, t;
      ge_elligator_st(g_T, gp, r, &s, &t);
      st[j] = ge_dcb_from_jacobi_st(s, t);
      HostDcbIO io; io.out = enc_a; io.base = i + j;
      dcb_put(io, 0, st[j]);
      RegPowTab pt;
      dcb_finish(pt, io, 1);
    }
    fe ab[4];
    for (int j = 0; j < 4; ++j) ab[j] = fe_mul_strict(st[j].p, st[j ^ 1].p);
    const fe inv = fe_invert(fe_mul_strict(ab[0], ab[2]));
    for (int j = 0; j < 4; ++j) dcb_encode_one(st[j], fe_mul(fe_mul(inv, ab[j ^ 2]), st[j ^ 1].p), enc_b + 8 * (i + j));
    for (int j = 0; j < 4; ++j) {
      const uint32_t* w = enc_a + 8 * (i + j);
      GivenPowers gp = powers_of(ge_decompress_den(w));
      ge g;
      const uint32_t bad = ge_decompress(g_T, gp, w, &g);
      status[i + j] = (uint8_t)bad;
      ge_store256(g, xyzt + 32 * (i + j));
      gp = powers_of(ge_compress_den(g));                          // ... and the generic compressor on the same terms
      uint32_t back[8];
      ge_compress(g_T, gp, g, back, bad == 0);
      if (memcmp(back, w, 32) != 0) status[i + j] |= 2;
    }
  }
}
// reference-form addition and negation (the API kernels k_add / k_neg / k_hash_to_curve use them)
void sim_group_misc(const uint32_t* p, const uint32_t* q, size_t n, uint32_t* sum, uint32_t* neg) {
  for (size_t i = 0; i < n; ++i) {
    ge a = ge_load256(p + 32 * i), b = ge_load256(q + 32 * i);
    ge_store256(ge_add(a, b), sum + 32 * i);
    ge_store256(ge_neg(a), neg + 32 * i);
    (void)fe_eq(fe_mul(a.x, b.y), fe_mul(b.x, a.y));
    (void)fe_invert(a.z);
  }
}
// the element-wise kernels' conversion-free forms (k_add, k_double, k_eq, k_neg, k_fq_op in d377.hip)
void sim_raw_forms(const uint32_t* p, const uint32_t* q, size_t n, uint32_t* sum, uint32_t* dbl, uint32_t* neg, uint8_t* eq,
                   uint8_t* neg_ok, uint32_t* fmul, uint32_t* fsqr, uint32_t* fadd, uint32_t* fsub, uint32_t* fneg, uint8_t* f_ok) {
  for (size_t i = 0; i < n; ++i) {
    const uint32_t* a = p + 32 * i;
    const uint32_t* b = q + 32 * i;
    ge_add_raw_words(a, b, false, sum + 32 * i);
    ge_double_raw_words(a, dbl + 32 * i);
    neg_ok[i] = ge_neg_words(a, neg + 32 * i) ? 1 : 0;
    eq[i] = ge_eq_raw_words(a, b) ? 1 : 0;
    // field ops on the X words of the two records
    fe_scaled_to_mont256_words(fe_mul(fe_from_words(a), fe_from_words(b)), FE_RAW2_TO_MONT256, fmul + 8 * i);
    fe_scaled_to_mont256_words(fe_sqr(fe_from_words(a)), FE_RAW2_TO_MONT256, fsqr + 8 * i);
    bool ok = fq_addsub_words(a, b, false, fadd + 8 * i);
    ok = fq_addsub_words(a, b, true, fsub + 8 * i) && ok;
    ok = fq_neg_words(a, fneg + 8 * i) && ok;
    f_ok[i] = ok ? 1 : 0;
  }
}
// msm.hip's per-lane chain for one bucket run, twice, and the reduction level above it: points -> cached affine
// records (a negative digit exchanges y+x / y-x as the loader does), the first point of a run taken as is
// (ge_from_cached_affine), mixed additions for the rest, then one full addition of the two partial sums.
// xyzt: n x 32 words (Z = 1 records: decompress output), negs: one byte per point.  n >= 2.
void sim_msm_bucket(const uint32_t* xyzt, const uint8_t* negs, size_t n, uint32_t* out) {
  auto run = [&](size_t lo, size_t hi) {
    ge acc = ge_identity();
    for (size_t i = lo; i < hi; ++i) {
      const ge p = ge_load256(xyzt + 32 * i);
      gea q = gea_from_affine(p.x, p.y);
      const bool neg = negs[i] != 0;
      if (neg) { fe t = q.ypx; q.ypx = q.ymx; q.ymx = t; }
      acc = (i == lo) ? ge_from_cached_affine(q, neg) : ge_add_affine(acc, q, neg, true);
    }
    return acc;
  };
  const ge a = run(0, n / 2), b = run(n / 2, n);
  ge_store256(ge_add(ge_add(a, b), ge_add(a, a)), out);          // a + b + 2a: partial sums meet in every position of ge_add
}
// Element - Element as k_add does it with negate = 1
void sim_raw_ge_sub(const uint32_t* p, const uint32_t* q, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) {
    ge_add_raw_words(p + 32 * i, q + 32 * i, true, out + 32 * i);
  }
}
// normalize_batch as one lane of k_to_affine walks it: all n records share one inversion
void sim_to_affine_raw(const uint32_t* xyzt, size_t n, uint32_t* xy) {
  std::vector<fe> prefix(n);
  fe p = fe_const(FE_ONE);
  for (size_t i = 0; i < n; ++i) {
    bool zz;
    const fe z = affine_raw_z(xyzt + 32 * i + 16, &zz);
    prefix[i] = p;
    p = fe_mul(p, z);
  }
  fe inv = fe_mul(fe_invert(p), fe_const(FE_TO_MONT256));
  for (size_t i = n; i-- > 0;) {
    bool zz;
    const fe z = affine_raw_z(xyzt + 32 * i + 16, &zz);
    const fe zi = fe_mul(inv, prefix[i]);
    inv = fe_mul(inv, z);
    affine_raw_finish(zi, xyzt + 32 * i, xyzt + 32 * i + 8, zz, xy + 16 * i);
  }
}
// Fq inverse three ways on Montgomery-256 words: divsteps (fe_invert, what the kernels use), the x^(q-2) ladder and the
// chain built from the square root's exponentiations
void sim_invert(const uint32_t* a, size_t n, uint32_t* gcd, uint32_t* ladder, uint32_t* chain) {
  for (size_t i = 0; i < n; ++i) {
    const fe x = fe_from_mont256_words(a + 8 * i);
    RegPowTab pt;
    fe_to_mont256_words(fe_invert(x), gcd + 8 * i);
    fe_to_mont256_words(fe_invert_pow(x), ladder + 8 * i);
    fe_to_mont256_words(fe_invert_chain(x, pt), chain + 8 * i);
  }
}
// the plain modular inverse on 29-bit limbs (integers in [0, q))
void sim_modinv_limbs29(const uint32_t* x, size_t n, uint32_t* y) {
  for (size_t i = 0; i < n; ++i) modinv_limbs29(x + 9 * i, y + 9 * i);
}
// decompression / compression / round trip with batched inverses for their square roots (the form the variable-base
// kernel uses for its decompression; the stand-alone kernels keep the inversion-free form, see d377.hip)
void sim_decompress_assisted(const uint32_t* enc, size_t n, uint32_t* xyzt, uint8_t* st) {
  std::vector<uint32_t> dummy(8 * (n + 1));
  dcb_rounds<1>(n, dummy.data(), false,
    [&](HostDcbIO& io, size_t i, int j) { dcb_put_den(io, 0, j, ge_decompress_den(enc + 8 * i)); },
    [&](HostDcbIO& io, size_t i, int j) {
      const fe inv = dcb_get_inv(io, 0, j);
      RegPowTab pt; ge g; uint32_t bad = ge_decompress(g_T, pt, enc + 8 * i, &g, &inv);
      st[i] = (uint8_t)bad;
      if (bad) memset(xyzt + 32 * i, 0, 128); else ge_store256(g, xyzt + 32 * i);
    });
}
void sim_compress_assisted(const uint32_t* xyzt, size_t n, uint32_t* enc) {
  dcb_rounds<1>(n, enc, false,
    [&](HostDcbIO& io, size_t i, int j) { dcb_put_den(io, 0, j, ge_compress_den(ge_load256(xyzt + 32 * i))); },
    [&](HostDcbIO& io, size_t i, int j) {
      const fe inv = dcb_get_inv(io, 0, j);
      RegPowTab pt; ge_compress(g_T, pt, ge_load256(xyzt + 32 * i), enc + 8 * i, true, &inv);
    });
}
// the kernels' form: one element at a time, inversion-free square roots
void sim_compress(const uint32_t* xyzt, size_t n, uint32_t* enc) {
  for (size_t i = 0; i < n; ++i) { RegPowTab pt; ge_compress(g_T, pt, ge_load256(xyzt + 32 * i), enc + 8 * i); }
}
void sim_roundtrip_assisted(const uint32_t* enc, size_t n, uint32_t* out, uint8_t* st) {
  dcb_rounds<1>(n, out, false,
    [&](HostDcbIO& io, size_t i, int j) { dcb_put_den(io, 0, j, ge_decompress_den(enc + 8 * i)); },
    [&](HostDcbIO& io, size_t i, int j) {
      const fe inv = dcb_get_inv(io, 0, j);
      RegPowTab pt; ge g; uint32_t bad = ge_decompress(g_T, pt, enc + 8 * i, &g, &inv);
      st[i] = (uint8_t)bad;
      if (bad) memset(out + 8 * i, 0, 32); else ge_compress(g_T, pt, g, out + 8 * i);
    });
}
// the chunked round trip of d377.hip (k_roundtrip_chunked): the decoding pass takes the inverses of its denominators, leaves
// (X, Y) of each point and the compressor's denominator behind; those are inverted together and the compressor runs with them
void sim_roundtrip_chunked(const uint32_t* enc, size_t n, uint32_t* out, uint8_t* st) {
  for (size_t base = 0; base < n; base += DCB_K) {
    HostDcbIO io; io.out = out; io.base = base;
    const int cnt = (int)((n - base) < (size_t)DCB_K ? (n - base) : (size_t)DCB_K);
    fe xs[DCB_K], ys[DCB_K];
    for (int j = 0; j < cnt; ++j) dcb_put_den(io, 0, j, ge_decompress_den(enc + 8 * (base + j)));
    dcb_invert_slot(io, 0, cnt);
    for (int j = 0; j < cnt; ++j) {
      const fe inv = dcb_get_inv(io, 0, j);
      RegPowTab pt; ge g; const uint32_t bad = ge_decompress(g_T, pt, enc + 8 * (base + j), &g, &inv);
      st[base + j] = (uint8_t)bad;
      if (bad) g = ge_identity();
      xs[j] = g.x; ys[j] = g.y;
      dcb_put_den(io, 0, j, ge_compress_den(g));
    }
    dcb_invert_slot(io, 0, cnt);
    for (int j = 0; j < cnt; ++j) {
      const fe inv = dcb_get_inv(io, 0, j);
      ge g; g.x = xs[j]; g.y = ys[j]; g.z = fe_const(FE_ONE); g.t = fe_mul(g.x, g.y);
      RegPowTab pt; uint32_t w[8];
      ge_compress(g_T, pt, g, w, st[base + j] == 0, &inv);
      if (st[base + j]) memset(out + 8 * (base + j), 0, 32); else memcpy(out + 8 * (base + j), w, 32);
    }
  }
}
void sim_encode_to_curve(const uint32_t* r0, size_t n, uint32_t* enc, uint32_t* xyzt) {
  dcb_rounds<1>(n, enc, true,
    [&](HostDcbIO& io, size_t i, int j) { dcb_put_den(io, 0, j, ge_elligator_den(fe_from_words_mod_order(r0 + 8 * i))); },
    [&](HostDcbIO& io, size_t i, int j) {
      const fe inv = dcb_get_inv(io, 0, j);
      RegPowTab pt; fe s, t;
      ge_elligator_st(g_T, pt, fe_from_words_mod_order(r0 + 8 * i), &s, &t, &inv);
      if (xyzt) ge_store256(ge_from_jacobi_st(s, t), xyzt + 32 * i);
      dcb_put(io, j, ge_dcb_from_jacobi_st(s, t));
    });
}
// the same through the generic compressor and the inversion-free square roots
void sim_encode_to_curve_sqrt(const uint32_t* r0, size_t n, uint32_t* enc) {
  for (size_t i = 0; i < n; ++i) {
    RegPowTab pt;
    ge_compress(g_T, pt, ge_elligator_map(g_T, pt, fe_from_words_mod_order(r0 + 8 * i)), enc + 8 * i);
  }
}
void sim_scalar_mul_var(const uint32_t* enc, const uint32_t* k, size_t n, uint32_t* out, uint8_t* st) {
  dcb_rounds<1>(n, out, true,
    [&](HostDcbIO& io, size_t i, int j) { dcb_put_den(io, 0, j, ge_decompress_den(enc + 8 * i)); },
    [&](HostDcbIO& io, size_t i, int j) {
      const fe inv = dcb_get_inv(io, 0, j);
      RegPowTab pt; ge g; uint32_t bad = ge_decompress(g_T, pt, enc + 8 * i, &g, &inv);
      st[i] = (uint8_t)bad;
      uint32_t kk[8], dg[8]; memcpy(kk, k + 8 * i, 32); fr_reduce_words(kk); fr_half_words(kk); fr_recode_signed16(kk, dg);
      HostTab tab; ge r = ge_scalar_mul_w4(g, dg, tab, DCB_WANT_T);  // [k/2]P; the state is that of its double
      dcb_put(io, j, ge_dcb_from_half(r, bad != 0));
    });
}
// d377_batch_msm_small's lane kernel (batch_msm.hip): n sums of m terms, Elements in (Montgomery-256 records), encodings out;
// one host "lane" walks the sums in rounds as a device lane does, the Straus chain is the device's own (straus.hpp)
struct HostStrausTab {
  gec e[8][9];
  uint32_t dig[BM_WINDOWS];
  void store(int p, int j, const gec& g) { e[p][j] = g; }
  gec load(int p, int j, bool swap) const { gec c = e[p][j]; if (swap) { fe t = c.ypx; c.ypx = c.ymx; c.ymx = t; } return c; }
  void dig_store(int w, uint32_t v) { dig[w] = v; }
  uint32_t dig_load(int w) const { return dig[w]; }
};
void sim_batch_msm(const uint32_t* xyzt, const uint32_t* k, int m, size_t n, uint32_t* out) {
  dcb_rounds<0>(n, out, true,
    [&](HostDcbIO&, size_t, int) {},
    [&](HostDcbIO& io, size_t i, int j) {
      HostStrausTab tab;
      const size_t first = i * (size_t)m;
      const ge r = straus_sum(tab, m,
        [&](int p, uint32_t kk[8]) { memcpy(kk, k + 8 * (first + p), 32); },
        [&](int p, ge* g) -> bool { *g = ge_load256(xyzt + 32 * (first + p)); return fe_is_zero(g->z); }, DCB_WANT_T);
      dcb_put(io, j, ge_dcb_from_half(r, false));
    });
}
void sim_scalar_mul_var_sqrt(const uint32_t* enc, const uint32_t* k, size_t n, uint32_t* out, uint8_t* st) {
  for (size_t i = 0; i < n; ++i) {
    RegPowTab pt; ge g; uint32_t bad = ge_decompress(g_T, pt, enc + 8 * i, &g);
    st[i] = (uint8_t)bad;
    if (bad) { memset(out + 8 * i, 0, 32); continue; }
    uint32_t kk[8], dg[8]; memcpy(kk, k + 8 * i, 32); fr_reduce_words(kk); fr_recode_signed16(kk, dg);
    HostTab tab; ge r = ge_scalar_mul_w4(g, dg, tab);
    ge_compress(g_T, pt, r, out + 8 * i);
  }
}
// The MSM's integer plan (msm_plan.hpp): the windows for width c, and the signed digits of n scalars exactly as the prepare
// kernels write them (scalar mod r, halved mod r, recoded).  shape = {W, nwide}; digits: n rows of 64 ints (W <= 63 used).
void sim_msm_digits(const uint32_t* k, size_t n, int c, int* shape, int* digits) {
  const WinShape ws = win_shape(c);
  shape[0] = ws.W; shape[1] = ws.nwide;
  for (size_t i = 0; i < n; ++i) {
    uint32_t w[8];
    for (int j = 0; j < 8; ++j) w[j] = k[8 * i + j];
    fr_reduce_words(w);
    fr_half_words(w);
    uint32_t carry = 0;
    for (int q = 0; q < ws.W; ++q) digits[64 * i + q] = msm_digit(w, q, ws, carry);
    digits[64 * i + 63] = (int)carry;                   // must end as 0
  }
}
// lanes that touch a bucket holding the entries [o, o + size) of its window when every lane takes L consecutive entries
void sim_span_plan(const uint32_t* o, const uint32_t* size, size_t nb, uint32_t L, uint32_t* first_lane, uint32_t* partials) {
  for (size_t b = 0; b < nb; ++b) { first_lane[b] = span_first_lane(o[b], L); partials[b] = span_partials(o[b], size[b], L); }
}
void sim_scalar_mul_base(const uint32_t* k, size_t n, uint32_t* out) {
  HostFTab ft{g_fbase.data()};
  dcb_rounds<0>(n, out, true,
    [&](HostDcbIO&, size_t, int) {},
    [&](HostDcbIO& io, size_t i, int j) {
      uint32_t kk[8]; memcpy(kk, k + 8 * i, 32); fr_reduce_words(kk); fr_half_words(kk);
      dcb_put(io, j, ge_dcb_from_half(ge_scalar_mul_base_w8(kk, ft, DCB_WANT_T), false));
    });
}
void sim_decompress(const uint32_t* enc, size_t n, uint32_t* xyzt, uint8_t* st) {
  for (size_t i = 0; i < n; ++i) {
    RegPowTab pt; ge g; uint32_t bad = ge_decompress(g_T, pt, enc + 8 * i, &g);
    st[i] = (uint8_t)bad;
    if (bad) memset(xyzt + 32 * i, 0, 128); else ge_store256(g, xyzt + 32 * i);
  }
}
void sim_roundtrip(const uint32_t* enc, size_t n, uint32_t* out, uint8_t* st) {
  for (size_t i = 0; i < n; ++i) {
    RegPowTab pt; ge g; uint32_t bad = ge_decompress(g_T, pt, enc + 8 * i, &g);
    st[i] = (uint8_t)bad;
    if (bad) memset(out + 8 * i, 0, 32); else ge_compress(g_T, pt, g, out + 8 * i);
  }
}
// hash_to_curve: two Elligator maps (their two square roots take batched inverses), an addition, the generic compressor
void sim_hash_to_curve(const uint32_t* r1, const uint32_t* r2, size_t n, uint32_t* enc) {
  dcb_rounds<2>(n, enc, false,
    [&](HostDcbIO& io, size_t i, int j) {
      dcb_put_den(io, 0, j, ge_elligator_den(fe_from_words_mod_order(r1 + 8 * i)));
      dcb_put_den(io, 1, j, ge_elligator_den(fe_from_words_mod_order(r2 + 8 * i)));
    },
    [&](HostDcbIO& io, size_t i, int j) {
      const fe i1 = dcb_get_inv(io, 0, j), i2 = dcb_get_inv(io, 1, j);
      RegPowTab pt;
      ge a = ge_elligator_map(g_T, pt, fe_from_words_mod_order(r1 + 8 * i), &i1);
      ge b = ge_elligator_map(g_T, pt, fe_from_words_mod_order(r2 + 8 * i), &i2);
      ge_compress(g_T, pt, ge_add(a, b), enc + 8 * i);
    });
}
// k_hash_to_curve as the kernel does it now: both maps as far as (s, t), the sum on the Jacobi quartic, the square-root-free
// compressor; exceptional pairs by the reference's route.  force_exceptional: every pair takes that route (its plumbing).
// st1 / st2 (optional): instead of mapping r1 / r2, take these (s, t) pairs (4 x 8 words: canonical s1, t1, s2, t2 per
// element) -- lets a test hand in points that hit the addition law's exceptional case.
void sim_hash_to_curve_quartic(const uint32_t* r1, const uint32_t* r2, size_t n, uint32_t* enc, int force_exceptional,
                               const uint32_t* st_pairs, uint8_t* exceptional_out) {
  dcb_rounds<2>(n, enc, true,
    [&](HostDcbIO& io, size_t i, int j) {
      dcb_put_den(io, 0, j, ge_elligator_den(fe_from_words_mod_order(r1 + 8 * i)));
      dcb_put_den(io, 1, j, ge_elligator_den(fe_from_words_mod_order(r2 + 8 * i)));
    },
    [&](HostDcbIO& io, size_t i, int j) {
      const fe i1 = dcb_get_inv(io, 0, j), i2 = dcb_get_inv(io, 1, j);
      RegPowTab pt;
      fe s1, t1, s2, t2;
      ge_elligator_st(g_T, pt, fe_from_words_mod_order(r1 + 8 * i), &s1, &t1, &i1);
      ge_elligator_st(g_T, pt, fe_from_words_mod_order(r2 + 8 * i), &s2, &t2, &i2);
      if (st_pairs) {
        s1 = fe_from_words_mod_order(st_pairs + 32 * i); t1 = fe_from_words_mod_order(st_pairs + 32 * i + 8);
        s2 = fe_from_words_mod_order(st_pairs + 32 * i + 16); t2 = fe_from_words_mod_order(st_pairs + 32 * i + 24);
      }
      bool exc;
      dcb_state st = ge_dcb_from_jacobi_sum(s1, t1, s2, t2, &exc);
      if (exceptional_out) exceptional_out[i] = exc ? 1 : 0;
      if (exc || force_exceptional) {
        uint32_t w[8];
        ge_compress(g_T, pt, ge_add(ge_from_jacobi_st(s1, t1), ge_from_jacobi_st(s2, t2)), w);
        st = dcb_from_encoding_words(w);
      }
      dcb_put(io, j, st);
    });
}
void sim_fr_half(const uint32_t* k, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) { memcpy(out + 8 * i, k + 8 * i, 32); fr_reduce_words(out + 8 * i); fr_half_words(out + 8 * i); }
}
#if defined(D377_BOUNDS)
// products / squarings executed since the last call (bounds build only): bench.py's per-element MAC counts
void sim_op_counts(unsigned long* mul, unsigned long* sqr) {
  *mul = op_counts().mul; *sqr = op_counts().sqr;
  op_counts().mul = 0; op_counts().sqr = 0;
}
#endif
// the kernels' Fr arithmetic (k_fr_op / k_fr_from_wide in d377.hip): op codes of include/decaf377_amd.h
void sim_fr_op(int op, const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out, uint8_t* st) {
  for (size_t i = 0; i < n; ++i) {
    uint32_t x[8], y[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r[8];
    memcpy(x, a + 8 * i, 32); fr_reduce_words(x);
    if (op <= 2) { memcpy(y, b + 8 * i, 32); fr_reduce_words(y); }
    st[i] = 0;
    switch (op) {
      case 0: fr_addmod(x, y, r); break;
      case 1: fr_submod(x, y, r); break;
      case 2: fr_mulmod(x, y, r); break;
      case 3: fr_mulmod(x, x, r); break;
      case 4: fr_submod(y, x, r); break;
      default: st[i] = fr_invmod(x, r) ? 0 : 1; break;
    }
    memcpy(out + 8 * i, r, 32);
  }
}
void sim_fr_from_wide(const uint32_t* in, int len, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) {
    uint32_t lo[8], hi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    memcpy(lo, in + (size_t)(len / 4) * i, 32);
    memcpy(hi, in + (size_t)(len / 4) * i + 8, (size_t)len - 32);
    fr_from_wide_words(lo, hi, out + 8 * i);
  }
}
void sim_fr_reduce(const uint32_t* k, size_t n, uint32_t* out) {
  for (size_t i = 0; i < n; ++i) { uint32_t kk[8]; memcpy(kk, k + 8 * i, 32); fr_reduce_words(kk); memcpy(out + 8 * i, kk, 32); }
}
}
