// straus.hpp -- the Straus chain of the batched small multiscalar sums (batch_msm.hip): sum_{p < m} k_p P_p with ONE chain of
// 252 doublings shared by the m points, signed 4-bit windows of k / 2 mod r, one table of cached 0 .. 8 P per point.  Plain
// per-lane arithmetic on curve.hpp's formulas, shared by the device kernel and the host simulation (tests/host_sim), whose
// bounds build walks it: the additions chained after additions and doublings here are the sequences the limb bounds must hold for.
#pragma once
#include <stdint.h>

#include "curve.hpp"

namespace d377 {

constexpr int BM_WINDOWS = 64;          // 63 signed 4-bit windows of k / 2 mod r < 2^251 and the recoding's carry (0 or 1)

// nibble p of a digit word: the signed digit of point p in that window
D377_HD int nibble_digit(uint32_t word, int p) { return (int)(((word >> (4 * p)) & 15u) ^ 8u) - 8; }

// ge_add_cached (curve.hpp) with the NEXT table entry requested behind the four products that read the current one: q's
// registers are free from there, and the gather has the addition's other four products (and, at a window's last point, the four
// doublings of the next window) to arrive in.  On the device the products are volatile asm streams, so the fences keep the
// loads behind them.
template <class Reload>
D377_HD ge ge_add_cached_reload(const ge& p, gec& q, bool neg, bool with_t, Reload reload) {
  fe a = fe_mul(fe_sub_nc(p.y, p.x), q.ymx);
  fe b = fe_mul(fe_add(p.y, p.x), q.ypx);
  fe c = fe_mul(p.t, q.kt);
  fe d = fe_mul(p.z, q.z2);
#if defined(__HIPCC__)
  asm volatile("" ::: "memory");
#endif
  reload(q);
#if defined(__HIPCC__)
  asm volatile("" ::: "memory");
#endif
  fe e = fe_sub(b, a), h = fe_add(b, a);
  fe dmc = fe_sub(d, c), dpc = fe_add(d, c);
  fe f = fe_select(neg, dpc, dmc), g = fe_select(neg, dmc, dpc);
  ge r;
  r.x = fe_mul(e, f); r.y = fe_mul(g, h); r.z = fe_mul(f, g);
  r.t = p.t;
  if (with_t) r.t = fe_mul(e, h);
  return r;
}

// The sum of one lane: tables of its m points, the digit words, the shared chain.  -> [1/2] of the sum (the caller encodes the
// double).  Tab: store(p, j, gec) / load(p, j, swap) -> gec (entry j of point p's table; swap: ypx / ymx exchanged, the entry of
// -jP), dig_store(w, word) / dig_load(w) (nibble p of word w = point p's digit in window w).
// load_scalar(p, k[8]): the 32 bytes of scalar p as words.  load_point(p, &g) -> dead (the point contributes nothing: an invalid
// Encoding, a record with Z = 0).  want_t: whether the caller reads T of the result.
template <class Tab, class LoadScalar, class LoadPoint>
D377_HD ge straus_sum(Tab& st, int m, LoadScalar load_scalar, LoadPoint load_point, bool want_t = false) {
  uint32_t deadmask = 0;
#pragma unroll 1
  for (int p = 0; p < m; ++p) {
    ge g;
    if (load_point(p, &g)) deadmask |= 1u << p;
    gec id;
    id.ypx = fe_const(FE_ONE); id.ymx = fe_const(FE_ONE); id.z2 = fe_dbl(fe_const(FE_ONE)); id.kt = fe_zero();
    st.store(p, 0, id);
    const gec pc = ge_to_cached(g);
    st.store(p, 1, pc);
    ge acc = ge_double_fast(g, true);
    st.store(p, 2, ge_to_cached(acc));
#pragma unroll 1
    for (int j = 3; j <= 8; ++j) {
      acc = ge_add_cached(acc, pc, false, true);
      st.store(p, j, ge_to_cached(acc));
    }
  }
  {
    // the digit words: W[w] collects nibble w of every point's recoded k / 2 mod r (registers: static indices); a dead point's
    // digits are 0, so it only ever meets its table's entry 0, the identity
    uint32_t W[BM_WINDOWS];
#pragma unroll
    for (int w = 0; w < BM_WINDOWS; ++w) W[w] = 0;
#pragma unroll 1
    for (int p = 0; p < m; ++p) {
      uint32_t k[8], dg[8];
      load_scalar(p, k);
      fr_reduce_words(k);
      fr_half_words(k);
      fr_recode_signed16(k, dg);
      const uint32_t live = ((deadmask >> p) & 1u) ? 0u : 15u;
#pragma unroll
      for (int wi = 0; wi < 8; ++wi)
#pragma unroll
        for (int b = 0; b < 8; ++b) W[8 * wi + b] |= ((dg[wi] >> (4 * b)) & live) << (4 * p);
    }
#pragma unroll
    for (int w = 0; w < BM_WINDOWS; ++w) st.dig_store(w, W[w]);
  }
  // the chain: (window, point) pairs from (63, 0) down to (0, m - 1); every addition requests the entry of the next pair
  ge r = ge_identity();
  uint32_t wn = st.dig_load(BM_WINDOWS - 1);
  int d = nibble_digit(wn, 0);
  bool neg = d < 0;
  gec e = st.load(0, neg ? -d : d, neg);
#pragma unroll 1
  for (int i = BM_WINDOWS - 1; i >= 0; --i) {
    const uint32_t wc = wn;
    wn = st.dig_load(i > 0 ? i - 1 : 0);
    if (i != BM_WINDOWS - 1) {
#pragma unroll 1
      for (int j = 0; j < 4; ++j) r = ge_double_neg(r, j == 3);   // (-2)^4 = 16; the additions read T
    }
#pragma unroll 1
    for (int p = 0; p < m; ++p) {
      const bool neg_cur = neg;
      const bool more = p + 1 < m;
      const int np = more ? p + 1 : 0;
      d = nibble_digit(more ? wc : wn, np);
      neg = d < 0;
      const int nj = neg ? -d : d;
      r = ge_add_cached_reload(r, e, neg_cur, more || (i == 0 && want_t), [&](gec& q) { q = st.load(np, nj, neg); });
    }
  }
  return r;
}

}  // namespace d377
