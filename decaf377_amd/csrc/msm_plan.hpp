// msm_plan.hpp -- the integer plan of the multi-scalar multiplication (msm.hip): how the 252 scalar bits are cut into
// windows, the signed digits of a scalar, and where the span sums leave a bucket's partial sums.  Plain integer code shared
// by the device kernels and the host simulation (tests/host_sim), which checks it against big-integer arithmetic.
#pragma once
#include <stdint.h>

#include "curve.hpp"

namespace d377 {

// The windows of the 252 scalar bits: W = ceil(252 / c) of them, the first `nwide` c bits wide and the rest c - 1, so that they
// tile the 252 bits exactly whatever c is (c = 16: twelve 16-bit and four 15-bit windows; 14 and 12 tile by themselves).  A
// uniform width with a ragged top window -- 12 significant bits at c = 16 -- would pile n / 2^10 points on each of a few
// hundred buckets; here the top window is at most one bit narrower than the others, and because k / 2 mod r < r < 2^250.23 its
// UNSIGNED digits (the top window is not wrapped) stay below 2^(width - 1.77) + 1: inside the 2^(width-1) buckets of its width.
struct WinShape {
  int c, W, nwide;
  D377_HD int width(int w) const { return w < nwide ? c : c - 1; }
  D377_HD int first_bit(int w) const { return w * c - (w > nwide ? w - nwide : 0); }
};
inline WinShape win_shape(int c) {
  const int W = (252 + c - 1) / c;
  return WinShape{c, W, W - (W * c - 252)};
}
// signed digit w of k (< 2^252): |digit| <= 2^(width - 1); the top window is not wrapped
D377_HD int msm_digit(const uint32_t k[8], int w, const WinShape& ws, uint32_t& carry) {
  const int bit = ws.first_bit(w), cw = ws.width(w);
  const int wi = bit >> 5, sh = bit & 31;
  uint64_t v = k[wi];
  if (wi + 1 < 8) v |= (uint64_t)k[wi + 1] << 32;
  uint32_t d = (uint32_t)((v >> sh) & ((1u << cw) - 1u)) + carry;
  carry = 0;
  if (w + 1 < ws.W && d >= (1u << (cw - 1))) { carry = 1; return (int)d - (1 << cw); }
  return (int)d;
}

// THE SPANS (msm.hip, k_msm_spans): lane k of a window takes the sorted entries [k L, (k + 1) L) and leaves one partial per
// bucket its span touches.  A bucket that holds the entries [o, o + size) of its window is touched by the lanes o / L ...
// (o + size - 1) / L:
D377_HD uint32_t span_first_lane(uint32_t o, uint32_t L) { return o / L; }
D377_HD uint32_t span_partials(uint32_t o, uint32_t size, uint32_t L) { return size != 0 ? (o + size - 1) / L - o / L + 1 : 0u; }

}  // namespace d377
