// msm.hip -- Element::vartime_multiscalar_mul as a real multi-scalar multiplication on gfx950.
//
// The reference (src/ark_curve/element/projective.rs:99-117) is a stub: a fold of
// `acc + scalar * point`.  Same result, different schedule: Pippenger's bucket method with
// signed c-bit windows, laid out for one-lane-per-bucket execution:
//
//   k_msm_prepare_* one lane per (point, scalar): decompress (Encodings), conversion (Elements with Z = 1) or batched-inversion
//                   normalisation (any Z) -> cached AFFINE record in HBM (128 bytes); scalar mod r -> W signed digits
//   k_msm_count     counting sort, pass 1: workgroup (window, slice) builds the histogram of |digit| over
//                   its slice of the points in LDS (the whole histogram of a window, <= 2^13 + 1 counters,
//                   fits) and writes it out
//   k_msm_scan1/2/3 per-bucket prefix over the slices, the exclusive prefix sums over the buckets of a window, and the plan of
//                   the bucket sums: entries per span lane, partials per bucket, groups of the reduction levels
//   k_msm_place1    pass 2, level 1: the same workgroup scatters each point's index (sign in bit 31) into the
//                   super-bucket (128 consecutive buckets) it belongs to; LDS atomics hand out the positions
//   k_msm_place2    level 2: workgroup (window, super-bucket) spreads its entries over the 128 bucket runs
//   k_msm_spans     one lane per span of L consecutive sorted entries, L = entries / resident lanes: mixed additions (7 M
//                   each), one partial per bucket the span touches -- one generation of lanes with equal work
//   k_msm_reduce    groups of 8, 8, 32 partial sums of a bucket: every level decides on the device whether it runs; a
//                   bucket that holds most of the points (many equal scalars) is cut down level by level
//   k_msm_buckets   one lane per bucket: sum of what is left of it (a few partials with random scalars)
//   k_msm_wsum_*    sum_b b * B_b per window by a pairwise tree of bit-sums, then Horner over the bits on four lanes
//                   (k_msm_chunks / k_msm_fold, the chunked running sums, remain for windows wider than 14 bits)
//   k_msm_final     Horner over the windows (c doublings per window) on four lanes per point (quad_ops.hpp), the encoding
//   k_msm_small(_sum)  batches of up to 4 x 16 x CUs points skip all of the above: one quad of lanes per point, see there
//
// Group-element outputs are canonical as encodings, so the result bytes equal the reference's
// whatever the summation order (the scatter order is non-deterministic; the sum is not).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/decaf377_amd.h"
#include "curve.hpp"
#include "device_util.hpp"
#include "dcb.hpp"
#include "msm_plan.hpp"
#include "quad_ops.hpp"
#include "row_ops.hpp"
#include "host_state.hpp"

using namespace d377;

namespace {

constexpr int PT_WORDS = 4 * SLOT;      // one cached or extended point record: 192 B
constexpr int CHUNK = 8;                // buckets per lane in k_msm_chunks (short chains: this phase is latency-bound)
constexpr int FOLD = 4;                 // points per lane in k_msm_fold (a serial chain per lane: short chains, more levels)
// Partial sums per lane in the further reduction levels (k_msm_reduce), level 2, 3, 4, and `skip`: a level runs only if some
// bucket still has more than this many partials; fewer are summed by the lane (or pair of lanes) that finishes the bucket.
// With the span sums a bucket of random scalars is left with 1 + size / L partials -- one to three -- so no level runs; the
// levels are for runs that hold most of the points (many equal scalars: one bucket with thousands of partials), which each
// level cuts by its group size.  (Rounds 3-4, one lane per <= seg points of a bucket: 4-30 partials per bucket, skip and
// the segment length swept at 2^16 ... 2^22, everything within 2 %.)
struct RedSizes { int g[3]; uint32_t skip; };
constexpr RedSizes RED_DEFAULT = {{8, 8, 32}, 32};

// Every input point is normalised to affine form once (Z = 1 already after decompression; one batched inversion
// per lane for Element inputs) and stored as a cached AFFINE record (device_util.hpp: pt_store_affine, 128 bytes,
// 128-byte aligned).  These records are gathered once per window in bucket order (n x W records: the MSM's dominant
// HBM traffic), so two 64-byte sectors instead of the three of a projective cached point, and a mixed addition
// (7 products) instead of 8.
__device__ __forceinline__ void pt_store_ext(uint32_t* p, const ge& g) {
  slot_store(p, g.x); slot_store(p + SLOT, g.y); slot_store(p + 2 * SLOT, g.z); slot_store(p + 3 * SLOT, g.t);
}
__device__ __forceinline__ ge pt_load_ext(const uint32_t* p) {
  ge g;
  g.x = slot_load(p); g.y = slot_load(p + SLOT); g.z = slot_load(p + 2 * SLOT); g.t = slot_load(p + 3 * SLOT);
  return g;
}

using row::RQ_WORDS;
using row::rq_store_point;
using row::rq_store_cached;
using row::rq_load_point;

// Digits travel as int16 up to 16-bit windows (|digit| <= 2^15: 2 bytes per point and window through the sort) and as int32 for
// the 17- and 18-bit windows of the largest batches (DT).
template <class DT>
__device__ __forceinline__ void msm_write_digits(const uint8_t* scalar32, size_t i, size_t n, const WinShape& ws, bool skip, DT* digits) {
  uint32_t k[8];
  load32(scalar32, i, k);
  fr_reduce_words(k);
  fr_half_words(k);                                     // the sum is formed with k/2 mod r and doubled at the end (k_msm_final)
  uint32_t carry = 0;
#pragma unroll 1
  for (int w = 0; w < ws.W; ++w) {
    int d = msm_digit(k, w, ws, carry);
    if (skip) d = 0;                                    // invalid points contribute nothing
    digits[(size_t)w * n + i] = (DT)d;
  }
}

// Encodings: one lane per point; decompression leaves Z = 1, so the affine record costs nothing extra.
template <class DT>
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)
k_msm_prepare_enc(SqrtTables T, const uint8_t* enc32, const uint8_t* scalar32, size_t n, WinShape ws,
                  uint32_t* pts, DT* digits, uint8_t* status) {
  D377_POW_LDS();
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t w[8];
    load32(enc32, i, w);
    ge g;
    const uint32_t bad = ge_decompress(T, pt, w, &g);
    status[i] = (uint8_t)bad;
    const fe x = fe_select(bad != 0, fe_zero(), g.x), y = fe_select(bad != 0, fe_const(FE_ONE), g.y);
    pt_store_affine(pts + i * AP_WORDS, gea_from_affine(x, y));
    msm_write_digits(scalar32, i, n, ws, bad != 0, digits);
  }
}

// The same in chunks with the square roots' denominators inverted together (dcb.hpp; as k_scalar_mul_var and
// k_decompress_chunked decode their points): 5-8 % fewer cycles per decompression; msm_launch uses it from DCB_ASSIST_MIN
// points per resident lane, like d377_batch_decompress (393 216 points on 256 CUs: the call -0.7 % there, -2 % at 2^20, -4 % at
// 2^22; profiles/r05_decompress_route_sweep.txt).
template <class DT>
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)
k_msm_prepare_enc_chunked(SqrtTables T, const uint8_t* enc32, const uint8_t* scalar32, size_t n, WinShape ws,
                          uint32_t* pts, DT* digits, uint8_t* status, DcbScratch dcb) {
  D377_POW_LDS();
  D377_DCB_BEGIN(status);
  dcb_rounds<1, false, false>(n, io, pt,
    [&](size_t i, int j) {
      uint32_t w[8];
      load32(enc32, i, w);
      dcb_put_den(io, 0, j, ge_decompress_den(w));
    },
    [&](size_t i, int, const uint32_t (*invw)[8], bool) {
      uint32_t w[8];
      load32(enc32, i, w);
      const fe inv = fe_from_words(invw[0]);
      ge g;
      const uint32_t bad = ge_decompress(T, pt, w, &g, &inv);
      status[i] = (uint8_t)bad;
      const fe x = fe_select(bad != 0, fe_zero(), g.x), y = fe_select(bad != 0, fe_const(FE_ONE), g.y);
      pt_store_affine(pts + i * AP_WORDS, gea_from_affine(x, y));
      msm_write_digits(scalar32, i, n, ws, bad != 0, digits);
    });
  D377_DCB_END();
}

// Elements.  Decompression output and affine inputs have Z = 1 (the record's words are those of 1 * 2^256 mod q); then
// the points are affine already and need no inversion.  k_msm_prepare_affine (one lane per point, the wide grid) does
// that work for every point whose Z is 1 and raises *flag when it meets any other Z; k_msm_prepare_el (Montgomery's
// trick, ~32 points per lane sharing one inversion, every point again) runs only when the flag is up and returns at
// once otherwise.  (Round 2 made the choice per wave inside the 32-per-lane kernel, which left the common Z = 1 case
// on a grid sized for sharing inversions it did not need: 0.48 ms per 2^22 points.)
// The records go through LDS both ways (device_util.hpp: wave_load_records128 / wave_store_records128): a lane that loads its own
// 128-byte Element and stores its own 128-byte affine record issues 6 + 8 instructions that each touch 64 lines and use an
// eighth of them; a wave moves its 64 records as eight coalesced 1 KiB instructions each way instead (HBM-priced kernel: 320
// bytes per point).
template <class DT>
__global__ void __launch_bounds__(BLOCK, 4)
k_msm_prepare_affine(SqrtTables T, const uint64_t* xyzt, const uint8_t* scalar32, size_t n, WinShape ws, uint32_t* pts, DT* digits,
                     uint32_t* flag) {
  (void)T;
  D377_RECORD_TILES(rec0) {
    // Once any wave has met a Z != 1, k_msm_prepare_el redoes the whole batch and nothing written here is used: the wave
    // that meets one raises the flag at once and leaves, and every other wave leaves when it sees the flag (a batch of
    // projective Elements -- sums, products -- spent 0.39 ms per 2^22 points in this kernel for nothing).
    if (*reinterpret_cast<const volatile uint32_t*>(flag) != 0) return;
    uint32_t a[32], w[32];
    wave_load_records128(xyzt, rec0, n, tile, lane, a);
    const size_t i = rec0 + (size_t)lane;
    const bool live = i < n;
    // The bucket route rebuilds 2dxy from X, Y, Z and never reads T; the small-batch route (k_msm_small) runs on the record's T.
    // A record with T Z != X Y would sum differently on the two: the check build counts such records (here, before the
    // early exits: this kernel sees every record of the batch unless a projective one sends it home) and k_msm_small's.
#if defined(D377_CHECK_INVARIANTS)
    {
      ge g;
      g.x = fe_from_mont256_words(a); g.y = fe_from_mont256_words(a + 8); g.z = fe_from_mont256_words(a + 16); g.t = fe_from_mont256_words(a + 24);
      D377_INVARIANT(T, g, live);
    }
#endif
    bool one = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) one &= a[16 + k] == ONE_MONT256_WORDS[k];
    if (__any(live && !one)) {
      if (lane == 0) atomicOr(flag, 1u);
      return;
    }
    const gea c = gea_from_affine(fe_from_mont256_words(a), fe_from_mont256_words(a + 8));
#pragma unroll
    for (int k = 0; k < NL; ++k) { w[k] = c.ypx.l[k]; w[NL + k] = c.ymx.l[k]; w[2 * NL + k] = c.kt.l[k]; }
#pragma unroll
    for (int k = 3 * NL; k < AP_WORDS; ++k) w[k] = 0;
    wave_store_records128(reinterpret_cast<uint64_t*>(pts), rec0, n, tile, lane, w);
    if (live) msm_write_digits(scalar32, i, n, ws, false, digits);
  }
}

// Elements with some Z != 1: Montgomery's trick per lane, as in k_to_affine -- forward pass multiplies the z's of the
// lane's grid-stride elements up, parking each prefix product in the element's own record slot; one inversion;
// the backward pass peels 1/z_i off, writes the affine record and the digits.  A record with z = 0 is no group
// element: it becomes the identity with digits 0.
template <class DT>
__global__ void __launch_bounds__(BLOCK, 4)
k_msm_prepare_el(const uint64_t* xyzt, const uint8_t* scalar32, size_t n, WinShape ws, uint32_t* pts, DT* digits,
                 const uint32_t* flag) {
  if (*flag == 0) return;
  const size_t Tn = (size_t)gridDim.x * BLOCK, t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool live = t < n;                               // (a lane without points still takes part in the wave's inversion)
  const uint8_t* b = reinterpret_cast<const uint8_t*>(xyzt);
  // The coordinates are used as they lie in memory (curve.hpp, "normalize_batch on raw records"): the words of z * 2^256
  // taken as limbs are z * 2^-5 in the internal radix, the powers of two the prefix products collect cancel in
  // 1 / z_k = inverse_k * prefix_(k-1) up to one 2^5, and x_raw * that is exactly the internal form of x / z -- no
  // conversion product for x, y or z (7 products per point instead of 11).
  fe p = fe_const(FE_ONE);
  size_t last = t;
  for (size_t i = t; i < n; i += Tn) {
    uint32_t w[8];
    bool zz;
    load32(b, 4 * i + 2, w);
    const fe z = affine_raw_z(w, &zz);
    slot_store(pts + i * AP_WORDS, p);
    p = fe_mul(p, z);
    last = i;
  }
  fe inv = row::fe_invert_lanes(p);                      // one inversion per wave (row_ops.hpp)
  if (!live) return;
  for (size_t i = last;; i -= Tn) {
    uint32_t w[8];
    bool zz;
    load32(b, 4 * i + 2, w);
    const fe z = affine_raw_z(w, &zz);
    const fe zi = fe_mul(inv, slot_load(pts + i * AP_WORDS));
    inv = fe_mul(inv, z);
    load32(b, 4 * i + 0, w);
    fe x = fe_mul(fe_from_words(w), zi);
    load32(b, 4 * i + 1, w);
    fe y = fe_mul(fe_from_words(w), zi);
    x = fe_select(zz, fe_zero(), x);
    y = fe_select(zz, fe_const(FE_ONE), y);
    pt_store_affine(pts + i * AP_WORDS, gea_from_affine(x, y));
    msm_write_digits(scalar32, i, n, ws, zz, digits);
    if (i == t) break;
  }
}

// Counting sort of the points of every window by |digit|.  The histogram of a whole window (nb <= 2^13 + 1
// counters for the widths pick_window chooses) lives in LDS, so neither pass issues a global atomic: round 1's
// histogram of returning global atomics (one per point and window, each its own L2 round trip) and the
// scattered rank reads cost 5.7 ms of a 12.6 ms MSM at 2^22.
constexpr int SORT_THREADS = 1024;
// slice s of window w covers points [s * per, min(n, (s + 1) * per)).  Windows wider than 16 bits have more buckets than LDS
// holds counters for (2^17 + 1 at 18 bits: 512 KiB): the bucket range is then cut into R parts of nbr buckets and workgroup
// (window, slice, part) counts the digits that fall into its part -- the digits are read R times (R = 2 at 17 bits, 4 at 18).
template <class DT>
__global__ void __launch_bounds__(SORT_THREADS) k_msm_count(const DT* digits, size_t n, int nb, int S, size_t per, int R, int nbr,
                                                            uint32_t* blockhist) {
  extern __shared__ uint32_t h[];
  constexpr int PER_VEC = 16 / (int)sizeof(DT);                  // digits per 16-byte load
  const int ws_ = blockIdx.x / R, part = blockIdx.x % R;
  const int w = ws_ / S, sl = ws_ % S;
  const int b_lo = part * nbr, b_hi = b_lo + nbr < nb ? b_lo + nbr : nb;
  for (int j = threadIdx.x; j < b_hi - b_lo; j += SORT_THREADS) h[j] = 0;
  __syncthreads();
  const size_t lo = (size_t)sl * per, hi = (lo + per < n) ? lo + per : n;
  const DT* dw = digits + (size_t)w * n;
  auto tally = [&](int d) {
    const int b = d < 0 ? -d : d;
    if (d != 0 && b >= b_lo && b < b_hi) atomicAdd(&h[b - b_lo], 1u);
  };
  // a 16-byte load of digits per lane where the row allows it (one 2-byte load per lane and trip left the kernel waiting on
  // memory latency: 1 TB/s); the unaligned head and the tail go one by one
  const size_t a0 = ((reinterpret_cast<uintptr_t>(dw + lo) + 15) & ~(uintptr_t)15) - reinterpret_cast<uintptr_t>(dw + lo);
  size_t head = lo + a0 / sizeof(DT);
  if (head > hi) head = hi;
  const size_t nvec = (hi - head) / PER_VEC;
  for (size_t i = lo + threadIdx.x; i < head; i += SORT_THREADS) tally((int)dw[i]);
  const uint4* dv = reinterpret_cast<const uint4*>(dw + head);
  for (size_t v = threadIdx.x; v < nvec; v += SORT_THREADS) {
    const uint4 q = dv[v];
    const uint32_t wq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (sizeof(DT) == 2) {
        tally((int)(int16_t)(wq[k] & 0xFFFFu));
        tally((int)(int16_t)(wq[k] >> 16));
      } else {
        tally((int)wq[k]);
      }
    }
  }
  for (size_t i = head + nvec * PER_VEC + threadIdx.x; i < hi; i += SORT_THREADS) tally((int)dw[i]);
  __syncthreads();
  uint32_t* out = blockhist + (size_t)ws_ * nb + b_lo;
  for (int j = threadIdx.x; j < b_hi - b_lo; j += SORT_THREADS) out[j] = h[j];
}

// Prefix sums of the sort and the plan of the bucket sums, three small kernels over workgroups (window, 1024 buckets).
// In: blockhist[w][s][b] = points of slice s in bucket b.  Out: blockhist[w][s][b] = points of the slices before s in
// bucket b; offs[w][0..nb] = exclusive prefix of the bucket sizes (offs[w][nb] = the window's entries); bsz[w][b] = the sizes.
//
// THE SPANS.  The sorted entries of a window lie in bucket order, and the lanes that sum them take SPANS of L consecutive
// entries whatever buckets those belong to (k_msm_spans): lane k of window w takes entries [k L, (k + 1) L) and leaves one
// partial sum per bucket its span touches.  Every lane has the same L additions to do -- no lane waits for a longer run,
// no half-full last segment per bucket, and with L = entries / resident lanes the whole sum is ONE generation of lanes that
// end together (one lane per <= seg points of ONE bucket, as before, ran in 5-10 generations of which the last was part
// empty, plus half a segment of idle additions per bucket: 11-18 % of the kernel at 2^20 and 2^22 points).  It also leaves
// few partials: a bucket touches 1 + size / L spans.  Bucket b of window w (entries [o, o + size)) has its partials in
// slots  lane0[w] + ne0[w] + ne[w][b] + (o / L) ...  + ((o + size - 1) / L)  where lane0 counts the lanes of the windows
// before, ne0 / ne[w][b] the non-empty buckets before b: slot = lane + rank of the bucket among the non-empty ones, a
// closed form (both grow by one along the entries), so the level needs no prefix array of its own.
// segoff[0][w][0..nb] = ne (exclusive prefix of "bucket is not empty" within the window); segoff[l][w][0..nb], l = 1..3 =
// exclusive prefix of ceil(partials / red.g[0]), ceil(that / red.g[1]), ...: the groups of the further reduction levels.
// k_msm_scan1: sizes, their prefix local to 1024 buckets, chunk totals; k_msm_scan2: the final offs, L from the total
// number of entries, the plan (partials per bucket, groups per level, the levels' maxima) with chunk-local prefixes;
// k_msm_scan3: adds the chunk carries and writes the per-window bases.
constexpr int REDUCE_LEVELS = 4;
// lvlmax (zeroed by the host before the launch) receives, per reduction level l and WINDOW w, the largest number of
// level-(l+1) partials any bucket of that window has -- lvlmax[l * LVL_STRIDE + w] -- and per level the largest over all
// windows, lvlmax[REDUCE_LEVELS * LVL_STRIDE + l].  A level returns at once when no window needs it, and leaves the
// windows alone whose buckets are down to `skip` partials: the lane that finishes the bucket adds those (k_msm_buckets).
constexpr int LVL_STRIDE = 64;
constexpr int LVL_WORDS = REDUCE_LEVELS * LVL_STRIDE + REDUCE_LEVELS;
constexpr int SEG_BLOCKS_PER_CU = 4;             // workgroups of k_msm_spans per CU (128 VGPRs: tests/test_codegen.py)
constexpr uint32_t SPAN_MIN = 8;                 // entries per lane at least (a lane's locate + store are worth ~1 addition)
// the plan's scalars: [0] L, [1] entries of all windows, [2] lanes of all windows
constexpr int META_WORDS = 4;
struct WinInfo { uint32_t len, lane0, ne0, lanes; };     // per window; entry [W] holds the totals in lane0 / ne0
__global__ void __launch_bounds__(1024) k_msm_scan1(uint32_t* blockhist, uint32_t* offs, uint32_t* bsz, uint32_t* tot, int nb, int S,
                                                    int nchunk) {
  __shared__ uint32_t part[1024];
  const int w = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk, t = threadIdx.x;
  const int len = nb + 1;
  const int j = chunk * 1024 + t;
  uint32_t* bh = blockhist + (size_t)w * S * nb;
  uint32_t c = 0;
  if (j < nb)
    for (int sl = 0; sl < S; ++sl) {                              // exclusive prefix over the slices, in place
      const uint32_t v = bh[(size_t)sl * nb + j];
      bh[(size_t)sl * nb + j] = c;
      c += v;
    }
  part[t] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t v = (t >= off) ? part[t - off] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  if (j < len) {                                                  // slot nb is the sentinel: total
    offs[(size_t)w * len + j] = part[t] - c;
    bsz[(size_t)w * len + j] = c;
  }
  if (t == 1023) tot[(size_t)w * nchunk + chunk] = part[1023];
}
__global__ void __launch_bounds__(1024) k_msm_scan2(uint32_t* offs, const uint32_t* bsz, uint32_t* segoff, const uint32_t* tot, uint32_t* tot2,
                                                    int nb, int W, int nchunk, uint32_t lanes_target, uint32_t forced_L, RedSizes red,
                                                    uint32_t* lvlmax, uint32_t* meta) {
  __shared__ uint32_t part[REDUCE_LEVELS][1024];
  __shared__ uint32_t bmax[REDUCE_LEVELS];
  __shared__ uint32_t s_E, s_carry;
  const int w = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk, t = threadIdx.x;
  const int len = nb + 1;
  const int j = chunk * 1024 + t;
  if (t < REDUCE_LEVELS) bmax[t] = 0;
  if (t == 0) { s_E = 0; s_carry = 0; }
  __syncthreads();
  {                                                               // W x nchunk <= 63 x 33 words: every workgroup sums them itself
    uint32_t e = 0, cy = 0;
    for (int k = t; k < W * nchunk; k += 1024) {
      const uint32_t v = tot[k];
      e += v;
      if (k / nchunk == w && k % nchunk < chunk) cy += v;
    }
    if (e) atomicAdd(&s_E, e);
    if (cy) atomicAdd(&s_carry, cy);
  }
  __syncthreads();
  const uint32_t E = s_E;
  // one generation: L = entries / lanes the kernel keeps resident (a window's last lane is part full: W lanes spare)
  uint32_t L = forced_L ? forced_L : (E + lanes_target - 1) / lanes_target;
  if (L < SPAN_MIN && !forced_L) L = SPAN_MIN;
  if (L < 1) L = 1;
  if (blockIdx.x == 0 && t == 0) { meta[0] = L; meta[1] = E; }
  uint32_t start = 0, size = 0;
  if (j < len) {
    start = offs[(size_t)w * len + j] + s_carry;
    size = bsz[(size_t)w * len + j];
    offs[(size_t)w * len + j] = start;
  }
  uint32_t own[1 + REDUCE_LEVELS];                               // [0] non-empty, [1] partials after the spans, [2..] groups per level
  own[0] = size != 0 ? 1u : 0u;
  own[1] = span_partials(start, size, L);
  for (int l = 2; l <= REDUCE_LEVELS; ++l) own[l] = (own[l - 1] + (uint32_t)red.g[l - 2] - 1) / (uint32_t)red.g[l - 2];
  part[0][t] = own[0];
  for (int l = 1; l < REDUCE_LEVELS; ++l) part[l][t] = own[l + 1];
  __syncthreads();
  for (int l = 1; l <= REDUCE_LEVELS; ++l)
    if (own[l] > 1) atomicMax(&bmax[l - 1], own[l]);
  __syncthreads();
  if (t < REDUCE_LEVELS && bmax[t] > 1) {
    atomicMax(&lvlmax[t * LVL_STRIDE + w], bmax[t]);
    atomicMax(&lvlmax[REDUCE_LEVELS * LVL_STRIDE + t], bmax[t]);
  }
  for (int off = 1; off < 1024; off <<= 1) {
    uint32_t v[REDUCE_LEVELS];
    for (int l = 0; l < REDUCE_LEVELS; ++l) v[l] = (t >= off) ? part[l][t - off] : 0u;
    __syncthreads();
    for (int l = 0; l < REDUCE_LEVELS; ++l) part[l][t] += v[l];
    __syncthreads();
  }
  if (j < len) {
    segoff[((size_t)0 * W + w) * len + j] = part[0][t] - own[0];
    for (int l = 1; l < REDUCE_LEVELS; ++l) segoff[((size_t)l * W + w) * len + j] = part[l][t] - own[l + 1];
  }
  if (t == 1023)
    for (int l = 0; l < REDUCE_LEVELS; ++l) tot2[((size_t)w * nchunk + chunk) * REDUCE_LEVELS + l] = part[l][1023];
}
__global__ void __launch_bounds__(1024) k_msm_scan3(const uint32_t* offs, uint32_t* segoff, const uint32_t* tot2, int nb, int W, int nchunk,
                                                    uint32_t* meta, WinInfo* winfo) {
  const int w = blockIdx.x / nchunk, chunk = blockIdx.x % nchunk, t = threadIdx.x;
  const int len = nb + 1;
  const int j = chunk * 1024 + t;
  if (blockIdx.x == 0) {                                          // the windows' bases: lanes and non-empty buckets before each
    __shared__ uint32_t s_len[64], s_lanes[64], s_ne[64];
    const uint32_t L = meta[0];
    if (t < W) {
      const uint32_t lw = offs[(size_t)t * len + nb];
      uint32_t ne = 0;
      for (int cch = 0; cch < nchunk; ++cch) ne += tot2[((size_t)t * nchunk + cch) * REDUCE_LEVELS + 0];
      s_len[t] = lw; s_lanes[t] = (lw + L - 1) / L; s_ne[t] = ne;
    }
    __syncthreads();
    if (t == 0) {
      uint32_t lane0 = 0, ne0 = 0;
      for (int k = 0; k < W; ++k) {
        winfo[k] = WinInfo{s_len[k], lane0, ne0, s_lanes[k]};
        lane0 += s_lanes[k]; ne0 += s_ne[k];
      }
      winfo[W] = WinInfo{0u, lane0, ne0, 0u};
      meta[2] = lane0;
    }
  }
  if (chunk == 0 || j >= len) return;
  uint32_t carry[REDUCE_LEVELS];
  for (int l = 0; l < REDUCE_LEVELS; ++l) carry[l] = 0;
  for (int k = 0; k < chunk; ++k)
    for (int l = 0; l < REDUCE_LEVELS; ++l) carry[l] += tot2[((size_t)w * nchunk + k) * REDUCE_LEVELS + l];
  for (int l = 0; l < REDUCE_LEVELS; ++l) segoff[((size_t)l * W + w) * len + j] += carry[l];
}

// Placement in two levels.  Scattering straight into the nb (8193 at c = 14) bucket runs of a window keeps
// W x S x nb partly written lines open at once -- far more than the L2s hold from 2^21 points up, so each 4-byte
// index left the chip as its own masked line write (1.7 ms of an 8.4 ms MSM at 2^22).  Level 1 scatters into
// super-buckets of SUPER consecutive buckets, level 2 takes one super-bucket per workgroup and spreads it over its
// SUPER bucket runs.  Either level handles its entries a tile at a time and sorts the tile by bin in LDS first, so
// that consecutive threads store to consecutive addresses (runs of ~TILE / bins entries): a wave store that touches
// 64 different lines costs the address coalescer 64 cycles, and three of those per entry were what was left of the
// sort (1.36 ms at 2^22) once the lines stayed in cache.
constexpr int SUPER_BITS = 7, SUPER = 1 << SUPER_BITS;
constexpr int MSM_MAX_WINDOW = 18;                                     // widest window: digits as int32 beyond 16 bits
constexpr int MAX_SUPER = ((1 << (MSM_MAX_WINDOW - 1)) + 1 + SUPER - 1) / SUPER;   // 1 025 super-buckets at 18 bits
constexpr int MAX_BINS = MAX_SUPER > SUPER ? MAX_SUPER : SUPER;
constexpr int TILE_PER_THREAD = 8, TILE = SORT_THREADS * TILE_PER_THREAD;
struct TileLds {
  uint32_t idx[TILE];                  // the tile's entries, sorted by bin
  uint16_t bin[TILE];
  uint8_t sub[TILE];
  uint32_t hist[MAX_BINS];             // entries of this tile per bin (zero between tiles)
  uint32_t start[MAX_BINS];            // first staged position of a bin
  uint32_t gbase[MAX_BINS];            // where this tile's part of a bin goes in the output
  uint32_t cur[MAX_BINS];              // running output cursor of a bin
  uint32_t total;
};
// load(i) -> the entry's raw word(s); decode(raw, i, &payload, &bin, &sub) -> false for an entry that is dropped (digit 0).
// The tile's TILE_PER_THREAD loads of a thread are issued together, before anything looks at their values: with the load
// inside the per-entry branch (as at first) each of the eight was its own dependent round trip to memory -- 15-24 us per
// 8 192-entry tile (level 2: 290 -> 228 us at 2^22 with the loads hoisted).  cur[] holds the output cursors.
// AHEAD (level 1: sixteen and more tiles per workgroup): the next tile's loads before this tile's stores, stores
// unconditional; without it (level 2: a super-bucket is two or three tiles) a tile fetches its own entries and only its
// real entries are stored -- there the clamped repeats of a part-filled last tile cost more than the order saves (measured:
// 155 -> 190 us at 2^22 with AHEAD, against 334 -> 250 us for level 1).
template <bool HAS_SUB, bool AHEAD, class Load, class Decode>
__device__ __forceinline__ void tile_scatter(TileLds& L, size_t lo, size_t hi, int nbins, Load load, Decode decode, uint32_t* out_idx, uint8_t* out_sub) {
  const int t = threadIdx.x;
  if (lo >= hi) return;
  // The NEXT tile's entries are requested before this tile's stores go out, and every access is unconditional (clamped to
  // the range): the kernels spend three quarters of their wave cycles waiting for memory (SQ_WAIT_ANY 0.76 / 0.65 of
  // SQ_WAVE_CYCLES, VALU busy 0.05: profiles/r05_stall_reasons_msm_2^22.txt), and a tile that starts by fetching its
  // entries first waits for the previous tile's scattered stores to drain -- the wait counter is in order -- and then for its
  // own loads.  With the loads ahead of the stores the counter lets the stores stay in flight.
  uint64_t raw[TILE_PER_THREAD];
  if (AHEAD) {
#pragma unroll
    for (int r = 0; r < TILE_PER_THREAD; ++r) {
      const size_t i = lo + (size_t)r * SORT_THREADS + t;
      raw[r] = load(i < hi ? i : hi - 1);
    }
  }
  for (size_t tile_lo = lo; tile_lo < hi; tile_lo += TILE) {
    uint32_t pay[TILE_PER_THREAD], rank[TILE_PER_THREAD];
    int bin[TILE_PER_THREAD];
    uint32_t sub[TILE_PER_THREAD];
    if (!AHEAD) {
      // the tile's loads of a thread are issued together, before anything looks at their values
#pragma unroll
      for (int r = 0; r < TILE_PER_THREAD; ++r) {
        const size_t i = tile_lo + (size_t)r * SORT_THREADS + t;
        raw[r] = i < hi ? load(i) : 0;
      }
    }
    // (Ranking the lanes of a wave that share a bin with one ballot per bin bit and a single LDS atomic per group was
    // built and measured: 318 -> 468 us and 228 -> 420 us for the two levels at 2^22 -- the returning atomics are not what
    // these kernels wait for.)
#pragma unroll
    for (int r = 0; r < TILE_PER_THREAD; ++r) {
      const size_t i = tile_lo + (size_t)r * SORT_THREADS + t;
      bin[r] = -1;
      if (i < hi && decode(raw[r], i, &pay[r], &bin[r], &sub[r])) rank[r] = atomicAdd(&L.hist[bin[r]], 1u); else bin[r] = -1;
    }
    __syncthreads();
    if (t < 64) {                                                   // one wave: exclusive prefix over the bins
      const int K = (nbins + 63) / 64, b0 = t * K;
      uint32_t own = 0;
      for (int k = 0; k < K; ++k) if (b0 + k < nbins) own += L.hist[b0 + k];
      uint32_t inc = own;
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(inc, off);
        if (t >= off) inc += v;
      }
      uint32_t run = inc - own;
      for (int k = 0; k < K; ++k) {
        const int b = b0 + k;
        if (b < nbins) {
          const uint32_t h = L.hist[b];
          L.start[b] = run; L.gbase[b] = L.cur[b]; L.cur[b] += h; L.hist[b] = 0;
          run += h;
        }
      }
      if (t == 63) L.total = inc;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TILE_PER_THREAD; ++r)
      if (bin[r] >= 0) {
        const uint32_t p = L.start[bin[r]] + rank[r];
        L.idx[p] = pay[r]; L.bin[p] = (uint16_t)bin[r];
        if (HAS_SUB) L.sub[p] = (uint8_t)sub[r];
      }
    __syncthreads();
    const uint32_t total = L.total;
    // the next tile's entries: in flight from here, across this tile's stores
    const size_t next_lo = tile_lo + TILE;
    if (AHEAD && next_lo < hi) {
#pragma unroll
      for (int r = 0; r < TILE_PER_THREAD; ++r) {
        const size_t i = next_lo + (size_t)r * SORT_THREADS + t;
        raw[r] = load(i < hi ? i : hi - 1);
      }
    }
    if (AHEAD) {
      if (total) {
#pragma unroll
        for (int r = 0; r < TILE_PER_THREAD; ++r) {
          uint32_t p = (uint32_t)r * SORT_THREADS + t;
          if (p >= total) p = total - 1;                            // (a repeat of the tile's last entry: the same word to the same place)
          const int b = L.bin[p];
          const uint32_t dst = L.gbase[b] + (p - L.start[b]);
          out_idx[dst] = L.idx[p];
          if (HAS_SUB) out_sub[dst] = L.sub[p];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < TILE_PER_THREAD; ++r) {
        const uint32_t p = (uint32_t)r * SORT_THREADS + t;
        if (p < total) {
          const int b = L.bin[p];
          const uint32_t dst = L.gbase[b] + (p - L.start[b]);
          out_idx[dst] = L.idx[p];
          if (HAS_SUB) out_sub[dst] = L.sub[p];
        }
      }
    }
    __syncthreads();
  }
}

// PACKED (batches of up to 2^24 points): the level-1 entry is ONE word -- sign, the 7 bits of the bucket within its
// super-bucket, 24 bits of point index -- instead of a word and a byte in two arrays (the byte stores came in runs of a few
// dozen bytes: 5 bytes written and 5 read per entry became 4 and 4).
constexpr size_t PACKED_MAX_POINTS = (size_t)1 << 24;
// (TileLds is 73 KiB with bins for 18-bit windows: beyond the 64 KiB of static LDS, so both levels take it as dynamic LDS)
template <bool PACKED, class DT>
__global__ void __launch_bounds__(SORT_THREADS) k_msm_place1(const DT* digits, size_t n, int nb, int S, size_t per,
                                                             const uint32_t* blockhist, const uint32_t* offs,
                                                             uint32_t* tmp_idx, uint8_t* tmp_sub) {
  extern __shared__ uint8_t tile_lds_[];
  TileLds& L = *reinterpret_cast<TileLds*>(tile_lds_);
  const int w = blockIdx.x / S, sl = blockIdx.x % S;
  const int nsuper = (nb + SUPER - 1) >> SUPER_BITS;
  const uint32_t* before_me = blockhist + (size_t)blockIdx.x * nb;      // points of earlier slices, per bucket
  const uint32_t* ow = offs + (size_t)w * (nb + 1);
  for (int j = threadIdx.x; j < nsuper; j += SORT_THREADS) { L.cur[j] = ow[j << SUPER_BITS]; L.hist[j] = 0; }
  __syncthreads();
  // points of earlier slices in the same super-bucket come first
  for (int j = threadIdx.x; j < nb; j += SORT_THREADS) {
    const uint32_t before = before_me[j];
    if (before) atomicAdd(&L.cur[j >> SUPER_BITS], before);
  }
  __syncthreads();
  const size_t lo = (size_t)sl * per, hi = (lo + per < n) ? lo + per : n;
  const DT* dw = digits + (size_t)w * n;
  tile_scatter<!PACKED, true>(L, lo, hi, nsuper,
                     [dw](size_t i) { return (uint64_t)(uint32_t)(int32_t)dw[i]; },
                     [](uint64_t raw, size_t i, uint32_t* pay, int* bin, uint32_t* sub) {
                       const int d = (int32_t)(uint32_t)raw;
                       if (d == 0) return false;
                       const int b = d < 0 ? -d : d;
                       *pay = (uint32_t)i | (d < 0 ? 0x80000000u : 0u);
                       if (PACKED) *pay |= (uint32_t)(b & (SUPER - 1)) << 24;
                       *bin = b >> SUPER_BITS;
                       *sub = (uint32_t)(b & (SUPER - 1));
                       return true;
                     },
                     tmp_idx + (size_t)w * n, tmp_sub + (size_t)w * n);
}

// workgroup (window, super-bucket): its entries are contiguous in tmp_*, at the positions the bucket runs will occupy
template <bool PACKED>
__global__ void __launch_bounds__(SORT_THREADS) k_msm_place2(const uint32_t* tmp_idx, const uint8_t* tmp_sub, size_t n, int nb,
                                                             const uint32_t* offs, uint32_t* idx) {
  extern __shared__ uint8_t tile_lds_[];
  TileLds& L = *reinterpret_cast<TileLds*>(tile_lds_);
  const int nsuper = (nb + SUPER - 1) >> SUPER_BITS;
  const int w = blockIdx.x / nsuper, B = blockIdx.x % nsuper;
  const uint32_t* ow = offs + (size_t)w * (nb + 1);
  const int first = B << SUPER_BITS, last = (first + SUPER < nb) ? first + SUPER : nb;
  if ((int)threadIdx.x < SUPER) { L.cur[threadIdx.x] = (int)threadIdx.x < last - first ? ow[first + threadIdx.x] : 0u; L.hist[threadIdx.x] = 0; }
  __syncthreads();
  const uint32_t* ti = tmp_idx + (size_t)w * n;
  const uint8_t* ts = tmp_sub + (size_t)w * n;
  tile_scatter<false, false>(L, ow[first], ow[last], SUPER,
                      [ti, ts](size_t i) { return PACKED ? (uint64_t)ti[i] : ((uint64_t)ti[i] | ((uint64_t)ts[i] << 32)); },
                      [](uint64_t raw, size_t, uint32_t* pay, int* bin, uint32_t* sub) {
                        if (PACKED) {
                          *pay = (uint32_t)raw & 0x80FFFFFFu;
                          *bin = (int)(((uint32_t)raw >> 24) & (SUPER - 1));
                        } else {
                          *pay = (uint32_t)raw;
                          *bin = (int)(raw >> 32);
                        }
                        *sub = 0;
                        return true;
                      },
                      idx + (size_t)w * n, (uint8_t*)nullptr);
}

// Lane gi of a reduction level -> (window, bucket, group within the bucket) by a short search in that level's
// prefix sums so[w][0..nb] (W <= 63 windows; groups of a window are contiguous).  False beyond the last group.
// wmax / skip (optional): the per-window maxima of the level's input; a lane of a window that needs no further level gets
// *w_out = -1 before it has searched anything.
__device__ __forceinline__ bool msm_locate(size_t gi, const uint32_t* so_all, int W, int nb, int* w_out, int* b_out, uint32_t* k_out,
                                           size_t* base_out, const uint32_t* wmax = nullptr, uint32_t skip = 0) {
  const int len = nb + 1;
  size_t base = 0;
  int w = 0;
  for (; w < W; ++w) {
    const uint32_t tot = so_all[(size_t)w * len + nb];
    if (gi < base + tot) break;
    base += tot;
  }
  if (w == W) return false;
  if (wmax && wmax[w] <= skip) { *w_out = -1; return true; }
  const uint32_t local = (uint32_t)(gi - base);
  const uint32_t* so = so_all + (size_t)w * len;
  int lo_b = 0, hi_b = nb;                             // largest b with so[b] <= local
  while (hi_b - lo_b > 1) {
    const int mid = (lo_b + hi_b) >> 1;
    if (so[mid] <= local) lo_b = mid; else hi_b = mid;
  }
  *w_out = w; *b_out = lo_b; *k_out = local - so[lo_b]; *base_out = base;
  return true;
}
__device__ __forceinline__ size_t msm_window_base(const uint32_t* so_all, int w, int nb) {   // groups of the windows before w
  size_t base = 0;
  for (int k = 0; k < w; ++k) base += so_all[(size_t)k * (nb + 1) + nb];
  return base;
}
// The span partials of bucket b of window w (the closed form above): first slot and count.
struct SpanPlan {
  const uint32_t* offs;          // [W][nb + 1]
  const uint32_t* ne;            // segoff level 0: non-empty buckets before b, within the window
  const WinInfo* winfo;
  uint32_t L;
};
__device__ __forceinline__ void span_partials_of(const SpanPlan& sp, int w, int b, int nb, size_t* first_slot, uint32_t* count) {
  const int len = nb + 1;
  const uint32_t o = sp.offs[(size_t)w * len + b], e = sp.offs[(size_t)w * len + b + 1];
  if (e == o) { *first_slot = 0; *count = 0; return; }
  const WinInfo wi = sp.winfo[w];
  *first_slot = (size_t)wi.lane0 + wi.ne0 + sp.ne[(size_t)w * len + b] + span_first_lane(o, sp.L);
  *count = span_partials(o, e - o, sp.L);
}

// One lane per span of L consecutive sorted entries of a window (see THE SPANS above): mixed additions (7 products each),
// the next record in flight while the current one is added, one partial sum stored per bucket the span touches.  A lane's
// first point is lifted from its record (4 products); after a bucket boundary inside the span the sum restarts from the
// identity with a full addition -- a branch there would have every wave that holds such a lane walk both paths.
__global__ void __launch_bounds__(BLOCK, SEG_BLOCKS_PER_CU)
k_msm_spans(const uint32_t* pts, const uint32_t* idx, SpanPlan sp, const uint32_t* meta, size_t n, int W, int nb, uint32_t* partial) {
  const int len = nb + 1;
  const uint32_t L = meta[0], lanes = meta[2];
  for (size_t gi = (size_t)blockIdx.x * BLOCK + threadIdx.x; gi < lanes; gi += (size_t)gridDim.x * BLOCK) {
    int w = 0;
    while (w + 1 < W && sp.winfo[w + 1].lane0 <= gi) ++w;
    const WinInfo wi = sp.winfo[w];
    const uint32_t pos = (uint32_t)(gi - wi.lane0) * L;
    uint32_t end = pos + L;
    if (end > wi.len) end = wi.len;
    if (pos >= end) continue;                            // (cannot happen: a window has ceil(len / L) lanes)
    const uint32_t* ow = sp.offs + (size_t)w * len;
    int lo_b = 0, hi_b = nb;                             // the bucket that holds entry pos: the largest b with ow[b] <= pos
    while (hi_b - lo_b > 1) {
      const int mid = (lo_b + hi_b) >> 1;
      if (ow[mid] <= pos) lo_b = mid; else hi_b = mid;
    }
    int b = lo_b;
    size_t slot = (size_t)gi + wi.ne0 + sp.ne[(size_t)w * len + b];
    uint32_t bend = ow[b + 1];
    const uint32_t* iw = idx + (size_t)w * n;
    // One record ahead, RAW: at the top of a step the buffer's words (entry j, requested a whole addition ago) are turned
    // into the cached point, the same registers are reloaded with entry j + 1 -- whose index was fetched an addition
    // earlier -- and the index of entry j + 2 is fetched; then the addition.  Nothing in flight is copied or looked at
    // before its turn.  (With the next record converted right behind its loads, as pt_load_affine does, the wave waited for
    // every gather before the addition that was to hide it: s_waitcnt vmcnt(5) and eighteen v_cndmask in front of the products.)
    uint32_t e_cur = iw[pos];
    uint32_t e_nxt = iw[pos + 1 < end ? pos + 1 : end - 1];
    gea_raw r = pt_load_affine_raw(pts + (size_t)(e_cur & 0x7FFFFFFFu) * AP_WORDS);
    uint32_t j = pos;                                    // entries [pos, j) are in acc (or flushed); r holds entry j
    // Issue priority by progress (dcb.hpp: dcb_progress_priority): the four waves of a SIMD have the same L additions to do and
    // the arbiter serves the oldest first, so they would end one after the other and the last run alone; a wave that is
    // behind outranks the ones ahead instead (3 for the first half of its span, 2 to 3/4, 1 to 31/32, then 0).
    int step = 0;                                        // additions done (the same in every lane of the wave)
    const int last = (int)(L / 32) > 0 ? (int)(L / 32) : 1;
    dcb_progress_priority_steps(0, (int)L, last);
    ge acc;
    {                                                    // the lane's first entry is lifted from its record (4 products)
      const bool neg = (e_cur >> 31) != 0;
      gea cur = gea_from_raw(r, neg);
      gea_pin(cur);
      e_cur = e_nxt;
      const uint32_t* rec = pts + (size_t)(e_nxt & 0x7FFFFFFFu) * AP_WORDS;
      // (unconditional, clamped to the span: loads under a branch leave the compiler's wait counters unknown at the join and
      // it waits for everything; the index first: the loop's back edge needs it, not the record behind it)
      e_nxt = iw[j + 2 < end ? j + 2 : end - 1];
      r = pt_load_affine_raw(rec);
      asm volatile("" ::: "memory");                     // ... and no load sinks below this line, to its first use after the addition
      acc = ge_from_cached_affine(cur, neg);
      ++j;
    }
#pragma unroll 1
    while (true) {
      if (j == bend || j == end) {                       // the bucket, or the span, ends here
        pt_store_ext(partial + slot * PT_WORDS, acc);
        if (j == end) break;
        ++b;
        while (ow[b + 1] <= j) ++b;                      // empty buckets in between (j < end <= ow[nb]: this stops)
        bend = ow[b + 1];
        ++slot;
        acc = ge_identity();
      }
      const bool neg = (e_cur >> 31) != 0;
      gea cur = gea_from_raw(r, neg);
      gea_pin(cur);
      e_cur = e_nxt;
      const uint32_t* rec = pts + (size_t)(e_nxt & 0x7FFFFFFFu) * AP_WORDS;
      // (unconditional, clamped to the span: loads under a branch leave the compiler's wait counters unknown at the join and
      // it waits for everything; the index first: the loop's back edge needs it, not the record behind it)
      e_nxt = iw[j + 2 < end ? j + 2 : end - 1];
      r = pt_load_affine_raw(rec);
      asm volatile("" ::: "memory");                     // ... and no load sinks below this line, to its first use after the addition
      acc = ge_add_affine(acc, cur, neg, true);
      ++j;
      dcb_progress_priority_steps(++step, (int)L, last);
    }
  }
}

// A further level of the same reduction: one lane per group of <= `red` partial sums of one bucket (so_out: prefix of the
// groups; the inputs of level 0 are the span partials, found by the closed form, those of later levels lie packed by
// so_in).  With random scalars a bucket has a handful of partials and no level runs; with many equal scalars (all
// coefficients 1, say) a run holds most of the n points, and every level cuts its partials by `red` instead of leaving them
// to one lane of k_msm_buckets (245 ms at 2^20 equal scalars).
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)
k_msm_reduce(const uint32_t* in, SpanPlan sp, const uint32_t* so_in, const uint32_t* so_out, int W, int nb, size_t max_groups, uint32_t* out,
             int red, const uint32_t* lvlmax, int level, uint32_t skip, const uint32_t* meta) {
  if (lvlmax[REDUCE_LEVELS * LVL_STRIDE + level] <= skip) return;   // every bucket is down to a few partials: k_msm_buckets adds those itself
  sp.L = meta[0];
  const int len = nb + 1;
  for (size_t gi = (size_t)blockIdx.x * BLOCK + threadIdx.x; gi < max_groups; gi += (size_t)gridDim.x * BLOCK) {
    int w, b;
    uint32_t k;
    size_t base;
    if (!msm_locate(gi, so_out, W, nb, &w, &b, &k, &base, lvlmax + level * LVL_STRIDE, skip)) break;
    if (w < 0) continue;                              // a window whose buckets need no further level
    size_t lo, hi;
    if (level == 0) {
      size_t first;
      uint32_t cnt;
      span_partials_of(sp, w, b, nb, &first, &cnt);
      lo = first + (size_t)k * (uint32_t)red; hi = lo + (uint32_t)red;
      if (hi > first + cnt) hi = first + cnt;
    } else {
      const size_t in_base = msm_window_base(so_in, w, nb);
      const uint32_t s0 = so_in[(size_t)w * len + b], s1 = so_in[(size_t)w * len + b + 1];
      lo = in_base + s0 + (size_t)k * (uint32_t)red; hi = lo + (uint32_t)red;
      if (hi > in_base + s1) hi = in_base + s1;
    }
    ge acc = ge_identity();
    if (lo < hi) {                                       // a group of one is copied, not added to the identity
      acc = pt_load_ext(in + lo * PT_WORDS);
      ge nx = acc;
      if (lo + 1 < hi) nx = pt_load_ext(in + (lo + 1) * PT_WORDS);
#pragma unroll 1
      for (size_t j = lo + 1; j < hi; ++j) {             // the next partial is in flight while this one is added
        const ge cur = nx;
        if (j + 1 < hi) nx = pt_load_ext(in + (j + 1) * PT_WORDS);
        acc = ge_add(acc, cur);
      }
    }
    pt_store_ext(out + gi * PT_WORDS, acc);
  }
}

// one lane per bucket: sum of what the last level that ran left of it (a few partials at most with random scalars).
// Level l + 1 ran for window w iff lvlmax[(l - 1) * LVL_STRIDE + w] > skip; the levels that ran for a window are a prefix.
struct MsmLevels {
  const uint32_t* buf[REDUCE_LEVELS];                     // partial sums after level 1 (the spans), 2, ...
};
// (With segments of 8-32 points a bucket was left with 4-30 partials and this kernel with a million additions -- a pair, and in
// one experiment a quad, of lanes per bucket shortened its chains; the spans leave 1 + size / L.)
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ ge ge_quad_perm(const ge& g) {
  ge r;
  r.x = fe_quad_perm<P0, P1, P2, P3>(g.x); r.y = fe_quad_perm<P0, P1, P2, P3>(g.y);
  r.z = fe_quad_perm<P0, P1, P2, P3>(g.z); r.t = fe_quad_perm<P0, P1, P2, P3>(g.t);
  return r;
}
// LANES = 2: a PAIR of lanes per bucket (lane q sums the partials q, q + 2, ...; one exchange step adds the two sums) -- for
// buckets that are left with many partials.  LANES = 1: one lane per bucket, no exchange -- when the spans leave a bucket one
// to three partials (L >= the run length: the usual case), where the pair's second lane and the exchange addition were more
// work than the sum itself (118 -> ~45 us at 2^22, 58 -> ~35 at 2^20).  msm_launch chooses by the expected partials per bucket.
template <int LANES>
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)
k_msm_buckets(MsmLevels lv, SpanPlan sp, const uint32_t* segoff_all, const uint32_t* lvlmax, uint32_t skip, int W, int nb, uint32_t* buckets,
              const uint32_t* meta) {
  static_assert(LANES == 1 || LANES == 2, "a bucket is summed by one lane or by a pair");
  constexpr int SHIFT = LANES == 2 ? 1 : 0;
  const int len = nb + 1;
  sp.L = meta[0];
  const size_t total = (size_t)W * nb;                      // buckets = groups of LANES lanes
  const int q = threadIdx.x & (LANES - 1);
  const size_t ngroups = ((size_t)gridDim.x * BLOCK) >> SHIFT;
  // every pair of a wave takes the same number of trips: the DPP exchange below needs both lanes active, so a pair past the
  // end redoes the last bucket and does not store
  const size_t trips = (total + ngroups - 1) / ngroups;
  size_t gq = ((size_t)blockIdx.x * BLOCK + threadIdx.x) >> SHIFT;
  for (size_t trip = 0; trip < trips; ++trip, gq += ngroups) {
    const size_t gi = gq < total ? gq : total - 1;
    const int w = (int)(gi / nb), b = (int)(gi % nb);
    int last = 0;
    while (last + 1 < REDUCE_LEVELS && lvlmax[last * LVL_STRIDE + w] > skip) ++last;
    const uint32_t* partial = lv.buf[last];
    size_t s0, s1;
    if (last == 0) {
      uint32_t cnt;
      span_partials_of(sp, w, b, nb, &s0, &cnt);
      s1 = s0 + cnt;
    } else {
      const uint32_t* segoff = segoff_all + (size_t)last * W * len;
      const size_t base = msm_window_base(segoff, w, nb);
      s0 = base + segoff[(size_t)w * len + b]; s1 = base + segoff[(size_t)w * len + b + 1];
    }
    s0 += (size_t)q;
    ge acc = ge_identity();
    if (s0 < s1) {
      acc = pt_load_ext(partial + s0 * PT_WORDS);
      ge nx = acc;
      if (s0 + LANES < s1) nx = pt_load_ext(partial + (s0 + LANES) * PT_WORDS);
#pragma unroll 1
      for (size_t j = s0 + LANES; j < s1; j += LANES) {   // the next partial is in flight while this one is added
        const ge cur = nx;
        if (j + LANES < s1) nx = pt_load_ext(partial + (j + LANES) * PT_WORDS);
        acc = ge_add(acc, cur);
      }
    }
    if (LANES == 2) acc = ge_add(acc, ge_quad_perm<1, 0, 3, 2>(acc));        // neighbours: sums 0 + 1
    if (q == 0 && gq < total) pt_store_ext(buckets + gi * PT_WORDS, acc);
  }
}

// lane (w, t): buckets lo..hi of window w (lo = 1 + t*CHUNK): sum_b b * B_b over the chunk
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)
k_msm_chunks(const uint32_t* buckets, int W, int nb, int nchunks, uint32_t* chunks) {
  const int total = W * nchunks;
  for (int gi = blockIdx.x * BLOCK + threadIdx.x; gi < total; gi += gridDim.x * BLOCK) {
    const int w = gi / nchunks, t = gi % nchunks;
    const int lo = 1 + t * CHUNK;
    int hi = lo + CHUNK - 1;
    if (hi > nb - 1) hi = nb - 1;
    ge run = ge_identity(), acc = ge_identity();
    if (hi >= lo) {
      run = pt_load_ext(buckets + ((size_t)w * nb + hi) * PT_WORDS);
      acc = run;
    }
#pragma unroll 1
    for (int b = hi - 1; b >= lo; --b) {
      run = ge_add(run, pt_load_ext(buckets + ((size_t)w * nb + b) * PT_WORDS));
      acc = ge_add(acc, run);                       // acc = sum (b - lo + 1) * B_b
    }
    // + (lo - 1) * run
    const uint32_t s = (uint32_t)(lo - 1);
    ge r = ge_identity();
#pragma unroll 1
    for (int bit = MSM_MAX_WINDOW - 1; bit >= 0; --bit) {   // s < 2^(c-1) + 1
      r = ge_double(r);
      if ((s >> bit) & 1u) r = ge_add(r, run);
    }
    pt_store_ext(chunks + (size_t)gi * PT_WORDS, ge_add(acc, r));
  }
}

// out[w][g] = sum of in[w][g*FOLD .. min((g+1)*FOLD, m))
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)
k_msm_fold(const uint32_t* in, int W, int m, int mout, uint32_t* out) {
  const int total = W * mout;
  for (int gi = blockIdx.x * BLOCK + threadIdx.x; gi < total; gi += gridDim.x * BLOCK) {
    const int w = gi / mout, g = gi % mout;
    int hi = (g + 1) * FOLD;
    if (hi > m) hi = m;
    ge acc = pt_load_ext(in + ((size_t)w * m + (size_t)g * FOLD) * PT_WORDS);
#pragma unroll 1
    for (int j = g * FOLD + 1; j < hi; ++j) acc = ge_add(acc, pt_load_ext(in + ((size_t)w * m + j) * PT_WORDS));
    pt_store_ext(out + (size_t)gi * PT_WORDS, acc);
  }
}

// ---- weighted bucket sums ------------------------------------------------------------------------
// S_w = sum_b b * B_b over the buckets of a window.  With U_j = the sum of the buckets whose index has bit j set,
// S_w = sum_j 2^j U_j, and the U_j come out of one pairwise tree: a node over 2^j consecutive buckets carries its total T
// and the bit-sums V_0 .. V_(j-1) of its own range; merging a left and a right node adds them position by position and
// appends V_j = T(right).  Level j has 2^(c-1-j) merges of j + 1 independent additions each -- 2 additions per bucket
// in all, like the running-sum trick, but c levels deep instead of a serial walk: the chunked running sums this
// replaces (8 buckets per lane, then a 16-bit double-and-add for the chunk's offset, then five 4-to-1 folds) were ~55
// dependent group operations, 245 us at every batch size.  k_msm_wsum_block takes 2^m buckets per workgroup through m
// levels in LDS (points as structure-of-arrays, 36 words each); k_msm_wsum_window merges the block nodes of a window
// (first level straight from global memory) and finishes with Horner over the bit-sums, in the lane-spread form.
// The tree's LEAVES are the buckets 1 .. 2^(c-1), numbered from 0 (bucket 0 is always empty: digit 0 places nothing), so
// the tree is c - 1 levels deep, a whole number of blocks, and S_w = sum_i (i + 1) L_i = sum_j 2^j V_j + T.  (Numbered
// by bucket index it was 2^(c-1) + 1 leaves: a level, a Horner step and a block per window for the one top bucket.)
constexpr int WS_M = 8;                          // 2^WS_M buckets per workgroup of k_msm_wsum_block / block2
constexpr int WS_M8 = 9;                         // 2^WS_M8 per workgroup of k_msm_wsum_block8 (eight buckets per lane)
constexpr int NODE_STRIDE = WS_M8 + 1;           // points per block node in global memory (a node of depth m uses m + 1)
constexpr int WS_THREADS = 1 << (WS_M - 2);      // one wave: a lane takes FOUR buckets through levels 0 and 1 in registers
constexpr int LP_WORDS = 4 * NL;
struct LdsPts {
  uint32_t* base;
  int cap;                                       // word k of point p lives at base[k * cap + p]
  __device__ __forceinline__ ge load(int p) const {
    ge g;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      g.x.l[i] = base[(0 * NL + i) * cap + p]; g.y.l[i] = base[(1 * NL + i) * cap + p];
      g.z.l[i] = base[(2 * NL + i) * cap + p]; g.t.l[i] = base[(3 * NL + i) * cap + p];
    }
    return g;
  }
  __device__ __forceinline__ void store(int p, const ge& g) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      base[(0 * NL + i) * cap + p] = g.x.l[i]; base[(1 * NL + i) * cap + p] = g.y.l[i];
      base[(2 * NL + i) * cap + p] = g.z.l[i]; base[(3 * NL + i) * cap + p] = g.t.l[i];
    }
  }
  // one coordinate (0 X, 1 Y, 2 Z, 3 T) of a point: what a lane of a quad holds (quad_ops.hpp)
  __device__ __forceinline__ fe load_coord(int p, int r) const {
    fe f;
#pragma unroll
    for (int i = 0; i < NL; ++i) f.l[i] = base[(r * NL + i) * cap + p];
    return f;
  }
  __device__ __forceinline__ void store_coord(int p, int r, const fe& f) {
#pragma unroll
    for (int i = 0; i < NL; ++i) base[(r * NL + i) * cap + p] = f.l[i];
  }
};
// one level of merges inside LDS: `cur` holds 2 * merges nodes of (j + 1) points, `nxt` receives merges nodes of (j + 2)
__device__ __forceinline__ void wsum_level(const LdsPts& cur, LdsPts& nxt, int j, int merges, int t, int nthreads) {
  const int per = j + 1;
  for (int l = t; l < merges * per; l += nthreads) {
    const int mu = l / per, tt = l - mu * per;
    const ge a = cur.load((2 * mu) * per + tt), b = cur.load((2 * mu + 1) * per + tt);
    nxt.store(mu * (per + 1) + tt, ge_add(a, b));
    if (tt == 0) nxt.store(mu * (per + 1) + per, b);          // V_j of the merged node = the right node's total
  }
}
// The same level with a QUAD of lanes per addition (quad_ops.hpp: lane r holds coordinate r; ~800 instructions per addition
// instead of ~1 900 on one lane) for the upper levels, which have fewer additions than the workgroup has quads to spare: a level
// of up to nthreads / 2 additions takes one or two rounds of quads, 1.5-3.0 us, instead of one round of lanes, 3.7 us.
// Every lane runs every trip (the DPP exchanges need whole quads); only the stores are predicated.
__device__ __forceinline__ void wsum_level_quads(const LdsPts& cur, LdsPts& nxt, int j, int merges, int t, int nthreads) {
  const int per = j + 1, role = t & 3, nq = nthreads >> 2, total = merges * per;
  for (int l0 = 0; l0 < total; l0 += nq) {
    const int l = l0 + (t >> 2);
    const bool live = l < total;
    const int ll = live ? l : 0;
    const int mu = ll / per, tt = ll - mu * per;
    const int pa = (2 * mu) * per + tt, pb = (2 * mu + 1) * per + tt;
    const fe a = cur.load_coord(pa, role), b = cur.load_coord(pb, role);
    const fe r = gq_add_with(a, gq_cached_slot(b, role), role, false);
    if (live) {
      nxt.store_coord(mu * (per + 1) + tt, role, r);
      if (tt == 0) nxt.store_coord(mu * (per + 1) + per, role, b);   // V_j of the merged node = the right node's total
    }
  }
}
__device__ __forceinline__ void wsum_level_any(const LdsPts& cur, LdsPts& nxt, int j, int merges, int t, int nthreads) {
  if (2 * merges * (j + 1) <= nthreads) wsum_level_quads(cur, nxt, j, merges, t, nthreads);
  else wsum_level(cur, nxt, j, merges, t, nthreads);
}
// One wave per workgroup and four buckets per lane: 18 windows x 33 blocks = 594 waves at c = 14, fewer than the chip has
// SIMDs, so every wave runs alone on its SIMD (with two buckets per lane there were 1170 waves on 1024 SIMDs and the
// kernel took as long as the SIMDs that got two: 100-130 us against 47 us for the same depth at c = 7).
constexpr int WSA_CAP0 = 3 << (WS_M - 2), WSA_CAP1 = 1 << (WS_M - 1);     // points after level 1 (192) and level 2 (128)
__global__ void __launch_bounds__(WS_THREADS)
k_msm_wsum_block(const uint32_t* buckets, int nb, int m, int nblk, uint32_t* nodes) {
  __shared__ uint32_t lds[(WSA_CAP0 + WSA_CAP1) * LP_WORDS];
  const int w = blockIdx.x / nblk, blk = blockIdx.x % nblk, t = threadIdx.x;
  LdsPts A{lds, WSA_CAP0}, B{lds + WSA_CAP0 * LP_WORDS, WSA_CAP1};
  const int M1 = 1 << (m - 2);                                 // nodes after level 1 (m >= 2)
  if (t < M1) {                                                // leaves 4t .. 4t + 3: T = B0 + B1 + B2 + B3, V_0 = B1 + B3, V_1 = B2 + B3
    const int b0 = (blk << m) + 4 * t + 1;                     // leaf i is bucket i + 1 (bucket 0 holds nothing: digit 0 is skipped)
    const uint32_t* src = buckets + ((size_t)w * nb + b0) * PT_WORDS;
    const ge p0 = b0 < nb ? pt_load_ext(src) : ge_identity();
    const ge p1 = b0 + 1 < nb ? pt_load_ext(src + PT_WORDS) : ge_identity();
    const ge p2 = b0 + 2 < nb ? pt_load_ext(src + 2 * PT_WORDS) : ge_identity();
    const ge p3 = b0 + 3 < nb ? pt_load_ext(src + 3 * PT_WORDS) : ge_identity();
    const ge r = ge_add(p2, p3);                               // total of the right pair
    A.store(3 * t, ge_add(ge_add(p0, p1), r));
    A.store(3 * t + 1, ge_add(p1, p3));
    A.store(3 * t + 2, r);
  }
  __syncthreads();
  LdsPts cur = A, nxt = B;
#pragma unroll 1
  for (int j = 2; j < m; ++j) {
    wsum_level_any(cur, nxt, j, M1 >> (j - 1), t, WS_THREADS);
    __syncthreads();
    const LdsPts tmp = cur; cur = nxt; nxt = tmp;
  }
  if (t <= m) pt_store_ext(nodes + (((size_t)w * nblk + blk) * NODE_STRIDE + t) * PT_WORDS, cur.load(t));
}

// The same with TWO waves per workgroup and two buckets per lane, for launches that leave every wave a SIMD of its own even so
// (2 x windows x blocks <= the chip's SIMDs: the 12-bit windows of batches below 2^20 points): levels 0 .. 2, which have
// 96-128 additions each, take one round of the lanes instead of two -- 8 dependent additions per lane instead of 11.
constexpr int WS2_THREADS = 1 << (WS_M - 1);
constexpr int WS2_CAP0 = 1 << WS_M, WS2_CAP1 = 3 << (WS_M - 2);            // points after level 0 (256) and level 1 (192)
static_assert((WS2_CAP0 + WS2_CAP1) * LP_WORDS * 4 <= 65536, "k_msm_wsum_block2's two point buffers fill the static LDS");
__global__ void __launch_bounds__(WS2_THREADS)
k_msm_wsum_block2(const uint32_t* buckets, int nb, int m, int nblk, uint32_t* nodes) {
  __shared__ uint32_t lds[(WS2_CAP0 + WS2_CAP1) * LP_WORDS];
  const int w = blockIdx.x / nblk, blk = blockIdx.x % nblk, t = threadIdx.x;
  LdsPts A{lds, WS2_CAP0}, B{lds + WS2_CAP0 * LP_WORDS, WS2_CAP1};
  const int M0 = 1 << (m - 1);                                 // nodes after level 0 (m >= 2)
  if (t < M0) {                                                // leaves 2t, 2t + 1: T = B0 + B1, V_0 = B1
    const int b0 = (blk << m) + 2 * t + 1;                     // leaf i is bucket i + 1
    const uint32_t* src = buckets + ((size_t)w * nb + b0) * PT_WORDS;
    const ge p0 = b0 < nb ? pt_load_ext(src) : ge_identity();
    const ge p1 = b0 + 1 < nb ? pt_load_ext(src + PT_WORDS) : ge_identity();
    A.store(2 * t, ge_add(p0, p1));
    A.store(2 * t + 1, p1);
  }
  __syncthreads();
  LdsPts cur = A, nxt = B;
#pragma unroll 1
  for (int j = 1; j < m; ++j) {
    wsum_level_any(cur, nxt, j, M0 >> j, t, WS2_THREADS);
    __syncthreads();
    const LdsPts tmp = cur; cur = nxt; nxt = tmp;
  }
  if (t <= m) pt_store_ext(nodes + (((size_t)w * nblk + blk) * NODE_STRIDE + t) * PT_WORDS, cur.load(t));
}

// EIGHT buckets per lane, 512 per one-wave workgroup, the levels in place: for the wide windows of large batches (c = 15, 16:
// 2^14 or 2^15 leaves per window).  k_msm_wsum_block keeps two point buffers, 46 KB of LDS -- three workgroups per CU -- and at
// c = 16 its 2 048 workgroups ran in 2.7 generations (190 us against 63 at c = 14).  Here a lane takes eight leaves through
// levels 0-2 in registers (11 additions, at most five points live) and leaves the node's four points in LDS; the levels above
// read their operands, wait for everybody, and write the merged nodes over them (a merged node is never longer than its two
// halves): ONE buffer of 256 points, 36 KB, four workgroups per CU -- one wave per SIMD, 1 024 workgroups at c = 16 in one
// generation.
constexpr int WS8_THREADS = 64, WS8_CAP = 4 * WS8_THREADS;
// level j in place: 2 * merges nodes of (j + 1) points -> merges nodes of (j + 2); at most two rounds of lanes
__device__ __forceinline__ void wsum_level_inplace(LdsPts& buf, int j, int merges, int t) {
  const int per = j + 1, total = merges * per;
  ge r0 = ge_identity(), r1 = ge_identity(), c0 = ge_identity(), c1 = ge_identity();
  const int l0 = t, l1 = t + WS8_THREADS;
  const int mu0 = l0 / per, tt0 = l0 - mu0 * per, mu1 = l1 / per, tt1 = l1 - mu1 * per;
  if (l0 < total) {
    const ge a = buf.load((2 * mu0) * per + tt0), b = buf.load((2 * mu0 + 1) * per + tt0);
    r0 = ge_add(a, b);
    c0 = b;
  }
  if (l1 < total) {
    const ge a = buf.load((2 * mu1) * per + tt1), b = buf.load((2 * mu1 + 1) * per + tt1);
    r1 = ge_add(a, b);
    c1 = b;
  }
  __syncthreads();                                             // every operand has been read
  if (l0 < total) {
    buf.store(mu0 * (per + 1) + tt0, r0);
    if (tt0 == 0) buf.store(mu0 * (per + 1) + per, c0);        // V_j of the merged node = the right node's total
  }
  if (l1 < total) {
    buf.store(mu1 * (per + 1) + tt1, r1);
    if (tt1 == 0) buf.store(mu1 * (per + 1) + per, c1);
  }
  __syncthreads();
}
__global__ void __launch_bounds__(WS8_THREADS)
k_msm_wsum_block8(const uint32_t* buckets, int nb, int m, int nblk, uint32_t* nodes) {
  __shared__ uint32_t lds[WS8_CAP * LP_WORDS];
  const int w = blockIdx.x / nblk, blk = blockIdx.x % nblk, t = threadIdx.x;
  LdsPts A{lds, WS8_CAP};
  {
    const int b0 = (blk << m) + 8 * t + 1;                     // leaf i is bucket i + 1 (m == WS_M8 here: 8 x 64 leaves)
    const uint32_t* src = buckets + ((size_t)w * nb + b0) * PT_WORDS;
    auto leaf = [&](int i) { return b0 + i < nb ? pt_load_ext(src + (size_t)i * PT_WORDS) : ge_identity(); };
    // T = sum of the eight, V_0 = odd leaves, V_1 = leaves 2 3 6 7, V_2 = leaves 4..7: 11 additions, leaves streamed
    ge s, a23;
    {
      const ge p1 = leaf(1), p3 = leaf(3);
      const ge o13 = ge_add(p1, p3);
      const ge a01 = ge_add(leaf(0), p1);
      a23 = ge_add(leaf(2), p3);
      s = ge_add(a01, a23);
      const ge p5 = leaf(5), p7 = leaf(7);
      A.store(4 * t + 1, ge_add(o13, ge_add(p5, p7)));
      const ge a45 = ge_add(leaf(4), p5), a67 = ge_add(leaf(6), p7);
      A.store(4 * t + 2, ge_add(a23, a67));
      const ge v2 = ge_add(a45, a67);
      A.store(4 * t + 3, v2);
      A.store(4 * t, ge_add(s, v2));
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int j = 3; j < m; ++j) wsum_level_inplace(A, j, WS8_THREADS >> (j - 2), t);
  if (t <= m) pt_store_ext(nodes + (((size_t)w * nblk + blk) * NODE_STRIDE + t) * PT_WORDS, A.load(t));
}

// The block nodes of one window -> S_w.  Levels m .. c-1 (none when one block covers the window), then Horner over the
// bit-sums with the cooperative doubling / addition of the tail below.
constexpr int WSB_THREADS = 512;
// The two point buffers of k_msm_wsum_window (dynamic LDS): the level that merges the block nodes leaves 2^(depth-m-1) nodes
// of m + 2 points, the next half as many of m + 3, and from there the levels shrink; the second buffer also carries the
// depth + 1 row records of the Horner chain.  Depth 13 (14-bit windows): 160 + 88 points, 35 KB; depth 15 (16-bit): 640 + 352, 140 KB.
inline int wsb_cap0(int depth, int m) { return depth > m ? (1 << (depth - m - 1)) * (m + 2) : m + 1; }
inline int wsb_cap1(int depth, int m) {
  int pts = depth > m + 1 ? (1 << (depth - m - 2)) * (m + 3) : 0;
  const int rec = ((depth + 1) * RQ_WORDS + LP_WORDS - 1) / LP_WORDS;       // the Horner chain's records, in points
  return pts > rec ? pts : rec;
}
// The levels between the block nodes and k_msm_wsum_window for trees deeper than 16 (windows of 17 and 18 bits: 2^16 or 2^17
// leaves, 128 or 256 block nodes per window -- k_msm_wsum_window's first merged level alone would be 200 KB of LDS): workgroup
// (window, group) merges 2^(m_out - m_in) consecutive block nodes of depth m_in into one node of depth m_out (m_out + 1
// points), first level from global memory, the rest in LDS.
constexpr int WSM_THREADS = 256;
constexpr int MID_STRIDE = 16;                   // points per node of the middle level (depth <= 15)
constexpr int WSM_CAP = 96;                      // points per LDS buffer: 8 nodes of 11 points after the first level of a 16-to-1 merge
__global__ void __launch_bounds__(WSM_THREADS)
k_msm_wsum_mid(const uint32_t* nodes_in, int m_in, int nblk_in, int m_out, int nblk_out, uint32_t* nodes_out) {
  __shared__ uint32_t lds[2 * WSM_CAP * LP_WORDS];
  const int w = blockIdx.x / nblk_out, grp = blockIdx.x % nblk_out, t = threadIdx.x;
  LdsPts X{lds, WSM_CAP}, Y{lds + WSM_CAP * LP_WORDS, WSM_CAP};
  const int fan = 1 << (m_out - m_in);                          // input nodes per output node (>= 2)
  const uint32_t* wn = nodes_in + ((size_t)w * nblk_in + (size_t)grp * fan) * NODE_STRIDE * PT_WORDS;
  {
    const int per = m_in + 1, merges = fan >> 1;                // level m_in, operands in global memory
    for (int l = t; l < merges * per; l += WSM_THREADS) {
      const int mu = l / per, tt = l - mu * per;
      const ge a = pt_load_ext(wn + ((size_t)(2 * mu) * NODE_STRIDE + tt) * PT_WORDS);
      const ge b = pt_load_ext(wn + ((size_t)(2 * mu + 1) * NODE_STRIDE + tt) * PT_WORDS);
      X.store(mu * (per + 1) + tt, ge_add(a, b));
      if (tt == 0) X.store(mu * (per + 1) + per, b);
    }
  }
  __syncthreads();
  LdsPts cur = X, nxt = Y;
#pragma unroll 1
  for (int j = m_in + 1; j < m_out; ++j) {
    wsum_level_any(cur, nxt, j, 1 << (m_out - j - 1), t, WSM_THREADS);
    __syncthreads();
    const LdsPts tmp = cur; cur = nxt; nxt = tmp;
  }
  if (t <= m_out) pt_store_ext(nodes_out + (((size_t)w * nblk_out + grp) * MID_STRIDE + t) * PT_WORDS, cur.load(t));
}

__global__ void __launch_bounds__(WSB_THREADS)
k_msm_wsum_window(const uint32_t* nodes, int c, int m, int nblk, int stride, int cap0, int cap1, uint32_t* sums) {
  // c here: the DEPTH of the window's tree, log2 of its leaves (the window width less one: see the leaves' numbering above)
  extern __shared__ uint32_t lds[];
  const int w = blockIdx.x, t = threadIdx.x;
  LdsPts X{lds, cap0}, Y{lds + (size_t)cap0 * LP_WORDS, cap1};
  const uint32_t* wn = nodes + (size_t)w * nblk * stride * PT_WORDS;
  LdsPts cur = X, nxt = Y;
  if (c == m) {                                                // the block node is the window's node
    if (t <= m) X.store(t, pt_load_ext(wn + (size_t)t * PT_WORDS));
  } else {
    const int per = m + 1, merges = 1 << (c - m - 1);         // level m, operands in global memory
    for (int l = t; l < merges * per; l += WSB_THREADS) {
      const int mu = l / per, tt = l - mu * per;
      const ge a = 2 * mu < nblk ? pt_load_ext(wn + ((size_t)(2 * mu) * stride + tt) * PT_WORDS) : ge_identity();
      const ge b = 2 * mu + 1 < nblk ? pt_load_ext(wn + ((size_t)(2 * mu + 1) * stride + tt) * PT_WORDS) : ge_identity();
      X.store(mu * (per + 1) + tt, ge_add(a, b));
      if (tt == 0) X.store(mu * (per + 1) + per, b);
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int j = m + 1; j < c; ++j) {
    wsum_level_any(cur, nxt, j, 1 << (c - j - 1), t, WSB_THREADS);
    __syncthreads();
    const LdsPts tmp = cur; cur = nxt; nxt = tmp;
  }
  // sum_i (i + 1) L_i = sum_j 2^j V_j + T.  Horner over the bit-sums: S = V_(c-1); S = 2 S + V_j; then + T -- one dependency
  // chain per window, so it runs in the lane-spread form (row_ops.hpp: the point across the four rows of ONE wave).  The
  // cached forms of the addends first, one lane each, as row records in the buffer that is free now; then wave 0 runs the chain.
  uint32_t* crec = nxt.base;                                   // c + 1 records of RQ_WORDS words (<= the smaller buffer)
  if (t < c - 1) rq_store_cached(crec + t * RQ_WORDS, cur.load(1 + t));   // point 1 + j of the node is V_j
  if (t == c - 1) rq_store_point(crec + (c - 1) * RQ_WORDS, cur.load(c)); // the top bit-sum: where the chain starts
  if (t == c) rq_store_cached(crec + c * RQ_WORDS, cur.load(0));          // point 0 is the total T
  __syncthreads();
  if (t >= 64) return;
  const row::RowK K = row::row_consts();
  const row::RowSel S = row::row_sel();
  uint32_t v = crec[(c - 1) * RQ_WORDS + t];
  bool negated = false;                                        // v holds -S after an odd number of sign-folded doublings
#pragma unroll 1
  for (int j = c - 2; j >= 0; --j) {
    v = row::rq_double_neg(v, S, K);
    negated = !negated;
    v = row::rq_add(v, crec + j * RQ_WORDS, S, negated, K);   // -2S - V_j, or 2S + V_j
  }
  v = row::rq_add(v, crec + c * RQ_WORDS, S, negated, K);     // -S - T, or S + T
  __shared__ uint32_t xrec[RQ_WORDS];
  xrec[t] = v;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // wave 0 alone is left (the others have returned): its own
  __builtin_amdgcn_wave_barrier();                             // LDS writes, in order, before its reads
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  ge r = rq_load_point(xrec);
  if (negated) r = ge_neg(r);
  if (t == 0) pt_store_ext(sums + (size_t)w * PT_WORDS, r);
}

// The square-root power table of a 64-thread workgroup, one LDS column per lane
struct Pow64 {
  uint32_t* col;
  __device__ __forceinline__ void put(int j, const fe& v) { for (int k = 0; k < NL; ++k) col[(j * NL + k) * 64] = v.l[k]; }
  __device__ __forceinline__ fe get(int j) const { fe r; for (int k = 0; k < NL; ++k) r.l[k] = col[(j * NL + k) * 64]; return r; }
};
// the square-root-free compressor's records for a single element
struct OneIO {
  uint32_t st[4][8], parked_[8], out[8];
  __device__ __forceinline__ void put(int s, int, const uint32_t* w) { for (int k = 0; k < 8; ++k) st[s][k] = w[k]; }
  __device__ __forceinline__ void get(int s, int, uint32_t* w) const { for (int k = 0; k < 8; ++k) w[k] = st[s][k]; }
  __device__ __forceinline__ void park(int, const uint32_t* w) { for (int k = 0; k < 8; ++k) parked_[k] = w[k]; }
  __device__ __forceinline__ void parked(int, uint32_t* w) const { for (int k = 0; k < 8; ++k) w[k] = parked_[k]; }
  __device__ __forceinline__ void emit(int, const uint32_t* w) { for (int k = 0; k < 8; ++k) out[k] = w[k]; }
};
// r = sum (k_i / 2) P_i: the result is its double, whose encoding needs no square root (curve.hpp, "compression
// without a square root") -- one element, so the inversion is not shared, but a divsteps inversion (~26 000
// instructions) is still well under the ~67 000 of a square root on this one dependent chain.  Every lane of the
// wave runs it, and the wave inverts together (row_ops.hpp fe_invert_wave: lane 0's value, 18 us instead of 31); `first`
// (lane 0) writes.
__device__ __forceinline__ void msm_emit_doubled(const ge& r, bool first, uint8_t* enc_out, uint64_t* xyzt_out) {
  OneIO io;
  dcb_put(io, 0, ge_dcb_from_half(r, false));
  dcb_finish_with(io, 1, [](const fe& c) { return row::fe_invert_wave(c); });
  if (first) {
    if (xyzt_out) store_ge_mont256(xyzt_out, 0, ge_double(r));
    store32(enc_out, 0, io.out);
  }
}

// Horner over the window sums S_w (lanes 0-3 of one wave), result as Element record and as encoding
__global__ void __launch_bounds__(64, 1)
k_msm_final(SqrtTables T, const uint32_t* sums, WinShape ws, uint8_t* enc_out, uint64_t* xyzt_out) {
  const int W = ws.W;
  if (blockIdx.x != 0) return;
  // cached forms of the window sums as row records, one lane each (W <= 63), then the chain in the lane-spread form
  // (row_ops.hpp): the running sum lies across the four rows of this wave, ~250 doublings at ~0.5 us instead of ~1 us on
  // four lanes
  __shared__ uint32_t crec[63 * RQ_WORDS];
  const int t = threadIdx.x;
  if (t < W - 1) rq_store_cached(crec + t * RQ_WORDS, pt_load_ext(sums + (size_t)t * PT_WORDS));
  if (t == W - 1) rq_store_point(crec + (W - 1) * RQ_WORDS, pt_load_ext(sums + (size_t)(W - 1) * PT_WORDS));
  __syncthreads();
  const row::RowK K = row::row_consts();
  const row::RowSel S = row::row_sel();
  uint32_t v = crec[(W - 1) * RQ_WORDS + t];
  bool negated = false;                                          // v holds minus the running sum
#pragma unroll 1
  for (int w = W - 2; w >= 0; --w) {
    const int c = ws.width(w);                                   // the running sum moves up by the width of the window it takes in
#pragma unroll 1
    for (int j = 0; j < c; ++j) v = row::rq_double_neg(v, S, K);
    if (c & 1) negated = !negated;                               // an odd number of sign-folded doublings
    v = row::rq_add(v, crec + w * RQ_WORDS, S, negated, K);
  }
  __syncthreads();
  crec[t] = v;                                                   // the first record is done with: back to whole elements
  __syncthreads();
  ge r = rq_load_point(crec);
  if (negated) r = ge_neg(r);
  (void)T;
  msm_emit_doubled(r, threadIdx.x == 0, enc_out, xyzt_out);
}

// ---- small batches: no buckets -------------------------------------------------------------------------------------
// (Rounds 3-4; since round 5 the bucket method's floor is 0.35 ms and msm_launch takes it from 3 073 / 4 097 points, see
// msm_small_max: this kernel is reached through the tuning override -- the route tests -- and stays as the cross-check.)
// Up to one quad of lanes per point and one wave per SIMD (n <= 4 x 16 x the CUs: 16384 on an MI355X) the bucket
// method had nothing to share: its kernels waited on their own dependency chains (the tree of bit-sums, the Horner chains
// of the windows, the 252 doublings of the tail: 0.55 ms whatever n) while most of the chip idled.  Here every quad
// computes [k_i / 2]P_i by itself -- the same 252 doublings, all points at once -- the 16 quads of a wave add their
// results up, and a second kernel sums the waves' partial results and encodes the double.  (Straus' interleaving would
// share the doublings between the points of a quad, which saves work and no time while every point has a quad.)
constexpr int MS_THREADS = 64, MS_QUADS = MS_THREADS / 4;      // k_msm_small: one wave per workgroup
constexpr int MSS_THREADS = 512, MSS_QUADS = MSS_THREADS / 4;  // k_msm_small_sum: one workgroup

__device__ __forceinline__ void gq_publish_slot(uint32_t* rec, const fe& v, int role) {
  const fe sl = gq_cached_slot(v, role);
#pragma unroll
  for (int i = 0; i < NL; ++i) rec[role * NL + i] = sl.l[i];
}

template <bool ENCODED>
__global__ void __launch_bounds__(MS_THREADS)
k_msm_small(SqrtTables T, const void* pts_in, const uint8_t* scalar32, size_t n, uint32_t* partial, uint8_t* status) {
  __shared__ uint32_t lds_pow_[ENCODED ? POW_TAB * NL * MS_THREADS : 1];
  __shared__ uint32_t tab[MS_QUADS * GQ_TAB_ENTRIES * GQ_WORDS];
  Pow64 pt;
  pt.col = lds_pow_ + threadIdx.x;
  const int role = threadIdx.x & 3, quad = threadIdx.x >> 2;
  const size_t e0 = (size_t)blockIdx.x * MS_QUADS + quad;
  const bool active = e0 < n;
  const size_t e = active ? e0 : n - 1;                          // idle quads redo the last point and contribute nothing
  uint32_t k[8], dg[8];
  load32(scalar32, e, k);
  fr_reduce_words(k);
  fr_half_words(k);                                              // the sum is formed with k/2 mod r and doubled at the end
  fr_recode_signed16(k, dg);
  ge g;
  bool skip = !active;
  if (ENCODED) {
    uint32_t w[8];
    load32(reinterpret_cast<const uint8_t*>(pts_in), e, w);
    const uint32_t bad = ge_decompress(T, pt, w, &g);            // every lane of the quad: the same chain of squarings
    if (active && role == 0) status[e] = (uint8_t)bad;
    skip |= bad != 0;                                            // invalid points contribute nothing
  } else {
    g = load_ge_mont256(reinterpret_cast<const uint64_t*>(pts_in), e);
    skip |= fe_is_zero(g.z);                                     // a record with z = 0 is no group element
    D377_INVARIANT(T, g, active && role == 0 && !skip);          // check build: T Z = X Y is what this route relies on
  }
  const fe id = gq_from_ge(ge_identity(), role);
  fe v = gq_scalar_mul_w4(gq_from_ge(g, role), dg, tab + quad * GQ_TAB_ENTRIES * GQ_WORDS, role);
  v = fe_select(skip, id, v);
  __syncthreads();                                               // the tables are done with: their area carries the exchange
  // sum over the 16 quads: after the step of width s, quad q holds the sum of quads q .. q + 2s - 1 (cyclically)
#pragma unroll 1
  for (int step = 1; step < MS_QUADS; step <<= 1) {
    gq_publish_slot(tab + quad * GQ_WORDS, v, role);
    __syncthreads();
    v = gq_add(v, tab + ((quad + step) & (MS_QUADS - 1)) * GQ_WORDS, role, false);
    __syncthreads();
  }
  if (quad == 0) slot_store(partial + (size_t)blockIdx.x * PT_WORDS + role * SLOT, v);   // X, Y, Z, T: pt_store_ext's layout
}

// ---- the smallest batches: a WAVE per point, or per few points ------------------------------------------------------------
// Up to one point per SIMD (4 x the CUs: 1 024 on an MI355X) even the quads leave most of the chip idle, and the call is one
// point's chain of 252 doublings.  In the lane-spread form (row_ops.hpp) that chain is half as long: every wave computes
// [k_i / 2]P_i with the point across its four rows, converts back and writes one partial; k_msm_small_sum adds them up.
// Up to MT_MAX points per SIMD a wave takes `m` points (wave b: points b, b + grid, ...) and shares the doublings between them
// (Straus): 252 doublings and m x 63 additions -- 0.12 + m x 0.04 ms against 0.35 ms for a quad per point.  The square roots of
// Encoding inputs run their power chains on the four ROWS of the wave at once, one point per row.
constexpr int MT_MAX = 4;
template <bool ENCODED>
__global__ void __launch_bounds__(64)
k_msm_tiny(SqrtTables T, const void* pts_in, const uint8_t* scalar32, size_t n, int m, uint32_t* partial, uint8_t* status) {
  __shared__ uint32_t tab[MT_MAX * row::RQ_TAB_ENTRIES * RQ_WORDS];
  __shared__ uint32_t xrec[2 * RQ_WORDS];
  __shared__ uint32_t sdg[MT_MAX][8];                                // the points' signed digits (wave-uniform reads in the loop)
  const int t = threadIdx.x;
  const row::RowK K = row::row_consts();
  const row::RowSel S = row::row_sel();
  // lane t looks after point (t & 3) of the wave's m: e_mine (clamped to a real point; `mine` says whether it is one)
  const int pj = t & 3;
  const size_t e_raw = (size_t)blockIdx.x + (size_t)pj * gridDim.x;
  const bool mine = pj < m && e_raw < n;
  const size_t e_mine = mine ? e_raw : (size_t)blockIdx.x;
  ge g;
  bool skip = !mine;
  if (ENCODED) {
    uint32_t w[8];
    load32(reinterpret_cast<const uint8_t*>(pts_in), e_mine, w);
    // the square roots' power chains (~300 products each) in the lane-spread form, row r = point r of the wave; their table
    // phases and the rest of the decompressions as whole-element code, lane t for point t & 3
    if (t < 4) row::row_store_from_fe(xrec + 16 * t, ge_decompress_den(w));
    __syncthreads();
    const row::RowPowers pw = row::row_sqrt_powers(xrec[t], tab, t, K);
    __syncthreads();
    xrec[t] = pw.v; xrec[RQ_WORDS + t] = pw.uv;
    __syncthreads();
    const fe pv = row::row_load_to_fe(xrec + 16 * pj), puv = row::row_load_to_fe(xrec + RQ_WORDS + 16 * pj);
    __syncthreads();
    const uint32_t bad = ge_decompress_from_powers(T, w, pv, puv, &g);
    if (t < 4 && mine) status[e_mine] = (uint8_t)bad;
    skip |= bad != 0;                                              // invalid points contribute nothing
  } else {
    g = load_ge_mont256(reinterpret_cast<const uint64_t*>(pts_in), e_mine);
    skip |= fe_is_zero(g.z);                                       // a record with z = 0 is no group element
    D377_INVARIANT(T, g, t < 4 && !skip);
  }
  // the points' digits (k / 2 mod r: the sum is doubled at the end) and tables, one point after the other
#pragma unroll 1
  for (int j = 0; j < m; ++j) {
    const size_t e = (size_t)blockIdx.x + (size_t)j * gridDim.x;
    const bool real = e < n;
    uint32_t k[8], dg[8];
    load32(scalar32, real ? e : (size_t)blockIdx.x, k);
    fr_reduce_words(k);
    fr_half_words(k);
    fr_recode_signed16(k, dg);
    // lanes j, j + 4, j + 8, j + 12 hold point j: each writes one coordinate
    if (pj == j && t < 16) row::row_store_from_fe(xrec + 16 * (t >> 2), fe_pick(t >> 2, g.x, g.y, g.z, g.t));
    __syncthreads();
    const bool dead = __shfl((int)skip, j) != 0 || !real;          // (wave-uniform: lane j's verdict on point j)
    if (t < 8) sdg[j][t] = dead ? 0x88888888u : dg[t];              // dead: every digit 0 (nibble 8 = value 0), the point contributes nothing
    row::rq_build_table(dead ? row::rq_identity(S) : xrec[t], tab + j * row::RQ_TAB_ENTRIES * RQ_WORDS, S, K);
    __syncthreads();
  }
  uint32_t v = row::rq_identity(S);
#pragma unroll 1
  for (int i = 63; i >= 0; --i) {
    if (i != 63) {
#pragma unroll 1
      for (int k = 0; k < 4; ++k) v = row::rq_double_neg(v, S, K); // four sign-folded doublings keep the sign
    }
#pragma unroll 1
    for (int j = 0; j < m; ++j) {
      const int d = fr_digit(sdg[j], i);
      if (d != 0) v = row::rq_add(v, tab + (j * row::RQ_TAB_ENTRIES + (d < 0 ? -d : d)) * RQ_WORDS, S, d < 0, K);
    }
  }
  __syncthreads();
  xrec[t] = v;
  __syncthreads();
  const ge r = rq_load_point(xrec);
  if (t == 0) pt_store_ext(partial + (size_t)blockIdx.x * PT_WORDS, r);
}

// the sum of the m partial results (one quad per MSS_QUADS of them, then a tree over the quads) and its encoding
__global__ void __launch_bounds__(MSS_THREADS)
k_msm_small_sum(SqrtTables T, const uint32_t* partial, int m, uint8_t* enc_out, uint64_t* xyzt_out) {
  __shared__ uint32_t xrec[MSS_QUADS * GQ_WORDS];
  const int role = threadIdx.x & 3, quad = threadIdx.x >> 2;
  const fe id = gq_from_ge(ge_identity(), role);
  fe v = quad < m ? slot_load(partial + (size_t)quad * PT_WORDS + role * SLOT) : id;
#pragma unroll 1
  for (int base = MSS_QUADS; base < m; base += MSS_QUADS) {
    const int j = base + quad;
    const fe u = j < m ? slot_load(partial + (size_t)j * PT_WORDS + role * SLOT) : id;
    gq_publish_slot(xrec + quad * GQ_WORDS, u, role);
    __syncthreads();
    v = gq_add(v, xrec + quad * GQ_WORDS, role, false);
    __syncthreads();
  }
  // quad 0 gathers: after the step of width s it holds the sum of quads 0 .. 2s - 1 (quads past m hold the identity;
  // what quads near the end pick up from the wrapped index never reaches quad 0)
  const int live = m < MSS_QUADS ? m : MSS_QUADS;
#pragma unroll 1
  for (int step = 1; step < live; step <<= 1) {
    gq_publish_slot(xrec + quad * GQ_WORDS, v, role);
    __syncthreads();
    v = gq_add(v, xrec + ((quad + step) & (MSS_QUADS - 1)) * GQ_WORDS, role, false);
    __syncthreads();
  }
  if (threadIdx.x >= 64) return;                                 // wave 0 encodes (no barrier from here on)
  (void)T;
  msm_emit_doubled(gq_to_ge(v), threadIdx.x == 0, enc_out, xyzt_out);       // lanes 0-3 hold the total
}

// sum of m Element records (partial results of several GPUs / ranks), one lane
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)
k_msm_combine(SqrtTables T, const uint64_t* xyzt, size_t m, uint8_t* enc_out, uint64_t* xyzt_out) {
  D377_POW_LDS();
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  ge r = ge_identity();
#pragma unroll 1
  for (size_t i = 0; i < m; ++i) r = ge_add(r, load_ge_mont256(xyzt, i));
  if (xyzt_out) store_ge_mont256(xyzt_out, 0, r);
  uint32_t w8[8];
  ge_compress(T, pt, r, w8);
  store32(enc_out, 0, w8);
}

// ------------------------------------------------------------------------------ host side ---
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Window width.  Only widths that tile the 252 scalar bits exactly are chosen (12, 14; the override also takes others):
// then the top window is as wide as the others and, because scalars are < r < 2^251, its unsigned
// digits spread over the same 2^(c-1) buckets as the signed digits of the other windows.  A ragged
// top window (e.g. c = 16: 11 significant bits) would pile n / 2^11 points on each of a few
// buckets and leave single lanes summing runs 16 times longer than everyone else's.
int pick_window(const DeviceState& d, size_t n) {
  // Measured on one MI355X (D377_TUNE_MSM_WINDOW sweep, Elements, whole call), 4 / 6 / 7 / 9 / 12 bits: 2^8 points 608 / 551 /
  // 556 / 543 / 544 us, 2^12: 657 / 607 / 599 / 578 / 577, 2^14: 739 / 683 / 671 / 622 / 587, 2^16: 1020 / 866 / 803 / 745 /
  // 686; 12 against 14 bits: 2^18 977 / 1007, 2^19 1310 / 1331, 2^20 2055 / 1985 us.  Since the bucket sums are a tree of
  // bit-sums (two kernels whose depth is the window width) a wide window costs little even when most of its buckets are
  // empty, and fewer windows are fewer additions per point and fewer chains side by side (the Horner doublings are 252
  // either way).  So: 12 bits below 2^20 points, 14 from there.
  // Round 5 (span sums, the tree to 16 bits, windows of mixed widths: no ragged top), 14 / 15 / 16 bits, one box: 2^20 1575 /
  // 1744 / 1734 us, 2^21 2844 / 2861 / 2959, 2^22 5585 / 5414 / 5299, 2^23 10713 / 10296 / 10013, 2^24 20966 / 20046 / 19203
  // (profiles/r05_msm_window_sweep.txt): 16 bits from 3 x 2^20 points.
  // At HEAD of round 5 (12 / 13 / 14 / 15 / 16 bits): 2^18 698 / 703 / 703 / 815 / 897 us, 2^19 1037 / 1029 / 1007 / 1070 / 1139,
  // 2^20 1685 / 1685 / 1522 / 1585 / 1638, 2^21 2981 / 2976 / 2720 / 2776 / 2784: 14 bits from 2^19 points.
  // Round 6: windows of 17 and 18 bits exist (int32 digits, the counting pass in parts, a middle level in the tree of bit-sums)
  // and do not pay up to 2^24 points -- 16 / 17 / 18 bits: 2^22 4828 / 4974 / 5641 us, 2^23 8953 / 9038 / 9559, 2^24 17310 /
  // 17275 / 17437 (profiles/r06_msm_window_sweep.txt): at 2^24 the span sums shrink by 1.5 ms from 16 to 18 bits and the
  // counting pass, the first placement level, the per-bucket sums and the tree's 3.5 times as many leaves take it back
  // (profiles/r06_msm_breakdown_16_17_18.txt).  They stay behind the override for batches beyond 2^24.
  const int c = n >= ((size_t)3 << 20) ? 16 : (n >= ((size_t)1 << 19) ? 14 : 12);
  return (int)d.tuned(D377_TUNE_MSM_WINDOW, c);              // developer override: 4 .. 18 (>= 4: at most 63 windows, k_msm_final's table of cached sums)
}

// Lanes of the span sums (k_msm_spans): as many as the device keeps resident at once -- what the runtime says the kernel's
// registers allow (4 workgroups per CU at 128 VGPRs: tests/test_codegen.py) -- less one per window (a window's last lane
// is part full).  k_msm_scan2 turns it into L = entries / lanes on the device, where the entries are known.
int grid_of(const DeviceState& d, size_t n) {
  size_t blocks = (n + BLOCK - 1) / BLOCK;
  size_t cap = (size_t)d.cus * 32;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// The workspace: grown when a call needs more (never inside a stream capture), handed from stream to stream by its guard.
// `held` releases it on every path out of the caller, error or not.
struct MsmHeld {
  ScratchGuard& g; hipStream_t s; bool held;
  int finish() { if (!held) return D377_OK; held = false; return g.release(s); }   // success path: a failed hand-over is an error
  ~MsmHeld() { if (held) (void)g.release(s); }
};
int msm_reserve(DeviceState& d, hipStream_t s, size_t bytes, MsmHeld& held) {
  int rc;
  if (bytes > d.msm.cap) {
    if (ScratchGuard::capturing(s))
      return fail(D377_ERR_ARG, "%s", "msm: the workspace must grow, which cannot happen inside a stream capture -- run one MSM of this size before capturing");
    if ((rc = d.msm.guard.drain())) return rc;          // a launch on another stream may still be using the old area
    if (d.msm.mem) {
      if (d.msm.guard.seen_capture) d.msm.retired.push_back(d.msm.mem);   // a captured graph may still point into it
      else HIP_TRY(hipFree(d.msm.mem));
    }
    d.msm.mem = nullptr; d.msm.cap = 0;
    HIP_TRY(hipMalloc(&d.msm.mem, bytes + bytes / 8));
    d.msm.cap = bytes + bytes / 8;
  }
  if ((rc = d.msm.guard.acquire(s))) return rc;         // one workspace per device: queue behind its last user
  held.held = true;
  return D377_OK;
}

// Batches up to this many points skip the buckets: a wave per 1 .. 3 points (Elements) or 1 .. 4 (Encodings), k_msm_tiny below.
// The bucket method's floor came down to 0.35 ms this round (span sums, the tree's block kernels), under the quads of
// k_msm_small, which took 4 097 .. 16 384 points in round 4 and are now reached by the tuning override only: Elements at
// 4 096 / 8 192 / 16 384 points 387 / 410 / 428 us on the small route against 353 / 366 / 378 with buckets, Encodings at
// 6 144 / 16 384 542 / 566 against 497 / 516; at 3 072 points the waves still win (346 / 358; Encodings at 4 096: 482 / 489)
// (profiles/r05_msm_small_route_sweep.txt).  D377_TUNE_MSM_SMALL_MAX: developer override (0 = never).
size_t msm_small_max(const DeviceState& d, bool encoded) {
  return (size_t)d.tuned(D377_TUNE_MSM_SMALL_MAX, (long long)d.cus * 4 * (encoded ? 4 : 3));
}

// Batches up to this many points take a wave per 1 .. MT_MAX points (k_msm_tiny): one wave per SIMD, up to 4 points each --
// measured against the quads (profiles/r04_size_sweep_msm.txt; Encodings at 3 072 / 4 096 points: 454 / 489 us against 534 / 537).
// D377_TUNE_MSM_TINY_MAX: developer override.
size_t msm_tiny_max(const DeviceState& d, bool encoded) {
  (void)encoded;
  return (size_t)d.tuned(D377_TUNE_MSM_TINY_MAX, (long long)d.cus * 4 * MT_MAX);
}

int msm_launch_small(DeviceState& d, hipStream_t s, bool encoded, const void* pts_in, const uint8_t* scalars, size_t n,
                     uint8_t* enc_out, uint64_t* xyzt_out, uint8_t* status) {
  const bool tiny = n <= msm_tiny_max(d, encoded);
  // points per wave on the wave route: as few as keep one wave per SIMD (more only when a developer override asks for it)
  size_t per_wave = (n + (size_t)d.cus * 4 - 1) / ((size_t)d.cus * 4);
  if (per_wave > MT_MAX) per_wave = MT_MAX;
  if (per_wave < 1) per_wave = 1;
  const size_t waves = (n + per_wave - 1) / per_wave;
  const size_t m = tiny ? waves : (n + MS_QUADS - 1) / MS_QUADS;
  MsmHeld held{d.msm.guard, s, false};
  int rc;
  if ((rc = msm_reserve(d, s, m * PT_WORDS * 4, held))) return rc;
  uint32_t* partial = (uint32_t*)d.msm.mem;
  const SqrtTables T = d.tables();
  if (tiny && encoded) hipLaunchKernelGGL(k_msm_tiny<true>, dim3((unsigned)waves), dim3(64), 0, s, T, pts_in, scalars, n, (int)per_wave, partial, status);
  else if (tiny) hipLaunchKernelGGL(k_msm_tiny<false>, dim3((unsigned)waves), dim3(64), 0, s, T, pts_in, scalars, n, (int)per_wave, partial, status);
  else if (encoded) hipLaunchKernelGGL(k_msm_small<true>, dim3((unsigned)m), dim3(MS_THREADS), 0, s, T, pts_in, scalars, n, partial, status);
  else hipLaunchKernelGGL(k_msm_small<false>, dim3((unsigned)m), dim3(MS_THREADS), 0, s, T, pts_in, scalars, n, partial, status);
  hipLaunchKernelGGL(k_msm_small_sum, dim3(1), dim3(MSS_THREADS), 0, s, T, partial, (int)m, enc_out, xyzt_out);
  HIP_TRY(hipGetLastError());
  return held.finish();
}

// everything on device pointers, enqueued on `s`
int msm_launch(DeviceState& d, hipStream_t s, bool encoded, const void* pts_in, const uint8_t* scalars, size_t n,
               uint8_t* enc_out, uint64_t* xyzt_out, uint8_t* status) {
  if (n >= ((size_t)1 << 31)) return fail(D377_ERR_ARG, "%s", "msm: n must be below 2^31");
  if (n && n <= msm_small_max(d, encoded) && n < ((size_t)1 << 24)) return msm_launch_small(d, s, encoded, pts_in, scalars, n, enc_out, xyzt_out, status);
  const int c = pick_window(d, n);
  const WinShape wshape = win_shape(c);                      // W windows of c or c - 1 bits that tile the 252 scalar bits
  const int W = wshape.W;
  const int nb = (1 << (c - 1)) + 1;                         // bucket indices 0 .. 2^(c-1)
  const int nchunks = (nb - 1 + CHUNK - 1) / CHUNK;
  // workspace carve-up
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  // slices per window of the counting sort: enough workgroups to cover the chip, never less than 8192 points each
  // (W * S <= the 2 workgroups of 1024 threads a CU holds: with one more slice per window, 18 x 29 = 522 workgroups on 512
  // places, the count and level-1 placement kernels ran a second generation for ten workgroups)
  int S = (int)d.tuned(D377_TUNE_MSM_SLICES, (long long)((size_t)2 * d.cus / (size_t)W));   // developer override (sweeps): 1 .. 4096
  if ((size_t)S > (n + 8191) / 8192) S = (int)((n + 8191) / 8192);
  if (S < 1) S = 1;
  const size_t per = (n + (size_t)S - 1) / (size_t)S;
  const size_t o_flag = carve(256);
  const size_t o_pts = carve(n * AP_WORDS * 4);
  const bool wide_digits = c > 16;                            // |digit| <= 2^(c-1): int16 up to 16-bit windows, int32 beyond
  const size_t o_dig = carve((size_t)W * n * (wide_digits ? 4 : 2));
  const size_t o_bh = carve((size_t)W * S * nb * 4);
  const size_t o_off = carve((size_t)W * (nb + 1) * 4);
  const size_t o_seg = carve((size_t)REDUCE_LEVELS * W * (nb + 1) * 4);
  const size_t o_bsz = carve((size_t)W * (nb + 1) * 4);
  const int scan_chunks = (nb + 1 + 1023) / 1024;
  const size_t o_tot = carve((size_t)W * scan_chunks * 4);
  const size_t o_tot2 = carve((size_t)W * scan_chunks * REDUCE_LEVELS * 4);
  const size_t o_meta = carve(META_WORDS * sizeof(uint32_t));
  const size_t o_winfo = carve((size_t)(W + 1) * sizeof(WinInfo));
  // the span sums: lanes resident at once (asked once per device), entries per lane when a developer forces them
  if (d.msm_span_blocks < 0) {
    int nblk = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, reinterpret_cast<const void*>(k_msm_spans), BLOCK, 0));
    d.msm_span_blocks = nblk < 1 ? 1 : (nblk > SEG_BLOCKS_PER_CU ? SEG_BLOCKS_PER_CU : nblk);
  }
  const size_t span_resident = (size_t)d.cus * d.msm_span_blocks * BLOCK;
  const uint32_t lanes_target = (uint32_t)(span_resident - (size_t)W);
  const uint32_t forced_L = (uint32_t)d.tuned(D377_TUNE_MSM_SEG, 0);             // developer override: entries per span lane
  // partial slots: one per lane and one per non-empty bucket (a lane stores one partial per bucket its span touches)
  const size_t span_lanes_max = forced_L ? ((size_t)n * W) / forced_L + (size_t)W + 1 : (((size_t)n * W) / SPAN_MIN + (size_t)W + 1 < span_resident
                                                                                          ? ((size_t)n * W) / SPAN_MIN + (size_t)W + 1 : span_resident);
  const size_t max_segs = span_lanes_max + (size_t)W * nb + 1;
  // the level-1 index array of the sort (W * n words) borrows the span partials' area, which is free until k_msm_spans
  const size_t par_bytes = max_segs * PT_WORDS * 4, tmp_bytes = (size_t)W * n * 4;
  const size_t o_par = carve(par_bytes > tmp_bytes ? par_bytes : tmp_bytes);
  const size_t o_idx = carve((size_t)W * n * 4);
  const size_t o_sub = carve((size_t)W * n);                            // level-1 placement: bucket index within the super-bucket
  const size_t o_bkt = carve((size_t)W * nb * PT_WORDS * 4);
  // further levels of the bucket reduction: groups of partials, then groups of those (never more than this many); every
  // level is launched and decides on the device whether it has anything to do (k_msm_scan1, lvlmax)
  RedSizes red = RED_DEFAULT;
  red.g[0] = (int)d.tuned(D377_TUNE_MSM_RED, red.g[0]);       // developer overrides (sweeps): 2 .. 64, 1 .. 64
  red.skip = (uint32_t)d.tuned(D377_TUNE_MSM_SKIP, red.skip);
  size_t max_g[REDUCE_LEVELS];
  size_t o_r[REDUCE_LEVELS];
  max_g[0] = max_segs; o_r[0] = o_par;
  for (int l = 1; l < REDUCE_LEVELS; ++l) {
    max_g[l] = max_g[l - 1] / (size_t)red.g[l - 1] + (size_t)W * nb;
    o_r[l] = carve(max_g[l] * PT_WORDS * 4);
  }
  const size_t o_lvl = carve(LVL_WORDS * sizeof(uint32_t));
  const size_t o_ch = carve((size_t)W * nchunks * PT_WORDS * 4);
  // ping-pong buffers of the 32-to-1 folds, sized from the fold sequence itself: the first fold writes
  // ceil(nchunks / FOLD) records per window into f0, the second ceil(that / FOLD) into f1, and so on
  const size_t m1 = (size_t)(nchunks + FOLD - 1) / FOLD, m2 = (m1 + FOLD - 1) / FOLD;
  const size_t o_f0 = carve((size_t)W * m1 * PT_WORDS * 4);
  const size_t o_f1 = carve((size_t)W * m2 * PT_WORDS * 4);
  // weighted bucket sums by the pairwise tree (every width: c <= 16); the chunked running sums remain as a developer
  // override (D377_TUNE_MSM_CHUNKED_SUMS), the tree's cross-check
  const bool tree = d.tuned(D377_TUNE_MSM_CHUNKED_SUMS, 0) == 0;
  // the tree's leaves are buckets 1 .. 2^(c-1) (bucket 0 is empty): depth c - 1, a whole number of blocks
  const int ws_depth = c - 1;
  // 512 leaves per workgroup (k_msm_wsum_block8) where 256 per workgroup would not fit the chip at once: three of those per CU
  const bool ws8 = ws_depth >= WS_M8 && (size_t)W * ((size_t)1 << (ws_depth - WS_M)) > (size_t)d.cus * 3;
  const int ws_m = ws8 ? WS_M8 : (ws_depth < WS_M ? ws_depth : WS_M);        // >= 2: window widths start at 4 here
  const int ws_nblk = 1 << (ws_depth - ws_m);
  const size_t o_nodes = carve(tree ? (size_t)W * ws_nblk * NODE_STRIDE * PT_WORDS * 4 : 0);
  // trees deeper than 15 (17- and 18-bit windows): a middle level merges the block nodes 8 or 16 to 1 (k_msm_wsum_mid), down
  // to 16 nodes per window for k_msm_wsum_window
  const bool ws_mid = tree && ws_depth > 15;
  const int ws_mid_m = ws_depth - 4, ws_mid_nblk = 16;
  const size_t o_mid = carve(ws_mid ? (size_t)W * ws_mid_nblk * MID_STRIDE * PT_WORDS * 4 : 0);
  const size_t o_sums = carve((size_t)W * PT_WORDS * 4);
  int rc;
  MsmHeld held{d.msm.guard, s, false};
  if ((rc = msm_reserve(d, s, off, held))) return rc;
  uint8_t* m = d.msm.mem;
  uint32_t* pts = (uint32_t*)(m + o_pts);
  void* dig_raw = m + o_dig;
  uint32_t *bh = (uint32_t*)(m + o_bh), *offs = (uint32_t*)(m + o_off);
  uint32_t *segoff = (uint32_t*)(m + o_seg), *partial = (uint32_t*)(m + o_par);
  uint32_t* idx = (uint32_t*)(m + o_idx);
  uint32_t* tmp_idx = partial;                                // free until k_msm_spans writes it
  uint8_t* tmp_sub = m + o_sub;
  uint32_t *bkt = (uint32_t*)(m + o_bkt), *ch = (uint32_t*)(m + o_ch), *f0 = (uint32_t*)(m + o_f0), *f1 = (uint32_t*)(m + o_f1);
  const SqrtTables T = d.tables();
  // the counting pass keeps a histogram of at most 2^15 + 1 buckets in LDS (128 KiB): wider windows are counted in R parts
  const int count_parts = (nb + (1 << 15)) / ((1 << 15) + 1);
  const int count_nbr = (nb + count_parts - 1) / count_parts;
  const size_t hist_bytes = (size_t)count_nbr * 4;
  // prepare (points -> affine records, scalars -> digits) and the counting pass, for the digit type of this window width
  auto front = [&](auto tag) -> int {
    using DT = decltype(tag);
    DT* dig = reinterpret_cast<DT*>(dig_raw);
    if (hist_bytes > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_count<DT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_bytes));
    if (n) {
      if (encoded) {
        const size_t chunked_min = (size_t)d.tuned(D377_TUNE_MSM_ENC_CHUNKED_MIN, (long long)(d.resident_lanes() * DCB_ASSIST_MIN));
        if (n >= chunked_min && d.msm_enc_chunked < 0) {
          // the chunked kernel claims lane sets of the scratch areas: only if its residency matches them (as d377_ctx_create
          // checks for the kernels of d377.hip); otherwise the wide kernel stays
          int nbk = 0, nbk32 = 0;
          HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, reinterpret_cast<const void*>(k_msm_prepare_enc_chunked<int16_t>), BLOCK, 0));
          HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk32, reinterpret_cast<const void*>(k_msm_prepare_enc_chunked<int32_t>), BLOCK, 0));
          d.msm_enc_chunked = (nbk >= 1 && nbk <= WAVES_PER_SIMD && nbk32 >= 1 && nbk32 <= WAVES_PER_SIMD) ? 1 : 0;
        }
        if (n >= chunked_min && d.msm_enc_chunked == 1) {
          const ChunkDeal cd = deal_chunks((n + BLOCK - 1) / BLOCK, (size_t)d.cus * WAVES_PER_SIMD, (size_t)DCB_K_LONG, (size_t)d.cus * 64);
          const size_t nchunks = cd.nchunks;
          DcbScratch dcb{d.dcb_scratch, d.slot_pool, d.cus * WAVES_PER_SIMD, (int)cd.per_lane, d.dcb_sets * BLOCK, (int)cd.extra, d.pool_health};
          dcb.prio = nchunks <= 2 * (size_t)d.cus * WAVES_PER_SIMD ? 1 : 0;      // as d377.hip's chunks_of: launches of one or two generations
          GuardScope vb{d.vb_guard, s};                       // the lane-set areas: queue behind their last user
          int r;
          if ((r = vb.acquire())) return r;
          hipLaunchKernelGGL(k_msm_prepare_enc_chunked<DT>, dim3((unsigned)nchunks), dim3(BLOCK), 0, s, T, (const uint8_t*)pts_in, scalars, n, wshape,
                             pts, dig, status, dcb);
          if ((r = vb.finish())) return r;
        } else {
          hipLaunchKernelGGL(k_msm_prepare_enc<DT>, dim3(grid_of(d, n)), dim3(BLOCK), 0, s, T, (const uint8_t*)pts_in, scalars, n, wshape, pts,
                             dig, status);
        }
      } else {
        uint32_t* zflag = (uint32_t*)(m + o_flag);
        HIP_TRY(hipMemsetAsync(zflag, 0, sizeof(uint32_t), s));
        hipLaunchKernelGGL(k_msm_prepare_affine<DT>, dim3(grid_of(d, n)), dim3(BLOCK), 0, s, T, (const uint64_t*)pts_in, scalars, n, wshape, pts,
                           dig, zflag);
        // ~32 elements per lane share one inversion, but never fewer lanes than one wave per SIMD (see k_to_affine)
        size_t lanes = (n + 31) / 32;
        const size_t fill = (size_t)d.cus * BLOCK;
        if (lanes < fill) lanes = fill < n ? fill : n;
        hipLaunchKernelGGL(k_msm_prepare_el<DT>, dim3((unsigned)((lanes + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, (const uint64_t*)pts_in,
                           scalars, n, wshape, pts, dig, zflag);
      }
    }
    hipLaunchKernelGGL(k_msm_count<DT>, dim3(W * S * count_parts), dim3(SORT_THREADS), hist_bytes, s, dig, n, nb, S, per, count_parts, count_nbr, bh);
    return D377_OK;
  };
  if ((rc = wide_digits ? front(int32_t{}) : front(int16_t{}))) return rc;
  uint32_t* tot = (uint32_t*)(m + o_tot);
  uint32_t* lvlmax = (uint32_t*)(m + o_lvl);
  HIP_TRY(hipMemsetAsync(lvlmax, 0, LVL_WORDS * sizeof(uint32_t), s));
  uint32_t *bsz = (uint32_t*)(m + o_bsz), *tot2 = (uint32_t*)(m + o_tot2), *meta = (uint32_t*)(m + o_meta);
  WinInfo* winfo = (WinInfo*)(m + o_winfo);
  hipLaunchKernelGGL(k_msm_scan1, dim3(W * scan_chunks), dim3(1024), 0, s, bh, offs, bsz, tot, nb, S, scan_chunks);
  hipLaunchKernelGGL(k_msm_scan2, dim3(W * scan_chunks), dim3(1024), 0, s, offs, bsz, segoff, tot, tot2, nb, W, scan_chunks, lanes_target,
                     forced_L, red, lvlmax, meta);
  hipLaunchKernelGGL(k_msm_scan3, dim3(W * scan_chunks), dim3(1024), 0, s, offs, segoff, tot2, nb, W, scan_chunks, meta, winfo);
  {
    // the two placement levels (TileLds: 73 KiB of dynamic LDS each)
    const bool packed = n <= PACKED_MAX_POINTS && d.tuned(D377_TUNE_MSM_SORT_PACKED, 1) != 0;
    const size_t tile_lds = sizeof(TileLds);
    auto place = [&](auto tag, auto packed_tag) -> int {
      using DT = decltype(tag);
      constexpr bool PK = decltype(packed_tag)::value;
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_place1<PK, DT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds));
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_place2<PK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds));
      hipLaunchKernelGGL((k_msm_place1<PK, DT>), dim3(W * S), dim3(SORT_THREADS), tile_lds, s, reinterpret_cast<const DT*>(dig_raw), n, nb, S, per, bh, offs,
                         tmp_idx, tmp_sub);
      hipLaunchKernelGGL(k_msm_place2<PK>, dim3(W * ((nb + SUPER - 1) / SUPER)), dim3(SORT_THREADS), tile_lds, s, tmp_idx, tmp_sub, n, nb, offs, idx);
      return D377_OK;
    };
    if (wide_digits) rc = packed ? place(int32_t{}, std::true_type{}) : place(int32_t{}, std::false_type{});
    else rc = packed ? place(int16_t{}, std::true_type{}) : place(int16_t{}, std::false_type{});
    if (rc) return rc;
  }
  const SpanPlan sp{offs, segoff, winfo, 0u};                  // (L travels in meta: the kernels read it there)
  hipLaunchKernelGGL(k_msm_spans, dim3(grid_of(d, span_lanes_max)), dim3(BLOCK), 0, s, pts, idx, sp, meta, n, W, nb, partial);
  const size_t so_stride = (size_t)W * (nb + 1);
  MsmLevels lv;
  lv.buf[0] = partial;
  for (int l = 1; l < REDUCE_LEVELS; ++l) {
    uint32_t* r = (uint32_t*)(m + o_r[l]);
    // (the levels only ever run for runs that hold most of the points: a small grid that strides, so that the launch that
    // finds nothing to do costs a few hundred workgroups, not thousands)
    int gr = grid_of(d, max_g[l]);
    if (gr > d.cus * 4) gr = d.cus * 4;
    hipLaunchKernelGGL(k_msm_reduce, dim3(gr), dim3(BLOCK), 0, s, lv.buf[l - 1], sp, segoff + (size_t)(l - 1) * so_stride,
                       segoff + (size_t)l * so_stride, W, nb, max_g[l], r, red.g[l - 1], lvlmax, l - 1, red.skip, meta);
    lv.buf[l] = r;
  }
  {
    // partials a bucket is left with when the scalars are random: 1 + its run / the entries per span lane
    const double run = (double)n / (double)(nb - 1);
    double Lest = forced_L ? (double)forced_L : (double)n * W / (double)lanes_target;
    if (!forced_L && Lest < (double)SPAN_MIN) Lest = (double)SPAN_MIN;
    if (1.0 + run / Lest <= 4.0)
      hipLaunchKernelGGL(k_msm_buckets<1>, dim3(grid_of(d, (size_t)W * nb)), dim3(BLOCK), 0, s, lv, sp, segoff, lvlmax, red.skip, W, nb, bkt, meta);
    else
      hipLaunchKernelGGL(k_msm_buckets<2>, dim3(grid_of(d, (size_t)W * nb * 2)), dim3(BLOCK), 0, s, lv, sp, segoff, lvlmax, red.skip, W, nb, bkt, meta);
  }
  const uint32_t* cur_in;
  if (tree) {
    uint32_t *nodes = (uint32_t*)(m + o_nodes), *sums = (uint32_t*)(m + o_sums);
    if (ws8)
      hipLaunchKernelGGL(k_msm_wsum_block8, dim3(W * ws_nblk), dim3(WS8_THREADS), 0, s, bkt, nb, ws_m, ws_nblk, nodes);
    else if ((size_t)2 * W * ws_nblk <= (size_t)d.cus * 4)     // two waves per block still leave every wave its own SIMD
      hipLaunchKernelGGL(k_msm_wsum_block2, dim3(W * ws_nblk), dim3(WS2_THREADS), 0, s, bkt, nb, ws_m, ws_nblk, nodes);
    else
      hipLaunchKernelGGL(k_msm_wsum_block, dim3(W * ws_nblk), dim3(WS_THREADS), 0, s, bkt, nb, ws_m, ws_nblk, nodes);
    const uint32_t* top_nodes = nodes;
    int top_m = ws_m, top_nblk = ws_nblk, top_stride = NODE_STRIDE;
    if (ws_mid) {
      uint32_t* mid = (uint32_t*)(m + o_mid);
      hipLaunchKernelGGL(k_msm_wsum_mid, dim3(W * ws_mid_nblk), dim3(WSM_THREADS), 0, s, nodes, ws_m, ws_nblk, ws_mid_m, ws_mid_nblk, mid);
      top_nodes = mid; top_m = ws_mid_m; top_nblk = ws_mid_nblk; top_stride = MID_STRIDE;
    }
    const int cap0 = wsb_cap0(ws_depth, top_m), cap1 = wsb_cap1(ws_depth, top_m);
    const size_t wsb_lds = (size_t)(cap0 + cap1) * LP_WORDS * sizeof(uint32_t);
    if (wsb_lds > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_wsum_window), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wsb_lds));
    hipLaunchKernelGGL(k_msm_wsum_window, dim3(W), dim3(WSB_THREADS), wsb_lds, s, top_nodes, ws_depth, top_m, top_nblk, top_stride, cap0, cap1, sums);
    cur_in = sums;
  } else {
    hipLaunchKernelGGL(k_msm_chunks, dim3(grid_of(d, (size_t)W * nchunks)), dim3(BLOCK), 0, s, bkt, W, nb, nchunks, ch);
    // fold chunk results down to one point per window
    cur_in = ch;
    int mcur = nchunks;
    uint32_t* bufs[2] = {f0, f1};
    int which = 0;
    const size_t fold_cap[2] = {m1, m2};
    while (mcur > 1) {
      const int mout = (mcur + FOLD - 1) / FOLD;
      if ((size_t)mout > fold_cap[which]) return fail(D377_ERR_ARG, "%s", "msm: fold buffer too small (internal)");
      uint32_t* o = bufs[which];
      hipLaunchKernelGGL(k_msm_fold, dim3(grid_of(d, (size_t)W * mout)), dim3(BLOCK), 0, s, cur_in, W, mcur, mout, o);
      cur_in = o; mcur = mout; which ^= 1;
    }
  }
  hipLaunchKernelGGL(k_msm_final, dim3(1), dim3(64), 0, s, T, cur_in, wshape, enc_out, xyzt_out);
  HIP_TRY(hipGetLastError());
  return held.finish();
}

// one device's share of a host batch: copies in, MSM, partial sum (Element record) and statuses out, synchronised
int msm_one(DeviceState& d, bool encoded, const uint8_t* pts_in, const uint8_t* scalars, size_t cnt, uint8_t* enc_out,
            uint64_t* partial_out, uint8_t* status) {
  HIP_TRY(hipSetDevice(d.id));
  int rc = D377_OK;
  SyncOnError guard{&rc, d.id, d.stream, nullptr};
  auto body = [&]() -> int {
    const size_t rec = encoded ? 32 : 128;
    int r;
    if ((r = ensure(d, 0, cnt * rec + 16))) return r;
    if ((r = ensure(d, 1, cnt * 32 + 16))) return r;
    if ((r = ensure(d, 2, 32 + 128))) return r;
    if ((r = ensure(d, 3, cnt + 16))) return r;
    StarveCheck starve{d, d.stream};
    if ((r = starve.before())) return r;
    if (cnt) {
      HIP_TRY(hipMemcpyAsync(d.buf[0], pts_in, cnt * rec, hipMemcpyHostToDevice, d.stream));
      HIP_TRY(hipMemcpyAsync(d.buf[1], scalars, cnt * 32, hipMemcpyHostToDevice, d.stream));
    }
    if ((r = msm_launch(d, d.stream, encoded, d.buf[0], d.buf[1], cnt, d.buf[2], (uint64_t*)(d.buf[2] + 32), d.buf[3]))) return r;
    HIP_TRY(hipMemcpyAsync(partial_out, d.buf[2] + 32, 128, hipMemcpyDeviceToHost, d.stream));
    if (encoded && cnt) HIP_TRY(hipMemcpyAsync(status, d.buf[3], cnt, hipMemcpyDeviceToHost, d.stream));
    if (enc_out) HIP_TRY(hipMemcpyAsync(enc_out, d.buf[2], 32, hipMemcpyDeviceToHost, d.stream));
    if ((r = starve.after())) return r;
    HIP_TRY(hipStreamSynchronize(d.stream));
    return starve.verdict();
  };
  rc = body();
  return rc;
}

int msm_host(d377_ctx* ctx, bool encoded, const void* pts_in, const uint8_t* scalars, size_t n, uint8_t* enc_out,
             uint64_t* xyzt_out, uint8_t* status) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  if (!enc_out || (n && (!pts_in || !scalars)) || (encoded && n && !status)) return fail(D377_ERR_ARG, "%s", "null buffer");
  std::lock_guard<std::mutex> lock(ctx->mu);
  const size_t nd = ctx->devs.size();
  const size_t rec = encoded ? 32 : 128;
  std::vector<uint64_t> partial(nd * 16, 0);
  int rc;
  if (nd == 1) {
    if ((rc = msm_one(ctx->devs[0], encoded, (const uint8_t*)pts_in, scalars, n, enc_out, partial.data(), status))) return rc;
    if (xyzt_out) memcpy(xyzt_out, partial.data(), 128);
    return D377_OK;
  }
  // several devices: one host thread each (pageable copies block their issuing thread), joined before
  // anything else happens; then one small cross-device reduction of the partial sums on device 0
  const size_t per = (n + nd - 1) / nd;
  std::vector<int> rcs(nd, D377_OK);
  std::vector<std::string> errs(nd);
  std::vector<std::thread> workers;
  const int delay = debug_device_delay_ms();
  size_t used = 0;
  for (size_t k = 0; k < nd; ++k) {
    const size_t lo = per * k;
    if (k > 0 && lo >= n) break;
    const size_t cnt = (lo >= n) ? 0 : ((lo + per <= n) ? per : n - lo);
    ++used;
    workers.emplace_back([&, k, lo, cnt]() {
      if (delay > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay));
      rcs[k] = msm_one(ctx->devs[k], encoded, (const uint8_t*)pts_in + lo * rec, scalars + lo * 32, cnt, nullptr,
                       partial.data() + 16 * k, (encoded && cnt) ? status + lo : nullptr);
      if (rcs[k] != D377_OK) errs[k] = d377_g_err;
    });
  }
  for (auto& w : workers) w.join();
  for (size_t k = 0; k < used; ++k)
    if (rcs[k] != D377_OK) return fail(rcs[k], "%s", errs[k].c_str());
  DeviceState& d = ctx->devs[0];
  HIP_TRY(hipSetDevice(d.id));
  rc = D377_OK;
  SyncOnError guard{&rc, d.id, d.stream, nullptr};
  auto combine = [&]() -> int {
    int r;
    if ((r = ensure(d, 0, used * 128))) return r;
    HIP_TRY(hipMemcpyAsync(d.buf[0], partial.data(), used * 128, hipMemcpyHostToDevice, d.stream));
    hipLaunchKernelGGL(k_msm_combine, dim3(1), dim3(64), 0, d.stream, d.tables(), (const uint64_t*)d.buf[0], used, d.buf[2],
                       (uint64_t*)(d.buf[2] + 32));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(enc_out, d.buf[2], 32, hipMemcpyDeviceToHost, d.stream));
    HIP_TRY(hipMemcpyAsync(partial.data(), d.buf[2] + 32, 128, hipMemcpyDeviceToHost, d.stream));
    HIP_TRY(hipStreamSynchronize(d.stream));
    return D377_OK;
  };
  rc = combine();
  if (rc) return rc;
  if (xyzt_out) memcpy(xyzt_out, partial.data(), 128);
  return D377_OK;
}

int msm_dev(d377_ctx* ctx, int dev, void* stream, bool encoded, const void* pts_in, const uint8_t* scalars, size_t n,
            uint8_t* enc_out, uint64_t* xyzt_out, uint8_t* status) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  if (dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "device index out of range");
  if (!enc_out || (n && (!pts_in || !scalars)) || (encoded && n && !status)) return fail(D377_ERR_ARG, "%s", "null buffer");
  if (!aligned16(pts_in) || !aligned16(scalars) || !aligned16(enc_out) || !aligned16(xyzt_out))
    return fail(D377_ERR_ARG, "%s", "device record buffers must be 16-byte aligned");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceState& d = ctx->devs[(size_t)dev];
  HIP_TRY(hipSetDevice(d.id));
  return msm_launch(d, (hipStream_t)stream, encoded, pts_in, scalars, n, enc_out, xyzt_out, status);
}

}  // namespace

extern "C" {

int d377_msm(d377_ctx* ctx, const uint64_t* xyzt, const uint8_t* scalar32, size_t n, uint8_t* enc32_out,
             uint64_t* xyzt_out) {
  return msm_host(ctx, false, xyzt, scalar32, n, enc32_out, xyzt_out, nullptr);
}
int d377_msm_encoded(d377_ctx* ctx, const uint8_t* enc32, const uint8_t* scalar32, size_t n, uint8_t* enc32_out,
                     uint64_t* xyzt_out, uint8_t* status) {
  return msm_host(ctx, true, enc32, scalar32, n, enc32_out, xyzt_out, status);
}
int d377_msm_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* xyzt, const uint8_t* scalar32, size_t n,
                 uint8_t* enc32_out, uint64_t* xyzt_out) {
  return msm_dev(ctx, dev, stream, false, xyzt, scalar32, n, enc32_out, xyzt_out, nullptr);
}
int d377_msm_encoded_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* enc32, const uint8_t* scalar32, size_t n,
                         uint8_t* enc32_out, uint64_t* xyzt_out, uint8_t* status) {
  return msm_dev(ctx, dev, stream, true, enc32, scalar32, n, enc32_out, xyzt_out, status);
}
int d377_sum_elements_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* xyzt, size_t m, uint8_t* enc32_out,
                          uint64_t* xyzt_out) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  if (dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "device index out of range");
  if (!enc32_out || (m && !xyzt)) return fail(D377_ERR_ARG, "%s", "null buffer");
  if (!aligned16(xyzt) || !aligned16(enc32_out) || !aligned16(xyzt_out))
    return fail(D377_ERR_ARG, "%s", "device record buffers must be 16-byte aligned");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceState& d = ctx->devs[(size_t)dev];
  HIP_TRY(hipSetDevice(d.id));
  hipLaunchKernelGGL(k_msm_combine, dim3(1), dim3(64), 0, (hipStream_t)stream, d.tables(), xyzt, m, enc32_out, xyzt_out);
  HIP_TRY(hipGetLastError());
  return D377_OK;
}

}  // extern "C"
