// d377.hip -- gfx950 kernels and the C ABI (include/decaf377_amd.h) of the decaf377 batch engine.
//
// One lane = one group element.  Every kernel loads its 32-byte records as two 16-byte
// vector loads per lane (a wave reads 2 KiB contiguous), keeps all field arithmetic in
// VGPRs (fq29.hpp), and writes 32-byte records back the same way.  Work per element is
// ~10^5 integer MACs against 64-97 bytes of traffic, so the kernels are VALU-bound; the
// memory system only matters for the per-lane window table of the variable-base kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/decaf377_amd.h"
#include "curve.hpp"
#include "device_util.hpp"
#include "dcb.hpp"
#include "quad_ops.hpp"
#include "row_ops.hpp"
#include "host_state.hpp"
#include "codec_chunked.hpp"

using namespace d377;

thread_local char d377_g_err[512] = "";

namespace {

// per-lane window table of the variable-base kernel in global scratch, laid out
// [entry][thread][4 slots x 12 words]: a wave stores one entry as 12 KiB contiguous, and a lane
// fetches its 192-byte (three 64-B sectors) record with 16-byte loads.  A negative digit
// swaps the ypx / ymx slots by address.
struct GlobalTab {
  uint32_t* base;
  size_t nthreads, tid;
  __device__ __forceinline__ void store(int j, const gec& c) {
    uint32_t* p = base + ((size_t)j * nthreads + tid) * VB_ENTRY_WORDS;
    slot_store(p, c.ypx); slot_store(p + SLOT, c.ymx); slot_store(p + 2 * SLOT, c.z2); slot_store(p + 3 * SLOT, c.kt);
  }
  __device__ __forceinline__ gec load(int j, bool swap) const {
    const uint32_t* p = base + ((size_t)j * nthreads + tid) * VB_ENTRY_WORDS;
    gec c;
    c.ypx = slot_load(p + (swap ? SLOT : 0));
    c.ymx = slot_load(p + (swap ? 0 : SLOT));
    c.z2 = slot_load(p + 2 * SLOT);
    c.kt = slot_load(p + 3 * SLOT);
    return c;
  }
};
// shared fixed-base comb FB[i][j] = affine cached j * 2^(FB_BITS i) * B, [FB_WINDOWS][FB_ENTRIES] 128-byte records
// (device_util.hpp: pt_load_affine -- two sectors per gather; round 2's three 48-byte slots cost 3.25)
template <int BITS>
struct FixedTab {
  const uint32_t* base;
  __device__ __forceinline__ gea load(int i, int j, bool swap) const {
    return pt_load_affine(base + ((size_t)i * FbShape<BITS>::entries + j) * FBW_ENTRY_WORDS, swap);
  }
};

// ------------------------------------------------------------------------- init kernels ---
__device__ fe fe_pow_u32(const fe& x, uint32_t e) {   // e >= 1
  int top = 31 - __clz((int)e);
  fe r = x;
#pragma unroll 1
  for (int i = top - 1; i >= 0; --i) {
    r = fe_sqr(r);
    if ((e >> i) & 1u) r = fe_mul(r, x);
  }
  return r;
}

// gtab[t][nu] = g^(nu * 2^(8t)), g = zeta^m  (src/ark_curve/invsqrt.rs:41-50)
__global__ void __launch_bounds__(BLOCK) k_init_gtab(uint32_t* gtab) {
  const int idx = blockIdx.x * BLOCK + threadIdx.x;
  if (idx >= 6 * 256) return;
  const int t = idx >> 8, nu = idx & 255;
  fe base = fe_sqr_n(fe_const(FE_SQRT_G), 8 * t);
  fe v = nu == 0 ? fe_const(FE_ONE) : fe_pow_u32(base, (uint32_t)nu);
  v = fe_mul_strict(v, fe_const(FE_ONE));       // same value, below 1.02q: keeps the s_lookup keys within x, x + q
  uint32_t* p = gtab + (size_t)idx * GT_STRIDE;
#pragma unroll
  for (int i = 0; i < NL; ++i) p[i] = v.l[i];
  p[9] = 0; p[10] = 0; p[11] = 0;
}

// s_lookup: keys g^-(nu * 2^39) (invsqrt.rs:27-39) in both tight representations -> nu.
// One block of 256 threads; thread 0 then inserts sequentially and counts collisions.
__global__ void __launch_bounds__(BLOCK) k_init_slookup(uint8_t* s_lookup, uint32_t* keys /*[256][2]*/,
                                                        int* collisions) {
  const int nu = threadIdx.x;
  fe b39 = fe_sqr_n(fe_const(FE_SQRT_G_INV), 39);
  fe v = nu == 0 ? fe_const(FE_ONE) : fe_pow_u32(b39, (uint32_t)nu);
  fe c = fe_reduce_once(fe_mul_strict(v, fe_const(FE_ONE)));
  keys[2 * nu] = s_hash_raw(c);
  uint32_t h2 = 0xFFFFFFFFu;
  if (c.l[NL - 1] < (1u << 16)) {        // x < 2^248: x + q is a possible product representation
    fe cq = fe_add(c, fe_const(Q_LIMBS));
    uint32_t carry = 0;
    fe n;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      uint32_t t = cq.l[i] + carry;
      if (i < NL - 1) { n.l[i] = t & MASK29; carry = t >> RB; } else n.l[i] = t;
    }
    h2 = s_hash_raw(n);
  }
  keys[2 * nu + 1] = h2;
  for (int i = threadIdx.x; i < (1 << S_HASH_BITS); i += BLOCK) s_lookup[i] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    int coll = 0;
    // owner map in the table itself is ambiguous for nu = 0, so track with a second pass
    for (int k = 0; k < 256; ++k)
      for (int r = 0; r < 2; ++r) {
        uint32_t h = keys[2 * k + r];
        if (h == 0xFFFFFFFFu) continue;
        for (int k2 = 0; k2 < k; ++k2)
          for (int r2 = 0; r2 < 2; ++r2)
            if (keys[2 * k2 + r2] == h) ++coll;
        s_lookup[h] = (uint8_t)k;
      }
    *collisions = coll;
  }
}

// FB[i][j] = j * 2^(FB_BITS i) * B in affine cached form, i < FB_WINDOWS, j < FB_ENTRIES.  A thread builds a RUN of
// FB_RUN consecutive multiples of one window: j0 * base by double-and-add, then one addition of the base per entry; the
// projective coordinates are parked in the entries' own records (27 limbs = a record's 27 words) and the run's Z's are
// inverted together (Montgomery's trick: one divsteps inversion per FB_RUN entries).  ~9 000 instructions per entry;
// one thread per entry with its own ladder from B and its own inversion was ~420 000 at 21-bit windows (12.6 M entries).
constexpr int FB_RUN = 16;
template <int FB_BITS>
__global__ void k_init_fbase_bases(uint32_t* bases) {          // bases[i] = 2^(FB_BITS i) * B as X, Y, Z, T: one thread
  constexpr int FB_WINDOWS = FbShape<FB_BITS>::windows;
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  ge p = ge_generator();
#pragma unroll 1
  for (int i = 0; i < FB_WINDOWS; ++i) {
    slot_store(bases + (size_t)i * 4 * SLOT, p.x); slot_store(bases + (size_t)i * 4 * SLOT + SLOT, p.y);
    slot_store(bases + (size_t)i * 4 * SLOT + 2 * SLOT, p.z); slot_store(bases + (size_t)i * 4 * SLOT + 3 * SLOT, p.t);
#pragma unroll 1
    for (int k = 0; k < FB_BITS; ++k) p = ge_double(p);
  }
}
template <int FB_BITS>
__global__ void __launch_bounds__(BLOCK) k_init_fbase(const uint32_t* bases, uint32_t* fb) {
  constexpr int FB_WINDOWS = FbShape<FB_BITS>::windows, FB_ENTRIES = FbShape<FB_BITS>::entries;
  constexpr int RUNS = (FB_ENTRIES + FB_RUN - 1) / FB_RUN;
  const size_t idx = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (idx >= (size_t)FB_WINDOWS * RUNS) return;
  const int i = (int)(idx / RUNS), j0 = (int)(idx % RUNS) * FB_RUN;
  ge base;
  base.x = slot_load(bases + (size_t)i * 4 * SLOT); base.y = slot_load(bases + (size_t)i * 4 * SLOT + SLOT);
  base.z = slot_load(bases + (size_t)i * 4 * SLOT + 2 * SLOT); base.t = slot_load(bases + (size_t)i * 4 * SLOT + 3 * SLOT);
  ge acc = ge_identity();
#pragma unroll 1
  for (int b = FB_BITS - 1; b >= 0; --b) {                     // acc = j0 * base
    acc = ge_double(acc);
    if ((j0 >> b) & 1) acc = ge_add(acc, base);
  }
  uint32_t* rec0 = fb + ((size_t)i * FB_ENTRIES + j0) * FBW_ENTRY_WORDS;
  fe prefix[FB_RUN];
  fe c = fe_const(FE_ONE);
#pragma unroll
  for (int r = 0; r < FB_RUN; ++r) {
    if (j0 + r < FB_ENTRIES) {
      uint32_t* q = rec0 + (size_t)r * FBW_ENTRY_WORDS;
#pragma unroll
      for (int k = 0; k < NL; ++k) { q[k] = acc.x.l[k]; q[NL + k] = acc.y.l[k]; q[2 * NL + k] = acc.z.l[k]; }
      prefix[r] = c;
      c = fe_mul(c, acc.z);
      acc = ge_add(acc, base);
    }
  }
  fe inv = fe_invert(c);
#pragma unroll
  for (int r = FB_RUN - 1; r >= 0; --r) {
    if (j0 + r < FB_ENTRIES) {
      uint32_t* q = rec0 + (size_t)r * FBW_ENTRY_WORDS;
      fe X, Y, Z;
#pragma unroll
      for (int k = 0; k < NL; ++k) { X.l[k] = q[k]; Y.l[k] = q[NL + k]; Z.l[k] = q[2 * NL + k]; }
      const fe zi = fe_mul(inv, prefix[r]);
      inv = fe_mul(inv, Z);
      pt_store_affine(q, gea_from_affine(fe_mul(X, zi), fe_mul(Y, zi)));
    }
  }
}

// --------------------------------------------------------------------------- batch kernels ---
// The kernels that work in chunks (dcb_rounds above) hand their square roots the inverses of their denominators.
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_sqrt_ratio_zeta(SqrtTables T, const uint8_t* num32,
                                                           const uint8_t* den32, size_t n,
                                                           uint8_t* root32, uint8_t* was_square, int min_curve_root, DcbScratch dcb) {
  D377_POW_LDS();
  D377_DCB_BEGIN(root32);
  dcb_rounds<1, false>(n, io, pt,
    [&](size_t i, int j) {
      uint32_t wd[8];
      load32(den32, i, wd);
      dcb_put_den(io, 0, j, fe_from_words_mod_order_strict(wd));
    },
    [&](size_t i, int j, const uint32_t (*invw)[8], bool have) {
      uint32_t wn[8], wd[8], wr[8];
      load32(num32, i, wn);
      load32(den32, i, wd);
      fe inv = fe_from_words(invw[0]);            // meaningless words when !have (never read then)
      fe r;
      const bool ws = fe_sqrt_ratio_zeta<false>(T, pt, fe_from_words_mod_order_strict(wn), fe_from_words_mod_order_strict(wd), &r,
                                                min_curve_root != 0, &inv, have);
      fe_to_bytes_words(r, wr);
      store32(root32, i, wr);
      was_square[i] = ws ? 1 : 0;
    });
  D377_DCB_END();
}

// Elements per resident lane from which decompress, compress and the round trip take their chunked forms.  Their shared
// inversion paid from 3 per lane (393 216 elements on 256 CUs); with issue priority by progress (dcb.hpp) the chunked forms lead
// from 2 (262 144: chunked / wide 0.97-0.99, 393 216: 0.94-0.96, 2^20: 0.90-0.95; profiles/r05_decompress_route_sweep.txt).
constexpr int CODEC_CHUNKED_MIN = 2;
// decompress, compress and the round trip run one element per lane on the wide grid, each square root in the
// reference's inversion-free form: in chunks with batched inverses they execute 4-8 % fewer instructions.  All three
// take the chunked form (k_decompress_chunked below; k_compress_chunked, k_roundtrip_chunked in codec_chunked.hip) from
// CODEC_CHUNKED_MIN elements per resident lane (measured before the priorities, when the rule switched at 3 x the resident lanes:
// decompress -4 % there, -5 % at 2^19, -7 % from 2^20 on; compress -1 / -1 / -6 / -4 %; round trip -3 / -3 / -3 / -5 %,
// -6 % at 2^22: profiles/r05_decompress_route_sweep.txt; rounds 2-4 switched the decompression at 2^21 and left the
// others on the wide grid, before a wave shared one inversion and before the rounds were dealt out evenly).
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_decompress(SqrtTables T, const uint8_t* enc32, size_t n,
                                                      uint64_t* xyzt, uint8_t* status) {
  D377_POW_LDS();
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t w[8];
    load32(enc32, i, w);
    ge g;
    const uint32_t bad = ge_decompress(T, pt, w, &g);
    status[i] = (uint8_t)bad;
    if (bad) {
      uint8_t* b = reinterpret_cast<uint8_t*>(xyzt);
      store32_zero(b, 4 * i); store32_zero(b, 4 * i + 1); store32_zero(b, 4 * i + 2); store32_zero(b, 4 * i + 3);
    } else {
      store_ge_mont256(xyzt, i, g);
    }
  }
}

// The same in chunks, the square roots' denominators inverted together (as k_scalar_mul_var decodes its points): fewer
// instructions per element, one generation of workgroups per 2^20 elements -- for batches of several generations.
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_decompress_chunked(SqrtTables T, const uint8_t* enc32, size_t n,
                                                              uint64_t* xyzt, uint8_t* status, DcbScratch dcb) {
  D377_POW_LDS();
  D377_DCB_BEGIN(xyzt);
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
      if (bad) {
        uint8_t* b = reinterpret_cast<uint8_t*>(xyzt);
        store32_zero(b, 4 * i); store32_zero(b, 4 * i + 1); store32_zero(b, 4 * i + 2); store32_zero(b, 4 * i + 3);
      } else {
        store_ge_mont256(xyzt, i, g);
      }
    });
  D377_DCB_END();
}

__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_compress(SqrtTables T, const uint64_t* xyzt, size_t n,
                                                    uint8_t* enc32) {
  D377_POW_LDS();
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t w[8];
    ge_compress(T, pt, load_ge_mont256(xyzt, i), w);
    store32(enc32, i, w);
  }
}

// decompress -> compress.  The compressor here is the generic one (its own square root, from the coordinates alone):
// a round trip that used the encoding it was given as the point's known preimage would not compress anything.
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_roundtrip(SqrtTables T, const uint8_t* enc32, size_t n,
                                                     uint8_t* out32, uint8_t* status) {
  D377_POW_LDS();
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t w[8];
    load32(enc32, i, w);
    ge g;
    const uint32_t bad = ge_decompress(T, pt, w, &g);
    ge_compress(T, pt, g, w, bad == 0);
    status[i] = (uint8_t)bad;
    if (bad) store32_zero(out32, i); else store32(out32, i, w);
  }
}

// The next kernels end in an encoding of a point whose isogeny preimage they know (a doubling, or the Elligator
// map's (s, t)): each lane leaves the four 32-byte state records of its elements in `dcb` and the square-root-free
// compressor finishes the round with one inversion.
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_encode_to_curve(SqrtTables T, const uint8_t* fq32, size_t n,
                                                           uint8_t* out32, DcbScratch dcb) {
  D377_POW_LDS();
  D377_DCB_BEGIN(out32);
  dcb_rounds<1, true>(n, io, pt,
    [&](size_t i, int j) {
      uint32_t w[8];
      load32(fq32, i, w);
      dcb_put_den(io, 0, j, ge_elligator_den(fe_from_words_mod_order(w)));
    },
    [&](size_t i, int j, const uint32_t (*invw)[8], bool have) {
      uint32_t w[8];
      load32(fq32, i, w);
      fe inv = fe_from_words(invw[0]);            // meaningless words when !have (never read then)
      fe s, t;
      ge_elligator_st(T, pt, fe_from_words_mod_order(w), &s, &t, &inv, have);
      D377_INVARIANT(T, ge_from_jacobi_st(s, t), true);
      dcb_put(io, j, ge_dcb_from_jacobi_st(s, t));
    });
  D377_DCB_END();
}

// hash_to_curve (elligator.rs:67-71): two maps (their square roots take batched inverses) and the encoding of the sum.
// The sum is formed on the Jacobi quartic the maps land on, where the decaf isogeny makes it a preimage of the Edwards
// sum, so the encoding needs no third square root (curve.hpp, ge_dcb_from_jacobi_sum): 1.8e8 -> 2.5e8 /s at 2^20.  A pair
// that hits the addition law's exceptional case (s1 s2 = +-1) goes the reference's way, Edwards addition and generic
// compression, and enters the batch as a finished encoding; the branch is taken by a wave only if one of its lanes needs it.
// (Handed nothing but the pair's input words BY VALUE: it maps both inputs again, in the reference's inversion-free form.
// References to the caller's (s, t) values would put those in scratch memory for EVERY element -- 144 bytes of stores per
// element on the hot path, which is what the first version did -- to save four square roots on a route no known input
// takes.  Inline since it left the per-element loops: as a call it cost its callers an argument block and six more
// registers in scratch -- k_hash_to_curve 80 -> 24 bytes per lane, what is left are addresses kept per chunk; the tiny
// kernels none at all -- for the same time, profiles/r05_ab_exceptional_inline.txt.)
struct Words8 { uint32_t w[8]; };
template <class PT>
__device__ __forceinline__ Words8 hash_exceptional_words(SqrtTables T, PT pt, Words8 a1, Words8 a2) {
  fe s1, t1, s2, t2, unused = fe_zero();
  ge_elligator_st(T, pt, fe_from_words_mod_order(a1.w), &s1, &t1, &unused, false);
  ge_elligator_st(T, pt, fe_from_words_mod_order(a2.w), &s2, &t2, &unused, false);
  Words8 r;
  ge_compress(T, pt, ge_add(ge_from_jacobi_st(s1, t1), ge_from_jacobi_st(s2, t2)), r.w);
  return r;
}
template <class PT>
__device__ __forceinline__ void hash_exceptional_pair(SqrtTables T, PT pt, const uint8_t* r1, const uint8_t* r2, size_t i, uint32_t w[8]) {
  Words8 a1, a2;
  load32(r1, i, a1.w);
  load32(r2, i, a2.w);
  const Words8 r = hash_exceptional_words(T, pt, a1, a2);
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = r.w[k];
}
// The exceptional pair in the chunked kernel: NOT in the per-element loop (a call there made the compiler spill 22 VGPRs
// around it and shaped the register allocation of the loop that every element walks).  An element that needs the route
// enters the chunk's compressor as the neutral state, its two input records are parked in the records of the state's
// numerators (slots 2, 3: read by the compressor for this element only, and its output is overwritten) -- the inputs
// themselves may be gone by then: a caller may hash in place -- and after the chunk's encodings have been written the
// lanes that flagged an element redo it the reference's way.  A wave enters only if one of its lanes flagged something.
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_hash_to_curve(SqrtTables T, const uint8_t* r1, const uint8_t* r2,
                                                         size_t n, uint8_t* out32, DcbScratch dcb) {
  D377_POW_LDS();
  D377_DCB_BEGIN(out32);
  uint32_t flagged = 0;                              // bit j: element j of the current chunk takes the exceptional route
  dcb_rounds<2, true>(n, io, pt,
    [&](size_t i, int j) {
      uint32_t w[8];
      load32(r1, i, w);
      dcb_put_den(io, 0, j, ge_elligator_den(fe_from_words_mod_order(w)));
      load32(r2, i, w);
      dcb_put_den(io, 1, j, ge_elligator_den(fe_from_words_mod_order(w)));
    },
    [&](size_t i, int j, const uint32_t (*invw)[8], bool have) {
      uint32_t w[8];
      load32(r1, i, w);
      fe inv = fe_from_words(invw[0]);            // meaningless words when !have (never read then)
      fe s1, t1, s2, t2;
      ge_elligator_st(T, pt, fe_from_words_mod_order(w), &s1, &t1, &inv, have);
      load32(r2, i, w);
      inv = fe_from_words(invw[1]);
      ge_elligator_st(T, pt, fe_from_words_mod_order(w), &s2, &t2, &inv, have);
      D377_INVARIANT(T, ge_from_jacobi_st(s1, t1), true);
      D377_INVARIANT(T, ge_from_jacobi_st(s2, t2), true);
      bool exceptional;
      dcb_state st = ge_dcb_from_jacobi_sum(s1, t1, s2, t2, &exceptional);
#if defined(D377_CHECK_INVARIANTS)
      exceptional |= (i & 3) == 3;                  // the debug build sends every fourth pair down the exceptional route, so that
                                                   // the GPU suite runs it (no input pair is known that takes it by itself)
#endif
      const dcb_state ne = dcb_neutral();
      st.p = fe_select(exceptional, ne.p, st.p); st.w = fe_select(exceptional, ne.w, st.w);
      st.n0 = fe_select(exceptional, ne.n0, st.n0); st.n1 = fe_select(exceptional, ne.n1, st.n1);
      dcb_put(io, j, st);
      if (exceptional) {                            // (a few loads and stores under a branch no known input takes)
        flagged |= 1u << j;
        load32(r1, i, w);
        io.put(2, j, w);
        load32(r2, i, w);
        io.put(3, j, w);
      }
    },
    [&](DcbIO& io2, int cnt) {
      if (__any(flagged != 0)) {
#pragma unroll 1
        for (int j = 0; j < cnt; ++j) {
          const bool mine = ((flagged >> j) & 1u) != 0;
          if (!__any(mine)) continue;
          Words8 a1, a2;
          io2.get(2, j, a1.w);                      // (the other lanes of the wave: whatever their records hold -- any words are inputs)
          io2.get(3, j, a2.w);
          const Words8 r = hash_exceptional_words(T, pt, a1, a2);
          if (mine) io2.emit(j, r.w);
        }
      }
      flagged = 0;
    });
  D377_DCB_END();
}

// [k]P = [2]([k/2 mod r]P): the window loop runs on k/2 and the encoding is that of the double (no square root)
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_scalar_mul_var(SqrtTables T, const uint8_t* enc32,
                                                          const uint8_t* scalar32, size_t n, uint8_t* out32,
                                                          uint8_t* status, uint32_t* scratch, DcbScratch dcb) {
  D377_POW_LDS();
  D377_DCB_BEGIN(out32);
  GlobalTab tab;
  tab.base = scratch;
  tab.nthreads = (size_t)dcb.nslots * BLOCK;            // the window tables exist for this kernel's sets only (vb_scratch)
  tab.tid = io.lane;
  dcb_rounds<1, true, false>(n, io, pt,
    [&](size_t i, int j) {
      uint32_t w[8];
      load32(enc32, i, w);
      dcb_put_den(io, 0, j, ge_decompress_den(w));
    },
    [&](size_t i, int j, const uint32_t (*invw)[8], bool have) {
      uint32_t w[8], k[8], dg[8];
      load32(enc32, i, w);
      load32(scalar32, i, k);
      const fe inv = fe_from_words(invw[0]);
      ge g;
      const uint32_t bad = ge_decompress(T, pt, w, &g, &inv);
      status[i] = (uint8_t)bad;
      fr_reduce_words(k);
      fr_half_words(k);
      fr_recode_signed16(k, dg);
      const ge r = ge_scalar_mul_w4(g, dg, tab, DCB_WANT_T);
      D377_INVARIANT(T, r, bad == 0);
      dcb_put(io, j, ge_dcb_from_half(r, bad != 0));      // failed lanes: neutral state, all-zero output
    });
  D377_DCB_END();
}

// Up to three workgroups per CU (168 VGPRs): this kernel's additions wait on table gathers from HBM (11 random 128-byte
// records per element of the 23-bit comb, L2 hit rate 0.25), which a third wave per SIMD hides a little better, and large batches share one
// inversion among FB_K = 16 elements per lane instead of 8 (launch(): OP_MUL_BASE chooses per call).
template <int BITS>
__global__ void __launch_bounds__(BLOCK, FB_SETS) k_scalar_mul_base(SqrtTables T, const uint32_t* fbase,
                                                           const uint8_t* scalar32, size_t n, uint8_t* out32, DcbScratch dcb) {
  D377_POW_LDS();                        // unused here (no square root): residency is capped by the launch's LDS padding
  D377_DCB_BEGIN(out32);
  FixedTab<BITS> ft{fbase};
  dcb_rounds<0, true>(n, io, pt,
    [&](size_t, int) {},
    [&](size_t i, int j, const uint32_t (*)[8], bool) {
      uint32_t k[8];
      load32(scalar32, i, k);
      fr_reduce_words(k);
      fr_half_words(k);
      const ge r = ge_scalar_mul_base_w8<BITS>(k, ft, DCB_WANT_T);
      D377_INVARIANT(T, r, true);
      dcb_put(io, j, ge_dcb_from_half(r, false));
    });
  D377_DCB_END();
}

// ---- small batches: one group element per QUAD of lanes -----------------------------------------------------------------
// Below ~2^14 elements the chip is not full with one lane per element and a call takes as long as ONE element's
// dependency chain: 1.06 ms for a variable-base multiplication, whatever the batch (profiles/r03_size_sweep.txt).  Most
// of that chain is group operations, and those split four ways (quad_ops.hpp: a doubling or an addition is two rounds of
// four independent products): a quad of lanes takes one element, each lane one product per round.  The square root of
// the decompression and the final inversion stay one dependent chain of squarings and are simply done by all four
// lanes alike.  The window table (cached 0 .. 8 P, the slots each lane multiplies by) lives in LDS, one per quad.
// Four times the lanes and ~2.4 x fewer instructions per lane: used while n <= the quads the chip holds at one wave
// per SIMD (launch(): small_max), where the other lanes would have idled anyway.
constexpr int SMALL_THREADS = 64;                              // one wave per workgroup, 16 elements
constexpr int SMALL_QUADS = SMALL_THREADS / 4;
static_assert(GQ_TAB_ENTRIES == VB_ENTRIES, "the quad chain's table is the per-lane chain's: 0 .. 8 times P");
struct OneDcbIO {                                              // the square-root-free compressor's records for a single element
  uint32_t st[4][8], parked_[8], out[8];
  __device__ __forceinline__ void put(int s, int, const uint32_t* w) { for (int k = 0; k < 8; ++k) st[s][k] = w[k]; }
  __device__ __forceinline__ void get(int s, int, uint32_t* w) const { for (int k = 0; k < 8; ++k) w[k] = st[s][k]; }
  __device__ __forceinline__ void park(int, const uint32_t* w) { for (int k = 0; k < 8; ++k) parked_[k] = w[k]; }
  __device__ __forceinline__ void parked(int, uint32_t* w) const { for (int k = 0; k < 8; ++k) w[k] = parked_[k]; }
  __device__ __forceinline__ void emit(int, const uint32_t* w) { for (int k = 0; k < 8; ++k) out[k] = w[k]; }
};
// One inversion for the sixteen quads of a wave: each quad holds one value p (its four lanes alike, never zero); every lane
// gets 1 / p of its own quad.  The product of the sixteen by an exchange tree over the quads (the partner's partial product at
// every level is what the way back multiplies by), the wave's inversion of the total (row_ops.hpp): 8 products, 36 shuffles
// and 17 us, against 31 us for sixteen inversions side by side on lanes.
__device__ __forceinline__ fe fe_shfl_xor(const fe& a, int mask) {
  fe r;
#pragma unroll
  for (int i = 0; i < NL; ++i) r.l[i] = (uint32_t)__shfl_xor((int)a.l[i], mask);
  return r;
}
__device__ __forceinline__ fe quads16_invert(const fe& p) {
  const fe q0 = fe_shfl_xor(p, 4), p1 = fe_mul_strict(p, q0);
  const fe q1 = fe_shfl_xor(p1, 8), p2 = fe_mul_strict(p1, q1);
  const fe q2 = fe_shfl_xor(p2, 16), p3 = fe_mul_strict(p2, q2);
  const fe q3 = fe_shfl_xor(p3, 32);
  const fe inv = row::fe_invert_wave(fe_mul_strict(p3, q3));      // the same total in every lane
  return fe_mul(fe_mul(fe_mul(fe_mul(inv, q3), q2), q1), q0);
}

// ELEMENT: the reference's own signature (Element * Fr -> Element, src/min_curve/ops.rs:89-95): records in and out, no
// square root at either end, so the whole chain splits four ways (0.88 -> ~0.36 ms per call).
template <bool ELEMENT>
__global__ void __launch_bounds__(SMALL_THREADS)
k_scalar_mul_var_small(SqrtTables T, const uint8_t* enc32, const uint8_t* scalar32, size_t n, uint8_t* out32, uint8_t* status) {
  __shared__ uint32_t lds_pow_[POW_TAB * NL * SMALL_THREADS];
  __shared__ uint32_t tab[SMALL_QUADS * VB_ENTRIES * GQ_WORDS];
  struct Pow64 {                                               // the square root's odd powers, one LDS column per lane
    uint32_t* col;
    __device__ __forceinline__ void put(int j, const fe& v) { for (int k = 0; k < NL; ++k) col[(j * NL + k) * SMALL_THREADS] = v.l[k]; }
    __device__ __forceinline__ fe get(int j) const { fe r; for (int k = 0; k < NL; ++k) r.l[k] = col[(j * NL + k) * SMALL_THREADS]; return r; }
  } pt;
  pt.col = lds_pow_ + threadIdx.x;
  const int role = threadIdx.x & 3, quad = threadIdx.x >> 2;
  uint32_t* qtab = tab + quad * VB_ENTRIES * GQ_WORDS;
  for (size_t base = (size_t)blockIdx.x * SMALL_QUADS; base < n; base += (size_t)gridDim.x * SMALL_QUADS) {
    const bool active = base + quad < n;
    const size_t e = active ? base + quad : n - 1;             // idle quads redo the last element and store nothing
    uint32_t w[8], k[8], dg[8];
    load32(scalar32, e, k);
    ge g;
    uint32_t bad = 0;
    fr_reduce_words(k);
    if (ELEMENT) {
      g = load_ge_mont256(reinterpret_cast<const uint64_t*>(enc32), e);
    } else {
      load32(enc32, e, w);
      bad = ge_decompress(T, pt, w, &g);                        // every lane of the quad: the same chain of squarings
      fr_half_words(k);                                        // [k]P = [2]([k/2 mod r]P): the encoding of a double needs no square root
    }
    fr_recode_signed16(k, dg);
    const fe v = gq_scalar_mul_w4(gq_from_ge(g, role), dg, qtab, role);
    const ge r = gq_to_ge(v);
    D377_INVARIANT(T, r, bad == 0);
    if (ELEMENT) {
      if (active && role == 0) store_ge_mont256(reinterpret_cast<uint64_t*>(out32), e, r);
    } else {
      const dcb_state st = ge_dcb_from_half(r, bad != 0);     // failed elements: neutral state, all-zero output
      uint32_t wo[8];
      dcb_encode_one(st, quads16_invert(st.p), wo);             // one inversion for the sixteen elements of the wave
      if (active && role == 0) {
        store32(out32, e, wo);
        status[e] = (uint8_t)bad;
      }
    }
    __syncthreads();                                           // the table is rewritten by the next element
  }
}

// The fixed-base multiplication on quads, for the same batch sizes: FB_WINDOWS mixed additions of comb entries are 2.7 us each
// on a lane and 1.1 us on a quad (the lane takes its slot of the entry straight from the table: the comb's records are
// cached affine points in the internal form already), and the sixteen encodings of a wave share one inversion: 0.098 ->
// ~0.06 ms per call.
// lane `role`'s slot of window i's comb entry for the digit d (the records hold Y + X, Y - X, 2dXY; Z = 1).
// (A free function, not a lambda inside the kernel: with the lambda hipcc compiled EVERY kernel of this file differently --
// k_scalar_mul_var 249 VGPRs and 84 SGPR spills instead of 256 and 47 -- which tools/resource_usage.sh is there to catch.)
template <int FB_BITS>
__device__ __forceinline__ fe fb_fetch_slot(const uint32_t* fbase, int role, int i, int d) {
  const int slot = gq_add_slot(role, d < 0);
  const uint32_t* src = fbase + ((size_t)i * FbShape<FB_BITS>::entries + (size_t)(d < 0 ? -d : d)) * FBW_ENTRY_WORDS + NL * (slot == 0 ? 1 : (slot == 1 ? 0 : 2));
  fe x = fe_const(FE_ONE);
  if (slot < 3) {
#pragma unroll
    for (int j = 0; j < NL; ++j) x.l[j] = src[j];
  }
  return x;
}
template <bool ELEMENT, int FB_BITS>
__global__ void __launch_bounds__(SMALL_THREADS)
k_scalar_mul_base_small(const uint32_t* fbase, const uint8_t* scalar32, size_t n, uint8_t* out) {
  constexpr int FB_WINDOWS = FbShape<FB_BITS>::windows;
  const int role = threadIdx.x & 3, quad = threadIdx.x >> 2;
  const size_t e_raw = (size_t)blockIdx.x * SMALL_QUADS + quad;   // grid = ceil(n / 16)
  const bool active = e_raw < n;
  const size_t e = active ? e_raw : n - 1;                       // idle quads redo the last element and store nothing
  uint32_t k[8];
  load32(scalar32, e, k);
  fr_reduce_words(k);
  if (!ELEMENT) fr_half_words(k);                                // [k]B = [2]([k/2 mod r]B): the encoding of a double needs no square root
  uint32_t carry = 0;
  int d = fb_digit<FB_BITS>(k, 0, carry);
  fe nxt = fb_fetch_slot<FB_BITS>(fbase, role, 0, d);
  fe v = gq_from_ge(ge_identity(), role);
#pragma unroll 1
  for (int i = 0; i < FB_WINDOWS; ++i) {
    const fe cur = nxt;
    const bool neg = d < 0;
    if (i + 1 < FB_WINDOWS) {                                    // the next entry is in flight during this addition
      d = fb_digit<FB_BITS>(k, i + 1, carry);
      nxt = fb_fetch_slot<FB_BITS>(fbase, role, i + 1, d);
    }
    v = gq_add_with(v, cur, role, neg);
  }
  const ge r = gq_to_ge(v);
  if (ELEMENT) {
    if (active && role == 0) store_ge_mont256(reinterpret_cast<uint64_t*>(out), e, r);
  } else {
    const dcb_state st = ge_dcb_from_half(r, false);
    uint32_t w[8];
    dcb_encode_one(st, quads16_invert(st.p), w);
    if (active && role == 0) store32(out, e, w);
  }
}

// The smallest batches -- up to one element per SIMD, 4 x the CUs -- give every element a WAVE: the group operations run in
// the lane-spread form (row_ops.hpp: the point across the four rows, a field product ~100 instructions on the critical path
// instead of ~200), so the chain of 252 doublings and 63 additions is half as long as on a quad; the square root of the
// decompression and the inversion of the encoding stay whole-element chains that every lane repeats.
template <bool ELEMENT>
__global__ void __launch_bounds__(64)
k_scalar_mul_var_tiny(SqrtTables T, const uint8_t* enc32, const uint8_t* scalar32, size_t n, uint8_t* out32, uint8_t* status) {
  __shared__ uint32_t tab[row::RQ_TAB_ENTRIES * row::RQ_WORDS];
  __shared__ uint32_t xrec[2 * row::RQ_WORDS];
  const int t = threadIdx.x;
  const row::RowK K = row::row_consts();
  const row::RowSel S = row::row_sel();
  const size_t e = blockIdx.x;                                   // grid = n
  uint32_t w[8], k[8], dg[8];
  load32(scalar32, e, k);
  ge g;
  uint32_t bad = 0;
  fr_reduce_words(k);
  if (ELEMENT) {
    g = load_ge_mont256(reinterpret_cast<const uint64_t*>(enc32), e);
  } else {
    load32(enc32, e, w);
    // the square root's power chains in the lane-spread form (every row the same element), its table phase and the rest of
    // the decompression as whole-element code
    if (t < 4) row::row_store_from_fe(xrec + 16 * t, ge_decompress_den(w));
    __syncthreads();
    const row::RowPowers pw = row::row_sqrt_powers(xrec[t], tab, t, K);
    __syncthreads();
    xrec[t] = pw.v; xrec[row::RQ_WORDS + t] = pw.uv;
    __syncthreads();
    const fe pv = row::row_load_to_fe(xrec), puv = row::row_load_to_fe(xrec + row::RQ_WORDS);
    __syncthreads();
    bad = ge_decompress_from_powers(T, w, pv, puv, &g);
    fr_half_words(k);                                            // [k]P = [2]([k/2 mod r]P): the encoding of a double needs no square root
  }
  fr_recode_signed16(k, dg);
  if (t < 4) row::row_store_from_fe(xrec + 16 * t, fe_pick(t, g.x, g.y, g.z, g.t));
  __syncthreads();
  const uint32_t v = row::rq_scalar_mul_w4(xrec[t], dg, tab, S, K);
  __syncthreads();
  xrec[t] = v;
  __syncthreads();
  const ge r = row::rq_load_point(xrec);
  D377_INVARIANT(T, r, bad == 0 && t == 0);
  if (ELEMENT) {
    if (t == 0) store_ge_mont256(reinterpret_cast<uint64_t*>(out32), e, r);
  } else {
    OneDcbIO io;
    dcb_put(io, 0, ge_dcb_from_half(r, bad != 0));               // failed elements: neutral state, all-zero output
    dcb_finish_with(io, 1, [](const fe& c) { return row::fe_invert_wave(c); });   // (every lane holds the same element)
    if (t == 0) {
      store32(out32, e, io.out);
      status[e] = (uint8_t)bad;
    }
  }
}

// The fixed-base multiplication of the same regime: FB_WINDOWS mixed additions of comb entries, which every lane of a
// lane-per-element wave does alone (~2.7 us each, then a 31 us inversion: 91 us per call whatever the batch).  One wave per
// scalar instead: the comb entries of its digits, fetched and turned into row records by 3 x FB_WINDOWS lanes side by side,
// then the additions in the lane-spread form (0.55 us each) and the wave's inversion (18 us).
template <bool ELEMENT, int FB_BITS>
__global__ void __launch_bounds__(64)
k_scalar_mul_base_tiny(const uint32_t* fbase, const uint8_t* scalar32, size_t n, uint8_t* out) {
  constexpr int FB_WINDOWS = FbShape<FB_BITS>::windows, FB_ENTRIES = FbShape<FB_BITS>::entries;
  __shared__ uint32_t crec[FB_WINDOWS * row::RQ_WORDS];
  __shared__ uint32_t xrec[row::RQ_WORDS];
  __shared__ int sdig[FB_WINDOWS];
  const int t = threadIdx.x;
  const row::RowK K = row::row_consts();
  const row::RowSel S = row::row_sel();
  const size_t e = blockIdx.x;                                   // grid = n
  uint32_t k[8];
  load32(scalar32, e, k);
  fr_reduce_words(k);
  if (!ELEMENT) fr_half_words(k);                                // [k]B = [2]([k/2 mod r]B): the encoding of a double needs no square root
  if (t == 0) {
    uint32_t carry = 0;
#pragma unroll 1
    for (int i = 0; i < FB_WINDOWS; ++i) sdig[i] = fb_digit<FB_BITS>(k, i, carry);
  }
  __syncthreads();
  // lane l: coordinate l % 4 of window l / 4's entry (Y + X, Y - X, 2dXY as stored; the fourth is Z = 1) -> the slot a row
  // multiplies by (row_ops.hpp rq_add: 0 Y - X, 1 Y + X, 2 2dT, 3 Z)
#pragma unroll 1
  for (int l = t; l < 4 * FB_WINDOWS; l += 64) {
    const int i = l >> 2, c = l & 3, d = sdig[i];
    uint32_t* rec = crec + i * row::RQ_WORDS + 16 * (c == 0 ? 1 : (c == 1 ? 0 : c));
    if (c == 3) {
      for (int j = 0; j < 16; ++j) rec[j] = j == 0 ? 1u : 0u;
    } else {
      const uint32_t* src = fbase + ((size_t)i * FB_ENTRIES + (size_t)(d < 0 ? -d : d)) * FBW_ENTRY_WORDS + NL * c;
      fe x;
#pragma unroll
      for (int j = 0; j < NL; ++j) x.l[j] = src[j];
      row::row_store_from_fe(rec, x);
    }
  }
  __syncthreads();
  uint32_t v = row::rq_identity(S);
#pragma unroll 1
  for (int i = 0; i < FB_WINDOWS; ++i) v = row::rq_add(v, crec + i * row::RQ_WORDS, S, sdig[i] < 0, K);
  xrec[t] = v;
  __syncthreads();
  const ge r = row::rq_load_point(xrec);
  if (ELEMENT) {
    if (t == 0) store_ge_mont256(reinterpret_cast<uint64_t*>(out), e, r);
  } else {
    OneDcbIO io;
    dcb_put(io, 0, ge_dcb_from_half(r, false));
    dcb_finish_with(io, 1, [](const fe& c) { return row::fe_invert_wave(c); });   // (every lane holds the same element)
    if (t == 0) store32(out, e, io.out);
  }
}

// ---- the smallest batches of the square-root family: FOUR elements per wave --------------------------------------------
// A square root is ~300 dependent products (the two fixed exponentiations) and a table phase; with one element per lane a
// call of any size up to the chip's lanes takes as long as that one chain, ~135 us.  Up to four elements per SIMD
// (4 x 4 x the CUs: 4 096 on an MI355X) the power chains run in the lane-spread form instead, one element on each ROW of the
// wave (row_ops.hpp: ~0.2 us per product instead of ~0.33), and everything else -- the table phase, the curve arithmetic
// around the root -- stays whole-element code that lane t runs for element t & 3 (curve.hpp: GivenPowers hands the powers to
// the same functions the other kernels call).  The kernels that end in an encoding invert once per wave, for all four
// elements (row_ops.hpp fe_invert_wave).
constexpr int TINY4 = 4;
struct Tiny4Lds {
  uint32_t tab[POW_TAB * 64];           // the odd powers of the sliding-window exponentiation, one word per lane and entry
  uint32_t xrec[2 * row::RQ_WORDS];     // row records in and out
};
struct Tiny4Lane {
  size_t e;                             // the element this lane works for (clamped to the batch)
  bool own;                             // lanes 0..3 of a wave write, if their element exists
};
__device__ __forceinline__ Tiny4Lane tiny4_lane(size_t n) {
  const size_t e = (size_t)blockIdx.x * TINY4 + (threadIdx.x & 3);
  return Tiny4Lane{e < n ? e : n - 1, threadIdx.x < TINY4 && e < n};
}
// lane t passes the argument(s) of element t & 3's square root (lanes 0..3 are read) and gets that element's powers back
template <bool WITH_NUM>
__device__ __forceinline__ GivenPowers tiny4_powers(Tiny4Lds& L, const fe& num, const fe& den, const row::RowK& K) {
  const int t = threadIdx.x;
  if (t < TINY4) {
    row::row_store_from_fe(L.xrec + 16 * t, den);
    if (WITH_NUM) row::row_store_from_fe(L.xrec + row::RQ_WORDS + 16 * t, num);
  }
  __syncthreads();
  const row::RowPowers pw = WITH_NUM ? row::row_sqrt_powers_num(L.xrec[row::RQ_WORDS + t], L.xrec[t], L.tab, t, K)
                                     : row::row_sqrt_powers(L.xrec[t], L.tab, t, K);
  __syncthreads();
  L.xrec[t] = pw.v; L.xrec[row::RQ_WORDS + t] = pw.uv;
  __syncthreads();
  GivenPowers g;
  g.v = row::row_load_to_fe(L.xrec + 16 * (t & 3));
  g.uv = row::row_load_to_fe(L.xrec + row::RQ_WORDS + 16 * (t & 3));
  __syncthreads();
  return g;
}
// The encodings of the wave's four elements from their compressor states (curve.hpp, "compression without a square root"):
// the product of the four p's by two exchanges inside the quad, one inversion by the wave, each lane's own share back.
__device__ __forceinline__ void tiny4_encode(const dcb_state& st, uint32_t w[8]) {
  const fe b = fe_quad_perm<1, 0, 3, 2>(st.p);
  const fe ab = fe_mul_strict(st.p, b);
  const fe cd = fe_quad_perm<2, 3, 0, 1>(ab);
  const fe inv = row::fe_invert_wave(fe_mul_strict(ab, cd));    // every quad holds the same four: lane 0's product is the wave's
  dcb_encode_one(st, fe_mul(fe_mul(inv, cd), b), w);
}

__global__ void __launch_bounds__(64)
k_sqrt_ratio_zeta_tiny(SqrtTables T, const uint8_t* num32, const uint8_t* den32, size_t n, uint8_t* root32, uint8_t* was_square,
                       int min_curve_root) {
  __shared__ Tiny4Lds L;
  const row::RowK K = row::row_consts();
  const Tiny4Lane me = tiny4_lane(n);
  uint32_t wn[8], wd[8], wr[8];
  load32(num32, me.e, wn);
  load32(den32, me.e, wd);
  const fe num = fe_from_words_mod_order_strict(wn), den = fe_from_words_mod_order_strict(wd);
  GivenPowers pt = tiny4_powers<true>(L, num, den, K);
  fe r;
  const bool ws = fe_sqrt_ratio_zeta<false>(T, pt, num, den, &r, min_curve_root != 0);
  fe_to_bytes_words(r, wr);
  if (me.own) {
    store32(root32, me.e, wr);
    was_square[me.e] = ws ? 1 : 0;
  }
}

// ROUNDTRIP: decompress -> compress (k_roundtrip), two square roots one after the other
template <bool ROUNDTRIP>
__global__ void __launch_bounds__(64)
k_decompress_tiny(SqrtTables T, const uint8_t* enc32, size_t n, uint8_t* out, uint8_t* status) {
  __shared__ Tiny4Lds L;
  const row::RowK K = row::row_consts();
  const Tiny4Lane me = tiny4_lane(n);
  uint32_t w[8];
  load32(enc32, me.e, w);
  GivenPowers pt = tiny4_powers<false>(L, fe_zero(), ge_decompress_den(w), K);
  ge g;
  const uint32_t bad = ge_decompress(T, pt, w, &g);
  if (ROUNDTRIP) {
    pt = tiny4_powers<false>(L, fe_zero(), ge_compress_den(g), K);
    ge_compress(T, pt, g, w, bad == 0 && me.own);           // (the check build counts an element once, not once per lane)
    if (me.own) {
      status[me.e] = (uint8_t)bad;
      if (bad) store32_zero(out, me.e); else store32(out, me.e, w);
    }
  } else if (me.own) {
    status[me.e] = (uint8_t)bad;
    if (bad) {
      store32_zero(out, 4 * me.e); store32_zero(out, 4 * me.e + 1); store32_zero(out, 4 * me.e + 2); store32_zero(out, 4 * me.e + 3);
    } else {
      store_ge_mont256(reinterpret_cast<uint64_t*>(out), me.e, g);
    }
  }
}

__global__ void __launch_bounds__(64) k_compress_tiny(SqrtTables T, const uint64_t* xyzt, size_t n, uint8_t* enc32) {
  __shared__ Tiny4Lds L;
  const row::RowK K = row::row_consts();
  const Tiny4Lane me = tiny4_lane(n);
  const ge p = load_ge_mont256(xyzt, me.e);
  GivenPowers pt = tiny4_powers<false>(L, fe_zero(), ge_compress_den(p), K);
  uint32_t w[8];
  ge_compress(T, pt, p, w, me.own);                           // (the check build counts an element once, not once per lane)
  if (me.own) store32(enc32, me.e, w);
}

// A pair that hits the exceptional case of the quartic's addition law goes the reference's way (as in k_hash_to_curve: whole-
// element square roots, their power table one LDS column per lane of the wave) and enters the encoder as a finished encoding.
// Called by the whole wave when any of its lanes needs it.
__device__ __forceinline__ void tiny_hash_exceptional(SqrtTables T, const uint8_t* r1, const uint8_t* r2, size_t e, bool exceptional, dcb_state& st) {
  __shared__ uint32_t lds_pow_[POW_TAB * NL * 64];
  struct Pow64 {
    uint32_t* col;
    __device__ __forceinline__ void put(int j, const fe& v) { for (int k = 0; k < NL; ++k) col[(j * NL + k) * 64] = v.l[k]; }
    __device__ __forceinline__ fe get(int j) const { fe r; for (int k = 0; k < NL; ++k) r.l[k] = col[(j * NL + k) * 64]; return r; }
  } lp;
  lp.col = lds_pow_ + threadIdx.x;
  uint32_t we[8];
  hash_exceptional_pair(T, lp, r1, r2, e, we);
  const dcb_state se = dcb_from_encoding_words(we);
  st.p = fe_select(exceptional, se.p, st.p); st.w = fe_select(exceptional, se.w, st.w);
  st.n0 = fe_select(exceptional, se.n0, st.n0); st.n1 = fe_select(exceptional, se.n1, st.n1);
}

// r2 null: encode_to_curve; otherwise hash_to_curve (two maps one after the other, the sum on the Jacobi quartic; the
// exceptional pair goes the reference's way as in k_hash_to_curve, with whole-element square roots)
__global__ void __launch_bounds__(64) k_map_to_curve_tiny(SqrtTables T, const uint8_t* r1, const uint8_t* r2, size_t n, uint8_t* out32) {
  __shared__ Tiny4Lds L;
  const row::RowK K = row::row_consts();
  const Tiny4Lane me = tiny4_lane(n);
  uint32_t w[8];
  load32(r1, me.e, w);
  fe r0 = fe_from_words_mod_order(w), s1, t1;
  GivenPowers pt = tiny4_powers<false>(L, fe_zero(), ge_elligator_den(r0), K);
  ge_elligator_st(T, pt, r0, &s1, &t1);
  D377_INVARIANT(T, ge_from_jacobi_st(s1, t1), true);
  dcb_state st;
  if (r2 == nullptr) {
    st = ge_dcb_from_jacobi_st(s1, t1);
  } else {
    fe s2, t2;
    load32(r2, me.e, w);
    r0 = fe_from_words_mod_order(w);
    pt = tiny4_powers<false>(L, fe_zero(), ge_elligator_den(r0), K);
    ge_elligator_st(T, pt, r0, &s2, &t2);
    D377_INVARIANT(T, ge_from_jacobi_st(s2, t2), true);
    bool exceptional;
    st = ge_dcb_from_jacobi_sum(s1, t1, s2, t2, &exceptional);
#if defined(D377_CHECK_INVARIANTS)
    exceptional |= (me.e & 3) == 3;                 // as in k_hash_to_curve: the debug build exercises the exceptional route
#endif
    if (__any(exceptional)) tiny_hash_exceptional(T, r1, r2, me.e, exceptional, st);
  }
  tiny4_encode(st, w);
  if (me.own) store32(out32, me.e, w);
}

// hash_to_curve with TWO pairs per wave (batches up to half the size): the four square roots of the wave are the two maps of
// two pairs, so one pass over the rows serves both maps of a pair (0.235 -> ~0.18 ms per call).  Lane t works for map t & 1
// of pair (t >> 1) & 1; the two lanes of a pair exchange their (s, t) and both form the sum.
__global__ void __launch_bounds__(64) k_hash_to_curve_tiny2(SqrtTables T, const uint8_t* r1, const uint8_t* r2, size_t n, uint8_t* out32) {
  __shared__ Tiny4Lds L;
  const row::RowK K = row::row_consts();
  const int t = threadIdx.x, which = t & 1;
  const size_t e_raw = (size_t)blockIdx.x * 2 + ((t >> 1) & 1);
  const size_t e = e_raw < n ? e_raw : n - 1;
  uint32_t w[8];
  load32(which ? r2 : r1, e, w);
  const fe r0 = fe_from_words_mod_order(w);
  fe s_own, t_own;
  GivenPowers pt = tiny4_powers<false>(L, fe_zero(), ge_elligator_den(r0), K);
  ge_elligator_st(T, pt, r0, &s_own, &t_own);
  D377_INVARIANT(T, ge_from_jacobi_st(s_own, t_own), t < 4);
  const fe s_oth = fe_quad_perm<1, 0, 3, 2>(s_own), t_oth = fe_quad_perm<1, 0, 3, 2>(t_own);
  const bool second = which != 0;                              // both lanes of a pair: (map of r1) + (map of r2), in that order
  bool exceptional;
  dcb_state st = ge_dcb_from_jacobi_sum(fe_select(second, s_oth, s_own), fe_select(second, t_oth, t_own),
                                        fe_select(second, s_own, s_oth), fe_select(second, t_own, t_oth), &exceptional);
#if defined(D377_CHECK_INVARIANTS)
  exceptional |= (e & 3) == 3;                                  // as in k_hash_to_curve: the debug build exercises the exceptional route
#endif
  if (__any(exceptional)) tiny_hash_exceptional(T, r1, r2, e, exceptional, st);
  tiny4_encode(st, w);                                          // (the quad's four values are p_A, p_A, p_B, p_B: any four non-zero values do)
  if (t < 4 && which == 0 && e_raw < n) store32(out32, e, w);
}

// The reference's own signatures for these operations take and return Elements (`Element * Fr`,
// src/min_curve/ops.rs:89-95; `Element::encode_to_curve`, `hash_to_curve`, src/min_curve/element.rs:235-244;
// `vartime_compress_to_field`, :163-181): the same per-lane code as above without the encoding step at either
// end.  An Element leaves as whatever projective representative the schedule here produces -- the group element
// (and so its encoding, and decaf equality) is the reference's; its X:Y:Z:T need not be.
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_scalar_mul_var_el(const uint64_t* xyzt, const uint8_t* scalar32, size_t n,
                                                                             uint64_t* out, uint32_t* scratch, DcbScratch dcb) {
  // chunked like the kernels above (one workgroup per per_lane x 256 elements, a claimed set of window tables)
  const int slot = dcb_claim(dcb);
  if (slot < 0) return;
  GlobalTab tab;
  tab.base = scratch;
  tab.nthreads = (size_t)dcb.nslots * BLOCK;
  tab.tid = (size_t)slot * BLOCK + threadIdx.x;
  const size_t chunk_elems = (size_t)dcb.per_lane * BLOCK;
  for (size_t chunk = blockIdx.x; chunk * chunk_elems < n; chunk += gridDim.x) {
    // a workgroup that walks several chunks (beyond the grid cap) draws a new ticket for each, as dcb_rounds does: the word of
    // a set in use keeps changing however long the launch (d377_ctx_reset_scratch frees only sets whose ticket stood still)
    if (chunk != blockIdx.x && threadIdx.x == 0)
      atomicExch(dcb.pool + slot, (int)(atomicAdd(dcb.health, 1u) & 0x7FFFFFFFu) + 1);
#pragma unroll 1
    for (int j = 0; j < dcb.per_lane; ++j) {
      const size_t i = chunk * chunk_elems + (size_t)j * BLOCK + threadIdx.x;
      if (i >= n) break;
      dcb_progress_priority(dcb.prio, j, dcb.per_lane);  // dcb.hpp: the wave that is behind outranks the one ahead
      uint32_t k[8], dg[8];
      load32(scalar32, i, k);
      const ge g = load_ge_mont256(xyzt, i);
      fr_reduce_words(k);
      fr_recode_signed16(k, dg);
      store_ge_mont256(out, i, ge_scalar_mul_w4(g, dg, tab));
    }
  }
  dcb_release(dcb, slot);
}
template <int BITS>
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_scalar_mul_base_el(const uint32_t* fbase, const uint8_t* scalar32, size_t n,
                                                                              uint64_t* out) {
  FixedTab<BITS> ft{fbase};
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t k[8];
    load32(scalar32, i, k);
    fr_reduce_words(k);
    store_ge_mont256(out, i, ge_scalar_mul_base_w8<BITS>(k, ft));
  }
}
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_compress_to_field(SqrtTables T, const uint64_t* xyzt, size_t n, uint64_t* out) {
  D377_POW_LDS();
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t w[8];
    ge_compress(T, pt, load_ge_mont256(xyzt, i), w);                 // canonical s
    fe_to_mont256_words(fe_from_words_mod_order(w), w);
    store32(reinterpret_cast<uint8_t*>(out), i, w);
  }
}
// second input null: encode_to_curve; otherwise hash_to_curve (two maps and an addition)
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_map_to_element(SqrtTables T, const uint8_t* r1, const uint8_t* r2, size_t n,
                                                                          uint64_t* out, DcbScratch dcb) {
  D377_POW_LDS();
  D377_DCB_BEGIN(out);
  dcb_rounds<2, false>(n, io, pt,
    [&](size_t i, int j) {
      uint32_t w[8];
      load32(r1, i, w);
      dcb_put_den(io, 0, j, ge_elligator_den(fe_from_words_mod_order(w)));
      if (r2) load32(r2, i, w);                            // without a second input slot 1 repeats slot 0 (unused)
      dcb_put_den(io, 1, j, ge_elligator_den(fe_from_words_mod_order(w)));
    },
    [&](size_t i, int j, const uint32_t (*invw)[8], bool have) {
      uint32_t w[8];
      load32(r1, i, w);
      fe inv = fe_from_words(invw[0]);            // meaningless words when !have (never read then)
      fe s1, t1;
      ge_elligator_st(T, pt, fe_from_words_mod_order(w), &s1, &t1, &inv, have);
      ge a;
      if (r2) {                                   // (s, t) of the first map across the second square root, as in k_hash_to_curve
        load32(r2, i, w);
        inv = fe_from_words(invw[1]);
        fe s2, t2;
        ge_elligator_st(T, pt, fe_from_words_mod_order(w), &s2, &t2, &inv, have);
          a = ge_add(ge_from_jacobi_st(s1, t1), ge_from_jacobi_st(s2, t2));
      } else {
        a = ge_from_jacobi_st(s1, t1);
      }
      D377_INVARIANT(T, a, true);
      store_ge_mont256(out, i, a);
    });
  D377_DCB_END();
}

// wide byte strings (48 or 64 bytes per record) -> Fq, optionally straight into the Elligator map
__device__ __forceinline__ void load_wide_words(const uint8_t* in, size_t i, int len, uint32_t lo[8], uint32_t hi[8]) {
  const uint4* p = reinterpret_cast<const uint4*>(in + (size_t)len * i);
  uint4 a = p[0], b = p[1], c = p[2];
  uint4 d = make_uint4(0, 0, 0, 0);
  if (len == 64) d = p[3];
  lo[0] = a.x; lo[1] = a.y; lo[2] = a.z; lo[3] = a.w; lo[4] = b.x; lo[5] = b.y; lo[6] = b.z; lo[7] = b.w;
  hi[0] = c.x; hi[1] = c.y; hi[2] = c.z; hi[3] = c.w; hi[4] = d.x; hi[5] = d.y; hi[6] = d.z; hi[7] = d.w;
}
__device__ __forceinline__ fe load_wide(const uint8_t* in, size_t i, int len) {
  uint32_t lo[8], hi[8];
  load_wide_words(in, i, len, lo, hi);
  return fe_from_wide_words(lo, hi);
}
__global__ void __launch_bounds__(BLOCK) k_fq_from_wide(const uint8_t* in, int len, size_t n, uint8_t* out32) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t w[8];
    fe_to_bytes_words(load_wide(in, i, len), w);
    store32(out32, i, w);
  }
}
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_encode_to_curve_wide(SqrtTables T, const uint8_t* in, int len,
                                                                                size_t n, uint8_t* out32, DcbScratch dcb) {
  D377_POW_LDS();
  D377_DCB_BEGIN(out32);
  dcb_rounds<1, true>(n, io, pt,
    [&](size_t i, int j) { dcb_put_den(io, 0, j, ge_elligator_den(fe_carry(load_wide(in, i, len)))); },
    [&](size_t i, int j, const uint32_t (*invw)[8], bool have) {
      fe inv = fe_from_words(invw[0]);            // meaningless words when !have (never read then)
      fe s, t;
      ge_elligator_st(T, pt, fe_carry(load_wide(in, i, len)), &s, &t, &inv, have);
      D377_INVARIANT(T, ge_from_jacobi_st(s, t), true);
      dcb_put(io, j, ge_dcb_from_jacobi_st(s, t));
    });
  D377_DCB_END();
}
// (x/z, y/z) as Montgomery-256 limbs: CurveGroup::normalize_batch (src/ark_curve/element.rs:74-81), with
// the batched inversion that name implies (Montgomery's trick).  A lane walks its grid-stride elements
// twice: forward it multiplies the z's up, parking each exclusive prefix product in the element's own
// 64-byte output slot; then ONE inversion (x^(q-2), ~380 field operations) of the lane's total; backward
// it peels 1/z_i = inverse * prefix_i, inverse *= z_i.  About 12 products per element plus the inversion
// shared by the lane's ~32 elements, against one inversion per element before.  A zero z (not a group
// element; Fq::inverse returns None there) yields a zero record and is kept out of the product.
constexpr int AFFINE_PER_LANE = 32;
// (The records are used as they lie in memory -- curve.hpp, "normalize_batch on raw records": 5 products per element
// instead of 12, 0.62 -> see profiles/README.md.)
__global__ void __launch_bounds__(BLOCK) k_to_affine(const uint64_t* xyzt, size_t n, uint64_t* xy) {
  const size_t T = (size_t)gridDim.x * BLOCK, t = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  const bool live = t < n;                                  // (a lane without elements still takes part in the wave's inversion)
  const uint8_t* b = reinterpret_cast<const uint8_t*>(xyzt);
  uint8_t* o = reinterpret_cast<uint8_t*>(xy);
  uint32_t* slots = reinterpret_cast<uint32_t*>(xy);       // 16 words per element
  // Both walks are one dependent chain of products per lane with a handful of waves per SIMD: the next element's words
  // are requested before the current element's products start, so the chain does not wait for memory at every step.
  fe p = fe_const(FE_ONE);
  size_t last = t;
  uint32_t zn[8] = {};
  if (live) load32(b, 4 * t + 2, zn);
  for (size_t i = t; i < n; i += T) {
    uint32_t zw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) zw[k] = zn[k];
    if (i + T < n) load32(b, 4 * (i + T) + 2, zn);
    bool zz;
    const fe z = affine_raw_z(zw, &zz);
    slot_store(slots + 16 * i, p);                          // product of this lane's earlier z's
    p = fe_mul(p, z);
    last = i;
  }
  fe inv = fe_mul(row::fe_invert_lanes(p), fe_const(FE_TO_MONT256));   // one inversion per wave (row_ops.hpp)
  if (!live) return;
  uint32_t xn[8], yn[8];
  load32(b, 4 * last + 2, zn);
  load32(b, 4 * last + 0, xn);
  load32(b, 4 * last + 1, yn);
  fe pn = slot_load(slots + 16 * last);
  for (size_t i = last;; i -= T) {
    uint32_t zw[8], xw[8], yw[8], w[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) { zw[k] = zn[k]; xw[k] = xn[k]; yw[k] = yn[k]; }
    const fe pre = pn;
    if (i != t) {                                           // the element below, while this one is worked on
      const size_t j = i - T;
      load32(b, 4 * j + 2, zn);
      load32(b, 4 * j + 0, xn);
      load32(b, 4 * j + 1, yn);
      pn = slot_load(slots + 16 * j);
    }
    bool zz;
    const fe z = affine_raw_z(zw, &zz);
    const fe zi = fe_mul(inv, pre);                         // (scaled) 1 / z_i
    inv = fe_mul(inv, z);
    affine_raw_finish(zi, xw, yw, zz, w);
    store32(o, 2 * i, w);
    store32(o, 2 * i + 1, w + 8);
    if (i == t) break;
  }
}

// Fq operations on Montgomery-256 records (src/fields/fq/u64/wrapper.rs:99-132)
__global__ void __launch_bounds__(BLOCK) k_fq_op(int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out,
                                                 uint8_t* status) {
  const uint8_t* ab = reinterpret_cast<const uint8_t*>(a);
  const uint8_t* bb = reinterpret_cast<const uint8_t*>(b);
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t w[8];
    load32(ab, i, w);
    uint32_t v[8], o[8];
    for (int k = 0; k < 8; ++k) v[k] = 0;
    if (op <= D377_FQ_MUL) load32(bb, i, v);
    uint32_t st = 0;
    bool done = false;
    // products: raw limbs carry 2^-5 each (curve.hpp, "records used without conversion"); sums on the words themselves
    if (op == D377_FQ_MUL || op == D377_FQ_SQUARE) {
      const fe x = fe_from_words(w);
      fe_scaled_to_mont256_words(op == D377_FQ_MUL ? fe_mul(x, fe_from_words(v)) : fe_sqr(x), FE_RAW2_TO_MONT256, o);
      done = true;
    } else if (op == D377_FQ_ADD || op == D377_FQ_SUB) {
      done = fq_addsub_words(w, v, op == D377_FQ_SUB, o);
    } else if (op == D377_FQ_NEG) {
      done = fq_neg_words(w, o);
    }
    if (!done) {                                          // inverse, or a non-canonical operand: the long way
      const fe x = fe_from_mont256_words(w);
      fe r;
      switch (op) {
        case D377_FQ_ADD: r = fe_carry(fe_add(x, fe_from_mont256_words(v))); break;
        case D377_FQ_SUB: r = fe_sub(x, fe_from_mont256_words(v)); break;
        case D377_FQ_NEG: r = fe_neg(x); break;
        default: r = fe_invert(x); st = fe_is_zero(x) ? 1u : 0u; break;   // 0^(q-2) = 0: zero record, status 1
      }
      fe_to_mont256_words(r, o);
    }
    store32(reinterpret_cast<uint8_t*>(out), i, o);
    if (status) status[i] = (uint8_t)st;
  }
}
// Fq::inverse on a batch (src/fields/fq/u64/wrapper.rs:104-112): ONE inversion per wave and trip (row_ops.hpp fe_invert_lanes:
// 12 products and a share of the wave's inversion per element instead of the lane's own ~26 000 instructions of divsteps).
// A zero has no inverse: zero record, status 1; it enters the wave's product as 1.
__global__ void __launch_bounds__(BLOCK) k_fq_inv(const uint64_t* a, size_t n, uint64_t* out, uint8_t* status) {
  const uint8_t* ab = reinterpret_cast<const uint8_t*>(a);
  const size_t stride = (size_t)gridDim.x * BLOCK;
  for (size_t base = (size_t)blockIdx.x * BLOCK; base < n; base += stride) {     // (uniform per workgroup: every lane is in the wave's inversion)
    const size_t i = base + threadIdx.x;
    const bool live = i < n;
    uint32_t w[8] = {}, o[8];
    if (live) load32(ab, i, w);
    const fe x = fe_from_mont256_words(w);
    const bool zero = !live || fe_is_zero(x);
    const fe r = row::fe_invert_lanes(fe_select(zero, fe_const(FE_ONE), x));
    fe_to_mont256_words(fe_select(zero, fe_zero(), r), o);
    if (live) {
      store32(reinterpret_cast<uint8_t*>(out), i, o);
      if (status) status[i] = zero ? 1 : 0;
    }
  }
}
__global__ void __launch_bounds__(BLOCK) k_fq_from_bytes_checked(const uint8_t* in, size_t n, uint64_t* out, uint8_t* status) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t w[8];
    load32(in, i, w);
    const bool bad = words_geq(w, FQ_MODULUS_W_LIT);
    const fe x = fe_from_words_mod_order(w);
    fe_to_mont256_words(x, w);
    if (bad) store32_zero(reinterpret_cast<uint8_t*>(out), i); else store32(reinterpret_cast<uint8_t*>(out), i, w);
    status[i] = bad ? 1 : 0;
  }
}
__global__ void __launch_bounds__(BLOCK) k_fq_to_bytes(const uint64_t* a, size_t n, uint8_t* out) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t w[8];
    load32(reinterpret_cast<const uint8_t*>(a), i, w);
    fe_to_bytes_words(fe_from_mont256_words(w), w);
    store32(out, i, w);
  }
}
// Fr::from_le_bytes_mod_order / from_bytes_checked on 32-byte strings (src/fields/fr.rs:82-107)
__global__ void __launch_bounds__(BLOCK) k_fr_bytes(const uint8_t* in, size_t n, uint8_t* out, uint8_t* status) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t k[8];
    load32(in, i, k);
    if (status) {                                   // checked: copy through when canonical
      const bool bad = words_geq(k, FR_ORDER_W_LIT);
      status[i] = bad ? 1 : 0;
      if (bad) store32_zero(out, i); else store32(out, i, k);
    } else {
      fr_reduce_words(k);
      store32(out, i, k);
    }
  }
}
// Fr arithmetic on 32-byte little-endian records (any value: reduced mod r first, as from_le_bytes_mod_order does)
__global__ void __launch_bounds__(BLOCK) k_fr_op(int op, const uint8_t* a, const uint8_t* b, size_t n, uint8_t* out, uint8_t* status) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t x[8], y[8], r[8];
    load32(a, i, x);
    fr_reduce_words(x);
    if (op <= D377_FQ_MUL) { load32(b, i, y); fr_reduce_words(y); }
    uint32_t st = 0;
    switch (op) {
      case D377_FQ_ADD: fr_addmod(x, y, r); break;
      case D377_FQ_SUB: fr_submod(x, y, r); break;
      case D377_FQ_MUL: fr_mulmod(x, y, r); break;
      case D377_FQ_SQUARE: fr_mulmod(x, x, r); break;
      case D377_FQ_NEG: {
        for (int k = 0; k < 8; ++k) y[k] = 0;
        fr_submod(y, x, r);
        break;
      }
      default: st = fr_invmod(x, r) ? 0u : 1u; break;               // zero record, status 1 for x = 0
    }
    store32(out, i, r);
    if (status) status[i] = (uint8_t)st;
  }
}
__global__ void __launch_bounds__(BLOCK) k_fr_from_wide(const uint8_t* in, int len, size_t n, uint8_t* out32) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t lo[8], hi[8], r[8];
    load_wide_words(in, i, len, lo, hi);
    fr_from_wide_words(lo, hi, r);
    store32(out32, i, r);
  }
}
__global__ void __launch_bounds__(BLOCK) k_neg(const uint64_t* p, size_t n, uint64_t* out) {
  D377_RECORD_TILES(rec0) {
    uint32_t a[32], r[32];
    wave_load_records128(p, rec0, n, tile, lane, a);
    if (!ge_neg_words(a, r)) {                            // a non-canonical record: reduce it the long way
      ge g;
      g.x = fe_from_mont256_words(a); g.y = fe_from_mont256_words(a + 8); g.z = fe_from_mont256_words(a + 16); g.t = fe_from_mont256_words(a + 24);
      g = ge_neg(g);
      fe_to_mont256_words(g.x, r); fe_to_mont256_words(g.y, r + 8); fe_to_mont256_words(g.z, r + 16); fe_to_mont256_words(g.t, r + 24);
    }
    wave_store_records128(out, rec0, n, tile, lane, r);
  }
}
// Element::is_identity: x == 0 (src/min_curve/element.rs:113-117)
__global__ void __launch_bounds__(BLOCK) k_is_identity(const uint64_t* p, size_t n, uint8_t* out) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK)
    out[i] = fe_is_zero(load_ge_mont256(p, i).x) ? 1 : 0;
}
// add / double / eq / neg read the records without converting them (curve.hpp, "records used without conversion"):
// these kernels move 256-384 bytes per element and the eight conversion products were most of their time.
// negate != 0: Element - Element = self + other.neg() (src/min_curve/ops.rs:43-49)
// k_add and k_neg move their records through LDS, coalesced (device_util.hpp, wave_load_records128): same box, 2^22 records,
// neg 0.25 -> 0.20 ms, add 0.49-0.52 -> 0.46 ms; the doubling (one input stream, ~2 500 instructions per element) is bound by
// its instructions and stays on plain loads (0.32 ms either way).  The reference's exact coordinates cost 11 products for
// an addition and 10 for a doubling (curve.hpp, ge_raw_efgh_to_words; 13 and 12 with one scaling product per coordinate:
// 0.53 and 0.43 ms).
__global__ void __launch_bounds__(BLOCK) k_add(const uint64_t* p, const uint64_t* q, size_t n, uint64_t* out, int negate) {
  D377_RECORD_TILES(rec0) {
    uint32_t a[32], b[32], r[32];
    wave_load_records128(p, rec0, n, tile, lane, a);
    wave_load_records128(q, rec0, n, tile, lane, b);
    ge_add_raw_words(a, b, negate != 0, r);
    wave_store_records128(out, rec0, n, tile, lane, r);
  }
}
__global__ void __launch_bounds__(BLOCK) k_double(const uint64_t* p, size_t n, uint64_t* out) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t a[32], r[32];
    load_record128(p, i, a);
    ge_double_raw_words(a, r);
    store_record128(out, i, r);
  }
}
// decaf equality: x1 * y2 == x2 * y1  (src/min_curve/element.rs:334-340)
__global__ void __launch_bounds__(BLOCK) k_eq(const uint64_t* p, const uint64_t* q, size_t n, uint8_t* eq) {
  const uint8_t* pb = reinterpret_cast<const uint8_t*>(p);
  const uint8_t* qb = reinterpret_cast<const uint8_t*>(q);
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint32_t a[32], b[32];                               // only X and Y are read
    load32(pb, 4 * i + 0, a); load32(pb, 4 * i + 1, a + 8);
    load32(qb, 4 * i + 0, b); load32(qb, 4 * i + 1, b + 8);
    eq[i] = ge_eq_raw_words(a, b) ? 1 : 0;
  }
}

// d377_ctx_reset_scratch: frees the lane sets whose tickets the host found unchanged (leaked by a launch that died)
__global__ void k_pool_release(int* pool, const int* idx, const int* ticket, int m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) atomicCAS(&pool[idx[i]], ticket[i], 0);
}
constexpr int RESET_WAIT_MS = 1500;
constexpr int RESET_MIN_MS = 250;        // between the two reads of the pool: many chunks of any kernel (the longest, 8 variable-base elements per lane, is ~15 ms)

// ------------------------------------------------------------------------------ host side ---
}  // namespace

int d377::debug_device_delay_ms() {
  const char* e = getenv("D377_DEBUG_DEVICE_DELAY_MS");
  return e ? atoi(e) : 0;
}

namespace {

// Batches up to this many elements take the quad-per-element kernel (k_scalar_mul_var_small).  One wave of 16 quads per
// SIMD is 16 x 4 x CUs elements (16 384); the kernel still wins a little beyond that: the Element form (20 KiB of LDS per
// wave: seven waves per CU) runs 28 672 elements in one generation, two waves sharing most SIMDs -- 0.60 ms against 0.88
// with one lane per element -- and the Encoding form (39 KiB: four waves per CU) two generations of 16 384 in 1.02 ms
// against 1.08.  Beyond those sizes one lane per element is the better use of the chip (measured, profiles/README.md).
// D377_TUNE_SMALL_MAX: developer override (0 switches the small-batch kernel off).
size_t small_batch_max(const DeviceState& d, bool element_form) {
  return (size_t)d.tuned(D377_TUNE_SMALL_MAX, (long long)d.cus * SMALL_QUADS * (element_form ? 7 : 8));
}

// Batches up to this many elements take one WAVE per element (k_scalar_mul_var_tiny): one wave per SIMD.  Capped by
// small_batch_max, so that switching the small-batch kernels off switches this one off too.  D377_TUNE_TINY_MAX: developer override.
size_t tiny_batch_max(const DeviceState& d) {
  const size_t v = (size_t)d.tuned(D377_TUNE_TINY_MAX, (long long)d.cus * 4), cap = small_batch_max(d, true);
  return v < cap ? v : cap;
}

// ... and the square-root family's batches of up to FOUR elements per wave (k_*_tiny above)
size_t tiny4_batch_max(const DeviceState& d) { return tiny_batch_max(d) * TINY4; }
unsigned tiny4_grid(size_t n) { return (unsigned)((n + TINY4 - 1) / TINY4); }

int grid_for(const DeviceState& d, size_t n) {
  // >> 256 workgroups when the batch allows it; capped so huge batches grid-stride
  size_t blocks = (n + BLOCK - 1) / BLOCK;
  size_t cap = (size_t)d.cus * 32;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

int init_tables(DeviceState& d, uint32_t* keys, int* coll) {
  hipLaunchKernelGGL(k_init_gtab, dim3(6), dim3(BLOCK), 0, d.stream, d.gtab);
  hipLaunchKernelGGL(k_init_slookup, dim3(1), dim3(BLOCK), 0, d.stream, d.s_lookup, keys, coll);
  HIP_TRY(hipGetLastError());
  int h_coll = -1;
  HIP_TRY(hipMemcpyAsync(&h_coll, coll, sizeof(int), hipMemcpyDeviceToHost, d.stream));
  HIP_TRY(hipStreamSynchronize(d.stream));
  if (h_coll != 0) return fail(D377_ERR_INIT, "s_lookup perfect hash self-check failed (%s)", "collisions");
  return D377_OK;
}

// The lane-set areas have WAVES_PER_SIMD sets per CU.  Ask the runtime how many workgroups of each chunked kernel fit
// on a CU (registers, the kernel's own LDS); where that is more than the sets, pad the launch with dynamic LDS until it
// is not (160 KiB per CU: a pad of a little over 160 / (sets + 1) KiB admits `sets` workgroups and no more).  A kernel
// that still exceeds the sets would only spin for a free set (dcb_claim), silently slower: refuse to start instead.
// The comb widths a context can ask for at run time (d377_ctx_create_ex), plus whatever width the library was built with:
// f(std::integral_constant<int, BITS>) runs with the kernels of that width.
template <class F>
int with_fb_bits(int bits, F&& f) {
  switch (bits) {
    case 18: return f(std::integral_constant<int, 18>{});
    case 21: return f(std::integral_constant<int, 21>{});
    case 23: return f(std::integral_constant<int, 23>{});
  }
  if (bits == FB_BITS) return f(std::integral_constant<int, FB_BITS>{});
  return fail(D377_ERR_ARG, "%s", "comb width: 18, 21 or 23 bits (or 0 for the library's default)");
}

// residency of the fixed-base kernel of the context's comb width: its wide launch (FB_SETS lane sets per CU) and its narrow one
int check_residency_fb(DeviceState& d, const void* fn) {
  const bool verbose = getenv("D377_DEBUG_RESIDENCY") != nullptr;
  {
    int nb = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, BLOCK, 0));
    d.chunk_lds[CK_MUL_BASE] = 0;
    const int sets = d.chunk_sets[CK_MUL_BASE];
    const int pad = (160 * 1024) / (sets + 1) + 1024;
    if (nb > sets) {
      if (pad > 64 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, pad));
      d.chunk_lds[CK_MUL_BASE] = pad;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, BLOCK, (size_t)pad));
    }
    if (verbose) fprintf(stderr, "d377: k_scalar_mul_base<%d>: %d workgroups per CU with %d bytes of LDS padding\n", d.fb_bits, nb, d.chunk_lds[CK_MUL_BASE]);
    d.chunk_blocks[CK_MUL_BASE] = nb;
    if (nb < 1 || nb > sets)
      return fail(D377_ERR_INIT, "residency of %s does not match the lane sets of the scratch areas", "k_scalar_mul_base");
  }
  // the narrow launch (WAVES_PER_SIMD workgroups per CU, up to FB_WIDE_GENERATIONS generation of full chunks): its own padding
  {
    const int pad = (160 * 1024) / (WAVES_PER_SIMD + 1) + 1024;
    int nb = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, BLOCK, (size_t)pad));
    if (verbose) fprintf(stderr, "d377: k_scalar_mul_base<%d>, narrow launch: %d workgroups per CU with %d bytes of LDS padding\n", d.fb_bits, nb, pad);
    if (nb < 1 || nb > WAVES_PER_SIMD)
      return fail(D377_ERR_INIT, "residency of %s does not match the lane sets of the scratch areas", "k_scalar_mul_base (narrow launch)");
    d.fb_narrow_lds = pad;
    if (pad > d.chunk_lds[CK_MUL_BASE] && pad > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, pad));
  }
  return D377_OK;
}

// The fixed-base comb of the context's width: allocated and built on the context's stream the first time it is asked for --
// by d377_ctx_create[_ex] for an eager context, by the first fixed-base call of a lazy one (d377_ctx_opts::comb_lazy: a
// caller that never multiplies by the generator never pays the table, 5.9 GB at 23 bits) -- and complete when this returns.
// `s`: the stream of the call that needs it; a capturing stream cannot allocate or synchronise, so a lazy context's FIRST
// fixed-base call must not be inside a capture (as for an MSM that would have to grow its workspace).  Caller holds ctx->mu.
int ensure_comb(DeviceState& d, hipStream_t s) {
  if (d.fbase) return D377_OK;
  if (s && ScratchGuard::capturing(s))
    return fail(D377_ERR_ARG, "%s", "the fixed-base comb of a lazy context is built by its first fixed-base call, which cannot be captured into a graph: make one eager call first");
  return with_fb_bits(d.fb_bits, [&](auto bits_c) -> int {
    constexpr int BITS = decltype(bits_c)::value;
    using Sh = FbShape<BITS>;
    int rc = check_residency_fb(d, reinterpret_cast<const void*>(k_scalar_mul_base<BITS>));
    if (rc) return rc;
    const size_t bytes = (size_t)Sh::windows * Sh::entries * FBW_ENTRY_WORDS * sizeof(uint32_t);
    uint32_t* fb = nullptr;
    if (hipMalloc(&fb, bytes) != hipSuccess) {
      (void)hipGetLastError();
      snprintf(d377_g_err, sizeof d377_g_err,
               "the %d-bit fixed-base comb needs %.2f GB of device memory and the allocation failed (device %d): "
               "d377_ctx_create_ex with comb_bits = 18 (0.24 GB) or 21 (1.6 GB), or free memory", BITS, (double)bytes / 1e9, d.id);
      return D377_ERR_HIP;
    }
    if (!d.fb_bases && hipMalloc(&d.fb_bases, (size_t)FbShape<8>::windows * 4 * SLOT * sizeof(uint32_t)) != hipSuccess) {
      (void)hipFree(fb);
      return fail(D377_ERR_HIP, "%s", "hipMalloc failed (comb window bases)");
    }
    hipLaunchKernelGGL(k_init_fbase_bases<BITS>, dim3(1), dim3(64), 0, d.stream, d.fb_bases);
    const size_t runs = (size_t)Sh::windows * ((Sh::entries + FB_RUN - 1) / FB_RUN);
    hipLaunchKernelGGL(k_init_fbase<BITS>, dim3((unsigned)((runs + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, d.stream, d.fb_bases, fb);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(d.stream);
    if (e != hipSuccess) { (void)hipFree(fb); return fail(D377_ERR_HIP, "building the fixed-base comb: %s", hipGetErrorString(e)); }
    d.fbase = fb;
    return D377_OK;
  });
}

int check_residency(DeviceState& d) {
  const void* fns[CK_COUNT];
  fns[CK_SQRT] = reinterpret_cast<const void*>(k_sqrt_ratio_zeta);
  fns[CK_ENCODE] = reinterpret_cast<const void*>(k_encode_to_curve);
  fns[CK_HASH] = reinterpret_cast<const void*>(k_hash_to_curve);
  fns[CK_MUL_VAR] = reinterpret_cast<const void*>(k_scalar_mul_var);
  fns[CK_MUL_BASE] = nullptr;                                  // per comb width: check_residency_fb, when the comb is built
  fns[CK_MUL_VAR_EL] = reinterpret_cast<const void*>(k_scalar_mul_var_el);
  fns[CK_MAP_EL] = reinterpret_cast<const void*>(k_map_to_element);
  fns[CK_ENCODE_WIDE] = reinterpret_cast<const void*>(k_encode_to_curve_wide);
  fns[CK_DECOMPRESS] = reinterpret_cast<const void*>(k_decompress_chunked);
  static const char* names[CK_COUNT] = {"k_sqrt_ratio_zeta", "k_encode_to_curve", "k_hash_to_curve", "k_scalar_mul_var",
                                        "k_scalar_mul_base", "k_scalar_mul_var_el", "k_map_to_element", "k_encode_to_curve_wide",
                                        "k_decompress_chunked"};
  const bool verbose = getenv("D377_DEBUG_RESIDENCY") != nullptr;
  for (int k = 0; k < CK_COUNT; ++k) {
    if (!fns[k]) continue;
    int nb = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fns[k], BLOCK, 0));
    d.chunk_lds[k] = 0;
    const int sets = d.chunk_sets[k];
    const int pad = (160 * 1024) / (sets + 1) + 1024;
    if (nb > sets) {
      if (pad > 64 * 1024) HIP_TRY(hipFuncSetAttribute(fns[k], hipFuncAttributeMaxDynamicSharedMemorySize, pad));
      d.chunk_lds[k] = pad;
      const int before = nb;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fns[k], BLOCK, (size_t)pad));
      if (verbose) fprintf(stderr, "d377: %s: %d workgroups per CU by registers / own LDS, %d with %d bytes of LDS padding\n", names[k], before, nb, pad);
    } else if (verbose) {
      fprintf(stderr, "d377: %s: %d workgroups per CU\n", names[k], nb);
    }
    d.chunk_blocks[k] = nb;
    if (nb < 1 || nb > sets)
      return fail(D377_ERR_INIT, "residency of %s does not match the lane sets of the scratch areas", names[k]);
  }
  return D377_OK;
}

int init_device(DeviceState& d) {
  HIP_TRY(hipSetDevice(d.id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, d.id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(D377_ERR_NO_DEVICE, "device is %s, this library is built for gfx950 only", prop.gcnArchName);
  d.cus = prop.multiProcessorCount;
  HIP_TRY(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&d.copy_stream, hipStreamNonBlocking));
  {
    // The stream of d377_ctx_health / d377_ctx_reset_scratch, at the highest priority: the runtime keeps a pool of hardware
    // queues per priority and lets streams share them when a process has more streams than queues (4), and a packet behind
    // a starving kernel in the SAME hardware queue waits for it -- a health call on an ordinary stream blocked for the
    // kernel's whole 10 s whenever the process held a second context (seen: 5 streams on 4 queues).  Only these short
    // control operations ever run at this priority.
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(&d.ctl_stream, hipStreamNonBlocking, greatest));
  }
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipEventCreateWithFlags(&d.ev_in[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d.ev_done[i], hipEventDisableTiming));
  }
  HIP_TRY(hipEventCreateWithFlags(&d.ev_shard, hipEventDisableTiming));
  int rc;
  if ((rc = d.vb_guard.init()) || (rc = d.msm.guard.init())) return rc;
  HIP_TRY(hipMalloc(&d.gtab, (size_t)6 * 256 * GT_STRIDE * sizeof(uint32_t)));
  HIP_TRY(hipMalloc(&d.s_lookup, (size_t)1 << S_HASH_BITS));
  HIP_TRY(hipMalloc(&d.inv_fail, sizeof(uint32_t)));
  HIP_TRY(hipMemsetAsync(d.inv_fail, 0, sizeof(uint32_t), d.stream));
  // variable-base window tables: one per resident lane, fixed grid, grid-stride over the batch
  d.vb_blocks = d.cus * WAVES_PER_SIMD;        // exactly the resident blocks: 2 per CU

  HIP_TRY(hipMalloc(&d.vb_scratch, (size_t)d.vb_blocks * BLOCK * VB_ENTRIES * VB_ENTRY_WORDS * sizeof(uint32_t)));
  // round records of the batched inversions: DCB_SLOTS x DCB_KMAX 32-byte records per lane of every lane set (480 MiB), and the
  // pool of lane sets the workgroups claim
  for (int k = 0; k < CK_COUNT; ++k) { d.chunk_sets[k] = WAVES_PER_SIMD; d.chunk_k[k] = DCB_K; }
  d.chunk_sets[CK_MUL_BASE] = FB_SETS;
  d.chunk_k[CK_MUL_BASE] = FB_K;
  d.dcb_sets = d.cus * DCB_SETS_MAX;
  HIP_TRY(hipMalloc(&d.dcb_scratch, (size_t)d.dcb_sets * BLOCK * DCB_SLOTS * DCB_KMAX * 32));
  HIP_TRY(hipMalloc(&d.slot_pool, (size_t)d.dcb_sets * sizeof(int)));
  HIP_TRY(hipMemsetAsync(d.slot_pool, 0, (size_t)d.dcb_sets * sizeof(int), d.stream));      // every set free; workgroups free what they claim
  HIP_TRY(hipMalloc(&d.pool_health, 4 * sizeof(uint32_t)));
  HIP_TRY(hipMemsetAsync(d.pool_health, 0, 4 * sizeof(uint32_t), d.stream));
  HIP_TRY(hipHostMalloc(&d.starve_host, 2 * sizeof(uint32_t), hipHostMallocDefault));
  d.starve_host[0] = d.starve_host[1] = 0;
  // pinned landing area of d377_ctx_health / d377_ctx_reset_scratch (pool words + the four health words): a copy into
  // pageable memory may wait for work of other streams inside the runtime, and these calls must not wait for kernels
  HIP_TRY(hipHostMalloc(&d.pool_host, ((size_t)d.dcb_sets + 4) * sizeof(int), hipHostMallocDefault));
  if ((rc = check_residency(d))) return rc;
  uint32_t* keys = nullptr;
  int* coll = nullptr;
  HIP_TRY(hipMalloc(&keys, 512 * sizeof(uint32_t)));
  if (hipMalloc(&coll, sizeof(int)) != hipSuccess) { (void)hipFree(keys); return fail(D377_ERR_HIP, "%s", "hipMalloc failed"); }
  rc = init_tables(d, keys, coll);
  (void)hipFree(keys);
  (void)hipFree(coll);
  if (rc == D377_OK && !d.fb_lazy) rc = ensure_comb(d, nullptr);
  return rc;
}

void free_device(DeviceState& d) {
  if (d.id < 0) return;
  (void)hipSetDevice(d.id);
  if (d.stream) (void)hipStreamSynchronize(d.stream);
  if (d.copy_stream) (void)hipStreamSynchronize(d.copy_stream);
  (void)d.vb_guard.drain();
  (void)d.msm.guard.drain();
  (void)hipFree(d.gtab); (void)hipFree(d.s_lookup); (void)hipFree(d.fbase); (void)hipFree(d.fb_bases); (void)hipFree(d.bm_scratch); (void)hipFree(d.vb_scratch); (void)hipFree(d.dcb_scratch); (void)hipFree(d.slot_pool); (void)hipFree(d.pool_health); (void)hipFree(d.inv_fail);
  if (d.starve_host) (void)hipHostFree(d.starve_host);
  d.starve_host = nullptr;
  if (d.pool_host) (void)hipHostFree(d.pool_host);
  d.pool_host = nullptr;
  for (int i = 0; i < 4; ++i) { (void)hipFree(d.buf[i]); (void)hipFree(d.buf2[i]); (void)hipFree(d.shard[i]); }
  for (int i = 0; i < 2; ++i) {
    if (d.ev_in[i]) (void)hipEventDestroy(d.ev_in[i]);
    if (d.ev_done[i]) (void)hipEventDestroy(d.ev_done[i]);
  }
  if (d.ev_shard) (void)hipEventDestroy(d.ev_shard);
  d.vb_guard.destroy();
  d.msm.guard.destroy();
  if (d.copy_stream) (void)hipStreamDestroy(d.copy_stream);
  if (d.ctl_stream) (void)hipStreamDestroy(d.ctl_stream);
  (void)hipFree(d.msm.mem);
  for (uint8_t* r : d.msm.retired) (void)hipFree(r);
  d.msm.retired.clear();
  if (d.stream) (void)hipStreamDestroy(d.stream);
}

// launches one op on device buffers; in0/in1 inputs, out0/out1 outputs (unused ones null).
// aux: the D377_FQ_* selector of OP_FQ_BIN / OP_FQ_UN / OP_FR_BIN / OP_FR_UN, the D377_SQRT_ROOT_* convention of OP_SQRT.
// The caller holds ctx->mu (the scratch guards are host state).
int launch(DeviceState& d, hipStream_t s, Op op, int aux, const void* in0, const void* in1, size_t n, void* out0, void* out1) {
  if (n == 0) return D377_OK;
  const SqrtTables T = d.tables();
  const int g = grid_for(d, n);
  // kernels that work in chunks of DCB_K x 256 elements: one workgroup per chunk (oversubscribed on purpose, see
  // DcbScratch), each claiming one of the vb_blocks resident lane sets of the per-device scratch areas
  // DCB_K elements per lane when the batch is large enough to fill the resident lane sets that way, fewer otherwise
  int gv = 0;
  DcbScratch dcb{};
  // The batch in rounds of BLOCK elements over the `places` workgroups that are resident at once (cus x sets).  Up to
  // kmax rounds per place everything runs in ONE generation, and the rounds are dealt out as evenly as they go: every
  // workgroup takes rounds / places of them and the first rounds % places workgroups one more (DcbScratch::extra).
  // (Chunks of ceil(n / resident lanes) elements per lane for everybody left a batch just above k x the resident lanes
  // with fewer workgroups of k + 1 rounds each: pairs of them shared a CU while other CUs held one, and the call took
  // as long as k + 1 full rounds -- profiles/r04_size_sweep.txt.)  Beyond that: the same deal over the fewest generations
  // that hold the rounds (host_state.hpp deal_chunks), as many workgroups as chunks (oversubscribed on purpose, see DcbScratch).
  auto chunks_of = [&](int sets, int kmax, int& grid, DcbScratch& sc) {
    const size_t places = (size_t)d.cus * (size_t)sets;
    const size_t rounds = (n + BLOCK - 1) / BLOCK;
    size_t per_lane, extra = 0, nchunks;
    if (d.is_tuned(D377_TUNE_CHUNK_PER_LANE)) {                             // developer override (sweeps, route tests): uniform chunks
      per_lane = (size_t)d.tuned(D377_TUNE_CHUNK_PER_LANE, 1);
      if (per_lane > (size_t)kmax) per_lane = (size_t)kmax;
      nchunks = (rounds + per_lane - 1) / per_lane;
    } else {
      const ChunkDeal c = deal_chunks(rounds, places, (size_t)kmax, (size_t)d.cus * 64);      // host_state.hpp
      per_lane = c.per_lane; extra = c.extra; nchunks = c.nchunks;
    }
    if (nchunks > (size_t)d.cus * 64) nchunks = (size_t)d.cus * 64;
    grid = (int)nchunks;
#ifdef D377_DCB_TICKETS                   // A/B only (dcb.hpp): the resident workgroups take the chunks by ticket
    if ((size_t)grid > places) grid = (int)places;
#endif
    sc = DcbScratch{d.dcb_scratch, d.slot_pool, d.cus * sets, (int)per_lane, d.dcb_sets * BLOCK, (int)extra, d.pool_health};
    // Issue priority by progress (dcb.hpp dcb_progress_priority) for launches of one or two generations of workgroups: there the
    // workgroups of a CU would end one after the other (-4 to -8 % at 2^20, -1 to -3 % at 1.5 and 2 x 2^20).  Longer launches
    // refill their CUs and only their last generation has that tail: within +-1 % either way at 2^22, so they stay as they
    // were measured (profiles/r05_ab_progress_priority.txt).
    sc.prio = nchunks <= 2 * places ? 1 : 0;
  };
  chunks_of(WAVES_PER_SIMD, DCB_K_LONG, gv, dcb);  // every chunked kernel but the fixed-base one (up to 2^20 on 256 CUs: <= DCB_K per lane either way)
#ifdef D377_DCB_TICKETS
  HIP_TRY(hipMemsetAsync(d.pool_health + 3, 0, sizeof(uint32_t), s));      // the launch's chunk counter
#endif
  GuardScope vb{d.vb_guard, s};            // released (event recorded) when this function returns, if it was acquired
  int rc;
  switch (op) {
    // Every kernel with a square root or an encoding keeps per-lane state in scratch areas that exist once per device
    // (round records of the batched inversions, window tables): never more lanes than those areas have (gv), and each
    // launch queues behind the areas' last user.
    case OP_SQRT:
      if (n <= tiny4_batch_max(d)) {                          // four elements per wave: power chains on the rows, no scratch
        hipLaunchKernelGGL(k_sqrt_ratio_zeta_tiny, dim3(tiny4_grid(n)), dim3(64), 0, s, T, (const uint8_t*)in0, (const uint8_t*)in1, n,
                           (uint8_t*)out0, (uint8_t*)out1, aux);
        break;
      }
      if ((rc = vb.acquire())) return rc;
      hipLaunchKernelGGL(k_sqrt_ratio_zeta, dim3(gv), dim3(BLOCK), d.chunk_lds[CK_SQRT], s, T, (const uint8_t*)in0, (const uint8_t*)in1, n,
                         (uint8_t*)out0, (uint8_t*)out1, aux, dcb);
      break;
    case OP_DECOMPRESS: {
      // from DCB_ASSIST_MIN elements per resident lane: the batched-inverse form (2^19 0.97 against 1.02 ms, 2^20 1.89 / 2.04,
      // 2^21 3.79 / 4.06, 2^22 7.66 / 8.10); below, the wide grid
      if (n <= tiny4_batch_max(d)) {
        hipLaunchKernelGGL(k_decompress_tiny<false>, dim3(tiny4_grid(n)), dim3(64), 0, s, T, (const uint8_t*)in0, n, (uint8_t*)out0, (uint8_t*)out1);
        break;
      }
      const size_t chunked_min = (size_t)d.tuned(D377_TUNE_DECOMPRESS_CHUNKED_MIN,
                                                 (long long)(d.resident_lanes() * CODEC_CHUNKED_MIN));
      if (n >= chunked_min) {
        if ((rc = vb.acquire())) return rc;
        hipLaunchKernelGGL(k_decompress_chunked, dim3(gv), dim3(BLOCK), d.chunk_lds[CK_DECOMPRESS], s, T, (const uint8_t*)in0, n, (uint64_t*)out0,
                           (uint8_t*)out1, dcb);
        break;
      }
      hipLaunchKernelGGL(k_decompress, dim3(g), dim3(BLOCK), 0, s, T, (const uint8_t*)in0, n, (uint64_t*)out0, (uint8_t*)out1);
      break;
    }
    case OP_COMPRESS:
      if (n <= tiny4_batch_max(d)) {
        hipLaunchKernelGGL(k_compress_tiny, dim3(tiny4_grid(n)), dim3(64), 0, s, T, (const uint64_t*)in0, n, (uint8_t*)out0);
        break;
      }
      if (n >= (size_t)d.tuned(D377_TUNE_DECOMPRESS_CHUNKED_MIN, (long long)(d.resident_lanes() * CODEC_CHUNKED_MIN)) &&   // as OP_DECOMPRESS
          codec_chunked_ok(d)) {                                                  // codec_chunked.hip
        if ((rc = vb.acquire())) return rc;
        if ((rc = codec_chunked_launch(d, s, false, in0, n, out0, nullptr, gv, dcb))) return rc;
        break;
      }
      hipLaunchKernelGGL(k_compress, dim3(g), dim3(BLOCK), 0, s, T, (const uint64_t*)in0, n, (uint8_t*)out0);
      break;
    case OP_ROUNDTRIP:
      if (n <= tiny4_batch_max(d)) {
        hipLaunchKernelGGL(k_decompress_tiny<true>, dim3(tiny4_grid(n)), dim3(64), 0, s, T, (const uint8_t*)in0, n, (uint8_t*)out0, (uint8_t*)out1);
        break;
      }
      if (n >= (size_t)d.tuned(D377_TUNE_DECOMPRESS_CHUNKED_MIN, (long long)(d.resident_lanes() * CODEC_CHUNKED_MIN)) &&   // as OP_DECOMPRESS
          codec_chunked_ok(d)) {
        if ((rc = vb.acquire())) return rc;
        if ((rc = codec_chunked_launch(d, s, true, in0, n, out0, out1, gv, dcb))) return rc;
        break;
      }
      hipLaunchKernelGGL(k_roundtrip, dim3(g), dim3(BLOCK), 0, s, T, (const uint8_t*)in0, n, (uint8_t*)out0, (uint8_t*)out1);
      break;
    case OP_MUL_BASE: {
      // Wide launch (3 workgroups per CU, 16 elements per inversion) beyond ONE generation of full narrow chunks (DCB_K elements per
      // resident lane: 2^20 on 256 CUs); up to there the narrow one (2 per CU, 8 per inversion), whose rounds are dealt out
      // evenly, is as fast or faster (196 608 elements: 166 against 186 us).  Beyond, the narrow launch needs a second
      // generation of workgroups and the wide one does not: wide / narrow 0.88 at 1.25 x 2^20, 0.95 at 1.5 x, 0.88 at
      // 1.75 x, 0.96 at 2^21, 0.94-0.95 at 3 x 2^20 and 2^22, warm clocks, alternating (profiles/r05_fb_wide_sweep.txt; the
      // threshold of rounds 3-4, two generations, was measured on the 18-bit comb before the rounds were dealt out evenly).
      if ((rc = ensure_comb(d, s))) return rc;                 // (a lazy context's first fixed-base call builds the table)
      if (n <= tiny_batch_max(d)) {                           // one scalar per wave, lane-spread arithmetic
        if ((rc = with_fb_bits(d.fb_bits, [&](auto b) -> int {
               hipLaunchKernelGGL((k_scalar_mul_base_tiny<false, decltype(b)::value>), dim3((unsigned)n), dim3(64), 0, s, d.fbase, (const uint8_t*)in0, n, (uint8_t*)out0);
               return D377_OK; }))) return rc;
        break;
      }
      if (n <= small_batch_max(d, false)) {                   // one scalar per quad of lanes
        if ((rc = with_fb_bits(d.fb_bits, [&](auto b) -> int {
               hipLaunchKernelGGL((k_scalar_mul_base_small<false, decltype(b)::value>), dim3((unsigned)((n + SMALL_QUADS - 1) / SMALL_QUADS)), dim3(SMALL_THREADS), 0, s,
                                  d.fbase, (const uint8_t*)in0, n, (uint8_t*)out0);
               return D377_OK; }))) return rc;
        break;
      }
      bool wide = n > d.resident_lanes() * DCB_K * FB_WIDE_GENERATIONS;
      if (d.is_tuned(D377_TUNE_FB_WIDE)) wide = d.tuned(D377_TUNE_FB_WIDE, 0) != 0;               // developer overrides (A/B)
      const int fk = (int)d.tuned(D377_TUNE_FB_K, wide ? FB_K : DCB_K);
      int gb;
      DcbScratch db;
      chunks_of(wide ? FB_SETS : WAVES_PER_SIMD, fk, gb, db);
      // dcb.hpp dcb_progress_priority in the wide launch: one generation only (-1 % at 1.25 x 2^20, -2 % at 2^21); two generations
      // of waves in step gather their table entries in bursts (+6 to +10 % at 2^22: profiles/r05_ab_progress_priority.txt)
      if (wide && gb > d.cus * FB_SETS) db.prio = 0;
      if ((rc = vb.acquire())) return rc;
      if ((rc = with_fb_bits(d.fb_bits, [&](auto b) -> int {
             hipLaunchKernelGGL((k_scalar_mul_base<decltype(b)::value>), dim3(gb), dim3(BLOCK), wide ? d.chunk_lds[CK_MUL_BASE] : d.fb_narrow_lds, s, T, d.fbase,
                                (const uint8_t*)in0, n, (uint8_t*)out0, db);
             return D377_OK; }))) return rc;
      break;
    }
    case OP_MUL_VAR:
      if (n <= tiny_batch_max(d)) {                           // one element per wave, lane-spread arithmetic
        hipLaunchKernelGGL(k_scalar_mul_var_tiny<false>, dim3((unsigned)n), dim3(64), 0, s, T, (const uint8_t*)in0, (const uint8_t*)in1, n,
                           (uint8_t*)out0, (uint8_t*)out1);
        break;
      }
      if (n <= small_batch_max(d, false)) {                   // one element per quad of lanes, table in LDS: no scratch, no hand-over
        hipLaunchKernelGGL(k_scalar_mul_var_small<false>, dim3((unsigned)((n + SMALL_QUADS - 1) / SMALL_QUADS)), dim3(SMALL_THREADS), 0, s, T,
                           (const uint8_t*)in0, (const uint8_t*)in1, n, (uint8_t*)out0, (uint8_t*)out1);
        break;
      }
      if ((rc = vb.acquire())) return rc;
      hipLaunchKernelGGL(k_scalar_mul_var, dim3(gv), dim3(BLOCK), d.chunk_lds[CK_MUL_VAR], s, T, (const uint8_t*)in0, (const uint8_t*)in1, n,
                         (uint8_t*)out0, (uint8_t*)out1, d.vb_scratch, dcb);
      break;
    case OP_ENCODE:
      if (n <= tiny4_batch_max(d)) {
        hipLaunchKernelGGL(k_map_to_curve_tiny, dim3(tiny4_grid(n)), dim3(64), 0, s, T, (const uint8_t*)in0, (const uint8_t*)nullptr, n, (uint8_t*)out0);
        break;
      }
      if ((rc = vb.acquire())) return rc;
      hipLaunchKernelGGL(k_encode_to_curve, dim3(gv), dim3(BLOCK), d.chunk_lds[CK_ENCODE], s, T, (const uint8_t*)in0, n, (uint8_t*)out0, dcb);
      break;
    case OP_HASH:
      if (n <= tiny4_batch_max(d) / 2) {                      // two pairs per wave: one pass over the rows for both maps
        hipLaunchKernelGGL(k_hash_to_curve_tiny2, dim3((unsigned)((n + 1) / 2)), dim3(64), 0, s, T, (const uint8_t*)in0, (const uint8_t*)in1, n, (uint8_t*)out0);
        break;
      }
      if (n <= tiny4_batch_max(d)) {
        hipLaunchKernelGGL(k_map_to_curve_tiny, dim3(tiny4_grid(n)), dim3(64), 0, s, T, (const uint8_t*)in0, (const uint8_t*)in1, n, (uint8_t*)out0);
        break;
      }
      if ((rc = vb.acquire())) return rc;
      hipLaunchKernelGGL(k_hash_to_curve, dim3(gv), dim3(BLOCK), d.chunk_lds[CK_HASH], s, T, (const uint8_t*)in0, (const uint8_t*)in1, n,
                         (uint8_t*)out0, dcb);
      break;
    case OP_ADD:
      hipLaunchKernelGGL(k_add, dim3(g), dim3(BLOCK), 0, s, (const uint64_t*)in0, (const uint64_t*)in1, n, (uint64_t*)out0, aux);
      break;
    case OP_DOUBLE:
      hipLaunchKernelGGL(k_double, dim3(g), dim3(BLOCK), 0, s, (const uint64_t*)in0, n, (uint64_t*)out0);
      break;
    case OP_EQ:
      hipLaunchKernelGGL(k_eq, dim3(g), dim3(BLOCK), 0, s, (const uint64_t*)in0, (const uint64_t*)in1, n, (uint8_t*)out0);
      break;
    case OP_WIDE48:
    case OP_WIDE64:
      hipLaunchKernelGGL(k_fq_from_wide, dim3(g), dim3(BLOCK), 0, s, (const uint8_t*)in0, op == OP_WIDE48 ? 48 : 64, n,
                         (uint8_t*)out0);
      break;
    case OP_ENCODE_WIDE48:
    case OP_ENCODE_WIDE64:
      if ((rc = vb.acquire())) return rc;
      hipLaunchKernelGGL(k_encode_to_curve_wide, dim3(gv), dim3(BLOCK), d.chunk_lds[CK_ENCODE_WIDE], s, T, (const uint8_t*)in0,
                         op == OP_ENCODE_WIDE48 ? 48 : 64, n, (uint8_t*)out0, dcb);
      break;
    case OP_AFFINE: {
      // ~AFFINE_PER_LANE elements per lane so that one inversion serves many, but never fewer lanes than one
      // wave per SIMD.  Measured (tools/attic/affine_bench.py, 2^20 elements): 0.28 / 0.40 / 0.65 / 1.14 ms at 1 / 2 / 4 / 8
      // blocks per CU -- every extra lane is an extra 380-operation inversion -- against 2.7 ms for one inversion
      // per element.  D377_TUNE_AFFINE_BLOCKS_PER_CU is a developer override for that sweep.
      size_t lanes = (n + AFFINE_PER_LANE - 1) / AFFINE_PER_LANE;
      const size_t fill = (size_t)d.cus * (size_t)d.tuned(D377_TUNE_AFFINE_BLOCKS_PER_CU, 1) * BLOCK;
      if (lanes < fill) lanes = fill < n ? fill : n;
      const int ga = (int)((lanes + BLOCK - 1) / BLOCK);
      hipLaunchKernelGGL(k_to_affine, dim3(ga), dim3(BLOCK), 0, s, (const uint64_t*)in0, n, (uint64_t*)out0);
      break;
    }
    case OP_FQ_BIN:
    case OP_FQ_UN:
      if (op == OP_FQ_UN && aux == D377_FQ_INVERSE)
        hipLaunchKernelGGL(k_fq_inv, dim3(g), dim3(BLOCK), 0, s, (const uint64_t*)in0, n, (uint64_t*)out0, (uint8_t*)out1);
      else
        hipLaunchKernelGGL(k_fq_op, dim3(g), dim3(BLOCK), 0, s, aux, (const uint64_t*)in0, (const uint64_t*)in1, n,
                           (uint64_t*)out0, (uint8_t*)out1);
      break;
    case OP_FQ_CHECKED:
      hipLaunchKernelGGL(k_fq_from_bytes_checked, dim3(g), dim3(BLOCK), 0, s, (const uint8_t*)in0, n, (uint64_t*)out0, (uint8_t*)out1);
      break;
    case OP_FQ_TO_BYTES:
      hipLaunchKernelGGL(k_fq_to_bytes, dim3(g), dim3(BLOCK), 0, s, (const uint64_t*)in0, n, (uint8_t*)out0);
      break;
    case OP_FR_MOD:
      hipLaunchKernelGGL(k_fr_bytes, dim3(g), dim3(BLOCK), 0, s, (const uint8_t*)in0, n, (uint8_t*)out0, (uint8_t*)nullptr);
      break;
    case OP_FR_CHECKED:
      hipLaunchKernelGGL(k_fr_bytes, dim3(g), dim3(BLOCK), 0, s, (const uint8_t*)in0, n, (uint8_t*)out0, (uint8_t*)out1);
      break;
    case OP_NEG:
      hipLaunchKernelGGL(k_neg, dim3(g), dim3(BLOCK), 0, s, (const uint64_t*)in0, n, (uint64_t*)out0);
      break;
    case OP_MUL_VAR_EL: {
      if (n <= tiny_batch_max(d)) {
        hipLaunchKernelGGL(k_scalar_mul_var_tiny<true>, dim3((unsigned)n), dim3(64), 0, s, T, (const uint8_t*)in0, (const uint8_t*)in1, n,
                           (uint8_t*)out0, (uint8_t*)nullptr);
        break;
      }
      if (n <= small_batch_max(d, true)) {
        hipLaunchKernelGGL(k_scalar_mul_var_small<true>, dim3((unsigned)((n + SMALL_QUADS - 1) / SMALL_QUADS)), dim3(SMALL_THREADS), 0, s, T,
                           (const uint8_t*)in0, (const uint8_t*)in1, n, (uint8_t*)out0, (uint8_t*)nullptr);
        break;
      }
      // no inversions here, so nothing argues for long chunks: one or two elements per lane keep the grid oversubscribed
      // at every batch size
      // (2^20 elements: 7.33e7/s with 8 per lane, 7.44 with 4, 7.59 with 2, 7.54 with 1)
      DcbScratch dv = dcb;
      dv.extra = 0;                                          // this kernel walks uniform chunks
      dv.per_lane = n > d.resident_lanes() ? 2 : 1;
      size_t nch = (n + (size_t)dv.per_lane * BLOCK - 1) / ((size_t)dv.per_lane * BLOCK);
      if (nch > (size_t)d.cus * 64) nch = (size_t)d.cus * 64;
      // `dcb` carries the priority rule of ITS deal (chunks of up to 16 per lane); this launch has its own shape -- 8 192 chunks
      // in 16 generations at 2^22 -- and the rule is about generations: priorities in launches of one or two of them
      dv.prio = nch <= 2 * (size_t)d.cus * WAVES_PER_SIMD ? 1 : 0;
      if ((rc = vb.acquire())) return rc;
      hipLaunchKernelGGL(k_scalar_mul_var_el, dim3((int)nch), dim3(BLOCK), d.chunk_lds[CK_MUL_VAR_EL], s, (const uint64_t*)in0, (const uint8_t*)in1, n,
                         (uint64_t*)out0, d.vb_scratch, dv);
      break;
    }
    case OP_MUL_BASE_EL:
      if ((rc = ensure_comb(d, s))) return rc;
      if ((rc = with_fb_bits(d.fb_bits, [&](auto b) -> int {
             constexpr int BITS = decltype(b)::value;
             if (n <= tiny_batch_max(d))
               hipLaunchKernelGGL((k_scalar_mul_base_tiny<true, BITS>), dim3((unsigned)n), dim3(64), 0, s, d.fbase, (const uint8_t*)in0, n, (uint8_t*)out0);
             else if (n <= small_batch_max(d, true))
               hipLaunchKernelGGL((k_scalar_mul_base_small<true, BITS>), dim3((unsigned)((n + SMALL_QUADS - 1) / SMALL_QUADS)), dim3(SMALL_THREADS), 0, s, d.fbase,
                                  (const uint8_t*)in0, n, (uint8_t*)out0);
             else
               hipLaunchKernelGGL((k_scalar_mul_base_el<BITS>), dim3(g), dim3(BLOCK), 0, s, d.fbase, (const uint8_t*)in0, n, (uint64_t*)out0);
             return D377_OK; }))) return rc;
      break;
    case OP_COMPRESS_FIELD:
      hipLaunchKernelGGL(k_compress_to_field, dim3(g), dim3(BLOCK), 0, s, T, (const uint64_t*)in0, n, (uint64_t*)out0);
      break;
    case OP_ENCODE_EL:
    case OP_HASH_EL:
      if ((rc = vb.acquire())) return rc;
      hipLaunchKernelGGL(k_map_to_element, dim3(gv), dim3(BLOCK), d.chunk_lds[CK_MAP_EL], s, T, (const uint8_t*)in0,
                         op == OP_HASH_EL ? (const uint8_t*)in1 : (const uint8_t*)nullptr, n, (uint64_t*)out0, dcb);
      break;
    case OP_FR_BIN:
    case OP_FR_UN:
      hipLaunchKernelGGL(k_fr_op, dim3(g), dim3(BLOCK), 0, s, aux, (const uint8_t*)in0, (const uint8_t*)in1, n, (uint8_t*)out0,
                         (uint8_t*)out1);
      break;
    case OP_FR_WIDE48:
    case OP_FR_WIDE64:
      hipLaunchKernelGGL(k_fr_from_wide, dim3(g), dim3(BLOCK), 0, s, (const uint8_t*)in0, op == OP_FR_WIDE48 ? 48 : 64, n,
                         (uint8_t*)out0);
      break;
    case OP_IS_IDENTITY:
      hipLaunchKernelGGL(k_is_identity, dim3(g), dim3(BLOCK), 0, s, (const uint64_t*)in0, n, (uint8_t*)out0);
      break;
  }
  HIP_TRY(hipGetLastError());
  return vb.finish();
}

struct OpShape { size_t in0, in1, out0, out1; };   // bytes per element
OpShape shape_of(Op op) {
  switch (op) {
    case OP_SQRT: return {32, 32, 32, 1};
    case OP_DECOMPRESS: return {32, 0, 128, 1};
    case OP_COMPRESS: return {128, 0, 32, 0};
    case OP_ROUNDTRIP: return {32, 0, 32, 1};
    case OP_MUL_BASE: return {32, 0, 32, 0};
    case OP_MUL_VAR: return {32, 32, 32, 1};
    case OP_ENCODE: return {32, 0, 32, 0};
    case OP_HASH: return {32, 32, 32, 0};
    case OP_ADD: return {128, 128, 128, 0};
    case OP_DOUBLE: return {128, 0, 128, 0};
    case OP_EQ: return {128, 128, 1, 0};
    case OP_WIDE48: return {48, 0, 32, 0};
    case OP_WIDE64: return {64, 0, 32, 0};
    case OP_ENCODE_WIDE48: return {48, 0, 32, 0};
    case OP_ENCODE_WIDE64: return {64, 0, 32, 0};
    case OP_AFFINE: return {128, 0, 64, 0};
    case OP_FQ_BIN: return {32, 32, 32, 1};
    case OP_FQ_UN: return {32, 0, 32, 1};
    case OP_FQ_CHECKED: return {32, 0, 32, 1};
    case OP_FQ_TO_BYTES: return {32, 0, 32, 0};
    case OP_FR_MOD: return {32, 0, 32, 0};
    case OP_FR_CHECKED: return {32, 0, 32, 1};
    case OP_NEG: return {128, 0, 128, 0};
    case OP_MUL_VAR_EL: return {128, 32, 128, 0};
    case OP_MUL_BASE_EL: return {32, 0, 128, 0};
    case OP_COMPRESS_FIELD: return {128, 0, 32, 0};
    case OP_ENCODE_EL: return {32, 0, 128, 0};
    case OP_HASH_EL: return {32, 32, 128, 0};
    case OP_FR_BIN: return {32, 32, 32, 1};
    case OP_FR_UN: return {32, 0, 32, 1};
    case OP_FR_WIDE48: return {48, 0, 32, 0};
    case OP_FR_WIDE64: return {64, 0, 32, 0};
    case OP_IS_IDENTITY: return {128, 0, 1, 0};
  }
  return {0, 0, 0, 0};
}

// Large batches are pipelined in chunks of 2^18 records: while chunk k runs on the compute stream,
// chunk k+1's inputs are copied in and chunk k-1's outputs are copied out on the copy stream
// (double-buffered staging), so PCIe time hides under the kernels.
constexpr size_t PIPE_CHUNK = (size_t)1 << 18;

int run_one_pipelined(DeviceState& d, Op op, int aux, const OpShape& sh, const void* in0, const void* in1, size_t n,
                      void* out0, void* out1) {
  int rc = D377_OK;
  SyncOnError guard{&rc, d.id, d.stream, d.copy_stream};
  auto body = [&]() -> int {
    int r;
    if ((r = ensure(d, 0, PIPE_CHUNK * sh.in0))) return r;
    if (sh.in1 && (r = ensure(d, 1, PIPE_CHUNK * sh.in1))) return r;
    if ((r = ensure(d, 2, PIPE_CHUNK * sh.out0))) return r;
    if (sh.out1 && (r = ensure(d, 3, PIPE_CHUNK * sh.out1))) return r;
    if ((r = ensure2(d, 0, PIPE_CHUNK * sh.in0))) return r;
    if (sh.in1 && (r = ensure2(d, 1, PIPE_CHUNK * sh.in1))) return r;
    if ((r = ensure2(d, 2, PIPE_CHUNK * sh.out0))) return r;
    if (sh.out1 && (r = ensure2(d, 3, PIPE_CHUNK * sh.out1))) return r;
    const size_t nchunks = (n + PIPE_CHUNK - 1) / PIPE_CHUNK;
    StarveCheck starve{d, d.stream};
    if ((r = starve.before())) return r;
    auto bufs = [&](size_t k) -> uint8_t** { return (k & 1) ? d.buf2 : d.buf; };
    auto drain = [&](size_t k) -> int {          // outputs of chunk k -> host (waits for its kernel)
      const size_t lo = k * PIPE_CHUNK, cnt = (lo + PIPE_CHUNK <= n) ? PIPE_CHUNK : n - lo;
      uint8_t** b = bufs(k);
      HIP_TRY(hipStreamWaitEvent(d.copy_stream, d.ev_done[k & 1], 0));
      HIP_TRY(hipMemcpyAsync((uint8_t*)out0 + lo * sh.out0, b[2], cnt * sh.out0, hipMemcpyDeviceToHost, d.copy_stream));
      if (sh.out1)
        HIP_TRY(hipMemcpyAsync((uint8_t*)out1 + lo * sh.out1, b[3], cnt * sh.out1, hipMemcpyDeviceToHost, d.copy_stream));
      HIP_TRY(hipStreamSynchronize(d.copy_stream));
      return D377_OK;
    };
    for (size_t k = 0; k < nchunks; ++k) {
      const size_t lo = k * PIPE_CHUNK, cnt = (lo + PIPE_CHUNK <= n) ? PIPE_CHUNK : n - lo;
      uint8_t** b = bufs(k);
      // this buffer set was last used by chunk k-2, which has been drained (host-synchronised) already
      HIP_TRY(hipMemcpyAsync(b[0], (const uint8_t*)in0 + lo * sh.in0, cnt * sh.in0, hipMemcpyHostToDevice, d.copy_stream));
      if (sh.in1)
        HIP_TRY(hipMemcpyAsync(b[1], (const uint8_t*)in1 + lo * sh.in1, cnt * sh.in1, hipMemcpyHostToDevice, d.copy_stream));
      HIP_TRY(hipEventRecord(d.ev_in[k & 1], d.copy_stream));
      HIP_TRY(hipStreamWaitEvent(d.stream, d.ev_in[k & 1], 0));
      if ((r = launch(d, d.stream, op, aux, b[0], b[1], cnt, b[2], b[3]))) return r;
      HIP_TRY(hipEventRecord(d.ev_done[k & 1], d.stream));
      if (k >= 1 && (r = drain(k - 1))) return r;
    }
    if ((r = starve.after())) return r;
    if ((r = drain(nchunks - 1))) return r;
    HIP_TRY(hipStreamSynchronize(d.stream));
    return starve.verdict();
  };
  rc = body();
  return rc;
}

// one device, one contiguous slice of a host batch: copies in, kernel, copies out, synchronised
int run_one(DeviceState& d, Op op, int aux, const OpShape& sh, const void* in0, const void* in1, size_t n, void* out0,
            void* out1) {
  if (n == 0) return D377_OK;
  HIP_TRY(hipSetDevice(d.id));
  if (n >= 2 * PIPE_CHUNK) return run_one_pipelined(d, op, aux, sh, in0, in1, n, out0, out1);
  int rc = D377_OK;
  SyncOnError guard{&rc, d.id, d.stream, nullptr};
  auto body = [&]() -> int {
    int r;
    if ((r = ensure(d, 0, n * sh.in0))) return r;
    if (sh.in1 && (r = ensure(d, 1, n * sh.in1))) return r;
    if ((r = ensure(d, 2, n * sh.out0))) return r;
    if (sh.out1 && (r = ensure(d, 3, n * sh.out1))) return r;
    StarveCheck starve{d, d.stream};
    if ((r = starve.before())) return r;
    HIP_TRY(hipMemcpyAsync(d.buf[0], in0, n * sh.in0, hipMemcpyHostToDevice, d.stream));
    if (sh.in1) HIP_TRY(hipMemcpyAsync(d.buf[1], in1, n * sh.in1, hipMemcpyHostToDevice, d.stream));
    if ((r = launch(d, d.stream, op, aux, d.buf[0], d.buf[1], n, d.buf[2], d.buf[3]))) return r;
    HIP_TRY(hipMemcpyAsync(out0, d.buf[2], n * sh.out0, hipMemcpyDeviceToHost, d.stream));
    if (sh.out1) HIP_TRY(hipMemcpyAsync(out1, d.buf[3], n * sh.out1, hipMemcpyDeviceToHost, d.stream));
    if ((r = starve.after())) return r;
    HIP_TRY(hipStreamSynchronize(d.stream));
    return starve.verdict();
  };
  rc = body();
  return rc;
}

// host-pointer path: contiguous slices over the context's devices.  With several devices each slice is
// driven by its own host thread: copies from pageable caller memory block the issuing thread, so a single
// thread would run the devices one after another.
int run_host(d377_ctx* ctx, Op op, int aux, const void* in0, const void* in1, size_t n, void* out0, void* out1) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  const OpShape sh = shape_of(op);
  if (n && (!in0 || (sh.in1 && !in1) || !out0 || (sh.out1 && !out1))) return fail(D377_ERR_ARG, "%s", "null buffer");
  if (n == 0) return D377_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  const size_t nd = ctx->devs.size();
  if (nd == 1) return run_one(ctx->devs[0], op, aux, sh, in0, in1, n, out0, out1);
  const size_t per = (n + nd - 1) / nd;
  std::vector<int> rcs(nd, D377_OK);
  std::vector<std::string> errs(nd);
  std::vector<std::thread> workers;
  const int delay = debug_device_delay_ms();
  for (size_t k = 0; k < nd; ++k) {
    const size_t lo = per * k;
    if (lo >= n) break;
    const size_t cnt = (lo + per <= n) ? per : n - lo;
    workers.emplace_back([&, k, lo, cnt]() {
      if (delay > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay));
      rcs[k] = run_one(ctx->devs[k], op, aux, sh, (const uint8_t*)in0 + lo * sh.in0,
                       sh.in1 ? (const uint8_t*)in1 + lo * sh.in1 : nullptr, cnt, (uint8_t*)out0 + lo * sh.out0,
                       sh.out1 ? (uint8_t*)out1 + lo * sh.out1 : nullptr);
      if (rcs[k] != D377_OK) errs[k] = d377_g_err;       // the worker's thread-local text
    });
  }
  for (auto& w : workers) w.join();                      // every device has drained before we return, error or not
  for (size_t k = 0; k < nd; ++k)
    if (rcs[k] != D377_OK) return fail(rcs[k], "%s", errs[k].c_str());
  return D377_OK;
}

int check_dev_args(d377_ctx* ctx, int dev, Op op, const void* in0, const void* in1, size_t n, const void* out0,
                   const void* out1) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  if (dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "device index out of range");
  const OpShape sh = shape_of(op);
  const bool out1_optional = (op == OP_FQ_BIN || op == OP_FQ_UN || op == OP_FR_BIN || op == OP_FR_UN);
  if (n && (!in0 || (sh.in1 && !in1) || !out0 || (sh.out1 && !out1 && !out1_optional))) return fail(D377_ERR_ARG, "%s", "null buffer");
  if (!aligned16(in0) || !aligned16(in1) || (sh.out0 >= 16 && !aligned16(out0)))
    return fail(D377_ERR_ARG, "%s", "device record buffers must be 16-byte aligned");
  return D377_OK;
}

int run_dev(d377_ctx* ctx, int dev, void* stream, Op op, int aux, const void* in0, const void* in1, size_t n, void* out0,
            void* out1) {
  int rc = check_dev_args(ctx, dev, op, in0, in1, n, out0, out1);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceState& d = ctx->devs[(size_t)dev];
  HIP_TRY(hipSetDevice(d.id));
  return launch(d, (hipStream_t)stream, op, aux, in0, in1, n, out0, out1);
}

// A batch that lives in the HBM of device `root`: the other devices of the context get contiguous
// slices by peer copies over xGMI, run the same kernel, and copy their outputs back; the root's own
// slice runs in place.  Nothing synchronises with the host: completion is ordered on `stream`.
int run_sharded_dev(d377_ctx* ctx, int root, void* stream, Op op, int aux, const void* in0, const void* in1, size_t n,
                    void* out0, void* out1) {
  int rc = check_dev_args(ctx, root, op, in0, in1, n, out0, out1);
  if (rc) return rc;
  if (n == 0) return D377_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  const OpShape sh = shape_of(op);
  const size_t nd = ctx->devs.size();
  DeviceState& R = ctx->devs[(size_t)root];
  hipStream_t s = (hipStream_t)stream;
  // slices are multiples of 16 records so every slice of a 1-byte-per-record array stays 16-byte aligned
  size_t per = (n + nd - 1) / nd;
  per = (per + 15) & ~(size_t)15;
  HIP_TRY(hipSetDevice(R.id));
  HIP_TRY(hipEventRecord(R.ev_shard, s));                 // the inputs are ready at this point of `stream`
  std::vector<size_t> used;
  // An error after work has been enqueued on other devices' streams: nothing may still be writing into the caller's
  // output buffers (or reading its inputs) once we return, so drain every stream touched so far and `stream` itself.
  struct DrainOnError {
    d377_ctx* ctx; std::vector<size_t>& used; size_t* cur; int root; hipStream_t s; int* rc;
    ~DrainOnError() {
      if (*rc == D377_OK) return;
      char saved[sizeof d377_g_err];
      memcpy(saved, d377_g_err, sizeof saved);
      std::vector<size_t> all(used);
      if (*cur != (size_t)-1) all.push_back(*cur);
      for (size_t k : all) { (void)hipSetDevice(ctx->devs[k].id); (void)hipStreamSynchronize(ctx->devs[k].stream); }
      (void)hipSetDevice(ctx->devs[(size_t)root].id);
      (void)hipStreamSynchronize(s);
      memcpy(d377_g_err, saved, sizeof saved);
    }
  };
  size_t cur_dev = (size_t)-1;
  int rc_final = D377_OK;
  DrainOnError drain{ctx, used, &cur_dev, root, s, &rc_final};
  auto body = [&]() -> int {
  for (size_t k = 0; k < nd; ++k) {
    const size_t lo = per * k;
    if (lo >= n) break;
    const size_t cnt = (lo + per <= n) ? per : n - lo;
    if ((int)k == root) continue;
    DeviceState& d = ctx->devs[k];
    cur_dev = k;
    HIP_TRY(hipSetDevice(d.id));
    if ((rc = ensure_shard(d, 0, cnt * sh.in0))) return rc;
    if (sh.in1 && (rc = ensure_shard(d, 1, cnt * sh.in1))) return rc;
    if ((rc = ensure_shard(d, 2, cnt * sh.out0))) return rc;
    if (sh.out1 && (rc = ensure_shard(d, 3, cnt * sh.out1))) return rc;
    HIP_TRY(hipStreamWaitEvent(d.stream, R.ev_shard, 0));
    HIP_TRY(hipMemcpyPeerAsync(d.shard[0], d.id, (const uint8_t*)in0 + lo * sh.in0, R.id, cnt * sh.in0, d.stream));
    if (sh.in1) HIP_TRY(hipMemcpyPeerAsync(d.shard[1], d.id, (const uint8_t*)in1 + lo * sh.in1, R.id, cnt * sh.in1, d.stream));
    if ((rc = launch(d, d.stream, op, aux, d.shard[0], d.shard[1], cnt, d.shard[2], d.shard[3]))) return rc;
    HIP_TRY(hipMemcpyPeerAsync((uint8_t*)out0 + lo * sh.out0, R.id, d.shard[2], d.id, cnt * sh.out0, d.stream));
    if (sh.out1 && out1) HIP_TRY(hipMemcpyPeerAsync((uint8_t*)out1 + lo * sh.out1, R.id, d.shard[3], d.id, cnt * sh.out1, d.stream));
    HIP_TRY(hipEventRecord(d.ev_shard, d.stream));
    used.push_back(k);
    cur_dev = (size_t)-1;
  }
  HIP_TRY(hipSetDevice(R.id));
  {
    const size_t lo = per * (size_t)root;
    if (lo < n) {
      const size_t cnt = (lo + per <= n) ? per : n - lo;
      if ((rc = launch(R, s, op, aux, (const uint8_t*)in0 + lo * sh.in0, sh.in1 ? (const uint8_t*)in1 + lo * sh.in1 : nullptr,
                       cnt, (uint8_t*)out0 + lo * sh.out0, (sh.out1 && out1) ? (uint8_t*)out1 + lo * sh.out1 : nullptr)))
        return rc;
    }
  }
  for (size_t k : used) HIP_TRY(hipStreamWaitEvent(s, ctx->devs[k].ev_shard, 0));   // `stream` continues once every slice is back
  return D377_OK;
  };
  rc_final = body();
  return rc_final;
}

}  // namespace

extern "C" {

const char* d377_version(void) { return "decaf377_amd 0.3.0 (gfx950)"; }
const char* d377_last_error(void) { return d377_g_err; }

int d377_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int d377_ctx_create(const int* device_ids, int n_dev, d377_ctx** out) { return d377_ctx_create_ex(device_ids, n_dev, nullptr, out); }
int d377_ctx_create_ex(const int* device_ids, int n_dev, const d377_ctx_opts* opts, d377_ctx** out) {
  if (!out) return fail(D377_ERR_ARG, "%s", "null out pointer");
  *out = nullptr;
  int comb_bits = FB_BITS, comb_lazy = 0;
  if (opts) {
    if (opts->size < sizeof(d377_ctx_opts)) return fail(D377_ERR_ARG, "%s", "d377_ctx_opts::size: set it to sizeof(d377_ctx_opts)");
    if (opts->comb_bits != 0) comb_bits = opts->comb_bits;
    comb_lazy = opts->comb_lazy;
    if (comb_lazy != 0 && comb_lazy != 1) return fail(D377_ERR_ARG, "%s", "d377_ctx_opts::comb_lazy: 0 or 1");
    int rcw = with_fb_bits(comb_bits, [](auto) -> int { return D377_OK; });
    if (rcw) return rcw;
  }
  int avail = d377_device_count();
  if (avail <= 0) return fail(D377_ERR_NO_DEVICE, "%s", "no HIP device visible");
  std::vector<int> ids;
  if (!device_ids || n_dev <= 0) ids.push_back(0);
  else ids.assign(device_ids, device_ids + n_dev);
  for (int id : ids)
    if (id < 0 || id >= avail) return fail(D377_ERR_ARG, "%s", "device id out of range");
  d377_ctx* ctx = new (std::nothrow) d377_ctx();
  if (!ctx) return fail(D377_ERR_ARG, "%s", "out of host memory");
  ctx->devs.resize(ids.size());
  for (size_t k = 0; k < ids.size(); ++k) {
    ctx->devs[k].id = ids[k];
    ctx->devs[k].tune = &ctx->tune;
    ctx->devs[k].fb_bits = comb_bits;
    ctx->devs[k].fb_lazy = comb_lazy != 0;
    int rc = init_device(ctx->devs[k]);
    if (rc != D377_OK) {
      char saved[sizeof d377_g_err];
      memcpy(saved, d377_g_err, sizeof saved);
      for (auto& d : ctx->devs) free_device(d);
      delete ctx;
      memcpy(d377_g_err, saved, sizeof saved);
      return rc;
    }
  }
  // peer access between the context's devices (xGMI): the sharded device-pointer path copies slices directly
  ctx->peer.assign(ids.size() * ids.size(), 0);
  for (size_t a = 0; a < ids.size(); ++a)
    for (size_t b = 0; b < ids.size(); ++b) {
      if (ids[a] == ids[b]) { ctx->peer[a * ids.size() + b] = 1; continue; }
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, ids[a], ids[b]) == hipSuccess && can) {
        (void)hipSetDevice(ids[a]);
        hipError_t e = hipDeviceEnablePeerAccess(ids[b], 0);
        if (e != hipSuccess) (void)hipGetLastError();      // already enabled is fine; copies fall back to staging otherwise
        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ctx->peer[a * ids.size() + b] = 2;
      }
    }
  *out = ctx;
  return D377_OK;
}

void d377_ctx_destroy(d377_ctx* ctx) {
  if (!ctx) return;
  for (auto& d : ctx->devs) free_device(d);
  delete ctx;
}
int d377_ctx_num_devices(const d377_ctx* ctx) { return ctx ? (int)ctx->devs.size() : 0; }
int d377_ctx_comb_info(d377_ctx* ctx, int dev, int* comb_bits, int* built, uint64_t* table_bytes) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  if (dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "device index out of range");
  std::lock_guard<std::mutex> lock(ctx->mu);
  const DeviceState& d = ctx->devs[(size_t)dev];
  if (comb_bits) *comb_bits = d.fb_bits;
  if (built) *built = d.fbase != nullptr;
  if (table_bytes) {
    const uint64_t windows = (uint64_t)((252 + d.fb_bits - 1) / d.fb_bits), entries = ((uint64_t)1 << (d.fb_bits - 1)) + 1;
    *table_bytes = windows * entries * FBW_ENTRY_WORDS * sizeof(uint32_t);
  }
  return D377_OK;
}
int d377_ctx_invariant_failures(d377_ctx* ctx, int dev, uint64_t* count) {
  if (!ctx || !count) return fail(D377_ERR_ARG, "%s", "null argument");
  if (dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "device index out of range");
#if defined(D377_CHECK_INVARIANTS)
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceState& d = ctx->devs[(size_t)dev];
  HIP_TRY(hipSetDevice(d.id));
  HIP_TRY(hipDeviceSynchronize());
  uint32_t v = 0;
  HIP_TRY(hipMemcpy(&v, d.inv_fail, sizeof v, hipMemcpyDeviceToHost));
  *count = v;
  return 1;                                   // the checks are compiled in
#else
  *count = 0;
  return D377_OK;                             // not a checking build
#endif
}
// key -> [lo, hi] of the values d377_ctx_set_tuning accepts (besides D377_TUNE_DEFAULT)
static bool tuning_range(int key, long long* lo, long long* hi) {
  const long long big = (long long)1 << 62;
  switch (key) {
    // routes with a wave or a quad of lanes per element: their grids are n or n / 16 workgroups, and 4 x TINY_MAX is
    // computed in 64 bits -- 2^24 elements keeps both far inside a grid dimension (and is 2^10 times where they stop paying)
    case D377_TUNE_SMALL_MAX: case D377_TUNE_MSM_SMALL_MAX: case D377_TUNE_MSM_TINY_MAX:
    case D377_TUNE_TINY_MAX: *lo = 0; *hi = (long long)1 << 24; return true;
    // thresholds FROM which a route is taken: a huge value means never
    case D377_TUNE_DECOMPRESS_CHUNKED_MIN:
    case D377_TUNE_MSM_ENC_CHUNKED_MIN: *lo = 0; *hi = big; return true;
    case D377_TUNE_FB_WIDE: case D377_TUNE_MSM_CHUNKED_SUMS: case D377_TUNE_MSM_SORT_PACKED: *lo = 0; *hi = 1; return true;
    case D377_TUNE_FB_K: *lo = 1; *hi = DCB_KMAX; return true;
    case D377_TUNE_AFFINE_BLOCKS_PER_CU: *lo = 1; *hi = 64; return true;
    case D377_TUNE_MSM_WINDOW: *lo = 4; *hi = 18; return true;
    case D377_TUNE_MSM_SEG: *lo = 1; *hi = 128; return true;
    case D377_TUNE_MSM_SLICES: *lo = 1; *hi = 4096; return true;
    case D377_TUNE_MSM_RED: *lo = 2; *hi = 64; return true;
    case D377_TUNE_MSM_SKIP: *lo = 1; *hi = 64; return true;
    case D377_TUNE_CHUNK_PER_LANE: *lo = 1; *hi = DCB_K; return true;
  }
  return false;
}
int d377_ctx_set_tuning(d377_ctx* ctx, int key, int64_t value) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  long long lo, hi;
  if (!tuning_range(key, &lo, &hi)) return fail(D377_ERR_ARG, "%s", "unknown D377_TUNE_* key");
  if (value != D377_TUNE_DEFAULT && (value < lo || value > hi)) return fail(D377_ERR_ARG, "%s", "tuning value outside the key's range");
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->tune.v[key] = value;
  return D377_OK;
}
int d377_ctx_get_tuning(d377_ctx* ctx, int key, int64_t* value) {
  if (!ctx || !value) return fail(D377_ERR_ARG, "%s", "null argument");
  long long lo, hi;
  if (!tuning_range(key, &lo, &hi)) return fail(D377_ERR_ARG, "%s", "unknown D377_TUNE_* key");
  std::lock_guard<std::mutex> lock(ctx->mu);
  *value = ctx->tune.v[key];
  return D377_OK;
}
// ---- the lane-set pool's way back (dcb.hpp) ----
static int read_pool(DeviceState& d, std::vector<int>& pool, uint32_t health[4]) {
  pool.resize((size_t)d.dcb_sets);
  HIP_TRY(hipMemcpyAsync(d.pool_host, d.slot_pool, pool.size() * sizeof(int), hipMemcpyDeviceToHost, d.ctl_stream));
  HIP_TRY(hipMemcpyAsync(d.pool_host + d.dcb_sets, d.pool_health, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, d.ctl_stream));
  HIP_TRY(hipStreamSynchronize(d.ctl_stream));              // the control stream carries nothing else
  memcpy(pool.data(), d.pool_host, pool.size() * sizeof(int));
  memcpy(health, d.pool_host + d.dcb_sets, 4 * sizeof(uint32_t));
  return D377_OK;
}
int d377_ctx_health(d377_ctx* ctx, int dev, int* sets_claimed, uint64_t* waited_long, uint64_t* gave_up) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  if (dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "device index out of range");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceState& d = ctx->devs[(size_t)dev];
  HIP_TRY(hipSetDevice(d.id));
  std::vector<int> pool;
  uint32_t h[4];
  int rc = read_pool(d, pool, h);
  if (rc) return rc;
  int c = 0;
  for (int v : pool) c += v != 0;
  if (sets_claimed) *sets_claimed = c;
  if (waited_long) *waited_long = h[1];
  if (gave_up) *gave_up = h[2];
  return D377_OK;
}
int d377_ctx_starved_counter_dev(d377_ctx* ctx, int dev, const uint32_t** counter_dev) {
  if (!ctx || !counter_dev) return fail(D377_ERR_ARG, "%s", "null argument");
  if (dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "device index out of range");
  *counter_dev = ctx->devs[(size_t)dev].pool_health + 2;
  return D377_OK;
}
// Is anything this context enqueued (or was handed a stream for) still running on the device?
static bool device_busy(DeviceState& d) {
  if (hipStreamQuery(d.stream) == hipErrorNotReady) return true;
  if (d.vb_guard.used && hipEventQuery(d.vb_guard.ev) == hipErrorNotReady) return true;
  if (d.msm.guard.used && hipEventQuery(d.msm.guard.ev) == hipErrorNotReady) return true;
  (void)hipGetLastError();
  return false;
}
static bool wait_idle(DeviceState& d, int ms) {
  const auto t0 = std::chrono::steady_clock::now();
  while (device_busy(d)) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(ms)) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  return true;
}
int d377_ctx_reset_scratch(d377_ctx* ctx, int dev, int* sets_released) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  if (dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "device index out of range");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceState& d = ctx->devs[(size_t)dev];
  HIP_TRY(hipSetDevice(d.id));
  if (sets_released) *sets_released = 0;
  std::vector<int> before, after;
  uint32_t h[4];
  int rc;
  if ((rc = read_pool(d, before, h))) return rc;
  // Whatever is in flight gets time to finish or to move on: a workgroup draws a new ticket for every chunk it starts
  // (dcb.hpp: dcb_rounds), so over RESET_MIN_MS -- many chunks -- the word of every set that is in use changes, whether
  // its launch is one this context can see or a replayed hipGraph, and however long the launch is.  A ticket that
  // stood still between the two reads belongs to no running workgroup: a leak.  Those (and only those) are freed, by
  // compare-and-swap against the ticket, on the copy stream beside whatever waits for them.
  const auto t0 = std::chrono::steady_clock::now();
  const bool idle = wait_idle(d, RESET_WAIT_MS);
  const auto waited = std::chrono::steady_clock::now() - t0;
  if (waited < std::chrono::milliseconds(RESET_MIN_MS)) std::this_thread::sleep_for(std::chrono::milliseconds(RESET_MIN_MS) - waited);
  if ((rc = read_pool(d, after, h))) return rc;
  std::vector<int> idx, val;
  for (size_t i = 0; i < after.size(); ++i)
    if (after[i] != 0 && after[i] == before[i]) { idx.push_back((int)i); val.push_back(after[i]); }
  if (!idx.empty()) {
    int *d_idx = nullptr, *d_val = nullptr;
    HIP_TRY(hipMalloc(&d_idx, idx.size() * sizeof(int)));
    if (hipMalloc(&d_val, val.size() * sizeof(int)) != hipSuccess) { (void)hipFree(d_idx); return fail(D377_ERR_HIP, "%s", "hipMalloc failed"); }
    hipError_t e = hipMemcpyAsync(d_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, d.ctl_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_val, val.data(), val.size() * sizeof(int), hipMemcpyHostToDevice, d.ctl_stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_pool_release, dim3((unsigned)((idx.size() + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, d.ctl_stream, d.slot_pool, d_idx, d_val,
                         (int)idx.size());
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(d.ctl_stream);
    (void)hipFree(d_idx); (void)hipFree(d_val);
    if (e != hipSuccess) return fail(D377_ERR_HIP, "reset_scratch: %s", hipGetErrorString(e));
    if (sets_released) *sets_released = (int)idx.size();
  }
  if (!idle && !wait_idle(d, 10 * RESET_WAIT_MS))
    return fail(D377_ERR_HIP, "%s", "reset_scratch: the device is still busy after the leaked lane sets were freed");
  return D377_OK;
}
int d377_debug_poison_pool(d377_ctx* ctx, int dev, int sets) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  if (dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "device index out of range");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceState& d = ctx->devs[(size_t)dev];
  HIP_TRY(hipSetDevice(d.id));
  if (sets < 0 || sets > d.dcb_sets) sets = d.dcb_sets;
  if (!wait_idle(d, RESET_WAIT_MS)) return fail(D377_ERR_ARG, "%s", "poison_pool: the device is busy");
  std::vector<int> v((size_t)sets, 0x7FFFFFFF);              // a ticket no workgroup will draw for a long time
  if (sets) HIP_TRY(hipMemcpy(d.slot_pool, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
  return D377_OK;
}
#ifdef D377_WG_TIMES
// developer variant only (dcb.hpp): the records of the last chunked kernel of THIS translation unit, 6 x u64 per workgroup
extern "C" int d377_debug_wg_times(unsigned long long* out, int workgroups) {
  if (!out || workgroups < 0 || workgroups > WG_TIMES_MAX) return D377_ERR_ARG;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_times), (size_t)workgroups * 6 * sizeof(unsigned long long)));
  return D377_OK;
}
#endif
int d377_ctx_chunk_residency(const d377_ctx* ctx, int dev, int* sets_per_cu, int* max_blocks_per_cu, int* lds_pad_bytes) {
  if (!ctx || dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "bad context or device index");
  const DeviceState& d = ctx->devs[(size_t)dev];
  int mb = 0, pad = 0;
  for (int k = 0; k < CK_COUNT; ++k) {
    if (d.chunk_blocks[k] > mb) mb = d.chunk_blocks[k];
    if (d.chunk_lds[k] > pad) pad = d.chunk_lds[k];
  }
  if (sets_per_cu) *sets_per_cu = d.cus ? d.dcb_sets / d.cus : 0;
  if (max_blocks_per_cu) *max_blocks_per_cu = mb;
  if (lds_pad_bytes) *lds_pad_bytes = pad;
  return D377_OK;
}
int d377_ctx_peer_access(const d377_ctx* ctx, int dev_a, int dev_b) {
  if (!ctx || dev_a < 0 || dev_b < 0 || (size_t)dev_a >= ctx->devs.size() || (size_t)dev_b >= ctx->devs.size()) return -1;
  return ctx->peer[(size_t)dev_a * ctx->devs.size() + (size_t)dev_b];
}
int d377_ctx_device_id(const d377_ctx* ctx, int dev) {
  if (!ctx || dev < 0 || (size_t)dev >= ctx->devs.size()) return -1;
  return ctx->devs[(size_t)dev].id;
}

static int sqrt_root_ok(int root) {
  if (root == D377_SQRT_ROOT_ARK || root == D377_SQRT_ROOT_MIN_CURVE) return D377_OK;
  return fail(D377_ERR_ARG, "%s", "unknown square-root convention (D377_SQRT_ROOT_ARK / D377_SQRT_ROOT_MIN_CURVE)");
}
int d377_batch_sqrt_ratio_zeta(d377_ctx* ctx, const uint8_t* num32, const uint8_t* den32, size_t n, uint8_t* root32,
                               uint8_t* was_square) {
  return run_host(ctx, OP_SQRT, D377_SQRT_ROOT_ARK, num32, den32, n, root32, was_square);
}
int d377_batch_sqrt_ratio_zeta_ex(d377_ctx* ctx, int root, const uint8_t* num32, const uint8_t* den32, size_t n,
                                  uint8_t* root32, uint8_t* was_square) {
  int rc = sqrt_root_ok(root);
  return rc ? rc : run_host(ctx, OP_SQRT, root, num32, den32, n, root32, was_square);
}
int d377_batch_decompress(d377_ctx* ctx, const uint8_t* enc32, size_t n, uint64_t* xyzt, uint8_t* status) {
  return run_host(ctx, OP_DECOMPRESS, 0, enc32, nullptr, n, xyzt, status);
}
int d377_batch_compress(d377_ctx* ctx, const uint64_t* xyzt, size_t n, uint8_t* enc32) {
  return run_host(ctx, OP_COMPRESS, 0, xyzt, nullptr, n, enc32, nullptr);
}
int d377_batch_roundtrip(d377_ctx* ctx, const uint8_t* enc32, size_t n, uint8_t* enc32_out, uint8_t* status) {
  return run_host(ctx, OP_ROUNDTRIP, 0, enc32, nullptr, n, enc32_out, status);
}
int d377_batch_scalar_mul_base(d377_ctx* ctx, const uint8_t* scalar32, size_t n, uint8_t* enc32_out) {
  return run_host(ctx, OP_MUL_BASE, 0, scalar32, nullptr, n, enc32_out, nullptr);
}
int d377_batch_scalar_mul_var(d377_ctx* ctx, const uint8_t* enc32, const uint8_t* scalar32, size_t n,
                              uint8_t* enc32_out, uint8_t* status) {
  return run_host(ctx, OP_MUL_VAR, 0, enc32, scalar32, n, enc32_out, status);
}
int d377_batch_encode_to_curve(d377_ctx* ctx, const uint8_t* fq32, size_t n, uint8_t* enc32_out) {
  return run_host(ctx, OP_ENCODE, 0, fq32, nullptr, n, enc32_out, nullptr);
}
int d377_batch_hash_to_curve(d377_ctx* ctx, const uint8_t* r1_32, const uint8_t* r2_32, size_t n, uint8_t* enc32_out) {
  return run_host(ctx, OP_HASH, 0, r1_32, r2_32, n, enc32_out, nullptr);
}

int d377_batch_add(d377_ctx* ctx, const uint64_t* p_xyzt, const uint64_t* q_xyzt, size_t n, uint64_t* out_xyzt) {
  return run_host(ctx, OP_ADD, 0, p_xyzt, q_xyzt, n, out_xyzt, nullptr);
}
int d377_batch_sub(d377_ctx* ctx, const uint64_t* p_xyzt, const uint64_t* q_xyzt, size_t n, uint64_t* out_xyzt) {
  return run_host(ctx, OP_ADD, 1, p_xyzt, q_xyzt, n, out_xyzt, nullptr);
}
int d377_batch_double(d377_ctx* ctx, const uint64_t* p_xyzt, size_t n, uint64_t* out_xyzt) {
  return run_host(ctx, OP_DOUBLE, 0, p_xyzt, nullptr, n, out_xyzt, nullptr);
}
int d377_batch_eq(d377_ctx* ctx, const uint64_t* p_xyzt, const uint64_t* q_xyzt, size_t n, uint8_t* equal) {
  return run_host(ctx, OP_EQ, 0, p_xyzt, q_xyzt, n, equal, nullptr);
}
int d377_batch_add_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, const uint64_t* q_xyzt, size_t n,
                       uint64_t* out_xyzt) {
  return run_dev(ctx, dev, stream, OP_ADD, 0, p_xyzt, q_xyzt, n, out_xyzt, nullptr);
}
int d377_batch_sub_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, const uint64_t* q_xyzt, size_t n,
                       uint64_t* out_xyzt) {
  return run_dev(ctx, dev, stream, OP_ADD, 1, p_xyzt, q_xyzt, n, out_xyzt, nullptr);
}
int d377_batch_double_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, size_t n, uint64_t* out_xyzt) {
  return run_dev(ctx, dev, stream, OP_DOUBLE, 0, p_xyzt, nullptr, n, out_xyzt, nullptr);
}
int d377_batch_eq_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, const uint64_t* q_xyzt, size_t n,
                      uint8_t* equal) {
  return run_dev(ctx, dev, stream, OP_EQ, 0, p_xyzt, q_xyzt, n, equal, nullptr);
}

int d377_batch_fq_op(d377_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out, uint8_t* status) {
  if (op < D377_FQ_ADD || op > D377_FQ_INVERSE) return fail(D377_ERR_ARG, "%s", "unknown Fq operation");
  std::vector<uint8_t> scratch;
  if (!status) { scratch.resize(n ? n : 1); status = scratch.data(); }
  return run_host(ctx, op <= D377_FQ_MUL ? OP_FQ_BIN : OP_FQ_UN, op, a, op <= D377_FQ_MUL ? b : nullptr, n, out, status);
}
int d377_batch_fq_op_dev(d377_ctx* ctx, int dev, void* stream, int op, const uint64_t* a, const uint64_t* b, size_t n,
                         uint64_t* out, uint8_t* status) {
  if (op < D377_FQ_ADD || op > D377_FQ_INVERSE) return fail(D377_ERR_ARG, "%s", "unknown Fq operation");
  if (op == D377_FQ_INVERSE && !status) return fail(D377_ERR_ARG, "%s", "INVERSE needs a status buffer");
  // status is optional on the device path except for INVERSE; the kernel skips a null pointer
  return run_dev(ctx, dev, stream, op <= D377_FQ_MUL ? OP_FQ_BIN : OP_FQ_UN, op, a, op <= D377_FQ_MUL ? b : nullptr, n, out,
                 status);
}
int d377_batch_fq_from_bytes_checked(d377_ctx* ctx, const uint8_t* bytes32, size_t n, uint64_t* out, uint8_t* status) {
  return run_host(ctx, OP_FQ_CHECKED, 0, bytes32, nullptr, n, out, status);
}
int d377_batch_fq_to_bytes(d377_ctx* ctx, const uint64_t* a, size_t n, uint8_t* bytes32) {
  return run_host(ctx, OP_FQ_TO_BYTES, 0, a, nullptr, n, bytes32, nullptr);
}
int d377_batch_fr_from_le_bytes_mod_order(d377_ctx* ctx, const uint8_t* bytes32, size_t n, uint8_t* fr32_out) {
  return run_host(ctx, OP_FR_MOD, 0, bytes32, nullptr, n, fr32_out, nullptr);
}
int d377_batch_fr_from_bytes_checked(d377_ctx* ctx, const uint8_t* bytes32, size_t n, uint8_t* fr32_out, uint8_t* status) {
  return run_host(ctx, OP_FR_CHECKED, 0, bytes32, nullptr, n, fr32_out, status);
}
int d377_batch_fq_from_bytes_checked_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes32, size_t n, uint64_t* out,
                                         uint8_t* status) {
  return run_dev(ctx, dev, stream, OP_FQ_CHECKED, 0, bytes32, nullptr, n, out, status);
}
int d377_batch_fq_to_bytes_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* a, size_t n, uint8_t* bytes32) {
  return run_dev(ctx, dev, stream, OP_FQ_TO_BYTES, 0, a, nullptr, n, bytes32, nullptr);
}
int d377_batch_fr_from_le_bytes_mod_order_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes32, size_t n,
                                              uint8_t* fr32_out) {
  return run_dev(ctx, dev, stream, OP_FR_MOD, 0, bytes32, nullptr, n, fr32_out, nullptr);
}
int d377_batch_fr_from_bytes_checked_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes32, size_t n, uint8_t* fr32_out,
                                         uint8_t* status) {
  return run_dev(ctx, dev, stream, OP_FR_CHECKED, 0, bytes32, nullptr, n, fr32_out, status);
}
int d377_batch_fr_op(d377_ctx* ctx, int op, const uint8_t* a32, const uint8_t* b32, size_t n, uint8_t* out32, uint8_t* status) {
  if (op < D377_FQ_ADD || op > D377_FQ_INVERSE) return fail(D377_ERR_ARG, "%s", "unknown Fr operation");
  std::vector<uint8_t> scratch;
  if (!status) { scratch.resize(n ? n : 1); status = scratch.data(); }
  return run_host(ctx, op <= D377_FQ_MUL ? OP_FR_BIN : OP_FR_UN, op, a32, op <= D377_FQ_MUL ? b32 : nullptr, n, out32, status);
}
int d377_batch_fr_op_dev(d377_ctx* ctx, int dev, void* stream, int op, const uint8_t* a32, const uint8_t* b32, size_t n,
                         uint8_t* out32, uint8_t* status) {
  if (op < D377_FQ_ADD || op > D377_FQ_INVERSE) return fail(D377_ERR_ARG, "%s", "unknown Fr operation");
  if (op == D377_FQ_INVERSE && !status) return fail(D377_ERR_ARG, "%s", "INVERSE needs a status buffer");
  return run_dev(ctx, dev, stream, op <= D377_FQ_MUL ? OP_FR_BIN : OP_FR_UN, op, a32, op <= D377_FQ_MUL ? b32 : nullptr, n, out32,
                 status);
}
int d377_batch_scalar_mul_var_element(d377_ctx* ctx, const uint64_t* p_xyzt, const uint8_t* scalar32, size_t n, uint64_t* out_xyzt) {
  return run_host(ctx, OP_MUL_VAR_EL, 0, p_xyzt, scalar32, n, out_xyzt, nullptr);
}
int d377_batch_scalar_mul_base_element(d377_ctx* ctx, const uint8_t* scalar32, size_t n, uint64_t* out_xyzt) {
  return run_host(ctx, OP_MUL_BASE_EL, 0, scalar32, nullptr, n, out_xyzt, nullptr);
}
int d377_batch_compress_to_field(d377_ctx* ctx, const uint64_t* p_xyzt, size_t n, uint64_t* fq_out) {
  return run_host(ctx, OP_COMPRESS_FIELD, 0, p_xyzt, nullptr, n, fq_out, nullptr);
}
int d377_batch_encode_to_curve_element(d377_ctx* ctx, const uint8_t* fq32, size_t n, uint64_t* out_xyzt) {
  return run_host(ctx, OP_ENCODE_EL, 0, fq32, nullptr, n, out_xyzt, nullptr);
}
int d377_batch_hash_to_curve_element(d377_ctx* ctx, const uint8_t* r1_32, const uint8_t* r2_32, size_t n, uint64_t* out_xyzt) {
  return run_host(ctx, OP_HASH_EL, 0, r1_32, r2_32, n, out_xyzt, nullptr);
}
int d377_batch_scalar_mul_var_element_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, const uint8_t* scalar32,
                                          size_t n, uint64_t* out_xyzt) {
  return run_dev(ctx, dev, stream, OP_MUL_VAR_EL, 0, p_xyzt, scalar32, n, out_xyzt, nullptr);
}
int d377_batch_scalar_mul_base_element_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* scalar32, size_t n,
                                           uint64_t* out_xyzt) {
  return run_dev(ctx, dev, stream, OP_MUL_BASE_EL, 0, scalar32, nullptr, n, out_xyzt, nullptr);
}
int d377_batch_compress_to_field_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, size_t n, uint64_t* fq_out) {
  return run_dev(ctx, dev, stream, OP_COMPRESS_FIELD, 0, p_xyzt, nullptr, n, fq_out, nullptr);
}
int d377_batch_encode_to_curve_element_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* fq32, size_t n, uint64_t* out_xyzt) {
  return run_dev(ctx, dev, stream, OP_ENCODE_EL, 0, fq32, nullptr, n, out_xyzt, nullptr);
}
int d377_batch_hash_to_curve_element_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* r1_32, const uint8_t* r2_32,
                                         size_t n, uint64_t* out_xyzt) {
  return run_dev(ctx, dev, stream, OP_HASH_EL, 0, r1_32, r2_32, n, out_xyzt, nullptr);
}
int d377_batch_neg(d377_ctx* ctx, const uint64_t* p_xyzt, size_t n, uint64_t* out_xyzt) {
  return run_host(ctx, OP_NEG, 0, p_xyzt, nullptr, n, out_xyzt, nullptr);
}
int d377_batch_is_identity(d377_ctx* ctx, const uint64_t* p_xyzt, size_t n, uint8_t* is_identity) {
  return run_host(ctx, OP_IS_IDENTITY, 0, p_xyzt, nullptr, n, is_identity, nullptr);
}
int d377_batch_neg_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, size_t n, uint64_t* out_xyzt) {
  return run_dev(ctx, dev, stream, OP_NEG, 0, p_xyzt, nullptr, n, out_xyzt, nullptr);
}
int d377_batch_is_identity_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, size_t n,
                               uint8_t* is_identity) {
  return run_dev(ctx, dev, stream, OP_IS_IDENTITY, 0, p_xyzt, nullptr, n, is_identity, nullptr);
}
// Element::IDENTITY / Element::GENERATOR in the external layout (src/min_curve/element.rs:53-81): the
// Montgomery (R = 2^256) limbs the reference writes down in its source
void d377_identity(uint64_t xyzt[16]) {
  static const uint64_t one[4] = {0x7d1c7ffffffffff3ULL, 0x7257f50f6ffffff2ULL, 0x16d81575512c0feeULL, 0x0d4bda322bbb9a9dULL};
  for (int i = 0; i < 16; ++i) xyzt[i] = 0;
  for (int i = 0; i < 4; ++i) { xyzt[4 + i] = one[i]; xyzt[8 + i] = one[i]; }
}
void d377_generator(uint64_t xyzt[16]) {
  static const uint64_t g[16] = {
      5825153684096051627ULL, 16988948339439369204ULL, 186539475124256708ULL, 1230075515893193738ULL,
      9786171649960077610ULL, 13527783345193426398ULL, 10983305067350511165ULL, 1251302644532346138ULL,
      0x7d1c7ffffffffff3ULL, 0x7257f50f6ffffff2ULL, 0x16d81575512c0feeULL, 0x0d4bda322bbb9a9dULL,
      7466800842436274004ULL, 14314110021432015475ULL, 14108125795146788134ULL, 1305086759679105397ULL};
  for (int i = 0; i < 16; ++i) xyzt[i] = g[i];
}

static int wide_op(size_t len, Op o48, Op o64, Op* out) {
  if (len == 48) { *out = o48; return D377_OK; }
  if (len == 64) { *out = o64; return D377_OK; }
  return fail(D377_ERR_ARG, "%s", "wide records must be 48 or 64 bytes (32-byte records use the plain entry points)");
}
int d377_batch_fq_from_wide_bytes(d377_ctx* ctx, const uint8_t* bytes, size_t len, size_t n, uint8_t* fq32_out) {
  Op op; int rc = wide_op(len, OP_WIDE48, OP_WIDE64, &op);
  return rc ? rc : run_host(ctx, op, 0, bytes, nullptr, n, fq32_out, nullptr);
}
int d377_batch_fr_from_wide_bytes(d377_ctx* ctx, const uint8_t* bytes, size_t len, size_t n, uint8_t* fr32_out) {
  Op op; int rc = wide_op(len, OP_FR_WIDE48, OP_FR_WIDE64, &op);
  return rc ? rc : run_host(ctx, op, 0, bytes, nullptr, n, fr32_out, nullptr);
}
int d377_batch_fr_from_wide_bytes_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes, size_t len, size_t n,
                                      uint8_t* fr32_out) {
  Op op; int rc = wide_op(len, OP_FR_WIDE48, OP_FR_WIDE64, &op);
  return rc ? rc : run_dev(ctx, dev, stream, op, 0, bytes, nullptr, n, fr32_out, nullptr);
}
int d377_batch_encode_to_curve_wide(d377_ctx* ctx, const uint8_t* bytes, size_t len, size_t n, uint8_t* enc32_out) {
  Op op; int rc = wide_op(len, OP_ENCODE_WIDE48, OP_ENCODE_WIDE64, &op);
  return rc ? rc : run_host(ctx, op, 0, bytes, nullptr, n, enc32_out, nullptr);
}
int d377_batch_to_affine(d377_ctx* ctx, const uint64_t* xyzt, size_t n, uint64_t* xy) {
  return run_host(ctx, OP_AFFINE, 0, xyzt, nullptr, n, xy, nullptr);
}
int d377_batch_fq_from_wide_bytes_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes, size_t len, size_t n,
                                      uint8_t* fq32_out) {
  Op op; int rc = wide_op(len, OP_WIDE48, OP_WIDE64, &op);
  return rc ? rc : run_dev(ctx, dev, stream, op, 0, bytes, nullptr, n, fq32_out, nullptr);
}
int d377_batch_encode_to_curve_wide_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes, size_t len, size_t n,
                                        uint8_t* enc32_out) {
  Op op; int rc = wide_op(len, OP_ENCODE_WIDE48, OP_ENCODE_WIDE64, &op);
  return rc ? rc : run_dev(ctx, dev, stream, op, 0, bytes, nullptr, n, enc32_out, nullptr);
}
int d377_batch_to_affine_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* xyzt, size_t n, uint64_t* xy) {
  return run_dev(ctx, dev, stream, OP_AFFINE, 0, xyzt, nullptr, n, xy, nullptr);
}

int d377_batch_sqrt_ratio_zeta_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* num32, const uint8_t* den32,
                                   size_t n, uint8_t* root32, uint8_t* was_square) {
  return run_dev(ctx, dev, stream, OP_SQRT, D377_SQRT_ROOT_ARK, num32, den32, n, root32, was_square);
}
int d377_batch_sqrt_ratio_zeta_ex_dev(d377_ctx* ctx, int dev, void* stream, int root, const uint8_t* num32,
                                      const uint8_t* den32, size_t n, uint8_t* root32, uint8_t* was_square) {
  int rc = sqrt_root_ok(root);
  return rc ? rc : run_dev(ctx, dev, stream, OP_SQRT, root, num32, den32, n, root32, was_square);
}
int d377_batch_decompress_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* enc32, size_t n, uint64_t* xyzt,
                              uint8_t* status) {
  return run_dev(ctx, dev, stream, OP_DECOMPRESS, 0, enc32, nullptr, n, xyzt, status);
}
int d377_batch_compress_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* xyzt, size_t n, uint8_t* enc32) {
  return run_dev(ctx, dev, stream, OP_COMPRESS, 0, xyzt, nullptr, n, enc32, nullptr);
}
int d377_batch_roundtrip_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* enc32, size_t n, uint8_t* enc32_out,
                             uint8_t* status) {
  return run_dev(ctx, dev, stream, OP_ROUNDTRIP, 0, enc32, nullptr, n, enc32_out, status);
}
int d377_batch_scalar_mul_base_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* scalar32, size_t n,
                                   uint8_t* enc32_out) {
  return run_dev(ctx, dev, stream, OP_MUL_BASE, 0, scalar32, nullptr, n, enc32_out, nullptr);
}
int d377_batch_scalar_mul_var_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* enc32, const uint8_t* scalar32,
                                  size_t n, uint8_t* enc32_out, uint8_t* status) {
  return run_dev(ctx, dev, stream, OP_MUL_VAR, 0, enc32, scalar32, n, enc32_out, status);
}
int d377_batch_encode_to_curve_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* fq32, size_t n,
                                   uint8_t* enc32_out) {
  return run_dev(ctx, dev, stream, OP_ENCODE, 0, fq32, nullptr, n, enc32_out, nullptr);
}
int d377_batch_hash_to_curve_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* r1_32, const uint8_t* r2_32,
                                 size_t n, uint8_t* enc32_out) {
  return run_dev(ctx, dev, stream, OP_HASH, 0, r1_32, r2_32, n, enc32_out, nullptr);
}

int d377_batch_sharded_dev(d377_ctx* ctx, int root_dev, void* stream, int op, const void* in0, const void* in1, size_t n,
                           void* out0, void* out1) {
  Op o;
  switch (op) {
    case D377_OP_SQRT_RATIO_ZETA: o = OP_SQRT; break;
    case D377_OP_DECOMPRESS: o = OP_DECOMPRESS; break;
    case D377_OP_COMPRESS: o = OP_COMPRESS; break;
    case D377_OP_ROUNDTRIP: o = OP_ROUNDTRIP; break;
    case D377_OP_SCALAR_MUL_BASE: o = OP_MUL_BASE; break;
    case D377_OP_SCALAR_MUL_VAR: o = OP_MUL_VAR; break;
    case D377_OP_ENCODE_TO_CURVE: o = OP_ENCODE; break;
    case D377_OP_HASH_TO_CURVE: o = OP_HASH; break;
    case D377_OP_SCALAR_MUL_VAR_ELEMENT: o = OP_MUL_VAR_EL; break;
    case D377_OP_SCALAR_MUL_BASE_ELEMENT: o = OP_MUL_BASE_EL; break;
    default: return fail(D377_ERR_ARG, "%s", "unknown D377_OP_* code");
  }
  return run_sharded_dev(ctx, root_dev, stream, o, 0, in0, in1, n, out0, out1);
}

}  // extern "C"
