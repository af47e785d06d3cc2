// device_util.hpp -- record I/O and table-slot helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve.hpp"

namespace d377 {

constexpr int BLOCK = 256;
#ifndef D377_WAVES_PER_SIMD
#define D377_WAVES_PER_SIMD 2
#endif
constexpr int WAVES_PER_SIMD = D377_WAVES_PER_SIMD;   // occupancy the kernels are built for (VGPR budget 512 / this; LDS = POW_TAB * 9 KiB per block)
constexpr int SLOT = 12;                     // one field element slot in a table entry: 9 limbs + 3 pad = 3 x 16 B
constexpr int VB_ENTRIES = 9;                // cached 0..8 times P
constexpr int VB_ENTRY_WORDS = 4 * SLOT;     // ypx, ymx, z2, kt: 192 B, 64-B aligned
constexpr int AP_WORDS = 32;                 // affine cached point record: ypx, ymx, kt = 27 limbs in a 128-byte, 128-byte-aligned slot
constexpr int FBW_ENTRY_WORDS = AP_WORDS;    // fixed-base comb entries are such records

// ------------------------------------------------------------------ record I/O helpers ---
__device__ __forceinline__ void load32(const uint8_t* base, size_t i, uint32_t w[8]) {
  const uint4* p = reinterpret_cast<const uint4*>(base) + 2 * i;
  uint4 a = p[0], b = p[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void store32(uint8_t* base, size_t i, const uint32_t w[8]) {
  uint4* p = reinterpret_cast<uint4*>(base) + 2 * i;
  p[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ void store32_zero(uint8_t* base, size_t i) {
  uint4* p = reinterpret_cast<uint4*>(base) + 2 * i;
  p[0] = make_uint4(0, 0, 0, 0);
  p[1] = make_uint4(0, 0, 0, 0);
}
__device__ __forceinline__ void store_ge_mont256(uint64_t* xyzt, size_t i, const ge& g) {
  uint8_t* b = reinterpret_cast<uint8_t*>(xyzt);
  uint32_t w[8];
  fe_to_mont256_words(g.x, w); store32(b, 4 * i + 0, w);
  fe_to_mont256_words(g.y, w); store32(b, 4 * i + 1, w);
  fe_to_mont256_words(g.z, w); store32(b, 4 * i + 2, w);
  fe_to_mont256_words(g.t, w); store32(b, 4 * i + 3, w);
}
__device__ __forceinline__ ge load_ge_mont256(const uint64_t* xyzt, size_t i) {
  const uint8_t* b = reinterpret_cast<const uint8_t*>(xyzt);
  uint32_t w[8];
  ge g;
  load32(b, 4 * i + 0, w); g.x = fe_from_mont256_words(w);
  load32(b, 4 * i + 1, w); g.y = fe_from_mont256_words(w);
  load32(b, 4 * i + 2, w); g.z = fe_from_mont256_words(w);
  load32(b, 4 * i + 3, w); g.t = fe_from_mont256_words(w);
  return g;
}

__device__ __forceinline__ void load_record128(const uint64_t* xyzt, size_t i, uint32_t w[32]) {
  const uint8_t* b = reinterpret_cast<const uint8_t*>(xyzt);
  load32(b, 4 * i + 0, w); load32(b, 4 * i + 1, w + 8); load32(b, 4 * i + 2, w + 16); load32(b, 4 * i + 3, w + 24);
}
__device__ __forceinline__ void store_record128(uint64_t* xyzt, size_t i, const uint32_t w[32]) {
  uint8_t* b = reinterpret_cast<uint8_t*>(xyzt);
  store32(b, 4 * i + 0, w); store32(b, 4 * i + 1, w + 8); store32(b, 4 * i + 2, w + 16); store32(b, 4 * i + 3, w + 24);
}

__device__ __forceinline__ void slot_store(uint32_t* p, const fe& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
  q[2] = make_uint4(v.l[8], 0, 0, 0);
}
__device__ __forceinline__ fe slot_load(const uint32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1], c = q[2];
  fe r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
  r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  r.l[8] = c.x;
  return r;
}

// Cached AFFINE point record: Y+X, Y-X (both carried: a negative digit swaps them), 2dXY -- 27 limbs packed into a
// 128-byte, 128-byte-aligned slot, fetched as seven 16-byte loads = exactly two 64-byte sectors per gather.  (Three
// padded 48-byte slots, 144 bytes, straddle sector boundaries: 3.25 sectors per gather on average.)  Shared by the
// MSM's point records and the fixed-base comb.
__device__ __forceinline__ void pt_store_affine(uint32_t* p, const gea& c) {
  uint32_t w[AP_WORDS];
#pragma unroll
  for (int i = 0; i < NL; ++i) { w[i] = c.ypx.l[i]; w[NL + i] = c.ymx.l[i]; w[2 * NL + i] = c.kt.l[i]; }
#pragma unroll
  for (int i = 3 * NL; i < AP_WORDS; ++i) w[i] = 0;
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < AP_WORDS / 4; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
// The record as it lies in memory, and the cached point made of it.  A kernel that gathers the NEXT record while it adds
// the current one keeps the raw words across the addition and converts them only when their turn comes: the conversion's
// selects read the loaded registers, and placed right behind the loads (as pt_load_affine does) they make the wave wait for
// the gather before the addition it was meant to hide behind (seen in k_msm_spans' code: s_waitcnt vmcnt(5) and eighteen
// v_cndmask between the loads and the first product).
struct gea_raw { uint4 v[7]; };
__device__ __forceinline__ gea_raw pt_load_affine_raw(const uint32_t* p) {
  gea_raw r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < 7; ++i) r.v[i] = q[i];
  return r;
}
__device__ __forceinline__ gea gea_from_raw(const gea_raw& r, bool swap) {
  uint32_t w[28];
#pragma unroll
  for (int i = 0; i < 7; ++i) { w[4 * i] = r.v[i].x; w[4 * i + 1] = r.v[i].y; w[4 * i + 2] = r.v[i].z; w[4 * i + 3] = r.v[i].w; }
  gea c;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    c.ypx.l[i] = swap ? w[NL + i] : w[i];
    c.ymx.l[i] = swap ? w[i] : w[NL + i];
    c.kt.l[i] = w[2 * NL + i];
  }
  return c;
}
__device__ __forceinline__ gea pt_load_affine(const uint32_t* p, bool swap) { return gea_from_raw(pt_load_affine_raw(p), swap); }
// "The cached point exists from here on, and no load moves above this line": placed between gea_from_raw and the reload of
// the same raw buffer, it keeps the compiler from sinking the conversion's selects below the new loads -- which would need
// the old and the new record in registers at once and a copy of words still in flight at the loop's back edge.
__device__ __forceinline__ void gea_pin(gea& c) {
#pragma unroll
  for (int i = 0; i < NL; ++i) asm volatile("" : "+v"(c.ypx.l[i]), "+v"(c.ymx.l[i]), "+v"(c.kt.l[i]) : : "memory");
}

// the 8 odd powers of the fixed exponentiation, one LDS column per lane (bank = lane: no conflicts)
struct LdsPowTab {
  uint32_t* col;                           // &lds[threadIdx.x]
  __device__ __forceinline__ void put(int j, const fe& v) {
#pragma unroll
    for (int k = 0; k < NL; ++k) col[(j * NL + k) * BLOCK] = v.l[k];
  }
  __device__ __forceinline__ fe get(int j) const {
    fe r;
#pragma unroll
    for (int k = 0; k < NL; ++k) r.l[k] = col[(j * NL + k) * BLOCK];
    return r;
  }
};
// ---- 128-byte records through LDS: coalesced --------------------------------------------------------------------------
// A lane that loads its own 128-byte record issues eight 16-byte loads at a lane stride of 128 bytes: every instruction
// touches 64 lines and uses an eighth of each.  The kernels that mostly move records (k_neg, k_add) load a wave's 64 records as
// eight fully coalesced 1 KiB instructions into LDS instead (16-byte chunks, XOR-swizzled so that neither the chunk-order
// writes nor the record-order reads conflict) and each lane picks its record up there; results go back the same way.  tools/attic/record_io_bench.hip, 2^22 records: one input and one output stream 4.5 ->
// 5.7 TB/s, two inputs and one output 3.8 -> 5.5 TB/s.  In-place calls stay safe: a wave reads all of its tile before it
// writes any of it.
constexpr int REC_TILE_CHUNKS = 64 * 8;                 // 64 records x 8 chunks of 16 bytes: 8 KiB per wave
__device__ __forceinline__ int rec_swz(int c) { return (c & ~7) | ((c & 7) ^ ((c >> 3) & 7)); }
__device__ __forceinline__ void rec_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// records [rec0, rec0 + 64) of `base`, as far as they exist (rec0 < n): this lane's record -> w (zeros past the end)
__device__ __forceinline__ void wave_load_records128(const uint64_t* base, size_t rec0, size_t n, uint4* tile, int lane, uint32_t w[32]) {
  const uint4* g = reinterpret_cast<const uint4*>(base) + 8 * rec0;
  const size_t left = n - rec0;
  const int chunks = left >= 64 ? REC_TILE_CHUNKS : (int)left * 8;
  uint4 r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { const int c = k * 64 + lane; r[k] = c < chunks ? g[c] : make_uint4(0, 0, 0, 0); }
#pragma unroll
  for (int k = 0; k < 8; ++k) tile[rec_swz(k * 64 + lane)] = r[k];
  rec_lds_fence();
#pragma unroll
  for (int k = 0; k < 8; ++k) { const uint4 v = tile[rec_swz(lane * 8 + k)]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
  rec_lds_fence();                                      // the tile is free again
}
__device__ __forceinline__ void wave_store_records128(uint64_t* base, size_t rec0, size_t n, uint4* tile, int lane, const uint32_t w[32]) {
  uint4* g = reinterpret_cast<uint4*>(base) + 8 * rec0;
  const size_t left = n - rec0;
  const int chunks = left >= 64 ? REC_TILE_CHUNKS : (int)left * 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) tile[rec_swz(lane * 8 + k)] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
  rec_lds_fence();
#pragma unroll
  for (int k = 0; k < 8; ++k) { const int c = k * 64 + lane; if (c < chunks) g[c] = tile[rec_swz(c)]; }
  rec_lds_fence();
}
// the walk of a workgroup's waves over the batch in tiles of 64 records (grid-stride)
#define D377_RECORD_TILES(rec0)                                                                       \
  __shared__ uint4 rec_tiles_[BLOCK / 64][REC_TILE_CHUNKS];                                          \
  const int lane = threadIdx.x & 63;                                                                 \
  uint4* const tile = rec_tiles_[threadIdx.x >> 6];                                                  \
  for (size_t rec0 = ((size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 64; rec0 < n; rec0 += (size_t)gridDim.x * BLOCK)

#define D377_POW_LDS()                                         \
  __shared__ uint32_t lds_pow_[POW_TAB * NL * BLOCK];                \
  LdsPowTab pt;                                                \
  pt.col = lds_pow_ + threadIdx.x


}  // namespace d377
