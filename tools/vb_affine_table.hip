// Experiment (round 5, review item 3): AFFINE window-table entries for the variable-base multiplication, built through a
// batched inversion -- the one open idea for k_scalar_mul_var that round 4 sized (+2-3 %) and did not build.
//
//   proj    the product kernel's loop: per element a table of 0..8 P as cached PROJECTIVE entries (192 B, 8 entries built
//           with 8-product additions), 63 windows of 4 doublings and one 8-product addition
//   affine  per CHUNK of 8 elements per lane: the multiples 1..8 P of all eight as projective points in scratch, the 64 Z's
//           inverted together (Montgomery's trick, ONE divsteps inversion per lane per chunk), every multiple turned into a
//           cached AFFINE record (128 B: the MSM's record), then the eight window loops with 7-product mixed additions
//
// Same inputs, alternating on one box, whole kernel timed (table build included); the two outputs are compared as
// group elements.  No square roots, no encodings: the loop the headline kernel spends 90 % of its instructions in.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Idecaf377_amd/csrc tools/vb_affine_table.hip -o tools/vb_affine_table
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "curve.hpp"
#include "device_util.hpp"
using namespace d377;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int CHUNK_E = 8;                       // elements per lane per shared inversion (the product's DCB_K)

struct GlobalTab {
  uint32_t* base; size_t nthreads, tid;
  __device__ __forceinline__ void store(int j, const gec& c) {
    uint32_t* p = base + ((size_t)j * nthreads + tid) * VB_ENTRY_WORDS;
    slot_store(p, c.ypx); slot_store(p + SLOT, c.ymx); slot_store(p + 2 * SLOT, c.z2); slot_store(p + 3 * SLOT, c.kt);
  }
  __device__ __forceinline__ gec load(int j, bool swap) const {
    const uint32_t* p = base + ((size_t)j * nthreads + tid) * VB_ENTRY_WORDS;
    gec c;
    c.ypx = slot_load(p + (swap ? SLOT : 0)); c.ymx = slot_load(p + (swap ? 0 : SLOT));
    c.z2 = slot_load(p + 2 * SLOT); c.kt = slot_load(p + 3 * SLOT);
    return c;
  }
};
__device__ __forceinline__ ge load_pt(const uint32_t* p) { ge g; g.x = slot_load(p); g.y = slot_load(p + SLOT); g.z = slot_load(p + 2 * SLOT); g.t = slot_load(p + 3 * SLOT); return g; }
__device__ __forceinline__ void store_pt(uint32_t* p, const ge& g) { slot_store(p, g.x); slot_store(p + SLOT, g.y); slot_store(p + 2 * SLOT, g.z); slot_store(p + 3 * SLOT, g.t); }

__global__ void __launch_bounds__(BLOCK, 2) k_proj(const uint32_t* pts, const uint8_t* scalar32, size_t n, uint32_t* out, uint32_t* scratch) {
  const size_t nthreads = (size_t)gridDim.x * BLOCK, tid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  for (size_t i = tid; i < n; i += nthreads) {
    uint32_t k[8], dg[8];
    load32(scalar32, i, k);
    fr_reduce_words(k);
    fr_recode_signed16(k, dg);
    GlobalTab tab; tab.base = scratch; tab.nthreads = nthreads; tab.tid = tid;
    store_pt(out + i * 48, ge_scalar_mul_w4(load_pt(pts + i * 48), dg, tab));
  }
}

// scratch of the affine form, lane-interleaved like the product's tables:
//   P[e][j] (j = 0..7: (j+1) P as X, Y, Z in three slots)   at  pbase + ((e*8 + j) * nthreads + tid) * 3*SLOT words
//   C[e][j] the exclusive prefix product of the Z's          at  cbase + ((e*8 + j) * nthreads + tid) * SLOT words
//   A[e][j] (j = 0..8) cached affine records, 128 B          at  abase + ((e*9 + j) * nthreads + tid) * AP_WORDS words
__global__ void __launch_bounds__(BLOCK, 2) k_affine(const uint32_t* pts, const uint8_t* scalar32, size_t n, uint32_t* out,
                                                    uint32_t* pbase, uint32_t* cbase, uint32_t* abase) {
  const size_t nthreads = (size_t)gridDim.x * BLOCK, tid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  for (size_t i0 = tid; i0 < n; i0 += nthreads * CHUNK_E) {
    // A: the multiples, projective, and the running product of their Z's
    fe c = fe_const(FE_ONE);
    int cnt = 0;
#pragma unroll 1
    for (int e = 0; e < CHUNK_E; ++e) {
      const size_t i = i0 + (size_t)e * nthreads;
      if (i >= n) break;
      cnt = e + 1;
      const ge p = load_pt(pts + i * 48);
      const gec pc = ge_to_cached(p);
      ge acc = p;
#pragma unroll 1
      for (int j = 0; j < 8; ++j) {
        if (j == 1) acc = ge_double_fast(p, true);
        else if (j > 1) acc = ge_add_cached(acc, pc, false, true);
        uint32_t* q = pbase + ((size_t)(e * 8 + j) * nthreads + tid) * (3 * SLOT);
        slot_store(q, acc.x); slot_store(q + SLOT, acc.y); slot_store(q + 2 * SLOT, acc.z);
        slot_store(cbase + ((size_t)(e * 8 + j) * nthreads + tid) * SLOT, c);
        c = fe_mul(c, acc.z);
      }
    }
    // B: one inversion, then every multiple as a cached affine record
    fe inv = fe_invert(c);
    gea id;
    id.ypx = fe_const(FE_ONE); id.ymx = fe_const(FE_ONE); id.kt = fe_zero();
#pragma unroll 1
    for (int e = cnt - 1; e >= 0; --e) {
      pt_store_affine(abase + ((size_t)(e * 9) * nthreads + tid) * AP_WORDS, id);
#pragma unroll 1
      for (int j = 7; j >= 0; --j) {
        const uint32_t* q = pbase + ((size_t)(e * 8 + j) * nthreads + tid) * (3 * SLOT);
        const fe z = slot_load(q + 2 * SLOT);
        const fe zi = fe_mul(inv, slot_load(cbase + ((size_t)(e * 8 + j) * nthreads + tid) * SLOT));
        inv = fe_mul(inv, z);
        const fe x = fe_mul(slot_load(q), zi), y = fe_mul(slot_load(q + SLOT), zi);
        pt_store_affine(abase + ((size_t)(e * 9 + j + 1) * nthreads + tid) * AP_WORDS, gea_from_affine(x, y));
      }
    }
    // C: the window loops
#pragma unroll 1
    for (int e = 0; e < cnt; ++e) {
      const size_t i = i0 + (size_t)e * nthreads;
      uint32_t k[8], dg[8];
      load32(scalar32, i, k);
      fr_reduce_words(k);
      fr_recode_signed16(k, dg);
      const uint32_t* tab = abase + ((size_t)(e * 9) * nthreads + tid) * AP_WORDS;
      int d = fr_digit(dg, 63);                      // 0 or 1
      ge r = ge_select(d != 0, load_pt(pts + i * 48), ge_identity());
#pragma unroll 1
      for (int w = 62; w >= 0; --w) {
        d = fr_digit(dg, w);
        const bool neg = d < 0;
        // raw words now, the cached point after the doublings (device_util.hpp: a record converted right behind its loads makes
        // the wave wait for the gather before the work that was to hide it)
        const gea_raw raw = pt_load_affine_raw(tab + (size_t)(neg ? -d : d) * nthreads * AP_WORDS);
        asm volatile("" ::: "memory");
#pragma unroll 1
        for (int j = 0; j < 4; ++j) r = ge_double_neg(r, j == 3);   // (-2)^4 = 16
        r = ge_add_affine(r, gea_from_raw(raw, neg), neg, w == 0);
      }
      store_pt(out + i * 48, r);
    }
  }
}
__global__ void k_fill(uint32_t* pts, size_t n) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  ge g = ge_generator();
  for (int j = 0; j < (int)(i % 5); ++j) g = ge_double(g);
  store_pt(pts + i * 48, g);
}
// the same group element?  (x1 z2 == x2 z1 and y1 z2 == y2 z1, up to the torsion the decaf quotient ignores: x1 y2 == y1 x2)
__global__ void k_cmp(const uint32_t* a, const uint32_t* b, size_t n, unsigned* bad) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const ge p = load_pt(a + i * 48), q = load_pt(b + i * 48);
  const fe l = fe_canon(fe_mul(p.x, q.y)), r = fe_canon(fe_mul(p.y, q.x));
  bool same = true;
  for (int k = 0; k < NL; ++k) same &= l.l[k] == r.l[k];
  if (!same) atomicAdd(bad, 1u);
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const size_t n = (size_t)1 << 22;
  const int blocks = p.multiProcessorCount * 2;
  const size_t lanes = (size_t)blocks * BLOCK;
  uint32_t *pts, *out_a, *out_b, *scratch, *pb, *cb, *ab; uint8_t* k; unsigned* bad;
  CK(hipMalloc(&pts, n * 192)); CK(hipMalloc(&out_a, n * 192)); CK(hipMalloc(&out_b, n * 192)); CK(hipMalloc(&k, n * 32)); CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&scratch, lanes * VB_ENTRIES * VB_ENTRY_WORDS * 4));
  CK(hipMalloc(&pb, lanes * CHUNK_E * 8 * 3 * SLOT * 4)); CK(hipMalloc(&cb, lanes * CHUNK_E * 8 * SLOT * 4)); CK(hipMalloc(&ab, lanes * CHUNK_E * 9 * AP_WORDS * 4));
  hipLaunchKernelGGL(k_fill, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, pts, n);
  std::vector<uint8_t> hk(n * 32); uint64_t s = 88172645463325252ull;
  for (auto& b : hk) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; b = (uint8_t)s; }
  CK(hipMemcpy(k, hk.data(), n * 32, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("window loop of the variable-base multiplication with its table build, 2^22 elements, 2 waves per SIMD, %d CUs; alternating\n", p.multiProcessorCount);
  for (int r = 0; r < 8; ++r) {
    const bool proj = (r & 1) == 0;
    CK(hipEventRecord(e0));
    if (proj) hipLaunchKernelGGL(k_proj, dim3(blocks), dim3(BLOCK), 0, 0, pts, k, n, out_a, scratch);
    else hipLaunchKernelGGL(k_affine, dim3(blocks), dim3(BLOCK), 0, 0, pts, k, n, out_b, pb, cb, ab);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-7s %8.2f ms\n", proj ? "proj" : "affine", ms);
  }
  CK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL(k_cmp, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, out_a, out_b, n, bad);
  unsigned hb = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
  printf("results that differ as group elements: %u of %zu\n", hb, n);
  return hb ? 2 : 0;
}
