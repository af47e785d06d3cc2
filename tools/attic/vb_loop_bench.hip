// Experiment: the variable-base window loop alone (table build + 63 windows, no square roots, no LDS) at 2, 3, 4
// waves per SIMD -- would splitting k_scalar_mul_var into decompress / loop / compress kernels pay?
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Idecaf377_amd/csrc tools/vb_loop_bench.hip -o tools/vb_loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "curve.hpp"
#include "device_util.hpp"
using namespace d377;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

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

template <int WAVES>
__global__ void __launch_bounds__(BLOCK, WAVES) k_loop(const uint32_t* pts, const uint8_t* scalar32, size_t n, uint32_t* out, uint32_t* scratch) {
  GlobalTab tab; tab.base = scratch; tab.nthreads = (size_t)gridDim.x * BLOCK; tab.tid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  for (size_t i = tab.tid; i < n; i += tab.nthreads) {
    uint32_t k[8], dg[8];
    load32(scalar32, i, k);
    fr_reduce_words(k);
    fr_recode_signed16(k, dg);
    ge g = load_pt(pts + i * 48);
    store_pt(out + i * 48, ge_scalar_mul_w4(g, dg, tab));
  }
}
__global__ void k_fill(uint32_t* pts, size_t n) {   // generator multiples as inputs (any curve points do)
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  ge g = ge_generator();
  for (int j = 0; j < (int)(i % 5); ++j) g = ge_double(g);
  store_pt(pts + i * 48, g);
}
template <int WAVES> int run(int cus, const uint32_t* pts, const uint8_t* k, size_t n, uint32_t* out) {
  uint32_t* scratch;
  const int blocks = cus * WAVES;
  CK(hipMalloc(&scratch, (size_t)blocks * BLOCK * VB_ENTRIES * VB_ENTRY_WORDS * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_loop<WAVES>, dim3(blocks), dim3(BLOCK), 0, 0, pts, k, n, out, scratch); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_loop<WAVES>, dim3(blocks), dim3(BLOCK), 0, 0, pts, k, n, out, scratch); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  uint32_t h[4]; CK(hipMemcpy(h, out + 48 * 12345, 16, hipMemcpyDeviceToHost));
  printf("window loop alone, %d waves/SIMD: %.2f ms per 2^22 (check %08x %08x)\n", WAVES, best, h[0], h[1]);
  CK(hipFree(scratch));
  return 0;
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const size_t n = (size_t)1 << 22;
  uint32_t *pts, *out; uint8_t* k;
  CK(hipMalloc(&pts, n * 192)); CK(hipMalloc(&out, n * 192)); CK(hipMalloc(&k, n * 32));
  hipLaunchKernelGGL(k_fill, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, pts, n);
  std::vector<uint8_t> hk(n * 32); uint64_t s = 88172645463325252ull;
  for (auto& b : hk) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; b = (uint8_t)s; }
  CK(hipMemcpy(k, hk.data(), n * 32, hipMemcpyHostToDevice));
  if (run<2>(p.multiProcessorCount, pts, k, n, out)) return 1;
  if (run<3>(p.multiProcessorCount, pts, k, n, out)) return 1;
  if (run<4>(p.multiProcessorCount, pts, k, n, out)) return 1;
  return 0;
}
