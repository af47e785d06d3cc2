// Experiment (round 4, review item 6): does the table traffic of the variable-base window loop cost clock?
// k_scalar_mul_var keeps a 9-entry window table per lane in global scratch and gathers one entry per window: 62 GB per
// 2^22-element launch for 0.4 GB of algorithmic bytes, while the chip sustains 2.2-2.3 GHz under it.  Here the same loop
// (table build + 63 windows of 4 doublings and one addition, no square roots) runs twice, alternating on one box:
//   gathers   the table in global scratch, as in the product kernel
//   registers the table replaced by register-resident stand-ins (same arithmetic, no table stores or loads)
// and every workgroup reads the shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime) around its
// work, so the run itself reports the frequency the chip sustained; tools/clock_vs_traffic.sh adds GRBM_GUI_ACTIVE per
// second from rocprofv3.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Idecaf377_amd/csrc tools/clock_vs_traffic.hip -o tools/clock_vs_traffic
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
// no memory behind it: two entries held in registers stand in for the nine (which one depends on the digit, so that the
// compiler cannot fold the choice away); the arithmetic of the loop is unchanged
struct RegTab {
  gec e0, e1;
  bool have = false;
  __device__ __forceinline__ void store(int j, const gec& c) { if (j & 1) e1 = c; else e0 = c; }
  __device__ __forceinline__ gec load(int j, bool swap) const {
    gec c = (j & 1) ? e1 : e0;
    if (swap) { const fe t = c.ypx; c.ypx = c.ymx; c.ymx = t; }
    return c;
  }
};
__device__ __forceinline__ ge load_pt(const uint32_t* p) { ge g; g.x = slot_load(p); g.y = slot_load(p + SLOT); g.z = slot_load(p + 2 * SLOT); g.t = slot_load(p + 3 * SLOT); return g; }
__device__ __forceinline__ void store_pt(uint32_t* p, const ge& g) { slot_store(p, g.x); slot_store(p + SLOT, g.y); slot_store(p + 2 * SLOT, g.z); slot_store(p + 3 * SLOT, g.t); }

template <bool GATHER>
__global__ void __launch_bounds__(BLOCK, 2) k_loop(const uint32_t* pts, const uint8_t* scalar32, size_t n, uint32_t* out, uint32_t* scratch,
                                                  unsigned long long* clocks) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  const size_t nthreads = (size_t)gridDim.x * BLOCK, tid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  for (size_t i = tid; i < n; i += nthreads) {
    uint32_t k[8], dg[8];
    load32(scalar32, i, k);
    fr_reduce_words(k);
    fr_recode_signed16(k, dg);
    ge g = load_pt(pts + i * 48);
    if (GATHER) {
      GlobalTab tab; tab.base = scratch; tab.nthreads = nthreads; tab.tid = tid;
      store_pt(out + i * 48, ge_scalar_mul_w4(g, dg, tab));
    } else {
      RegTab tab;
      store_pt(out + i * 48, ge_scalar_mul_w4(g, dg, tab));
    }
  }
  if (threadIdx.x == 0) {
    atomicAdd(&clocks[0], clock64() - c0);            // shader clock ticks over the workgroup's life
    atomicAdd(&clocks[1], wall_clock64() - w0);       // 100 MHz ticks over the same span
  }
}
__global__ void k_fill(uint32_t* pts, size_t n) {
  size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  ge g = ge_generator();
  for (int j = 0; j < (int)(i % 5); ++j) g = ge_double(g);
  store_pt(pts + i * 48, g);
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const size_t n = (size_t)1 << 22;
  const int blocks = p.multiProcessorCount * 2;
  uint32_t *pts, *out, *scratch; uint8_t* k; unsigned long long* clocks;
  CK(hipMalloc(&pts, n * 192)); CK(hipMalloc(&out, n * 192)); CK(hipMalloc(&k, n * 32)); CK(hipMalloc(&clocks, 16));
  CK(hipMalloc(&scratch, (size_t)blocks * BLOCK * VB_ENTRIES * VB_ENTRY_WORDS * 4));
  hipLaunchKernelGGL(k_fill, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, pts, n);
  std::vector<uint8_t> hk(n * 32); uint64_t s = 88172645463325252ull;
  for (auto& b : hk) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; b = (uint8_t)s; }
  CK(hipMemcpy(k, hk.data(), n * 32, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("window loop of the variable-base multiplication, 2^22 elements, 2 waves per SIMD, %d CUs; alternating\n", p.multiProcessorCount);
  for (int r = 0; r < 8; ++r) {
    const bool gather = (r & 1) == 0;
    CK(hipMemset(clocks, 0, 16));
    CK(hipEventRecord(e0));
    if (gather) hipLaunchKernelGGL(k_loop<true>, dim3(blocks), dim3(BLOCK), 0, 0, pts, k, n, out, scratch, clocks);
    else hipLaunchKernelGGL(k_loop<false>, dim3(blocks), dim3(BLOCK), 0, 0, pts, k, n, out, scratch, clocks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clocks, 16, hipMemcpyDeviceToHost));
    uint32_t chk[2]; CK(hipMemcpy(chk, out + 48 * 12345, 8, hipMemcpyDeviceToHost));
    printf("%-9s %8.2f ms   s_memtime / s_memrealtime = %.4f  -> %7.1f MHz if s_memtime counts shader cycles   (check %08x)\n",
           gather ? "gathers" : "registers", ms, (double)h[0] / (double)h[1], (double)h[0] / (double)h[1] * 100.0, chk[0]);
  }
  return 0;
}
