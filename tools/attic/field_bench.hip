// Cycles per fe_mul / fe_sqr per SIMD at 1, 2, 4, 8 waves per SIMD (register-only chains of the hand-written streams).
// Build: hipcc -O3 --offload-arch=gfx950 -Idecaf377_amd/csrc tools/field_bench.hip -o tools/field_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "fq29.hpp"
using namespace d377;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

__device__ __forceinline__ fe ld(const uint32_t* p, int t) { fe a; for (int i = 0; i < 9; ++i) a.l[i] = p[i * 256 + t] & MASK29; return a; }
__device__ __forceinline__ void st(uint32_t* p, int t, const fe& a) { for (int i = 0; i < 9; ++i) p[i * 256 + t] = a.l[i]; }

__global__ void __launch_bounds__(256) k_sqr(const uint32_t* in, uint32_t* out) {
  fe a = ld(in, threadIdx.x);
#pragma unroll 1
  for (int i = 0; i < ITERS; ++i) a = fe_sqr(a);
  st(out + blockIdx.x % 2 * 9 * 256, threadIdx.x, a);
}
__global__ void __launch_bounds__(256) k_sqr2(const uint32_t* in, uint32_t* out) {   // 2 a^2 (fe_sqr2x)
  fe a = ld(in, threadIdx.x);
#pragma unroll 1
  for (int i = 0; i < ITERS; ++i) a = fe_sqr2x(a);
  st(out + blockIdx.x % 2 * 9 * 256, threadIdx.x, a);
}
__global__ void __launch_bounds__(256) k_mul_strict(const uint32_t* in, uint32_t* out) {
  fe a = ld(in, threadIdx.x), b = ld(in + 9 * 256, threadIdx.x);
#pragma unroll 1
  for (int i = 0; i < ITERS; ++i) a = fe_mul_strict(a, b);
  st(out + blockIdx.x % 2 * 9 * 256, threadIdx.x, a);
}
__global__ void __launch_bounds__(256) k_mul(const uint32_t* in, uint32_t* out) {
  fe a = ld(in, threadIdx.x), b = ld(in + 9 * 256, threadIdx.x);
#pragma unroll 1
  for (int i = 0; i < ITERS; ++i) a = fe_mul(a, b);
  st(out + blockIdx.x % 2 * 9 * 256, threadIdx.x, a);
}
__global__ void __launch_bounds__(256) k_sub(const uint32_t* in, uint32_t* out) {
  fe a = ld(in, threadIdx.x), b = ld(in + 9 * 256, threadIdx.x);
#pragma unroll 1
  for (int i = 0; i < ITERS * 4; ++i) a = fe_sub(a, b);
  st(out + blockIdx.x % 2 * 9 * 256, threadIdx.x, a);
}
typedef void (*kern_t)(const uint32_t*, uint32_t*);
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  uint32_t *in, *out; CK(hipMalloc(&in, 18 * 256 * 4)); CK(hipMalloc(&out, 18 * 256 * 4));
  CK(hipMemset(in, 0x5a, 18 * 256 * 4));
  struct { const char* n; kern_t k; double ops; } ks[] = {{"fe_sqr (168 instr)", k_sqr, ITERS}, {"fe_sqr2x (169)", k_sqr2, ITERS},
                                                          {"fe_mul (196)", k_mul, ITERS}, {"fe_mul_strict (205)", k_mul_strict, ITERS},
                                                          {"fe_sub (45)", k_sub, ITERS * 4.0}};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("cycles per operation per SIMD at 2.3 GHz (wave-ops: one op for 64 lanes)\n%-20s", "op \\ waves/SIMD");
  for (int w : {1, 2, 4, 8}) printf("   w=%d", w);
  printf("\n");
  for (auto& k : ks) {
    printf("%-20s", k.n);
    for (int w : {1, 2, 4, 8}) {
      hipLaunchKernelGGL(k.k, dim3(cus * w), dim3(256), 0, 0, in, out); CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k.k, dim3(cus * w), dim3(256), 0, 0, in, out); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double cyc = best * 1e-3 * 2.3e9 / (k.ops * w);
      printf(" %6.0f", cyc);
    }
    printf("\n");
  }
  return 0;
}
