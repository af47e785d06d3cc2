// Do integer-MAC waves and FP64-FMA waves on one SIMD overlap, or share one pipe?
// Blocks of 512 threads (2 waves per SIMD): variant "int" = all waves v_mad_u64_u32, "f64" = all
// waves v_fma_f64, "mix" = waves 0-3 integer, waves 4-7 FP64 (one of each per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 32768, U = 8;
__device__ __forceinline__ void run_int(uint32_t* out, uint32_t seed) {
  uint32_t a = seed * (threadIdx.x + 1) | 1u, b = seed ^ (threadIdx.x * 2654435761u);
  uint64_t r[U];
  for (int k = 0; k < U; ++k) r[k] = a + k;
  for (int i = 0; i < ITERS; ++i)
#pragma unroll
    for (int k = 0; k < U; ++k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b) : "vcc");
  uint64_t s = 0; for (int k = 0; k < U; ++k) s ^= r[k];
  if (s == 0x1234) out[threadIdx.x] = (uint32_t)s;
}
__device__ __forceinline__ void run_f64(uint32_t* out, uint32_t seed) {
  double a = 1.0 + 1e-9 * (seed + threadIdx.x), b = 1e-12 * threadIdx.x;
  double r[U];
  for (int k = 0; k < U; ++k) r[k] = a + k;
  for (int i = 0; i < ITERS; ++i)
#pragma unroll
    for (int k = 0; k < U; ++k) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(r[k]) : "v"(a), "v"(b));
  double s = 0; for (int k = 0; k < U; ++k) s += r[k];
  if (s == 0.1234) out[threadIdx.x] = 1;
}
__device__ __forceinline__ void run_f32(uint32_t* out, uint32_t seed) {
  float a = 1.0f + 1e-6f * (seed + threadIdx.x), b = 1e-7f * threadIdx.x;
  float r[U];
  for (int k = 0; k < U; ++k) r[k] = a + k;
  for (int i = 0; i < ITERS; ++i)
#pragma unroll
    for (int k = 0; k < U; ++k) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(r[k]) : "v"(a), "v"(b));
  float s = 0; for (int k = 0; k < U; ++k) s += r[k];
  if (s == 0.1234f) out[threadIdx.x] = 1;
}
// mode bits per half: 0 = idle, 1 = int, 2 = f64, 3 = f32
__global__ void __launch_bounds__(512) k(uint32_t* out, uint32_t seed, int lo_mode, int hi_mode) {
  const int mode = (threadIdx.x < 256) ? lo_mode : hi_mode;     // wave-uniform
  if (mode == 1) run_int(out, seed); else if (mode == 2) run_f64(out, seed); else if (mode == 3) run_f32(out, seed);
}
int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); int cus = p.multiProcessorCount;
  uint32_t* out; CK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* nm[4] = {"idle", "int(mad_u64_u32)", "f64(fma)", "f32(fma)"};
  int cfg[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 2}, {1, 3}, {2, 3}};
  for (auto& c : cfg) {
    hipLaunchKernelGGL(k, dim3(cus), dim3(512), 0, 0, out, 7u, c[0], c[1]); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(cus), dim3(512), 0, 0, out, 7u, c[0], c[1]); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("waves0-3: %-18s waves4-7: %-18s  %8.3f ms\n", nm[c[0]], nm[c[1]], best);
  }
  return 0;
}
