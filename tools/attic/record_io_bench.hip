// record_io_bench.hip -- how should a lane get its 128-byte record?  Micro-benchmark behind the element-wise group
// kernels (k_add, k_double, k_neg: d377.hip), which move 256-384 bytes per element around a handful of field products.
//   lane:   every lane loads its own record with eight global_load_dwordx4 (lane stride 128 B: each instruction touches 64
//           lines and uses 16 bytes of each), and stores the same way
//   staged: a wave loads its 64 records as eight fully coalesced 1 KiB instructions into LDS (XOR-swizzled 16-byte chunks),
//           each lane then reads its own record with ds_read_b128; stores go back the same way
// `work` rounds of a dependent 32-bit multiply-add chain per word stand in for the field arithmetic.
// usage: record_io_bench [log2 n = 22] [work = 0]      prints GB/s (read + written bytes) per variant and stream count
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int BLOCK = 256;

__device__ __forceinline__ void churn(uint4 r[8], int work) {
  for (int w = 0; w < work; ++w)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      r[k].x = r[k].x * 0x9E3779B1u + r[k].y; r[k].y = r[k].y * 0x85EBCA77u + r[k].z;
      r[k].z = r[k].z * 0xC2B2AE3Du + r[k].w; r[k].w = r[k].w * 0x27D4EB2Fu + r[k].x;
    }
}

template <int NIN>
__global__ void __launch_bounds__(BLOCK) k_lane(const uint4* a, const uint4* b, uint4* out, size_t n, int work) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
    uint4 r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = a[8 * i + k];
    if (NIN == 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { const uint4 t = b[8 * i + k]; r[k].x ^= t.x; r[k].y += t.y; r[k].z ^= t.z; r[k].w += t.w; }
    }
    churn(r, work);
#pragma unroll
    for (int k = 0; k < 8; ++k) out[8 * i + k] = r[k];
  }
}

// chunk c (16 bytes) of the wave's 64-record tile lives at LDS chunk (c & ~7) | ((c & 7) ^ ((c >> 3) & 7))
__device__ __forceinline__ int swz(int c) { return (c & ~7) | ((c & 7) ^ ((c >> 3) & 7)); }

template <int NIN>
__global__ void __launch_bounds__(BLOCK) k_staged(const uint4* a, const uint4* b, uint4* out, size_t n, int work) {
  __shared__ uint4 tile[BLOCK / 64][512];                        // 8 KiB per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4* t = tile[wave];
  const size_t nwaves = (size_t)gridDim.x * (BLOCK / 64);
  for (size_t w0 = ((size_t)blockIdx.x * (BLOCK / 64) + wave) * 64; w0 < n; w0 += nwaves * 64) {   // n is a multiple of 64 here
    uint4 r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = a[8 * w0 + k * 64 + lane];          // coalesced: 1 KiB per instruction
#pragma unroll
    for (int k = 0; k < 8; ++k) t[swz(k * 64 + lane)] = r[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = t[swz(lane * 8 + k)];               // own record
    if (NIN == 2) {
      uint4 s[8];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = b[8 * w0 + k * 64 + lane];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[swz(k * 64 + lane)] = s[k];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 8; ++k) { const uint4 q = t[swz(lane * 8 + k)]; r[k].x ^= q.x; r[k].y += q.y; r[k].z ^= q.z; r[k].w += q.w; }
    }
    churn(r, work);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 8; ++k) t[swz(lane * 8 + k)] = r[k];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 8; ++k) out[8 * w0 + k * 64 + lane] = t[swz(k * 64 + lane)];
    __builtin_amdgcn_wave_barrier();
  }
}

template <class F>
static float time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 22;
  const int work = argc > 2 ? atoi(argv[2]) : 0;
  const size_t n = (size_t)1 << lg;
  uint4 *a, *b, *o1, *o2;
  if (hipMalloc(&a, n * 128) != hipSuccess || hipMalloc(&b, n * 128) != hipSuccess || hipMalloc(&o1, n * 128) != hipSuccess ||
      hipMalloc(&o2, n * 128) != hipSuccess) { printf("no device / memory\n"); return 2; }
  hipMemset(a, 0x5A, n * 128); hipMemset(b, 0x3C, n * 128);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  printf("n = 2^%d records of 128 bytes, work = %d\n", lg, work);
  for (int per_cu : {8, 16, 32}) {
    const int grid = prop.multiProcessorCount * per_cu;
    const float l1 = time_ms([&] { hipLaunchKernelGGL(k_lane<1>, dim3(grid), dim3(BLOCK), 0, 0, a, b, o1, n, work); });
    const float s1 = time_ms([&] { hipLaunchKernelGGL(k_staged<1>, dim3(grid), dim3(BLOCK), 0, 0, a, b, o2, n, work); });
    const float l2 = time_ms([&] { hipLaunchKernelGGL(k_lane<2>, dim3(grid), dim3(BLOCK), 0, 0, a, b, o1, n, work); });
    const float s2 = time_ms([&] { hipLaunchKernelGGL(k_staged<2>, dim3(grid), dim3(BLOCK), 0, 0, a, b, o2, n, work); });
    printf("grid %5d: 1 in + 1 out  lane %7.3f ms %6.0f GB/s   staged %7.3f ms %6.0f GB/s | 2 in + 1 out  lane %7.3f ms %6.0f GB/s   staged %7.3f ms %6.0f GB/s\n",
           grid, l1, n * 256 / (l1 * 1e6), s1, n * 256 / (s1 * 1e6), l2, n * 384 / (l2 * 1e6), s2, n * 384 / (s2 * 1e6));
  }
  // same bytes out of both forms?
  hipLaunchKernelGGL(k_lane<2>, dim3(1024), dim3(BLOCK), 0, 0, a, b, o1, n, work);
  hipLaunchKernelGGL(k_staged<2>, dim3(1024), dim3(BLOCK), 0, 0, a, b, o2, n, work);
  hipDeviceSynchronize();
  uint4 h1[16], h2[16];
  hipMemcpy(h1, o1 + 8 * 12345, sizeof h1, hipMemcpyDeviceToHost); hipMemcpy(h2, o2 + 8 * 12345, sizeof h2, hipMemcpyDeviceToHost);
  bool same = true;
  for (int i = 0; i < 16; ++i) same &= h1[i].x == h2[i].x && h1[i].y == h2[i].y && h1[i].z == h2[i].z && h1[i].w == h2[i].w;
  printf("outputs agree: %s\n", same ? "yes" : "NO");
  return same ? 0 : 1;
}
