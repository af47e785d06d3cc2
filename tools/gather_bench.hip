// gather_bench.hip -- what does the memory system give RANDOM small records?  The fixed-base kernel reads 11 random 128-byte comb
// records per scalar from a 5.9 GB table (2.0 TB/s at 2^20 scalars per 0.75 ms); this harness reads random records and does
// nothing else, to see where that stands against the rate such reads can have at all.
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/gather_bench.hip -o tools/gather_bench
//   run:   tools/gather_bench                      -> table on stdout (profiles/rNN_gather_bench.txt)
// Per lane: `inflight` independent records requested, then consumed (xor into a checksum), `rounds` times; every lane its own
// random stream (xorshift).  Record sizes 64 and 128 bytes (4 or 7-8 x 16-byte loads per lane: every load instruction has the
// 64 lanes in 64 different lines) and, for 128 bytes, the cooperative form (8 lanes per record, one line per 8 lanes and
// instruction).  Occupancy as the fixed-base kernel's: 2 or 3 workgroups of 256 per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xs(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

template <int REC_BYTES, int INFLIGHT>
__global__ void __launch_bounds__(256) k_gather(const uint4* table, uint32_t mask, int rounds, uint32_t* sink) {
  constexpr int CH = REC_BYTES / 16;
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int r = 0; r < rounds; ++r) {
    uint4 v[INFLIGHT][CH];
#pragma unroll
    for (int f = 0; f < INFLIGHT; ++f) {
      const size_t rec = xs(s) & mask;
      const uint4* p = table + rec * CH;
#pragma unroll
      for (int c = 0; c < CH; ++c) v[f][c] = p[c];
    }
#pragma unroll
    for (int f = 0; f < INFLIGHT; ++f)
#pragma unroll
      for (int c = 0; c < CH; ++c) { acc.x ^= v[f][c].x; acc.y += v[f][c].y; acc.z ^= v[f][c].z; acc.w += v[f][c].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

// 128-byte records, the wave cooperating: instruction k fetches the records of lanes 8k .. 8k + 7 whole, through LDS
__global__ void __launch_bounds__(256) k_gather_coop(const uint4* table, uint32_t mask, int rounds, uint32_t* sink) {
  __shared__ uint4 tile[4][512];
  const int lane = threadIdx.x & 63;
  uint4* t = tile[threadIdx.x >> 6];
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int sub = lane >> 3, c = (lane & 7) ^ sub;
  for (int r = 0; r < rounds; ++r) {
    const uint32_t rec = xs(s) & mask;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t rr = (uint32_t)__shfl((int)rec, 8 * k + sub);
      const uint4* src = table + (size_t)rr * 8 + c;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(t + k * 64), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) { const uint4 v = t[lane * 8 + (q ^ (lane & 7))]; acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

template <class F>
static double time_ms(F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  uint32_t* sink; CHECK(hipMalloc(&sink, 64));
  printf("%s, %d CUs; random records, every lane its own stream; GB/s of record bytes and records/s\n", prop.gcnArchName, cus);
  for (int lg : {21, 26, 29, 32}) {                    // table bytes: 2 MiB (L2), 64 MiB (Infinity Cache), 512 MiB, 4 GiB
    const size_t bytes = (size_t)1 << lg;
    uint4* table; CHECK(hipMalloc(&table, bytes));
    CHECK(hipMemset(table, 1, bytes));
    for (int wg_per_cu : {2, 3}) {
      const int grid = cus * wg_per_cu;
      const int rounds = 256;
      auto report = [&](const char* name, int rec, int inflight, double ms) {
        const double recs = (double)grid * 256 * rounds * inflight;
        printf("  table 2^%d B  %d WG/CU  %-28s %8.3f ms  %7.1f GB/s  %6.2f G records/s\n", lg, wg_per_cu, name, ms, recs * rec / ms / 1e6, recs / ms / 1e6);
      };
      const uint32_t m128 = (uint32_t)(bytes / 128 - 1), m64 = (uint32_t)(bytes / 64 - 1);
      report("128 B, 1 in flight per lane", 128, 1, time_ms([&] { hipLaunchKernelGGL((k_gather<128, 1>), dim3(grid), dim3(256), 0, 0, table, m128, rounds, sink); }));
      report("128 B, 2 in flight per lane", 128, 2, time_ms([&] { hipLaunchKernelGGL((k_gather<128, 2>), dim3(grid), dim3(256), 0, 0, table, m128, rounds, sink); }));
      report("128 B, 4 in flight per lane", 128, 4, time_ms([&] { hipLaunchKernelGGL((k_gather<128, 4>), dim3(grid), dim3(256), 0, 0, table, m128, rounds, sink); }));
      report("128 B, wave-cooperative", 128, 1, time_ms([&] { hipLaunchKernelGGL(k_gather_coop, dim3(grid), dim3(256), 0, 0, table, m128, rounds, sink); }));
      report("64 B, 1 in flight per lane", 64, 1, time_ms([&] { hipLaunchKernelGGL((k_gather<64, 1>), dim3(grid), dim3(256), 0, 0, table, m64, rounds, sink); }));
      report("64 B, 2 in flight per lane", 64, 2, time_ms([&] { hipLaunchKernelGGL((k_gather<64, 2>), dim3(grid), dim3(256), 0, 0, table, m64, rounds, sink); }));
      report("64 B, 4 in flight per lane", 64, 4, time_ms([&] { hipLaunchKernelGGL((k_gather<64, 4>), dim3(grid), dim3(256), 0, 0, table, m64, rounds, sink); }));
    }
    CHECK(hipFree(table));
  }
  return 0;
}
