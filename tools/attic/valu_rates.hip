// Register-only VALU issue-rate micro-benchmark for gfx950 (MI355X).
// Gives the measured denominator for the integer-multiply roofline
// (SURVEY.md section 8d): ops/s chip-wide for each candidate instruction.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 65536;
constexpr int UNROLL = 8;   // independent chains per lane

// 32-bit ops: d = op(a, b) or d = op(a, b, d)
#define KERNEL32(NAME, ASM)                                                        \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {        \
  uint32_t a = seed * (threadIdx.x + 1) | 1u, b = seed ^ (threadIdx.x * 2654435761u); \
  uint32_t r[UNROLL]; uint64_t acc64[2] = {a, b};                                  \
  _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) r[k] = a + k;                 \
  for (int i = 0; i < ITERS; ++i) {                                                \
    _Pragma("unroll") for (int k = 0; k < UNROLL; ++k)                             \
      asm volatile(ASM : "+v"(r[k]), "+v"(acc64[k & 1]) : "v"(a), "v"(b) : "vcc");      \
  }                                                                                \
  uint32_t s = 0;                                                                  \
  _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) s ^= r[k];                    \
  s ^= (uint32_t)(acc64[0] ^ acc64[1]);                                            \
  if (s == 0x12345678u) out[threadIdx.x] = s;                                      \
}

KERNEL32(k_add_u32,        "v_add_u32 %0, %2, %0")
KERNEL32(k_fma_f32,        "v_fma_f32 %0, %2, %3, %0")
KERNEL32(k_mul_lo_u32,     "v_mul_lo_u32 %0, %2, %0")
KERNEL32(k_mul_hi_u32,     "v_mul_hi_u32 %0, %2, %0")
KERNEL32(k_mad_u32_u24,    "v_mad_u32_u24 %0, %2, %3, %0")
KERNEL32(k_mul_u32_u24,    "v_mul_u32_u24 %0, %2, %0")
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %2, %0")
KERNEL32(k_addc,           "v_add_co_u32 %0, vcc, %2, %0\n\tv_addc_co_u32 %0, vcc, %3, %0, vcc")
KERNEL32(k_dot4_u32_u8,    "v_dot4_u32_u8 %0, %2, %3, %0")
KERNEL32(k_dot2_u32_u16,   "v_dot2_u32_u16 %0, %2, %3, %0")
KERNEL32(k_pk_mad_u16,     "v_pk_mad_u16 %0, %2, %3, %0")
KERNEL32(k_mad_u16,        "v_mad_u16 %0, %2, %3, %0")
KERNEL32(k_lshl_or,        "v_lshl_or_b32 %0, %2, 3, %0")
KERNEL32(k_alignbit,       "v_alignbit_b32 %0, %2, %0, 29")
KERNEL32(k_and_or,         "v_and_or_b32 %0, %2, %3, %0")
KERNEL32(k_add3,           "v_add3_u32 %0, %2, %3, %0")
KERNEL32(k_cndmask,        "v_cndmask_b32 %0, %2, %0, vcc")
KERNEL32(k_cmp_cndmask,    "v_cmp_lt_u32 vcc, %2, %0\n\tv_cndmask_b32 %0, %3, %0, vcc")
KERNEL32(k_mad64_carry,    "v_mad_u64_u32 %1, vcc, %2, %0, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc")
KERNEL32(k_sub_co,         "v_sub_co_u32 %0, vcc, %0, %2\n\tv_subb_co_u32 %0, vcc, %0, %3, vcc")

// 64-bit accumulators
#define KERNEL64(NAME, ASM)                                                        \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {        \
  uint32_t a = seed * (threadIdx.x + 1) | 1u, b = seed ^ (threadIdx.x * 2654435761u); \
  uint64_t r[UNROLL];                                                              \
  uint64_t a64 = ((uint64_t)a << 20) | b;                                          \
  _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) r[k] = a + k;                 \
  for (int i = 0; i < ITERS; ++i) {                                                \
    _Pragma("unroll") for (int k = 0; k < UNROLL; ++k)                             \
      asm volatile(ASM : "+v"(r[k]) : "v"(a), "v"(b), "v"(a64) : "vcc");           \
  }                                                                                \
  uint64_t s = 0;                                                                  \
  _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) s ^= r[k];                    \
  if (s == 0x12345678u) out[threadIdx.x] = (uint32_t)s;                            \
}

KERNEL64(k_mad_u64_u32,  "v_mad_u64_u32 %0, vcc, %1, %2, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %3, 0, %0")
KERNEL64(k_lshrrev_b64,  "v_lshrrev_b64 %0, 1, %0")

// f64
#define KERNELF64(NAME, ASM)                                                       \
__global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {        \
  double a = 1.0 + 1e-9 * (double)(seed + threadIdx.x), b = 1e-12 * threadIdx.x;   \
  double r[UNROLL];                                                                \
  _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) r[k] = a + k;                 \
  for (int i = 0; i < ITERS; ++i) {                                                \
    _Pragma("unroll") for (int k = 0; k < UNROLL; ++k)                             \
      asm volatile(ASM : "+v"(r[k]) : "v"(a), "v"(b));                             \
  }                                                                                \
  double s = 0;                                                                    \
  _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) s += r[k];                    \
  if (s == 0.12345678) out[threadIdx.x] = 1;                                       \
}
KERNELF64(k_fma_f64, "v_fma_f64 %0, %1, %0, %2")
KERNELF64(k_add_f64, "v_add_f64 %0, %1, %0")
KERNELF64(k_mul_f64, "v_mul_f64 %0, %1, %0")


__global__ void __launch_bounds__(256) k_clock(uint64_t* out, uint32_t seed) {
  uint32_t a = seed * (threadIdx.x + 1) | 1u;
  uint32_t r[UNROLL]; uint64_t acc = a;
  _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) r[k] = a + k;
  uint64_t t0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < ITERS; ++i) {
    _Pragma("unroll") for (int k = 0; k < UNROLL; ++k)
      asm volatile("v_mad_u64_u32 %1, vcc, %2, %0, %1\n\tv_add_u32 %0, %2, %0" : "+v"(r[k]), "+v"(acc) : "v"(a) : "vcc");
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
  uint32_t s = 0;
  _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) s ^= r[k];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = rt1 - rt0; out[2] = s ^ acc; }
}
typedef void (*kern_t)(uint32_t*, uint32_t);
struct Entry { const char* name; kern_t k; int ops_per_asm; };

int main(int argc, char** argv) {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  int cus = p.multiProcessorCount;
  printf("device %s CUs %d clock %d kHz\n", p.name, cus, p.clockRate);
  uint32_t* out; CK(hipMalloc(&out, 4096));
  std::vector<Entry> es = {
    {"v_add_u32", k_add_u32, 1}, {"v_fma_f32", k_fma_f32, 1},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mad_u64_u32", k_mad_u64_u32, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1},
    {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1},
    {"v_add_co+v_addc_co (2 instr)", k_addc, 2},
    {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1},
    {"v_dot4_u32_u8", k_dot4_u32_u8, 1}, {"v_dot2_u32_u16", k_dot2_u32_u16, 1},
    {"v_pk_mad_u16", k_pk_mad_u16, 1}, {"v_mad_u16", k_mad_u16, 1},
    {"v_lshl_or_b32", k_lshl_or, 1}, {"v_alignbit_b32", k_alignbit, 1},
    {"v_and_or_b32", k_and_or, 1}, {"v_add3_u32", k_add3, 1}, {"v_cndmask_b32", k_cndmask, 1}, {"v_cmp+v_cndmask (2)", k_cmp_cndmask, 2}, {"mad_u64_u32+addc (2)", k_mad64_carry, 2}, {"sub_co+subb_co (2)", k_sub_co, 2},
    {"v_fma_f64", k_fma_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_mul_f64", k_mul_f64, 1},
  };
  {
    uint64_t* cb; CK(hipMalloc(&cb, 64));
    for (int w : {1, 4, 8}) {
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_clock, dim3(cus * w), dim3(256), 0, 0, cb, 777u);
      CK(hipDeviceSynchronize());
      uint64_t h[3]; CK(hipMemcpy(h, cb, 24, hipMemcpyDeviceToHost));
      printf("in-kernel clock under mad_u64_u32 load, %d waves/SIMD: %.1f MHz (memtime %llu ticks / memrealtime %llu x10ns)\n", w,
             (double)h[0] / ((double)h[1] / 100.0), (unsigned long long)h[0], (unsigned long long)h[1]);
    }
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // waves per SIMD sweep: blocks of 256 threads = 4 waves = 1 wave per SIMD per block
  int wps_list[] = {1, 2, 4, 8};
  printf("%-32s", "instr \\ waves/SIMD");
  for (int w : wps_list) printf("  w=%d: Ginstr/s  cyc/wave-instr@2.4GHz |", w);
  printf("\n");
  for (auto& e : es) {
    printf("%-32s", e.name);
    for (int w : wps_list) {
      int blocks = cus * w;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);  // warmup
      CK(hipDeviceSynchronize());
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u + rep);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      double lane_instr = (double)blocks * 256 * ITERS * UNROLL * e.ops_per_asm;
      double ginstr = lane_instr / (best * 1e-3) / 1e9;           // lane-instructions per second (G)
      // cycles per wave-instruction per SIMD at nominal 2.4 GHz
      double wave_instr_per_simd = (double)w * ITERS * UNROLL * e.ops_per_asm;
      double cyc = (best * 1e-3) * 2.4e9 / wave_instr_per_simd;
      printf("  %10.1f  %6.2f |", ginstr, cyc);
    }
    printf("\n");
  }
  return 0;
}
