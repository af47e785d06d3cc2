// VALU issue-cost micro-benchmark, second edition (round 2): settles what the integer-MAC roofline
// denominator is and which instruction classes are cheap on gfx950.
//   * every body is 32 instructions long and the loop is unrolled x4 (128 instructions per backward
//     branch), so loop overhead cannot inflate the per-instruction figure (round 1 used 8 per branch);
//   * cycles come from s_memtime inside the kernel (wave 0 of block 0), i.e. real shader cycles at the
//     clock the chip actually sustained, not from wall time and a nominal clock;
//   * operand-kind variants of v_mad_u64_u32 (VGPR / SGPR / inline-constant src; gfx9 VOP3 takes no literal), dependent vs independent
//     accumulator chains, and MAC + cheap-op mixes in one wave.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_mix.hip -o tools/valu_mix
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;   // x 4 x 32 = 262144 instructions per wave

// registers every body may use: 8 x u64 accumulators A0..A7 (%0..%7), 8 x u32 R0..R7 (%8..%15),
// inputs a (%16, VGPR), b (%17, VGPR), s (%18, SGPR)
#define OPS                                                                                           \
  "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]),     \
  "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7])      \
  : "v"(a), "v"(b), "s"(s) : "vcc"

#define KERNEL(NAME, BODY)                                                                            \
  __global__ void __launch_bounds__(256) NAME(uint64_t* out, uint32_t seed, uint32_t sval) {          \
    uint32_t a = (seed * (threadIdx.x + 1)) | 1u, b = seed ^ (threadIdx.x * 2654435761u);            \
    uint32_t s = sval;                                                                                \
    uint64_t A[8]; uint32_t R[8];                                                                     \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) { A[k] = a + k; R[k] = b + k; }                     \
    const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                 \
    for (int i = 0; i < ITERS; ++i) {                                                                 \
      asm volatile(BODY : OPS); asm volatile(BODY : OPS); asm volatile(BODY : OPS); asm volatile(BODY : OPS); \
    }                                                                                                 \
    const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                 \
    uint64_t x = 0;                                                                                   \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) x ^= A[k] ^ R[k];                                   \
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; }                                    \
    if (x == 0x123456789abcull) out[1 + threadIdx.x] = x;                                             \
  }

// 8 instructions over the 8 accumulators / registers, then x4 = 32
#define X4(S) S S S S
#define MAC8(SRC0, SRC1)                                                                              \
  "v_mad_u64_u32 %0, vcc, " SRC0 ", " SRC1 ", %0\n v_mad_u64_u32 %1, vcc, " SRC0 ", " SRC1 ", %1\n"   \
  "v_mad_u64_u32 %2, vcc, " SRC0 ", " SRC1 ", %2\n v_mad_u64_u32 %3, vcc, " SRC0 ", " SRC1 ", %3\n"   \
  "v_mad_u64_u32 %4, vcc, " SRC0 ", " SRC1 ", %4\n v_mad_u64_u32 %5, vcc, " SRC0 ", " SRC1 ", %5\n"   \
  "v_mad_u64_u32 %6, vcc, " SRC0 ", " SRC1 ", %6\n v_mad_u64_u32 %7, vcc, " SRC0 ", " SRC1 ", %7\n"
#define OP8_2(OP, SRC)                                                                                \
  OP " %8, " SRC ", %8\n " OP " %9, " SRC ", %9\n " OP " %10, " SRC ", %10\n " OP " %11, " SRC ", %11\n" \
  OP " %12, " SRC ", %12\n " OP " %13, " SRC ", %13\n " OP " %14, " SRC ", %14\n " OP " %15, " SRC ", %15\n"
#define OP8_3(OP, S0, S1)                                                                             \
  OP " %8, " S0 ", " S1 ", %8\n " OP " %9, " S0 ", " S1 ", %9\n " OP " %10, " S0 ", " S1 ", %10\n "   \
  OP " %11, " S0 ", " S1 ", %11\n " OP " %12, " S0 ", " S1 ", %12\n " OP " %13, " S0 ", " S1 ", %13\n " \
  OP " %14, " S0 ", " S1 ", %14\n " OP " %15, " S0 ", " S1 ", %15\n"
#define OP8_64(OP, SH)                                                                                \
  OP " %0, " SH ", %0\n " OP " %1, " SH ", %1\n " OP " %2, " SH ", %2\n " OP " %3, " SH ", %3\n"      \
  OP " %4, " SH ", %4\n " OP " %5, " SH ", %5\n " OP " %6, " SH ", %6\n " OP " %7, " SH ", %7\n"

// ---- v_mad_u64_u32 operand kinds, independent accumulators
KERNEL(k_mac_vv, X4(MAC8("%16", "%17")))
KERNEL(k_mac_vs, X4(MAC8("%16", "%18")))
KERNEL(k_mac_sv, X4(MAC8("%18", "%17")))
KERNEL(k_mac_vinl, X4(MAC8("%16", "1")))
// accumulator-dependent operand (as in the m*q products: src0 = low word of another accumulator's past value)
KERNEL(k_mac_rr, X4(
  "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %10, %11, %1\n v_mad_u64_u32 %2, vcc, %12, %13, %2\n"
  "v_mad_u64_u32 %3, vcc, %14, %15, %3\n v_mad_u64_u32 %4, vcc, %9, %10, %4\n v_mad_u64_u32 %5, vcc, %11, %12, %5\n"
  "v_mad_u64_u32 %6, vcc, %13, %14, %6\n v_mad_u64_u32 %7, vcc, %15, %8, %7\n"))
// ---- dependent chains: 1, 2, 4 accumulators
KERNEL(k_mac_dep1, X4(
  "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %0, vcc, %10, %11, %0\n v_mad_u64_u32 %0, vcc, %12, %13, %0\n"
  "v_mad_u64_u32 %0, vcc, %14, %15, %0\n v_mad_u64_u32 %0, vcc, %9, %10, %0\n v_mad_u64_u32 %0, vcc, %11, %12, %0\n"
  "v_mad_u64_u32 %0, vcc, %13, %14, %0\n v_mad_u64_u32 %0, vcc, %15, %8, %0\n"))
KERNEL(k_mac_dep2, X4(
  "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %10, %11, %1\n v_mad_u64_u32 %0, vcc, %12, %13, %0\n"
  "v_mad_u64_u32 %1, vcc, %14, %15, %1\n v_mad_u64_u32 %0, vcc, %9, %10, %0\n v_mad_u64_u32 %1, vcc, %11, %12, %1\n"
  "v_mad_u64_u32 %0, vcc, %13, %14, %0\n v_mad_u64_u32 %1, vcc, %15, %8, %1\n"))
KERNEL(k_mac_dep4, X4(
  "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %10, %11, %1\n v_mad_u64_u32 %2, vcc, %12, %13, %2\n"
  "v_mad_u64_u32 %3, vcc, %14, %15, %3\n v_mad_u64_u32 %0, vcc, %9, %10, %0\n v_mad_u64_u32 %1, vcc, %11, %12, %1\n"
  "v_mad_u64_u32 %2, vcc, %13, %14, %2\n v_mad_u64_u32 %3, vcc, %15, %8, %3\n"))
KERNEL(k_imac_vv, X4(
  "v_mad_i64_i32 %0, vcc, %16, %17, %0\n v_mad_i64_i32 %1, vcc, %16, %17, %1\n v_mad_i64_i32 %2, vcc, %16, %17, %2\n"
  "v_mad_i64_i32 %3, vcc, %16, %17, %3\n v_mad_i64_i32 %4, vcc, %16, %17, %4\n v_mad_i64_i32 %5, vcc, %16, %17, %5\n"
  "v_mad_i64_i32 %6, vcc, %16, %17, %6\n v_mad_i64_i32 %7, vcc, %16, %17, %7\n"))
// ---- candidate cheap ops (32-bit encodings unless the mnemonic says _e64)
KERNEL(k_add_u32, X4(OP8_2("v_add_u32", "%16")))
KERNEL(k_add_u32_e64, X4(OP8_2("v_add_u32_e64", "%16")))
KERNEL(k_add_u32_lit, X4(OP8_2("v_add_u32", "0x60000008")))
KERNEL(k_sub_u32, X4(OP8_2("v_sub_u32", "%16")))
KERNEL(k_subrev_u32, X4(OP8_2("v_subrev_u32", "%16")))
KERNEL(k_and_b32, X4(OP8_2("v_and_b32", "%16")))
KERNEL(k_and_b32_lit, X4(OP8_2("v_and_b32", "0x1fffffff")))
KERNEL(k_and_b32_e64, X4(OP8_2("v_and_b32_e64", "%16")))
KERNEL(k_or_b32, X4(OP8_2("v_or_b32", "%16")))
KERNEL(k_xor_b32, X4(OP8_2("v_xor_b32", "%16")))
KERNEL(k_lshrrev_b32, X4(OP8_2("v_lshrrev_b32", "29")))
KERNEL(k_lshlrev_b32, X4(OP8_2("v_lshlrev_b32", "1")))
KERNEL(k_min_u32, X4(OP8_2("v_min_u32", "%16")))
KERNEL(k_mov_b32, X4("v_mov_b32 %8, %9\n v_mov_b32 %9, %10\n v_mov_b32 %10, %11\n v_mov_b32 %11, %12\n"
                     "v_mov_b32 %12, %13\n v_mov_b32 %13, %14\n v_mov_b32 %14, %15\n v_mov_b32 %15, %16\n"))
KERNEL(k_not_b32, X4("v_not_b32 %8, %8\n v_not_b32 %9, %9\n v_not_b32 %10, %10\n v_not_b32 %11, %11\n"
                     "v_not_b32 %12, %12\n v_not_b32 %13, %13\n v_not_b32 %14, %14\n v_not_b32 %15, %15\n"))
KERNEL(k_add_co_u32, X4(OP8_2("v_add_co_u32", "vcc, %16")))
KERNEL(k_addc_co_u32, X4(
  "v_addc_co_u32 %8, vcc, %16, %8, vcc\n v_addc_co_u32 %9, vcc, %16, %9, vcc\n v_addc_co_u32 %10, vcc, %16, %10, vcc\n"
  "v_addc_co_u32 %11, vcc, %16, %11, vcc\n v_addc_co_u32 %12, vcc, %16, %12, vcc\n v_addc_co_u32 %13, vcc, %16, %13, vcc\n"
  "v_addc_co_u32 %14, vcc, %16, %14, vcc\n v_addc_co_u32 %15, vcc, %16, %15, vcc\n"))
KERNEL(k_cndmask, X4(OP8_2("v_cndmask_b32", "%16") ))   // vcc read only (e32: vcc implicit)
KERNEL(k_fma_f32, X4(OP8_3("v_fma_f32", "%16", "%17")))
KERNEL(k_fmac_f32, X4("v_fmac_f32 %8, %16, %17\n v_fmac_f32 %9, %16, %17\n v_fmac_f32 %10, %16, %17\n v_fmac_f32 %11, %16, %17\n"
                      "v_fmac_f32 %12, %16, %17\n v_fmac_f32 %13, %16, %17\n v_fmac_f32 %14, %16, %17\n v_fmac_f32 %15, %16, %17\n"))
KERNEL(k_mul_lo_u32, X4(OP8_2("v_mul_lo_u32", "%16")))
KERNEL(k_mul_hi_u32, X4(OP8_2("v_mul_hi_u32", "%16")))
KERNEL(k_mul_u32_u24, X4(OP8_2("v_mul_u32_u24", "%16")))
KERNEL(k_mad_u32_u24, X4(OP8_3("v_mad_u32_u24", "%16", "%17")))
KERNEL(k_add3_u32, X4(OP8_3("v_add3_u32", "%16", "%17")))
KERNEL(k_and_or_b32, X4(OP8_3("v_and_or_b32", "%16", "%17")))
KERNEL(k_lshl_add_u32, X4(OP8_3("v_lshl_add_u32", "%16", "3")))
KERNEL(k_bfe_u32, X4("v_bfe_u32 %8, %8, 3, 29\n v_bfe_u32 %9, %9, 3, 29\n v_bfe_u32 %10, %10, 3, 29\n v_bfe_u32 %11, %11, 3, 29\n"
                     "v_bfe_u32 %12, %12, 3, 29\n v_bfe_u32 %13, %13, 3, 29\n v_bfe_u32 %14, %14, 3, 29\n v_bfe_u32 %15, %15, 3, 29\n"))
KERNEL(k_alignbit, X4(OP8_3("v_alignbit_b32", "%16", "%17")))
KERNEL(k_lshrrev_b64, X4(OP8_64("v_lshrrev_b64", "29")))
KERNEL(k_ashrrev_i64, X4(OP8_64("v_ashrrev_i64", "29")))
KERNEL(k_lshl_add_u64, X4(
  "v_lshl_add_u64 %0, %1, 0, %0\n v_lshl_add_u64 %1, %2, 0, %1\n v_lshl_add_u64 %2, %3, 0, %2\n v_lshl_add_u64 %3, %4, 0, %3\n"
  "v_lshl_add_u64 %4, %5, 0, %4\n v_lshl_add_u64 %5, %6, 0, %5\n v_lshl_add_u64 %6, %7, 0, %6\n v_lshl_add_u64 %7, %0, 0, %7\n"))
KERNEL(k_pk_add_u16, X4(OP8_2("v_pk_add_u16", "%16")))
// ---- mixes inside one wave: do a MAC and a cheap op cost the sum of their issue costs?
#define MAC4A "v_mad_u64_u32 %0, vcc, %16, %17, %0\n v_mad_u64_u32 %1, vcc, %16, %17, %1\n v_mad_u64_u32 %2, vcc, %16, %17, %2\n v_mad_u64_u32 %3, vcc, %16, %17, %3\n"
#define MAC4B "v_mad_u64_u32 %4, vcc, %16, %17, %4\n v_mad_u64_u32 %5, vcc, %16, %17, %5\n v_mad_u64_u32 %6, vcc, %16, %17, %6\n v_mad_u64_u32 %7, vcc, %16, %17, %7\n"
#define AND4A "v_and_b32 %8, %16, %8\n v_and_b32 %9, %16, %9\n v_and_b32 %10, %16, %10\n v_and_b32 %11, %16, %11\n"
#define AND4B "v_and_b32 %12, %16, %12\n v_and_b32 %13, %16, %13\n v_and_b32 %14, %16, %14\n v_and_b32 %15, %16, %15\n"
// 16 MAC + 16 and, blocked (8 MAC, 8 and, ...)
KERNEL(k_mix_block, MAC4A MAC4B AND4A AND4B MAC4A MAC4B AND4A AND4B)
// alternating MAC, and, MAC, and
#define ALT(K, J) "v_mad_u64_u32 %" #K ", vcc, %16, %17, %" #K "\n v_and_b32 %" #J ", %16, %" #J "\n"
KERNEL(k_mix_alt, X4(ALT(0, 8) ALT(1, 9) ALT(2, 10) ALT(3, 11)) )
// 24 MAC + 8 and (3:1, close to the multiplier's real mix)
KERNEL(k_mix_3to1, MAC4A MAC4B MAC4A AND4A MAC4B MAC4A MAC4B AND4B)
// 16 MAC + 16 v_lshrrev_b64 on other accumulators
KERNEL(k_mix_shift, MAC4A "v_lshrrev_b64 %4, 29, %4\n v_lshrrev_b64 %5, 29, %5\n v_lshrrev_b64 %6, 29, %6\n v_lshrrev_b64 %7, 29, %7\n"
                    MAC4A "v_lshrrev_b64 %4, 29, %4\n v_lshrrev_b64 %5, 29, %5\n v_lshrrev_b64 %6, 29, %6\n v_lshrrev_b64 %7, 29, %7\n"
                    MAC4A "v_lshrrev_b64 %4, 29, %4\n v_lshrrev_b64 %5, 29, %5\n v_lshrrev_b64 %6, 29, %6\n v_lshrrev_b64 %7, 29, %7\n"
                    MAC4A "v_lshrrev_b64 %4, 29, %4\n v_lshrrev_b64 %5, 29, %5\n v_lshrrev_b64 %6, 29, %6\n v_lshrrev_b64 %7, 29, %7\n")
typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t);
struct Entry { const char* name; kern_t k; };

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("device %s CUs %d\n", p.gcnArchName, cus);
  uint64_t* out; CK(hipMalloc(&out, 4096 * 8));
#define E(k) {#k, k}
  std::vector<Entry> es = {
    E(k_mac_vv), E(k_mac_vs), E(k_mac_sv), E(k_mac_vinl), E(k_mac_rr), E(k_mac_dep1), E(k_mac_dep2), E(k_mac_dep4),
    E(k_imac_vv),
    E(k_add_u32), E(k_add_u32_e64), E(k_add_u32_lit), E(k_sub_u32), E(k_subrev_u32), E(k_and_b32), E(k_and_b32_lit), E(k_and_b32_e64),
    E(k_or_b32), E(k_xor_b32), E(k_lshrrev_b32), E(k_lshlrev_b32), E(k_min_u32), E(k_mov_b32), E(k_not_b32), E(k_add_co_u32),
    E(k_addc_co_u32), E(k_cndmask), E(k_fma_f32), E(k_fmac_f32), E(k_mul_lo_u32), E(k_mul_hi_u32), E(k_mul_u32_u24), E(k_mad_u32_u24),
    E(k_add3_u32), E(k_and_or_b32), E(k_lshl_add_u32), E(k_bfe_u32), E(k_alignbit), E(k_lshrrev_b64), E(k_ashrrev_i64),
    E(k_lshl_add_u64), E(k_pk_add_u16),
    E(k_mix_block), E(k_mix_alt), E(k_mix_3to1), E(k_mix_shift),
  };
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int wps[] = {1, 2, 3, 4, 8};
  printf("cycles per wave-instruction per SIMD: in-kernel s_memtime (wave 0) / wall-clock at 2.4 GHz\n%-16s", "kernel");
  for (int w : wps) printf(" |   w=%d  memtime   wall", w);
  printf("\n");
  const double ninstr = (double)ITERS * 4 * 32;
  for (auto& e : es) {
    printf("%-16s", e.name + 2);
    for (int w : wps) {
      const int blocks = cus * w;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u, 0x108c0000u);
      CK(hipDeviceSynchronize());
      float best = 1e30f; uint64_t ticks = ~0ull;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u + rep, 0x108c0000u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        uint64_t t; CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
        if (ms < best) best = ms;
        if (t < ticks) ticks = t;
      }
      printf(" |       %8.2f %6.2f", (double)ticks / (ninstr * w), best * 1e-3 * 2.4e9 / (ninstr * w));
    }
    printf("\n");
  }
  return 0;
}
