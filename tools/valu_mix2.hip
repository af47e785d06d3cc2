// Follow-up to valu_mix.hip: (1) what an s_nop after every dependent MAC costs (hipcc pads every
// consumer of an inline-asm result with one on gfx950), (2) VGPR bank effects on v_mad_u64_u32
// with hard-coded registers.  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_mix2.hip -o tools/valu_mix2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;

#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","vcc"
#define KERNEL(NAME, BODY)                                                                            \
  __global__ void __launch_bounds__(256) NAME(uint64_t* out, uint32_t seed) {                         \
    uint32_t a = (seed * (threadIdx.x + 1)) | 1u;                                                     \
    asm volatile("v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n"    \
                 "v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n"    \
                 "v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n"    \
                 "v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n"    \
                 "v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n"    \
                 "v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n"    \
                 "v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n"    \
                 "v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n"    \
                 "v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n"    \
                 "v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n" :: "v"(a) : CLOB);                         \
    const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                 \
    for (int i = 0; i < ITERS; ++i) {                                                                 \
      asm volatile(BODY ::: CLOB); asm volatile(BODY ::: CLOB); asm volatile(BODY ::: CLOB); asm volatile(BODY ::: CLOB); \
    }                                                                                                 \
    const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                 \
    uint32_t x;                                                                                       \
    asm volatile("v_xor_b32 %0, v10, v12\n v_xor_b32 %0, %0, v14\n v_xor_b32 %0, %0, v16\n v_xor_b32 %0, %0, v11\n" : "=v"(x) :: CLOB); \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                        \
    if (x == 0x12345678u) out[1 + threadIdx.x] = x;                                                   \
  }
#define X4(S) S S S S
#define X8(S) S S S S S S S S
// one dependent chain on v[10:11]; sources walk through registers
// conflict-free: acc banks 2,3 (v10,v11); src0 bank 0 (v12,v16,..), src1 bank 1 (v13,v17,...)
KERNEL(k_dep_free, X4(
  "v_mad_u64_u32 v[10:11], vcc, v12, v13, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v16, v17, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v20, v21, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v24, v25, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v28, v29, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v32, v33, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v36, v37, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v40, v41, v[10:11]\n"))
// same with an s_nop 0 after every MAC (what hipcc emits around the pinned accumulator)
KERNEL(k_dep_free_nop, X4(
  "v_mad_u64_u32 v[10:11], vcc, v12, v13, v[10:11]\n s_nop 0\n v_mad_u64_u32 v[10:11], vcc, v16, v17, v[10:11]\n s_nop 0\n"
  "v_mad_u64_u32 v[10:11], vcc, v20, v21, v[10:11]\n s_nop 0\n v_mad_u64_u32 v[10:11], vcc, v24, v25, v[10:11]\n s_nop 0\n"
  "v_mad_u64_u32 v[10:11], vcc, v28, v29, v[10:11]\n s_nop 0\n v_mad_u64_u32 v[10:11], vcc, v32, v33, v[10:11]\n s_nop 0\n"
  "v_mad_u64_u32 v[10:11], vcc, v36, v37, v[10:11]\n s_nop 0\n v_mad_u64_u32 v[10:11], vcc, v40, v41, v[10:11]\n s_nop 0\n"))
// src0 and src1 in the same bank (bank 0), acc banks 2,3
KERNEL(k_dep_s01, X4(
  "v_mad_u64_u32 v[10:11], vcc, v12, v16, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v16, v20, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v20, v24, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v24, v28, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v28, v32, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v32, v36, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v36, v40, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v40, v12, v[10:11]\n"))
// src0 in the accumulator's low bank (bank 2), src1 in the accumulator's high bank (bank 3)
KERNEL(k_dep_sacc, X4(
  "v_mad_u64_u32 v[10:11], vcc, v14, v15, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v18, v19, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v22, v23, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v26, v27, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v30, v31, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v34, v35, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v38, v39, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v42, v43, v[10:11]\n"))
// everything in two banks: src0 bank 2, src1 bank 2, acc banks 2,3
KERNEL(k_dep_worst, X4(
  "v_mad_u64_u32 v[10:11], vcc, v14, v18, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v18, v22, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v22, v26, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v26, v30, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v30, v34, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v34, v38, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v38, v42, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v42, v14, v[10:11]\n"))
// product-scanning pattern as the compiler emits it: a_i (consecutive regs) x b_j (consecutive regs)
KERNEL(k_dep_scan, X4(
  "v_mad_u64_u32 v[10:11], vcc, v12, v28, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v13, v27, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v14, v26, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v15, v25, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v16, v24, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v17, v23, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v18, v22, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v19, v21, v[10:11]\n"))
// one VGPR source + SGPR source (the m*q terms), sources walk
KERNEL(k_dep_sgpr, X4(
  "v_mad_u64_u32 v[10:11], vcc, v12, s4, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v13, s5, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v14, s6, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v15, s7, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v16, s4, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v17, s5, v[10:11]\n"
  "v_mad_u64_u32 v[10:11], vcc, v18, s6, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v19, s7, v[10:11]\n"))
// two interleaved chains, conflict-free
KERNEL(k_dep2_free, X4(
  "v_mad_u64_u32 v[10:11], vcc, v12, v13, v[10:11]\n v_mad_u64_u32 v[14:15], vcc, v16, v17, v[14:15]\n"
  "v_mad_u64_u32 v[10:11], vcc, v20, v21, v[10:11]\n v_mad_u64_u32 v[14:15], vcc, v24, v25, v[14:15]\n"
  "v_mad_u64_u32 v[10:11], vcc, v28, v29, v[10:11]\n v_mad_u64_u32 v[14:15], vcc, v32, v33, v[14:15]\n"
  "v_mad_u64_u32 v[10:11], vcc, v36, v37, v[10:11]\n v_mad_u64_u32 v[14:15], vcc, v40, v41, v[14:15]\n"))
// the full column tail, dependent: neg, and, mac(m,1), shift
KERNEL(k_col_tail4, X8(
  "v_sub_u32 v12, 0, v10\n v_and_b32 v12, 0x1fffffff, v12\n v_mad_u64_u32 v[10:11], vcc, v12, 1, v[10:11]\n v_lshrrev_b64 v[10:11], 29, v[10:11]\n"))
// proposed: neg (32-bit), mac(m',1), shift
KERNEL(k_col_tail3, X8(
  "v_sub_u32 v12, 0, v10\n v_mad_u64_u32 v[10:11], vcc, v12, 1, v[10:11]\n v_lshrrev_b64 v[10:11], 29, v[10:11]\n v_mad_u64_u32 v[10:11], vcc, v16, v17, v[10:11]\n"))

typedef void (*kern_t)(uint64_t*, uint32_t);
struct Entry { const char* name; kern_t k; double instr_per_body; };
int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  uint64_t* out; CK(hipMalloc(&out, 4096 * 8));
#define E(k, n) {#k, k, n}
  std::vector<Entry> es = {E(k_dep_free, 32), E(k_dep_free_nop, 32), E(k_dep_s01, 32), E(k_dep_sacc, 32), E(k_dep_worst, 32), E(k_dep_scan, 32),
                           E(k_dep_sgpr, 32), E(k_dep2_free, 32), E(k_col_tail4, 32), E(k_col_tail3, 32)};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int wps[] = {1, 2, 3, 4, 8};
  printf("cycles per VALU wave-instruction per SIMD (s_nop not counted): w=1 in-kernel s_memtime; all w: wall-clock at 2.4 GHz\n%-16s  memtime(w=1)", "kernel");
  for (int w : wps) printf("  wall w=%d", w);
  printf("\n");
  for (auto& e : es) {
    const double ninstr = (double)ITERS * 4 * e.instr_per_body;
    printf("%-16s", e.name + 2);
    for (int w : wps) {
      const int blocks = cus * w;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u);
      CK(hipDeviceSynchronize());
      float best = 1e30f; uint64_t ticks = ~0ull;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u + rep);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        uint64_t t; CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
        if (ms < best) best = ms;
        if (t < ticks) ticks = t;
      }
      if (w == 1) printf("  %12.2f", (double)ticks / ninstr);
      printf("  %8.2f", best * 1e-3 * 2.4e9 / (ninstr * w));
    }
    printf("\n");
  }
  return 0;
}
