// GPU self-test of the four-lane group operations of msm.hip (gq_double_neg, gq_add) against the single-lane formulas
// of curve.hpp, straight-line and inside loops (the Horner chains), built and run by tests/test_msm.py on the GPU box.
// It exists because these operations once passed in isolation and failed inside a loop: hipcc's DPP combiner folded the
// quad_perm moves into v_subrev_u32_dpp instructions that came back computed on the lane's own value (msm.hip,
// fe_quad_perm).  Test infrastructure only: it includes the kernels' translation unit to reach its internal functions.
#include "../../decaf377_amd/csrc/msm.hip"
thread_local char d377_g_err[512] = "";
int d377::debug_device_delay_ms() { return 0; }

__device__ bool same_point(const ge& a, const ge& b) {   // projective equality, T included
  return fe_eq(fe_mul(a.x, b.z), fe_mul(b.x, a.z)) && fe_eq(fe_mul(a.y, b.z), fe_mul(b.y, a.z)) && fe_eq(fe_mul(a.t, b.z), fe_mul(b.t, a.z));
}
__global__ void k_test(int* out) {
  __shared__ uint32_t rec[GQ_WORDS], rid[GQ_WORDS];
  const int role = threadIdx.x & 3;
  const ge g = ge_generator(), p = ge_double(g), q = ge_add(p, g), id = ge_identity();
  if (threadIdx.x == 0) gq_store_cached(rec, q);
  if (threadIdx.x == 1) gq_store_cached(rid, id);
  __syncthreads();
  int ok = 0;
  // one operation each
  if (same_point(ge_double_neg(p, true), gq_to_ge(gq_double_neg(gq_from_ge(p, role), role)))) ok |= 1;
  if (same_point(ge_add(p, q), gq_to_ge(gq_add(gq_from_ge(p, role), rec, role, false)))) ok |= 2;
  if (same_point(ge_sub_pts(p, q), gq_to_ge(gq_add(gq_from_ge(p, role), rec, role, true)))) ok |= 4;
  if (same_point(q, gq_to_ge(gq_add(gq_from_ge(id, role), rec, role, false)))) ok |= 8;
  // repeated doublings, compared after every step and after a loop
  {
    fe u = gq_from_ge(p, role);
    ge ref = p;
    bool all = true;
    for (int j = 0; j < 5; ++j) {
      u = gq_double_neg(u, role);
      ref = ge_double_neg(ref, true);
      all = all && same_point(gq_to_ge(u), ref);
    }
#pragma unroll 1
    for (int j = 0; j < 14; ++j) { u = gq_double_neg(u, role); ref = ge_double_neg(ref, true); }
    if (all && same_point(gq_to_ge(u), ref)) ok |= 16;
    if (same_point(gq_to_ge(gq_add(u, rec, role, true)), ge_sub_pts(ref, q))) ok |= 32;
  }
  // the tail's shape: identity, 61 x (4 doublings, + identity), 4 doublings, + Q  ==  Q
  {
    fe v = gq_from_ge(id, role);
    int steps = 0;
#pragma unroll 1
    for (int w = 0; w < 61; ++w) {
#pragma unroll 1
      for (int j = 0; j < 4; ++j) v = gq_double_neg(v, role);
      v = gq_add(v, rid, role, (w & 1) != 0);
      if (same_point(gq_to_ge(v), id)) ++steps;
    }
#pragma unroll 1
    for (int j = 0; j < 4; ++j) v = gq_double_neg(v, role);
    v = gq_add(v, rec, role, false);
    if (steps == 61 && same_point(gq_to_ge(v), q)) ok |= 64;
  }
  out[threadIdx.x] = ok;
}
int main() {
  int* d = nullptr;
  if (hipMalloc(&d, 64 * sizeof(int)) != hipSuccess) { printf("no device\n"); return 2; }
  hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, d);
  int h[64];
  if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); return 2; }
  int bad = 0;
  for (int i = 0; i < 64; ++i) bad += h[i] != 127;
  printf(bad ? "GQ_SELFTEST_FAIL lane0=%d bad_lanes=%d\n" : "GQ_SELFTEST_OK %d %d\n", h[0], bad);
  return bad ? 1 : 0;
}
