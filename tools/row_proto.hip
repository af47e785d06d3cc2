// row_proto.hip -- test and timing harness of the lane-spread field arithmetic (decaf377_amd/csrc/row_ops.hpp).  Dev tool:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared tools/row_proto.hip -o build/row_proto.so ; python tools/row_proto.py
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../decaf377_amd/csrc/curve.hpp"
#include "../decaf377_amd/csrc/quad_ops.hpp"
#include "../decaf377_amd/csrc/row_ops.hpp"

using namespace d377;

__global__ void k_row_mul(const uint32_t* a, const uint32_t* b, uint32_t* out, int n) {
  const row::RowK K = row::row_consts();
  const int prod = blockIdx.x * 4 + (threadIdx.x >> 4), j = threadIdx.x & 15;
  const int p = prod < n ? prod : n - 1;
  const uint32_t r = row::row_mul(a[p * 16 + j], b[p * 16 + j], K);
  if (prod < n) out[p * 16 + j] = r;
}
// a dependent chain of products on ONE wave: x <- x * y, iters times
__global__ void __launch_bounds__(64) k_row_chain(const uint32_t* a, const uint32_t* b, uint32_t* out, int iters) {
  const row::RowK K = row::row_consts();
  uint32_t x = a[threadIdx.x], y = b[threadIdx.x];
#pragma unroll 1
  for (int i = 0; i < iters; ++i) x = row::row_mul(x, y, K);
  out[threadIdx.x] = x;
}
__global__ void __launch_bounds__(64) k_lane_chain(const uint32_t* a, const uint32_t* b, uint32_t* out, int iters) {
  fe x, y;
  for (int i = 0; i < NL; ++i) { x.l[i] = a[threadIdx.x * NL + i] & MASK29; y.l[i] = b[threadIdx.x * NL + i] & MASK29; }
  x.l[NL - 1] &= 0xFFFFF; y.l[NL - 1] &= 0xFFFFF;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) x = fe_mul(x, y);
  for (int i = 0; i < NL; ++i) out[threadIdx.x * NL + i] = x.l[i];
}
__global__ void __launch_bounds__(64) k_lane_sqr_chain(const uint32_t* a, uint32_t* out, int iters) {
  fe x;
  for (int i = 0; i < NL; ++i) x.l[i] = a[threadIdx.x * NL + i] & MASK29;
  x.l[NL - 1] &= 0xFFFFF;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) x = fe_sqr(x);
  for (int i = 0; i < NL; ++i) out[threadIdx.x * NL + i] = x.l[i];
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return -1; } } while (0)

extern "C" int row_proto_mul(const uint32_t* a, const uint32_t* b, uint32_t* out, int n) {
  uint32_t *da, *db, *dout;
  const size_t bytes = (size_t)n * 16 * 4;
  CK(hipMalloc(&da, bytes)); CK(hipMalloc(&db, bytes)); CK(hipMalloc(&dout, bytes));
  CK(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_row_mul, dim3((n + 3) / 4), dim3(64), 0, 0, da, db, dout, n);
  CK(hipGetLastError());
  CK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
  return 0;
}
// which: 0 = row product chain, 1 = one-lane-per-element fe_mul chain, 2 = fe_sqr chain; -> ms for `iters` dependent products
extern "C" int row_proto_chain(int which, int iters, const uint32_t* a, const uint32_t* b, uint32_t* out, float* ms) {
  uint32_t *da, *db, *dout;
  const size_t bytes = 64 * 16 * 4;
  CK(hipMalloc(&da, bytes)); CK(hipMalloc(&db, bytes)); CK(hipMalloc(&dout, bytes));
  CK(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    if (which == 0) hipLaunchKernelGGL(k_row_chain, dim3(1), dim3(64), 0, 0, da, db, dout, iters);
    else if (which == 1) hipLaunchKernelGGL(k_lane_chain, dim3(1), dim3(64), 0, 0, da, db, dout, iters);
    else hipLaunchKernelGGL(k_lane_sqr_chain, dim3(1), dim3(64), 0, 0, da, dout, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
  }
  CK(hipEventElapsedTime(ms, e0, e1));
  CK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
  return 0;
}

// what the two row swaps do, lane by lane: a = lane, b = 100 + lane
__global__ void k_swap_probe(uint32_t* out) {
  const uint32_t a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1]; out[128 + threadIdx.x] = s[0]; out[192 + threadIdx.x] = s[1];
}
extern "C" int row_proto_swap_probe(uint32_t* out) {
  uint32_t* d;
  CK(hipMalloc(&d, 256 * 4));
  hipLaunchKernelGGL(k_swap_probe, dim3(1), dim3(64), 0, 0, d);
  CK(hipMemcpy(out, d, 256 * 4, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return 0;
}

// group operations on the four rows of a wave: v = (X, Y, Z, T) as 64 words, q = cached record (64 words)
__global__ void __launch_bounds__(64) k_row_point(const uint32_t* v_in, const uint32_t* qrec, uint32_t* out) {
  __shared__ uint32_t rec[64];
  const row::RowK K = row::row_consts();
  const row::RowSel S = row::row_sel();
  rec[threadIdx.x] = qrec[threadIdx.x];
  __syncthreads();
  const uint32_t v = v_in[threadIdx.x];
  out[threadIdx.x] = row::rq_double_neg(v, S, K);
  out[64 + threadIdx.x] = row::rq_add(v, rec, S, false, K);
  out[128 + threadIdx.x] = row::rq_add(v, rec, S, true, K);
  out[192 + threadIdx.x] = row::rq_double_neg(row::rq_double_neg(v, S, K), S, K);
}
__global__ void __launch_bounds__(64) k_row_dbl_chain(const uint32_t* v_in, uint32_t* out, int iters) {
  const row::RowK K = row::row_consts();
  const row::RowSel S = row::row_sel();
  uint32_t v = v_in[threadIdx.x];
#pragma unroll 1
  for (int i = 0; i < iters; ++i) v = row::rq_double_neg(v, S, K);
  out[threadIdx.x] = v;
}
__global__ void __launch_bounds__(64) k_quad_dbl_chain(const uint32_t* a, uint32_t* out, int iters) {
  fe x;
  for (int i = 0; i < NL; ++i) x.l[i] = a[threadIdx.x * NL + i] & MASK29;
  x.l[NL - 1] &= 0xFFFFF;
  const int role = threadIdx.x & 3;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) x = gq_double_neg(x, role);
  for (int i = 0; i < NL; ++i) out[threadIdx.x * NL + i] = x.l[i];
}
extern "C" int row_proto_point(const uint32_t* v, const uint32_t* q, uint32_t* out) {
  uint32_t *dv, *dq, *dout;
  CK(hipMalloc(&dv, 256)); CK(hipMalloc(&dq, 256)); CK(hipMalloc(&dout, 1024));
  CK(hipMemcpy(dv, v, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dq, q, 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_row_point, dim3(1), dim3(64), 0, 0, dv, dq, dout);
  CK(hipGetLastError());
  CK(hipMemcpy(out, dout, 1024, hipMemcpyDeviceToHost));
  (void)hipFree(dv); (void)hipFree(dq); (void)hipFree(dout);
  return 0;
}
// which: 0 = doubling chain on rows, 1 = doubling chain on quads
extern "C" int row_proto_dbl_chain(int which, int iters, const uint32_t* a, uint32_t* out, float* ms) {
  uint32_t *da, *dout;
  const size_t bytes = 64 * 16 * 4;
  CK(hipMalloc(&da, bytes)); CK(hipMalloc(&dout, bytes));
  CK(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    if (which == 0) hipLaunchKernelGGL(k_row_dbl_chain, dim3(1), dim3(64), 0, 0, da, dout, iters);
    else hipLaunchKernelGGL(k_quad_dbl_chain, dim3(1), dim3(64), 0, 0, da, dout, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
  }
  CK(hipEventElapsedTime(ms, e0, e1));
  CK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(dout);
  return 0;
}

// the wave's inversion (fe_invert_wave) beside the lane's (fe_invert): one wave per element, 64 words out per element --
// canonical limbs of 1/x by the wave, by the lane, of x * (1/x), and of one
__global__ void __launch_bounds__(64) k_row_invert(const uint32_t* a, uint32_t* out, int n) {
  const int e = blockIdx.x;
  if (e >= n) return;
  fe x;
  for (int i = 0; i < NL; ++i) x.l[i] = a[e * NL + i] & MASK29;
  x.l[NL - 1] &= 0xFFFFF;
  const fe iw = row::fe_invert_wave(x), il = fe_invert(x);
  const fe cw = fe_canon(iw), cl = fe_canon(il), pr = fe_canon(fe_mul(x, iw)), one = fe_canon(fe_const(FE_ONE));
  if (threadIdx.x == 0)
    for (int i = 0; i < NL; ++i) {
      out[e * 64 + i] = cw.l[i]; out[e * 64 + 9 + i] = cl.l[i]; out[e * 64 + 18 + i] = pr.l[i]; out[e * 64 + 27 + i] = one.l[i];
    }
}
__global__ void __launch_bounds__(64) k_invert_chain(const uint32_t* a, uint32_t* out, int iters, int which) {
  fe x;
  for (int i = 0; i < NL; ++i) x.l[i] = a[i] & MASK29;
  x.l[NL - 1] &= 0xFFFFF;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) x = fe_add(which == 0 ? row::fe_invert_wave(x) : fe_invert(x), fe_const(FE_ONE));
  for (int i = 0; i < NL; ++i) out[threadIdx.x * NL + i] = x.l[i];
}
extern "C" int row_proto_invert(const uint32_t* a, uint32_t* out, int n) {
  uint32_t *da, *dout;
  CK(hipMalloc(&da, (size_t)n * NL * 4)); CK(hipMalloc(&dout, (size_t)n * 64 * 4));
  CK(hipMemcpy(da, a, (size_t)n * NL * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dout, 0, (size_t)n * 64 * 4));
  hipLaunchKernelGGL(k_row_invert, dim3(n), dim3(64), 0, 0, da, dout, n);
  CK(hipGetLastError());
  CK(hipMemcpy(out, dout, (size_t)n * 64 * 4, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(dout);
  return 0;
}
// which: 0 = the wave's inversion, 1 = the lane's; -> ms for `iters` dependent inversions
extern "C" int row_proto_invert_chain(int which, int iters, const uint32_t* a, uint32_t* out, float* ms) {
  uint32_t *da, *dout;
  CK(hipMalloc(&da, 64 * 4)); CK(hipMalloc(&dout, 64 * NL * 4));
  CK(hipMemcpy(da, a, NL * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_invert_chain, dim3(1), dim3(64), 0, 0, da, dout, iters, which);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
  }
  CK(hipEventElapsedTime(ms, e0, e1));
  CK(hipMemcpy(out, dout, 64 * NL * 4, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(dout);
  return 0;
}

// the rounds of one wave inversion, for the Python model: 25 x 80 words (the 64 lanes' limbs, then eta, u, v, q, r, md, me, f0, g0)
__global__ void __launch_bounds__(64) k_row_invert_trace(const uint32_t* a, uint32_t* dbg) {
  fe x;
  for (int i = 0; i < NL; ++i) x.l[i] = a[i] & MASK29;
  x.l[NL - 1] &= 0xFFFFF;
  const fe iw = row::fe_invert_wave_impl<true>(x, dbg);
  if (threadIdx.x == 0) { const fe c = fe_canon(iw); for (int i = 0; i < NL; ++i) dbg[25 * 80 + i] = c.l[i]; }
}
extern "C" int row_proto_invert_trace(const uint32_t* a, uint32_t* out) {
  uint32_t *da, *dout;
  CK(hipMalloc(&da, NL * 4)); CK(hipMalloc(&dout, (25 * 80 + 16) * 4));
  CK(hipMemcpy(da, a, NL * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dout, 0, (25 * 80 + 16) * 4));
  hipLaunchKernelGGL(k_row_invert_trace, dim3(1), dim3(64), 0, 0, da, dout);
  CK(hipGetLastError());
  CK(hipMemcpy(out, dout, (25 * 80 + 16) * 4, hipMemcpyDeviceToHost));
  (void)hipFree(da); (void)hipFree(dout);
  return 0;
}
