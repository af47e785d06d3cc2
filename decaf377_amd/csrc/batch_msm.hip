// batch_msm.hip -- many SMALL multiscalar sums at once (gfx950): out[i] = sum_{j < m} scalar[i m + j] * point[i m + j], m = 1 .. 8.
//
// The reference's Element::vartime_multiscalar_mul (src/ark_curve/element/projective.rs:99-117) is a fold of `acc + scalar *
// point`, and the shape its own test exercises is a 3-term sum per case (tests/operations.rs:44-60).  d377_msm (msm.hip) is ONE
// sum per call -- Pippenger, with a floor of a quarter of a millisecond -- so a caller with 2^16 independent 3-term sums had to
// compose three scalar-multiplication batches and two addition batches.  Here every sum is one Straus chain: the m points of a
// sum share ONE chain of 252 doublings, each window adds one entry of each point's table of cached 0 .. 8 P (signed 4-bit
// digits of k / 2 mod r, the encoding of the double needs no square root: curve.hpp).  63 x 4 doublings + 64 m additions + the
// m tables, against m x (63 x 4 doublings + 63 additions + a table): 0.51 of the composition's instructions at m = 3.
//
//   k_batch_msm_lane   one lane per sum, in chunks like k_scalar_mul_var (dcb.hpp): the m tables of a lane and its digit words
//                      live in a per-device scratch area that exists once per resident lane (m x 1 728 + 256 bytes per lane,
//                      grown on first use), the sums of a chunk share one inversion per wave
//   k_batch_msm_wave   one WAVE per sum in the lane-spread form (row_ops.hpp), tables in LDS: batches up to one sum per SIMD,
//                      where a call is as long as one chain
//
// A translation unit of its own, like codec_chunked.hip: the register tables of d377.hip's kernels are measured artefacts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/decaf377_amd.h"
#include "curve.hpp"
#include "device_util.hpp"
#include "dcb.hpp"
#include "quad_ops.hpp"
#include "row_ops.hpp"
#include "straus.hpp"
#include "host_state.hpp"

using namespace d377;

namespace {

constexpr int BM_MAX = D377_BATCH_MSM_MAX_TERMS;
static_assert(BM_MAX == 8, "a window's digits of the m points of a sum are the eight nibbles of one word");
static_assert(VB_ENTRIES == 9, "straus_sum stores entries 0 .. 8 of every point's table");

// Scratch of one resident lane: tables [point][entry][lane] as k_scalar_mul_var's (a wave stores one entry as 12 KiB
// contiguous; a negative digit swaps the ypx / ymx slots by address), and the digit words [window][lane]: nibble p of word w =
// the signed digit of point p in window w.
struct StrausTab {
  uint32_t* tab;
  uint32_t* dig;
  size_t nthreads, tid;
  __device__ __forceinline__ uint32_t* entry(int p, int j) const { return tab + (((size_t)p * VB_ENTRIES + j) * nthreads + tid) * VB_ENTRY_WORDS; }
  __device__ __forceinline__ void store(int p, int j, const gec& c) {
    uint32_t* q = entry(p, j);
    slot_store(q, c.ypx); slot_store(q + SLOT, c.ymx); slot_store(q + 2 * SLOT, c.z2); slot_store(q + 3 * SLOT, c.kt);
  }
  __device__ __forceinline__ gec load(int p, int j, bool swap) const {
    const uint32_t* q = entry(p, j);
    gec c;
    c.ypx = slot_load(q + (swap ? SLOT : 0));
    c.ymx = slot_load(q + (swap ? 0 : SLOT));
    c.z2 = slot_load(q + 2 * SLOT);
    c.kt = slot_load(q + 3 * SLOT);
    return c;
  }
  __device__ __forceinline__ void dig_store(int w, uint32_t v) { dig[(size_t)w * nthreads + tid] = v; }
  __device__ __forceinline__ uint32_t dig_load(int w) const { return dig[(size_t)w * nthreads + tid]; }
};
template <bool ENCODED>
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)
k_batch_msm_lane(SqrtTables T, const void* pts_in, const uint8_t* scalar32, int m, size_t n, uint8_t* out32, uint64_t* xyzt_out,
                 uint8_t* status, uint32_t* tab, uint32_t* dig, DcbScratch dcb) {
  __shared__ uint32_t lds_pow_[ENCODED ? POW_TAB * NL * BLOCK : 1];
  LdsPowTab pt;
  pt.col = lds_pow_ + (ENCODED ? threadIdx.x : 0);
  D377_DCB_BEGIN(out32);
  StrausTab st{tab, dig, (size_t)dcb.nslots * BLOCK, io.lane};
  dcb_rounds<0, true>(n, io, pt,
    [&](size_t, int) {},
    [&](size_t i, int j, const uint32_t (*)[8], bool) {
      const size_t first = i * (size_t)m;
      const ge r = straus_sum(st, m, [&](int p, uint32_t k[8]) { load32(scalar32, first + (size_t)p, k); }, [&](int p, ge* g) -> bool {
        if (ENCODED) {
          uint32_t w[8];
          load32(reinterpret_cast<const uint8_t*>(pts_in), first + (size_t)p, w);
          const uint32_t bad = ge_decompress(T, pt, w, g);
          status[first + (size_t)p] = (uint8_t)bad;
          return bad != 0;
        }
        *g = load_ge_mont256(reinterpret_cast<const uint64_t*>(pts_in), first + (size_t)p);
        D377_INVARIANT(T, *g, !fe_is_zero(g->z));
        return fe_is_zero(g->z);                                   // a record with Z = 0 is no group element: the identity
      }, DCB_WANT_T);
      if (xyzt_out) store_ge_mont256(xyzt_out, i, ge_double_fast(r, true));   // the chain ran on k / 2: the sum is the double
      dcb_put(io, j, ge_dcb_from_half(r, false));
    });
  D377_DCB_END();
}

// ---- one wave per sum: the chain in the lane-spread form (row_ops.hpp), as msm.hip's k_msm_tiny, m tables in LDS --------------
struct OneIO {                                                   // the square-root-free compressor's records for a single element
  uint32_t st[4][8], parked_[8], out[8];
  __device__ __forceinline__ void put(int s, int, const uint32_t* w) { for (int k = 0; k < 8; ++k) st[s][k] = w[k]; }
  __device__ __forceinline__ void get(int s, int, uint32_t* w) const { for (int k = 0; k < 8; ++k) w[k] = st[s][k]; }
  __device__ __forceinline__ void park(int, const uint32_t* w) { for (int k = 0; k < 8; ++k) parked_[k] = w[k]; }
  __device__ __forceinline__ void parked(int, uint32_t* w) const { for (int k = 0; k < 8; ++k) w[k] = parked_[k]; }
  __device__ __forceinline__ void emit(int, const uint32_t* w) { for (int k = 0; k < 8; ++k) out[k] = w[k]; }
};
using row::RQ_WORDS;
// the square roots of a group of Encodings keep their POW_TAB odd powers (64 words each) in the table of the group's first
// point, which is built after them
static_assert(POW_TAB * 64 <= row::RQ_TAB_ENTRIES * RQ_WORDS, "row_sqrt_powers' scratch must fit in one point's LDS table");
template <bool ENCODED>
__global__ void __launch_bounds__(64)
k_batch_msm_wave(SqrtTables T, const void* pts_in, const uint8_t* scalar32, int m, size_t n, uint8_t* out32, uint64_t* xyzt_out,
                 uint8_t* status) {
  extern __shared__ uint32_t tab[];                                // m tables of RQ_TAB_ENTRIES x RQ_WORDS words (dynamic: 2 304 bytes per term)
  __shared__ uint32_t xrec[2 * RQ_WORDS];
  __shared__ uint32_t sdg[BM_MAX][8];                              // the points' signed digits (wave-uniform reads in the loop)
  const int t = threadIdx.x;
  const row::RowK K = row::row_consts();
  const row::RowSel S = row::row_sel();
  const size_t first = (size_t)blockIdx.x * (size_t)m;            // grid = n
  (void)n;
  // points in groups of four: lane t looks after point base + (t & 3) of the group (the square roots of Encodings run their
  // power chains on the four rows of the wave, one point per row)
#pragma unroll 1
  for (int base = 0; base < m; base += 4) {
    const int pj = t & 3;
    const bool mine = base + pj < m;
    const size_t e_mine = first + (size_t)(mine ? base + pj : 0);
    ge g;
    bool skip = !mine;
    if (ENCODED) {
      uint32_t w[8];
      load32(reinterpret_cast<const uint8_t*>(pts_in), e_mine, w);
      if (t < 4) row::row_store_from_fe(xrec + 16 * t, ge_decompress_den(w));
      __syncthreads();
      const row::RowPowers pw = row::row_sqrt_powers(xrec[t], tab + base * row::RQ_TAB_ENTRIES * RQ_WORDS, t, K);   // (this group's tables: not built yet)
      __syncthreads();
      xrec[t] = pw.v; xrec[RQ_WORDS + t] = pw.uv;
      __syncthreads();
      const fe pv = row::row_load_to_fe(xrec + 16 * pj), puv = row::row_load_to_fe(xrec + RQ_WORDS + 16 * pj);
      __syncthreads();
      const uint32_t bad = ge_decompress_from_powers(T, w, pv, puv, &g);
      if (t < 4 && mine) status[e_mine] = (uint8_t)bad;
      skip |= bad != 0;
    } else {
      g = load_ge_mont256(reinterpret_cast<const uint64_t*>(pts_in), e_mine);
      skip |= fe_is_zero(g.z);
      D377_INVARIANT(T, g, t < 4 && !skip);
    }
#pragma unroll 1
    for (int j = 0; j < 4 && base + j < m; ++j) {
      uint32_t k[8], dg[8];
      load32(scalar32, first + (size_t)(base + j), k);
      fr_reduce_words(k);
      fr_half_words(k);
      fr_recode_signed16(k, dg);
      if (pj == j && t < 16) row::row_store_from_fe(xrec + 16 * (t >> 2), fe_pick(t >> 2, g.x, g.y, g.z, g.t));
      __syncthreads();
      const bool dead = __shfl((int)skip, j) != 0;                  // (wave-uniform: lane j's verdict on point base + j)
      if (t < 8) sdg[base + j][t] = dead ? 0u : dg[t];              // dead: every digit 0
      row::rq_build_table(dead ? row::rq_identity(S) : xrec[t], tab + (base + j) * row::RQ_TAB_ENTRIES * RQ_WORDS, S, K);
      __syncthreads();
    }
  }
  uint32_t v = row::rq_identity(S);
#pragma unroll 1
  for (int i = 63; i >= 0; --i) {
    if (i != 63) {
#pragma unroll 1
      for (int k = 0; k < 4; ++k) v = row::rq_double_neg(v, S, K);  // four sign-folded doublings keep the sign
    }
#pragma unroll 1
    for (int j = 0; j < m; ++j) {
      const int d = fr_digit(sdg[j], i);
      if (d != 0) v = row::rq_add(v, tab + (j * row::RQ_TAB_ENTRIES + (d < 0 ? -d : d)) * RQ_WORDS, S, d < 0, K);
    }
  }
  __syncthreads();
  xrec[t] = v;
  __syncthreads();
  const ge r = row::rq_load_point(xrec);
  if (xyzt_out && t == 0) store_ge_mont256(xyzt_out, blockIdx.x, ge_double_fast(r, true));
  OneIO io;
  dcb_put(io, 0, ge_dcb_from_half(r, false));
  dcb_finish_with(io, 1, [](const fe& c) { return row::fe_invert_wave(c); });   // (every lane holds the same element)
  if (t == 0) store32(out32, blockIdx.x, io.out);
}

// ------------------------------------------------------------------------------ host side ---
// bytes of scratch per resident lane for sums of m terms
size_t scratch_bytes(const DeviceState& d, int m) {
  return d.resident_lanes() * ((size_t)m * VB_ENTRIES * VB_ENTRY_WORDS + BM_WINDOWS) * sizeof(uint32_t);
}

// everything on device pointers, enqueued on `s`; the caller holds ctx->mu
int batch_msm_launch(DeviceState& d, hipStream_t s, bool encoded, const void* pts_in, const uint8_t* scalars, size_t m, size_t n,
                     uint8_t* out32, uint64_t* xyzt_out, uint8_t* status) {
  if (n == 0) return D377_OK;
  const SqrtTables T = d.tables();
  // up to four sums per SIMD: a wave per sum (the lane kernel needs two sums per lane of the chip before it is the better use of
  // it: 2^12 three-term sums 1.41 ms on lanes).  D377_TUNE_TINY_MAX, the developer override of every wave-per-element route,
  // scales this one too.
  const size_t wave_max = 4 * (size_t)d.tuned(D377_TUNE_TINY_MAX, (long long)d.cus * 4);
  if (n <= wave_max) {
    const size_t lds = m * row::RQ_TAB_ENTRIES * RQ_WORDS * sizeof(uint32_t);
    if (encoded) hipLaunchKernelGGL(k_batch_msm_wave<true>, dim3((unsigned)n), dim3(64), lds, s, T, pts_in, scalars, (int)m, n, out32, xyzt_out, status);
    else hipLaunchKernelGGL(k_batch_msm_wave<false>, dim3((unsigned)n), dim3(64), lds, s, T, pts_in, scalars, (int)m, n, out32, xyzt_out, status);
    HIP_TRY(hipGetLastError());
    return D377_OK;
  }
  // residency of the lane kernel against the lane sets (as d377_ctx_create checks the kernels of d377.hip), once per device
  const void* fn = encoded ? reinterpret_cast<const void*>(k_batch_msm_lane<true>) : reinterpret_cast<const void*>(k_batch_msm_lane<false>);
  int& lds = d.bm_lds[encoded ? 1 : 0];
  if (lds < 0) {
    int nb = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, BLOCK, 0));
    int pad = 0;
    if (nb > WAVES_PER_SIMD) {
      pad = (160 * 1024) / (WAVES_PER_SIMD + 1) + 1024;
      if (pad > 64 * 1024) HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, pad));
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, BLOCK, (size_t)pad));
    }
    if (nb < 1 || nb > WAVES_PER_SIMD)
      return fail(D377_ERR_INIT, "residency of %s does not match the lane sets of the scratch areas", "k_batch_msm_lane");
    lds = pad;
  }
  GuardScope vb{d.vb_guard, s};                              // the lane-set areas and this scratch: queue behind their last user
  int rc;
  const size_t need = scratch_bytes(d, (int)m);
  if (need > d.bm_cap) {
    if (ScratchGuard::capturing(s))
      return fail(D377_ERR_ARG, "%s", "batch_msm_small: the table scratch must grow, which cannot happen inside a stream capture -- run one call with this many terms first");
    if ((rc = d.vb_guard.drain())) return rc;               // a launch on another stream may still be using the old area
    if (d.bm_scratch) HIP_TRY(hipFree(d.bm_scratch));
    d.bm_scratch = nullptr; d.bm_cap = 0;
    if (hipMalloc(&d.bm_scratch, need) != hipSuccess) {
      (void)hipGetLastError();
      return fail(D377_ERR_HIP, "%s", "batch_msm_small: hipMalloc of the table scratch failed (0.23 GB per term on 256 CUs)");
    }
    d.bm_cap = need;
  }
  if ((rc = vb.acquire())) return rc;
  const size_t places = (size_t)d.cus * WAVES_PER_SIMD, rounds = (n + BLOCK - 1) / BLOCK;
  const ChunkDeal c = deal_chunks(rounds, places, (size_t)DCB_K, (size_t)d.cus * 64);
  DcbScratch dcb{d.dcb_scratch, d.slot_pool, d.cus * WAVES_PER_SIMD, (int)c.per_lane, d.dcb_sets * BLOCK, (int)c.extra, d.pool_health};
  dcb.prio = c.nchunks <= 2 * places ? 1 : 0;               // as d377.hip's chunks_of
  uint32_t* tab = d.bm_scratch;
  uint32_t* dig = tab + d.resident_lanes() * (size_t)m * VB_ENTRIES * VB_ENTRY_WORDS;
  if (encoded)
    hipLaunchKernelGGL(k_batch_msm_lane<true>, dim3((unsigned)c.nchunks), dim3(BLOCK), lds, s, T, pts_in, scalars, (int)m, n, out32, xyzt_out, status, tab, dig, dcb);
  else
    hipLaunchKernelGGL(k_batch_msm_lane<false>, dim3((unsigned)c.nchunks), dim3(BLOCK), lds, s, T, pts_in, scalars, (int)m, n, out32, xyzt_out, status, tab, dig, dcb);
  HIP_TRY(hipGetLastError());
  return vb.finish();
}

int check_terms(size_t m) {
  if (m < 1 || m > (size_t)BM_MAX) return fail(D377_ERR_ARG, "%s", "batch_msm_small: 1 .. 8 terms per sum (D377_BATCH_MSM_MAX_TERMS); d377_msm for one long sum");
  return D377_OK;
}

// one device's slice of a host batch: copies in, kernel, copies out, synchronised
int batch_msm_one(DeviceState& d, bool encoded, const uint8_t* pts_in, const uint8_t* scalars, size_t m, size_t n, uint8_t* out32,
                  uint64_t* xyzt_out, uint8_t* status) {
  if (n == 0) return D377_OK;
  HIP_TRY(hipSetDevice(d.id));
  int rc = D377_OK;
  SyncOnError guard{&rc, d.id, d.stream, nullptr};
  auto body = [&]() -> int {
    const size_t rec = encoded ? 32 : 128, terms = n * m;
    int r;
    if ((r = ensure(d, 0, terms * rec))) return r;
    if ((r = ensure(d, 1, terms * 32))) return r;
    if ((r = ensure(d, 2, n * (xyzt_out ? 32 + 128 : 32)))) return r;      // the Encodings, then the Element records
    if (encoded && (r = ensure(d, 3, terms))) return r;
    StarveCheck starve{d, d.stream};
    if ((r = starve.before())) return r;
    HIP_TRY(hipMemcpyAsync(d.buf[0], pts_in, terms * rec, hipMemcpyHostToDevice, d.stream));
    HIP_TRY(hipMemcpyAsync(d.buf[1], scalars, terms * 32, hipMemcpyHostToDevice, d.stream));
    uint64_t* xyzt_dev = xyzt_out ? reinterpret_cast<uint64_t*>(d.buf[2] + n * 32) : nullptr;
    if ((r = batch_msm_launch(d, d.stream, encoded, d.buf[0], d.buf[1], m, n, d.buf[2], xyzt_dev, d.buf[3]))) return r;
    HIP_TRY(hipMemcpyAsync(out32, d.buf[2], n * 32, hipMemcpyDeviceToHost, d.stream));
    if (xyzt_out) HIP_TRY(hipMemcpyAsync(xyzt_out, xyzt_dev, n * 128, hipMemcpyDeviceToHost, d.stream));
    if (encoded) HIP_TRY(hipMemcpyAsync(status, d.buf[3], terms, hipMemcpyDeviceToHost, d.stream));
    if ((r = starve.after())) return r;
    HIP_TRY(hipStreamSynchronize(d.stream));
    return starve.verdict();
  };
  rc = body();
  return rc;
}

// host pointers: contiguous slices of the SUMS over the context's devices, one host thread per device (as d377.hip's run_host)
int batch_msm_host(d377_ctx* ctx, bool encoded, const void* pts_in, const uint8_t* scalars, size_t m, size_t n, uint8_t* out32,
                   uint64_t* xyzt_out, uint8_t* status) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  int rc = check_terms(m);
  if (rc) return rc;
  if (n && (!pts_in || !scalars || !out32 || (encoded && !status))) return fail(D377_ERR_ARG, "%s", "null buffer");
  if (n == 0) return D377_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  const size_t nd = ctx->devs.size(), rec = encoded ? 32 : 128;
  if (nd == 1) return batch_msm_one(ctx->devs[0], encoded, (const uint8_t*)pts_in, scalars, m, n, out32, xyzt_out, status);
  const size_t per = (n + nd - 1) / nd;
  std::vector<int> rcs(nd, D377_OK);
  std::vector<std::string> errs(nd);
  std::vector<std::thread> workers;
  const int delay = debug_device_delay_ms();
  for (size_t k = 0; k < nd; ++k) {
    const size_t lo = per * k;
    if (lo >= n) break;
    const size_t cnt = (lo + per <= n) ? per : n - lo;
    workers.emplace_back([&, k, lo, cnt]() {
      if (delay > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay));
      rcs[k] = batch_msm_one(ctx->devs[k], encoded, (const uint8_t*)pts_in + lo * m * rec, scalars + lo * m * 32, m, cnt, out32 + lo * 32,
                             xyzt_out ? xyzt_out + lo * 16 : nullptr, encoded ? status + lo * m : nullptr);
      if (rcs[k] != D377_OK) errs[k] = d377_g_err;
    });
  }
  for (auto& w : workers) w.join();
  for (size_t k = 0; k < nd; ++k)
    if (rcs[k] != D377_OK) return fail(rcs[k], "%s", errs[k].c_str());
  return D377_OK;
}

int batch_msm_dev(d377_ctx* ctx, int dev, void* stream, bool encoded, const void* pts_in, const uint8_t* scalars, size_t m, size_t n,
                  uint8_t* out32, uint64_t* xyzt_out, uint8_t* status) {
  if (!ctx) return fail(D377_ERR_ARG, "%s", "null context");
  if (dev < 0 || (size_t)dev >= ctx->devs.size()) return fail(D377_ERR_ARG, "%s", "device index out of range");
  int rc = check_terms(m);
  if (rc) return rc;
  if (n && (!pts_in || !scalars || !out32 || (encoded && !status))) return fail(D377_ERR_ARG, "%s", "null buffer");
  if (!aligned16(pts_in) || !aligned16(scalars) || !aligned16(out32) || !aligned16(xyzt_out))
    return fail(D377_ERR_ARG, "%s", "device record buffers must be 16-byte aligned");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceState& d = ctx->devs[(size_t)dev];
  HIP_TRY(hipSetDevice(d.id));
  return batch_msm_launch(d, (hipStream_t)stream, encoded, pts_in, scalars, m, n, out32, xyzt_out, status);
}

}  // namespace

extern "C" {

int d377_batch_msm_small(d377_ctx* ctx, const uint64_t* xyzt, const uint8_t* scalar32, size_t m, size_t n, uint8_t* enc32_out,
                         uint64_t* xyzt_out) {
  return batch_msm_host(ctx, false, xyzt, scalar32, m, n, enc32_out, xyzt_out, nullptr);
}
int d377_batch_msm_small_encoded(d377_ctx* ctx, const uint8_t* enc32, const uint8_t* scalar32, size_t m, size_t n, uint8_t* enc32_out,
                                 uint64_t* xyzt_out, uint8_t* status) {
  return batch_msm_host(ctx, true, enc32, scalar32, m, n, enc32_out, xyzt_out, status);
}
int d377_batch_msm_small_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* xyzt, const uint8_t* scalar32, size_t m, size_t n,
                             uint8_t* enc32_out, uint64_t* xyzt_out) {
  return batch_msm_dev(ctx, dev, stream, false, xyzt, scalar32, m, n, enc32_out, xyzt_out, nullptr);
}
int d377_batch_msm_small_encoded_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* enc32, const uint8_t* scalar32, size_t m,
                                     size_t n, uint8_t* enc32_out, uint64_t* xyzt_out, uint8_t* status) {
  return batch_msm_dev(ctx, dev, stream, true, enc32, scalar32, m, n, enc32_out, xyzt_out, status);
}

}  // extern "C"
