// codec_chunked.hip -- compress and the decompress -> compress round trip in chunks with batched inverses (gfx950).
//
// A translation unit of its own on purpose: added to d377.hip these two kernels moved the register allocation of every kernel
// there (k_scalar_mul_var 256 VGPRs / 75 SGPR spills -> 246 / 104; all within +-1 % in time, profiles/r05_ab_codegen_shift.txt),
// and the headline kernels' tables are measured artefacts (tests/test_codegen.py).  d377.hip's launch rules call in through
// codec_chunked.hpp.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/decaf377_amd.h"
#include "curve.hpp"
#include "device_util.hpp"
#include "dcb.hpp"
#include "row_ops.hpp"
#include "host_state.hpp"
#include "codec_chunked.hpp"

using namespace d377;

namespace {

// compress and the round trip in chunks, like k_decompress_chunked: the generic compressor's square root takes the inverse of its
// denominator from the chunk's batched inversion too.  In the round trip that denominator exists only once the point is
// decoded, so a chunk has two inversions: the decoding pass leaves (X, Y) of the point -- Z = 1, T = XY -- as raw limbs in
// the element's records 1-3 and the compressor's denominator in record 0 (whose inverse it has just consumed), and after
// the pass the lanes invert those together and compress (dcb_rounds' post step).  Same bytes as k_compress / k_roundtrip.
__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_compress_chunked(SqrtTables T, const uint64_t* xyzt, size_t n,
                                                            uint8_t* enc32, DcbScratch dcb) {
  D377_POW_LDS();
  D377_DCB_BEGIN(enc32);
  dcb_rounds<1, false, false>(n, io, pt,
    [&](size_t i, int j) { dcb_put_den(io, 0, j, ge_compress_den(load_ge_mont256(xyzt, i))); },
    [&](size_t i, int, const uint32_t (*invw)[8], bool) {
      const fe inv = fe_from_words(invw[0]);
      uint32_t w[8];
      ge_compress(T, pt, load_ge_mont256(xyzt, i), w, true, &inv);
      store32(enc32, i, w);
    });
  D377_DCB_END();
}

__global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD) k_roundtrip_chunked(SqrtTables T, const uint8_t* enc32, size_t n,
                                                             uint8_t* out32, uint8_t* status, DcbScratch dcb) {
  D377_POW_LDS();
  D377_DCB_BEGIN(out32);
  uint32_t badmask = 0;                              // bit j: element j of the current chunk did not decode
  dcb_rounds<1, false, false>(n, io, pt,
    [&](size_t i, int j) {
      uint32_t w[8];
      load32(enc32, i, w);
      dcb_put_den(io, 0, j, ge_decompress_den(w));
    },
    [&](size_t i, int j, const uint32_t (*invw)[8], bool) {
      uint32_t w[8];
      load32(enc32, i, w);
      const fe inv = fe_from_words(invw[0]);
      ge g;
      const uint32_t bad = ge_decompress(T, pt, w, &g, &inv);
      status[i] = (uint8_t)bad;
      if (bad) { badmask |= 1u << j; g = ge_identity(); }
      uint32_t pk[24] = {};
#pragma unroll
      for (int k = 0; k < NL; ++k) { pk[k] = g.x.l[k]; pk[NL + k] = g.y.l[k]; }
      io.put(1, j, pk); io.put(2, j, pk + 8); io.put(3, j, pk + 16);
      dcb_put_den(io, 0, j, ge_compress_den(g));
    },
    [&](DcbIO& io2, int cnt) {
      dcb_invert_slot_with(io2, 0, cnt, [](const fe& c) { return row::fe_invert_lanes(c); });
#pragma unroll 1
      for (int j = 0; j < cnt; ++j) {
        uint32_t pk[24], iw[8], w[8];
        io2.get(1, j, pk); io2.get(2, j, pk + 8); io2.get(3, j, pk + 16); io2.get(0, j, iw);
        ge g;
#pragma unroll
        for (int k = 0; k < NL; ++k) { g.x.l[k] = pk[k]; g.y.l[k] = pk[NL + k]; }
        g.z = fe_const(FE_ONE);
        g.t = fe_mul(g.x, g.y);
        const fe inv = fe_from_words(iw);
        const bool bad = ((badmask >> j) & 1u) != 0;
        ge_compress(T, pt, g, w, !bad, &inv);
        if (bad) store32_zero(io2.out32, io2.base + (size_t)j * BLOCK); else io2.emit(j, w);
      }
      badmask = 0;
    });
  D377_DCB_END();
}

}  // namespace

namespace d377 {

bool codec_chunked_ok(DeviceState& d) {
  if (d.codec_chunked < 0) {
    int a = 0, b = 0;
    const bool asked = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void*>(k_compress_chunked), BLOCK, 0) == hipSuccess &&
                       hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void*>(k_roundtrip_chunked), BLOCK, 0) == hipSuccess;
    d.codec_chunked = (asked && a >= 1 && a <= WAVES_PER_SIMD && b >= 1 && b <= WAVES_PER_SIMD) ? 1 : 0;
  }
  return d.codec_chunked == 1;
}

int codec_chunked_launch(DeviceState& d, hipStream_t s, bool roundtrip, const void* in0, size_t n, void* out0, void* out1, int grid,
                         const DcbScratch& dcb) {
  const SqrtTables T = d.tables();
  if (roundtrip)
    hipLaunchKernelGGL(k_roundtrip_chunked, dim3((unsigned)grid), dim3(BLOCK), 0, s, T, (const uint8_t*)in0, n, (uint8_t*)out0, (uint8_t*)out1, dcb);
  else
    hipLaunchKernelGGL(k_compress_chunked, dim3((unsigned)grid), dim3(BLOCK), 0, s, T, (const uint64_t*)in0, n, (uint8_t*)out0, dcb);
  HIP_TRY(hipGetLastError());
  return D377_OK;
}

}  // namespace d377
