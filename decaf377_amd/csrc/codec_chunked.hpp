// codec_chunked.hpp -- what d377.hip's launch rules see of codec_chunked.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "dcb.hpp"
#include "host_state.hpp"

namespace d377 {
// May the kernels of codec_chunked.hip claim lane sets of the scratch areas (their residency per CU is within the sets)?
// Asked of the runtime once per device; false keeps the caller on its wide-grid kernels.
bool codec_chunked_ok(DeviceState& d);
// compress (roundtrip = false: in0 = Element records, out0 = encodings) or the round trip (in0 = encodings, out0 = encodings,
// out1 = status bytes) in chunks, `grid` workgroups dealt out as `dcb` says, enqueued on `s`.  The caller holds the
// scratch areas' guard.
int codec_chunked_launch(DeviceState& d, hipStream_t s, bool roundtrip, const void* in0, size_t n, void* out0, void* out1, int grid,
                         const DcbScratch& dcb);
}  // namespace d377
