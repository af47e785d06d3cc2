// host_state.hpp -- per-context / per-device host state shared by the translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <mutex>
#include <vector>

#include "../../include/decaf377_amd.h"
#include "curve.hpp"

extern thread_local char d377_g_err[512];

namespace d377 {

inline int fail(int code, const char* fmt, const char* detail) {
  snprintf(d377_g_err, sizeof d377_g_err, fmt, detail);
  return code;
}
#define HIP_TRY(expr)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) return ::d377::fail(D377_ERR_HIP, #expr ": %s", hipGetErrorString(e_)); \
  } while (0)

// workspace of the multi-scalar multiplication (msm.hip), grow-only
struct MsmWorkspace {
  uint8_t* mem = nullptr;
  size_t cap = 0;
};

struct DeviceState {
  int id = -1;
  int cus = 0;
  uint32_t* gtab = nullptr;
  uint8_t* s_lookup = nullptr;
  uint32_t* fbase = nullptr;
  uint32_t* vb_scratch = nullptr;
  int vb_blocks = 0;
  hipStream_t stream = nullptr;          // compute stream of the host-pointer entry points
  hipStream_t copy_stream = nullptr;     // PCIe copies of the pipelined host path
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
  // grow-only staging buffers for the host-pointer entry points (set 1 = second half of the
  // double buffer used when a large batch is pipelined chunk by chunk)
  uint8_t* buf[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap[4] = {0, 0, 0, 0};
  uint8_t* buf2[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap2[4] = {0, 0, 0, 0};
  MsmWorkspace msm;
  SqrtTables tables() const { return SqrtTables{gtab, s_lookup}; }
};

inline int ensure(DeviceState& d, int slot, size_t bytes) {
  if (bytes <= d.cap[slot]) return D377_OK;
  if (d.buf[slot]) HIP_TRY(hipFree(d.buf[slot]));
  d.buf[slot] = nullptr; d.cap[slot] = 0;
  size_t want = bytes + bytes / 4 + 4096;
  HIP_TRY(hipMalloc(&d.buf[slot], want));
  d.cap[slot] = want;
  return D377_OK;
}

inline int ensure2(DeviceState& d, int slot, size_t bytes) {
  if (bytes <= d.cap2[slot]) return D377_OK;
  if (d.buf2[slot]) HIP_TRY(hipFree(d.buf2[slot]));
  d.buf2[slot] = nullptr; d.cap2[slot] = 0;
  HIP_TRY(hipMalloc(&d.buf2[slot], bytes + 4096));
  d.cap2[slot] = bytes + 4096;
  return D377_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace d377

struct d377_ctx {
  std::vector<d377::DeviceState> devs;
  std::mutex mu;
};
