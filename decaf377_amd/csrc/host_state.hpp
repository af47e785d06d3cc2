// host_state.hpp -- per-context / per-device host state shared by the translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include <vector>

#include "../../include/decaf377_amd.h"
#include "curve.hpp"
#include "device_util.hpp"

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

// A per-device scratch area that kernels launched on ANY stream may use (the variable-base window
// tables, the MSM workspace).  Users are serialised on the device: a launch first makes its stream wait
// for the previous user's completion event and records its own afterwards, so two `_dev` calls on
// different streams (or a host-path call on the context's private stream while a `_dev` launch is in
// flight) queue up instead of racing.  Host-side access to this struct is under d377_ctx::mu.
struct ScratchGuard {
  hipEvent_t ev = nullptr;
  bool used = false;
  int init() { HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); return D377_OK; }
  void destroy() { if (ev) (void)hipEventDestroy(ev); ev = nullptr; used = false; }
  // A stream that is being captured into a hipGraph takes no part in the hand-over (an event recorded outside
  // the capture cannot be waited on inside it): within the graph the launches keep their stream order.  What that
  // means for a replay that overlaps other calls is area-specific: the lane-set areas (window tables, inversion
  // records) are claimed atomically by every workgroup and never reset, so overlapping kernels simply share them;
  // the MSM workspace is exclusive, and a caller who replays a graph containing an MSM while another MSM of the same
  // device is in flight on a different stream must order the two (include/decaf377_amd.h, "Threads and streams").
  static bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
  }
  bool seen_capture = false;             // some launch on this area has been captured into a graph (see MsmWorkspace)
  int acquire(hipStream_t s) {
    if (capturing(s)) { seen_capture = true; return D377_OK; }
    if (used) HIP_TRY(hipStreamWaitEvent(s, ev, 0));
    return D377_OK;
  }
  int release(hipStream_t s) { if (capturing(s)) return D377_OK; HIP_TRY(hipEventRecord(ev, s)); used = true; return D377_OK; }
  int drain() { if (used) HIP_TRY(hipEventSynchronize(ev)); return D377_OK; }   // before freeing the area
};

// Holds a scratch area for one launch: queues the launch behind the area's last user and records the hand-over event
// on EVERY path out of the launch once the area has been acquired -- also when a later step of the launch fails, so
// that the next user on another stream still queues behind whatever was enqueued.
struct GuardScope {
  ScratchGuard& g;
  hipStream_t s;
  bool held = false;
  int acquire() { int rc = g.acquire(s); held = rc == D377_OK; return rc; }
  // the success path hands over explicitly and reports a failed event record (the next user would not queue behind this
  // launch); the destructor covers the error paths
  int finish() { if (!held) return D377_OK; held = false; return g.release(s); }
  ~GuardScope() { if (held) (void)g.release(s); }
};

// workspace of the multi-scalar multiplication (msm.hip), grow-only.  Once an MSM has been captured into a hipGraph the
// graph holds pointers into the workspace it saw: from then on a workspace that is outgrown is retired (kept until the
// context is destroyed) instead of freed, so a replay never touches freed memory.
struct MsmWorkspace {
  uint8_t* mem = nullptr;
  size_t cap = 0;
  ScratchGuard guard;
  std::vector<uint8_t*> retired;
};

// The kernels that work in chunks and claim a lane set of the per-device scratch areas (d377.hip: dcb_claim).  At most
// chunk_sets[k] workgroups of kernel k may be resident per CU -- that is how many lane sets it may claim; d377_ctx_create checks
// each with hipOccupancyMaxActiveBlocksPerMultiprocessor and pads the launch's LDS allocation for a kernel whose
// registers and own LDS would let more in (chunk_lds, bytes of dynamic LDS per launch).
// Developer tuning (d377_ctx_set_tuning): one value per D377_TUNE_* key, D377_TUNE_DEFAULT = the built-in rule.  Lives in
// the context, written and read under d377_ctx::mu (every launch path holds it); nothing on a call path reads the
// process environment.
struct Tuning {
  int64_t v[D377_TUNE_COUNT];
  Tuning() { for (int k = 0; k < D377_TUNE_COUNT; ++k) v[k] = D377_TUNE_DEFAULT; }
};

enum ChunkKernel { CK_SQRT, CK_ENCODE, CK_HASH, CK_MUL_VAR, CK_MUL_BASE, CK_MUL_VAR_EL, CK_MAP_EL, CK_ENCODE_WIDE, CK_DECOMPRESS, CK_COUNT };

struct DeviceState {
  int id = -1;
  int cus = 0;
  int chunk_lds[CK_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};
  int chunk_blocks[CK_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};     // resident workgroups per CU with that padding (occupancy query)
  int chunk_sets[CK_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};       // lane sets per CU the kernel may claim = the residency it is launched for
  int chunk_k[CK_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0};          // elements per lane per shared inversion
  int fb_narrow_lds = 0;                 // LDS padding of the fixed-base kernel's narrow launch (WAVES_PER_SIMD workgroups per CU)
  int msm_enc_chunked = -1;              // msm.hip: may the chunked decoding pass run (its residency matches the lane sets)?  -1 = not asked yet
  int codec_chunked = -1;                // codec_chunked.hip: may its kernels claim lane sets (their residency matches them)?  -1 = not asked yet
  int msm_span_blocks = -1;              // msm.hip: workgroups of k_msm_spans a CU holds (occupancy query), -1 = not asked yet
  int dcb_sets = 0;                      // lane sets of the round-record area and its pool (the largest chunk_sets x CUs)
  uint32_t* bm_scratch = nullptr;        // batch_msm.hip: tables and digit words of the batched small sums, per resident lane; grown on first use
  size_t bm_cap = 0;
  int bm_lds[2] = {-1, -1};              // batch_msm.hip: LDS padding of its lane kernel (Element / Encoding form), -1 = not asked yet
  uint32_t* gtab = nullptr;
  uint8_t* s_lookup = nullptr;
  uint32_t* fbase = nullptr;             // the fixed-base comb: null until built (d377.hip ensure_comb: at context creation, or by the first fixed-base call of a lazy context)
  uint32_t* fb_bases = nullptr;          // 2^(fb_bits i) B, the comb builder's window bases
  int fb_bits = FB_BITS;                 // the comb's width: the build's default, or d377_ctx_opts::comb_bits (18 / 21 / 23)
  bool fb_lazy = false;                  // d377_ctx_opts::comb_lazy
  uint32_t* vb_scratch = nullptr;
  uint8_t* dcb_scratch = nullptr;        // round records of the batched inversions (curve.hpp: dcb_invert_slot, dcb_finish)
  int* slot_pool = nullptr;              // which of the lane sets of the scratch areas are claimed, and by which ticket (dcb.hpp, DcbScratch)
  uint32_t* pool_health = nullptr;       // ticket counter, workgroups that waited long for a set, workgroups that gave up (dcb.hpp)
  int* pool_host = nullptr;              // pinned: where d377_ctx_health / reset_scratch read the pool and the health words
  uint32_t* starve_host = nullptr;       // pinned: the gave-up counter before / after a host-pointer call's kernels (StarveCheck)
  int vb_blocks = 0;
  uint32_t* inv_fail = nullptr;          // device counter of the -DD377_CHECK_INVARIANTS build (always allocated)
  ScratchGuard vb_guard;
  hipStream_t stream = nullptr;          // compute stream of the host-pointer entry points
  hipStream_t copy_stream = nullptr;     // PCIe copies of the pipelined host path
  hipStream_t ctl_stream = nullptr;      // highest priority: d377_ctx_health / d377_ctx_reset_scratch (a hardware queue no kernel of ours waits in)
  hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
  // grow-only staging buffers for the host-pointer entry points (set 1 = second half of the
  // double buffer used when a large batch is pipelined chunk by chunk)
  uint8_t* buf[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap[4] = {0, 0, 0, 0};
  uint8_t* buf2[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t cap2[4] = {0, 0, 0, 0};
  // staging of the sharded device-pointer path (slices of a batch that lives on another device)
  uint8_t* shard[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t shard_cap[4] = {0, 0, 0, 0};
  hipEvent_t ev_shard = nullptr;
  MsmWorkspace msm;
  const Tuning* tune = nullptr;          // the owning context's overrides
  // the override for `key`, or `dflt` when the built-in rule is in force
  long long tuned(int key, long long dflt) const { return (tune && tune->v[key] >= 0) ? tune->v[key] : dflt; }
  bool is_tuned(int key) const { return tune && tune->v[key] >= 0; }
  // lanes the chunked kernels keep resident (WAVES_PER_SIMD workgroups of BLOCK lanes per CU): the unit the batch-size
  // thresholds of the launch rules are written in, so that they follow the device's CU count
  size_t resident_lanes() const { return (size_t)cus * WAVES_PER_SIMD * BLOCK; }
  SqrtTables tables() const { return SqrtTables{gtab, s_lookup, inv_fail}; }
};

// How a chunked kernel's rounds (of BLOCK elements) are dealt out to its workgroups.  `places` workgroups are resident at a
// time and a chunk holds at most kmax rounds (elements per lane).  Up to one round per place: a workgroup per round.
// Otherwise G = the fewest generations of resident workgroups that can hold the rounds: exactly G x places chunks, the rounds
// dealt out evenly, the first `extra` chunks one round longer (DcbScratch::extra) -- every generation is full and there is
// no last generation of a few stragglers.  (Chunks of kmax left 5 x 2^18 elements with 128 of them: sqrt_ratio_zeta 2.54 ms
// against 2.13 dealt evenly, variable base 24.1 against 19.4: profiles/r05_chunk_between_generations.txt.)  Whole multiples
// of places x kmax rounds -- 2^20, 2^21, 2^22 elements on 256 CUs -- come out as chunks of kmax, as before.  Beyond `cap`
// chunks the workgroups walk several chunks of kmax.
struct ChunkDeal { size_t per_lane, extra, nchunks; };
inline ChunkDeal deal_chunks(size_t rounds, size_t places, size_t kmax, size_t cap) {
  ChunkDeal c{1, 0, rounds};
  if (rounds <= places) return c;
  const size_t gens = (rounds + places * kmax - 1) / (places * kmax);
  if (gens * places <= cap) {
    c.nchunks = gens * places; c.per_lane = rounds / c.nchunks; c.extra = rounds % c.nchunks;
  } else {
    c.per_lane = kmax; c.nchunks = (rounds + kmax - 1) / kmax;
    if (c.nchunks > cap) c.nchunks = cap;
  }
  return c;
}

inline int grow(uint8_t*& p, size_t& cap, size_t bytes, size_t slack) {
  if (bytes <= cap) return D377_OK;
  if (p) HIP_TRY(hipFree(p));
  p = nullptr; cap = 0;
  HIP_TRY(hipMalloc(&p, bytes + slack));
  cap = bytes + slack;
  return D377_OK;
}
inline int ensure(DeviceState& d, int slot, size_t bytes) { return grow(d.buf[slot], d.cap[slot], bytes, bytes / 4 + 4096); }
inline int ensure2(DeviceState& d, int slot, size_t bytes) { return grow(d.buf2[slot], d.cap2[slot], bytes, 4096); }
inline int ensure_shard(DeviceState& d, int slot, size_t bytes) { return grow(d.shard[slot], d.shard_cap[slot], bytes, bytes / 4 + 4096); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Synchronises a set of streams when the enclosing function returns with an error after work has been
// enqueued: no copy into caller memory (or into a local vector) may still be in flight once we return.
struct SyncOnError {
  int* rc;
  int device;
  hipStream_t a, b;
  ~SyncOnError() {
    if (*rc == D377_OK) return;
    char saved[sizeof d377_g_err];
    memcpy(saved, d377_g_err, sizeof saved);            // keep the first error text
    (void)hipSetDevice(device);
    if (a) (void)hipStreamSynchronize(a);
    if (b) (void)hipStreamSynchronize(b);
    memcpy(d377_g_err, saved, sizeof saved);
  }
};

// A host-pointer call must not return D377_OK with records a starved workgroup never wrote (dcb.hpp: after
// DCB_GIVE_UP_TICKS without a lane set a workgroup counts itself in health[2] and leaves).  The call copies the counter
// to pinned memory on its stream before its first kernel and after its last; the call's own final synchronisation
// covers both copies, and verdict() turns a counter that moved into D377_ERR_STARVED.  (A `_dev` launch of another
// stream that starves during the call fails it too: the device's scratch pool is unhealthy either way.)
struct StarveCheck {
  DeviceState& d;
  hipStream_t s;
  int before() { HIP_TRY(hipMemcpyAsync(&d.starve_host[0], d.pool_health + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s)); return D377_OK; }
  int after() { HIP_TRY(hipMemcpyAsync(&d.starve_host[1], d.pool_health + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s)); return D377_OK; }
  int verdict() const {                                     // after the stream has been synchronised
    if (d.starve_host[1] == d.starve_host[0]) return D377_OK;
    snprintf(d377_g_err, sizeof d377_g_err,
             "%u workgroup(s) found no free lane set for 10 s and left their output records unwritten (device %d): "
             "d377_ctx_health / d377_ctx_reset_scratch", d.starve_host[1] - d.starve_host[0], d.id);
    return D377_ERR_STARVED;
  }
};

// op codes of the generic launcher in d377.hip (shared with the sharded path)
enum Op { OP_SQRT, OP_DECOMPRESS, OP_COMPRESS, OP_ROUNDTRIP, OP_MUL_BASE, OP_MUL_VAR, OP_ENCODE, OP_HASH, OP_ADD, OP_DOUBLE,
          OP_EQ, OP_WIDE48, OP_WIDE64, OP_ENCODE_WIDE48, OP_ENCODE_WIDE64, OP_AFFINE, OP_NEG, OP_IS_IDENTITY, OP_FQ_BIN,
          OP_FQ_UN, OP_FQ_CHECKED, OP_FQ_TO_BYTES, OP_FR_MOD, OP_FR_CHECKED, OP_MUL_VAR_EL, OP_MUL_BASE_EL, OP_COMPRESS_FIELD,
          OP_ENCODE_EL, OP_HASH_EL, OP_FR_BIN, OP_FR_UN, OP_FR_WIDE48, OP_FR_WIDE64 };

// D377_DEBUG_DEVICE_DELAY_MS (tests only): every per-device worker of a multi-device host call sleeps this
// long before it touches its device, which makes "the devices work concurrently" observable on a box
// with a single GPU listed twice.
int debug_device_delay_ms();

}  // namespace d377

struct d377_ctx {
  std::vector<d377::DeviceState> devs;
  std::mutex mu;
  d377::Tuning tune;
  std::vector<int> peer;                  // [a][b]: 1 = the same physical device, 2 = peer access a -> b enabled by d377_ctx_create, 0 = none
};
