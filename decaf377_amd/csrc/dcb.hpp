// dcb.hpp -- the chunked kernels' shared machinery (device only): lane sets of the per-device scratch areas claimed by
// workgroups, the round records of the batched inversions, and the walk of a workgroup through its chunks.
// Used by the batch kernels of d377.hip and by the MSM's decoding pass (msm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "curve.hpp"
#include "device_util.hpp"
#include "row_ops.hpp"

namespace d377 {

// Lane sets and elements per inversion are per kernel: every chunked kernel runs WAVES_PER_SIMD workgroups per CU with
// DCB_K elements per lane, except the fixed-base multiplication (FB_SETS, FB_K: see k_scalar_mul_base).  The scratch
// layout is sized for the largest of each and is the same for all of them.
constexpr int DCB_KMAX = 16;
#ifndef D377_FB_SETS
#define D377_FB_SETS 3                // (4: 128 VGPRs, 13 of them spilled, the same time at 2^20 ... 2^23: profiles/r05_ab_fb_sets.txt)
#endif
constexpr int FB_SETS = D377_FB_SETS, FB_K = 16;
constexpr int FB_WIDE_GENERATIONS = 1;            // generations of full narrow chunks beyond which the fixed-base kernel takes FB_SETS / FB_K
constexpr int DCB_SETS_MAX = FB_SETS > WAVES_PER_SIMD ? FB_SETS : WAVES_PER_SIMD;
// Beyond one generation of DCB_K elements per resident lane (2^20 on 256 CUs) a chunk grows to DCB_K_LONG per lane before
// the launch takes a further generation of workgroups: with issue priority by progress (below) a launch of one or two
// generations has next to no tail, so fewer, longer generations win -- 8 -> 16 per lane: variable base -1.4 % at 2^22
// (61.2 -> 60.4 ms, same box), -2.2 % at 3 x 2^20, sqrt -3 to -7 %, encode_to_curve / hash_to_curve -2 % at 2^22, unchanged
// at 2^23 (profiles/r05_ab_progress_priority.txt; under the arbiter's own order 16 per lane had LOST 1.6 %: curve.hpp).
constexpr int DCB_K_LONG = DCB_KMAX;
static_assert(DCB_K <= DCB_KMAX && FB_K <= DCB_KMAX, "the scratch layout has DCB_KMAX record rows per slot");

// Round storage of the batched inversions (curve.hpp: dcb_invert_slot, dcb_finish): [5 slots][DCB_KMAX][lanes] 32-byte
// records in global scratch, so that a wave reads and writes 2 KiB contiguous.  Slots 0..3: the denominators of the
// round's square roots / their inverses, later the compressor's state; slot 4: prefix products.  The compressor
// parks its prefix products in the output records of the elements they belong to.
//
// The areas (these records, the variable-base window tables) exist once per RESIDENT lane -- 2 workgroups per CU, 3 for
// the fixed-base kernel's wide launch -- but the grid is oversubscribed: a workgroup takes one chunk of DCB_K x 256 consecutive elements (DCB_K per lane)
// and there are as many workgroups as chunks.  Measured at 2^22 variable-base elements: 64.2 ms with exactly the
// resident workgroups walking 32 elements per lane each, 61.4 ms with four generations of workgroups of 8 per lane,
// although the latter pays four times as many inversions (profiles/README.md).  A workgroup therefore claims one of
// the area's `nslots` lane sets when it starts (an atomic on a small pool) and frees it when it is done.  The launch
// configuration keeps at most `nslots` workgroups of these kernels resident (d377_ctx_create checks it with the
// occupancy query and pads a kernel's LDS allocation when its registers alone would let more in: d377.hip
// check_residency; msm.hip asks the same question for its decoding pass before it uses it), so a free set normally exists; should residency ever exceed the sets -- two such kernels from different
// streams sharing a CU -- the extra workgroup sleeps and retries until a holder, which never waits on anything,
// finishes.  Claims are atomic and nothing resets the pool between launches, so kernels from different streams
// (a replayed hipGraph next to an eager call) can share the areas safely.
//
// A claim is the claiming workgroup's ticket (a serial number, never 0), so that the host can tell a set that one
// workgroup has held for seconds -- leaked by a launch that died -- from sets that change hands (d377_ctx_reset_scratch).
// A workgroup that finds no free set for DCB_STUCK_TICKS counts itself in health[1] and keeps waiting; after
// DCB_GIVE_UP_TICKS it counts itself in health[2] and leaves without touching its elements, so that no launch spins
// forever on a pool whose sets were leaked.  Outputs of such a launch are unwritten, and the call FAILS: the
// host-pointer entry points read health[2] on their stream before and after their kernels and return
// D377_ERR_STARVED when it moved (d377.hip: StarveCheck); a `_dev` caller does the same with the counter's device
// address (d377_ctx_starved_counter_dev) or asks d377_ctx_health.
struct DcbScratch {
  uint8_t* rec;        // [DCB_SLOTS][DCB_KMAX][lanes] 32-byte records, lanes = all the lane sets of the device x BLOCK
  int* pool;           // one word per lane set, 0 = free, else the holder's ticket (cleared at context creation; every workgroup frees what it claimed)
  int nslots;          // the sets THIS kernel may claim: the first nslots (its resident workgroups: 2 or 3 per CU)
  int per_lane;        // elements per lane in a chunk, 1 .. the kernel's K: smaller for small batches, so that the grid still fills the chip
  int lanes;           // the layout's lane count: the same for every kernel, so a set is the same memory whoever claims it
  int extra;           // launches of one workgroup per chunk: the first `extra` workgroups take per_lane + 1 elements per lane (0: every chunk alike)
  uint32_t* health;    // [0] ticket counter, [1] workgroups that waited DCB_STUCK_TICKS for a set, [2] workgroups that gave up
  int prio = 1;        // issue priority by progress (dcb_progress_priority below); 0: the launch leaves the arbiter alone
};
constexpr uint64_t DCB_STUCK_TICKS = 25000000ull;        // wall_clock64() ticks (100 MHz): 0.25 s
constexpr uint64_t DCB_GIVE_UP_TICKS = 1000000000ull;    // 10 s
struct DcbIO {
  uint8_t* scratch;
  uint8_t* out32;
  size_t nlanes, lane, base;            // lane of the claimed set; the chunk's j-th element of this lane is record base + j * BLOCK
  int slot, per_lane, extra;
  int* claim;                           // this workgroup's word of the pool (holds its ticket)
  uint32_t* tickets;                    // the ticket counter (health[0])
  int prio;                             // DcbScratch::prio
  __device__ __forceinline__ size_t rec(int sl, int j) const { return (size_t)(sl * DCB_KMAX + j) * nlanes + lane; }
  __device__ __forceinline__ void put(int sl, int j, const uint32_t w[8]) { store32(scratch, rec(sl, j), w); }
  __device__ __forceinline__ void get(int sl, int j, uint32_t w[8]) const { load32(scratch, rec(sl, j), w); }
  __device__ __forceinline__ void park(int j, const uint32_t w[8]) { store32(out32, base + (size_t)j * BLOCK, w); }
  __device__ __forceinline__ void parked(int j, uint32_t w[8]) const { load32(out32, base + (size_t)j * BLOCK, w); }
  __device__ __forceinline__ void emit(int j, const uint32_t w[8]) { store32(out32, base + (size_t)j * BLOCK, w); }
};
constexpr int DCB_SLOTS = 5;

// -> the claimed set, or -1 when the workgroup gave up (every lane of the workgroup gets the same answer)
__device__ __forceinline__ int dcb_claim(const DcbScratch& sc) {
  __shared__ int s_slot;
  if (threadIdx.x == 0) {
    const int ticket = (int)(atomicAdd(&sc.health[0], 1u) & 0x7FFFFFFFu) + 1;
#ifdef D377_SLOT_ROT                                     // experiment: a workgroup's first choice of lane set shifted by one (set parity against XCD parity)
    int s = (int)((blockIdx.x + 1u) % (unsigned)sc.nslots);
#else
    int s = (int)(blockIdx.x % (unsigned)sc.nslots);
#endif
    int tries = 0;
    uint64_t t0 = 0;
    bool counted = false;
    while (atomicCAS(&sc.pool[s], 0, ticket) != 0) {
      s = s + 1 == sc.nslots ? 0 : s + 1;
      if (++tries >= sc.nslots) {                                                // a whole lap without a free set: back off
        __builtin_amdgcn_s_sleep(32);
        tries = 0;
        const uint64_t now = wall_clock64();
        if (t0 == 0) t0 = now;
        if (!counted && now - t0 > DCB_STUCK_TICKS) { atomicAdd(&sc.health[1], 1u); counted = true; }
        if (now - t0 > DCB_GIVE_UP_TICKS) { atomicAdd(&sc.health[2], 1u); s = -1; break; }
      }
    }
    s_slot = s;
  }
  __syncthreads();
  return s_slot;
}
__device__ __forceinline__ void dcb_release(const DcbScratch& sc, int slot) {
  __syncthreads();                       // every lane of the workgroup is done with the set
  if (threadIdx.x == 0) {
    __threadfence();
    atomicExch(&sc.pool[slot], 0);
  }
}

// A workgroup's walk through its chunks (normally one): phase 0 leaves the denominators of the chunk's square roots
// in records 0 .. NINV-1, they are inverted together (one divsteps inversion per lane), phase 1 does the element's
// work with those inverses, and when the operation ends in an encoding of a point whose isogeny preimage it knows
// (FINISH) the square-root-free compressor closes the chunk.
// phase1(i, j, inv, have): inv[s] = the eight words of 1 / (denominator s of element j) (fe_from_words makes them a field
// element) when `have`; otherwise there are no inverses this launch: take the square roots in the reference's
// inversion-free form.
// A shared inversion costs ~26 000 instructions per lane plus 4 products per element; the chain it replaces in each
// square root (den^(2^47-1), 46 S + 9 M) ~9 500: with fewer than DCB_ASSIST_MIN elements per lane the inversion loses,
// and at those batch sizes (n <= 2 x the resident lanes) its latency is the whole call.  SMALL_OK = false keeps a
// kernel on the always-assisted form (k_scalar_mul_var: its codegen is left exactly as it was measured).
constexpr int DCB_ASSIST_MIN = 3;
// Issue priority by progress.  The waves that share a SIMD (one of each resident workgroup) do the same work, and the
// arbiter serves the OLDEST first: left alone, the older wave runs at a lone wave's rate (0.87 of the slots), the younger
// one gets the rest (0.10), and once the older workgroup is done the younger runs alone for 0.89 of its chunk --
// measured per workgroup at 2^20 (one generation: tools/wg_times.py, profiles/r05_wg_times.txt): durations from 980 to
// 1 850 us for identical work, the kernel as long as the slowest.  A launch of several generations hides it (the CU is
// refilled), a launch of one pays ~5 %.  So a wave lowers its priority as it gets through its chunk -- 3 for the first
// half, 2 for the next quarter, then 1, and 0 for the last element -- and the wave that is behind always outranks the one
// ahead: they take turns and end within the last stretch of each other.  2^20 elements (profiles/r05_ab_progress_priority.txt):
// sqrt -8 %, encode_to_curve / hash_to_curve -6 / -7 %, round trip -4 %, variable and fixed base -5 %; 2^22: within +-1 %.
// The waves of a SIMD then run in step, which the fixed-base kernel's WIDE launch does not like beyond one generation (its
// additions wait on table gathers, and waves in step gather in bursts: +4 to +10 % at 2^22), so a launch can leave the arbiter
// alone (`on`): the host asks for priorities in launches of one or two generations of workgroups (d377.hip: chunks_of).
#ifndef D377_DCB_PRIORITY
#define D377_DCB_PRIORITY 1
#endif
// (s_setprio is a scalar instruction: under a condition the compiler holds in vector registers it is predicated by EXEC
// only -- it runs whatever the mask says -- so every condition around one is forced into scalar registers first.)
__device__ __forceinline__ void dcb_progress_priority(int on, int j, int per_lane) {   // all uniform over the wave
#if D377_DCB_PRIORITY
  on = __builtin_amdgcn_readfirstlane(on); j = __builtin_amdgcn_readfirstlane(j); per_lane = __builtin_amdgcn_readfirstlane(per_lane);
  if (!on) return;
  const int left = per_lane - j;                   // elements of the chunk still to do, this one included
  if (left * 2 > per_lane) __builtin_amdgcn_s_setprio(3);
  else if (left * 4 > per_lane) __builtin_amdgcn_s_setprio(2);
  else if (left > 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#else
  (void)on; (void)j; (void)per_lane;
#endif
}
// The same at a finer grain, for a kernel that counts steps of its own (k_msm_spans: additions): `done` of `total` steps are
// behind the wave; 3 up to half of them, 2 up to three quarters, 1 up to the last `last`, then 0.  Called once per step.
// (At the grain of the variable-base kernels' windows instead of their elements: -1 to -3 % between 2^17 and 2^19, but +2 %
// at 2^16 and, for the Element form, from 2^20 on: not kept -- profiles/r05_ab_progress_priority.txt.)
__device__ __forceinline__ void dcb_progress_priority_steps(int done, int total, int last) {
#if D377_DCB_PRIORITY
  done = __builtin_amdgcn_readfirstlane(done); total = __builtin_amdgcn_readfirstlane(total); last = __builtin_amdgcn_readfirstlane(last);
  if (done == 0) __builtin_amdgcn_s_setprio(3);
  else if (done == total / 2) __builtin_amdgcn_s_setprio(2);
  else if (done == total - total / 4) __builtin_amdgcn_s_setprio(1);
  else if (done == total - last) __builtin_amdgcn_s_setprio(0);
#else
  (void)done; (void)total; (void)last;
#endif
}
// post(io, cnt): runs after a chunk's outputs have been written (a kernel's rare fix-ups: k_hash_to_curve).
struct DcbNoPost { __device__ __forceinline__ void operator()(DcbIO&, int) const {} };
template <int NINV, bool FINISH, bool SMALL_OK = true, class PT, class P0, class P1, class PF = DcbNoPost>
__device__ __forceinline__ void dcb_rounds(size_t n, DcbIO& io, PT& pt, P0 phase0, P1 phase1, PF post = PF()) {
  constexpr int NW = NINV > 0 ? NINV : 1;
  // chunk c covers the elements from BLOCK x (c x per_lane + min(c, extra)): the first `extra` chunks are one round longer
#ifdef D377_DCB_UNIFORM_AB                                                          // A/B only: round 3's loop (uniform chunks)
  const int per_lane = io.per_lane;
  const bool assist = NINV > 0 && (!SMALL_OK || per_lane >= DCB_ASSIST_MIN);
  const size_t CHUNK = (size_t)per_lane * BLOCK;
  for (size_t chunk = blockIdx.x; chunk * CHUNK < n; chunk += gridDim.x) {
    io.base = chunk * CHUNK + threadIdx.x;
#elif defined(D377_DCB_TICKETS)
  // A/B only (tools/build_variant.sh tickets -DD377_DCB_TICKETS): as many workgroups as are resident, each taking chunks by
  // TICKET from a per-launch counter (health[3], zeroed by the host before the launch) until they run out -- an XCD whose
  // clock is higher takes more chunks than a slower one, where chunks dealt by workgroup id give every XCD the same number.
  const bool assist = NINV > 0 && (!SMALL_OK || io.per_lane + (io.extra != 0 ? 1 : 0) >= DCB_ASSIST_MIN);
  __shared__ unsigned s_chunk_;
  unsigned taken = 0;
  for (;;) {
    if (threadIdx.x == 0) s_chunk_ = atomicAdd(io.tickets + 3, 1u);
    __syncthreads();
    const unsigned chunk = s_chunk_;
    __syncthreads();
    const bool longer = chunk < (unsigned)io.extra;
    int per_lane = io.per_lane + (longer ? 1 : 0);
    const size_t first = ((size_t)chunk * (unsigned)io.per_lane + (longer ? chunk : (unsigned)io.extra)) * BLOCK;
    if (first >= n) break;
    io.base = first + threadIdx.x;
    if (taken++ != 0 && threadIdx.x == 0)
      atomicExch(io.claim, (int)(atomicAdd(io.tickets, 1u) & 0x7FFFFFFFu) + 1);
#else
  // (a launch with extra != 0 has as many workgroups as chunks: only a workgroup's first chunk can be a long one)
  const bool longer = blockIdx.x < (unsigned)io.extra;
  int per_lane = io.per_lane + (longer ? 1 : 0);
  size_t first = ((size_t)blockIdx.x * (unsigned)io.per_lane + (longer ? blockIdx.x : (unsigned)io.extra)) * BLOCK;
  // uniform over the launch (the compiler specialises the loop on it): shared inversions as soon as SOME workgroup has
  // DCB_ASSIST_MIN elements per lane
  const bool assist = NINV > 0 && (!SMALL_OK || io.per_lane + (io.extra != 0 ? 1 : 0) >= DCB_ASSIST_MIN);
  for (unsigned chunk = blockIdx.x; first < n;
       chunk += gridDim.x, per_lane = io.per_lane, first = ((size_t)chunk * (unsigned)io.per_lane + (unsigned)io.extra) * BLOCK) {
    io.base = first + threadIdx.x;
    // A workgroup that walks several chunks draws a new ticket for each: the word in the pool then changes as long as its
    // holder makes progress, however long the launch (d377_ctx_reset_scratch frees only sets whose ticket stood still).
    if (chunk != blockIdx.x && threadIdx.x == 0)
      atomicExch(io.claim, (int)(atomicAdd(io.tickets, 1u) & 0x7FFFFFFFu) + 1);
#endif
    int cnt = 0;
    dcb_progress_priority(io.prio, 0, per_lane);
#pragma unroll 1
    for (int j = 0; j < per_lane; ++j) {
      const size_t i = io.base + (size_t)j * BLOCK;
      if (i >= n) break;
      if (assist) phase0(i, j);
      cnt = j + 1;
    }
    if (assist) {
#pragma unroll 1
      for (int sl = 0; sl < NINV; ++sl) {
#if defined(D377_DCB_LANE_INVERSIONS)                  // A/B: every lane its own divsteps inversion (rounds 2-4)
        dcb_invert_slot(io, sl, cnt);
#else                                                  // one inversion per WAVE (row_ops.hpp fe_invert_lanes); all lanes are here
        dcb_invert_slot_with(io, sl, cnt, [](const fe& c) { return row::fe_invert_lanes(c); });
#endif
      }
    }
#pragma unroll 1
    for (int j = 0; j < cnt; ++j) {
      dcb_progress_priority(io.prio, j, per_lane);
      uint32_t cur[NW][8] = {};
      if (assist) {
#pragma unroll
        for (int sl = 0; sl < NINV; ++sl) io.get(sl, j, cur[sl]);
      }
      phase1(io.base + (size_t)j * BLOCK, j, cur, assist);
    }
#if defined(D377_DCB_LANE_INVERSIONS)
    if (FINISH) dcb_finish(pt, io, cnt);
#else
    if (FINISH) dcb_finish_with(io, cnt, [](const fe& c) { return row::fe_invert_lanes(c); });
#endif
    post(io, cnt);
  }
}
// Developer variant (tools/build_variant.sh wgtimes -DD377_WG_TIMES; tools/wg_times.py): every workgroup of a chunked kernel of
// d377.hip leaves its entry time, the time it held a lane set, its end time (wall_clock64: 100 MHz) and where it ran
// (XCC_ID, HW_ID).  Not in the product build.
#ifdef D377_WG_TIMES
constexpr int WG_TIMES_MAX = 16384;
static __device__ unsigned long long g_wg_times[WG_TIMES_MAX * 6];   // entry, set claimed, end (100 MHz); XCC_ID | HW_ID; clock64() at claim and end
#define D377_WG_T0() const unsigned long long wg_t0_ = wall_clock64()
#define D377_WG_T1()                                                                              \
  if (threadIdx.x == 0 && blockIdx.x < (unsigned)WG_TIMES_MAX) {                                  \
    g_wg_times[blockIdx.x * 6 + 0] = wg_t0_;                                                      \
    g_wg_times[blockIdx.x * 6 + 1] = wall_clock64();                                              \
    g_wg_times[blockIdx.x * 6 + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg(6164) << 32) | /* XCC_ID[3:0] */ \
                                     (unsigned)__builtin_amdgcn_s_getreg(63492);                   /* HW_ID */       \
    g_wg_times[blockIdx.x * 6 + 4] = clock64();                                                   \
  }
#define D377_WG_T2()                                                                              \
  if (threadIdx.x == 0 && blockIdx.x < (unsigned)WG_TIMES_MAX) {                                  \
    g_wg_times[blockIdx.x * 6 + 2] = wall_clock64();                                              \
    g_wg_times[blockIdx.x * 6 + 5] = clock64();                                                   \
  }
#else
#define D377_WG_T0()
#define D377_WG_T1()
#define D377_WG_T2()
#endif
#define D377_DCB_BEGIN(out_ptr)                                                                   \
  D377_WG_T0();                                                                                   \
  const int dcb_slot_ = dcb_claim(dcb);                                                           \
  D377_WG_T1();                                                                                   \
  if (dcb_slot_ < 0) return;                      /* no set for DCB_GIVE_UP_TICKS: counted in health[2] */ \
  DcbIO io{dcb.rec, reinterpret_cast<uint8_t*>(out_ptr), (size_t)dcb.lanes,                      \
           (size_t)dcb_slot_ * BLOCK + threadIdx.x, 0, dcb_slot_, dcb.per_lane, dcb.extra,       \
           dcb.pool + dcb_slot_, dcb.health, dcb.prio}
#define D377_DCB_END() dcb_release(dcb, dcb_slot_); D377_WG_T2()

}  // namespace d377
