/*
 * decaf377_amd.h -- C ABI of the MI355X batch group-operation engine for decaf377.
 *
 * The reference crate (penumbra-zone/decaf377 v0.10.1) has no FFI or plugin interface of its
 * own (it is a #![no_std] Rust library, src/lib.rs:1-31); the boundary below is what a
 * Rust `extern "C"` block for this hot path would bind (INTEGRATION.md shows that block).
 * Every entry point is a pure map over packed, unpadded, element-major arrays: element i of
 * every output equals the reference function applied to element i of the inputs.
 *
 * Records
 *   Encoding   32 bytes, `Encoding(pub [u8; 32])`                 src/ark_curve/encoding.rs:14-15
 *   Fq / Fr    32 bytes little-endian                              src/fields/fq.rs:90-115, fr.rs:82-107
 *   Element    16 x u64 = X, Y, Z, T, each 4 Montgomery limbs (R = 2^256, fully reduced), the
 *              in-memory value of `Fq::from_montgomery_limbs`      src/fields/fq/u64/wrapper.rs:82-85
 *   status     1 byte per element: 0 = Ok, 1 = EncodingError::InvalidEncoding   src/error.rs:1-5
 *              (failed elements get all-zero output records)
 *
 * Return value: 0 on success, negative D377_ERR_* otherwise; per-element failures are only
 * reported through `status`.  Buffers belong to the caller and are never retained.  An output buffer may
 * be the input buffer of the same record size (in place); until a call completes its output records may hold
 * intermediate values.
 *
 * Threads and streams.  Calls on one context are serialised by a mutex inside the context, so
 * several host threads may share it; distinct contexts are independent.  The `_dev` entry points
 * return as soon as the work is enqueued, and calls on DIFFERENT streams may be in flight at the
 * same time: the per-device scratch areas some kernels use (the variable-base window tables, the records of
 * the batched inversions behind scalar_mul_var / scalar_mul_base / encode_to_curve[_wide] / hash_to_curve /
 * sqrt_ratio_zeta, the MSM workspace) are handed from one launch to the next by events on the device, so such launches queue
 * up behind each other instead of racing -- results are the same as if the calls had been made one
 * after another; only their overlap is lost.  Kernels that use no scratch overlap freely.
 * `_dev` calls only enqueue kernels (no host synchronisation, no allocation once the workspaces have grown
 * to the batch size), so a sequence of them can be captured into a hipGraph and replayed.  A capturing stream takes
 * no part in the event hand-over (the graph keeps its own order), so for a REPLAY that overlaps other calls on the same
 * device:
 *   - the lane-set areas (window tables, inversion records: every batch operation above) are safe -- each workgroup
 *     claims a free set atomically and nothing resets the pool between launches, so a replay and an eager call, or two
 *     replays, simply share them;
 *   - the MSM workspace is exclusive: do not let a replay that contains d377_msm*_dev overlap another MSM of the same
 *     device on a different stream (order them with the same stream or your own events);
 *   - a workspace never shrinks, and once an MSM has been captured an outgrown workspace is kept until
 *     d377_ctx_destroy instead of freed, so an old graph stays valid after later, larger calls; an MSM that would have
 *     to grow the workspace DURING a capture fails with D377_ERR_ARG (run one call of that size first).
 *
 * Three families:
 *   d377_batch_*          host pointers; the library copies to the context's GPU(s), shards
 *                         contiguous slices over them when the context owns several (one host
 *                         thread per device, so the devices run concurrently), and copies back.
 *   d377_batch_*_dev      device pointers (16-byte aligned, resident on the context's device
 *                         `dev`), enqueued on `stream` (a hipStream_t, NULL = default stream) with
 *                         no host synchronisation: for callers that keep batches in HBM.
 *   d377_batch_sharded_dev  a batch resident in the HBM of ONE device of a multi-GPU context:
 *                         contiguous slices go to the other devices by peer copies over xGMI, every
 *                         device runs the kernel, the outputs come back; ordered on `stream`.
 */
#ifndef DECAF377_AMD_H
#define DECAF377_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D377_OK 0
#define D377_ERR_HIP (-1)        /* a HIP runtime call failed; see d377_last_error() */
#define D377_ERR_ARG (-2)        /* null / misaligned pointer, bad device index */
#define D377_ERR_NO_DEVICE (-3)  /* no usable gfx950 device */
#define D377_ERR_INIT (-4)       /* device table self-check failed at context creation */
#define D377_ERR_STARVED (-5)    /* workgroups of this call found no free lane set for 10 s and wrote no output: the
                                    output buffers are NOT valid (d377_ctx_health, d377_ctx_reset_scratch below) */

typedef struct d377_ctx d377_ctx;

/* Library / device discovery. */
const char* d377_version(void);
int d377_device_count(void);
/* Human-readable text for the most recent error on this thread. */
const char* d377_last_error(void);

/* Builds the read-only device tables (Sarkar square-root tables of
 * src/ark_curve/invsqrt.rs:14-66, fixed-base table of Element::GENERATOR) once per device
 * and allocates the per-device scratch (about 6.7 GB of HBM per device: 0.73 GB of window tables and records of the
 * batched inversions, and the 5.9 GB fixed-base comb of 23-bit windows, built in 16 ms).  device_ids == NULL,
 * n_dev == 0 -> device 0.  The comb is the one cost a caller may not want: in the reference Element::GENERATOR is a
 * constant (src/min_curve/element.rs:61-81) and `GENERATOR * Fr` costs nothing until it is used -- d377_ctx_create_ex
 * makes it so here too. */
int d377_ctx_create(const int* device_ids, int n_dev, d377_ctx** out);
/* The same with options (NULL = d377_ctx_create's defaults).
 *   size        sizeof(d377_ctx_opts) as the caller was compiled (fields added later default to 0)
 *   comb_bits   width of the fixed-base comb: 0 = the library's default (23), or 18 / 21 / 23 --
 *               0.24 / 1.6 / 5.9 GB per device, 14 / 12 / 11 additions per scalar (18 is 9-12 % slower than 23 on
 *               d377_batch_scalar_mul_base*, profiles/r05_ab_fixed_base_wide.txt); results are the same bytes
 *   comb_lazy   0: the comb is built at context creation.  1: by the first fixed-base call on each device
 *               (d377_batch_scalar_mul_base[_element][_dev], d377_batch_sharded_dev with those operations), which then
 *               allocates, builds (16 ms at 23 bits) and synchronises once -- so that first call must not be inside a
 *               stream capture (D377_ERR_ARG); a context that never multiplies by the generator holds 0.73 GB per device.
 * A failed comb allocation is D377_ERR_HIP with a text that names the width and the smaller ones.
 * d377_ctx_comb_info: the width of device `dev`'s comb, whether it has been built, and its size in bytes (any pointer
 * may be NULL). */
typedef struct d377_ctx_opts {
  size_t size;
  int comb_bits;
  int comb_lazy;
} d377_ctx_opts;
int d377_ctx_create_ex(const int* device_ids, int n_dev, const d377_ctx_opts* opts, d377_ctx** out);
int d377_ctx_comb_info(d377_ctx* ctx, int dev, int* comb_bits, int* built, uint64_t* table_bytes);
void d377_ctx_destroy(d377_ctx* ctx);
int d377_ctx_num_devices(const d377_ctx* ctx);
int d377_ctx_device_id(const d377_ctx* ctx, int dev);
/* What d377_ctx_create arranged between two devices of the context: 2 = they are distinct GPUs and peer access
 * dev_a -> dev_b (xGMI) was enabled, 1 = the same physical GPU listed twice, 0 = distinct GPUs without peer access
 * (d377_batch_sharded_dev then relies on the runtime's staged copies), -1 = bad index. */
int d377_ctx_peer_access(const d377_ctx* ctx, int dev_a, int dev_b);
/* The lane-set scratch areas hold `sets_per_cu` sets per compute unit (3: the fixed-base kernel claims among all of
 * them, the other kernels among the first 2 per CU); d377_ctx_create asks the runtime
 * (hipOccupancyMaxActiveBlocksPerMultiprocessor) how many workgroups of each kernel that claims a set can be resident
 * per CU, pads a kernel's launch with dynamic LDS when its registers alone would admit more than the sets it may claim,
 * and fails with D377_ERR_INIT if one still exceeds them.  Reports the numbers it settled on: the largest residency over those
 * kernels (<= sets_per_cu) and the largest LDS padding in use (0 when none was needed).  Any pointer may be NULL. */
int d377_ctx_chunk_residency(const d377_ctx* ctx, int dev, int* sets_per_cu, int* max_blocks_per_cu, int* lds_pad_bytes);
/* The lane-set pool's health and its way back.  Every workgroup of a chunked kernel claims one lane set of the scratch
 * areas and frees it when it is done; a launch that dies mid-kernel (a fault in another kernel of the process, a killed
 * graph) would leave its sets claimed for the life of the context.  Later launches then wait: a workgroup that finds
 * no free set for 0.25 s is counted in *waited_long and keeps waiting; after 10 s it is counted in *gave_up and leaves
 * WITHOUT writing its output records (so no launch can spin forever).  That is never a silent wrong answer -- the
 * reference's fallible operations always return a Result (src/ark_curve/encoding.rs:34-60):
 *   - every host-pointer entry point (d377_batch_*, d377_msm[_encoded]) reads the gave-up counter on its stream before
 *     its first and after its last kernel and returns D377_ERR_STARVED when it moved; the output buffers are then invalid;
 *   - a `_dev` call only enqueues, so its caller checks: d377_ctx_starved_counter_dev gives the DEVICE address of the
 *     32-bit counter (valid for the life of the context, only ever incremented); copy it on the call's stream before and
 *     after the call (two 4-byte copies, no synchronisation of their own) and compare once the stream has been
 *     synchronised, or call d377_ctx_health afterwards.  Either check is MANDATORY after any `_dev` call that took
 *     longer than 10 s before its outputs are trusted; a call that finishes sooner cannot have starved.
 * d377_ctx_health reports the sets claimed right now (0 on an idle device) and both counters since the context was created; it
 * does not wait for running kernels.  d377_ctx_reset_scratch frees leaked sets: it reads the pool, waits for the work
 * this context knows of (at least 0.25 s, at most 1.5 s), reads it again and frees exactly the sets whose ticket did
 * not change in between -- a workgroup draws a new ticket for every chunk it starts (milliseconds of work), so a
 * ticket that stood still belongs to no running workgroup, however long its launch and whether or not the context can
 * see it (a replayed hipGraph) -- then waits for starved work to finish.  *sets_released (may be NULL) = how many it
 * freed; 0 on a healthy context.
 * d377_debug_poison_pool marks `sets` sets (< 0: all) as claimed by nobody: the test hook for the calls above. */
int d377_ctx_health(d377_ctx* ctx, int dev, int* sets_claimed, uint64_t* waited_long, uint64_t* gave_up);
int d377_ctx_reset_scratch(d377_ctx* ctx, int dev, int* sets_released);
int d377_ctx_starved_counter_dev(d377_ctx* ctx, int dev, const uint32_t** counter_dev);
int d377_debug_poison_pool(d377_ctx* ctx, int dev, int sets);
/* Debug builds (-DD377_CHECK_INVARIANTS, the counterpart of the reference's debug assertions in
 * Element::new, src/min_curve/element.rs:104-110, and is_on_curve, src/ark_curve/on_curve.rs:14-39): how many
 * group elements failed the curve equation / T Z = X Y / Z != 0 after decompression, the Elligator map or on
 * their way into compression since the context was created.  Synchronises the device.  Returns 1 when the
 * checks are compiled in, 0 (and *count = 0) in a normal build, negative on error. */
int d377_ctx_invariant_failures(d377_ctx* ctx, int dev, uint64_t* count);

/* Developer interface: launch-rule overrides.  The library picks kernels, chunk shapes and MSM parameters from the
 * batch size and the device's CU count; these keys replace one rule each for A/B measurements and for tests that force
 * a route over sizes that would not take it (every route gives the same bytes).  Values are validated, stored in the
 * context, and read under its mutex by every later call on it (calls already enqueued keep what they were launched
 * with); D377_TUNE_DEFAULT restores the built-in rule.  Nothing in the library reads the process environment on a call
 * path (D377_DEBUG_* diagnostics at context creation and in the multi-device test hook aside).
 *   SMALL_MAX              scalar_mul_var[_element], scalar_mul_base[_element]: batches up to this many elements take the quad-per-element
 *                          kernel (0 = never; it also caps TINY_MAX)
 *   DECOMPRESS_CHUNKED_MIN decompress, compress, roundtrip: batches from this many elements run in chunks with shared inversions
 *   FB_WIDE                scalar_mul_base: 0 = narrow launch (2 workgroups per CU), 1 = wide (3 per CU) at every size the lane kernel takes
 *   FB_K                   scalar_mul_base: elements per lane per shared inversion, 1..16
 *   AFFINE_BLOCKS_PER_CU   to_affine: workgroups per CU that share the batch, >= 1
 *   MSM_WINDOW             msm: window width in bits, 4..18
 *   MSM_SEG                msm: sorted entries per span lane, 1..128 (built-in: entries / resident lanes, at least 8)
 *   MSM_SMALL_MAX          msm: batches up to this many points skip the buckets (0 = never)
 *   MSM_SLICES             msm: slices per window of the counting sort, 1..4096
 *   MSM_SORT_PACKED        msm: 0 = the sort's level-1 entries as a word and a byte (the form batches above 2^24 points take), 1 = one
 *                          packed word (refused above 2^24 points)
 *   MSM_RED / MSM_SKIP     msm: group size of the second reduction level (2..64) / leftovers a bucket lane sums itself (1..64)
 *   MSM_CHUNKED_SUMS       msm: 1 = weighted bucket sums by chunked running sums instead of the tree of bit-sums
 *   MSM_ENC_CHUNKED_MIN    msm_encoded: batches from this many points decode with shared inversions
 *   TINY_MAX               scalar_mul_var[_element], scalar_mul_base[_element]: batches up to this many elements take one wave per
 *                          element (0 = never); sqrt_ratio_zeta, decompress, compress, roundtrip, encode_to_curve, hash_to_curve:
 *                          batches up to FOUR times this many take four elements per wave (hash_to_curve up to twice: two pairs)
 *   MSM_TINY_MAX           msm: batches up to this many points take a wave per one to four points (0 = never; at most MSM_SMALL_MAX applies)
 *   CHUNK_PER_LANE         chunked kernels (sqrt_ratio_zeta, encode_to_curve[_wide], hash_to_curve, scalar_mul_var): elements per lane per chunk, 1..8
 * Returns D377_ERR_ARG for an unknown key or a value outside its range. */
#define D377_TUNE_SMALL_MAX 0
#define D377_TUNE_DECOMPRESS_CHUNKED_MIN 1
#define D377_TUNE_FB_WIDE 2
#define D377_TUNE_FB_K 3
#define D377_TUNE_AFFINE_BLOCKS_PER_CU 4
#define D377_TUNE_MSM_WINDOW 5
#define D377_TUNE_MSM_SEG 6
#define D377_TUNE_MSM_SMALL_MAX 7
#define D377_TUNE_MSM_SLICES 8
#define D377_TUNE_MSM_RED 9
#define D377_TUNE_MSM_SKIP 10
#define D377_TUNE_MSM_CHUNKED_SUMS 11
#define D377_TUNE_MSM_ENC_CHUNKED_MIN 12
#define D377_TUNE_CHUNK_PER_LANE 13
#define D377_TUNE_MSM_TINY_MAX 14
#define D377_TUNE_TINY_MAX 15
#define D377_TUNE_MSM_SORT_PACKED 16
#define D377_TUNE_COUNT 17
#define D377_TUNE_DEFAULT (-1)
int d377_ctx_set_tuning(d377_ctx* ctx, int key, int64_t value);
int d377_ctx_get_tuning(d377_ctx* ctx, int key, int64_t* value);

/* Fq::sqrt_ratio_zeta(num, den) -> (was_square, root)        src/ark_curve/invsqrt.rs:75-166
 * num32/den32: 32-byte strings reduced mod q like Fq::from_le_bytes_mod_order.
 * The crate's two backends return different roots (same flag, root negated about half the time):
 *   D377_SQRT_ROOT_ARK        the default `arkworks` backend, Sarkar's table method   src/ark_curve/invsqrt.rs:75-166
 *   D377_SQRT_ROOT_MIN_CURVE  `Fq::non_arkworks_sqrt_ratio_zeta`, constant-time Tonelli-Shanks seeded
 *                             with 11^m                       src/min_curve/invsqrt.rs:11-95, src/fields/fq.rs:62-67
 * d377_batch_sqrt_ratio_zeta returns the ARK root; the _ex forms take the convention.  Every group-level
 * output (encodings) is the same under either: decompress / compress / Elligator fix the sign themselves. */
#define D377_SQRT_ROOT_ARK 0
#define D377_SQRT_ROOT_MIN_CURVE 1
int d377_batch_sqrt_ratio_zeta(d377_ctx* ctx, const uint8_t* num32, const uint8_t* den32, size_t n,
                               uint8_t* root32, uint8_t* was_square);
int d377_batch_sqrt_ratio_zeta_ex(d377_ctx* ctx, int root, const uint8_t* num32, const uint8_t* den32, size_t n,
                                  uint8_t* root32, uint8_t* was_square);
/* Encoding::vartime_decompress                                src/ark_curve/encoding.rs:32-83 */
int d377_batch_decompress(d377_ctx* ctx, const uint8_t* enc32, size_t n, uint64_t* xyzt, uint8_t* status);
/* Element::vartime_compress                                   src/ark_curve/encoding.rs:91-128 */
int d377_batch_compress(d377_ctx* ctx, const uint64_t* xyzt, size_t n, uint8_t* enc32);
/* vartime_decompress then vartime_compress (tests/encoding.rs:97-107 round trip) */
int d377_batch_roundtrip(d377_ctx* ctx, const uint8_t* enc32, size_t n, uint8_t* enc32_out, uint8_t* status);
/* Element::GENERATOR * Fr::from_le_bytes_mod_order(scalar) -> Encoding
 *                                                             src/min_curve/ops.rs:89-95 */
int d377_batch_scalar_mul_base(d377_ctx* ctx, const uint8_t* scalar32, size_t n, uint8_t* enc32_out);
/* Encoding::vartime_decompress(enc)? * Fr::from_le_bytes_mod_order(scalar) -> Encoding */
int d377_batch_scalar_mul_var(d377_ctx* ctx, const uint8_t* enc32, const uint8_t* scalar32, size_t n,
                              uint8_t* enc32_out, uint8_t* status);
/* Element::encode_to_curve(Fq::from_le_bytes_mod_order(r)) -> Encoding
 *                                                             src/ark_curve/elligator.rs:15-62,74-76 */
int d377_batch_encode_to_curve(d377_ctx* ctx, const uint8_t* fq32, size_t n, uint8_t* enc32_out);
/* Element::hash_to_curve(r1, r2) -> Encoding                  src/ark_curve/elligator.rs:67-71 */
int d377_batch_hash_to_curve(d377_ctx* ctx, const uint8_t* r1_32, const uint8_t* r2_32, size_t n,
                             uint8_t* enc32_out);

/* The same operations with the reference's own signatures: Elements in, Elements out, no encoding step.
 *   Element * Fr                                               src/min_curve/ops.rs:89-95, element.rs:138-157
 *   Element::GENERATOR * Fr
 *   Element::vartime_compress_to_field -> Fq (4 Montgomery limbs) src/min_curve/element.rs:163-181
 *   Element::encode_to_curve / hash_to_curve -> Element         src/min_curve/element.rs:190-244
 * An Element returned by the scalar multiplications is some extended representative of the reference's group
 * element (same encoding, equal under d377_batch_eq); its X:Y:Z:T are those of the schedule here (signed
 * windows), not of the reference's double-and-add.  encode_to_curve / hash_to_curve return the coordinates
 * the reference's formulas give.  scalar32: any 32 bytes, reduced mod r. */
int d377_batch_scalar_mul_var_element(d377_ctx* ctx, const uint64_t* p_xyzt, const uint8_t* scalar32, size_t n,
                                      uint64_t* out_xyzt);
int d377_batch_scalar_mul_base_element(d377_ctx* ctx, const uint8_t* scalar32, size_t n, uint64_t* out_xyzt);
int d377_batch_compress_to_field(d377_ctx* ctx, const uint64_t* p_xyzt, size_t n, uint64_t* fq_out);
int d377_batch_encode_to_curve_element(d377_ctx* ctx, const uint8_t* fq32, size_t n, uint64_t* out_xyzt);
int d377_batch_hash_to_curve_element(d377_ctx* ctx, const uint8_t* r1_32, const uint8_t* r2_32, size_t n,
                                     uint64_t* out_xyzt);

/* Element + Element, Element - Element (= self + other.neg()), Element::double, decaf equality
 * (x1*y2 == x2*y1) on in-memory elements
 *            src/min_curve/element.rs:291-322, 119-136, 334-340; src/min_curve/ops.rs:15-87;
 *            ark: element/projective.rs:65-70
 * Results are the same extended coordinates the reference formulas produce. */
int d377_batch_add(d377_ctx* ctx, const uint64_t* p_xyzt, const uint64_t* q_xyzt, size_t n, uint64_t* out_xyzt);
int d377_batch_sub(d377_ctx* ctx, const uint64_t* p_xyzt, const uint64_t* q_xyzt, size_t n, uint64_t* out_xyzt);
int d377_batch_double(d377_ctx* ctx, const uint64_t* p_xyzt, size_t n, uint64_t* out_xyzt);
int d377_batch_eq(d377_ctx* ctx, const uint64_t* p_xyzt, const uint64_t* q_xyzt, size_t n, uint8_t* equal);

/* Fq::from_le_bytes_mod_order on 48- or 64-byte strings (hash outputs)      src/fields/fq.rs:90-102
 * -> canonical 32-byte Fq; and the same fused into encode_to_curve.  len must be 48 or 64. */
int d377_batch_fq_from_wide_bytes(d377_ctx* ctx, const uint8_t* bytes, size_t len, size_t n, uint8_t* fq32_out);
int d377_batch_encode_to_curve_wide(d377_ctx* ctx, const uint8_t* bytes, size_t len, size_t n, uint8_t* enc32_out);
/* CurveGroup::normalize_batch / into_affine                                   src/ark_curve/element.rs:74-85
 * xy: n x 8 u64 = affine x, y as 4 Montgomery limbs each. */
int d377_batch_to_affine(d377_ctx* ctx, const uint64_t* xyzt, size_t n, uint64_t* xy);

/* Element::vartime_multiscalar_mul(scalars, points) = sum_i scalar_i * point_i
 *                                                 src/ark_curve/element/projective.rs:99-117
 * (a fold of scalar multiplications in the reference; here a Pippenger bucket MSM, and for batches small enough to
 * give every point four lanes -- up to 64 per compute unit, 16384 on an MI355X -- one scalar multiplication per point
 * followed by a tree sum).  The sum is returned as its canonical Encoding (enc32_out, 32 bytes) and, if xyzt_out !=
 * NULL, as one Element record (some extended representative of the same group element).  Element records are read
 * as the extended coordinates they are (T Z = X Y, as every Element the reference or this library produces has
 * them); a record with Z = 0 is no group element and counts as the identity.
 * d377_msm takes in-memory Elements; d377_msm_encoded takes Encodings, reports invalid ones in
 * status[] and leaves them out of the sum.  n < 2^31.  With a multi-GPU context each device sums a
 * contiguous slice and the partial sums are added on the first device. */
int d377_msm(d377_ctx* ctx, const uint64_t* xyzt, const uint8_t* scalar32, size_t n, uint8_t* enc32_out,
             uint64_t* xyzt_out);
int d377_msm_encoded(d377_ctx* ctx, const uint8_t* enc32, const uint8_t* scalar32, size_t n, uint8_t* enc32_out,
                     uint64_t* xyzt_out, uint8_t* status);

/* MANY small multiscalar sums at once: sum i = scalar[i m] * point[i m] + ... + scalar[i m + m - 1] * point[i m + m - 1] for
 * i < n, m terms per sum, 1 <= m <= D377_BATCH_MSM_MAX_TERMS; enc32_out[i] = the canonical Encoding of sum i and, if
 * xyzt_out != NULL, xyzt_out[i] = sum i as an Element record (n x 16 u64; some extended representative, as d377_msm's).  This is
 * Element::vartime_multiscalar_mul (src/ark_curve/element/projective.rs:99-117) in the shape the reference's own test
 * exercises it -- a 3-term sum per case, tests/operations.rs:44-60 -- for callers that hold many such sums (d377_msm is ONE
 * long sum per call).  Each sum is one Straus chain: its m points share the 252 doublings, so a 3-term sum costs about half
 * of three scalar multiplications and two additions.  Points and scalars are term-major within a sum: n x m records each.
 * d377_batch_msm_small takes Elements (read as the extended coordinates they are, T Z = X Y; a record with Z = 0 counts as
 * the identity); _encoded takes Encodings, reports invalid ones in status[i m + j] (n x m bytes) and leaves them out of
 * their sum, like d377_msm_encoded.  Scratch: 0.23 GB of HBM per term on a 256-CU device, allocated on the first call
 * with that many terms (so that call must not be inside a stream capture).  A multi-GPU context slices the SUMS. */
#define D377_BATCH_MSM_MAX_TERMS 8
int d377_batch_msm_small(d377_ctx* ctx, const uint64_t* xyzt, const uint8_t* scalar32, size_t m, size_t n, uint8_t* enc32_out,
                         uint64_t* xyzt_out);
int d377_batch_msm_small_encoded(d377_ctx* ctx, const uint8_t* enc32, const uint8_t* scalar32, size_t m, size_t n,
                                 uint8_t* enc32_out, uint64_t* xyzt_out, uint8_t* status);

/* Fq field operations on in-memory elements (4 Montgomery u64 limbs, R = 2^256, fully reduced), the
 * unit everything above is built from             src/fields/fq/u64/wrapper.rs:99-132, fq/ops.rs
 * op: D377_FQ_ADD / SUB / MUL (binary, b != NULL) and D377_FQ_SQUARE / NEG / INVERSE (unary, b NULL).
 * INVERSE mirrors `Fq::inverse() -> Option<Fq>`: status[i] = 1 and a zero record for a zero input
 * (status may be NULL for the other ops).  d377_batch_fq_from_bytes_checked mirrors
 * Fq::from_bytes_checked (src/fields/fq.rs:108-115): status 1 for non-canonical strings. */
#define D377_FQ_ADD 0
#define D377_FQ_SUB 1
#define D377_FQ_MUL 2
#define D377_FQ_SQUARE 3
#define D377_FQ_NEG 4
#define D377_FQ_INVERSE 5
int d377_batch_fq_op(d377_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out,
                     uint8_t* status);
int d377_batch_fq_from_bytes_checked(d377_ctx* ctx, const uint8_t* bytes32, size_t n, uint64_t* out, uint8_t* status);
int d377_batch_fq_to_bytes(d377_ctx* ctx, const uint64_t* a, size_t n, uint8_t* bytes32);

/* Fr byte handling (the only part of Fr the hot path uses)            src/fields/fr.rs:82-107
 * from_le_bytes_mod_order: 32 raw bytes -> canonical 32-byte scalar (value mod r);
 * from_bytes_checked: status 1 for strings >= r (the record is copied through when canonical). */
int d377_batch_fr_from_le_bytes_mod_order(d377_ctx* ctx, const uint8_t* bytes32, size_t n, uint8_t* fr32_out);
int d377_batch_fr_from_bytes_checked(d377_ctx* ctx, const uint8_t* bytes32, size_t n, uint8_t* fr32_out, uint8_t* status);

/* Fr arithmetic                                                   src/fields/fr/u64/wrapper.rs:76-108
 * Scalars travel as 32-byte little-endian strings (Fr::to_bytes_le, wrapper.rs:63-70), as in the scalar
 * multiplications above: inputs are any 32 bytes (reduced mod r first), outputs canonical.  op: the D377_FQ_*
 * selectors (ADD / SUB / MUL binary, b32 != NULL; SQUARE / NEG / INVERSE unary, b32 NULL).  INVERSE mirrors
 * `Fr::inverse() -> Option<Fr>`: status[i] = 1 and a zero record for a zero input (status may be NULL for the
 * other ops).  d377_batch_fr_from_wide_bytes: Fr::from_le_bytes_mod_order on 48- or 64-byte strings (hash
 * outputs, `Fr::rand`)                                           src/fields/fr.rs:82-94, 118-126 */
int d377_batch_fr_op(d377_ctx* ctx, int op, const uint8_t* a32, const uint8_t* b32, size_t n, uint8_t* out32,
                     uint8_t* status);
int d377_batch_fr_from_wide_bytes(d377_ctx* ctx, const uint8_t* bytes, size_t len, size_t n, uint8_t* fr32_out);

/* -Element (x, t negated), Element::is_identity (x == 0), and the constants Element::IDENTITY /
 * Element::GENERATOR as one 16 x u64 record each      src/min_curve/element.rs:324-332, 113-117, 53-81 */
int d377_batch_neg(d377_ctx* ctx, const uint64_t* p_xyzt, size_t n, uint64_t* out_xyzt);
int d377_batch_is_identity(d377_ctx* ctx, const uint64_t* p_xyzt, size_t n, uint8_t* is_identity);
void d377_identity(uint64_t xyzt[16]);
void d377_generator(uint64_t xyzt[16]);

/* Device-pointer forms (same semantics). */
int d377_batch_sqrt_ratio_zeta_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* num32,
                                   const uint8_t* den32, size_t n, uint8_t* root32, uint8_t* was_square);
int d377_batch_sqrt_ratio_zeta_ex_dev(d377_ctx* ctx, int dev, void* stream, int root, const uint8_t* num32,
                                      const uint8_t* den32, size_t n, uint8_t* root32, uint8_t* was_square);
int d377_batch_decompress_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* enc32, size_t n,
                              uint64_t* xyzt, uint8_t* status);
int d377_batch_compress_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* xyzt, size_t n,
                            uint8_t* enc32);
int d377_batch_roundtrip_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* enc32, size_t n,
                             uint8_t* enc32_out, uint8_t* status);
int d377_batch_scalar_mul_base_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* scalar32,
                                   size_t n, uint8_t* enc32_out);
int d377_batch_scalar_mul_var_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* enc32,
                                  const uint8_t* scalar32, size_t n, uint8_t* enc32_out, uint8_t* status);
int d377_batch_encode_to_curve_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* fq32, size_t n,
                                   uint8_t* enc32_out);
int d377_batch_hash_to_curve_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* r1_32,
                                 const uint8_t* r2_32, size_t n, uint8_t* enc32_out);

int d377_batch_add_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, const uint64_t* q_xyzt,
                       size_t n, uint64_t* out_xyzt);
int d377_batch_sub_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, const uint64_t* q_xyzt,
                       size_t n, uint64_t* out_xyzt);
int d377_batch_double_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, size_t n,
                          uint64_t* out_xyzt);
int d377_batch_eq_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, const uint64_t* q_xyzt,
                      size_t n, uint8_t* equal);

int d377_batch_fq_op_dev(d377_ctx* ctx, int dev, void* stream, int op, const uint64_t* a, const uint64_t* b, size_t n,
                         uint64_t* out, uint8_t* status);
int d377_batch_neg_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, size_t n, uint64_t* out_xyzt);
int d377_batch_is_identity_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, size_t n,
                               uint8_t* is_identity);
int d377_batch_fq_from_wide_bytes_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes, size_t len,
                                      size_t n, uint8_t* fq32_out);
int d377_batch_encode_to_curve_wide_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes, size_t len,
                                        size_t n, uint8_t* enc32_out);
int d377_batch_to_affine_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* xyzt, size_t n, uint64_t* xy);
int d377_batch_scalar_mul_var_element_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt,
                                          const uint8_t* scalar32, size_t n, uint64_t* out_xyzt);
int d377_batch_scalar_mul_base_element_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* scalar32, size_t n,
                                           uint64_t* out_xyzt);
int d377_batch_compress_to_field_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* p_xyzt, size_t n,
                                     uint64_t* fq_out);
int d377_batch_encode_to_curve_element_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* fq32, size_t n,
                                           uint64_t* out_xyzt);
int d377_batch_hash_to_curve_element_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* r1_32,
                                         const uint8_t* r2_32, size_t n, uint64_t* out_xyzt);
int d377_batch_fq_from_bytes_checked_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes32, size_t n,
                                         uint64_t* out, uint8_t* status);
int d377_batch_fq_to_bytes_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* a, size_t n, uint8_t* bytes32);
int d377_batch_fr_from_le_bytes_mod_order_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes32, size_t n,
                                              uint8_t* fr32_out);
int d377_batch_fr_from_bytes_checked_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes32, size_t n,
                                         uint8_t* fr32_out, uint8_t* status);
int d377_batch_fr_op_dev(d377_ctx* ctx, int dev, void* stream, int op, const uint8_t* a32, const uint8_t* b32, size_t n,
                         uint8_t* out32, uint8_t* status);
int d377_batch_fr_from_wide_bytes_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* bytes, size_t len,
                                      size_t n, uint8_t* fr32_out);
/* MSM on device buffers (the workspace grows inside the context on first use of a larger n). */
int d377_msm_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* xyzt, const uint8_t* scalar32, size_t n,
                 uint8_t* enc32_out, uint64_t* xyzt_out);
int d377_msm_encoded_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* enc32, const uint8_t* scalar32,
                         size_t n, uint8_t* enc32_out, uint64_t* xyzt_out, uint8_t* status);
int d377_batch_msm_small_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* xyzt, const uint8_t* scalar32, size_t m,
                             size_t n, uint8_t* enc32_out, uint64_t* xyzt_out);
int d377_batch_msm_small_encoded_dev(d377_ctx* ctx, int dev, void* stream, const uint8_t* enc32, const uint8_t* scalar32,
                                     size_t m, size_t n, uint8_t* enc32_out, uint64_t* xyzt_out, uint8_t* status);
/* Sum of m Element records (e.g. the per-rank partial sums of a sharded MSM after an all-gather). */
int d377_sum_elements_dev(d377_ctx* ctx, int dev, void* stream, const uint64_t* xyzt, size_t m,
                          uint8_t* enc32_out, uint64_t* xyzt_out);


/* Multi-GPU for a batch that already lives in HBM (contiguous slices, scatter of inputs and
 * gather of outputs only, no other exchange).  in0/in1/out0/out1 are the buffers the matching
 * d377_batch_*_dev entry point takes, in its order (unused ones NULL), resident on device `root_dev` of
 * the context; slice k of the batch runs on device k (peer copies over xGMI for k != root_dev, in place
 * for the root), and `stream` (a stream of the root device) continues once every slice is back. */
#define D377_OP_SQRT_RATIO_ZETA 0
#define D377_OP_DECOMPRESS 1
#define D377_OP_COMPRESS 2
#define D377_OP_ROUNDTRIP 3
#define D377_OP_SCALAR_MUL_BASE 4
#define D377_OP_SCALAR_MUL_VAR 5
#define D377_OP_ENCODE_TO_CURVE 6
#define D377_OP_HASH_TO_CURVE 7
#define D377_OP_SCALAR_MUL_VAR_ELEMENT 8
#define D377_OP_SCALAR_MUL_BASE_ELEMENT 9
int d377_batch_sharded_dev(d377_ctx* ctx, int root_dev, void* stream, int op, const void* in0, const void* in1, size_t n,
                           void* out0, void* out1);

#ifdef __cplusplus
}
#endif
#endif /* DECAF377_AMD_H */
