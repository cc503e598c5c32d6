/*
 * b3wit.h — C-ABI of the MI355X-native batched witness generator for the reference's BLAKE3
 * circom circuits (libb3wit.so).  Plain C: pointers and sizes only, no HIP / torch types.
 *
 * Every entry point names the piece of the reference interface it stands in for
 * (paths relative to the reference repo banyancomputer/hot-proofs-blake3-circom):
 *   WC  = blake3_nova_js/witness_calculator.js  (the circom-emitted loader, the drop-in boundary)
 *   WASM exports = the functions WC calls on the compiled circuit instance (same file, cited lines)
 *
 * Status codes: 0 ok; 1..6 are the circom runtime exception codes WC maps to text at
 * WC:21-37 (1 Signal not found, 2 Too many signals set, 3 Signal already set, 4 Assert Failed,
 * 5 Not enough memory, 6 Input signal array access exceeds the size); >= 100 are runtime errors
 * of this library.  No function throws; a ctx / batch is not thread-safe, distinct ones are.
 */
#ifndef B3WIT_H
#define B3WIT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Circuits = the committed circom builds of the reference. */
#define B3W_CIRCUIT_COMPRESSION_BN254 0 /* build/blake3_compression/blake3_compression_js/blake3_compression.wasm  N=24093 */
#define B3W_CIRCUIT_NOVA_BN254        1 /* build/blake3_nova_js/blake3_nova.wasm (--prime bn128, O2)               N=23291 */
#define B3W_CIRCUIT_NOVA_VESTA        2 /* build/blake3_nova_pasta_js/blake3_nova_pasta.wasm (--prime vesta, O2)   N=23291 */
#define B3W_CIRCUIT_NOVA_BN254_O1     3 /* build/blake3_nova/blake3_nova_js/blake3_nova.wasm (circomkit build)      N=24614 */
#define B3W_CIRCUIT_UNKNOWN         (-1)

#define B3W_OK                     0
#define B3W_E_SIGNAL_NOT_FOUND     1
#define B3W_E_TOO_MANY_SIGNALS     2
#define B3W_E_SIGNAL_ALREADY_SET   3
#define B3W_E_ASSERT_FAILED        4
#define B3W_E_NOT_ENOUGH_MEMORY    5
#define B3W_E_ARRAY_ACCESS         6
#define B3W_E_BAD_ARGUMENT       100
#define B3W_E_NO_DEVICE          101 /* no HIP device / HIP runtime error: the product has no CPU path */
#define B3W_E_HIP                102
#define B3W_E_DOMAIN             103 /* batch status only: record outside the batch kernels' domain (DESIGN.md "Input domain");
                                        b3w_calc_witness evaluates such inputs with the exact kernel instead */
#define B3W_E_NOT_ALL_INPUTS     104 /* WC:166-168 "Not all inputs have been set" */
#define B3W_E_RCCL               105 /* librccl missing or a collective failed — RCCL's, the host transport's or the caller's
                                        (b3w_last_error has the text) */

typedef struct b3w_ctx b3w_ctx;
typedef struct b3w_batch b3w_batch;

/* Library / ABI version (major<<16 | minor). */
uint32_t b3w_abi_version(void);

/* Which committed circuit is this WASM?  Replaces WebAssembly.compile(code) at WC:7: the
 * builder receives the .wasm bytes; the native path keys on their sha256. */
int32_t b3w_identify_wasm(const uint8_t *code, size_t len);

/* Replaces WebAssembly.instantiate + `new WitnessCalculator(instance, sanityCheck)` (WC:19,78,
 * ctor WC:109-125).  device = HIP device ordinal (>= 0).  Fails with B3W_E_NO_DEVICE when the
 * HIP runtime has no such device — there is no CPU fallback. */
int32_t b3w_create(int32_t circuit, int32_t device, b3w_ctx **out);
void b3w_destroy(b3w_ctx *ctx);

/* WASM exports getFieldNumLen32 / getRawPrime / getWitnessSize / getInputSize /
 * getVersion+getMinorVersion+getPatchVersion (read by the ctor WC:112-122 and WC:166). */
int32_t b3w_info(const b3w_ctx *ctx, uint32_t *n32, uint8_t prime_le[32], uint32_t *witness_size,
                 uint32_t *input_size, uint32_t version[3]);

/* WASM export getInputSignalSize(hMSB,hLSB) (WC:141): number of values the input signal whose
 * FNV-1a-64 name hash (WC:325-337) is `fnv1a64_of_name` takes; 0 when the circuit has no such input. */
int32_t b3w_input_signal_size(const b3w_ctx *ctx, uint64_t fnv1a64_of_name);

/* One witness = WASM exports init + setInputSignal x inputs + getWitness x N as driven by
 * _doCalculateWitness (WC:131-169) and calculateBinWitness (WC:190-205).
 *   name_hashes[k], counts[k] : FNV-1a-64 of the k-th input name and how many values it carries
 *   values_le32               : sum(counts) field elements, 32-byte little-endian, already
 *                               reduced into [0,p) (WC:319-323 normalize), in key order
 *   out_body                  : witness_size*32 bytes, canonical little-endian elements
 * Any field elements are accepted: canonical records run through the batch kernel, everything else
 * through the exact (256-bit field arithmetic) device kernel — both on the GPU.
 * Size errors mirror WC:142-150 (B3W_E_TOO_MANY_SIGNALS / B3W_E_ARRAY_ACCESS /
 * B3W_E_NOT_ALL_INPUTS); a failed circuit assert returns B3W_E_ASSERT_FAILED and b3w_last_error gives
 * the reference WASM's own trace text ("Assert Failed.\nError in template Bits34_1 line: 201\n...").
 * ORDER (WC:136-160): keys are taken as given; per key the size check, then its values; the circuit runs when the
 * last missing input has been set, before the keys BEHIND the completing one are looked at.  So a failed assert
 * wins over the fault of a later key, the fault of an earlier key over the assert; when a later key is refused
 * (an unknown name with values: B3W_E_TOO_MANY_SIGNALS) the circuit HAS run and out_body holds the witness — the
 * JS shim and the Python mirror call with the keys up to the completing one, log what the circuit logs, and look
 * at the rest themselves (tests/golden/order.json, tests/test_order_parity.py). */
int32_t b3w_calc_witness(b3w_ctx *ctx, const uint64_t *name_hashes, const uint32_t *counts,
                         const uint8_t *values_le32, uint32_t nkeys, uint8_t *out_body);

/* The 76-byte .wtns v2 preamble calculateWTNSBin builds at WC:215-262 ("wtns", version 2,
 * 2 sections, section 1 {n8, prime, nWitness}, section 2 id + length). */
int32_t b3w_write_wtns_header(const b3w_ctx *ctx, uint8_t out[76]);

/* Text of the last error on this ctx (the trace WC appends after "Assert Failed.\n", WC:41,38). */
int32_t b3w_last_error(const b3w_ctx *ctx, char *buf, size_t len);

/* ---- batch fast path (no counterpart in the reference: it computes one witness per call) ----
 * Inputs are packed u32 records, all words canonical (< 2^32):
 *   compression (28 words): h[8] m[16] t[2] b d
 *   nova        (32 words): n_blocks block_count h[8] chunk_idx_low chunk_idx_high leaf_depth
 *                           total_depth depth m[16] b
 * Witness bodies stay in HBM: n bodies of witness_size*32 bytes, body i at out + i*pitch.   */

/* Kernel launch only; every pointer is a DEVICE pointer, `stream` is a hipStream_t (NULL = default
 * stream).  No allocation, no synchronisation: safe inside stream capture.
 *   d_records : n * input_size u32                     d_bodies : n * pitch bytes (pitch % 32 == 0,
 *   d_public  : n * public_words u32 or NULL             pitch >= witness_size*32)
 *   d_status  : n int32 or NULL (0 ok, 4 assert failed, 103 outside the fast-path domain) */
int32_t b3w_batch_run_device(b3w_ctx *ctx, const uint32_t *d_records, uint32_t n, uint8_t *d_bodies,
                             uint64_t pitch, uint32_t *d_public, int32_t *d_status, void *stream);

/* Number of u32 public-output words per witness written to d_public: compression 16 (out[16]),
 * nova 15 (w[1..15]: n_blocks_out block_count_out h_out[8] total_depth_out depth_out
 * chunk_idx_low_out chunk_idx_high_out leaf_depth_out). */
uint32_t b3w_public_words(const b3w_ctx *ctx);

/* Convenience wrappers that own device buffers (hipMalloc) for up to `capacity` witnesses. */
int32_t b3w_batch_alloc(b3w_ctx *ctx, uint32_t capacity, uint64_t pitch /* 0 = witness_size*32 */, b3w_batch **out);
void b3w_batch_free(b3w_batch *batch);
/* H2D of n records + kernel on `stream`, then stream sync. */
int32_t b3w_batch_run(b3w_batch *batch, const uint32_t *host_records, uint32_t n, void *stream);
/* D2H of the public outputs (n * public_words u32) and per-witness status (n int32, may be NULL). */
int32_t b3w_batch_outputs(b3w_batch *batch, uint32_t *host_public, int32_t *host_status);
/* D2H of one full body (witness_size*32 bytes). */
int32_t b3w_batch_fetch(b3w_batch *batch, uint32_t index, uint8_t *out_body);
/* Zero-copy handle for on-GPU consumers: device pointer of body 0, and the pitch. */
void *b3w_batch_device_ptr(b3w_batch *batch, uint64_t *pitch);

/* ---- body-buffer placement (no counterpart in the reference) ------------------------------------
 * MI355X HBM consists of three classes of physical memory (DESIGN.md "Placement"): the witness kernels' store
 * pattern — one stream per body — runs about 25 % faster when the bodies being written at any moment are spread
 * over two classes than when they all sit in one, which is where a plain hipMalloc puts them.  b3w_bodies_alloc
 * returns a linear device buffer of at least `bytes` bytes whose 256 MiB pieces alternate between two classes
 * (found by timing short store probes against two reference pieces while the buffer is assembled through the HIP
 * virtual-memory API);
 * *placement reports what was achieved — and "mixed" is only reported when ONE REAL witness launch of this context's
 * circuit into the buffer is at least 10 % faster than into a plain hipMalloc buffer (measured once per context; a buffer
 * that fails the check stays usable and is reported as INTERLEAVED: its pieces do alternate, but on this box, today,
 * plain buffers were as fast — no speed claim; B3W_PLACE_CHECK=0 skips the check).  Use the pointer like
 * any device pointer (kernels, hipMemcpy); release it
 * with b3w_bodies_free.  B3W_PLACEMENT=plain in the environment turns the search off.
 * b3w_batch_alloc places its body buffer this way. */
#define B3W_PLACEMENT_PLAIN 0 /* one class (no search, search failed, or a buffer below 512 MiB) */
#define B3W_PLACEMENT_MIXED 1 /* alternating classes, and measured >= 10 % faster than plain buffers */
#define B3W_PLACEMENT_INTERLEAVED 2 /* alternating classes, measured gain below 10 % (this box's plain buffers were fast themselves) */
int32_t b3w_bodies_alloc(b3w_ctx *ctx, uint64_t bytes, void **d_ptr, int32_t *placement);
int32_t b3w_bodies_free(b3w_ctx *ctx, void *d_ptr);
/* The allocator keeps up to 3 x 12 GiB of classified-but-unused physical memory per device for the next buffer (and two
 * 256 MiB reference pieces for good); b3w_bodies_trim returns that reserve to the driver. */
void b3w_bodies_trim(void);
/* Memory a CONTEXT keeps: when a b3w_chain is destroyed its ring buffers (batch_steps bodies each — 12 GB for 16 384 nova steps)
 * stay with the context for the next chain of the same ring geometry, because placed buffers use up address space for good.
 * Bounds: spares of one ring size at a time (a chain of another geometry releases them first) and at most 26 GiB in all.
 * Neither torch nor RCCL can see that memory; b3w_ctx_trim(ctx) releases it (to the placement pool — follow with
 * b3w_bodies_trim to hand it to the driver), b3w_destroy does the same. */
int32_t b3w_ctx_trim(b3w_ctx *ctx);
/* Bounds of the placement allocator, in GiB (negative = leave as is): `search_gib` = new physical memory one search may
 * touch transiently beyond the buffer itself (default and most: 160, and never more than half of what is free — released again at
 * the end of the search; what ends a search first is its time limit, b3w_bodies_search_limit); `pool_gib` = labelled memory kept pooled for later buffers, all three labels together (default 12).
 * Also B3W_PLACE_SEARCH_GIB / B3W_PLACE_POOL_GIB in the environment.  Other allocators in the process (torch, RCCL) cannot
 * see pooled memory: b3w_bodies_trim hands it back. */
void b3w_bodies_configure(int64_t search_gib, int64_t pool_gib);
/* What placement costs.  A search ends — and the buffer is plain — when it has not found a second class of memory after
 * `seconds` (default 5; B3W_PLACE_SEARCH_S; <= 0 = no limit) as well as when it runs out of its GiB budget.
 * b3w_bodies_search_stats: out[0] = seconds spent inside searching b3w_bodies_alloc calls on the ctx's device so far (probes,
 * seam checks), out[1] = GiB of new physical memory those searches created (most of it released again), out[2] = searches that
 * ended on the time limit, out[3] = the time limit, out[4] = seconds spent in the real-kernel check of "mixed" buffers (the three
 * plain yardstick buffers are measured once per context). */
void b3w_bodies_search_limit(double seconds);
int32_t b3w_bodies_search_stats(const b3w_ctx *ctx, double out[5]);
/* Where a search's time goes: seconds this process has spent inside hipMemCreate (out[0]), hipMemMap + hipMemSetAccess (out[1]), the
 * timed store probes (out[2]) and hipMemUnmap + hipMemRelease (out[3]) on the ctx's device.  The first is the driver's: a call
 * returns in microseconds until the process's footprint crosses some tens of GiB and then one call takes seconds
 * (profiles/r05/place_cost.log). */
int32_t b3w_bodies_search_breakdown(const b3w_ctx *ctx, double out[4]);
/* Harness helper: the pure-store ceiling of a body buffer — GB/s of `iters` passes (after 2 untimed ones) of kernels that do
 * nothing but the witness kernels' stores into n bodies at d_bodies + i * pitch (16-byte aligned; pitch 0 = witness_size * 32):
 * shape 0 = body streams, one wave per 4 bodies and 1 KiB per body and step (the fused kernels' EXPAND phase without trace, slot
 * table or LDS), 1 = the same per 8 bodies, 2 = the runtime's fill shape (256 workgroups over 4 KiB tiles; the sweep kernels'),
 * 3 / 4 = PACED persistent body streams (512 single-wave workgroups taking groups of 4 / 8 bodies in turn, four vector-ALU instructions
 * in front of every store: a store-only kernel without any issues faster than HBM drains and fills it SLOWER — the best store-only
 * shapes of tools/ubench/store_sweep.hip on a placed buffer), 5 = 8 bodies per wave, paced, one wave per group, 6 / 7 = the fill-ordered
 * witness kernel's store order (variant 200: one contiguous 4 MiB window chip-wide) paced by s_sleep / by vector-ALU instructions (the best of five / six
 * paces each: the rate is a cliff in the pace) — the only store-only shapes that fill a caller's plain buffer as fast as a placed one (7.0 TB/s).
 * What a witness kernel's achieved bandwidth on the SAME buffer is to be read against (bench.py: roofline.of_measured_ceiling).
 * HIP events on `stream`; waits for them.  The buffer's contents are overwritten. */
int32_t b3w_bodies_store_rate(b3w_ctx *ctx, void *d_bodies, uint32_t n, uint64_t pitch, int32_t shape, uint32_t iters, void *stream,
                              double *gb_per_s);
/* out[0] address-space arena of the ctx's device in bytes, out[1] of it used up (never reused), out[2] pooled bytes,
 * out[3] bytes of live placed buffers, out[4] their number, out[5] physical 256 MiB handles created so far. */
int32_t b3w_bodies_stats(const b3w_ctx *ctx, uint64_t out[6]);
/* Placement of a batch's own body buffer. */
int32_t b3w_batch_placement(const b3w_batch *batch);

/* Timing helper for harnesses: records HIP events around `iters` back-to-back launches of the
 * batch kernel on `stream` and returns the average kernel time in milliseconds (device pointers
 * as in b3w_batch_run_device). */
int32_t b3w_batch_time_device(b3w_ctx *ctx, const uint32_t *d_records, uint32_t n, uint8_t *d_bodies,
                              uint64_t pitch, uint32_t *d_public, int32_t *d_status, void *stream,
                              uint32_t iters, float *avg_ms);

/* On-device consumer #1: the rank-1 constraint check  <A_k, z> * <B_k, z> - <C_k, z> = 0  for every constraint k of an
 * R1CS and every witness body z in HBM — what the reference's consumers do with a witness first: circom_tester's
 * expectPass / checkConstraints (test/blake3_hash.test.ts:36) and synthesize_with_vec, which enforces every R1CS row over
 * the witness variables (rust_fold/src/utils.rs:17-88, rows read from the circuit's .r1cs by circom-scotia).
 * The constraint system is the caller's: `r1cs_image` is a complete iden3 .r1cs file image (format version 1: header,
 * constraints, wire map) over the ctx's field with nWires = witness_size — the circuit's own .r1cs where the maintainer
 * has it (the reference checkout does not: .MISSING_LARGE_BLOBS), or one this repository derives (tools/gen_r1cs.py ->
 * hot-proofs-blake3-circom_amd/constraints/: blake3_compression.r1cs.gz and blake3_nova_bn254_o1.r1cs.gz from the circuit
 * text with circom's signal numbering rule; blake3_nova_bn254.r1cs.gz and blake3_nova_vesta.r1cs.gz — the O2 builds — by
 * aligning the reference's O2 witnesses with its O1 witnesses and eliminating the missing wires; gunzip first).
 * The check is independent of the witness kernels: full field arithmetic on the 32-byte elements as they lie in the
 * body, no knowledge of the circuit beyond the file.
 *   d_violations[i] = number of constraints body i violates (0 = a valid witness); an element >= p counts as a violation
 *                     of every constraint that reads it;   d_first[i] (may be NULL) = lowest violated constraint index
 *                     (file order), 0xFFFFFFFF if none. */
typedef struct b3w_r1cs b3w_r1cs;
int32_t b3w_r1cs_create(b3w_ctx *ctx, const uint8_t *r1cs_image, size_t len, b3w_r1cs **out);
int32_t b3w_r1cs_info(const b3w_r1cs *r1cs, uint32_t *n_constraints, uint32_t *n_wires, uint64_t *n_terms,
                      uint32_t *n_pub_out, uint32_t *n_pub_in, uint32_t *n_prv_in);
/* Which formulation the checks of this system take: 1 = the tile kernels (rows local enough: at most 1 024 wires outside
 * any tile of 1 024 consecutive wires — every circom circuit seen so far), 0 = the gather kernel (any system). */
int32_t b3w_r1cs_is_tiled(const b3w_r1cs *r1cs);
void b3w_r1cs_destroy(b3w_r1cs *r1cs);
/* d_bodies 16-byte aligned, pitch a multiple of 16 (0 = witness_size * 32).
 * Stream capture: b3w_batch_run_device, b3w_batch_verify_device and b3w_r1cs_check_device only enqueue kernels on `stream`
 * (no allocation, no synchronisation, no memset nodes), so a caller may capture a loop of small batches into a hipGraph and
 * replay it (tests/test_gpu_graph_capture.py); b3w_batch_commit_device too once its scratch has grown to the batch size, and
 * b3w_r1cs_check_device after its first call on that stream (which allocates the stream's deferred-row scratch, a fixed
 * 27 MB for these systems): check once before capturing — a first check on a stream that is being captured is refused
 * (B3W_E_BAD_ARGUMENT, b3w_last_error says why) instead of allocating inside the capture.  A system keeps the scratch of at most
 * eight streams: when a ninth comes, the least recently used one is released once its last check has finished (streams seen
 * under capture keep theirs until b3w_r1cs_destroy).  One check at a time per stream. */
int32_t b3w_r1cs_check_device(b3w_ctx *ctx, const b3w_r1cs *r1cs, const uint8_t *d_bodies, uint32_t n, uint64_t pitch,
                              uint32_t *d_violations, uint32_t *d_first, void *stream);
/* The same on the witnesses of the last b3w_batch_run; host arrays of n entries (host_first may be NULL). */
int32_t b3w_batch_r1cs_check(b3w_batch *batch, const b3w_r1cs *r1cs, uint32_t *host_violations, uint32_t *host_first);
/* A ready-made consumer for the chained pass (b3w_chain_run_leaves / run_parents): checks every batch of step witnesses
 * against the step circuit's constraints while it sits in the ring — synthesize_with_vec's job (rust_fold/src/utils.rs:17-88)
 * for every step of the fold.  `user` = a b3w_r1cs_sink; d_violations has room for every step of the pass (step order).
 * `next` / `next_user` (may be NULL) name a second consumer that runs on the same batch afterwards, e.g. b3w_commit_consumer
 * with its b3w_commit_sink: witness -> constraint check -> commitment, nothing but counts and points kept. */
typedef struct {
  b3w_ctx *ctx;
  const b3w_r1cs *r1cs;
  uint32_t *d_violations;
  void (*next)(void *user, const uint8_t *d_bodies, uint64_t pitch, uint64_t first_step, uint32_t count, void *stream);
  void *next_user;
  int32_t error; /* first non-zero status of a check launch, 0 = none */
} b3w_r1cs_sink;
void b3w_r1cs_consumer(void *user, const uint8_t *d_bodies, uint64_t pitch, uint64_t first_step, uint32_t count, void *stream);

/* A cheaper tamper check of n witness bodies in HBM (NOT independent evidence: it shares the witness kernels' trace code
 * and slot table).  The circuits are deterministic (the inputs fix every
 * signal), so a body satisfies all constraints iff it equals the witness recomputed from its own input
 * slots; the kernel reads each body once (HBM-read bound) and compares every 16-byte unit.
 * d_mismatch[i] = number of differing units of body i: 0 = valid witness; 0xFFFFFFFF = the body's inputs are
 * rejected by the circuit or are not plain 32-bit values (outside this path's domain).  Stands where the
 * reference's consumers check a witness against the R1CS (circom_tester expectPass, test/blake3_hash.test.ts:36;
 * synthesize_with_vec's constraints, rust_fold/src/utils.rs:17-88). */
int32_t b3w_batch_verify_device(b3w_ctx *ctx, const uint8_t *d_bodies, uint32_t n, uint64_t pitch, uint32_t *d_mismatch,
                                void *stream);

/* On-device consumer #2: Pedersen commitments C_i = sum_s w_i[s] * G[s - first_slot] over slots s >= first_slot of n
 * witness bodies in HBM — what the folding prover does with a step witness right after `synthesize`
 * (rust_fold/src/main.rs:166-179 -> arecibo's prove_step commits to W).  The group is the one
 * whose scalar field is the circuit's field: BN254 G1 for the bn128 circuits, the Pallas curve for the --prime vesta
 * build (short Weierstrass, a = 0).  The generators are the caller's (arecibo's commitment key): affine points,
 * x then y, 32-byte little-endian each, standard (non-Montgomery) form, one per committed slot.
 * b3w_commit_key_create precomputes 2^k * G for the slots that hold more than one bit, so committing a witness is
 * "add the points of its set bits" (no doublings).  Output: n affine points (x, y as above; all zero = the point
 * at infinity); d_status (may be NULL): 0 ok, 103 = a bit slot of that body does not hold 0 or 1 (not a body of the
 * batch kernels: commitment not meaningful). */
#define B3W_CURVE_BN254_G1 0 /* y^2 = x^3 + 3 over q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47 */
#define B3W_CURVE_PALLAS   1 /* y^2 = x^3 + 5 over 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001: the Pallas curve
                               * (pasta_curves; generator (-1, 2)), whose SCALAR field is the prime circom calls "vesta" — the group
                               * arecibo's PallasEngine commits in for the --prime vesta build (rust_fold/src/main.rs:366) */
#define B3W_CURVE_VESTA    B3W_CURVE_PALLAS /* older name of the same id, after the circuit's prime; kept for source compatibility */
typedef struct b3w_commit_key b3w_commit_key;
int32_t b3w_commit_key_create(b3w_ctx *ctx, int32_t curve, uint32_t first_slot, const uint8_t *host_generators, b3w_commit_key **out);
/* The same with the table's window width chosen by the caller: every `window_bits` consecutive bits of the witness share one
 * table of 2^window_bits - 1 precomputed subset sums.  12: 1.2-1.3 GB of HBM, 0.03 s set-up;  16: 14-15 GB, 0.3 s, a
 * quarter fewer point additions per witness (+15-18 % throughput);  18 (r05): 51 GB for an unfolded nova key, 25 for a folded
 * one, 1 s, another ninth fewer additions (+7.6 %) — HBM spent for VALU work on a 288 GB card.  0 = automatic
 * (b3w_commit_key_create): the environment's B3W_COMMIT_WINDOW, else the widest of 18, 16, 12 whose table takes at most a
 * quarter of the free device memory. */
int32_t b3w_commit_key_create_ex(b3w_ctx *ctx, int32_t curve, uint32_t first_slot, const uint8_t *host_generators, uint32_t window_bits,
                                 b3w_commit_key **out);
/* FOLDED keys.  A witness satisfies its circuit's LINEAR constraints, so a slot that is a linear combination of others — every
 * 32-bit word of these circuits is the sum of its bit slots — need not be committed by itself: with w_k = sum_j a_kj w_j,
 * w_k G_k = sum_j w_j (a_kj G_k), i.e. the caller adds a_kj G_k to the generator of slot j once and marks slot k as folded
 * (folded[k - first_slot] != 0: no table windows, the slot's bytes are not even read for their value).  The commitment of every
 * body that satisfies those relations is the same point; half of the point additions are gone (46 289 -> 23 377 virtual slots
 * for blake3_compression).  folded[k] = 0x80 | i keeps ONE virtual slot of a 32-bit word: bit i of its value, with the generator
 * given for it (the O2 builds: a word is 31 bit slots + 2^i * the bit circom's O2 pass took out of the witness).
 * A body that violates the relations (not a witness) gets the commitment of the witness its remaining slots
 * determine: run the constraint check where that matters.  The Python binding derives the relations from an .r1cs image and folds
 * the generators (fold.py: CommitKey(..., fold=image)).  b3w_slot_widths: bits a slot can hold (1, 32, 64, 256) as the
 * commitment kernel cuts it into virtual slots. */
int32_t b3w_commit_key_create_folded(b3w_ctx *ctx, int32_t curve, uint32_t first_slot, const uint8_t *host_generators,
                                     const uint8_t *folded /* witness_size - first_slot flags, or NULL */, uint32_t window_bits,
                                     b3w_commit_key **out);
int32_t b3w_slot_widths(b3w_ctx *ctx, uint16_t *out_bits /* witness_size */);
uint32_t b3w_commit_key_window(const b3w_commit_key *key);     /* 12, 16 or 18 */
/* Statistics for harnesses: while counting is on (on != 0 resets and starts, 0 stops; both wait for the device), every commit
 * launch with this key adds its number of mixed point additions (one per non-zero window and per tabulated inverse: 8 field
 * multiplications + 2 squarings each — the work the kernel's VALU roofline is priced in) to a device counter.
 * b3w_commit_key_counts: out[0] = additions, out[1] = witnesses committed, since counting was switched on (waits for the device). */
int32_t b3w_commit_key_count(b3w_commit_key *key, int32_t on);
int32_t b3w_commit_key_counts(const b3w_commit_key *key, uint64_t out[2]);
void b3w_commit_key_destroy(b3w_commit_key *key);
/* d_bodies and d_points 16-byte aligned, pitch a multiple of 16 (0 = witness_size * 32). */
int32_t b3w_batch_commit_device(b3w_ctx *ctx, const b3w_commit_key *key, const uint8_t *d_bodies, uint32_t n, uint64_t pitch,
                                uint8_t *d_points /* n * 64 bytes */, int32_t *d_status, void *stream);
/* The same for the witnesses of the last b3w_batch_run; host_points receives n * 64 bytes, host_status n int32 (may be NULL). */
int32_t b3w_batch_commit(b3w_batch *batch, const b3w_commit_key *key, uint8_t *host_points, int32_t *host_status);

/* Commitments straight from the input records, without the witness bodies: a witness is an expansion of a 3.7-11 KB
 * trace image through the slot table, so its bits — all the commitment needs — are pieces of image words.  Equal to
 * b3w_batch_run_device followed by b3w_batch_commit_device, at the speed of the point additions alone (the 771 KB body
 * is neither written nor read).  d_public (may be NULL) and d_status (required) receive what b3w_batch_run_device
 * writes; a record whose status is not 0 gets the point at infinity (all zero).
 * O2 nova circuits: the 67 IsZero inverses of a step (256 virtual bit slots each, 39 % of a folded key) are 1/k of small signed
 * k the record determines, so the key also holds, per gadget, the points (+-1/k) G for |k| <= 2 047 (18 MB, 10 ms of set-up)
 * and this path adds ONE point per gadget instead of sixteen windows; a larger |k| goes through the windows as before.
 * b3w_batch_commit_device does the same with bodies: k from the body's input slots, the point taken only when the body's inverse
 * slot holds exactly +-1/k — a body that says anything else is committed to as it stands.
 * Same points either way (tests/test_gpu_commit.py); B3W_COMMIT_INVTAB=0 builds keys without the tables. */
int32_t b3w_commit_records_device(b3w_ctx *ctx, const b3w_commit_key *key, const uint32_t *d_records, uint32_t n, uint8_t *d_points,
                                  uint32_t *d_public, int32_t *d_status, void *stream);

/* The same from and to host buffers (n records of 28 / 32 words; n * 64 bytes of points; public outputs and status may be NULL). */
int32_t b3w_commit_records(b3w_ctx *ctx, const b3w_commit_key *key, const uint32_t *host_records, uint32_t n, uint8_t *host_points,
                           uint32_t *host_public, int32_t *host_status);

/* A ready-made consumer for the chained pass (b3w_chain_run_leaves / run_parents below): commits every batch of step
 * witnesses while it sits in the ring, so that of a 28 TB pass only one 64-byte point per step is kept.
 * `user` = a b3w_commit_sink whose d_points has room for every step of the pass (n_leaf + n_parent points, step order);
 * d_status may be NULL. */
typedef struct {
  b3w_ctx *ctx;
  const b3w_commit_key *key;
  uint8_t *d_points;
  int32_t *d_status;
  int32_t error; /* first non-zero status of a commit launch, 0 = none */
} b3w_commit_sink;
void b3w_commit_consumer(void *user, const uint8_t *d_bodies, uint64_t pitch, uint64_t first_step, uint32_t count, void *stream);

/* Same check on the witnesses of the last b3w_batch_run; host_mismatch receives n counts. */
int32_t b3w_batch_verify(b3w_batch *batch, uint32_t *host_mismatch);

/* Streaming .wtns writer (the hand-off the reference does one file at a time: generate_witness.js:15-18,
 * circomkit `witness` in test/witness_gen.test.ts:41, then `snarkjs groth16 prove` reads the file, :47-49): witnesses
 * [first, first+count) of the last b3w_batch_run are copied to the host through two pinned staging buffers (D2H of chunk
 * k+1 overlaps the file writes of chunk k) and written as <dir>/<prefix><index>.wtns, each byte-identical to
 * calculateWTNSBin's image (76-byte header + body, one writev per file).  The files are written by `threads` writer threads
 * (0 = B3W_WTNS_THREADS, else min(16, cores)).  Witnesses whose status is not 0 are skipped.  Returns the number of files
 * written in *written.  Rates: profiles/r04/wtns_writer.log (PCIe D2H alone: about 56 GB/s = 72 k witnesses/s). */
int32_t b3w_batch_write_wtns(b3w_batch *batch, uint32_t first, uint32_t count, const char *dir, const char *prefix,
                             uint32_t *written);
int32_t b3w_batch_write_wtns_ex(b3w_batch *batch, uint32_t first, uint32_t count, const char *dir, const char *prefix,
                                uint32_t threads, uint32_t *written);

/* Choose the fastest bit-identical kernel variant for THIS output buffer: the body-stream kernels (variants 0 / 3 / 8: fastest
 * on a buffer from b3w_bodies_alloc), the fill-ordered fused kernel (200; compression circuit and nova O2 builds: fastest on a
 * caller's own plain hipMalloc / torch buffer — 7.0-7.2 against 5.5 TB/s on one-class memory — and the default there even without
 * this call), the same paced lighter (201: fastest of all where its pace holds, which only a measurement on
 * the buffer says) and the two-kernel sweep path (100), DESIGN.md "Witness kernels".  An integrator that brings its own buffers calls this once per (context, buffer).  Runs and times each candidate on the caller's device
 * buffers, which end up holding the correct witnesses, and keeps the winner in the ctx for later
 * b3w_batch_run_device calls of more than 2 560 witnesses.  Batches up to 2 560 witnesses follow the default policy — SLICED
 * (several waves per body, DESIGN.md "Batch size") or, from 128 witnesses / 512 nova steps on, the fill order — unless B3W_VARIANT says otherwise; for such an n the call only times that
 * shape and reports it (*chosen_variant = 20 + waves per body).  Allocates the sweep scratch on first use; not for stream
 * capture. */
int32_t b3w_batch_autotune_device(b3w_ctx *ctx, const uint32_t *d_records, uint32_t n, uint8_t *d_bodies,
                                  uint64_t pitch, uint32_t *d_public, int32_t *d_status, void *stream,
                                  int32_t *chosen_variant, float *chosen_ms);

/* ---- chained ("nova fold") mode: step-input planner -----------------------------------------
 * Device counterpart of the reference's per-step input construction: Blake3BlockCompressCircuit::
 * {new, update_for_step, format_input} (rust_fold/src/blake3_circuit.rs:160-289), Blake3CompressPubIO::new
 * (:83-110) and the sibling-CV extraction of hash_with_path (rust_fold/src/blake3_hash.rs:17-93).
 * Because the chaining value is plain BLAKE3, the records of ALL steps of ALL chunks are produced
 * by a native pre-pass and every step witness becomes independent (b3w_batch_run_device on the
 * records).  Records use the nova batch input format.  All pointers are device pointers. */
uint64_t b3w_chain_num_chunks(uint64_t preimage_len);       /* ceil(len / 1024), at least 1 */
uint64_t b3w_chain_num_leaf_steps(uint64_t preimage_len);   /* total number of 64-byte blocks */
uint32_t b3w_chain_path_len(uint64_t chunk, uint64_t n_chunks); /* parents above `chunk` in BLAKE3's tree */

/* Leaf steps of chunks [first_chunk, first_chunk + n_chunks_local): 16 records per chunk (fewer for the
 * last, partial chunk), record j of chunk c at d_records + ((c - first_chunk)*16 + j)*32 words; the chunk
 * chaining values (8 words each) go to d_chunk_cvs.  d_preimage points at byte first_chunk*1024. */
int32_t b3w_chain_plan_leaves_device(b3w_ctx *ctx, const uint8_t *d_preimage, uint64_t preimage_len, uint64_t first_chunk,
                                     uint32_t n_chunks_local, uint32_t *d_records, uint32_t *d_chunk_cvs, void *stream);

/* BLAKE3 tree over n_chunks chunk CVs.  d_levels: (2*n_chunks + 64) * 8 words; the caller fills level 0
 * (words [0, 8*n_chunks)) with the chunk CVs, level j+1 follows level j.  d_root receives the root
 * chaining value = the BLAKE3 hash words (for n_chunks == 1 the chunk CV already is the root). */
int32_t b3w_chain_tree_device(b3w_ctx *ctx, uint32_t *d_levels, uint64_t n_chunks, uint32_t *d_root, void *stream);

/* Parent steps of the paths of chunks [first_chunk, +n_chunks_local), for ANY chunk count, planned the way the
 * reference's driver plans them (hash_with_path, rust_fold/src/blake3_hash.rs:58-84: the PathNode at height g carries the
 * node's right child CV when bit g of the chunk index is clear, else its left child CV; format_input, blake3_circuit.rs:
 * 230-245: that CV as m[0..7], zeros above, b = 64; h and depth come from the previous step).  Chunk c gets
 * b3w_chain_path_len(c, n) records, bottom up (record g: height g, depth = path_len - 1 - g), starting at row
 * b3w_chain_parent_row(c, n) - b3w_chain_parent_row(first_chunk, n) of d_records; in a complete tree that is row
 * (c - first_chunk) * log2(n).  The step circuit takes left/right from the same index bits (Blake3GetDownLeftPath,
 * circuits/blake3_nova.circom:47-84), which is the leaf's real position only when b3w_chain_path_provable(c, n) = 1
 * (every leaf of a complete tree; in an incomplete tree the leaves of the leading power-of-two subtree and those later ones
 * whose position happens to agree): exactly those paths end in h_out = BLAKE3(preimage).  For the other leaves the records are
 * still the reference's, and so is the (wrong) final value — tests/golden/incomplete_trees.nova_vesta.json holds the
 * reference WASM's transcript.  d_levels must have been through b3w_chain_tree_device. */
uint64_t b3w_chain_num_parent_steps(uint64_t preimage_len, uint64_t first_chunk, uint64_t n_chunks_local);
uint64_t b3w_chain_parent_row(uint64_t chunk, uint64_t n_chunks);       /* chunk == n_chunks: the total */
int32_t b3w_chain_path_provable(uint64_t chunk, uint64_t n_chunks);
int32_t b3w_chain_plan_parents_device(b3w_ctx *ctx, const uint32_t *d_levels, uint64_t n_chunks, uint64_t preimage_len,
                                      uint64_t first_chunk, uint32_t n_chunks_local, uint32_t *d_records, void *stream);

/* ---- multi-GPU exchange: RCCL over xGMI, for hosts that do not bring their own collectives ------------
 * One process per GPU.  Witness bodies never leave the GPU that produced them; what the fold needs
 * from every rank are the per-step public outputs (h_out ...: 15 or 16 words per step) and, in chained mode, the
 * chunk chaining values (8 words per chunk).  b3w_comm_allgather is ncclAllGather on the caller's stream.
 * librccl is loaded at run time on first use (an RCCL already loaded into the process is reused).
 *   rank 0:  b3w_comm_unique_id(id)  -> hand the 128 bytes to the other ranks (file, socket, environment)
 *   all:     b3w_comm_create(ctx, id, rank, nranks, &comm)
 *            b3w_comm_allgather(comm, d_local, d_all, bytes_per_rank, stream)     d_all = nranks * bytes_per_rank
 * The Python harness uses torch.distributed ("nccl" = RCCL) for the same exchange instead. */
typedef struct b3w_comm b3w_comm;
#define B3W_COMM_ID_BYTES 128
int32_t b3w_comm_unique_id(uint8_t id[B3W_COMM_ID_BYTES]);
int32_t b3w_comm_create(b3w_ctx *ctx, const uint8_t id[B3W_COMM_ID_BYTES], int32_t rank, int32_t nranks, b3w_comm **out);
/* Two more transports behind the same b3w_comm, so that everything below that takes one — b3w_batch_allgather_public,
 * b3w_chain_run_parents_sharded, b3w_chain_allgather_hout[_host] — also runs where RCCL cannot: several ranks on ONE GPU
 * (rehearsals of an N-GPU job, tests of the rank > 0 paths), hosts without librccl, or a caller that already has collectives.
 *   b3w_comm_create_host      the processes of one host, through a POSIX shared-memory segment: every rank passes the same
 *                             `name` ("/something-unique-to-the-job"; rank 0 creates the segment, the others wait for it, the
 *                             name is removed again once all are attached).  An all-gather is D2H into pinned memory, the
 *                             segment, H2D — it WAITS for `stream` (not for stream capture).  Every wait gives up after
 *                             B3W_HOSTCOMM_TIMEOUT_S seconds (default 120) with B3W_E_RCCL on all ranks.
 *   b3w_comm_create_external  the caller's collective: `allgather(user, d_send, d_recv, bytes_per_rank, stream)` must leave
 *                             d_recv[r * bytes_per_rank ..) = rank r's d_send[0 .. bytes_per_rank) for every r, ordered on `stream`
 *                             like a kernel launched there (it may also simply wait for the stream), and return 0; any other
 *                             value fails the calling entry point with B3W_E_RCCL.  Device pointers; called on the thread
 *                             that called into the library, with ctx's device current. */
typedef int32_t (*b3w_allgather_fn)(void *user, const void *d_send, void *d_recv, uint64_t bytes_per_rank, void *stream);
int32_t b3w_comm_create_host(b3w_ctx *ctx, const char *name, int32_t rank, int32_t nranks, b3w_comm **out);
int32_t b3w_comm_create_external(b3w_ctx *ctx, int32_t rank, int32_t nranks, b3w_allgather_fn allgather, void *user, b3w_comm **out);
int32_t b3w_comm_rank(const b3w_comm *comm);
int32_t b3w_comm_size(const b3w_comm *comm);
void b3w_comm_destroy(b3w_comm *comm);
int32_t b3w_comm_allgather(b3w_comm *comm, const void *d_send, void *d_recv, uint64_t bytes_per_rank, void *stream);
/* The public outputs of the last b3w_batch_run of every rank (all ranks ran the same number n of witnesses):
 * host_all receives nranks * n * public_words u32 in rank order.  Device-side gather + one D2H. */
int32_t b3w_batch_allgather_public(b3w_batch *batch, b3w_comm *comm, uint32_t *host_all);

/* ---- chained mode: the whole pass, natively ---------------------------------------------------
 * What rust_fold/src/main.rs:41-203 does one step at a time for one chunk path, for ALL steps of the chunk range
 * [first_chunk, first_chunk + n_chunks_local) of a preimage (one rank's share; a single GPU takes all chunks):
 *   b3w_chain_run_leaves   pageable or pinned host slices of the preimage -> HBM on a copy stream, overlapped with
 *                          the leaf planner and the nova witness kernels of earlier slices on `stream`
 *   (multi-GPU: all-gather the chunk chaining values of b3w_chain_local_cvs across ranks — 32 B per chunk)
 *   b3w_chain_run_parents  BLAKE3 tree over all chunk CVs, parent-step records of the local chunks' paths (any chunk
 *                          count: b3w_chain_plan_parents_device), their witnesses
 * Witness bodies go through a ring of `ring` placed buffers (b3w_bodies_alloc) of `batch_steps` bodies each: a
 * 1 GiB preimage is 28 TB of witness.  After each batch `consumer` (may be NULL) is called with the device pointer
 * of the batch; it must enqueue its work on `stream` — the buffer is overwritten `ring` batches later.
 * Step records, public outputs (15 words per step) and status live in device arrays owned by the object: leaf
 * steps first (chunk order, block order), then the parent steps (chunk order, height order).  Nothing is
 * synchronised: results are valid once `stream` has drained. */
typedef struct b3w_chain b3w_chain;
typedef void (*b3w_batch_consumer)(void *user, const uint8_t *d_bodies, uint64_t pitch, uint64_t first_step, uint32_t count,
                                   void *stream);
int32_t b3w_chain_create(b3w_ctx *ctx, uint64_t preimage_len, uint64_t first_chunk, uint32_t n_chunks_local,
                         uint32_t batch_steps, uint32_t ring, int32_t with_parents, b3w_chain **out);
void b3w_chain_destroy(b3w_chain *chain);
/* Commitments only: from now on the pass computes one commitment per step straight from the step records
 * (b3w_commit_records_device) into d_points (n_leaf + n_parent points of 64 bytes, step order) and writes no witness
 * bodies; the consumer arguments of the run calls are ignored.  key = NULL switches back to bodies. */
int32_t b3w_chain_commit_only(b3w_chain *chain, const b3w_commit_key *key, uint8_t *d_points);
/* The same commitments (from the step records, at the speed of the point additions) WHILE the bodies are written as usual —
 * constraint check and consumers see every batch: the fold-shaped pass "witness -> check -> commit" where the commitment does not
 * read the 745 KB body back (b3w_batch_commit_device / b3w_commit_consumer do: they are for bodies the library did not make).
 * A step's commitment from its record equals the commitment of the body the witness kernel writes for it (tests/test_gpu_commit.py).
 * The commit kernels are bound by the vector ALUs, the witness kernels by HBM writes, so the commitments get a stream of the
 * chain's own beside the witness kernels; b3w_chain_commit_overlap says how far that goes. */
int32_t b3w_chain_commit_from_records(b3w_chain *chain, const b3w_commit_key *key, uint8_t *d_points);
/* Where the commitments of b3w_chain_commit_from_records run, per batch of steps (call before or after it; it holds for the chain):
 *   SERIAL  on the caller's stream, in front of the batch's witness kernel: nothing runs side by side
 *   FREE    on the chain's commit stream, ordered only behind the batch's planner: they run beside the witness kernels of this and
 *           later batches and `stream` waits for them at the end of each run call (4.5 M steps/s against 3.9 serial, 64 MiB)
 *   GATED   on the commit stream beside THIS batch's witness kernel only: whatever reads the batch on `stream` afterwards — the
 *           constraint check of b3w_chain_check_constraints, the consumer — starts when both have finished and has the device to
 *           itself, and the next batch's commitments start behind it (the check wants every vector register and most of the LDS of
 *           a CU: beside it the commit kernel only gets in its way)
 *   AUTO    (default) GATED when the chain checks constraints or the run call has a consumer, else FREE */
#define B3W_COMMIT_OVERLAP_AUTO   (-1)
#define B3W_COMMIT_OVERLAP_SERIAL 0
#define B3W_COMMIT_OVERLAP_FREE   1
#define B3W_COMMIT_OVERLAP_GATED  2
int32_t b3w_chain_commit_overlap(b3w_chain *chain, int32_t mode);
/* d_points = NULL above: the chain keeps the points itself; this copies them (n_leaf + n_parent times 64 bytes) to the host. */
int32_t b3w_chain_commitments(b3w_chain *chain, uint8_t *host_points, void *stream);
/* Constraint check inside the chained pass: after this call (r1cs = a system of the chain's context; NULL turns it off) every
 * batch of step witnesses is checked against the step circuit while it sits in the ring, before the consumer sees it;
 * b3w_chain_violations copies the per-step counts (n_leaf + n_parent entries, step order; 0 = the step satisfies every
 * constraint) to the host once `stream` has drained. */
int32_t b3w_chain_check_constraints(b3w_chain *chain, const b3w_r1cs *r1cs);
int32_t b3w_chain_violations(b3w_chain *chain, uint32_t *host_violations, void *stream);
uint32_t *b3w_chain_violations_device(b3w_chain *chain);   /* device: n_leaf + n_parent counts, NULL before b3w_chain_check_constraints */
int32_t b3w_chain_run_leaves(b3w_chain *chain, const uint8_t *host_preimage /* byte 0 of the WHOLE preimage */,
                             b3w_batch_consumer consumer, void *user, void *stream);
int32_t b3w_chain_run_parents(b3w_chain *chain, const uint32_t *d_all_chunk_cvs /* n_chunks*8 words; NULL = the local ones,
                              single rank */, b3w_batch_consumer consumer, void *user, void *stream);
/* Contiguous, balanced chunk ranges (the first n_chunks % nranks ranks take one extra chunk): what rank `rank` passes
 * to b3w_chain_create as first_chunk / n_chunks_local. */
void b3w_chain_shard(uint64_t n_chunks, int32_t rank, int32_t nranks, uint64_t *first_chunk, uint32_t *n_chunks_local);
/* b3w_chain_run_parents for a sharded pass: all-gathers the chunk chaining values over `comm` (32 B per chunk; shards
 * padded to the largest — equal shards are gathered where they lie, without a copy) and continues with the tree and this
 * rank's parent steps.  The chain must have been created with this rank's b3w_chain_shard range, and b3w_chain_run_leaves
 * of the same pass must have been called.  The exchange, the tree and the parent plan run on a stream of the CHAIN's —
 * beside the leaf witness kernels still queued on `stream`, which only joins for the parent witnesses — so an external
 * transport's callback (b3w_comm_create_external) is handed that stream, not the caller's. */
int32_t b3w_chain_run_parents_sharded(b3w_chain *chain, b3w_comm *comm, b3w_batch_consumer consumer, void *user, void *stream);
/* The fold's exchange in chained mode (BASELINE config 4: "RCCL gather of h_out"): the folding driver consumes z_{i+1} = the
 * public outputs of step i (Blake3CompressPubIO::to_vec, rust_fold/src/blake3_circuit.rs:111-123, fed back at
 * rust_fold/src/main.rs:166-179), of which h_out — public words 2..9 — is the running chaining value.  One ncclAllGather over
 * `comm` of this rank's h_out rows (packed to 8 words a row, shards padded to the largest) and a scatter into
 *   d_leaf_hout   : b3w_chain_num_leaf_steps(len) * 8 u32, GLOBAL step order (chunk, block): row 16 c + blocks(c) - 1 is chunk
 *                   c's chaining value
 *   d_parent_hout : b3w_chain_parent_row(n_chunks, n_chunks) * 8 u32, (chunk, height) order: the last row of a provable chunk
 *                   path is BLAKE3(preimage)
 * on every rank (either pointer may be NULL).  Enqueued on `stream` behind the pass; exchange buffers are allocated on the
 * first sharded call of a chain and kept, so a pass that has run once neither allocates nor synchronises. */
int32_t b3w_chain_allgather_hout(b3w_chain *chain, b3w_comm *comm, uint32_t *d_leaf_hout, uint32_t *d_parent_hout, void *stream);
/* The same into host arrays (for bindings that hold no device memory: Node).  Waits for `stream`. */
int32_t b3w_chain_allgather_hout_host(b3w_chain *chain, b3w_comm *comm, uint32_t *host_leaf_hout, uint32_t *host_parent_hout, void *stream);
/* How long the two exchanges of the last sharded pass took on this rank's device, in milliseconds (HIP events on the stream each ran on):
 * out_ms[0] = chunk chaining values (staging + all-gather + compaction, b3w_chain_run_parents_sharded), out_ms[1] = h_out (pack +
 * all-gather + scatter, b3w_chain_allgather_hout); 0 for an exchange that has not run.  Waits for those events. */
int32_t b3w_chain_exchange_ms(b3w_chain *chain, float out_ms[2]);
int32_t b3w_chain_info(const b3w_chain *chain, uint64_t *n_leaf_steps, uint64_t *n_parent_steps, uint64_t *n_chunks,
                       uint32_t *path_len, int32_t *placement);
/* Waits for `stream`, then copies the results to the host: (n_leaf + n_parent) * 15 public-output words, as many
 * status words, the 8 root words (after run_parents).  Any pointer may be NULL. */
int32_t b3w_chain_outputs(b3w_chain *chain, uint32_t *host_public, int32_t *host_status, uint32_t *host_root, void *stream);
uint32_t *b3w_chain_records(b3w_chain *chain);     /* device: (n_leaf + n_parent) * 32 u32 */
uint32_t *b3w_chain_public(b3w_chain *chain);      /* device: (n_leaf + n_parent) * 15 u32 */
int32_t *b3w_chain_status(b3w_chain *chain);       /* device: (n_leaf + n_parent) int32 */
uint32_t *b3w_chain_local_cvs(b3w_chain *chain);   /* device: n_chunks_local * 8 u32 */
uint32_t *b3w_chain_root(b3w_chain *chain);        /* device: 8 u32 = BLAKE3(preimage) words, after run_parents */

#ifdef __cplusplus
}
#endif
#endif /* B3WIT_H */
