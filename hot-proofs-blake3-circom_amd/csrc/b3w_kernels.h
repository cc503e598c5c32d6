// b3w_kernels.h — host-visible launch entry of b3w_kernels.hip
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

// variants >= B3W_VARIANT_SWEEP use the two-kernel path (TRACE -> HBM scratch -> linear SWEEP of the output)
#define B3W_VARIANT_SWEEP 100
// variants B3W_VARIANT_SLICED + s (s = 2 .. 64): one body per wave and s waves per body, each storing 1/s of its tiles (small batches)
#define B3W_VARIANT_SLICED 20
// variant B3W_VARIANT_REGIONFILL: the fill-ordered fused kernel for caller-owned (plain) buffers — 256 workgroups of four storing waves and
// a tracer wave over absolute 128 KiB regions, one contiguous 4 MiB window chip-wide (b3w_kernels.hip "REGIONFILL"); compression and nova O2 circuits
#define B3W_VARIANT_REGIONFILL 200
// the same kernel paced lighter: on the edge of the cliff below which the chip-wide store window frays — the fastest where it
// holds (7.17-7.19 TB/s at 4 096 witnesses, up to 7.38 from 16 384 on), 6.7-7.2 where it does not: a choice for the autotuner, which times it on the buffer
// at hand, not for the default
#define B3W_VARIANT_REGIONFILL_LIGHT 201
#define B3W_REGIONFILL_GRID 256  // one workgroup per CU, workgroup i on XCD i % 8
// d_aux of the nova circuits: [0,8) prime, [8] TABLE_N, [16 + 8k, +8) k^-1 mod p for k < 2048, then (O2) the slot numbers of the 67 IsZero inverses,
// then — for the four offsets a body may have in a 128-byte line — how many slots lie in lines that hold an inverse, and those slots as (slot, table entry)
#define B3W_AUX_WIDE_SLOTS (16 + 8 * 2048)
#define B3W_AUX_LINE_COUNTS (B3W_AUX_WIDE_SLOTS + 68)
#define B3W_AUX_LINE_LISTS (B3W_AUX_LINE_COUNTS + 4)      // uint2[4][B3W_LINE_LIST_MAX]
#define B3W_LINE_LIST_MAX 160
// then the aliases of the nova image words from 1 024 on (from | to << 16): see b3w_create, the fill-ordered kernel's 16-bit table
#define B3W_AUX_ALIAS_COUNT (B3W_AUX_LINE_LISTS + 2 * 4 * B3W_LINE_LIST_MAX)
#define B3W_AUX_ALIAS_LIST (B3W_AUX_ALIAS_COUNT + 1)
#define B3W_ALIAS_MAX 192
#define B3W_SWEEP_GRID 256       // one 256-thread workgroup per CU, tile = 4 KiB: the runtime fill kernel's shape
#define B3W_SWEEP_LOGC 13
#define B3W_SWEEP_CHUNK (1u << B3W_SWEEP_LOGC)   // witnesses per TRACE+SWEEP pair = row length of the scratch

extern "C" int b3w_launch_batch(int kind, int variant, const uint32_t *d_recs, uint32_t n, uint8_t *d_out,
                                uint64_t pitch, const uint32_t *d_table, uint32_t nwit, uint32_t *d_pub,
                                int32_t *d_status, const void *d_aux, uint32_t *d_scratch, uint32_t scratch_cap,
                                hipStream_t stream);

// b3w_exact.hip: one witness with arbitrary field-element inputs (full circom field semantics)
extern "C" int b3w_launch_exact(int nova, const uint32_t *d_inputs, const uint32_t *d_prime, const uint32_t *d_table,
                                uint32_t nwit, uint8_t *d_out, uint32_t *d_status, hipStream_t stream);

// b3w_plan.hip: chained-mode step-input planner
extern "C" uint32_t b3w_plan_path_len(uint64_t chunk, uint64_t nchunks);
extern "C" int b3w_launch_plan_leaves(const uint8_t *d_pre, uint64_t total_len, uint64_t first_chunk, uint32_t nlocal,
                                      uint64_t nchunks, uint32_t *d_recs, uint32_t *d_chunk_cv, hipStream_t stream);
extern "C" int b3w_launch_plan_merge(const uint32_t *d_left, const uint32_t *d_right, uint32_t stride_words, uint64_t npairs,
                                     uint32_t root, uint32_t *d_parents, hipStream_t stream);
extern "C" int b3w_launch_plan_parents(const uint32_t *d_levels, uint64_t nchunks, uint64_t first_chunk, uint32_t nlocal,
                                       uint32_t last_chunk_blocks, uint32_t *d_recs, hipStream_t stream);
extern "C" uint32_t b3w_plan_tree_first_level(uint64_t nchunks);               // the level the one-launch tree kernel starts at
extern "C" int b3w_launch_plan_tree(uint32_t *d_levels, uint64_t nchunks, uint32_t l0, uint32_t *d_root, uint64_t first_chunk, uint32_t plan_nlocal,
                                    uint32_t last_chunk_blocks, uint32_t *d_recs, hipStream_t stream);
extern "C" uint64_t b3w_plan_parent_row(uint64_t chunk, uint64_t nchunks);     // chunk == nchunks: all parent steps
extern "C" int b3w_plan_path_provable(uint64_t chunk, uint64_t nchunks);
// the fold's exchange (h_out = public words 2 .. 9 of a step): strided rows -> wire format, gathered rank blocks -> global step order
extern "C" int b3w_launch_pack_hout(const uint32_t *d_pub, uint64_t row0, uint64_t count, uint32_t *d_dst, hipStream_t stream);
extern "C" int b3w_launch_unpack_hout(const uint32_t *d_gathered, uint64_t block_words, uint64_t par_off, const uint64_t *d_tab, uint32_t nranks,
                                      uint64_t max_rows, uint32_t *d_leaf_all, uint32_t *d_par_all, hipStream_t stream);

extern "C" int b3w_launch_trace(int kind, const uint32_t *d_recs, uint32_t cn, uint32_t *d_images, uint32_t row, const uint32_t *d_table,
                                uint32_t nwit, uint32_t *d_pub, int32_t *d_status, const void *d_aux, hipStream_t stream);
extern "C" int b3w_launch_verify(int kind, const uint32_t *d_in_slots, uint32_t n, const uint8_t *d_bodies, uint64_t pitch,
                                 const uint32_t *d_table, uint32_t nwit, uint32_t *d_mismatch, const void *d_aux,
                                 hipStream_t stream);

// b3w_placement.hip: body buffers assembled from two classes of HBM (HIP virtual-memory API)
extern "C" int b3w_place_alloc(int device, uint64_t bytes, int want_mixed, void **out, int *mixed, float *rates);
extern "C" int b3w_place_free(void *ptr);
extern "C" int b3w_place_is_mixed(const void *ptr);                            // 1: inside a placed buffer of alternating classes
extern "C" void b3w_place_trim(void);
extern "C" void b3w_place_configure(int64_t search_gib, int64_t pool_gib);      // < 0: leave as is
extern "C" void b3w_place_stats(int device, uint64_t out[6]);
extern "C" void b3w_place_search_limit(double seconds);                         // <= 0: none
extern "C" void b3w_place_search_stats(int device, double out[4]);             // seconds, GiB walked, time-outs, the limit
extern "C" void b3w_place_cost_breakdown(int device, double out[4]);           // seconds in hipMemCreate, map + access, probes, unmap + release
extern "C" int b3w_place_store_rate(uint8_t *buf, uint64_t pitch, uint32_t n, uint32_t body_bytes, int shape, uint32_t iters, hipStream_t stream, double *gbs);

// b3w_commit.hip: Pedersen commitments of witness bodies (on-device consumer).
// Window width W (virtual slots per window, 2^W - 1 tabulated subset sums each) is a property of the key:
//   12: 1.2 GB table, 0.03 s set-up;   16: 14 GB table, 0.3 s set-up, a quarter fewer additions per witness (+15-18 %);   18: 51 GB, 1 s, +7.6 % more
#define B3W_COMMIT_WINDOW_SMALL 12
#define B3W_COMMIT_WINDOW_LARGE 16
#define B3W_COMMIT_WINDOW_XL 18          // (r05) 4 x the table of 16 (a folded nova key: 25 GB, an unfolded one 51) for a ninth fewer additions: +7.6 % —
                                         // HBM for VALU work, the card has 288 GB.  (20 bits, 16 x the table: SLOWER than 16, 7.01 against 7.17 M steps/s —
                                         // the gathers over 90-180 GB become the limit; not built.  profiles/r05/commit_windows_rates.log)
#define B3W_COMMIT_WINDOW_OK(w) ((w) == B3W_COMMIT_WINDOW_SMALL || (w) == B3W_COMMIT_WINDOW_LARGE || (w) == B3W_COMMIT_WINDOW_XL)
#define B3W_COMMIT_ENTRIES(W) ((1u << (W)) - 1u)
#define B3W_COMMIT_SUM_WORDS 36        // per witness between the commit and the normalise kernel: X, Y, ZZ, ZZZ in nine 29-bit limbs each
// Field of the curve's coordinates:
struct B3wCurve {
  uint32_t p[8];      // modulus, little-endian limbs
  uint32_t r2[8];     // 2^512 mod p   (into Montgomery form)
  uint32_t one[8];    // 2^256 mod p   (1 in Montgomery form)
  uint32_t pm2[8];    // p - 2         (Fermat inversion exponent)
  uint32_t inv;       // -p^-1 mod 2^32
};
extern "C" int b3w_launch_commit_setup(const uint32_t *d_gens, const uint32_t *d_first_v, const uint32_t *d_nbits, uint32_t nslots,
                                       uint32_t *d_points, const B3wCurve *curve, hipStream_t stream);
extern "C" int b3w_launch_commit_windows(const uint32_t *d_points, uint32_t nwin, uint32_t window, uint32_t *d_table, const B3wCurve *curve, hipStream_t stream);
extern "C" int b3w_launch_commit(const uint8_t *d_bodies, uint32_t n, uint64_t pitch, uint32_t first_slot, uint32_t nslots,
                                 const uint32_t *d_slotdesc /* first virtual slot | width code (0 bit, 1 32, 2 64, 3 256, 4 folded away, 8 + b bit b of a word) << 24;
                                                               IsZero gadget | 5 << 24: a nova inverse slot with tabulated points */,
                                 const uint32_t *d_images /* null: read the bodies; else TRACE images (b3w_launch_trace), bodies unused */,
                                 uint32_t img_row, const uint32_t *d_runs /* pairs: v0 | (len - 1) << 24, image word | shift << 16 */, uint32_t nruns,
                                 const uint32_t *d_table, uint32_t nwin, uint32_t window, uint32_t *d_sums, uint8_t *d_out,
                                 int32_t *d_status, const uint32_t *d_invtab /* or null: the O2 nova circuits' inverse point tables */, uint32_t inv_nk,
                                 const uint32_t *d_invmeta /* bodies mode with code-5 slots: [0, 67) committed slot of gadget j, [67, 71) witness slots of
                                                              n_blocks, block_count, total_depth, depth, [71, 138) first virtual slot of gadget j's slot */,
                                 const uint32_t *d_aux /* ... and the context's scalars: [0, 8) the prime, [16 + 8 k, + 8) 1 / k */,
                                 unsigned long long *d_adds /* or null: += the mixed additions of this launch (statistics) */,
                                 const B3wCurve *curve, hipStream_t stream);
// (d_out = null: the sums stay sums — X, Y, ZZ, ZZZ in d_sums, B3W_COMMIT_SUM_WORDS per witness — and are made points later, many at
// once: a normalisation is one field inversion, 380 DEPENDENT multiplications, and the launch that does a batch's worth of them right
// behind its commit kernel is one wave per CU busy for 130 us with the machine idle around it — 6 % of a commit-only pass.
// b3w_launch_commit_normalize: four witnesses a thread share one inversion (Montgomery's trick), any number of witnesses at once.)
extern "C" int b3w_launch_commit_normalize(const uint32_t *d_sums, uint64_t n, uint8_t *d_out, const B3wCurve *curve, hipStream_t stream);
// O2 nova circuits, records mode: invtab[j * nk + mag - 1] = (1 / mag) * G of the slot holding IsZero gadget j's inverse
// (d_inverses: 8 words per magnitude, standard form — the witness kernels' table; d_inv_slot[j] = committed slot index or ~0)
#define B3W_NOVA_ISZERO 67
extern "C" int b3w_launch_commit_invtab(const uint32_t *d_gens, const uint32_t *d_inv_slot, const uint32_t *d_inverses, uint32_t njobs, uint32_t nk,
                                        uint32_t *d_invtab, const B3wCurve *curve, hipStream_t stream);

// b3w_r1cs.hip: rank-1 constraint check of witness bodies (on-device consumer #1: A z * B z - C z = 0 for every row)
#include "b3w_r1cs_defs.h"          // B3wField, B3W_R1CS_TILE, B3W_R1CS_NOT_SMALL (shared with the HIP-free host code)
// rows: m x {first term, terms in A, in B, in C} (terms of a row are stored A then B then C); row_id: the row's index in the
// .r1cs file (rows are sorted by shape); term k = wires[k], cids[k] (0: coefficient +1, 1: -1, else d_coefR[16 * cid ..] =
// the coefficient (8 words), then coefficient * 2^256 mod p (8 words))
extern "C" int b3w_launch_r1cs(const uint8_t *d_bodies, uint32_t n, uint64_t pitch, uint32_t m, const uint32_t *d_rows,
                               const uint32_t *d_row_id, const uint32_t *d_wires, const uint16_t *d_cids, const uint32_t *d_coefR,
                               const B3wField *field, uint32_t *d_violations, uint32_t *d_first, hipStream_t stream);
// tiles: tile t = wires [t * B3W_R1CS_TILE, +B3W_R1CS_TILE); ntiles x {first row, rows, first outside wire, outside wires}
// coef_small[cid] = the coefficient as a signed integer when |c| < 2^62 (c or c - p), else B3W_R1CS_NOT_SMALL
// the tiled formulations (stream, walk): the same tiles with 8-byte elements in LDS and integer arithmetic only; the rows a workgroup
// cannot decide that way (an element of 2^63 or more, a coefficient that is no small integer, |<A,z>| or |<B,z>| of 2^63 or more) are
// marked in `scratch` and evaluated by a second launch with the gather kernel's field arithmetic.  Same verdicts as the gather
// kernel.  row_k[r] = the gather formulation's row number of tile row r.
struct B3wR1csSystem {
  uint32_t nwires, ntiles, max_ext, max_tile_terms, max_tile_rows, ncoef;
  const uint32_t *tiles, *tile_terms, *ext, *rows, *row_id, *row_k, *terms, *coefs;
  const long long *coef_small;
  const uint32_t *g_rows, *g_wires;
  const uint16_t *g_cids;
  // the stream kernel's program (b3w_r1cs_host.h): row descriptors by class, per tile {first general word, words, rows, -}, the
  // general rows' words and meta words, per coefficient the bound an element must stay below
  uint32_t max_g_words, max_g_rows, smask_groups;
  const uint32_t *srows, *sgdesc, *sgwords, *sgmeta;
  const unsigned long long *smask;                 // per tile x smask_groups: elements the tile's rows take for bits
  const unsigned long long *scost;                 // ntiles + 1 prefix sums of the tiles' relative unit costs
};
// the WALK kernel's program (b3w_r1cs_walk.hip, b3w_r1cs_host.h): device pointers
struct B3wWalk {
  uint32_t ntiles, exp_slots, max_gen, max_ent, ncoef, static_words, max_rows, signed_elems;   // signed_elems: the kernel instantiation that takes p - k for -k
  const uint32_t *tile;                  // B3W_WT_WORDS per tile
  const unsigned long long *mask;        // 16 per tile
  const uint16_t *exp;
  const uint4 *runs;
  const uint32_t *run_row, *ent_w, *ent_m, *row_id;
  const unsigned long long *stat;        // static_words per tile
  const long long *coef_small;
  const uint32_t *static_k, *static_id;  // the always-deferred rows as a list: {first pair, pairs, linear, has C} per row, then their unique terms
                                         // as pairs {wire, coefficient id | parts << 16} (b3w_r1cs_api.cpp); constraint numbers
  uint32_t nstatic, pad2;
  uint32_t static_d0[4];                 // the first one's descriptor (a kernel argument of the deferred kernel)
  uint32_t p[8];                         // the field's prime: the walk kernel takes an element p - k (k < 2^62) for the small number -k
};
// The WALK kernel (default where the system fits): a workgroup walks whole bodies tile after tile, earlier tiles' wires come from an
// export area in LDS — no outside wire is gathered from HBM.  `sysw` = the system with the WALK row order in tiles / row_k / row_id
// (the deferred kernel's view).  Returns -6 when the program does not fit (LDS).
extern "C" int b3w_launch_r1cs_walk(const uint8_t *d_bodies, uint32_t n, uint64_t pitch, const B3wWalk *walk, const B3wR1csSystem *sysw, const B3wField *field,
                                    unsigned long long *d_scratch, uint32_t *d_violations, uint32_t *d_first, hipStream_t stream);
extern "C" size_t b3w_r1cs_walk_scratch_bytes(const B3wWalk *walk);
#define B3W_R1CS_SLAB 8192u                                  // bodies per launch pair at most: bounds the scratch (64 MB at most)
extern "C" size_t b3w_r1cs_scratch_bytes(const B3wR1csSystem *sys);
// the STREAM kernel (default where the system fits): persistent 512-thread workgroups, two to a CU, over the tile-major unit list;
// elements fetched into registers one unit ahead, one barrier per unit (b3w_r1cs.hip); same scratch blocks (word 0 = which mask
// words were stored), same deferred kernel, same verdicts.
// Returns -6 when the system does not fit (more than 2 048 rows or 480 outside wires per tile, or no room in LDS).
extern "C" int b3w_launch_r1cs_stream(const uint8_t *d_bodies, uint32_t n, uint64_t pitch, const B3wR1csSystem *sys, const B3wField *field,
                                      unsigned long long *d_scratch, uint32_t *d_violations, uint32_t *d_first, hipStream_t stream);
