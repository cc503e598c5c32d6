// b3w_r1cs.hip — on-device rank-1 constraint check of witness bodies: for every constraint (A, B, C) of an R1CS and every
// body z in HBM, is <A,z> * <B,z> - <C,z> = 0 in the circuit's field?
//
// This is what the reference's consumers do with a witness before anything else: circom_tester's expectPass /
// checkConstraints (test/blake3_hash.test.ts:36) and the Nova driver's synthesize_with_vec, which enforces
// every R1CS row over the witness variables (rust_fold/src/utils.rs:17-88).  The constraint system is DATA (an iden3
// .r1cs image parsed in b3w_r1cs_api.cpp: for blake3_compression the one tools/gen_r1cs.py derives from the circuit text
// and checks against the build's .sym and the committed witness); nothing in here knows BLAKE3, the slot tables or the
// TRACE code of the witness kernels, so a wrong witness kernel or a wrong slot table cannot hide behind it.
//
// Arithmetic: full 256-bit field elements, eight 32-bit limbs, Montgomery multiplication (CIOS) with the modulus passed
// at run time (BN254 scalar field or the Vesta base field).  A term coef * z[wire] is montmul(coef * R, z) = coef * z;
// coefficients +1 / -1 (93 % of the terms of the compression system) are an addition / subtraction.  The tile formulations put
// an exact integer fast path in front of that (small coefficients times small elements summed in 128 bits).
//
// FOUR FORMULATIONS, same verdicts (tests/test_gpu_r1cs.py runs all of them over clean, corrupted and random inputs):
//  * the WALK kernel (round 4; default where the system fits: b3w_r1cs_walk.hip): a persistent workgroup walks whole
//    bodies tile after tile, the wires a tile's rows need from earlier tiles come out of an export area in LDS — HBM traffic = the
//    bodies, nothing else; truth-table rows in runs of 32 decided by one lane each.
//  * the STREAM kernel (B3W_R1CS_GATHER=4; round 3's default): persistent workgroups over the
//    tile-major list of (tile, body) units, elements fetched into registers one unit ahead, one barrier per unit, the host's
//    stream program instead of row-by-row evaluation; what it cannot decide goes to the same deferred kernel as the lean pair's.
//  * the LEAN pair (B3W_R1CS_GATHER=3; round 2's default): circom constraints are local, so a workgroup takes (body, tile
//    of 1 024 consecutive wires), streams the tile from HBM once plus the few wires outside the tile its rows mention (<= 137
//    per tile for the derived systems, listed per tile by the host: +6 % reads, L2 hits), keeps 8 bytes per element in LDS
//    together with bit-packed copies, the tile's term list and the small coefficients, and decides the rows as exact integers;
//    recomposition rows "word = sum 2^i bit_i" are folded into BIT RUNS.  What the integer case does not cover is marked and
//    evaluated by a second launch (b3w_r1cs_deferred_kernel) with the field arithmetic.  A body is read from HBM once.
//  * the GATHER kernel (B3W_R1CS_GATHER=1, and any system whose rows are not local enough: more than 1 024 outside wires for
//    some tile).
//
// Gather kernel — mapping: thread = (constraint, body), 256 consecutive rows per workgroup; rows are sorted by shape (terms in A, B, C)
// on the host so that the 64 lanes of a wave run the same trip counts.  The workgroups of ONE body all land on one XCD
// (the hardware deals consecutive workgroup ids round-robin over the 8 XCDs: id -> body = 8 * (id / (8 * RB)) + id % 8,
// row block = (id / 8) % RB), so a body is fetched from HBM into one L2 once and its ~4.9 reads per element (once per
// term) are L2 hits.  The constraint stream (117 760 terms, 0.9 MB for blake3_compression) is shared by all bodies and
// stays in L2; neighbouring rows read neighbouring slots (the bit runs), so most wave loads are whole lines.
// Values decide how much arithmetic a term costs, never what it computes: an element that is 0 contributes nothing, an
// element that is 1 contributes the coefficient itself, a product with a factor 0 or 1 needs no multiplication — the
// general Montgomery path is taken for everything else (any field element is handled exactly).  A valid witness of these
// circuits is 98 % bits, so the check of a valid batch is gather-bound (L2/TA), not ALU-bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include "b3w_kernels.h"
#include "b3w_r1cs_device.h"

namespace {

__global__ __launch_bounds__(256) void b3w_r1cs_kernel(const uint8_t *__restrict__ bodies, uint64_t pitch, uint32_t n, uint32_t m,
                                                       uint32_t row_blocks, const uint4 *__restrict__ rows /* off, nA, nB, nC */,
                                                       const uint32_t *__restrict__ row_id, const uint32_t *__restrict__ wires,
                                                       const uint16_t *__restrict__ cids, const uint32_t *__restrict__ coefR,
                                                       B3wField F, uint32_t *__restrict__ violations, uint32_t *__restrict__ first) {
  // all row blocks of a body on one XCD (see the header): groups of 8 bodies x row_blocks workgroups
  const uint32_t per_group = 8u * row_blocks;
  const uint32_t b = (blockIdx.x / per_group) * 8u + (blockIdx.x & 7u);
  const uint32_t r = ((blockIdx.x % per_group) >> 3) * 256u + threadIdx.x;
  if (b >= n) return;                                    // whole workgroup: the last group of a ragged batch
  const uint8_t *body = bodies + (uint64_t)b * pitch;
  bool bad = false;
  uint32_t id = 0xFFFFFFFFu;
  if (r < m) {
    bad = gather_row(body, rows[r], wires, cids, coefR, F);
    id = row_id[r];
  }
  const uint64_t mask = __ballot(bad);
  if (mask) {
    uint32_t mine = bad ? id : 0xFFFFFFFFu;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) mine = min(mine, (uint32_t)__shfl_xor((int)mine, s));
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(&violations[b], (uint32_t)__popcll(mask));
      if (first) atomicMin(&first[b], mine);
    }
  }
}

// (r02's LEAN pair — (body, tile) workgroups with 8-byte elements and integer sums in LDS, the rows they could not decide
// marked for b3w_r1cs_deferred_kernel — was the default of round 2 and a comparison formulation until round 4; the stream kernel
// below took its element packing and its deferred kernel, the walk kernel its place.  Removed in round 5: HISTORY.md.)

// ---- the STREAM kernel (default): persistent workgroups, one barrier per (body, tile) unit ------------------------------------
// The lean kernel is (body, tile) workgroups that load, pack, evaluate and leave: its HBM loads are in flight only part of a
// workgroup's life, and the sum of load time and evaluate time — not their maximum — is what a tile costs (profiles/r02: waves
// parked 53-58 % of their cycles).  Here a 512-thread workgroup (two to a CU) walks a contiguous, cost-weighted range of the
// tile-major unit list (tile, body).  Per tile switch (one or two per workgroup) it fetches what is per TILE — row descriptors,
// word list, masks, fetch offsets — so the steady-state loop has no global load but the elements themselves.
// FETCH.  No raw image of the tile in LDS (rounds of r03 had one, filled by LDS-DMA: 48 KB a workgroup, one unit in flight per
// workgroup and a barrier between fetch and pack).  A wave fetches the elements IT packs into registers, one unit ahead: a group
// of 64 elements is two CONTIGUOUS 1 KiB loads (a lane below 32 takes the low 16 bytes of its element, the lane 32 above it the
// high 16: 8 cache lines per instruction, every byte used), v_permlane32_swap then puts the low halves of the 64 elements into
// one register set and the high halves into the other.  (One element per lane — 32-byte stride, twice — is 16 half-used lines
// per instruction, and the texture path is what sixteen waves queue for: 0.75 -> 0.65 ms per 4 096 bodies.)  The outside wires
// of a tile are a gather (a lane per wire), the slowest thing here to come back: fetched a whole unit earlier.
// PIPELINE.  Elements, bit words and the general rows' sums exist twice in LDS (unit parity), so within one iteration
//     fetch tile elements of unit i + 1 | words and rows of unit i | verdicts of unit i - 1 | pack unit i + 1 | barrier
// nothing depends on anything else of the same iteration: ONE barrier per unit, and what a wave has more of than its neighbours
// (general words, outside wires, general rows' verdicts — dealt to different waves) is not waited for phase by phase.
// ROWS: the host's STREAM PROGRAM (b3w_r1cs_host.cpp).  Booleanity rows (64 % of all rows) and the bit-ness of every
// truth-table operand: the host's per-tile masks of "elements taken for bits" ANDed with the pack's "neither 0 nor 1" ballots —
// one scalar AND per 64 elements; only if that (or wire 0 not being 1) finds something does the unit take the row-by-row road
// (`anomaly`).  TRUTH-TABLE rows — a row over at most five wires that hold bits is a function of five bits, tabulated by the host
// with exact field arithmetic and looked up from the elements' bits (every XOR gate of these circuits: 87 % of the other rows).
// GENERAL rows (first in every tile: waves 0, 1 own them) one WORD per lane — every lane multiplies its coefficient and element
// and adds the product into its row's sums in LDS (ds_add_u64) — then the row's owner lane compares A * B with C.  A product that
// would not stay below 2^103, an element that is no bit in a truth-table row or a coefficient that is no small integer defers
// the row to the deferred kernel's field arithmetic; ALWAYS-DEFERRED rows have the same mask bits for every body of the tile.
// Everything a wave does here costs every other wave of its SIMD four cycles per instruction, so the steady state is written for
// instruction count: wave-uniform values live in SGPRs (the wave number comes through readfirstlane), whole classes of rows are
// decided by masks, a wave runs only the code of the row classes it owns, and a wave whose rows all hold skips the verdicts.

// A value the compiler must take as new at this point: what is derived from a row descriptor (LDS addresses of its operands, of its
// sums, the address of its id) is then computed where it is used, unit after unit, instead of once per tile into registers that
// stay occupied through the whole pipeline — 74 VGPRs of such loop invariants against 49 the loop itself works in, and the
// descriptors spilled to scratch to make room (every scratch reload waits for vmcnt(0): for every load in flight).
__device__ __forceinline__ uint4 fresh(uint4 d) {
  asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
  return d;
}

template <int WAVES, bool STAMPS>
__global__ __launch_bounds__(64 * WAVES, 4) void b3w_r1cs_stream_kernel(   // (second argument: waves per SIMD the register budget must allow — two 8-wave workgroups or one of 16)
    const uint8_t *__restrict__ bodies, uint64_t pitch, uint32_t n, B3wR1csSystem S, unsigned long long *__restrict__ scratch, uint32_t block_words,
    uint32_t *__restrict__ violations, uint32_t *__restrict__ first, uint32_t dbg_arg, unsigned long long *__restrict__ stamps) {
  constexpr uint32_t THREADS = 64u * WAVES;
  const uint32_t dbg = STAMPS ? dbg_arg : 0u;               // (the experiments' switches exist in the diagnostic instantiation only)
  constexpr int EG = 16 / WAVES;                           // groups of 64 tile elements a wave packs (1 or 2)
  constexpr int RP = 32 / WAVES;                           // row passes at most (a tile has at most 2 048 rows)
  extern __shared__ __align__(16) unsigned char smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);                  // (an SGPR: what depends on it alone is scalar code)
  // the diagnostic instantiation only (B3W_R1CS_STAMPS=1): cycles per phase, summed over the units of the middle workgroup, per wave
  // (a kernel of its own: the sums and the clock are 20 SGPRs the measured one has better uses for)
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  const bool stamping = STAMPS && stamps != nullptr && blockIdx.x == gridDim.x / 2u;
#define B3W_STAMP(k) do { if (STAMPS && stamping) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); ph[k] += t_now - t_prev; t_prev = t_now; } } while (0)
  // LDS: [parity] elements, [parity] bit words, [parity] general rows' sums {A, B, C} x {low, high} and flags; the tile's word list
  // with its meta words; the coefficients; summary and anomaly words
  const uint32_t ext_cap = (S.max_ext + 32u) & ~31u;                               // room for the outside wires (+ at least one spare element)
  const uint32_t ne = B3W_R1CS_TILE + ext_cap;
  const uint32_t groups = ((ne + 63u) >> 6) + 1u;                                  // (one spare pair: a run reads its group and the next)
  const uint32_t gr2 = (S.max_g_rows + 1u) & ~1u;
  unsigned long long *el0 = reinterpret_cast<unsigned long long *>(smem);
  unsigned long long *packed0 = el0 + 2u * ne;
  unsigned long long *gsum0 = packed0 + 4u * groups;
  uint32_t *gflag0 = reinterpret_cast<uint32_t *>(gsum0 + 12u * S.max_g_rows);
  uint32_t *gwords = gflag0 + 2u * gr2;
  uint32_t *gmeta = gwords + ((S.max_g_words + 4u) & ~3u);
  long long *lcoef = reinterpret_cast<long long *>(gmeta + ((S.max_g_words + 4u) & ~3u));
  unsigned long long *lsum = reinterpret_cast<unsigned long long *>(lcoef + ((S.ncoef + 1u) & ~1u));      // [unit parity]: bit w = the unit's mask word w is not zero (and was stored)
  uint32_t *lanom = reinterpret_cast<uint32_t *>(lsum + 2);                        // [unit mod 3] != 0: something the masks cannot vouch for in this (body, tile)
  // the row descriptors of the passes a lane does NOT keep in registers (rows PRE * THREADS and up of a tile: few tiles have any)
  constexpr int PRE = RP < 2 ? RP : 2;
  uint4 *lrow_hi = reinterpret_cast<uint4 *>(lanom + 4);
  for (uint32_t k = tid; k < S.ncoef; k += THREADS) lcoef[k] = S.coef_small[k];
  if (tid < 4) packed0[(tid >> 1) * 2u * groups + 2u * (groups - 1u) + (tid & 1u)] = 0ull;      // the spare pairs
  // who does what beside the equal shares (two element groups, every row pass): the general words go to the LAST waves (whose last
  // row pass is the short one), the groups of outside wires (at most 8) to waves 2, 3, ..., the general rows' verdicts are
  // waves 0 and 1's (the host puts those rows first), the summary word is wave 1's
  const uint32_t nxg = (ext_cap + 63u) >> 6;
  const uint32_t xg = (wave + (uint32_t)WAVES - 2u) % (uint32_t)WAVES;             // this wave's group of outside wires, if it has one
  const bool packs_ext = xg < nxg;

  // units, tile-major: u = tile * n + body; a workgroup takes a contiguous range of equal COST (S.scost: a tile with many general
  // words costs more than one of booleanity rows): position x of n * (total cost) lies in the tile t with n * scost[t] <= x, at body
  // (x - n * scost[t]) / (cost of t)
  auto unit_at = [&](const uint32_t k) -> uint64_t {       // first unit of workgroup k (k = gridDim.x: one past the last unit)
    if (k >= gridDim.x) return (uint64_t)S.ntiles * n;
    const uint64_t total = S.scost[S.ntiles] * n;
    const uint64_t x = total / gridDim.x * k + total % gridDim.x * k / gridDim.x;
    uint32_t t = 0;
    while (t + 1u < S.ntiles && S.scost[t + 1u] * n <= x) t++;
    const uint64_t body = (x - S.scost[t] * n) / (S.scost[t + 1u] - S.scost[t]);
    return (uint64_t)t * n + (body < n ? body : n - 1u);
  };
  uint64_t u = unit_at(blockIdx.x);
  const uint64_t u_end = unit_at(blockIdx.x + 1u);
  while (u < u_end) {
    const uint32_t tile = (uint32_t)(u / n), b_lo = (uint32_t)(u - (uint64_t)tile * n);
    const uint32_t m = (uint32_t)((u_end - u) < (uint64_t)(n - b_lo) ? (u_end - u) : (uint64_t)(n - b_lo));      // bodies of this tile
    // ---- per tile: descriptors, the word list, the masks, the fetch offsets.  (Nothing is in flight here: the pipeline below drains.)
    lds_barrier();                                                                 // the previous tile's last summary words are out, its word list is free
    const uint4 td = reinterpret_cast<const uint4 *>(S.tiles)[tile];               // {first row, rows, first outside wire, outside wires}
    const uint4 gd = reinterpret_cast<const uint4 *>(S.sgdesc)[tile];              // {first entry of the general rows, entries, general rows, entries that are bit runs (a multiple of 64; they stand first)}
    const uint32_t t0 = tile * B3W_R1CS_TILE;
    const uint32_t n_local = S.nwires - t0 < B3W_R1CS_TILE ? S.nwires - t0 : B3W_R1CS_TILE;
    const uint4 *srows = reinterpret_cast<const uint4 *>(S.srows);
    // this lane's rows (pass p: row p * THREADS + tid) and what the wave owns in each pass (wave-uniform: SGPRs)
    // (ONE scalar of flag bits — p: the wave has truth-table rows in pass p, 8 + p: general rows, 16 + p: always-deferred rows — not
    // a lane mask per flag and pass: the loop below has more wave-uniform state than there are SGPRs, and what does not fit is
    // kept in the lanes of a VGPR, a v_readlane per use)
    uint4 pre[PRE];
    uint32_t wflags = 0;
#pragma unroll
    for (int p = 0; p < RP; p++) {
      const uint32_t r = (uint32_t)p * THREADS + tid;
      const uint4 d = r < td.y ? srows[td.x + r] : make_uint4(0, 0, 0, 0);
      if (p < PRE) pre[p < PRE ? p : 0] = d;
      else if (r < ((S.max_tile_rows + 63u) & ~63u)) lrow_hi[r - (uint32_t)PRE * THREADS] = d;      // (whole waves: a lane behind the tile's rows reads an all-zero descriptor)
      wflags |= (__ballot((d.y >> 29) == 1u) != 0ull ? 1u : 0u) << p | (__ballot((d.y >> 28) == 1u) != 0ull ? 0x100u : 0u) << p |
                (__ballot((d.y >> 30) == 1u) != 0ull ? 0x10000u : 0u) << p;
    }
    auto row_of = [&](const int p) -> uint4 {              // this lane's row of pass p (registers, or LDS for the passes behind)
      if (p < PRE) return pre[p < PRE ? p : 0];
      return lrow_hi[(uint32_t)(p - PRE) * THREADS + tid];
    };
    wflags = __builtin_amdgcn_readfirstlane(wflags);
    const uint32_t my_rows = td.y > wave * 64u ? (td.y - wave * 64u + THREADS - 1u) / THREADS : 0u;      // row passes in which this wave has rows
#define has_rows(p) ((uint32_t)(p) < my_rows)
#define has_tt(p) ((wflags >> (p)) & 1u)
#define has_gen(p) ((wflags >> (8 + (p))) & 1u)
#define has_dmask(p) ((wflags >> (16 + (p))) & 1u)
    unsigned long long mbit[EG];                             // elements of this wave's groups the tile's rows take for bits
#pragma unroll
    for (int q = 0; q < EG; q++) mbit[q] = S.smask[(size_t)tile * S.smask_groups + wave + (uint32_t)q * WAVES];
    const unsigned long long xbit = packs_ext ? S.smask[(size_t)tile * S.smask_groups + 16u + xg] : 0ull;
    for (uint32_t k = tid; k < gd.y + 1u; k += THREADS) {    // (+ 1: a lane reads the word behind its own)
      gwords[k] = S.sgwords[gd.x + k];
      gmeta[k] = S.sgmeta[gd.x + k];
    }
    if (tid < 3) lanom[tid] = 0u;
    if (tid < 2) lsum[tid] = 0ull;
    for (uint32_t k = tid; k < 6u * gd.z; k += THREADS) { gsum0[k] = 0ull; gsum0[6u * S.max_g_rows + k] = 0ull; }      // (afterwards every general row's owner lane zeroes its own sums)
    for (uint32_t k = tid; k < gd.z; k += THREADS) { gflag0[k] = 0u; gflag0[gr2 + k] = 0u; }
    // fetch offsets: group g = elements 64 g ... 64 g + 63 as two blocks of 32; lane l of a block takes the low (l < 32) or high
    // 16 bytes of element (l & 31) of the block.  (An element behind the last tile's end: that tile's last element again.)
    uint4 rlo[EG], rhi[EG], xlo = make_uint4(0, 0, 0, 0), xhi = make_uint4(0, 0, 0, 0);
    uint32_t eoff[EG][2], xoff = 0;
#pragma unroll
    for (int q = 0; q < EG; q++) {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t e = (wave + (uint32_t)q * WAVES) * 64u + (uint32_t)h * 32u + (lane & 31u);
        eoff[q][h] = (t0 + (e < n_local ? e : n_local - 1u)) * 32u + (lane >> 5) * 16u;
      }
      rlo[q] = rhi[q] = make_uint4(0, 0, 0, 0);
    }
    const uint32_t xj = xg * 64u + lane;                     // outside wires: a lane per wire, lanes without one masked off
    const bool xact = packs_ext && xj < td.w;
    if (xact) xoff = S.ext[td.z + xj] * 32u;
    // every global load above has landed before the pipeline starts: inside it the compiler must find nothing of its own to wait for
    // but the fetches
#pragma unroll
    for (int p = 0; p < PRE; p++) asm volatile("" :: "v"(pre[p].x), "v"(pre[p].y), "v"(pre[p].z), "v"(pre[p].w));
#pragma unroll
    for (int q = 0; q < EG; q++) asm volatile("" :: "s"((uint32_t)mbit[q]), "s"((uint32_t)(mbit[q] >> 32)));
    asm volatile("" :: "s"((uint32_t)xbit), "s"((uint32_t)(xbit >> 32)), "v"(xoff));
    auto body_of = [&](const uint32_t k) -> const uint8_t * {      // body of unit k of this tile (behind the tile's end: the last one — fetched, packed, never looked at: no condition between a fetch and its use)
      const uint32_t kk = (dbg & 12u) ? 0u : k < m ? k : m - 1u;   // (experiments 4, 8: the tile's first body over and over — no HBM traffic, the same instructions)
      return bodies + (uint64_t)(b_lo + kk) * pitch;
    };
    auto fetch_ext = [&](const uint32_t k) {
      const uint8_t *body = body_of(k);
      if (xact) {
        xlo = ldg16<false>(body, xoff);
        xhi = ldg16<false>(body, xoff + 16u);
      }
    };
    auto fetch_tile = [&](const uint32_t k) {
      const uint8_t *body = body_of(k);
#pragma unroll
      for (int q = 0; q < EG; q++) {
        rlo[q] = ldg16<true>(body, eoff[q][0]);              // (until the pack: elements + 0 ... 31, both halves)
        rhi[q] = ldg16<true>(body, eoff[q][1]);              // (... + 32 ... 63)
      }
    };
    // ---- pack (unit k): 32-byte elements -> 8 bytes (bit 63 = "not below 2^63") + two bit words per 64 elements; anything the
    // masks take for a bit and that is none raises the anomaly flag, and so does wire 0 not being 1
    auto pack = [&](const uint32_t k, const uint32_t k3) {
      // every staged register is taken here, by every wave, whatever it goes on to use: the compiler waits for the fetch at this
      // one point (and knows it has) — a fetch it could not prove consumed would cost a vmcnt(0) where the NEXT fetch is issued
#pragma unroll
      for (int q = 0; q < EG; q++)
        asm volatile("" :: "v"(rlo[q].x), "v"(rlo[q].y), "v"(rlo[q].z), "v"(rlo[q].w), "v"(rhi[q].x), "v"(rhi[q].y), "v"(rhi[q].z), "v"(rhi[q].w));
      asm volatile("" :: "v"(xlo.x), "v"(xlo.y), "v"(xlo.z), "v"(xlo.w), "v"(xhi.x), "v"(xhi.y), "v"(xhi.z), "v"(xhi.w));
      if (dbg & 2u) return;
      unsigned long long *el = el0 + (k & 1u) * ne, *packed = packed0 + (k & 1u) * 2u * groups;
      bool flag = false;
#pragma unroll
      for (int q = 0; q < EG; q++) {
        const uint32_t g = wave + (uint32_t)q * WAVES, e = g * 64u + lane;         // the wave holds elements 64 g ... 64 g + 63
        uint4 lo = rlo[q], hi = rhi[q];
        halves_apart(lo.x, hi.x); halves_apart(lo.y, hi.y); halves_apart(lo.z, hi.z); halves_apart(lo.w, hi.w);
        const unsigned long long z = e < n_local ? lean_pack(lo, hi) : 0ull;
        el[e] = z;
        const unsigned long long ones = __ballot(z == 1ull), bads = __ballot(z > 1ull);
        if (lane == 0) { packed[2 * g] = ones; packed[2 * g + 1] = bads; }
        flag = flag || (bads & mbit[q]) != 0ull || (tile == 0 && g == 0 && !(ones & 1ull));          // (wire 0 is element 0 of tile 0)
      }
      if (packs_ext) {                                                             // (whole waves, for the ballots)
        const unsigned long long z = xj < td.w ? lean_pack(xlo, xhi) : 0ull;
        if (xj < ext_cap) el[B3W_R1CS_TILE + xj] = z;
        const unsigned long long ones = __ballot(z == 1ull), bads = __ballot(z > 1ull);
        if (lane == 0) { packed[2 * (16u + xg)] = ones; packed[2 * (16u + xg) + 1] = bads; }
        flag = flag || (bads & xbit) != 0ull || (tile != 0 && xg == 0 && !(ones & 1ull));            // (outside wire 0 of every other tile)
      }
      if (flag && lane == 0) lanom[k3] = 1u;
    };
    // ---- general rows, one ENTRY per lane: chunk c of 64 entries goes to wave WAVES - 1 - c mod WAVES; a lane adds coefficient *
    // element (a term) or the value of a bit run into its row's part sum.  The host puts the tile's runs first (padded to whole
    // chunks), so a chunk is of one kind and a wave runs one kind's code for it.  A part sum is two 64-bit counters {low, high}
    // worth low + high * 2^52: a contribution below 2^54 goes to `low` whole, a larger one (the dyadic row scaling of the O2 systems
    // makes 2^30 * word) is cut at bit 52 — products stay below 2^103 and a row has at most 256 entries, so no counter overflows
    // and the sum is exact.
    auto words = [&](const uint32_t k) {
      const unsigned long long *el = el0 + (k & 1u) * ne, *packed = packed0 + (k & 1u) * 2u * groups;
      unsigned long long *gsum = gsum0 + (k & 1u) * 6u * S.max_g_rows;
      uint32_t *gflag = gflag0 + (k & 1u) * gr2;
      for (uint32_t c0 = ((uint32_t)WAVES - 1u - wave) * 64u; c0 < gd.y; c0 += THREADS) {
        const uint32_t iw = c0 + lane;
        const bool act = iw < gd.y;
        const uint32_t w = gwords[act ? iw : 0u], mt = gmeta[act ? iw : 0u];
        const bool live = act && !(mt & 4u);                 // (4: a null entry of the padding)
        unsigned long long *sum = gsum + 6u * (mt >> 8) + 2u * (mt & 3u);
        if (c0 < gd.w) {                                     // a chunk of bit runs: first element | length << 16 | shift << 23 | negative << 29
          const uint32_t idx0 = w & 0xFFFFu, len = (w >> 16) & 0x7Fu, sh = (w >> 23) & 0x3Fu;
          const bool neg = (w >> 29) & 1u;
          const uint32_t g = idx0 >> 6, r = idx0 & 63u;
          const unsigned long long one_lo = packed[2 * g], bad_lo = packed[2 * g + 1], one_hi = packed[2 * g + 2], bad_hi = packed[2 * g + 3];
          const unsigned long long mask = len == 64u ? ~0ull : (1ull << len) - 1ull;
          const unsigned long long ones = ((one_lo >> r) | (r ? one_hi << (64u - r) : 0ull)) & mask;
          const unsigned long long bads = ((bad_lo >> r) | (r ? bad_hi << (64u - r) : 0ull)) & mask;
          const unsigned long long v = ones << sh;           // (below 2^62: the host checks shift + length)
          if (live) {
            if (bads) atomicOr(&gflag[mt >> 8], 1u);         // (an element of the run that is no bit)
            else if (v < (1ull << 54)) atomicAdd(sum, neg ? 0ull - v : v);
            else {
              const unsigned long long v0 = v & ((1ull << 52) - 1ull), v1 = v >> 52;
              atomicAdd(sum, neg ? 0ull - v0 : v0);
              atomicAdd(sum + 1, neg ? 0ull - v1 : v1);
            }
          }
        } else {                                             // a chunk of terms: element | coefficient id << 16
          const unsigned long long z = el[live ? w & 0xFFFFu : 0u];
          const long long c = lcoef[live ? w >> 16 : 0u];
          const unsigned long long mag = c < 0 ? 0ull - (unsigned long long)c : (unsigned long long)c;
          const unsigned long long lo = mag * z, hi = __umul64hi(mag, z);
          const bool neg = c < 0;
          const bool ok = c != B3W_R1CS_NOT_SMALL && !(z >> 63) && hi < (1ull << 39);      // (bit 63 of an element = "not below 2^63")
          if (live) {
            if (!ok) atomicOr(&gflag[mt >> 8], 1u);
            else if (hi == 0ull && lo < (1ull << 54)) atomicAdd(sum, neg ? 0ull - lo : lo);
            else {                                           // cut at bit 52: value = low + high * 2^52, low in [0, 2^52)
              __int128 v = (__int128)(((unsigned __int128)hi << 64) | lo);
              if (neg) v = -v;
              atomicAdd(sum, (unsigned long long)v & ((1ull << 52) - 1ull));
              atomicAdd(sum + 1, (unsigned long long)(long long)(v >> 52));
            }
          }
        }
      }
    };
    // ---- this wave's own rows (unit k).  Usual road: the masks have vouched for every bit, so booleanity rows hold, and a
    // truth-table row is its table indexed by the operands' low bits.  Anomaly road: row by row, with the bit-ness of every operand
    // looked at.  Returns bit p: row of pass p deferred, bit 8 + p: violated.
    auto rows = [&](const uint32_t k, const bool anomaly) -> uint32_t {
      const unsigned long long *el = el0 + (k & 1u) * ne;
      const uint32_t *el32 = reinterpret_cast<const uint32_t *>(el);
      uint32_t verdict = 0;
      auto table_row = [&](const uint4 d) {
        const uint32_t kk = (d.y >> 16) & 7u;
        const uint32_t a = (el32[2u * (d.x & 0xFFFFu)] & 1u) | (el32[2u * (d.x >> 16)] & 1u) << 1 | (el32[2u * (d.z & 0xFFFFu)] & 1u) << 2 |
                           (el32[2u * (d.z >> 16)] & 1u) << 3 | (el32[2u * (d.y & 0xFFFFu)] & 1u) << 4;      // (unused positions name element 0: masked)
        return (d.y >> 29) == 1u && !((d.w >> (a & ((1u << kk) - 1u))) & 1u);
      };
      auto careful_row = [&](const uint4 d, bool *defer, bool *bad) {
        const bool w0_is_one = el[tile == 0 ? 0 : B3W_R1CS_TILE] == 1ull;
        if (d.y >> 31) {                                     // booleanity  z * (1 - z) = 0: is the element 0 or 1
          if (w0_is_one) *bad = el[d.w] > 1ull;              // (an element of 2^63 or more is no bit)
          else *defer = true;                                // (wire 0 is not 1: nothing here means what it should — field arithmetic)
        }
        if ((d.y >> 29) == 1u) {
          const uint32_t kk = (d.y >> 16) & 7u;
          const unsigned long long z0 = el[d.x & 0xFFFFu], z1 = el[d.x >> 16], z2 = el[d.z & 0xFFFFu], z3 = el[d.z >> 16], z4 = el[d.y & 0xFFFFu];
          const unsigned long long nonbit = (z0 | (kk > 1 ? z1 : 0ull) | (kk > 2 ? z2 : 0ull) | (kk > 3 ? z3 : 0ull) | (kk > 4 ? z4 : 0ull)) >> 1;
          const uint32_t a = ((uint32_t)z0 & 1u) | ((uint32_t)z1 & 1u) << 1 | ((uint32_t)z2 & 1u) << 2 | ((uint32_t)z3 & 1u) << 3 | ((uint32_t)z4 & 1u) << 4;
          *defer = nonbit != 0ull || !w0_is_one;
          *bad = !*defer && !((d.w >> (a & ((1u << kk) - 1u))) & 1u);
        }
      };
#pragma unroll
      for (int p = 0; p < RP; p++) {
        if (!has_rows(p)) continue;
        if (!anomaly) {
          if (has_tt(p) && table_row(row_of(p))) verdict |= 0x100u << p;
        } else {
          bool defer = false, bad = false;
          careful_row(row_of(p), &defer, &bad);              // (a lane without a row holds an all-zero descriptor: no class)
          verdict |= (defer ? 1u : 0u) << p | (bad ? 0x100u : 0u) << p;
        }
      }
      return verdict;
    };
    // ---- the verdicts of unit k (one iteration after its words were added): general rows from their finished sums (the owner
    // lane zeroes them for unit k + 2), the mask words of the deferred rows, the violation counts.  What goes to the deferred kernel,
    // per (body, tile): word 0 of the block = which mask words are not zero; only those are stored.  A wave whose rows of this unit
    // are all decided and hold (every wave of almost every unit) leaves at once: one compare.
    auto verdicts = [&](const uint32_t k, const uint32_t verdict, const bool anomaly) {
      if (!(wflags & 0xFFFF00u) && !anomaly && __ballot(verdict != 0u) == 0ull) return;
      const uint32_t b = b_lo + k;
      unsigned long long *block = scratch + ((size_t)b * S.ntiles + tile) * block_words;
      unsigned long long *gsum = gsum0 + (k & 1u) * 6u * S.max_g_rows;
      uint32_t *gflag = gflag0 + (k & 1u) * gr2;
      uint32_t nbad = 0, low = 0xFFFFFFFFu, badmask = 0;
#pragma unroll
      for (int p = 0; p < RP; p++) {
        if (!has_rows(p)) continue;
        bool defer = (verdict >> p) & 1u, bad = (verdict >> (8 + p)) & 1u;
        if (has_gen(p)) {
          const uint4 d = row_of(p);
          if ((d.y >> 28) == 1u) {
            // (written out in 64-bit halves: the compiler's __int128 version of the same — range compares, a 128 x 128 product — was
            // 120 vector instructions on the one wave every other wave of the workgroup then waits for)
            unsigned long long *sum = gsum + 6u * d.x;
            unsigned long long a_lo, b_lo, c_lo;
            long long a_hi, b_hi, c_hi;
            part_sum(sum[0], sum[1], a_lo, a_hi);
            part_sum(sum[2], sum[3], b_lo, b_hi);
            part_sum(sum[4], sum[5], c_lo, c_hi);
            // A and B must fit 64 signed bits (then |A * B - C| < 2^127 < p: "= 0 mod p" is "= 0")
            defer = gflag[d.x] != 0u || (dbg & 32u) != 0u || a_hi != ((long long)a_lo >> 63) || b_hi != ((long long)b_lo >> 63);
            bad = !defer && (a_lo * b_lo != c_lo || __mul64hi((long long)a_lo, (long long)b_lo) != c_hi);
#pragma unroll
            for (int q = 0; q < 6; q++) sum[q] = 0ull;
            gflag[d.x] = 0u;
          }
        }
        if (bad) { nbad++; badmask |= 1u << p; }
        const unsigned long long mask = (has_gen(p) || anomaly ? __ballot(defer) : 0ull) | (has_dmask(p) ? __ballot((row_of(p).y >> 30) == 1u) : 0ull);      // (always-deferred rows: the same mask bits for every body)
        if (mask != 0ull && lane == 0) {                     // (wave-uniform; word (row - first) / 64)
          block[1 + (uint32_t)p * WAVES + wave] = mask;
          atomicOr(&lsum[k & 1u], 1ull << ((uint32_t)p * WAVES + wave));
        }
      }
      if (__ballot(nbad != 0) != 0ull) {                   // (rare: a body that violates something)
#pragma unroll
        for (int p = 0; p < RP; p++)
          if ((badmask >> p) & 1u) low = min(low, S.row_id[td.x + (uint32_t)p * THREADS + fresh(make_uint4(tid, 0, 0, 0)).x]);
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) {
          nbad += (uint32_t)__shfl_xor((int)nbad, sh);
          low = min(low, (uint32_t)__shfl_xor((int)low, sh));
        }
        if (lane == 0) {
          atomicAdd(&violations[b], nbad);
          if (first) atomicMin(&first[b], low);
        }
      }
    };
    // word 0 of unit k's block, by one lane, once every wave's verdicts of the unit are behind a barrier
    auto summary = [&](const uint32_t k) {
      if (wave == 1u % (uint32_t)WAVES && lane == 0) {
        scratch[((size_t)(b_lo + k) * S.ntiles + tile) * block_words] = lsum[k & 1u];
        lsum[k & 1u] = 0ull;
      }
    };
    // ---- the pipeline
    fetch_ext(0);
    fetch_tile(0);
    lds_barrier();                                                                 // the word list (and, the first time, the coefficients) in place, sums and flags zero
    pack(0, 0);
    fetch_tile(1);
    fetch_ext(1);
    lds_barrier();
    uint32_t pend_verdict = 0, i3 = 0;                       // (i3 = i mod 3)
    bool pend_anomaly = false;
    for (uint32_t i = 0; i < m; i++) {
      if (STAMPS && stamping) t_prev = __builtin_amdgcn_s_memtime();
      if (i >= 2u && !(dbg & 1u)) summary(i - 2u);                                 // (its verdicts ran in the last iteration, behind the last barrier)
      B3W_STAMP(0);
      // the row descriptors count as new in every iteration (see fresh(): nothing derived from them is to be kept around the loop)
#pragma unroll
      for (int p = 0; p < PRE; p++) asm volatile("" : "+v"(pre[p].x), "+v"(pre[p].y), "+v"(pre[p].z), "+v"(pre[p].w));
      const uint32_t i3n = i3 == 2u ? 0u : i3 + 1u, i3nn = i3n == 2u ? 0u : i3n + 1u;
      const bool anomaly = __builtin_amdgcn_readfirstlane(lanom[i3]) != 0u;
      if (tid == 0) lanom[i3nn] = 0u;                        // (read one unit ago by everyone; the pack of the NEXT iteration may set it)
      uint32_t verdict = 0;
      if (!(dbg & 1u)) {
        if (!(dbg & 32u)) words(i);
        B3W_STAMP(1);
        verdict = rows(i, anomaly);
        B3W_STAMP(2);
        if (i) verdicts(i - 1u, pend_verdict, pend_anomaly);
        B3W_STAMP(3);
      }
      pack(i + 1u, i3n);
      fetch_tile(i + 2u);                                                          // the registers just packed out of take the unit after: in flight through the barrier and all of the next iteration
      fetch_ext(i + 2u);
      B3W_STAMP(4);
      pend_verdict = verdict; pend_anomaly = anomaly; i3 = i3n;
      lds_barrier();
      B3W_STAMP(5);
    }
    if (!(dbg & 1u)) {
      verdicts(m - 1u, pend_verdict, pend_anomaly);
      lds_barrier();
      if (m >= 2u) summary(m - 2u);
      summary(m - 1u);
    }
    u += m;
  }
  if (STAMPS && stamping && lane == 0)
    for (int k = 0; k < 8; k++) stamps[wave * 8 + k] = ph[k];
#undef B3W_STAMP
#undef has_rows
#undef has_tt
#undef has_gen
#undef has_dmask
}

__global__ __launch_bounds__(64) void b3w_r1cs_deferred_kernel(const uint8_t *__restrict__ bodies, uint64_t pitch, uint32_t n, B3wR1csSystem S,
                                                               const unsigned long long *__restrict__ scratch, uint32_t block_words, B3wField F,
                                                               uint32_t *__restrict__ violations, uint32_t *__restrict__ first, bool sparse) {
  // a wave looks at B3W_R1CS_DEFERRED_TILES consecutive tiles of one body (measured with 4: the same 22 us for a batch of
  // blake3_compression as with 1 -- the launch is bound by the first load of each wave, not by the 98 304 dispatches)
  const uint32_t tgroups = (S.ntiles + B3W_R1CS_DEFERRED_TILES - 1u) / B3W_R1CS_DEFERRED_TILES;
  const uint32_t per_group = 8u * tgroups;
  const uint32_t b = (blockIdx.x / per_group) * 8u + (blockIdx.x & 7u);
  const uint32_t tile0 = ((blockIdx.x % per_group) >> 3) * B3W_R1CS_DEFERRED_TILES;
  if (b >= n) return;
  const uint8_t *body = bodies + (uint64_t)b * pitch;
  uint32_t nbad = 0, low = 0xFFFFFFFFu;
  bool any = false;
  for (uint32_t tile = tile0; tile < tile0 + B3W_R1CS_DEFERRED_TILES && tile < S.ntiles; tile++)
    any = deferred_tile(body, b, tile, S, scratch, block_words, F, sparse, nbad, low) || any;
  if (!any) return;
  deferred_report(nbad, low, b, violations, first);
}
// the result arrays start from "no violation": a kernel rather than hipMemsetAsync, so that the whole check is made of
// kernel nodes when a caller captures it into a hipGraph (memset nodes of a captured graph were seen to leave garbage)
__global__ void b3w_r1cs_init_kernel(uint32_t *__restrict__ violations, uint32_t *__restrict__ first, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    violations[i] = 0;
    if (first) first[i] = 0xFFFFFFFFu;
  }
}

}  // namespace

static int r1cs_init_results(uint32_t *d_violations, uint32_t *d_first, uint32_t n, hipStream_t stream) {
  hipLaunchKernelGGL(b3w_r1cs_init_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_violations, d_first, n);
  return (int)hipGetLastError();
}


extern "C" int b3w_launch_r1cs(const uint8_t *d_bodies, uint32_t n, uint64_t pitch, uint32_t m, const uint32_t *d_rows,
                               const uint32_t *d_row_id, const uint32_t *d_wires, const uint16_t *d_cids, const uint32_t *d_coefR,
                               const B3wField *field, uint32_t *d_violations, uint32_t *d_first, hipStream_t stream) {
  if (!n || !m) return 0;
  hipError_t e = (hipError_t)r1cs_init_results(d_violations, d_first, n, stream);
  if (e != hipSuccess) return (int)e;
  const uint32_t row_blocks = (m + 255) / 256;
  // one-dimensional grid, bodies in slabs that keep the workgroup count below 2^31
  const uint32_t slab = (0x7FFFFFFFu / row_blocks) & ~7u;
  for (uint32_t b0 = 0; b0 < n; b0 += slab) {
    const uint32_t nb = n - b0 < slab ? n - b0 : slab;
    const uint32_t groups = (nb + 7) / 8;
    hipLaunchKernelGGL(b3w_r1cs_kernel, dim3(groups * 8 * row_blocks), dim3(256), 0, stream, d_bodies + (uint64_t)b0 * pitch, pitch, nb, m,
                       row_blocks, reinterpret_cast<const uint4 *>(d_rows), d_row_id, d_wires, d_cids, d_coefR, *field, d_violations + b0,
                       d_first ? d_first + b0 : nullptr);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

static inline uint32_t lean_block_words(const B3wR1csSystem *sys) { return 1u + 4u * ((sys->max_tile_rows + 255u) >> 8); }

// bodies per launch pair: B3W_R1CS_SLAB, fewer for a system whose scratch blocks are large (many rows per tile) so that the scratch
// stays below 64 MB, and never so many that the grid leaves 31 bits
static inline uint32_t lean_slab(const B3wR1csSystem *sys) {
  const uint64_t per_body = (uint64_t)sys->ntiles * lean_block_words(sys) * 8;
  uint64_t slab = (64ull << 20) / per_body;
  if (slab > B3W_R1CS_SLAB) slab = B3W_R1CS_SLAB;
  const uint64_t grid_cap = 0x7FFFFFFFu / sys->ntiles;
  if (slab > grid_cap) slab = grid_cap;
  slab &= ~7ull;
  return slab < 8 ? 8u : (uint32_t)slab;
}

extern "C" size_t b3w_r1cs_scratch_bytes(const B3wR1csSystem *sys) {
  return (size_t)lean_slab(sys) * sys->ntiles * lean_block_words(sys) * 8;
}

// LDS of the stream kernel for a system: elements, bit words, general rows' sums and flags (each twice: unit parity) + word list +
// coefficients + summary and anomaly words
static inline size_t stream_smem(const B3wR1csSystem *sys) {
  const uint32_t ext_cap = (sys->max_ext + 32u) & ~31u;
  const uint32_t groups = ((B3W_R1CS_TILE + ext_cap + 63u) >> 6) + 1u;
  return 2u * (size_t)(B3W_R1CS_TILE + ext_cap) * 8u + 2u * (size_t)groups * 16u + 2u * (size_t)sys->max_g_rows * 48u +
         2u * (size_t)((sys->max_g_rows + 1u) & ~1u) * 4u + 2u * (size_t)((sys->max_g_words + 4u) & ~3u) * 4u + (size_t)((sys->ncoef + 1u) & ~1u) * 8u + 32u +
         (sys->max_tile_rows > 1024u ? (size_t)(((sys->max_tile_rows + 63u) & ~63u) - 1024u) * 16u : 0u);      // (+ the row descriptors behind the first 1 024 of a tile)
}

// 0 = launched; -6 = this system does not fit the stream kernel (the caller takes the gather kernel)
extern "C" int b3w_launch_r1cs_stream(const uint8_t *d_bodies, uint32_t n, uint64_t pitch, const B3wR1csSystem *sys, const B3wField *field,
                                      unsigned long long *d_scratch, uint32_t *d_violations, uint32_t *d_first, hipStream_t stream) {
  if (!n || !sys->ntiles) return 0;
  if (!d_scratch) return -5;
  static const int env_waves = getenv("B3W_R1CS_WAVES") ? atoi(getenv("B3W_R1CS_WAVES")) : 0;
  static const int env_wgs = getenv("B3W_R1CS_WGS") ? atoi(getenv("B3W_R1CS_WGS")) : 0;         // workgroups per CU
  static const int env_grid = getenv("B3W_R1CS_GRID") ? atoi(getenv("B3W_R1CS_GRID")) : 0;
  // The experiment switches exist only in a diagnostic build (-DB3W_R1CS_DIAG: B3W_BUILD_DIAG=1 python -m ...build): B3W_R1CS_DBG
  // switches phases of the kernel off (times only — such a launch marks EVERY body as violating, it cannot pass for a check) and
  // B3W_R1CS_STAMPS prints per-phase cycle counts after each launch (synchronises; refused while the stream is capturing).
#ifdef B3W_R1CS_DIAG
  static const uint32_t env_dbg = getenv("B3W_R1CS_DBG") ? (uint32_t)atoi(getenv("B3W_R1CS_DBG")) : 0u;     // 1 no words / rows / verdicts, 2 no pack, 4 or 8 every unit fetches the tile's first body (no HBM traffic), 32 no general words
#else
  static const uint32_t env_dbg = 0u;
#endif
  if (sys->max_tile_rows > 2048u || sys->max_ext > 480u) return -6;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  // the shapes: waves per workgroup; the default is the first
  struct Shape { int waves; const void *fn; };
#ifdef B3W_R1CS_DIAG
  static const bool want_stamps = (getenv("B3W_R1CS_STAMPS") && atoi(getenv("B3W_R1CS_STAMPS"))) || env_dbg != 0u;      // (the diagnostic instantiation: phase stamps, experiment switches)
  static const bool print_stamps = getenv("B3W_R1CS_STAMPS") && atoi(getenv("B3W_R1CS_STAMPS"));
  static const Shape shapes[] = {{8, want_stamps ? reinterpret_cast<const void *>(&b3w_r1cs_stream_kernel<8, true>) : reinterpret_cast<const void *>(&b3w_r1cs_stream_kernel<8, false>)},
                                 {16, reinterpret_cast<const void *>(&b3w_r1cs_stream_kernel<16, false>)}};
  if (print_stamps) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return (int)hipErrorStreamCaptureUnsupported; }
  }
#else
  static const bool print_stamps = false;
  static const Shape shapes[] = {{8, reinterpret_cast<const void *>(&b3w_r1cs_stream_kernel<8, false>)},
                                 {16, reinterpret_cast<const void *>(&b3w_r1cs_stream_kernel<16, false>)}};
#endif
  struct PerDevice { int cus = 0, lds = 0; bool attr[2] = {false, false}; };
  static PerDevice per[64];
  static std::mutex mu;
  const size_t smem = stream_smem(sys);
  int pick = -1, cus = 0, lds = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    PerDevice &pd = per[dev & 63];
    if (!pd.cus) {
      if ((e = hipDeviceGetAttribute(&pd.cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return (int)e;
      if ((e = hipDeviceGetAttribute(&pd.lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev)) != hipSuccess) return (int)e;
      if (pd.lds < 160 * 1024) pd.lds = 64 * 1024;         // (gfx950: 160 KB per workgroup; anything else: be modest)
    }
    if (smem > (size_t)pd.lds) return -6;
    for (int k = 0; k < (int)(sizeof shapes / sizeof shapes[0]) && pick < 0; k++)
      if (!env_waves || shapes[k].waves == env_waves) pick = k;
    if (pick < 0) return -6;
    if (!pd.attr[pick]) {
      if ((e = hipFuncSetAttribute(shapes[pick].fn, hipFuncAttributeMaxDynamicSharedMemorySize, pd.lds)) != hipSuccess) return (int)e;
      pd.attr[pick] = true;
    }
    cus = pd.cus; lds = pd.lds;
  }
  // workgroups per CU: what the LDS holds, and sixteen waves (the kernels are built for four waves per SIMD: 128 VGPRs)
  int wgs = (int)((size_t)lds / smem);
  if (wgs > 16 / shapes[pick].waves) wgs = 16 / shapes[pick].waves;
  if (env_wgs > 0) wgs = env_wgs;
  if (wgs < 1) wgs = 1;
  e = (hipError_t)r1cs_init_results(d_violations, d_first, n, stream);
  if (e != hipSuccess) return (int)e;
  // diagnostics (B3W_R1CS_STAMPS=1): per-phase cycle sums of workgroup 0, printed after every launch (synchronises: not for timing runs)
  static unsigned long long *d_stamps = nullptr;
  if (print_stamps && !d_stamps && hipMalloc((void **)&d_stamps, 16 * 8 * 8) != hipSuccess) d_stamps = nullptr;
  const uint32_t bw = lean_block_words(sys);
  const uint32_t slab = lean_slab(sys);
  for (uint32_t b0 = 0; b0 < n; b0 += slab) {
    const uint32_t nb = n - b0 < slab ? n - b0 : slab;
    const uint64_t units = (uint64_t)nb * sys->ntiles;
    uint32_t grid = env_grid > 0 ? (uint32_t)env_grid : (uint32_t)(cus * wgs);     // persistent workgroups
    if (grid > units) grid = (uint32_t)units;
    void *args[] = {(void *)&d_bodies, (void *)&pitch, (void *)&nb, (void *)sys, (void *)&d_scratch, (void *)&bw, (void *)&d_violations, (void *)&d_first,
                    (void *)&env_dbg, (void *)&d_stamps};
    const uint8_t *bodies0 = d_bodies + (uint64_t)b0 * pitch;
    uint32_t *viol0 = d_violations + b0, *first0 = d_first ? d_first + b0 : nullptr;
    args[0] = (void *)&bodies0; args[6] = (void *)&viol0; args[7] = (void *)&first0;
    e = hipLaunchKernel(shapes[pick].fn, dim3(grid), dim3(64u * (uint32_t)shapes[pick].waves), args, smem, stream);
    if (e != hipSuccess) return (int)e;
    if (d_stamps) {
      unsigned long long h[128];
      if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(h, d_stamps, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
        const double un = (double)(units / grid);
        static const char *name[8] = {"fetch-issue", "words", "own-rows", "verdicts", "wait+pack", "barrier", "-", "-"};
        fprintf(stderr, "b3w_r1cs_stream stamps (cycles per unit, workgroup %u of %u, %d waves, %g units):\n", grid / 2, grid, shapes[pick].waves, un);
        for (int k = 0; k < 8; k++) {
          fprintf(stderr, "  %-15s", name[k]);
          for (int w = 0; w < shapes[pick].waves; w++) fprintf(stderr, " %5.0f", (double)h[w * 8 + k] / un);
          fprintf(stderr, "\n");
        }
      }
    }
    if (env_dbg) {                                         // an experiment that skips phases leaves no valid masks behind: no verdict
      if ((e = hipMemsetAsync(d_violations + b0, 0xFF, (size_t)nb * 4, stream)) != hipSuccess) return (int)e;
      continue;
    }
    const dim3 dgrid(((nb + 7) / 8) * 8 * ((sys->ntiles + B3W_R1CS_DEFERRED_TILES - 1u) / B3W_R1CS_DEFERRED_TILES));
    hipLaunchKernelGGL(b3w_r1cs_deferred_kernel, dgrid, dim3(64), 0, stream, d_bodies + (uint64_t)b0 * pitch, pitch, nb, *sys, d_scratch, bw,
                       *field, d_violations + b0, d_first ? d_first + b0 : nullptr, true);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
