// b3w_r1cs_walk.hip — the WALK formulation of the on-device constraint check (round 4; the default where a system fits) and the
// deferred kernel behind it; b3w_r1cs.hip has the other three formulations and the story of all four.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include "b3w_kernels.h"
#include "b3w_r1cs_device.h"

namespace {
// two sums over the wave at the price of one: the lower half of the wave reduces a, the upper half b
__device__ __forceinline__ void wave_sum2(Fe &a, Fe &b, const uint32_t p[8]) {
  const bool upper = (threadIdx.x & 32u) != 0u;
  Fe keep = upper ? b : a, give = upper ? a : b;
#pragma unroll
  for (int i = 0; i < 8; i++) give.l[i] = (uint32_t)__shfl_xor((int)give.l[i], 32);
  fe_add(keep, give, p);
#pragma unroll
  for (int sh = 16; sh > 0; sh >>= 1) {
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.l[i] = (uint32_t)__shfl_xor((int)keep.l[i], sh);
    fe_add(keep, o, p);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    a.l[i] = (uint32_t)__builtin_amdgcn_readlane((int)keep.l[i], 0);
    b.l[i] = (uint32_t)__builtin_amdgcn_readlane((int)keep.l[i], 32);
  }
}
// An always-deferred row of the walk kernel's list, by one wave: d = {first pair, pairs, linear, has C terms}, pairs = the row's
// UNIQUE terms {wire, coefficient id | parts << 16} (b3w_r1cs_api.cpp) — a term that stands in A and in B (X (X - 1) = 0) is multiplied
// once and added twice; element and coefficient are loaded side by side, the next pair's while this one is multiplied.
__device__ __forceinline__ bool static_row_wave(const uint8_t *body, const uint4 d, const uint2 *pairs, const uint32_t *coefR, const B3wField &F) {
  bool wild = false;
  const uint32_t lane = threadIdx.x & 63u, n = d.y;
  Fe az, bz, cz;
#pragma unroll
  for (int i = 0; i < 8; i++) az.l[i] = bz.l[i] = cz.l[i] = 0;
  uint32_t meta = 0;
  uint4 zlo = make_uint4(0, 0, 0, 0), zhi = zlo;
  Fe cf;
  auto issue = [&](uint32_t k) {                            // (k < n)
    const uint2 pr = pairs[d.x + k];
    meta = pr.y;
    const uint4 *q = reinterpret_cast<const uint4 *>(body + (size_t)pr.x * 32);
    zlo = q[0]; zhi = q[1];
    cf = load_fe(coefR + (size_t)(pr.y & 0xFFFFu) * 16 + 8);
  };
  if (lane < n) issue(lane);
  for (uint32_t k = lane; k < n; k += 64u) {
    Fe z;
    z.l[0] = zlo.x; z.l[1] = zlo.y; z.l[2] = zlo.z; z.l[3] = zlo.w;
    z.l[4] = zhi.x; z.l[5] = zhi.y; z.l[6] = zhi.z; z.l[7] = zhi.w;
    const uint32_t c = meta & 0xFFFFu, parts = meta >> 16;
    const Fe cfk = cf;
    if (k + 64u < n) issue(k + 64u);
    if (fe_geq(z, F.p)) {
      wild = true;
      do fe_sub_p(z, F.p); while (fe_geq(z, F.p));
    }
    if (fe_is_zero(z)) continue;
    Fe t = z;
    if (c >= 2u) t = mont_mul(cfk, z, F);                   // (coef * R) * z / R
    if (c == 1u) {                                          // - z  =  + (p - z)   (z != 0)
      Fe pm;
#pragma unroll
      for (int i = 0; i < 8; i++) pm.l[i] = F.p[i];
      fe_sub(pm, t, F.p);
      t = pm;
    }
    if (parts & 1u) fe_add(az, t, F.p);
    if (parts & 2u) fe_add(bz, t, F.p);
    if (parts & 4u) fe_add(cz, t, F.p);
  }
  wave_sum2(az, bz, F.p);
  if (d.w) cz = wave_sum(cz, F.p);                          // (wave-uniform)
  return row_violated(az, bz, cz, d.z != 0u, F) || __ballot(wild) != 0;
}

// behind the walk kernel: one workgroup of TWO waves per BODY, told by one word which of the body's tiles have deferred rows at all
// (bit t = tile t) — a batch of valid blake3_compression witnesses is 4 096 workgroups that load a zero and leave.  The waves share a
// flagged tile's mask words.  Everything here is a chain of three dependent loads and a field multiplication, 145 VGPRs allow twelve
// waves on a CU, so what counts is that no resident wave idles: a nova step's 134 wide records are three rounds of wave 0, its
// always-deferred row of 133 terms three rounds of wave 1 (four waves per body, three of them waiting for the fourth: 50 us per
// 4 096 nova bodies; two: 37 us).  What is left is arithmetic, not waiting: 1.5e7 wave instructions per launch, a third of them
// quarter-rate multiplications (133 Montgomery products per body), keep the chip's 1 024 SIMDs busy for most of those 37 us — issuing
// the loads of a level side by side (below, and in gather_row_wave) and a build squeezed to 128 VGPRs (4 waves per SIMD, 104 bytes
// of scratch) each moved it by 1 us.
#define B3W_WALK_DEFERRED_WAVES 2u
// what the deferred kernel needs of the system (B3wR1csSystem by value was 122 scalar spills in this kernel)
struct B3wDeferredView {
  uint32_t ntiles;
  const uint32_t *tiles, *row_k, *row_id, *g_rows, *g_wires;
  const uint16_t *g_cids;
  const uint32_t *coefs;
};
__global__ __launch_bounds__(64 * B3W_WALK_DEFERRED_WAVES) void b3w_r1cs_walk_deferred_kernel(const uint8_t *__restrict__ bodies, uint64_t pitch, uint32_t n, B3wDeferredView S,
                                                                    const unsigned long long *__restrict__ scratch, uint32_t block_words,
                                                                    const unsigned long long *__restrict__ body_flags,
                                                                    const unsigned long long *__restrict__ wide_recs,
                                                                    const uint32_t *__restrict__ static_d /* 4 words per row, then the rows' unique terms as pairs: static_row_wave */, uint4 static_d0,
                                                                    const uint32_t *__restrict__ static_id, uint32_t nstatic, B3wField F, uint32_t *__restrict__ violations, uint32_t *__restrict__ first) {
  const uint32_t b = blockIdx.x, wave = threadIdx.x >> 6;
  const uint8_t *body = bodies + (uint64_t)b * pitch;
  uint32_t nbad = 0, low = 0xFFFFFFFFu;
  // Everything below is latency: what can be asked for at once, is.  The body's word and — on speculation, the place is valid memory
  // whatever it holds — this lane's first wide record go out side by side; the element a record points to is loaded only once the
  // word has said that the record exists.
  // (with always-deferred rows the last wave has its share in them: the records are the other waves')
  const uint32_t rec_waves = nstatic ? B3W_WALK_DEFERRED_WAVES - 1u : B3W_WALK_DEFERRED_WAVES;
  const bool rec_lane = wave < rec_waves && threadIdx.x < B3W_WALK_WIDE_CAP;
  unsigned long long flags = body_flags[b];
  unsigned long long rec[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
  if (rec_lane) {
    const unsigned long long *at = wide_recs + ((size_t)b * B3W_WALK_WIDE_CAP + threadIdx.x) * 5u;
#pragma unroll
    for (int q = 0; q < 5; q++) rec[q] = at[q];
  }
  // the system's ALWAYS-deferred rows (a coefficient that is no small integer: one row of 133 terms in each O2 nova system), for every
  // body, by the last wave, straight from the kernel's own list — no flag, no block, no mask word stands between the launch and the
  // row's terms; the first one's descriptor is a kernel argument
  if (wave == B3W_WALK_DEFERRED_WAVES - 1u)
    for (uint32_t sr = 0; sr < nstatic; sr++) {
      const uint4 d = sr ? reinterpret_cast<const uint4 *>(static_d)[sr] : static_d0;
      const bool bad = static_row_wave(body, d, reinterpret_cast<const uint2 *>(static_d), S.coefs, F);
      if ((threadIdx.x & 63u) == 0 && bad) { nbad++; low = min(low, static_id[sr]); }
    }
  if (flags == 0ull) {                                       // (wave-uniform)
    if (nstatic) deferred_report(nbad, low, b, violations, first);
    return;
  }
  // WIDE RECORDS: rows with ONE term s * W (s = +-1, W an element of 2^63 or more, as it lies in the body): with the other terms'
  // sums a, b, c — small integers, from the walk kernel — the row says (a + sW) b = c, a (b + sW) = c or a b = c + sW, i.e. k W = d
  // with k = s b, s a or s and d = c - a b or a b - c: decided by small_product_is (|k| < 2^32; else the general road).  An element
  // that is no canonical representative (>= p) violates the row, as everywhere.
  const uint32_t nwide = (uint32_t)(flags >> 56);
  flags &= (1ull << 56) - 1ull;
  for (uint32_t wr = threadIdx.x; wr < nwide && wave < rec_waves; wr += 64u * rec_waves) {
    if (wr != threadIdx.x) {
      const unsigned long long *at = wide_recs + ((size_t)b * B3W_WALK_WIDE_CAP + wr) * 5u;
#pragma unroll
      for (int q = 0; q < 5; q++) rec[q] = at[q];
    }
    const long long ra = (long long)rec[0], rb = (long long)rec[1];
    const unsigned long long c_lo = rec[2];
    const long long c_hi = (long long)rec[3];
    const unsigned long long gw = rec[4];
    const uint32_t part = (uint32_t)(gw >> 16) & 3u, tile = (uint32_t)(gw >> 32) & 0xFFu, row = (uint32_t)(gw >> 40);
    const bool sneg = (gw >> 18) & 1ull;
    bool wild = false;
    const Fe Wv = load_z(body, tile * B3W_R1CS_TILE + (uint32_t)(gw & 0xFFFFull), F, &wild);
    const __int128 ab = (__int128)ra * (__int128)rb, cc = (__int128)(((unsigned __int128)(unsigned long long)c_hi << 64) | c_lo);
    const __int128 d = part == 2u ? ab - cc : cc - ab;
    long long k = part == 0u ? rb : part == 1u ? ra : 1ll;
    if (sneg) k = -k;
    const unsigned long long kmag = k < 0 ? 0ull - (unsigned long long)k : (unsigned long long)k;
    const unsigned __int128 dmag = d < 0 ? (unsigned __int128)(-d) : (unsigned __int128)d;
    bool bad;
    if (wild) bad = true;
    else if (kmag >> 32) bad = gather_row(body, reinterpret_cast<const uint4 *>(S.g_rows)[S.row_k[row]], S.g_wires, S.g_cids, S.coefs, F);
    else if (kmag == 0ull) bad = dmag != 0;
    else {
      Fe dv;                                                   // d mod p  (|d| < 2^127 < p)
      dv.l[0] = (uint32_t)dmag; dv.l[1] = (uint32_t)(dmag >> 32); dv.l[2] = (uint32_t)(dmag >> 64); dv.l[3] = (uint32_t)(dmag >> 96);
      dv.l[4] = dv.l[5] = dv.l[6] = dv.l[7] = 0u;
      if (d < 0) {
        Fe pm;
#pragma unroll
        for (int q = 0; q < 8; q++) pm.l[q] = F.p[q];
        fe_sub(pm, dv, F.p);
        dv = pm;
      }
      bad = !small_product_is((uint32_t)kmag, k < 0, Wv, dv, F);
    }
    if (bad) { nbad++; low = min(low, S.row_id[row]); }
  }
  while (flags) {
    const uint32_t tile = (uint32_t)__ffsll((long long)flags) - 1u;
    flags &= flags - 1ull;
    (void)deferred_tile(body, b, tile, S, scratch, block_words, F, true, nbad, low, wave, B3W_WALK_DEFERRED_WAVES);
  }
  deferred_report(nbad, low, b, violations, first);
}

// ---- WALK kernel (round 4) -------------------------------------------------------------------------------------------------------
// The stream kernel above walks the units tile-major: a workgroup keeps one tile's program and sees body after body.  Every row of
// a tile that mentions a wire of another tile makes it GATHER that wire from HBM — 32 bytes wanted, a 128-byte line fetched, and
// the line was read once already as part of its own tile by another workgroup at another time: 13 % (compression) to 18 % (nova)
// more bytes from HBM than the bodies hold (profiles/r03/r1cs_check.json), on a kernel that sits on the HBM roofline.
// Here a workgroup walks WHOLE BODIES, tile after tile (units body-major).  A row belongs to the tile of its HIGHEST wire (host:
// b3w_r1cs_host.cpp, "WALK program"), so every other wire it mentions lies in the same tile or in one the workgroup has already
// had in LDS for this body: each tile EXPORTS the elements later tiles mention (1 500 - 1 600 per body) into an export area in LDS,
// 8 bytes each plus a bit-packed copy, and a row names an element as "local e" or "export slot s".  No outside wire is ever
// fetched; HBM traffic = the bodies.  The tile's program now changes with every unit, so it is made small enough to be read from
// L2 unit after unit, one unit ahead, straight into registers:
//   * booleanity rows (64 % of the rows): still one AND of the tile's must-be-bit mask with the pack's ballots, no descriptor;
//   * truth-table rows come in RUNS — the 32 XOR gates of a word have operands a_i, b_i, out_i that advance by one element from
//     row to row — and ONE lane decides a run of up to 32 rows from three 32-bit cut-outs of the bit-packed elements with a
//     bit-sliced evaluation of the table (16 + 15 v_bfi): 7 671 rows of blake3_compression are 480 descriptors of 16 bytes;
//   * general rows as before: one ENTRY (term or bit run) per lane into per-row sums in LDS, verdict by the row's owner lane.
// A (body, tile) in which something is no bit where the masks say bit (or wire 0 is not 1) goes to the deferred kernel with ALL its
// rows, and so does every later tile of that body (they may import from it); a general row with an element of 2^63 or more (the
// field inverses of a nova step) is deferred alone.  The deferred kernel and its scratch blocks are the stream kernel's.
// One workgroup owns a body: its violation count is a plain store, there is no initialisation kernel.

// An element as the walk kernel keeps it, 8 bytes: a SIGNED small number — v < 2^63 as it stands, and p - k (0 < k <= 2^62) as -k:
// unsimplified systems hold their small negative numbers that way (the circomkit nova build: 121 rows a step over such wires, which the
// unsigned form sent to the deferred kernel row by row) — or B3W_WALK_WIDE for everything else (a field inverse).  The test for
// "near p" is one compare of the top limb; the subtraction behind it runs only in a wave that has such an element.
#define B3W_WALK_WIDE 0x8000000000000000ull
// (p7 = the prime's top limb, a scalar; the whole prime lies in LDS for the rare road: eight more scalars held across the kernel's
// loop were 30 more scalar spills, 2 % of a nova check)
__device__ __forceinline__ unsigned long long walk_pack(const uint4 lo, const uint4 hi, const uint32_t p7, const uint32_t *p /* LDS */) {
  const uint32_t wide = lo.z | lo.w | hi.x | hi.y | hi.z | hi.w | (lo.y & 0x80000000u);
  unsigned long long z = wide ? B3W_WALK_WIDE : (unsigned long long)lo.x | (unsigned long long)lo.y << 32;
  if (hi.w == p7) {                                          // (ONE branch, and rare in an optimised system's bodies: p - k has the prime's top limb)
    const uint32_t e[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint32_t d[8], borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint64_t t = (uint64_t)p[i] - e[i] - borrow;
      d[i] = (uint32_t)t;
      borrow = (uint32_t)(t >> 63);
    }
    const unsigned long long k = (unsigned long long)d[0] | (unsigned long long)d[1] << 32;
    const bool fits = !borrow && !(d[2] | d[3] | d[4] | d[5] | d[6] | d[7]) && k != 0ull && k <= (1ull << 62);      // (k = 0: the element is p itself, no witness value)
    z = fits ? 0ull - k : z;
  }
  return z;
}
// 32 bits of a bit-packed array from bit `idx` on (the array has a spare word behind its last)
__device__ __forceinline__ uint32_t cut32(const unsigned long long *words, uint32_t idx) {
  const uint32_t g = idx >> 6, r = idx & 63u;
  const unsigned long long lo = words[g], hi = words[g + 1];
  return (uint32_t)(r ? (lo >> r) | (hi << (64u - r)) : lo);
}
// a 5-input truth table over 32 rows at once: bit j of the result = table[x0_j + 2 x1_j + 4 x2_j + 8 x3_j + 16 x4_j]
__device__ __forceinline__ uint32_t table32(uint32_t table, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4) {
  uint32_t l[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint32_t t0 = 0u - ((table >> (2 * j)) & 1u), t1 = 0u - ((table >> (2 * j + 1)) & 1u);
    l[j] = (x0 & t1) | (~x0 & t0);
  }
#pragma unroll
  for (int j = 0; j < 8; j++) l[j] = (x1 & l[2 * j + 1]) | (~x1 & l[2 * j]);
#pragma unroll
  for (int j = 0; j < 4; j++) l[j] = (x2 & l[2 * j + 1]) | (~x2 & l[2 * j]);
#pragma unroll
  for (int j = 0; j < 2; j++) l[j] = (x3 & l[2 * j + 1]) | (~x3 & l[2 * j]);
  return (x4 & l[1]) | (~x4 & l[0]);
}

#ifndef B3W_WALK_ABLATE
#define B3W_WALK_ABLATE 0      // measurements only (tools/ubench/walk_ablate.py builds the variants; never set in a product build). Leaves OUT: 1 the
#endif                         // truth-table runs, 2 the verdicts, 4 the entries, 8 the exports, 32 the scratch blocks, 2048 the wide records, 4096 their stores
template <int NE, bool SIGNED, bool STAMPS = false>          // SIGNED: elements p - k count as -k (walk_pack); chunks of 64 general entries a wave takes at most (the tile with most: NE * 512); STAMPS: the diagnostic build's
__global__ __launch_bounds__(512, 4) void b3w_r1cs_walk_kernel(const uint8_t *__restrict__ bodies, uint64_t pitch, uint32_t n, B3wWalk W,
                                                               unsigned long long *__restrict__ scratch, uint32_t block_words,
                                                               unsigned long long *__restrict__ body_flags /* per body: bit t = tile t has deferred rows; bits 56 up: wide records */,
                                                               unsigned long long *__restrict__ wide_recs /* per body B3W_WALK_WIDE_CAP x 5 words */,
                                                               uint32_t *__restrict__ violations, uint32_t *__restrict__ first,
                                                               unsigned long long *__restrict__ stamps /* STAMPS: per wave 8 cycle sums of the middle workgroup */) {
  constexpr uint32_t WAVES = 8, THREADS = 512, T = B3W_R1CS_TILE;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
  const bool stamping = STAMPS && stamps != nullptr && blockIdx.x == gridDim.x / 2u;
#define B3W_WSTAMP(k) do { if (STAMPS && stamping) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); ph[k] += t_now - t_prev; t_prev = t_now; } } while (0)
  extern __shared__ __align__(16) unsigned char smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // LDS: elements [parity][1 024]; export area; bit-packed "is 1" words of both (+ a spare word each); the general rows' sums
  // {A, B, C} x {low, high} and flags [parity]; the deferred-row mask words of a unit [parity]; tile table; coefficients; counters
  const uint32_t xw = (W.exp_slots >> 6) + 1u, gr2 = (W.max_gen + 1u) & ~1u;
  unsigned long long *el0 = reinterpret_cast<unsigned long long *>(smem);
  unsigned long long *xel = el0 + 2u * T;
  unsigned long long *ones0 = xel + W.exp_slots;             // [parity][18]
  unsigned long long *xones = ones0 + 36u;
  unsigned long long *gsum0 = xones + xw;
  unsigned long long *dmask0 = gsum0 + 12u * W.max_gen;      // [parity][8]
  long long *lcoef = reinterpret_cast<long long *>(dmask0 + 20u);      // (behind the three body-flag words)
  unsigned long long *lmask = reinterpret_cast<unsigned long long *>(lcoef + ((W.ncoef + 1u) & ~1u));      // the tiles' must-be-bit masks
  unsigned long long *lstat = lmask + 16u * W.ntiles;        // ... and always-deferred rows
  uint32_t *gflag0 = reinterpret_cast<uint32_t *>(lstat + (size_t)W.static_words * W.ntiles);
  uint32_t *gwide0 = gflag0 + 2u * gr2;                      // [parity][general row]: the row's ONE term +-1 * (element of 2^63 or more), see entries
  uint32_t *ltile = gwide0 + 2u * gr2;
  uint32_t *lanom = ltile + W.ntiles * B3W_WT_WORDS;         // [unit mod 3]
  uint32_t *cnt = lanom + 4;                                 // [body mod 3]: violations, then lowest violated row
  uint32_t *lprime = cnt + 9;                                // the field's prime (walk_pack)
  const uint32_t p7 = W.p[7];
  unsigned long long *bflag = dmask0 + 16u;                  // [body mod 3]: tiles with deferred rows
  for (uint32_t k = tid; k < W.ncoef; k += THREADS) lcoef[k] = W.coef_small[k];
  for (uint32_t k = tid; k < W.ntiles * B3W_WT_WORDS; k += THREADS) ltile[k] = W.tile[k];
  for (uint32_t k = tid; k < 16u * W.ntiles; k += THREADS) lmask[k] = W.mask[k];
  for (uint32_t k = tid; k < W.static_words * W.ntiles; k += THREADS) lstat[k] = W.stat[k];
  for (uint32_t k = tid; k < 12u * W.max_gen; k += THREADS) gsum0[k] = 0ull;
  for (uint32_t k = tid; k < 4u * gr2; k += THREADS) gflag0[k] = 0u;      // (flags and wide-term words)
  for (uint32_t k = tid; k < xw; k += THREADS) xones[k] = 0ull;
  for (uint32_t k = tid; k < W.exp_slots; k += THREADS) xel[k] = 0ull;
  if (tid < 36) ones0[tid] = 0ull;
  if (tid < 19) dmask0[tid] = 0ull;                          // (and the body flags)
  if (tid < 4) lanom[tid] = 0u;
  if (tid < 3) { cnt[tid] = 0u; cnt[3 + tid] = 0xFFFFFFFFu; cnt[6 + tid] = 0u; }      // (violations, lowest violated row, wide records)
  if (tid < 8) lprime[tid] = W.p[tid];
  // this workgroup's bodies: whole ones, contiguous, as even as whole bodies go
  const uint32_t b0 = (uint32_t)((uint64_t)n * blockIdx.x / gridDim.x), b1 = (uint32_t)((uint64_t)n * (blockIdx.x + 1u) / gridDim.x);
  const uint32_t m = (b1 - b0) * W.ntiles;                    // units
  if (m == 0) return;
  lds_barrier();
  // unit k of this workgroup = body b0 + k / ntiles, tile k % ntiles: three cursors walk ahead of each other (fetch, pack, evaluate)
  // (a cursor carries its tile's table row in the lanes of one VGPR — lane k holds word k — so that a field costs a v_readlane,
  // not an LDS round trip: one ds_read per unit and cursor)
  struct Cursor { uint32_t body, tile, rec; };
  auto load_rec = [&](Cursor &c) { c.rec = ltile[c.tile * B3W_WT_WORDS + (lane & 15u)]; };
  auto advance = [&](Cursor &c) { if (++c.tile == W.ntiles) { c.tile = 0; c.body++; } load_rec(c); };
#define TW(c, field) ((uint32_t)__builtin_amdgcn_readlane((int)(c).rec, (field)))
  // ---- fetch (unit k): this wave's two groups of 64 elements, each as two blocks of 32 — lanes 0-31 the low, 32-63 the high 16 bytes
  uint4 rlo[2], rhi[2];
  uint32_t foff[2][2];                                       // this lane's four places in a tile (bytes): constant, but for the end of the last tile
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int h = 0; h < 2; h++) foff[q][h] = ((wave + (uint32_t)q * WAVES) * 64u + (uint32_t)h * 32u + (lane & 31u)) * 32u + (lane >> 5) * 16u;
  auto fetch = [&](const Cursor c) {                        // (a cursor behind the last unit points at the last unit again: fetched, packed, never looked at)
    const uint8_t *body = bodies + (uint64_t)c.body * pitch + (uint64_t)TW(c, B3W_WT_SRC) * (T * 32u);      // (a unit = a tile, or one of several over the same tile)
    const uint32_t lim = (TW(c, B3W_WT_NLOCAL) - 1u) * 32u + (lane >> 5) * 16u;      // (an element behind the tile's end: its last element again)
#pragma unroll
    for (int q = 0; q < 2; q++) {
      rlo[q] = ldg16<true>(body, min(foff[q][0], lim));
      rhi[q] = ldg16<true>(body, min(foff[q][1], lim));
    }
  };
  // ---- who does what beside the equal shares (fetch and pack): every wave's instruction stream is a chain of LDS round trips that
  // ends at the unit's barrier, so the extra kinds of work go to DIFFERENT waves — the general rows' verdicts to waves 0 ... (the
  // rows' lanes), the scratch block and the body's results to wave 1, the truth-table runs to waves 3 ..., the exports to waves 4 ...,
  // the general entries to the last waves (chunk c to wave 7 - c mod 8)
  const uint32_t ewave = (WAVES - 1u) - wave;                 // this wave's first chunk of entries
  const uint32_t rtid = (tid + THREADS - 3u * 64u) % THREADS, rwave = (wave + WAVES - 3u) % WAVES;      // this lane's run, were there that many
  const uint32_t xtid = (tid + THREADS - 4u * 64u) % THREADS, xwave = (wave + WAVES - 4u) % WAVES;      // ... and its export
  // ---- the program of a unit, read one unit ahead from L2 into registers: this wave's entry chunks, its run, its export
  uint32_t pe_w[NE], pe_m[NE], px = 0;
  uint4 prun = make_uint4(0, 0, 0, 0);
  auto program = [&](const Cursor c) {
    const uint32_t ent_off = TW(c, B3W_WT_ENT_OFF), ent_n = TW(c, B3W_WT_ENT_N);
#pragma unroll
    // (only what this wave will use — the loads stand in FRONT of the next fetch, where a wait for them never waits for a fetch)
    for (int q = 0; q < NE; q++) {
      const uint32_t iw = (ewave + (uint32_t)q * WAVES) * 64u + lane;
      if ((ewave + (uint32_t)q * WAVES) * 64u < ent_n) {                                   // (wave-uniform)
        pe_w[q] = W.ent_w[ent_off + iw];                                                   // (behind the tile's entries: the next tile's or the spare)
        pe_m[q] = iw < ent_n ? W.ent_m[ent_off + iw] : 4u;
      }
    }
    const uint32_t run_n = TW(c, B3W_WT_RUN_N);
    if (rwave * 64u < run_n) prun = W.runs[TW(c, B3W_WT_RUN_OFF) + rtid];
    const uint32_t exp_n = TW(c, B3W_WT_EXP_N);
    if (xwave * 64u < exp_n) px = W.exp[TW(c, B3W_WT_EXP_OFF) + xtid];
  };
  // ---- pack (unit k, into parity k & 1): 32-byte elements -> 8 bytes + the "is 1" word of each group of 64; something the tile's
  // rows take for a bit that is none, or wire 0 not being 1, raises the unit's anomaly flag
  auto pack = [&](const Cursor c, const bool real, const uint32_t par, const uint32_t k3) {      // (real: not a repeat of the last unit)
#pragma unroll
    for (int q = 0; q < 2; q++)
      asm volatile("" :: "v"(rlo[q].x), "v"(rlo[q].y), "v"(rlo[q].z), "v"(rlo[q].w), "v"(rhi[q].x), "v"(rhi[q].y), "v"(rhi[q].z), "v"(rhi[q].w));
    const uint32_t tile = c.tile;
    const uint32_t n_local = TW(c, B3W_WT_NLOCAL);
    unsigned long long *el = el0 + par * T, *ones = ones0 + par * 18u;
    bool flag = false;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint32_t g = wave + (uint32_t)q * WAVES, e = g * 64u + lane;
      uint4 lo = rlo[q], hi = rhi[q];
      halves_apart(lo.x, hi.x); halves_apart(lo.y, hi.y); halves_apart(lo.z, hi.z); halves_apart(lo.w, hi.w);
      const unsigned long long z = e < n_local ? (SIGNED ? walk_pack(lo, hi, p7, lprime) : lean_pack(lo, hi)) : 0ull;
      el[e] = z;
      const unsigned long long is1 = __ballot(z == 1ull), bads = __ballot(z > 1ull);
      if (lane == 0) ones[g] = is1;
      const unsigned long long mbit = lmask[tile * 16u + g];
      flag = flag || (bads & mbit) != 0ull || (tile == 0 && g == 0 && !(is1 & 1ull));
    }
    if (flag && lane == 0 && real) lanom[k3] = 1u;
  };
  // ---- general entries of unit k (its program in pe_*): a lane adds coefficient * element (a term) or the value of a bit run
  // into its row's part sum; a part sum is two counters {low, high} worth low + high * 2^52 (see the stream kernel)
  auto entries = [&](const Cursor c, const uint32_t par) {
    const unsigned long long *el = el0 + par * T, *ones = ones0 + par * 18u;
    unsigned long long *gsum = gsum0 + par * 6u * W.max_gen;
    uint32_t *gflag = gflag0 + par * gr2, *gwide = gwide0 + par * gr2;
    const uint32_t ent_n = TW(c, B3W_WT_ENT_N), ent_runs = TW(c, B3W_WT_ENT_RUNS);
#pragma unroll
    for (int q = 0; q < NE; q++) {
      const uint32_t c0 = (ewave + (uint32_t)q * WAVES) * 64u;
      if (c0 >= ent_n) break;                                  // (wave-uniform)
      const uint32_t w = pe_w[q], mt = pe_m[q];
      const bool live = !(mt & 4u);
      unsigned long long *sum = gsum + 6u * (mt >> 8) + 2u * (mt & 3u);
      if (c0 < ent_runs) {                                     // a chunk of bit runs: first element | length << 16 | shift << 23 | negative << 29
        const uint32_t idx0 = w & 0xFFFFu, len = (w >> 16) & 0x7Fu, sh = (w >> 23) & 0x3Fu;
        const bool neg = (w >> 29) & 1u;
        const unsigned long long *src = idx0 < T ? ones : xones;
        const uint32_t i0 = idx0 < T ? idx0 : idx0 - T, g = i0 >> 6, r = i0 & 63u;
        const unsigned long long lo = src[g], hi = src[g + 1];
        const unsigned long long mask = len == 64u ? ~0ull : (1ull << len) - 1ull;
        const unsigned long long v = (((lo >> r) | (r ? hi << (64u - r) : 0ull)) & mask) << sh;      // (below 2^62: the host checks shift + length; no-bit elements: the masks)
        if (live) {
          if (v < (1ull << 54)) atomicAdd(sum, neg ? 0ull - v : v);
          else {
            const unsigned long long v0 = v & ((1ull << 52) - 1ull), v1 = v >> 52;
            atomicAdd(sum, neg ? 0ull - v0 : v0);
            atomicAdd(sum + 1, neg ? 0ull - v1 : v1);
          }
        }
      } else {                                                 // a chunk of terms: element | coefficient id << 16, or element | k << 16 | negative << 22 | 1 << 31 for +-2^k
        const uint32_t idx = live ? w & 0xFFFFu : 0u;
        const unsigned long long z = idx < T ? el[idx] : xel[idx - T];      // (walk_pack: below 2^63 as it stands; bit 63: -k, or B3W_WALK_WIDE)
        unsigned long long mag, lo, hi;
        bool neg, small;
        if (__ballot(live && !(w >> 31)) == 0ull) {            // every term of the chunk shifts (98 % of all terms do; the host puts the others last): no
          const uint32_t k = (w >> 16) & 63u;                  // coefficient from the table, no 64 x 64 multiplication
          neg = (w >> 22) & 1u;
          lo = z << k;
          hi = k ? z >> (64u - k) : 0ull;
          mag = 1ull << k;
          small = true;
        } else {
          long long cf;
          if (w >> 31) { const long long one_k = 1ll << ((w >> 16) & 63u); cf = (w >> 22) & 1u ? -one_k : one_k; }
          else cf = lcoef[live ? w >> 16 : 0u];
          mag = cf < 0 ? 0ull - (unsigned long long)cf : (unsigned long long)cf;
          lo = mag * z; hi = __umul64hi(mag, z);
          neg = cf < 0;
          small = cf != B3W_R1CS_NOT_SMALL;
        }
        const bool ok = small && !(z >> 63) && hi < (1ull << 39);      // (the road every term of an optimised system's valid body takes)
        // ONE term +-1 * (a local element that is no small number) — the field inverse of an IsZero gadget, "in * inv = 1 - out" — stays
        // out of the sums and is remembered with its row: the verdict lane then hands the deferred kernel the row's small sums and the
        // element's place (a WIDE RECORD) instead of the row — one load there, not a walk through the row's terms
        const bool wide1 = (SIGNED ? z == B3W_WALK_WIDE : (z >> 63) != 0ull) && mag == 1ull && idx < T;
        if (live) {
          if (wide1) atomicAdd(&gwide[mt >> 8], 1u << 24 | (mt & 3u) << 16 | (neg ? 1u << 18 : 0u) | idx);      // (bits 24 up count such terms: the verdict lane takes exactly one)
          else if (!ok) {
            // off the road.  With SIGNED elements a NEGATIVE small one (walk_pack: p - k as -k; an unsimplified system's bodies are full
            // of them) is multiplied again by its magnitude and added with the other sign; everything else marks its row for the
            // deferred kernel
            bool negel = false;
            if constexpr (SIGNED) {
              const unsigned long long za = 0ull - z;
              const unsigned long long n_lo = mag * za, n_hi = __umul64hi(mag, za);
              negel = small && (z >> 63) && z != B3W_WALK_WIDE && n_hi < (1ull << 39);
              __int128 v = (__int128)(((unsigned __int128)n_hi << 64) | n_lo);
              if (!neg) v = -v;
              if (negel) { atomicAdd(sum, (unsigned long long)v & ((1ull << 52) - 1ull)); atomicAdd(sum + 1, (unsigned long long)(long long)(v >> 52)); }
            }
            if (!negel) atomicOr(&gflag[mt >> 8], 1u);
          } else if (hi == 0ull && lo < (1ull << 54)) atomicAdd(sum, neg ? 0ull - lo : lo);
          else {
            __int128 v = (__int128)(((unsigned __int128)hi << 64) | lo);
            if (neg) v = -v;
            atomicAdd(sum, (unsigned long long)v & ((1ull << 52) - 1ull));
            atomicAdd(sum + 1, (unsigned long long)(long long)(v >> 52));
          }
        }
      }
    }
  };
  // ---- truth-table runs of unit k: lane j of the workgroup takes run j; returns violated rows (bit i = row i of the run)
  auto run_bits = [&](const uint4 d, const uint32_t par, const bool mine) -> uint32_t {
    const unsigned long long *ones = ones0 + par * 18u;
    const uint32_t len = ((d.w >> 16) & 31u) + 1u, nops = (d.w >> 21) & 7u, strides = (d.w >> 24) & 31u;
    const uint32_t idx[5] = {d.y & 0xFFFFu, d.y >> 16, d.z & 0xFFFFu, d.z >> 16, d.w & 0xFFFFu};
    uint32_t x[5];
#pragma unroll
    for (int o = 0; o < 5; o++) {
      const uint32_t i = idx[o];
      const uint32_t v = cut32(i < T ? ones : xones, i < T ? i : i - T);
      x[o] = (uint32_t)o < nops ? ((strides >> o) & 1u ? v : 0u - (v & 1u)) : 0u;
    }
    const uint32_t holds = table32(d.x, x[0], x[1], x[2], x[3], x[4]);
    return mine ? ~holds & (len == 32u ? ~0u : (1u << len) - 1u) : 0u;
  };
  // ---- exports of unit k: lane j takes export j — element px of the tile to slot slot0 + j, and the bits of 64 of them as one word
  auto exports = [&](const Cursor c, const uint32_t par) {
    const uint32_t exp_n = TW(c, B3W_WT_EXP_N), slot0 = TW(c, B3W_WT_EXP_SLOT0);
    if (xwave * 64u >= exp_n) return;                          // (wave-uniform)
    const unsigned long long z = xtid < exp_n ? el0[par * T + px] : 0ull;
    xel[slot0 + xtid] = z;
    const unsigned long long is1 = __ballot(z == 1ull);
    if (lane == 0) xones[(slot0 >> 6) + xwave] = is1;
  };
  // ---- verdicts of the general rows of unit k (one iteration after its entries were added): row g -> lane g of the workgroup
  unsigned long long wrec[5] = {0, 0, 0, 0, 0};
  size_t wrec_at = 0;                                        // (1 + the record's place: 0 = this lane has none pending)
  // A violated row goes straight into its body's counters (LDS) INSIDE the branch that found it: a lane's "lowest violated row" is a
  // load (the row's constraint number), and a load whose result is still under way where the branches meet makes the compiler wait
  // there for everything in flight (s_waitcnt vmcnt(0)) — the NEXT unit's fetch too, in the verdict phase of every wave and every
  // iteration (the ISA had it until r05; inside the branch the wait is the violating wave's alone).
  auto report = [&](const uint32_t body, const uint32_t nrows, const uint32_t row_at) {
    atomicAdd(&cnt[body % 3u], nrows);
    atomicMin(&cnt[3u + body % 3u], W.row_id[row_at]);
  };
  auto verdicts = [&](const Cursor c, const uint32_t par, const bool careful) {
    const uint32_t gen_n = TW(c, B3W_WT_GEN_N);
    if (wave * 64u >= gen_n) return;                           // (wave-uniform)
    unsigned long long *sum = gsum0 + par * 6u * W.max_gen + 6u * tid;
    uint32_t *gflag = gflag0 + par * gr2, *gwide = gwide0 + par * gr2;
    bool defer = false, bad = false;
    if (tid < gen_n) {
      unsigned long long a_lo, b_lo, c_lo;
      long long a_hi, b_hi, c_hi;
      part_sum(sum[0], sum[1], a_lo, a_hi);
      part_sum(sum[2], sum[3], b_lo, b_hi);
      part_sum(sum[4], sum[5], c_lo, c_hi);
      const uint32_t gw = (B3W_WALK_ABLATE & 2048) ? 0u : gwide[tid];
      defer = gflag[tid] != 0u || (gw >> 24) > 1u || a_hi != ((long long)a_lo >> 63) || b_hi != ((long long)b_lo >> 63);
      bad = !defer && !gw && (a_lo * b_lo != c_lo || __mul64hi((long long)a_lo, (long long)b_lo) != c_hi);
#pragma unroll
      for (int q = 0; q < 6; q++) sum[q] = 0ull;
      gflag[tid] = 0u;
      if (gw) {
        gwide[tid] = 0u;
        // (the deferred kernel forms c - a b in 128 bits: |a b| < 2^126 because a and b passed for 64-bit numbers above, and c must
        // leave room too — nothing else bounds a foreign system's C part; a row whose c does not stays a deferred ROW)
        if (!defer && !careful && (c_hi < -(1ll << 61) || c_hi >= (1ll << 61))) defer = true;
        if (!defer && !careful) {                              // a wide record, while the body's list has room; else the row itself
          const uint32_t slot = atomicAdd(&cnt[6u + c.body % 3u], 1u);
          if (slot < B3W_WALK_WIDE_CAP) {                      // (kept in registers: the stores go out behind the pack — see the pipeline)
            wrec_at = ((size_t)c.body * B3W_WALK_WIDE_CAP + slot) * 5u + 1u;
            wrec[0] = a_lo; wrec[1] = b_lo; wrec[2] = c_lo; wrec[3] = (unsigned long long)c_hi;
            wrec[4] = (unsigned long long)(gw & 0xFFFFFFu) | (unsigned long long)TW(c, B3W_WT_SRC) << 32 | (unsigned long long)(TW(c, B3W_WT_ROW0) + tid) << 40;
          } else defer = true;
        }
      }
    }
    if (careful) return;                                       // (every row of this unit goes to the deferred kernel: nothing is counted here)
    const unsigned long long dm = __ballot(defer);
    if (dm != 0ull && lane == 0) dmask0[par * 8u + wave] = dm;
    if (bad) report(c.body, 1u, TW(c, B3W_WT_ROW0) + tid);
  };
  // ---- the scratch block of unit k for the deferred kernel, by wave 1 once every wave's verdicts of the unit are behind a barrier:
  // word 0 = which mask words follow (bit w = word w is stored and not zero); a careful unit: all its rows
  auto summary = [&](const Cursor c, const uint32_t par, const bool careful) {
    if (wave != 1u) return;
    const uint32_t nrows = TW(c, B3W_WT_NROWS), words = (nrows + 63u) >> 6;
    unsigned long long v = 0ull;
    if (lane < words) {
      if (lane < 8u) { v = dmask0[par * 8u + lane]; dmask0[par * 8u + lane] = 0ull; }
      // (the ALWAYS-deferred rows are not in the blocks: the deferred kernel takes them for every body from its own list)
      if (careful) v = (lane + 1u < words || !(nrows & 63u) ? ~0ull : (1ull << (nrows & 63u)) - 1ull) & ~lstat[c.tile * W.static_words + lane];
    }
    unsigned long long *block = scratch + ((size_t)c.body * W.ntiles + c.tile) * block_words;
    const unsigned long long head = __ballot(v != 0ull);
    if (v != 0ull) block[1u + lane] = v;
    if (lane == 0 && head) { block[0] = head; bflag[c.body % 3u] |= 1ull << c.tile; }      // (an unflagged tile's block is never read)
  };
  // ---- a body's result, once the verdicts of its last unit are behind a barrier: one lane stores what the workgroup counted
  auto flush = [&](const uint32_t body) {                     // (the lane that writes the scratch blocks: its body flags are complete)
    if (wave == 1u && lane == 0) {
      const uint32_t s = body % 3u;
      violations[body] = cnt[s];
      if (first) first[body] = cnt[3u + s];
      const uint32_t nwide = cnt[6u + s] < B3W_WALK_WIDE_CAP ? cnt[6u + s] : B3W_WALK_WIDE_CAP;
      body_flags[body] = bflag[s] | (unsigned long long)nwide << 56;
      cnt[s] = 0u; cnt[3u + s] = 0xFFFFFFFFu; cnt[6u + s] = 0u; bflag[s] = 0ull;
    }
  };
  // ---- the pipeline.  Iteration i: entries, runs, exports of unit i | verdicts of unit i - 1 | scratch block of unit i - 2 |
  // pack unit i + 1 | program of unit i + 1, fetch of unit i + 2 | barrier.
  Cursor cf{b0, 0, 0}, cp{b0, 0, 0}, ce{b0, 0, 0};            // fetch, pack, evaluate
  load_rec(cf);
  cp.rec = ce.rec = cf.rec;
  const Cursor last{b1 - 1u, W.ntiles - 1u, 0};
  auto step = [&](Cursor &c) { if (c.body != last.body || c.tile != last.tile) advance(c); };      // (stays on the last unit)
  fetch(cf); step(cf);                                        // unit 0
  program(cp);
  pack(cp, true, 0u, 0u); step(cp);
  fetch(cf); step(cf);                                        // unit 1
  lds_barrier();
  Cursor prev = ce, prev2 = ce;
  bool sticky = false, prev_careful = false, prev2_careful = false;
  uint32_t i3 = 0;
  for (uint32_t i = 0; i < m; i++) {
    unsigned long long t_unit = 0;
    if (STAMPS && stamping) { t_prev = __builtin_amdgcn_s_memtime(); t_unit = t_prev; }
    const uint32_t unit_tile = ce.tile;
    const uint32_t par = i & 1u;
    const uint32_t i3n = i3 == 2u ? 0u : i3 + 1u, i3nn = i3n == 2u ? 0u : i3n + 1u;
    const bool anomaly = __builtin_amdgcn_readfirstlane(lanom[i3]) != 0u;
    if (tid == 0) lanom[i3nn] = 0u;
    sticky = ce.tile == 0u ? anomaly : (sticky || anomaly);   // (an anomalous tile taints the rest of its body: later tiles import from it)
    if (i >= 2u) {
      if (!(B3W_WALK_ABLATE & 32)) summary(prev2, par, prev2_careful);
      if (prev2.tile == W.ntiles - 1u) flush(prev2.body);     // (its verdicts ran in the last iteration, behind the last barrier)
    }
    B3W_WSTAMP(0);
    if (!(B3W_WALK_ABLATE & 4)) entries(ce, par);
    B3W_WSTAMP(1);
    if (!(B3W_WALK_ABLATE & 1)) {
      const uint32_t run_n = TW(ce, B3W_WT_RUN_N);
      if (rwave * 64u < run_n) {
        const uint32_t viol = run_bits(prun, par, rtid < run_n);
        if (viol && !sticky) report(ce.body, (uint32_t)__popc(viol), W.run_row[TW(ce, B3W_WT_RUN_OFF) + rtid] + (uint32_t)__ffs((int)viol) - 1u);
      }
    }
    B3W_WSTAMP(2);
    if (!(B3W_WALK_ABLATE & 8)) exports(ce, par);
    B3W_WSTAMP(3);
    if (i && !(B3W_WALK_ABLATE & 2)) verdicts(prev, par ^ 1u, prev_careful);
    B3W_WSTAMP(4);
    pack(cp, i + 1u < m, par ^ 1u, i3n);
    B3W_WSTAMP(5);
    // (a wide record's stores go out HERE: behind the wait for the fetch the pack consumes, in front of the next loads — the compiler
    // cannot count conditional stores, so a wait that follows them is a wait for all of them)
    if (__ballot(wrec_at != 0) != 0ull) {
      if (wrec_at && !(B3W_WALK_ABLATE & 4096)) {
        unsigned long long *rec = wide_recs + (wrec_at - 1u);
#pragma unroll
        for (int q = 0; q < 5; q++) rec[q] = wrec[q];
      }
      if (B3W_WALK_ABLATE & 4096) asm volatile("" :: "v"(wrec[0]), "v"(wrec[1]), "v"(wrec[2]), "v"(wrec[3]), "v"(wrec[4]));
      wrec_at = 0;
    }
    program(cp);
    fetch(cf);
    prev2 = prev; prev2_careful = prev_careful;
    prev = ce; prev_careful = sticky;
    ce = cp; cp = cf;                                         // (the cursors follow one another: one table row read per unit)
    step(cf);
    i3 = i3n;
    B3W_WSTAMP(6);
    lds_barrier();
    B3W_WSTAMP(7);
    // (per tile: cycles of the iterations that evaluated it, barrier to barrier, by wave 0 — stamps[64 + tile]; [128 + tile]: how many)
    if (STAMPS && stamping && tid == 0) { stamps[64u + unit_tile] += __builtin_amdgcn_s_memtime() - t_unit; stamps[128u + unit_tile] += 1ull; }
  }
  if (STAMPS && stamping && lane == 0)
    for (int k = 0; k < 8; k++) stamps[wave * 8 + k] = ph[k];
  {
    verdicts(prev, (m - 1u) & 1u, prev_careful);
    if (wrec_at) {
      unsigned long long *rec = wide_recs + (wrec_at - 1u);
#pragma unroll
      for (int q = 0; q < 5; q++) rec[q] = wrec[q];
    }
  }
  lds_barrier();
  if (m >= 2u) { summary(prev2, m & 1u, prev2_careful); if (prev2.tile == W.ntiles - 1u) flush(prev2.body); }
  lds_barrier();
  summary(prev, (m - 1u) & 1u, prev_careful);
  flush(prev.body);
#undef B3W_WSTAMP
}
#undef TW

}  // namespace

// ---- WALK launch
static inline uint32_t walk_block_words(const B3wWalk *w) { return 2u + ((w->max_rows + 63u) >> 6); }
static inline size_t walk_smem(const B3wWalk *w) {
  const size_t xw = (w->exp_slots >> 6) + 1u, gr2 = (w->max_gen + 1u) & ~1u;
  return 8u * (2u * (size_t)B3W_R1CS_TILE + w->exp_slots + 36u + xw + 12u * (size_t)w->max_gen + 20u + ((w->ncoef + 1u) & ~1u) + 16u * (size_t)w->ntiles +
               (size_t)w->static_words * w->ntiles) +
         4u * (4u * gr2 + (size_t)w->ntiles * B3W_WT_WORDS + 4u + 9u + 8u) + 32u;
}
extern "C" size_t b3w_r1cs_walk_scratch_bytes(const B3wWalk *w) {      // blocks | body flags | wide records
  return ((size_t)B3W_R1CS_SLAB * w->ntiles * walk_block_words(w) + B3W_R1CS_SLAB + (size_t)B3W_R1CS_SLAB * B3W_WALK_WIDE_CAP * 5u) * 8;
}

extern "C" int b3w_launch_r1cs_walk(const uint8_t *d_bodies, uint32_t n, uint64_t pitch, const B3wWalk *walk, const B3wR1csSystem *sysw, const B3wField *field,
                                    unsigned long long *d_scratch, uint32_t *d_violations, uint32_t *d_first, hipStream_t stream) {
  if (!n || !walk->ntiles) return 0;
  if (!d_scratch) return -5;
  static const int env_grid = getenv("B3W_R1CS_GRID") ? atoi(getenv("B3W_R1CS_GRID")) : 0;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const uint32_t ne = (walk->max_ent + 511u) / 512u;         // entry chunks per wave
  if (ne > 4u || walk->ntiles > 56u) return -6;
  const bool sg = walk->signed_elems != 0u;
  const void *fn = sg ? (ne <= 1u ? reinterpret_cast<const void *>(&b3w_r1cs_walk_kernel<1, true>) : ne == 2u ? reinterpret_cast<const void *>(&b3w_r1cs_walk_kernel<2, true>)
                         : ne == 3u ? reinterpret_cast<const void *>(&b3w_r1cs_walk_kernel<3, true>) : reinterpret_cast<const void *>(&b3w_r1cs_walk_kernel<4, true>))
                      : (ne <= 1u ? reinterpret_cast<const void *>(&b3w_r1cs_walk_kernel<1, false>) : ne == 2u ? reinterpret_cast<const void *>(&b3w_r1cs_walk_kernel<2, false>)
                         : ne == 3u ? reinterpret_cast<const void *>(&b3w_r1cs_walk_kernel<3, false>) : reinterpret_cast<const void *>(&b3w_r1cs_walk_kernel<4, false>));
  unsigned long long *d_stamps = nullptr;
#ifdef B3W_R1CS_DIAG
  // diagnostic build only: B3W_R1CS_STAMPS=1 prints per-phase cycle sums of the middle workgroup after every launch (synchronises)
  static const bool print_stamps = getenv("B3W_R1CS_STAMPS") && atoi(getenv("B3W_R1CS_STAMPS"));
  static unsigned long long *d_stamps_buf = nullptr;
  if (print_stamps) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return (int)hipErrorStreamCaptureUnsupported; }
    if (!d_stamps_buf && hipMalloc((void **)&d_stamps_buf, 192 * 8) != hipSuccess) d_stamps_buf = nullptr;
    d_stamps = d_stamps_buf;
    if (d_stamps && hipMemsetAsync(d_stamps, 0, 192 * 8, stream) != hipSuccess) return (int)hipGetLastError();
    fn = ne <= 1u ? reinterpret_cast<const void *>(&b3w_r1cs_walk_kernel<1, false, true>) : reinterpret_cast<const void *>(&b3w_r1cs_walk_kernel<3, false, true>);
    if (ne == 2u || ne > 3u) return -6;
  }
#endif
  struct PerDevice { int cus = 0, lds = 0; bool attr[8] = {false, false, false, false, false, false, false, false}; };      // (per instantiation: NE x SIGNED)
  static PerDevice per[64];
  static std::mutex mu;
  const size_t smem = walk_smem(walk);
  const uint32_t inst = (ne ? ne - 1u : 0u) + (sg ? 4u : 0u);
  int cus = 0, lds = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    PerDevice &pd = per[dev & 63];
    if (!pd.cus) {
      if ((e = hipDeviceGetAttribute(&pd.cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return (int)e;
      if ((e = hipDeviceGetAttribute(&pd.lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev)) != hipSuccess) return (int)e;
      if (pd.lds < 160 * 1024) pd.lds = 64 * 1024;
    }
    if (smem > (size_t)pd.lds) return -6;
    if (!pd.attr[inst]) {
      if ((e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, pd.lds)) != hipSuccess) return (int)e;
      pd.attr[inst] = true;
    }
    cus = pd.cus; lds = pd.lds;
  }
  int wgs = (int)((size_t)lds / smem);
  if (wgs < 2) return -6;                                    // (one workgroup per CU reads at 4 TB/s: the stream kernel does better — the circomkit nova build, 425 general rows in one tile)
  if (wgs > 2) wgs = 2;
  const uint32_t bw = walk_block_words(walk);
  const B3wDeferredView view{sysw->ntiles, sysw->tiles, sysw->row_k, sysw->row_id, sysw->g_rows, sysw->g_wires, sysw->g_cids, sysw->coefs};
  for (uint32_t b0 = 0; b0 < n; b0 += B3W_R1CS_SLAB) {
    const uint32_t nb = n - b0 < B3W_R1CS_SLAB ? n - b0 : B3W_R1CS_SLAB;
    uint32_t grid = env_grid > 0 ? (uint32_t)env_grid : (uint32_t)(cus * wgs);
    if (grid > nb) grid = nb;                                // (whole bodies per workgroup)
    const uint8_t *bodies0 = d_bodies + (uint64_t)b0 * pitch;
    uint32_t *viol0 = d_violations + b0, *first0 = d_first ? d_first + b0 : nullptr;
    unsigned long long *flags = d_scratch + (size_t)B3W_R1CS_SLAB * walk->ntiles * bw, *wide = flags + B3W_R1CS_SLAB;
    void *args[] = {(void *)&bodies0, (void *)&pitch, (void *)&nb, (void *)walk, (void *)&d_scratch, (void *)&bw, (void *)&flags, (void *)&wide, (void *)&viol0, (void *)&first0, (void *)&d_stamps};
    e = hipLaunchKernel(fn, dim3(grid), dim3(512), args, smem, stream);
    if (e != hipSuccess) return (int)e;
    if (d_stamps) {
      unsigned long long h[192];
      if (hipStreamSynchronize(stream) == hipSuccess && hipMemcpy(h, d_stamps, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
        const double un = (double)nb / grid * walk->ntiles;
        static const char *name[8] = {"top+summary", "entries", "runs", "exports", "verdicts", "wait+pack", "program+fetch", "barrier"};
        fprintf(stderr, "b3w_r1cs_walk stamps (cycles per unit, workgroup %u of %u, %g units):\n", grid / 2, grid, un);
        for (int k = 0; k < 8; k++) {
          fprintf(stderr, "  %-15s", name[k]);
          for (int w = 0; w < 8; w++) fprintf(stderr, " %5.0f", (double)h[w * 8 + k] / un);
          fprintf(stderr, "\n");
        }
        fprintf(stderr, "  cycles per iteration by the tile it evaluated:");
        for (uint32_t t = 0; t < walk->ntiles && t < 64u; t++) fprintf(stderr, " %u:%.0f", t, h[128 + t] ? (double)h[64 + t] / (double)h[128 + t] : 0.0);
        fprintf(stderr, "\n");
      }
    }
    // (on a second stream beside the next slab's walk kernel — two halves of the scratch in turn — the deferred kernel made a check of
    // 65 536 nova bodies SLOWER, 8.4 -> 9.2 ms: HISTORY.md, round 4)
    hipLaunchKernelGGL(b3w_r1cs_walk_deferred_kernel, dim3(nb), dim3(64 * B3W_WALK_DEFERRED_WAVES), 0, stream, bodies0, pitch, nb, view, d_scratch, bw, flags, wide, walk->static_k,
                       make_uint4(walk->static_d0[0], walk->static_d0[1], walk->static_d0[2], walk->static_d0[3]), walk->static_id, walk->nstatic, *field,
                       viol0, first0);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
