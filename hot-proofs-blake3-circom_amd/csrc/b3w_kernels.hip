// b3w_kernels.hip — gfx950 (MI355X, CDNA4) witness kernels for the reference's BLAKE3 circom circuits.
//
// One 64-lane wavefront per workgroup computes W witnesses in two phases:
//
//   TRACE   4 lanes per witness run the BLAKE3 compression as the circuit arithmetises it
//           (column G's on lanes 0..3 of a quad, diagonals after a quad rotate), and park every
//           distinct signal VALUE ("atom", b3w_atoms.h) of the witness in a ~3.7 KB LDS image:
//           per half-G the 34/33-bit add-with-carry sums and the rotate-xor words.
//           HalfFunG: circuits/blake3_compression.circom:72-100; Bits34/Bits33/ToBits:
//           circuits/blake3_common.circom:142-203; SingleRound :128-161; permutation :15-26.
//   EXPAND  all 64 lanes stream the flat witness vector to HBM: a per-circuit slot table
//           (slot -> LDS word, shift, mode; built from layouts/*.layout) says which bit or word
//           of the image each 32-byte slot holds.  The body is written in 1 KiB tiles aligned to
//           128-byte lines in absolute addresses; a lane pair owns one slot of a tile and each lane
//           stores one 16-byte half, so every wave store instruction is eight whole lines
//           (global_store_dwordx4 x 64 lanes).  A wave takes W bodies that start at the same offset
//           into a line (WaveBodies), so one lane -> slot mapping serves them all.
//   VERIFY  (MODE 2) the same walk reading the bodies back and comparing (on-device consumer).
//
// The work is integer/bit expansion bound by HBM writes (770 976 B written for 112 B read per
// compression witness); there is no contraction, so no MFMA.  Where the bodies live in HBM matters
// as much as the kernel: see b3w_placement.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "b3w_atoms.h"
#include "b3w_kernels.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int r) { return __builtin_rotateright32(x, r); }

// lane j of every quad reads lane P[j] of its quad (DPP quad_perm, no LDS traffic)
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ uint32_t quad_perm(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, false);
}

// Message schedule: round r uses msg_r[j] = m[PERM_r[j]], PERM_0 = id, PERM_{r+1}[j] = PERM_r[sigma[j]]
// (Blake3Permute, circuits/blake3_common.circom:15-26, chained as in blake3_compression.circom:197-209).
// Packed 4 bits per entry.
__host__ __device__ constexpr uint64_t sched_pack(int r) {
  const int sigma[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
  int p[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  for (int i = 0; i < r; i++) {
    int q[16] = {};
    for (int j = 0; j < 16; j++) q[j] = p[sigma[j]];
    for (int j = 0; j < 16; j++) p[j] = q[j];
  }
  uint64_t v = 0;
  for (int j = 0; j < 16; j++) v |= (uint64_t)p[j] << (4 * j);
  return v;
}

// One HalfFunG on this lane's (a,b,c,d) with message word m; parks the 8 trace words of half-G k.
template <int R1, int R2>
__device__ __forceinline__ void half_g(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, uint32_t m,
                                       uint32_t *Lk) {
  const uint64_t s1 = (uint64_t)a + b + m;          // add1.inp (fits 34 bits)
  const uint32_t A = (uint32_t)s1;                  // add1.out_word
  const uint32_t D2 = rotr32(d ^ A, R1);            // rxor2.out_word
  const uint64_t s3 = (uint64_t)c + D2;             // add3.inp (fits 33 bits)
  const uint32_t C = (uint32_t)s3;                  // add3.out_word
  const uint32_t B4 = rotr32(b ^ C, R2);            // rxor4.out_word
  *reinterpret_cast<uint4 *>(Lk) = make_uint4(A, (uint32_t)(s1 >> 32), C, (uint32_t)(s3 >> 32));
  *reinterpret_cast<uint4 *>(Lk + 4) = make_uint4(D2, d, B4, b);
  a = A; d = D2; c = C; b = B4;
}

template <int R>
__device__ __forceinline__ void round_fn(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, const uint32_t *L,
                                         uint32_t *Lw, int col) {
  constexpr uint64_t P = sched_pack(R);
  const uint32_t *M = L + B3W_A_M;
  // columns: G_col on (col, 4+col, 8+col, 12+col), x = msg[2col], y = msg[2col+1]
  {
    const uint32_t mx = M[(P >> (8 * col)) & 15], my = M[(P >> (8 * col + 4)) & 15];
    uint32_t *Lk = Lw + B3W_LDS_HG + 8 * ((R * 8 + col) * 2);
    half_g<16, 12>(a, b, c, d, mx, Lk);
    half_g<8, 7>(a, b, c, d, my, Lk + 8);
  }
  // diagonals: G_{4+col} on (col, 4+(col+1)%4, 8+(col+2)%4, 12+(col+3)%4)
  b = quad_perm<1, 2, 3, 0>(b);
  c = quad_perm<2, 3, 0, 1>(c);
  d = quad_perm<3, 0, 1, 2>(d);
  {
    const uint32_t mx = M[(P >> (32 + 8 * col)) & 15], my = M[(P >> (36 + 8 * col)) & 15];
    uint32_t *Lk = Lw + B3W_LDS_HG + 8 * ((R * 8 + 4 + col) * 2);
    half_g<16, 12>(a, b, c, d, mx, Lk);
    half_g<8, 7>(a, b, c, d, my, Lk + 8);
  }
  b = quad_perm<3, 0, 1, 2>(b);
  c = quad_perm<2, 3, 0, 1>(c);
  d = quad_perm<1, 2, 3, 0>(d);
}

// Blake3Compression on the H/M/T/B/D atoms already parked in the image L (words 1..28).
// Called by the 4 lanes of a quad (col = lane & 3).  Writes half-G atoms and O[16].
__device__ __forceinline__ void trace_compression(uint32_t *L, int col, uint32_t *pub /* 16 words or null */) {
  const uint32_t IVc = col == 0 ? 0x6A09E667u : col == 1 ? 0xBB67AE85u : col == 2 ? 0x3C6EF372u : 0xA54FF53Au;
  uint32_t a = L[B3W_A_H + col], b = L[B3W_A_H + 4 + col], c = IVc, d = L[B3W_A_T + col];  // T0 T1 B D contiguous
  const uint32_t h_lo = a, h_hi = b;
  round_fn<0>(a, b, c, d, L, L, col);
  round_fn<1>(a, b, c, d, L, L, col);
  round_fn<2>(a, b, c, d, L, L, col);
  round_fn<3>(a, b, c, d, L, L, col);
  round_fn<4>(a, b, c, d, L, L, col);
  round_fn<5>(a, b, c, d, L, L, col);
  round_fn<6>(a, b, c, d, L, L, col);
  // out[k] = v[k]^v[k+8] (k<8), v[k]^h[k-8] (k>=8)   blake3_compression.circom:213-227
  const uint32_t o0 = a ^ c, o1 = b ^ d, o2 = c ^ h_lo, o3 = d ^ h_hi;
  L[B3W_A_O + col] = o0;
  L[B3W_A_O + 4 + col] = o1;
  L[B3W_A_O + 8 + col] = o2;
  L[B3W_A_O + 12 + col] = o3;
  if (col == 0) L[B3W_A_ONE] = 1u;
  if (pub) { pub[col] = o0; pub[4 + col] = o1; pub[8 + col] = o2; pub[12 + col] = o3; }
}

template <bool NT>
__device__ __forceinline__ void store16(uint8_t *p, uint4 v) {
  const u32x4 x = {v.x, v.y, v.z, v.w};
  if (NT) __builtin_nontemporal_store(x, reinterpret_cast<u32x4 *>(p));
  else *reinterpret_cast<u32x4 *>(p) = x;
}

// EXPAND: stream W images to HBM through the slot table.
//
// emit_group stores group g (32 slots, 1 KiB per witness) of every active witness of the wave.
// FULL = all W witnesses active and all 32 slots inside the body: no per-lane / per-witness tests.
template <int W, int WORDS, bool WIDE, bool NT, bool FULL>
__device__ __forceinline__ void emit_group(const uint32_t *lds, uint32_t e, uint8_t *__restrict__ out,
                                           const uint64_t (&woff)[W], uint32_t byte_off, bool in, uint32_t nact,
                                           const uint32_t *okmask) {
  const uint32_t par = threadIdx.x & 1;
  const uint32_t src = e & 0xFFFu, sh = (e >> 12) & 31u, mode = (e >> 17) & 3u;
  const uint32_t off = (WIDE && mode == B3W_MODE_W256) ? src + 4u * par : src;
  const bool live = (par == 0) || (WIDE && mode == B3W_MODE_W256);
  const uint32_t m0 = live ? (mode == B3W_MODE_BIT ? 1u : 0xFFFFFFFFu) : 0u;
  const uint32_t m1 = (live && mode >= B3W_MODE_W64) ? 0xFFFFFFFFu : 0u;
  const uint32_t m23 = (WIDE && mode == B3W_MODE_W256) ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int w = 0; w < W; ++w) {
    if (FULL || ((uint32_t)w < nact && (!okmask || okmask[w]))) {
      const uint32_t *L = lds + w * WORDS + off;
      uint4 v;
      v.x = (L[0] >> sh) & m0;
      v.y = L[1] & m1;
      if (WIDE) { v.z = L[2] & m23; v.w = L[3] & m23; } else { v.z = 0; v.w = 0; }
      if (FULL || in) store16<NT>(out + woff[w] + byte_off, v);
    }
  }
}

// Which witnesses a wave computes.  Bodies sit `pitch` bytes apart, so body i starts (i * pitch) mod 128 bytes into
// a 128-byte line; a wave takes W bodies `stride` apart with stride = 128 / gcd(pitch mod 128, 128) (1, 2 or 4), all
// starting at the SAME offset into a line, so that one shifted lane->slot mapping serves them all (expand below).
struct WaveBodies {
  uint32_t first, stride;
  __device__ __forceinline__ uint32_t operator()(int w) const { return first + (uint32_t)w * stride; }
};
template <int W>
__device__ __forceinline__ WaveBodies wave_bodies(uint32_t bid, uint32_t stride) {
  return WaveBodies{(bid / stride) * (stride * W) + bid % stride, stride};
}
// number of active witnesses of the wave (they form a prefix: the index grows with w)
template <int W>
__device__ __forceinline__ uint32_t wave_active(const WaveBodies &wb, uint32_t n) {
  uint32_t a = 0;
#pragma unroll
  for (int w = 0; w < W; ++w) a += wb(w) < n ? 1u : 0u;
  return a;
}

// U = groups per software-pipeline stage: the slot-table words of the next U groups are loaded
// before the U*W stores of the current ones are issued, so a wave waits on memory once per U KiB*W.
//
// Tiles are aligned to 128-byte lines in ABSOLUTE addresses: with j = (body start mod 128) / 32, tile k of a body
// covers its slots [32k - j, 32k - j + 32), so every wave store is eight whole lines (an unshifted 1 KiB store from
// a body that starts 32 bytes into a line touches nine, two of them partially: measured 6 % slower).  The table
// has 4 pad entries in front for the (masked) lanes of tile 0 that lie before the body.
template <int W, int WORDS, bool WIDE, bool NT>
__device__ __forceinline__ void expand(const uint32_t *lds, const uint32_t *__restrict__ table, uint32_t nwit,
                                       uint8_t *__restrict__ out, uint64_t pitch, const WaveBodies wb, uint32_t n,
                                       const uint32_t *okmask /* per-w LDS flags or null */, bool all_ok,
                                       uint32_t slice = 0, uint32_t slices = 1 /* this wave stores tiles [slice, slice + 1) * per */) {
  constexpr int U = 4;
  const int lane = threadIdx.x;
  const uint32_t nact = wave_active<W>(wb, n);
  const uint64_t start = reinterpret_cast<uint64_t>(out) + (uint64_t)wb.first * pitch;
  const uint32_t j = (start & 31) ? 0u : (uint32_t)(start >> 5) & 3u;     // 16-byte aligned buffers: unshifted
  const uint32_t ntiles = (nwit + j + 31) >> 5;
  const uint32_t kf0 = j ? 1u : 0u, kf1 = (nwit + j) >> 5;                // full tiles [kf0, kf1)
  uint64_t woff[W];                                    // byte offset of this lane's 16 B in tile k of witness w
#pragma unroll
  for (int w = 0; w < W; ++w) woff[w] = (uint64_t)wb(w) * pitch + (uint32_t)lane * 16u - 32u * j;
  const uint32_t *tp = table + (lane >> 1) - j;        // entry of this lane's slot in tile 0
  uint32_t k = 0, kend_all = ntiles;
  if (slices > 1) {                                    // SLICED launch: a body's tiles are shared out to `slices` waves, in fours
    const uint32_t per = ((ntiles + slices - 1) / slices + (U - 1)) & ~(uint32_t)(U - 1);
    k = slice * per < ntiles ? slice * per : ntiles;
    kend_all = k + per < ntiles ? k + per : ntiles;
#pragma unroll
    for (int w = 0; w < W; ++w) woff[w] += (uint64_t)k * 1024u;
  }
  const uint32_t kf1s = kf1 < kend_all ? kf1 : kend_all;
  auto ragged = [&](uint32_t kend) {
    for (; k < kend; ++k) {
      const uint32_t slot = k * 32 + (lane >> 1) - j;   // wraps above nwit for the lanes before the body
      emit_group<W, WORDS, WIDE, NT, false>(lds, tp[k * 32], out, woff, 0, slot < nwit, nact, okmask);
#pragma unroll
      for (int w = 0; w < W; ++w) woff[w] += 1024u;
    }
  };
  if (nact == (uint32_t)W && all_ok) {
    ragged(kf0 < kend_all ? kf0 : kend_all);
    uint32_t cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = tp[(k + u) * 32];   // table is padded by 2U groups past the last slot
    for (; k + U <= kf1s; k += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) nxt[u] = tp[(k + U + u) * 32];
#pragma unroll
      for (int u = 0; u < U; ++u) emit_group<W, WORDS, WIDE, NT, true>(lds, cur[u], out, woff, u * 1024u, true, nact, nullptr);
#pragma unroll
      for (int w = 0; w < W; ++w) woff[w] += U * 1024u;
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
  }
  ragged(kend_all);                                      // ragged waves, and the last (partial) tiles
}

// VERIFY (on-device consumer): instead of storing, read the body back and compare it with what its own
// input slots determine.  The circuits are deterministic — the inputs fix every signal — so "body equals
// the witness recomputed from the body's input slots" is the same statement as "every constraint holds".
// Returns this lane's count of differing 16-byte units per witness in cnt[].
template <int W, int WORDS, bool WIDE>
__device__ __forceinline__ void expand_verify(const uint32_t *lds, const uint32_t *__restrict__ table, uint32_t nwit,
                                              const uint8_t *__restrict__ bodies, uint64_t pitch, const WaveBodies wb, uint32_t n,
                                              uint32_t (&cnt)[W]) {
  constexpr int U = 4;                                   // tiles per iteration: U*W 16-byte loads in flight per lane
  const int lane = threadIdx.x;
  const uint32_t par = lane & 1;
  const uint32_t nact = wave_active<W>(wb, n);
  // line-aligned tiles exactly as in expand(): tile k = slots [32k - j, 32k - j + 32), j from the first body's start
  const uint64_t start = reinterpret_cast<uint64_t>(bodies) + (uint64_t)wb.first * pitch;
  const uint32_t j = (start & 31) ? 0u : (uint32_t)(start >> 5) & 3u;
  const uint32_t ntiles = (nwit + j + 31) >> 5;
#pragma unroll
  for (int w = 0; w < W; ++w) cnt[w] = 0;
  for (uint32_t k0 = 0; k0 < ntiles; k0 += U) {
    uint32_t e[U];
    u32x4 got[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t k = k0 + u < ntiles ? k0 + u : ntiles - 1;            // clamped: the tail re-reads the last tile
      const int32_t slot = (int32_t)(k * 32 + (lane >> 1)) - (int32_t)j;
      const uint32_t sl = slot < 0 ? 0u : (uint32_t)slot < nwit ? (uint32_t)slot : nwit - 1;   // clamped inside the body
      e[u] = table[sl];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const uint32_t body = wb((uint32_t)w < nact ? w : 0);              // clamped: inactive witnesses re-read body 0
        got[u][w] = *reinterpret_cast<const u32x4 *>(bodies + (uint64_t)body * pitch + (uint64_t)sl * 32 + par * 16);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int32_t slot = (int32_t)((k0 + u) * 32 + (lane >> 1)) - (int32_t)j;
      const bool in = k0 + u < ntiles && slot >= 0 && (uint32_t)slot < nwit;
      const uint32_t src = e[u] & 0xFFFu, sh = (e[u] >> 12) & 31u, mode = (e[u] >> 17) & 3u;
      const uint32_t off = (WIDE && mode == B3W_MODE_W256) ? src + 4u * par : src;
      const bool live = (par == 0) || (WIDE && mode == B3W_MODE_W256);
      const uint32_t m0 = live ? (mode == B3W_MODE_BIT ? 1u : 0xFFFFFFFFu) : 0u;
      const uint32_t m1 = (live && mode >= B3W_MODE_W64) ? 0xFFFFFFFFu : 0u;
      const uint32_t m23 = (WIDE && mode == B3W_MODE_W256) ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const uint32_t *L = lds + w * WORDS + off;
        const uint32_t x = (L[0] >> sh) & m0, y = L[1] & m1;
        const uint32_t z = WIDE ? (L[2] & m23) : 0u, t = WIDE ? (L[3] & m23) : 0u;
        const bool bad = ((got[u][w].x ^ x) | (got[u][w].y ^ y) | (got[u][w].z ^ z) | (got[u][w].w ^ t)) != 0;
        cnt[w] += (in && (uint32_t)w < nact && bad) ? 1u : 0u;
      }
    }
  }
}

// records for VERIFY mode come from the body's own input slots (in_slots[j] = slot holding record word j);
// a slot that is not a plain 32-bit value cannot be checked on this path: flagged in nc
__device__ __forceinline__ uint32_t body_input_word(const uint8_t *body, uint32_t slot, uint32_t &nc) {
  const uint4 *p = reinterpret_cast<const uint4 *>(body + (uint64_t)slot * 32);
  const uint4 lo = p[0], hi = p[1];
  if (lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) nc = 1;
  return lo.x;
}

// wave reduction of the per-lane counters, one global word per witness
template <int W>
__device__ __forceinline__ void publish_counts(uint32_t (&cnt)[W], uint32_t *__restrict__ mismatch, const WaveBodies wb, uint32_t n,
                                               const uint32_t *flags /* per-w: nonzero = not verifiable / rejected */) {
#pragma unroll
  for (int w = 0; w < W; ++w) {
    uint32_t c = cnt[w];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (threadIdx.x == 0 && wb(w) < n) mismatch[wb(w)] = flags[w] ? 0xFFFFFFFFu : c;
  }
}

// ------------------------------------------------------------------ two-kernel ("sweep") path
// The fused kernels above leave every wave streaming its own bodies: thousands of independent write
// streams whose instantaneous positions load the HBM channels unevenly.  Measured on MI355X
// (tools/ubench/store_patterns.hip, addr_sensitivity.py): 5.3-6.7 TB/s depending on where the output
// buffer happens to sit, while a linear sweep in which workgroup b of exactly 256 (one per CU) writes
// the 4 KiB tiles b, b+256, b+512, ... — the shape of the runtime's own fill kernel — holds 6.5-7.0
// TB/s on every buffer: at any instant the chip writes one contiguous 1 MiB window that covers all
// channels evenly.  The sweep path splits the work to get that shape:
//   TRACE kernel  = the trace phase of the fused kernels; the images go to an HBM scratch
//                   (3.7-11 KB per witness, ~1 % of the body), stored word-major ("transposed":
//                   word j of witness w at scr[j*C + w]) so that the words one window needs are
//                   spread over all L2 channels instead of sitting in two hot 4 KB images.
//   SWEEP kernel  = walks the output in fill order; every lane gathers the 1-4 image words of its
//                   16 bytes from the scratch (L2 resident) through a register-only software
//                   pipeline: slot-table words two steps ahead, image words one step ahead.

// TRACE kernel epilogue: scatter this wave's W images into the word-major scratch; row
// B3W_LDS_OKWORD holds 1 for a valid witness.  C = witnesses per scratch row.
template <int W, int WORDS>
__device__ __forceinline__ void dump_images(uint32_t *lds, uint32_t *__restrict__ scratch, uint32_t C, uint32_t wit0,
                                            uint32_t n, const uint32_t *okf) {
  const int lane = threadIdx.x;
  if (lane < W) lds[lane * WORDS + B3W_LDS_OKWORD] = okf ? okf[lane] : 1u;
  __syncthreads();
  constexpr int ROWS = 64 / W;                       // image words handled per wave instruction
  const int w = lane % W, jr = lane / W;
  if (wit0 + w < n) {
#pragma unroll 4
    for (int j = jr; j < WORDS; j += ROWS) scratch[(uint64_t)j * C + wit0 + w] = lds[w * WORDS + j];
  }
}

struct SweepPos {          // tile t of the sweep: first body it touches and the byte offset inside it
  uint32_t t;              // < 2^32 tiles (16 TB)
  int32_t rem;             // (t*4096 - lead) - w_lo*pitch; negative only inside the lead-in of tile 0
  uint32_t w_lo;
};

// wave-uniform by construction (derived from blockIdx and kernel arguments): pin to SGPRs
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// one lane's pipeline registers for a step of K tiles (one 32-byte slot per lane and tile), plus the
// step's wave-uniform tile positions, computed once and carried from the fetch stage to the emit stage
template <int K, int NW>
struct SweepRegs {
  uint32_t e[K]; uint32_t a[K][NW]; uint32_t ok[K];
  uint32_t t[K], wlo[K]; int32_t rem[K];     // uniform
  bool fast;                                  // uniform: every tile of the step is interior to one body
};

// Shape (measured, tools/ubench/store_shapes2/3.hip): exactly 256 workgroups, 4 KiB tiles, tile t of
// workgroup b = b + 256k; 128 lanes x one 32-byte slot each; PAIRS wave-pairs per workgroup, pair q owning a
// contiguous share (SPLIT) or every PAIRS-th (interleaved) of the workgroup's tile sequence.  One wave per
// SIMD has to issue everything, so the loop is kept lean: the whole slot table lives in LDS (one workgroup
// per CU, so the 96 KB are free), 32-bit position arithmetic in SGPRs (pitch < 2^30, scratch rows of 2^LOGC
// witnesses), a three-stage register pipeline (table words, image words one step later, emit one more step later), rare second/wide words
// fetched only by waves that hold such a slot, and an interior-tile fast path without per-lane position logic.
template <bool WIDE, int K, int LOGC, int PAIRS, bool SPLIT>
__global__ __launch_bounds__(128 * PAIRS, 1) void b3w_sweep_kernel(const uint32_t *__restrict__ scr, uint32_t n,
                                                                   uint8_t *__restrict__ out_base, uint32_t lead, uint32_t pitch,
                                                                   const uint32_t *__restrict__ table, uint32_t nwit) {
  constexpr int NW = WIDE ? 8 : 2;                       // image words a slot can need
  extern __shared__ __attribute__((aligned(16))) uint32_t tab[];   // the slot table, nwit words
  for (uint32_t i = threadIdx.x; i < nwit; i += 128 * PAIRS) tab[i] = table[i];
  const uint32_t tid = threadIdx.x % 128u, tid32 = tid * 32u, pair = uni(threadIdx.x / 128u);
  const uint32_t G = SPLIT ? gridDim.x : gridDim.x * PAIRS;   // tile stride of one wave-pair
  const uint32_t body = 32u * nwit;
  const uint64_t total = (uint64_t)lead + (uint64_t)n * pitch;
  const uint32_t ntiles_all = (uint32_t)((total + 4095) >> 12);
  // tiles of this workgroup: b, b+G0, ...; SPLIT: pair q owns sequence indices [q*share, (q+1)*share)
  const uint32_t G0 = gridDim.x;
  const uint32_t nseq = ntiles_all > blockIdx.x ? (ntiles_all - blockIdx.x + G0 - 1) / G0 : 0;
  const uint32_t share = (nseq + PAIRS - 1) / PAIRS;
  const uint32_t first_seq = SPLIT ? pair * share : pair;
  const uint32_t end_seq = SPLIT ? (first_seq + share < nseq ? first_seq + share : nseq) : nseq;
  const uint32_t ntiles = uni(end_seq > 0 ? blockIdx.x + (end_seq - 1) * G0 + 1 : 0);   // exclusive tile bound of this pair
  const uint64_t stride = (uint64_t)G * 4096;
  const uint32_t dq = (uint32_t)(stride / pitch), dr = (uint32_t)(stride % pitch);
  // running position of the fetch stage (wave-uniform)
  uint32_t pt = uni(blockIdx.x + G0 * first_seq), pw;
  int32_t prem;
  {
    const int64_t pos = (int64_t)((uint64_t)pt * 4096) - (int64_t)lead;
    if (pos < 0) { pw = 0; prem = (int32_t)pos; }
    else { pw = uni((uint32_t)((uint64_t)pos / pitch)); prem = (int32_t)uni((uint32_t)((uint64_t)pos % pitch)); }
  }
  __syncthreads();                                       // table in LDS
  // a lane's slot of a tile at (rem, w_lo): which witness, byte offset in its body, does it exist
  auto locate = [&](uint32_t t, int32_t rem, uint32_t w_lo, uint32_t &w, uint32_t &r32) {
    int32_t r = rem + (int32_t)tid32;
    w = w_lo;
    if (r >= (int32_t)pitch) { r -= (int32_t)pitch; w++; }
    r32 = (uint32_t)r;
    return t < ntiles && r32 < body && w < n;            // r < 0 wraps above body
  };
  // stage 1: positions of the step's K tiles and their slot-table words (LDS)
  auto do_table = [&](SweepRegs<K, NW> &rg) {
    bool fast = true;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      rg.t[k] = pt; rg.rem[k] = prem; rg.wlo[k] = pw;
      fast = fast && pt < ntiles && prem >= 0 && (uint32_t)prem + 4096u <= body && pw < n;
      const int32_t r = prem + (int32_t)dr;
      const bool c = r >= (int32_t)pitch;
      pt = uni(pt + G);
      pw = uni(pw + dq + (c ? 1u : 0u));
      prem = (int32_t)uni((uint32_t)(c ? r - (int32_t)pitch : r));
    }
    rg.fast = fast;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if (fast) {
        rg.e[k] = (tab + ((uint32_t)rg.rem[k] >> 5))[tid];     // uniform base + lane
      } else {
        uint32_t w, r32;
        const bool in = locate(rg.t[k], rg.rem[k], rg.wlo[k], w, r32);
        rg.e[k] = tab[in ? (r32 >> 5) : 0u];
      }
    }
  };
  // stage 2 (one step later, so the LDS latency is off the gather's critical path): image words from the
  // word-major scratch (1-2 cache lines per wave and word).  Loads are unconditional with clamped addresses
  // so the register sets stay in VGPRs.
  auto do_words = [&](SweepRegs<K, NW> &rg) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      uint32_t w = rg.wlo[k];
      if (!rg.fast) {
        uint32_t r32;
        const bool in = locate(rg.t[k], rg.rem[k], rg.wlo[k], w, r32);
        w = in ? w : 0u;
      }
      const uint32_t e = rg.e[k];
      const uint32_t row = (e & 0xFFFu) << LOGC, mode = (e >> 17) & 3u;
      const uint32_t *col = scr + w;                     // fast path: uniform base
      rg.a[k][0] = col[row];
      // second word (add1.inp / add3.inp whole values, ~1 % of the slots) and the 256-bit slots (IsZero inverses,
      // 0.3 %): fetched only by waves that hold such a slot
      if (__builtin_amdgcn_ballot_w64(mode >= B3W_MODE_W64)) {
#pragma unroll
        for (int x = 1; x < NW; ++x) rg.a[k][x] = col[row + ((uint32_t)x << LOGC)];
      } else {
#pragma unroll
        for (int x = 1; x < NW; ++x) rg.a[k][x] = 0;
      }
      rg.ok[k] = WIDE ? col[(uint32_t)B3W_LDS_OKWORD << LOGC] : 1u;   // only the nova circuits reject steps
    }
  };
  // stage 3: shape the 32 bytes and store them
  auto do_emit = [&](const SweepRegs<K, NW> &rg) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t e = rg.e[k];
      const uint32_t sh = (e >> 12) & 31u, mode = (e >> 17) & 3u;
      const uint32_t m0 = mode == B3W_MODE_BIT ? 1u : 0xFFFFFFFFu;
      const uint32_t m1 = mode >= B3W_MODE_W64 ? 0xFFFFFFFFu : 0u;
      uint4 lo, hi;
      lo.x = (rg.a[k][0] >> sh) & m0;
      lo.y = rg.a[k][1] & m1;
      if constexpr (WIDE) {
        const uint32_t m2 = mode == B3W_MODE_W256 ? 0xFFFFFFFFu : 0u;
        lo.z = rg.a[k][2] & m2; lo.w = rg.a[k][3] & m2;
        hi = make_uint4(rg.a[k][4] & m2, rg.a[k][5] & m2, rg.a[k][6] & m2, rg.a[k][7] & m2);
      } else {
        lo.z = 0; lo.w = 0;
        hi = make_uint4(0, 0, 0, 0);
      }
      bool in = true;
      if (!rg.fast) { uint32_t w, r32; in = locate(rg.t[k], rg.rem[k], rg.wlo[k], w, r32); }
      uint8_t *dst = out_base + (uint64_t)rg.t[k] * 4096;   // uniform base, lane offset tid*32
      if (in && (!WIDE || rg.ok[k] != 0)) {                // rejected step: body untouched
        store16<false>(dst + tid32, lo);
        store16<false>(dst + tid32 + 16, hi);
      }
    }
  };

  SweepRegs<K, NW> r0, r1, r2;
  do_table(r0);                                        // step 0
  do_table(r1);                                        // step 1
  do_words(r0);                                        // step 0
  // step s: table words of step s+2, image words of step s+1, emit step s
  while (true) {
    if (r0.t[0] >= ntiles) break;
    do_table(r2); do_words(r1); do_emit(r0);
    if (r1.t[0] >= ntiles) break;
    do_table(r0); do_words(r2); do_emit(r1);
    if (r2.t[0] >= ntiles) break;
    do_table(r1); do_words(r0); do_emit(r2);
  }
}

// ------------------------------------------------------------------ compression circuit
// MODE 0: witnesses -> bodies (fused);  1: images -> scratch (TRACE kernel of the sweep path);
//      2: VERIFY — recs = in_slots table, out = bodies to check, pub = per-witness mismatch counts
// SL (MODE 0 only): SLICED launch for small batches — `slices` consecutive workgroups take the same W bodies, each recomputes
// their (cheap) traces and stores 1/slices of the tiles.  A lone body streams at 13 GB/s per wave, so a batch of a few hundred
// bodies is latency-bound with one wave per body; sliced, it has slices x as many store streams in flight.
// PERSIST (MODE 0): a FIXED grid (gridDim.x a multiple of the stride) whose waves take the groups of W bodies in turn — however large the
// batch, the same number of body streams is in flight (used by the nova O2 kernel for very large batches, see b3w_launch_batch).
template <int W, bool NT, int MODE, bool SL = false, bool PERSIST = false>
__global__ __launch_bounds__(64) void b3w_compression_kernel(const uint32_t *__restrict__ recs, uint32_t n,
                                                             uint8_t *__restrict__ out, uint64_t pitch,
                                                             const uint32_t *__restrict__ table, uint32_t nwit,
                                                             uint32_t *__restrict__ pub, int32_t *__restrict__ status,
                                                             uint32_t stride /* WaveBodies; 1 for MODE 1 */, uint32_t slices = 1) {
  constexpr int WORDS = B3W_LDS_WORDS_COMP;
  __shared__ __attribute__((aligned(16))) uint32_t lds[W * WORDS + 4];   // +4: expand reads src+1 unconditionally
  const int lane = threadIdx.x;
  const uint32_t slice = SL ? blockIdx.x % slices : 0u;
  if (SL && slice != 0) { pub = nullptr; status = nullptr; }             // slice 0 reports for the body
 for (uint32_t bid = blockIdx.x;; bid += gridDim.x) {
  const WaveBodies wb = wave_bodies<W>(SL ? bid / slices : bid, stride);
  if (wb.first >= n) return;                           // the grid is rounded up to whole stride groups: nothing (more) for this wave
  const uint32_t wit0 = wb.first;                      // MODE 1 runs with stride 1
  // stage the 28-word records of this wave's witnesses into atoms H M T B D (image words 1..28)
  __shared__ uint32_t ncf[W];      // VERIFY: an input slot of the body is not a plain 32-bit value
  if (lane < W) ncf[lane] = 0;
  if (MODE == 2) __syncthreads();
  for (int i = lane; i < W * 28; i += 64) {
    const int w = i / 28, j = i - w * 28;
    if (wb(w) < n) {
      uint32_t v;
      if (MODE == 2) {
        uint32_t nc = 0;
        v = body_input_word(out + (uint64_t)wb(w) * pitch, recs[j], nc);
        if (nc) ncf[w] = 1;
      } else {
        v = recs[(uint64_t)wb(w) * 28 + j];
      }
      lds[w * WORDS + B3W_A_H + j] = v;
    }
  }
  __syncthreads();
  {
    const int w = lane >> 2, col = lane & 3;
    if (w < W && wb(w) < n) {
      trace_compression(lds + w * WORDS, col, (MODE != 2 && pub) ? pub + (uint64_t)wb(w) * 16 : nullptr);
      if (MODE != 2 && status && col == 0) status[wb(w)] = 0;     // canonical u32 inputs cannot fail an assert
    }
  }
  __syncthreads();
  if (MODE == 1) dump_images<W, WORDS>(lds, reinterpret_cast<uint32_t *>(out), (uint32_t)pitch, wit0, n, nullptr);   // out = scratch, pitch = its row length
  else if (MODE == 2) {
    uint32_t cnt[W];
    expand_verify<W, WORDS, false>(lds, table, nwit, out, pitch, wb, n, cnt);
    publish_counts<W>(cnt, pub, wb, n, ncf);
  } else if (SL) expand<W, WORDS, false, NT>(lds, table, nwit, out, pitch, wb, n, nullptr, true, slice, slices);
  else expand<W, WORDS, false, NT>(lds, table, nwit, out, pitch, wb, n, nullptr, true);
  if (!PERSIST) return;
  __syncthreads();                                     // the images are overwritten by the next group's
 }
}


// ------------------------------------------------------------------ nova step circuit
// Blake3Nova(0) (circuits/blake3_nova.circom:169-267) = control logic over small integers + one
// Blake3Compression.  For canonical u32 inputs every signal is a bit or a word EXCEPT the IsZero
// inverses (circomlib IsZero: inv <-- in!=0 ? 1/in : 0): 67 per step, all of small signed integers
//   -depth, -block_count, n_blocks-1-block_count, total_depth-i-2-depth (i<64),
// and, in the circomkit (O1) build, the negative differences themselves (p - |k|).

// d_aux: words [0,8) prime (little-endian limbs), [8] TABLE_N, [16 + 8k, +8) = k^-1 mod p, k < TABLE_N
#define B3W_AUX_TABLE 16

struct U256 { uint32_t l[8]; };

__device__ __forceinline__ U256 u256_sub(const U256 &a, const U256 &b) {
  U256 r;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t d = (uint64_t)a.l[i] - b.l[i] - br;
    r.l[i] = (uint32_t)d;
    br = (uint32_t)(d >> 63);
  }
  return r;
}

__device__ __forceinline__ U256 u256_small(uint64_t x) {
  U256 r;
  r.l[0] = (uint32_t)x; r.l[1] = (uint32_t)(x >> 32);
#pragma unroll
  for (int i = 2; i < 8; ++i) r.l[i] = 0;
  return r;
}

// k mod p for a small signed integer
__device__ __forceinline__ U256 u256_signed(int64_t k, const U256 &P) {
  return k >= 0 ? u256_small((uint64_t)k) : u256_sub(P, u256_small((uint64_t)(-k)));
}

// 1/k mod p for 1 <= k < 2^32 without field arithmetic: with t = -p^-1 mod k, (p*t + 1)/k is an
// exact quotient below p whose product with k is 1 mod p.
__device__ __forceinline__ U256 inv_small_general(uint32_t k, const U256 &P) {
  uint64_t r = 0;
#pragma unroll
  for (int i = 7; i >= 0; --i) r = ((r << 32) | P.l[i]) % k;      // r = p mod k
  // x = r^-1 mod k (extended Euclid; gcd(r,k) = 1 because p is prime and k < p)
  int64_t x0 = 0, x1 = 1;
  uint64_t a = k, b = r;
  while (b > 1) {
    const uint64_t q = a / b, tt = a - q * b;
    a = b; b = tt;
    const int64_t tx = x0 - (int64_t)q * x1;
    x0 = x1; x1 = tx;
  }
  int64_t x = x1 % (int64_t)k;
  if (x < 0) x += k;
  const uint64_t t = ((uint64_t)k - (uint64_t)x) % k;              // p*t == -1 (mod k)
  uint32_t prod[9];
  uint64_t carry = 1;                                               // the "+ 1"
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t cur = (uint64_t)P.l[i] * t + carry;
    prod[i] = (uint32_t)cur;
    carry = cur >> 32;
  }
  prod[8] = (uint32_t)carry;
  U256 out;
  uint64_t rem = 0;
#pragma unroll
  for (int i = 8; i >= 0; --i) {
    const uint64_t cur = (rem << 32) | prod[i];
    const uint64_t qd = cur / k;
    rem = cur - qd * k;
    if (i < 8) out.l[i] = (uint32_t)qd;
  }
  return out;
}

// inv <-- k != 0 ? 1/k : 0 in the field, k a small signed integer.  false = |k| outside the supported range.
__device__ __forceinline__ bool inv_signed(int64_t k, const U256 &P, const uint32_t *__restrict__ aux, U256 &out) {
  const uint64_t mag = k < 0 ? (uint64_t)(-k) : (uint64_t)k;
  if (mag == 0) { out = u256_small(0); return true; }
  if (mag >> 32) { out = u256_small(0); return false; }
  const uint32_t tn = aux[8];
  U256 v;
  if (mag < tn) {
    const uint4 *tp = reinterpret_cast<const uint4 *>(aux + B3W_AUX_TABLE + 8 * mag);
    const uint4 lo = tp[0], hi = tp[1];
    v.l[0] = lo.x; v.l[1] = lo.y; v.l[2] = lo.z; v.l[3] = lo.w;
    v.l[4] = hi.x; v.l[5] = hi.y; v.l[6] = hi.z; v.l[7] = hi.w;
  } else {
    v = inv_small_general((uint32_t)mag, P);
  }
  out = k < 0 ? u256_sub(P, v) : v;
  return true;
}

__device__ __forceinline__ void lds_put256(uint32_t *L, const U256 &v) {
  *reinterpret_cast<uint4 *>(L) = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  *reinterpret_cast<uint4 *>(L + 4) = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// One IsZero gadget of a step (job j of 67) on the image L whose record is staged at B3W_LDS_NV: the gadget's flag, and — INV — its
// inverse (and, for the unsimplified build, the differences themselves) as field elements into the image's wide part.  INV = false is the
// fill-ordered kernel's tracer: its images carry no wide part (the 67 256-bit slots are written by another launch).
// -> false: |k| outside the supported range.
template <bool O1, bool INV>
__device__ __forceinline__ bool nova_iszero_job(uint32_t *L, int j, const U256 &P, const uint32_t *__restrict__ aux) {
  const uint32_t *in = L + B3W_LDS_NV;
  const int64_t depth = in[NV_DEPTH];
  int64_t k, in1 = 0;
  uint32_t flag_atom, isz_wide = 0, in1_wide = 0;
  if (j == 0) { k = -depth; flag_atom = NV_IS_ROOT; isz_wide = NV_ROOT_ISZ_IN; }                     // check_root (:19-23)
  else if (j == 1) { k = -(int64_t)in[NV_BLOCK_COUNT]; flag_atom = NV_E0; isz_wide = NV_E0_ISZ_IN; }  // check_block_counts[0] (:136-141)
  else if (j == 2) { in1 = (int64_t)in[NV_N_BLOCKS] - 1; k = in1 - (int64_t)in[NV_BLOCK_COUNT];       // check_block_counts[1] (:142-144)
                     flag_atom = NV_E1; isz_wide = NV_E1_ISZ_IN; in1_wide = NV_E1_IN1; }
  else { const int i = j - 3; in1 = (int64_t)in[NV_TOTAL_DEPTH] - i - 2; k = in1 - depth;              // eqs[i] (:65-72)
         flag_atom = NV_EQ_OUT + i; isz_wide = NV_EQ_ISZ_IN + i; in1_wide = NV_EQ_IN1 + i; }
  L[B3W_LDS_NV + flag_atom] = (k == 0) ? 1u : 0u;
  if (!INV) return ((uint64_t)(k < 0 ? -k : k) >> 32) == 0;
  U256 inv;
  const bool ok = inv_signed(k, P, aux, inv);
  if (!O1) {
    lds_put256(L + B3W_LDS_WIDE + 8 * j, inv);                 // O2 keeps only the inverses: wide index = job index
  } else {
    const uint32_t inv_wide = j < 3 ? (uint32_t)j : (uint32_t)(NV_EQ_INV - NV_NARROW_COUNT + (j - 3));
    lds_put256(L + B3W_LDS_WIDE + 8 * inv_wide, inv);
    lds_put256(L + B3W_LDS_WIDE + 8 * (isz_wide - NV_NARROW_COUNT), u256_signed(k, P));
    if (in1_wide) lds_put256(L + B3W_LDS_WIDE + 8 * (in1_wide - NV_NARROW_COUNT), u256_signed(in1, P));
  }
  return ok;
}

// The FLAGS of the 67 IsZero gadgets in closed form, by the four lanes of the step's quad (the fill-ordered kernel's tracer: no inverses,
// and 67 lane-jobs with four LDS reads each cost it more than the compression trace).  Same values as nova_iszero_job<*, false> for every
// j (tests: every body of the fill-ordered path is compared with the body-stream kernels', which run the jobs).
// -> true: an IsZero argument is out of the supported range (|k| >= 2^32).
__device__ __forceinline__ bool nova_iszero_flags_quad(uint32_t *L, int col) {
  uint32_t *nv = L + B3W_LDS_NV;
  const int64_t depth = nv[NV_DEPTH], block_count = nv[NV_BLOCK_COUNT], n_blocks = nv[NV_N_BLOCKS], total_depth = nv[NV_TOTAL_DEPTH];
  if (col == 0) {
    nv[NV_IS_ROOT] = depth == 0 ? 1u : 0u;                               // k = -depth
    nv[NV_E0] = block_count == 0 ? 1u : 0u;                              // k = -block_count
    nv[NV_E1] = n_blocks - 1 - block_count == 0 ? 1u : 0u;               // k = n_blocks - 1 - block_count
  }
  const int64_t istar = total_depth - 2 - depth;                         // eqs[i]: k = total_depth - i - 2 - depth = istar - i
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = col * 16 + q;
    nv[NV_EQ_OUT + i] = istar == i ? 1u : 0u;
  }
  const int64_t k2 = n_blocks - 1 - block_count;
  const int64_t mag2 = k2 < 0 ? -k2 : k2, magl = istar < 0 ? -istar : istar, magh = istar - 63 < 0 ? 63 - istar : istar - 63;
  return (mag2 >> 32) != 0 || (magl >> 32) != 0 || (magh >> 32) != 0;    // (|istar - i| is largest at i = 0 or i = 63)
}

// Flags, message / chaining-value selection of a step: the four lanes of a quad (col) on the step's image L, after its IsZero jobs.
// -> the step's status (0, 4 = the circuit's assert, 103 = outside the kernels' domain); the inputs of the compression are in place.
__device__ __forceinline__ int32_t nova_select(uint32_t *L, int col, bool dom_a /* an IsZero argument was out of range */) {
  uint32_t *nv = L + B3W_LDS_NV;
  const uint32_t n_blocks = nv[NV_N_BLOCKS], block_count = nv[NV_BLOCK_COUNT], cil = nv[NV_CIL], cih = nv[NV_CIH];
  const uint32_t leaf_depth = nv[NV_LEAF_DEPTH], total_depth = nv[NV_TOTAL_DEPTH], depth = nv[NV_DEPTH];
  // Blake3NovaTreePath_CheckDepth (:13-45): LessThan(8) / GreaterEqThan(8) through Num2Bits(9)
  const int64_t cp = (int64_t)depth + 257 - (int64_t)leaf_depth;       // check_parent.n2b.in
  const int64_t ed = (int64_t)leaf_depth + 255 - (int64_t)depth;       // exceed_depth.lt.n2b.in
  const bool assert_fail = cp < 0 || cp >= 512 || ed < 0 || ed >= 512 || ((ed >> 8) & 1) == 0;
  const uint32_t parent = 1u - (uint32_t)((cp >> 8) & 1);
  const uint32_t is_root = depth == 0 ? 1u : 0u;
  const uint32_t e0 = block_count == 0 ? 1u : 0u;
  const uint32_t e1 = ((int64_t)n_blocks - 1 == (int64_t)block_count) ? 1u : 0u;
  const bool dom = dom_a || depth == 0xFFFFFFFFu || (block_count == 0xFFFFFFFFu && !parent);
  const int32_t st = assert_fail ? 4 : dom ? 103 : 0;
  // Blake3GetFlag (:122-167)
  const uint32_t last = e1 & (1u - parent), first = e0 & (1u - parent);
  const uint32_t ur_tmp = parent | e1, ur_flag = ur_tmp & is_root;
  const uint32_t dflag = first + 2u * last + 8u * ur_flag + 4u * parent;
  // Blake3GetDownLeftPath (:47-84): eqs[i].out = [i == istar]; bit_at_depth is a running sum
  const int64_t istar = (int64_t)total_depth - 2 - (int64_t)depth;
  const uint64_t chunk_idx = (uint64_t)cil | ((uint64_t)cih << 32);
  const bool has_star = istar >= 0 && istar < 64;
  const uint32_t nb_star = has_star ? 1u - (uint32_t)((chunk_idx >> (istar & 63)) & 1) : 0u;
  const uint32_t dl = parent ? nb_star : 1u;                            // (1-parent) + parent*bit_at_depth[63]
  const uint32_t cdd = last | parent, decr = cdd & (1u - is_root);
  if (col == 0) {
    nv[NV_BLOCK_COUNT_OUT] = block_count + (1u - parent);               // :251
    nv[NV_DEPTH_OUT] = depth - decr;                                    // :262
    nv[NV_IS_PARENT] = parent;
    nv[NV_CP_IN1] = leaf_depth - 1u;
    nv[NV_CP_N2B_IN] = (uint32_t)cp;
    nv[NV_ED_IN1] = depth + 1u;
    nv[NV_ED_N2B_IN] = (uint32_t)ed;
    nv[NV_ED_OUT] = 0u;
    nv[NV_NOT_ROOT] = 1u - is_root;
    nv[NV_NOT_PARENT] = 1u - parent;
    nv[NV_IS_LAST_BLOCK] = last;
    nv[NV_FIRST] = first;
    nv[NV_UR_TMP] = ur_tmp;
    nv[NV_UR_FLAG] = ur_flag;
    nv[NV_DL] = dl;
    nv[NV_CDD_OUT] = cdd;
    nv[NV_DECR_DEPTH] = decr;
    L[B3W_LDS_CHUNK_IDX] = cil;
    L[B3W_LDS_CHUNK_IDX + 1] = cih;
    L[B3W_A_T] = parent ? 0u : cil;                                     // :244-245
    L[B3W_A_T + 1] = parent ? 0u : cih;
    L[B3W_A_B] = nv[NV_B];
    L[B3W_A_D] = dflag;
  }
  // Blake3GetFinal_m (:86-120) and h_compression (:229-233): products of a word and a bit
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = col + 4 * q;
    const uint32_t hsel = nv[NV_H + (i & 7)], msel = nv[NV_M + (i & 7)], mi = nv[NV_M + i];
    const uint32_t take_h = i < 8 ? dl : 1u - dl;                        // tmp_down = h * (dl | 1-dl)
    const uint32_t td = take_h ? hsel : 0u;
    const uint32_t mp = (take_h ? 0u : msel) + td;                       // m_is_parent
    const uint32_t tp = parent ? mp : 0u;
    nv[NV_TMP_DOWN + i] = td;
    nv[NV_M_IS_PARENT + i] = mp;
    nv[NV_TMP_IS_PAR + i] = tp;
    L[B3W_A_M + i] = (parent ? 0u : mi) + tp;                            // out_m
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = col + 4 * q;
    const uint32_t iv = i == 0 ? 0x6A09E667u : i == 1 ? 0xBB67AE85u : i == 2 ? 0x3C6EF372u : i == 3 ? 0xA54FF53Au
                      : i == 4 ? 0x510E527Fu : i == 5 ? 0x9B05688Cu : i == 6 ? 0x1F83D9ABu : 0x5BE0CD19u;
    const uint32_t tiv = parent ? iv : 0u;
    nv[NV_TMPIV + i] = tiv;
    L[B3W_A_H + i] = (parent ? 0u : nv[NV_H + i]) + tiv;                 // h_compression
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = col * 16 + q;
    nv[NV_BIT_AT_DEPTH + i] = (has_star && (int64_t)i >= istar) ? nb_star : 0u;
  }
  return st;
}

// MODE 3 (O2): the lines of every body that hold its 67 256-bit slots only — the field inverses of the IsZero gadgets, their slot numbers
// behind the inverse table in `aux` — beside a launch of the fill-ordered kernel, whose images carry no wide part and which leaves exactly
// those lines alone; rejected steps are left alone as everywhere.
template <int KIND, int W, bool NT, int MODE, bool SL = false, bool PERSIST = false>
__global__ __launch_bounds__(64) void b3w_nova_kernel(const uint32_t *__restrict__ recs, uint32_t n,
                                                      uint8_t *__restrict__ out, uint64_t pitch,
                                                      const uint32_t *__restrict__ table, uint32_t nwit,
                                                      uint32_t *__restrict__ pub, int32_t *__restrict__ status,
                                                      const uint32_t *__restrict__ aux, uint32_t stride, uint32_t slices = 1) {
  constexpr bool O1 = KIND == B3W_KIND_NOVA_O1;
  constexpr int WORDS = O1 ? B3W_LDS_WORDS_NOVA_O1 : B3W_LDS_WORDS_NOVA_O2;
  __shared__ __attribute__((aligned(16))) uint32_t lds[W * WORDS + 4];
  __shared__ uint32_t okf[W];      // 1 = witness computed, stream it out
  __shared__ uint32_t domf[W];     // 1 = an IsZero argument fell outside the supported magnitude
  const int lane = threadIdx.x;
  const uint32_t slice = SL ? blockIdx.x % slices : 0u;                  // SLICED launch: see b3w_compression_kernel
  if (SL && slice != 0) { pub = nullptr; status = nullptr; }
  __shared__ uint32_t ncf[W];      // VERIFY: an input slot of the body is not a plain 32-bit value
  U256 P;
#pragma unroll
  for (int i = 0; i < 8; ++i) P.l[i] = aux[i];
 for (uint32_t bid = blockIdx.x;; bid += gridDim.x) {                    // PERSIST: see b3w_compression_kernel
  const WaveBodies wb = wave_bodies<W>(SL ? bid / slices : bid, stride);
  if (wb.first >= n) return;                           // the grid is rounded up to whole stride groups: nothing (more) for this wave
  const uint32_t wit0 = wb.first;                      // MODE 1 runs with stride 1
  if (lane < W) { okf[lane] = 0; domf[lane] = 0; ncf[lane] = 0; }
  if (MODE == 2) __syncthreads();
  for (int i = lane; i < W * 32; i += 64) {
    const int w = i >> 5, j = i & 31;
    if (wb(w) < n) {
      uint32_t v;
      if (MODE == 2) {
        uint32_t nc = 0;
        v = body_input_word(out + (uint64_t)wb(w) * pitch, recs[j], nc);
        if (nc) ncf[w] = 1;
      } else {
        v = recs[(uint64_t)wb(w) * 32 + j];
      }
      lds[w * WORDS + B3W_LDS_NV + j] = v;
    }
  }
  __syncthreads();

  // ---- A: the 67 IsZero gadgets of each step, one per lane-job
  for (int t = lane; t < 67 * W; t += 64) {
    const int w = t / 67, j = t - w * 67;
    if (wb(w) >= n) continue;
    if (!nova_iszero_job<O1, true>(lds + w * WORDS, j, P, aux)) domf[w] = 1;
  }
  __syncthreads();

  // ---- B: flags, message / chaining-value selection; 4 lanes per witness
  const int w = lane >> 2, col = lane & 3;
  const bool active = w < W && wb(w) < n;
  uint32_t *L = lds + (active ? w : 0) * WORDS;
  if (active) {
    const int32_t st = nova_select(L, col, domf[w] != 0);
    if (col == 0) {
      okf[w] = st == 0 ? 1u : 0u;
      if (MODE != 2 && status) status[wb(w)] = st;
    }
  }
  __syncthreads();
  if (MODE == 3) {
    // the 128-byte lines (absolute addresses) that hold one of the body's 67 256-bit slots, whole — as far as they lie in the body —:
    // lane pair = one slot of such a line, each lane a 16-byte half.  What stands beside the inverses are select flags and
    // step inputs (b3w_ctx.cpp checks it: nothing of the compression trace, which this launch does not run).
    // The host lists them per offset of a body in a line (aux: B3W_AUX_LINE_LISTS): all of a body's list entries are loaded before its
    // first store — a load behind a store comes back behind it.
    const uint2 *lists = reinterpret_cast<const uint2 *>(aux + B3W_AUX_LINE_LISTS);
    const uint32_t half = (uint32_t)lane & 1u;
#pragma unroll
    for (int ww = 0; ww < W; ++ww) {
      if (wb(ww) < n && okf[ww]) {
        uint8_t *base = out + (uint64_t)wb(ww) * pitch;
        const uint32_t ph = (uint32_t)(reinterpret_cast<uint64_t>(base) >> 5) & 3u;
        const uint32_t cnt = aux[B3W_AUX_LINE_COUNTS + ph];
        constexpr int IT = B3W_LINE_LIST_MAX / 32;
        uint2 it[IT];
#pragma unroll
        for (int u = 0; u < IT; ++u) {
          const uint32_t idx = ((uint32_t)lane >> 1) + 32u * u;
          it[u] = idx < cnt ? lists[ph * B3W_LINE_LIST_MAX + idx] : make_uint2(0xFFFFFFFFu, 0u);
        }
#pragma unroll
        for (int u = 0; u < IT; ++u) {
          if (it[u].x < nwit) {
            const uint32_t e = it[u].y, src = e & 0xFFFu, sh = (e >> 12) & 31u, mode = (e >> 17) & 3u;
            const uint32_t *L = lds + ww * WORDS + src + (mode == B3W_MODE_W256 ? 4u * half : 0u);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (mode == B3W_MODE_W256) v = make_uint4(L[0], L[1], L[2], L[3]);
            else if (!half) { v.x = mode == B3W_MODE_BIT ? (L[0] >> sh) & 1u : L[0]; v.y = mode == B3W_MODE_W64 ? L[1] : 0u; }
            store16<false>(base + (uint64_t)it[u].x * 32 + 16 * half, v);
          }
        }
      }
    }
    if (!PERSIST) return;
    __syncthreads();
    continue;
  }
  if (active && okf[w]) trace_compression(L, col, nullptr);
  __syncthreads();
  bool all_ok = true;
#pragma unroll
  for (int x = 0; x < W; ++x) all_ok = all_ok && (okf[x] != 0);
  if (MODE != 2 && active && okf[w] && pub) {
    // w[1..15]: n_blocks_out block_count_out h_out[8] total_depth_out depth_out chunk_idx_low_out chunk_idx_high_out leaf_depth_out
    uint32_t *pw = pub + (uint64_t)wb(w) * 15;
    const uint32_t *nv = L + B3W_LDS_NV;
    if (col == 0) { pw[0] = nv[NV_N_BLOCKS]; pw[1] = nv[NV_BLOCK_COUNT_OUT]; pw[10] = nv[NV_TOTAL_DEPTH]; pw[11] = nv[NV_DEPTH_OUT]; }
    if (col == 1) { pw[12] = nv[NV_CIL]; pw[13] = nv[NV_CIH]; pw[14] = nv[NV_LEAF_DEPTH]; }
    pw[2 + col] = L[B3W_A_O + col];
    pw[6 + col] = L[B3W_A_O + 4 + col];
  }
  if (MODE == 1) dump_images<W, WORDS>(lds, reinterpret_cast<uint32_t *>(out), (uint32_t)pitch, wit0, n, okf);    // out = scratch, pitch = its row length
  else if (MODE == 2) {
    // a body whose inputs the circuit rejects, or that this path cannot read, verifies as "all wrong"
    if (lane < W) ncf[lane] |= okf[lane] ? 0u : 1u;
    __syncthreads();
    uint32_t cnt[W];
    expand_verify<W, WORDS, true>(lds, table, nwit, out, pitch, wb, n, cnt);
    publish_counts<W>(cnt, pub, wb, n, ncf);
  } else if (SL) expand<W, WORDS, true, NT>(lds, table, nwit, out, pitch, wb, n, okf, all_ok, slice, slices);
  else expand<W, WORDS, true, NT>(lds, table, nwit, out, pitch, wb, n, okf, all_ok);
  if (!PERSIST) return;
  __syncthreads();                                     // images and flags are overwritten by the next group's
 }
}


// ------------------------------------------------------------------ fill-ordered fused kernel ("REGIONFILL") — caller-owned buffers
// Where the fused kernels above leave thousands of body streams in flight, a plain hipMalloc buffer — all of it in ONE class of HBM
// (b3w_placement.hip) — takes 5.4-5.5 TB/s from them; the runtime's fill shape (256 workgroups of four waves, one per CU, workgroup b
// writing the 4 KiB blocks b, b + 256, ...) gets 6.4-6.5 out of the same memory, and the store-only sweep of round 6
// (tools/ubench/store_sweep.hip, profiles/r06/store_sweep*.log) says what of that shape matters: four-wave workgroups that store whole
// 4 KiB blocks, ONE workgroup per CU (twice as many: 5.3), every XCD writing the blocks of its own residue class (block index mod 8 =
// workgroup index mod 8: the dispatcher deals workgroups to the XCDs round-robin; without that: 5.5), and one compact window chip-wide.
// The fill order itself would need a trace per block (a workgroup's consecutive blocks lie 1 MiB = 1.4 bodies apart); a workgroup that
// stays in a 128 KiB region for four steps needs one per four, and that order ("A256x4" in the sweep) was the best of all: 6.56 TB/s.
//   * waves 0..3 of a workgroup store a block (1 KiB each; lane pair = one slot, as in EXPAND) from the images in LDS through the slot
//     table in LDS: their only vector-memory instructions are stores, and they never wait for one (a table word loaded from L2 by the
//     same wave would be waited for IN ORDER behind the wave's own stores — what holds the two-kernel sweep at 5.5 TB/s);
//   * wave 4 is the TRACER: while the others store the units of one half of the image buffer it computes the images of the next NH
//     units into the other half (four lanes per body, the same trace_compression), records prefetched a half ahead — one barrier per
//     NH units.
// Bit-identical to the other variants (tests/test_gpu_parity.py); b3w_batch_autotune_device times it on the buffer it is given.
// (Also built and measured in round 6, and taken out: the same workgroup dealt BODY-major — groups of 8 m workgroups per body, a workgroup
// the body's blocks = u (mod 8 m): 5.8-6.0 TB/s on one-class memory for m = 1 .. 8, profiles/r06/variant_scan_bf*.log.)
// The slot table in LDS, 16 bits a slot (so that the images fit beside it), made by the host (b3w_create):
//   image word (10 bits: below 1 024), then 5 bits that are the shift of a BIT slot or — under the word flag, bit 15 — the kind of a word slot:
//   0 one word, 1 two words, 2 (nova O2) a 256-bit slot, left to the wide-slot launch with the rest of its line.
// A nova image has 1 184 narrow words, but its slots mention only some 820 of them: the mentioned words from 1 024 on have an ALIAS below
// 1 024 in a word no slot mentions (b3w_create picks them; the tracer copies, alias_copy below), and the table names the alias — the same
// 16 bits as the compression circuit's, where a flag bit more (first kept in a bitmap beside the table) cost the nova storing waves eleven
// instructions a unit of the ~ 200 they are bound by.
// The host checks that the circuit's table can be said that way (fill_ok) before it offers the variant.
template <bool NOVA>
__device__ __forceinline__ void fill_store(uint32_t e, uint32_t w0, uint32_t w1, uint32_t par, bool in, uint8_t *p) {
  const uint32_t f = (e >> 10) & 31u;
  const bool word = (e & 0x8000u) != 0;
  uint4 v;
  v.x = par ? 0u : (word ? w0 : (w0 >> f) & 1u);
  v.y = (!par && word && (f & 1u)) ? w1 : 0u;
  v.z = 0; v.w = 0;
  if (in) store16<false>(p, v);
}
// The same store through a BUFFER resource over the body (base, body bytes): a lane whose offset lies outside — in front of the body
// (negative, i.e. huge), behind it, or made so on purpose (a rejected step; a line that holds a 256-bit slot) — is dropped by the hardware's
// range check: no branch and no exec-mask juggling per store (the nova storers were bound by their own instruction stream: 25 branches a unit).
// A 256-bit slot is not this launch's, and NEITHER ARE THE OTHER SLOTS OF ITS 128-BYTE LINE (eight lanes: the wave's KiB is aligned in
// absolute addresses): a line written in part costs the memory system far more than its bytes — the 67 slots of a nova body lie in 35 lines,
// 0.6 % of the body, and leaving 32-byte holes in them held the whole kernel at 6.4 instead of 6.9 TB/s.  The wide-slot launch writes those
// lines whole (b3w_nova_kernel MODE 3); of a line that straddles two bodies each body's unit decides for its own slots.
// ZONE: the wave's KiB may hold such a line (decided per store from scalars: the 67 slots lie within 256 slots of each other, nine of a
// body's 728 KiB); outside the zone a store costs neither the ballot nor the test for a 256-bit slot.
template <bool ZONE>
__device__ __forceinline__ void fill_store_nova(uint32_t e, uint32_t w0, uint32_t w1, uint32_t par, bool ok, __amdgpu_buffer_rsrc_t rsrc, uint32_t rel,
                                                uint32_t body, uint32_t lane) {
  const uint32_t f = (e >> 10) & 31u;
  const bool word = (e & 0x8000u) != 0;
  u32x4 v;
  v.x = par ? 0u : (word ? w0 : (w0 >> f) & 1u);
  v.y = (!par && word && (f & 1u)) ? w1 : 0u;
  v.z = 0; v.w = 0;
  bool skip = false;
  if (ZONE) {
    const uint64_t wide = __builtin_amdgcn_ballot_w64(word && f == 2u && rel < body);    // (a lane outside the body read garbage for e)
    skip = ((uint32_t)(wide >> (lane & 56u)) & 0xFFu) != 0;
  }
  const uint32_t off = (ok && !skip) ? rel : 0xFFFFFFF0u;
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)off, 0, 0);
}

// A body's public outputs and status go out through a STORING wave (from the image: they are part of it), not through the tracer:
// the tracer's vector-memory counter then holds loads only — behind a saturated write stream a store is acknowledged microseconds
// later, and the tracer would wait for its own output stores in order before it may use the next half's records.
#define B3W_LDS_STWORD 46   // unused pad word of the image: the step's status (nova, fill-ordered kernel)
template <bool NOVA>
__device__ __forceinline__ void fill_report(const uint32_t *img, uint32_t w, uint32_t lane, uint32_t *__restrict__ pub, int32_t *__restrict__ status) {
  if (!NOVA) {
    if (pub && lane < 16) pub[(uint64_t)w * 16 + lane] = img[B3W_A_O + lane];
    if (status && lane == 16) status[w] = 0;                                // canonical u32 inputs cannot fail an assert
    return;
  }
  // w[1..15]: n_blocks_out block_count_out h_out[8] total_depth_out depth_out chunk_idx_low_out chunk_idx_high_out leaf_depth_out
  if (pub && lane < 15 && img[B3W_LDS_OKWORD]) {
    const uint32_t *nv = img + B3W_LDS_NV;
    const uint32_t src = lane == 0 ? B3W_LDS_NV + NV_N_BLOCKS : lane == 1 ? B3W_LDS_NV + NV_BLOCK_COUNT_OUT : lane < 10 ? B3W_A_O + (lane - 2)
                       : lane == 10 ? B3W_LDS_NV + NV_TOTAL_DEPTH : lane == 11 ? B3W_LDS_NV + NV_DEPTH_OUT : lane == 12 ? B3W_LDS_NV + NV_CIL
                       : lane == 13 ? B3W_LDS_NV + NV_CIH : B3W_LDS_NV + NV_LEAF_DEPTH;
    (void)nv;
    pub[(uint64_t)w * 15 + lane] = img[src];
  }
  if (status && lane == 16) status[w] = (int32_t)img[B3W_LDS_STWORD];
}

// the tracer's half: records of `cnt` bodies (w_of(k)) from registers into the images; while it traces, the NEXT half's records are
// already on their way (a dependent global load costs as much as the whole trace).  NREC words a record, to image word DST of a
// WORDS-word image.
template <int NH, int NREC>
struct TracerRecs { uint32_t r[(NH * NREC + 63) / 64]; };
template <int NH, int NREC, typename WOf>
__device__ __forceinline__ void tracer_load(TracerRecs<NH, NREC> &t, const uint32_t *__restrict__ recs, uint32_t cnt, uint32_t lane, WOf w_of) {
#pragma unroll
  for (int c = 0; c < (NH * NREC + 63) / 64; ++c) {
    const uint32_t i = lane + 64u * c, k = i / NREC, j = i - k * NREC;
    t.r[c] = k < cnt ? recs[(uint64_t)w_of(k) * NREC + j] : 0u;
  }
}
template <int NH, int NREC, int WORDS, int DST>
__device__ __forceinline__ void tracer_put(const TracerRecs<NH, NREC> &t, uint32_t *half, uint32_t cnt, uint32_t lane) {
#pragma unroll
  for (int c = 0; c < (NH * NREC + 63) / 64; ++c) {
    const uint32_t i = lane + 64u * c, k = i / NREC, j = i - k * NREC;
    if (k < cnt) half[k * WORDS + DST + j] = t.r[c];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The deal: the buffer is cut into absolute 128 KiB REGIONS of 32 blocks; the eight workgroups of a group (one per XCD) share a region,
// workgroup x storing its blocks = x (mod 8) — four steps —, and the 32 groups take 32 consecutive regions: chip-wide ONE contiguous
// 4 MiB window that moves on every four steps.  A region lies in one body or across the border of two: the UNIT of work is
// (region, body), NH units per half of the image buffer.
// The units a workgroup's group visits are listed by the TRACER, 64 region visits at a time (a lane per visit: which body reaches into
// the region — one integer division —, does a second one start in it; a ballot numbers the units), into a ring of descriptors in LDS
// that the storing waves read: a wave walking the regions by itself spends more scalar branches per unit than the unit has stores.
// All positions in SLOT units (32 bytes) from the start of the region that holds the buffer's first byte: 32-bit arithmetic (the
// launch checks n * pitch < 2^37 bytes).
#define B3W_RF_RING 256
struct RegionList {
  uint32_t n, body32, pitch32, off0, end;   // slots per body, from body to body, of the first body's start, one past the last body's last slot
  uint32_t grp, v_next, produced;           // this workgroup's group; next visit to list; units listed so far
  int32_t zlo, zhi, xoff;                   // nova: a body's lowest / highest 256-bit slot; this workgroup's block in a region, in slots (zlo > zhi: none)
  bool done;
};
// descriptor flags beside the body number (below 2^29: n * pitch < 2^37 bytes, a body at least a region)
#define B3W_RF_STARTS 0x80000000u    // the body starts in this region: the unit reports its outputs and status
#define B3W_RF_NEAR 0x40000000u      // nova: one of the workgroup's four KiB of this unit may hold a line with a 256-bit slot (fill_store_nova<true>)
#define B3W_RF_REJECT 0x20000000u    // nova: the step was rejected (set by the tracer once the unit is traced): nothing of it is stored
#define B3W_RF_BODY 0x1FFFFFFFu
__device__ __forceinline__ void region_list_batch(RegionList &s, uint2 *ring, uint32_t lane) {
  const uint32_t v = s.v_next + lane, j = s.grp + 32u * v;
  const bool valid = (uint64_t)j << 12 < s.end;
  const uint32_t rs = j << 12, re = rs + 4096u;
  const uint32_t w_lo = (valid && rs >= s.off0 + s.body32) ? (rs - s.off0 - s.body32) / s.pitch32 + 1u : 0u;   // bodies that end before the region
  const uint32_t bs = s.off0 + w_lo * s.pitch32;
  const bool e1 = valid && w_lo < s.n && bs < re;                       // (bs >= re: the region lies in a gap between two bodies, pitch > body)
  const bool e2 = e1 && w_lo + 1 < s.n && bs + s.pitch32 < re;          // a second body starts in the region
  const uint64_t b1 = __builtin_amdgcn_ballot_w64(e1), b2 = __builtin_amdgcn_ballot_w64(e2), lt = (1ull << lane) - 1;
  const uint32_t pos = s.produced + __builtin_popcountll(b1 & lt) + __builtin_popcountll(b2 & lt);
  // descriptor: x = body | flags, y = bytes from the body's start to the region's (negative: the body starts inside it)
  // (near: the workgroup's waves store the body's slots [t, t + 32) + 1 024 r, t = rel + xoff + 32 sub, sub and r = 0 .. 3)
  auto near = [&](int32_t rel) { const int32_t t = rel + s.xoff; return (t + 96 + 3072 + 35 > s.zlo && t < s.zhi + 4) ? B3W_RF_NEAR : 0u; };
  if (e1) ring[pos % B3W_RF_RING] = make_uint2(w_lo | (bs >= rs ? B3W_RF_STARTS : 0u) | near((int32_t)(rs - bs)), (uint32_t)((int32_t)(rs - bs) << 5));
  if (e2) ring[(pos + 1) % B3W_RF_RING] = make_uint2((w_lo + 1) | B3W_RF_STARTS | near((int32_t)(rs - bs - s.pitch32)), (uint32_t)((int32_t)(rs - bs - s.pitch32) << 5));
  s.produced = uni(s.produced + (uint32_t)__builtin_popcountll(b1) + (uint32_t)__builtin_popcountll(b2));
  s.v_next += 64;
  s.done = s.done || __builtin_amdgcn_ballot_w64(!valid) != 0;
}

// KIND: the compression circuit, or the nova O2 builds — whose images here are the NARROW part only (4.7 instead of 6.9 KB: the tracer
// computes the IsZero gadgets' flags, not their inverses; the 67 256-bit slots of a body are skipped and written by a second small
// launch, b3w_nova_kernel MODE 3).
// (Two groups of four storing waves were measured for the nova storers, which do more per slot: the storers' own time halves, the store
// rate FALLS — 5.2 against 5.9 TB/s —, eight storing waves a CU being no fill shape any more.)
template <int KIND, int NH>
__global__ __launch_bounds__(320, 1) void b3w_regionfill_kernel(const uint32_t *__restrict__ recs, uint32_t n,
                                                                uint8_t *__restrict__ out, uint64_t pitch,
                                                                const uint32_t *__restrict__ table, uint32_t nwit,
                                                                uint32_t *__restrict__ pub, int32_t *__restrict__ status, uint32_t pace,
                                                                const uint32_t *__restrict__ aux) {
  constexpr bool NOVA = KIND != B3W_KIND_COMP;
  constexpr int WORDS = NOVA ? B3W_LDS_WIDE : B3W_LDS_WORDS_COMP, NREC = NOVA ? 32 : 28, RDST = NOVA ? B3W_LDS_NV : B3W_A_H, R = 4;
  extern __shared__ __attribute__((aligned(16))) uint32_t bf_lds[];
  __shared__ uint2 ring[B3W_RF_RING];                                       // unit descriptors, unit k at k % ring
  __shared__ uint32_t cnt_ring[4];                                          // units of half h at h % 4
  uint16_t *tab = reinterpret_cast<uint16_t *>(bf_lds);
  const uint32_t tabw = ((nwit + 7u) & ~7u) / 2;
  uint32_t *lds = bf_lds + tabw;
  const uint32_t wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t x = blockIdx.x & 7u, grp = blockIdx.x >> 3;              // XCD residue; group (32 of them: gridDim.x = 256)
  const uint32_t body = 32u * nwit;
  // The 16-bit table (made by the host, behind the 32-bit one: b3w_create) into LDS — by the four STORING waves, 32 entries a thread in
  // flight (four 16-byte loads at a time: one entry per load and iteration, as this loop first stood, was 75 dependent L2 round trips = 40 us
  // of every launch), while the tracer wave is already listing, loading and tracing the first half: the first barrier below is both "table
  // there" and "half 0 there".
  if (wave < 4) {
    const uint32_t padded = ((nwit + 31u) / 32u + 8u) * 32u;                // entries of the 32-bit table (build_slot_table)
    const uint4 *t8 = reinterpret_cast<const uint4 *>(table + padded);      // (16-byte aligned: padded is a multiple of 32 words)
    uint4 *tab8 = reinterpret_cast<uint4 *>(tab);
    const uint32_t n8 = (nwit + 7u) / 8u;
    for (uint32_t base8 = 0; base8 < n8; base8 += 256u * 4u) {
      uint4 v[4];
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        const uint32_t i8 = base8 + 256u * uu + threadIdx.x;
        v[uu] = i8 < n8 ? t8[i8] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        const uint32_t i8 = base8 + 256u * uu + threadIdx.x;
        if (i8 < n8) tab8[i8] = v[uu];
      }
    }
  }
  if (wave == 4) {
    // ---- TRACER: lists the units two halves ahead, loads the records one half ahead, traces half h into image half h & 1; one barrier
    // per half, the last one for a half that is not full (possibly empty)
    const uint32_t q = lane >> 2, col = lane & 3;
    RegionList rl;
    rl.n = n; rl.body32 = nwit; rl.pitch32 = (uint32_t)(pitch >> 5);
    rl.off0 = (uint32_t)((reinterpret_cast<uint64_t>(out) & ((1ull << 17) - 1)) >> 5);
    rl.end = rl.off0 + (n - 1) * rl.pitch32 + rl.body32;
    rl.grp = grp; rl.v_next = 0; rl.produced = 0; rl.done = false;
    rl.xoff = (int32_t)(x << 7);
    rl.zlo = NOVA ? (int32_t)aux[B3W_AUX_WIDE_SLOTS] : 0x7FFFFFFF;          // (the slot numbers ascend: b3w_create checks it)
    rl.zhi = NOVA ? (int32_t)aux[B3W_AUX_WIDE_SLOTS + B3W_NOVA_ISZERO - 1] : -1;
    auto list_until = [&](uint32_t target) {
      while (!rl.done && rl.produced < target) region_list_batch(rl, ring, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    auto cnt_of = [&](uint32_t h) { return rl.produced <= h * NH ? 0u : (rl.produced - h * NH < (uint32_t)NH ? rl.produced - h * NH : (uint32_t)NH); };
    // nova: the aliases of the image words from 1 024 on (aux: from | to << 16), three a lane; copied once a half is traced
    uint32_t al[3] = {0, 0, 0};
    if (NOVA) {
      const uint32_t na = aux[B3W_AUX_ALIAS_COUNT];
#pragma unroll
      for (int u = 0; u < 3; ++u) al[u] = lane + 64u * u < na ? aux[B3W_AUX_ALIAS_LIST + lane + 64u * u] : 0u;
    }
    TracerRecs<NH, NREC> tr;
    list_until(2 * NH);
    tracer_load<NH, NREC>(tr, recs, cnt_of(0), lane, [&](uint32_t k) { return ring[k % B3W_RF_RING].x & B3W_RF_BODY; });
    for (uint32_t h = 0;; ++h) {
      uint32_t *half = lds + (h & 1) * NH * WORDS;
      list_until((h + 3) * NH);                                             // (ring: at most 3 NH + 127 units between the oldest in use and the newest)
      const uint32_t cnt = cnt_of(h);
      if (lane == 0) cnt_ring[h & 3] = cnt;
      tracer_put<NH, NREC, WORDS, RDST>(tr, half, cnt, lane);
      tracer_load<NH, NREC>(tr, recs, cnt_of(h + 1), lane, [&](uint32_t k) { return ring[((h + 1) * NH + k) % B3W_RF_RING].x & B3W_RF_BODY; });
      if (!NOVA) {
        if (q < cnt) trace_compression(half + q * WORDS, col, nullptr);     // (outputs: the storers', see fill_report)
      } else {
        if (q < cnt) {
          uint32_t *L = half + q * WORDS;
          const bool dom_a = nova_iszero_flags_quad(L, (int)col);           // (flags only: no field arithmetic, no table of inverses)
          const int32_t st = nova_select(L, (int)col, dom_a);
          if (col == 0) {
            L[B3W_LDS_OKWORD] = st == 0 ? 1u : 0u; L[B3W_LDS_STWORD] = (uint32_t)st;
            if (st != 0) ring[(h * NH + q) % B3W_RF_RING].x |= B3W_RF_REJECT;   // (this unit's descriptor: nobody else's to write, and read behind the barrier)
          }
          if (st == 0) trace_compression(L, col, nullptr);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // (the whole wave copies what its quads have written)
        __builtin_amdgcn_wave_barrier();
        {
        uint32_t av[NH * 3];                                                  // (all reads, then all writes: one LDS round trip, not 33)
#pragma unroll
        for (int qq = 0; qq < NH; ++qq)
#pragma unroll
          for (int u = 0; u < 3; ++u) av[qq * 3 + u] = ((uint32_t)qq < cnt && al[u]) ? half[qq * WORDS + (al[u] & 0xFFFFu)] : 0u;
#pragma unroll
        for (int qq = 0; qq < NH; ++qq)
#pragma unroll
          for (int u = 0; u < 3; ++u) if ((uint32_t)qq < cnt && al[u]) half[qq * WORDS + (al[u] >> 16)] = av[qq * 3 + u];
        }
      }
      __syncthreads();
      if (cnt < (uint32_t)NH) break;
    }
    return;
  }
  // ---- STORERS: per half, after its barrier: the units' descriptors from the ring; table words one unit ahead
  __builtin_amdgcn_s_setprio(3);                                             // (wave 0 shares its SIMD with the tracer: the stores go first)
  const uint32_t par = lane & 1u;
  const uint32_t sub = wave;                                                // which KiB of a block this storing wave takes
  // (a lane's byte offset into a region in step 0: (x << 12) + (sub << 10) + (lane << 4); step r: + r * 32 KiB)
  // A unit goes through three stages, each an LDS round trip behind the one before: S1 descriptor -> table words; S2 image words (and the
  // image's ok word); S3 shape and store.  Three register sets rotate so that S1 of unit i + 2 and S2 of unit i + 1 are in flight while
  // unit i is stored (a lone wave per SIMD has nobody else to hide the round trips behind: exposed, they were a quarter of a unit's time).
  struct Ent { uint32_t wq; int32_t rel0, relw; uint32_t e[R]; };             // relw: rel0 of the wave's lane 0 (a scalar)
  struct Wd { uint32_t w0[R], w1[R]; };
  auto s1 = [&](uint32_t k, Ent &en) {                                       // (the descriptor read a further stage ahead was measured: slower)
    const uint2 d = ring[k % B3W_RF_RING];
    en.wq = uni(d.x);
    en.relw = (int32_t)uni(d.y) + (int32_t)((x << 12) + (sub << 10));
    en.rel0 = en.relw + (int32_t)(lane << 4);
    // the four steps' slots are 1 024 apart (a step is 32 KiB further on): ONE address and immediate offsets for the table words.  No clamping: a lane in front of or behind the body reads whatever
    // lies there — or zeros outside the LDS allocation — and its store is suppressed (range check / predicate) anyway.
    const int32_t s0 = en.rel0 >> 5;
    const uint16_t *tp = tab + s0;
#pragma unroll
    for (int r = 0; r < R; ++r) en.e[r] = tp[1024 * r];
  };
  auto s2 = [&](const Ent &en, Wd &wd, const uint32_t *img) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t *L = img + (en.e[r] & 0x3FFu);
      wd.w0[r] = L[0]; wd.w1[r] = L[1];
    }
  };
  auto s3 = [&](const Ent &en, const Wd &wd, const uint32_t *img) {
    // PACE, see the launch: sleeps (to the next 64-clock grid point each) and finer single steps.  Compression: once per REGION visit, not per
    // unit — where a body starts inside the region the visit is two units (the end of one body, the head of the next) that store one region's
    // bytes together: paced twice, the groups that meet more borders fell behind the others (r06: per visit, the same pace holds on every box
    // measured where per unit one of three lost the window at 32 768 witnesses).  (The nova instantiation keeps the pace per unit: its code
    // is sensitive to every branch in this loop — the same test there cost 9 %.)
    if (NOVA || !(en.wq & B3W_RF_STARTS)) {
      for (uint32_t z = 0; z < (pace & 15u); ++z) __builtin_amdgcn_s_sleep(1);
      uint32_t pv = lane;
      for (uint32_t z = 0; z < (pace >> 4); ++z) asm volatile("v_add_u32 %0, %0, 1" : "+v"(pv));
    }
    const uint32_t w = en.wq & B3W_RF_BODY;
    if ((en.wq >> 31) && x == 0 && sub == 0) fill_report<NOVA>(img, w, lane, pub, status);    // the unit in which the body starts reports for it
    uint8_t *dst = out + (uint64_t)w * pitch;
    if (NOVA) {
      const bool ok = !(en.wq & B3W_RF_REJECT);                               // a rejected step's body is left alone
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)body, 0x00020000);
      // near the 256-bit slots?  (the tracer's word in the descriptor, a scalar: one branch a unit)
      if (en.wq & B3W_RF_NEAR) {
#pragma unroll
        for (int r = 0; r < R; ++r) fill_store_nova<true>(en.e[r], wd.w0[r], wd.w1[r], par, ok, rsrc, (uint32_t)(en.rel0 + (int32_t)(r << 15)), body, lane);
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) fill_store_nova<false>(en.e[r], wd.w0[r], wd.w1[r], par, ok, rsrc, (uint32_t)(en.rel0 + (int32_t)(r << 15)), body, lane);
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const uint32_t rel = (uint32_t)(en.rel0 + (int32_t)(r << 15));
        fill_store<NOVA>(en.e[r], wd.w0[r], wd.w1[r], par, rel < body, dst + rel);
      }
    }
  };
  for (uint32_t h = 0;; ++h) {
    __syncthreads();                                                         // image half h & 1, descriptors and count of half h are there
    const uint32_t cnt = uni(cnt_ring[h & 3]);
    const uint32_t *half = lds + (h & 1) * NH * WORDS;
    const uint32_t k0 = h * NH;
    Ent e0, e1, e2;
    Wd d0, d1, d2;
    if (0 < cnt) s1(k0, e0);
    if (1 < cnt) s1(k0 + 1, e1);
    if (0 < cnt) s2(e0, d0, half);
    for (uint32_t i = 0; i < cnt; i += 3) {
      if (i + 2 < cnt) s1(k0 + i + 2, e2);
      if (i + 1 < cnt) s2(e1, d1, half + (i + 1) * WORDS);
      s3(e0, d0, half + i * WORDS);
      if (i + 1 >= cnt) break;
      if (i + 3 < cnt) s1(k0 + i + 3, e0);
      if (i + 2 < cnt) s2(e2, d2, half + (i + 2) * WORDS);
      s3(e1, d1, half + (i + 1) * WORDS);
      if (i + 2 >= cnt) break;
      if (i + 4 < cnt) s1(k0 + i + 4, e1);
      if (i + 3 < cnt) s2(e0, d0, half + (i + 3) * WORDS);
      s3(e2, d2, half + (i + 2) * WORDS);
    }
    if (cnt < (uint32_t)NH) break;
  }
}

}  // namespace

// ------------------------------------------------------------------ launch
namespace {
template <bool WIDE, int K>
int launch_sweep(const uint32_t *d_images, uint32_t n, uint8_t *d_out, uint64_t pitch, const uint32_t *d_table,
                 uint32_t nwit, hipStream_t stream) {
  const uintptr_t addr = reinterpret_cast<uintptr_t>(d_out);
  if (pitch >= (1ull << 30) || (addr & 31)) return -5;        // one 32-byte slot per lane: bodies must be 32-byte aligned
  const uint32_t lead = (uint32_t)(addr & 4095);
  static const int shape = getenv("B3W_SWEEP_SHAPE") ? atoi(getenv("B3W_SWEEP_SHAPE")) : 2;   // 2 = two wave-pairs, split
  const size_t smem = (size_t)nwit * 4 + 16;                 // the slot table
  constexpr size_t B3W_SWEEP_MAX_SMEM = 24614 * 4 + 16;       // the largest circuit's table (nova O1)
  if (smem > B3W_SWEEP_MAX_SMEM) return -5;
#define B3W_SWEEP_LAUNCH(PAIRS, SPLIT)                                                                              \
  {                                                                                                                 \
    /* the attribute is per device: one bit per device ordinal (setting it twice in a race is harmless) */         \
    static std::atomic<uint64_t> attr_done{0};                                                                      \
    int dev = 0;                                                                                                    \
    (void)hipGetDevice(&dev);                                                                                       \
    const uint64_t bit = 1ull << (dev & 63);                                                                        \
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {                                                       \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&b3w_sweep_kernel<WIDE, K, B3W_SWEEP_LOGC, PAIRS, SPLIT>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3W_SWEEP_MAX_SMEM);     \
      if (e != hipSuccess) return (int)e;                                                                           \
      attr_done.fetch_or(bit, std::memory_order_release);                                                           \
    }                                                                                                               \
    hipLaunchKernelGGL((b3w_sweep_kernel<WIDE, K, B3W_SWEEP_LOGC, PAIRS, SPLIT>), dim3(B3W_SWEEP_GRID), dim3(128 * PAIRS), smem, \
                       stream, d_images, n, d_out - lead, lead, (uint32_t)pitch, d_table, nwit);                    \
  }
  switch (shape) {
    case 1: B3W_SWEEP_LAUNCH(1, false); break;
    case 3: B3W_SWEEP_LAUNCH(4, true); break;
    case 4: B3W_SWEEP_LAUNCH(3, true); break;
    case 0: B3W_SWEEP_LAUNCH(2, false); break;
    default: B3W_SWEEP_LAUNCH(2, true); break;
  }
#undef B3W_SWEEP_LAUNCH
  return (int)hipGetLastError();
}
}  // namespace

// VERIFY launch: d_in_slots[j] = slot of record word j; d_mismatch[i] = differing 16-byte units of body i
// (0 = the body is the witness its own inputs determine; 0xFFFFFFFF = rejected inputs / not checkable here)
extern "C" int b3w_launch_verify(int kind, const uint32_t *d_in_slots, uint32_t n, const uint8_t *d_bodies, uint64_t pitch,
                                 const uint32_t *d_table, uint32_t nwit, uint32_t *d_mismatch, const void *d_aux,
                                 hipStream_t stream) {
  if (n == 0) return 0;
  uint8_t *bodies = const_cast<uint8_t *>(d_bodies);
  const uint32_t pm = (uint32_t)(pitch >> 5) & 3u, stride = (pitch & 31) ? 1u : pm == 0 ? 1u : pm == 2 ? 2u : 4u;   // WaveBodies
#define B3W_VGRID(WV) dim3((n + stride * WV - 1) / (stride * WV) * stride)
  if (kind == B3W_KIND_COMP)
    hipLaunchKernelGGL((b3w_compression_kernel<4, false, 2>), B3W_VGRID(4), dim3(64), 0, stream, d_in_slots, n, bodies, pitch,
                       d_table, nwit, d_mismatch, (int32_t *)nullptr, stride);
  else if (kind == B3W_KIND_NOVA_O2)
    hipLaunchKernelGGL((b3w_nova_kernel<B3W_KIND_NOVA_O2, 2, false, 2>), B3W_VGRID(2), dim3(64), 0, stream, d_in_slots, n,
                       bodies, pitch, d_table, nwit, d_mismatch, (int32_t *)nullptr, (const uint32_t *)d_aux, stride);
  else
    hipLaunchKernelGGL((b3w_nova_kernel<B3W_KIND_NOVA_O1, 2, false, 2>), B3W_VGRID(2), dim3(64), 0, stream, d_in_slots, n,
                       bodies, pitch, d_table, nwit, d_mismatch, (int32_t *)nullptr, (const uint32_t *)d_aux, stride);
#undef B3W_VGRID
  return (int)hipGetLastError();
}

// TRACE kernel alone: the images of `cn` <= row witnesses go to d_images, word-major (word j of witness w at d_images[j * row + w];
// row B3W_LDS_OKWORD = 1 for a valid witness), with the public outputs and the status words.  The first half of the
// two-kernel path; also what the commitment consumer needs of a witness (b3w_commit.hip, records mode).
extern "C" int b3w_launch_trace(int kind, const uint32_t *d_recs, uint32_t cn, uint32_t *d_images, uint32_t row, const uint32_t *d_table,
                                uint32_t nwit, uint32_t *d_pub, int32_t *d_status, const void *d_aux, hipStream_t stream) {
  if (cn == 0) return 0;
  if (cn > row) return -4;
  if (kind == B3W_KIND_COMP)
    hipLaunchKernelGGL((b3w_compression_kernel<16, false, 1>), dim3((cn + 15) / 16), dim3(64), 0, stream, d_recs, cn,
                       reinterpret_cast<uint8_t *>(d_images), (uint64_t)row, d_table, nwit, d_pub, d_status, 1u);
  else if (kind == B3W_KIND_NOVA_O2)
    hipLaunchKernelGGL((b3w_nova_kernel<B3W_KIND_NOVA_O2, 4, false, 1>), dim3((cn + 3) / 4), dim3(64), 0, stream, d_recs, cn,
                       reinterpret_cast<uint8_t *>(d_images), (uint64_t)row, d_table, nwit, d_pub, d_status, (const uint32_t *)d_aux, 1u);
  else if (kind == B3W_KIND_NOVA_O1)
    hipLaunchKernelGGL((b3w_nova_kernel<B3W_KIND_NOVA_O1, 2, false, 1>), dim3((cn + 1) / 2), dim3(64), 0, stream, d_recs, cn,
                       reinterpret_cast<uint8_t *>(d_images), (uint64_t)row, d_table, nwit, d_pub, d_status, (const uint32_t *)d_aux, 1u);
  else return -2;
  return (int)hipGetLastError();
}

extern "C" int b3w_launch_batch(int kind, int variant, const uint32_t *d_recs, uint32_t n, uint8_t *d_out,
                                uint64_t pitch, const uint32_t *d_table, uint32_t nwit, uint32_t *d_pub,
                                int32_t *d_status, const void *d_aux, uint32_t *d_scratch, uint32_t scratch_cap,
                                hipStream_t stream) {
  if (n == 0) return 0;
  if (variant >= B3W_VARIANT_SWEEP && variant < B3W_VARIANT_REGIONFILL) {
    // TRACE kernel -> scratch, SWEEP kernel -> bodies, in chunks of scratch_cap witnesses
    if (!d_scratch || scratch_cap != (1u << B3W_SWEEP_LOGC)) return -4;
    if (kind != B3W_KIND_COMP && !d_aux) return -3;
    uint32_t *d_images = d_scratch;                                        // word-major images
    for (uint32_t c0 = 0; c0 < n; c0 += scratch_cap) {
      const uint32_t cn = n - c0 < scratch_cap ? n - c0 : scratch_cap;
      uint32_t *pub_c = d_pub ? d_pub + (uint64_t)c0 * (kind == B3W_KIND_COMP ? 16 : 15) : nullptr;
      int32_t *st_c = d_status ? d_status + c0 : nullptr;
      uint8_t *out_c = d_out + (uint64_t)c0 * pitch;
      int rc = b3w_launch_trace(kind, d_recs + (uint64_t)c0 * (kind == B3W_KIND_COMP ? 28 : 32), cn, d_images, scratch_cap, d_table, nwit, pub_c,
                                st_c, d_aux, stream);
      if (rc) return rc;
      if (kind == B3W_KIND_COMP) rc = launch_sweep<false, 8>(d_images, cn, out_c, pitch, d_table, nwit, stream);
      else rc = launch_sweep<true, 4>(d_images, cn, out_c, pitch, d_table, nwit, stream);
      if (rc) return rc;
    }
    return (int)hipGetLastError();
  }
  if (variant == B3W_VARIANT_REGIONFILL || variant == B3W_VARIANT_REGIONFILL_LIGHT) {
    // 256 workgroups (one per CU; workgroup i on XCD i % 8); 32-byte aligned bodies (a lane pair is one slot); compression and nova O2
    if (kind != B3W_KIND_COMP && kind != B3W_KIND_NOVA_O2) return -1;
    if (kind == B3W_KIND_NOVA_O2 && !d_aux) return -3;
    if ((reinterpret_cast<uintptr_t>(d_out) & 31) || (pitch & 31) || pitch >= (1ull << 30)) return -5;
    if (32ull * nwit < (1ull << 17) || (uint64_t)n * pitch + (1ull << 20) >= (1ull << 37)) return -5;      // a region meets two bodies at most; 32-bit slot positions
    // slot table (16 bits a slot) [+ one bit a slot] + 2 x NH images: 14 of 3.7 KB (compression), 11 of the 4.7 KB narrow nova image
    constexpr int NH_C = 14, NH_N = 11;
    const bool nova = kind != B3W_KIND_COMP;
    const size_t smem = (size_t)((nwit + 7u) & ~7u) * 2 +
                        (size_t)(2 * (nova ? NH_N * B3W_LDS_WIDE : NH_C * B3W_LDS_WORDS_COMP) + 4) * 4;
    constexpr size_t B3W_FILL_MAX_SMEM = 160 * 1024 - 4096;      // (+ the kernel's static descriptor ring)
    if (smem > B3W_FILL_MAX_SMEM) return -5;
    static std::atomic<uint64_t> attr_done{0};                   // per device: one bit per ordinal, as for the sweep kernels
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(attr_done.load(std::memory_order_acquire) & bit)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&b3w_regionfill_kernel<B3W_KIND_COMP, NH_C>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)B3W_FILL_MAX_SMEM);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&b3w_regionfill_kernel<B3W_KIND_NOVA_O2, NH_N>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)B3W_FILL_MAX_SMEM);
      if (e != hipSuccess) return (int)e;
      attr_done.fetch_or(bit, std::memory_order_release);
    }
    // PACE: every storing wave waits per region visit (compression) / per unit (nova) — pace % 16 x s_sleep 1 (to the chip's next 64-clock
    // grid point each) and pace / 16 single vector-ALU steps (~ 25 clocks each with their loop).  The storers are built to issue as little as
    // they can, and then they are too fast: unpaced, the workgroups run at whatever the memory system lets each of them have, drift apart, and
    // the one compact window the fill order lives on frays — 6.6 (compression) TB/s on one-class memory.  Paced to just under the memory's
    // rate they stay in step, and the rate is a CLIFF in the pace: 6.6 below it, 7.15-7.2 (4 096 witnesses) / 7.3-7.35 (16 384 and more) on its
    // edge, -1 % for every further step, on placed, one-class and hipMalloc buffers alike — but on the edge a launch now and then falls off
    // (6.7-7.2), and the edge moves by a step from box to box.  Compression: one sleep and two steps per visit (33) held in all 72 cases —
    // three boxes x {256 ... 32 768 witnesses} x {placed, one-class, torch.empty} — at 7.11-7.19 / 7.26-7.30; three steps (48) are the edge
    // (7.15-7.21 / 7.3-7.35; fell to 7.06 once): variant 201 (tools/ubench/pace_robust.py, profiles/r06/pace_robust*.log, pace_visit*.log).
    // Nova (per unit): edge at 33 — 7.0-7.07 TB/s, or 6.5-6.7 —; 18 held in all twelve cases at 6.74-6.97 (pace_robust_nova*.log).
    // B3W_FILL_PACE overrides (measurements).
    const char *pace_s = getenv("B3W_FILL_PACE");                      // (read per launch: pace_scan.py changes it inside one process)
    const int pace_env = pace_s ? atoi(pace_s) : -1;
    const uint32_t pace = pace_env >= 0 ? (uint32_t)pace_env : variant == B3W_VARIANT_REGIONFILL_LIGHT ? (nova ? 33u : 48u) : nova ? 18u : 33u;
    if (!nova) {
      hipLaunchKernelGGL((b3w_regionfill_kernel<B3W_KIND_COMP, NH_C>), dim3(B3W_REGIONFILL_GRID), dim3(320), smem, stream, d_recs, n, d_out, pitch,
                         d_table, nwit, d_pub, d_status, pace, (const uint32_t *)nullptr);
    } else {
      hipLaunchKernelGGL((b3w_regionfill_kernel<B3W_KIND_NOVA_O2, NH_N>), dim3(B3W_REGIONFILL_GRID), dim3(320), smem, stream, d_recs, n, d_out, pitch,
                         d_table, nwit, d_pub, d_status, pace, (const uint32_t *)d_aux);
      // ... and the lines that hold the 67 field inverses of every body (35 lines, 0.6 % of its bytes): the body-stream kernel's gadget and
      // select phases, and those lines' stores alone
      hipLaunchKernelGGL((b3w_nova_kernel<B3W_KIND_NOVA_O2, 2, false, 3>), dim3((n + 1) / 2), dim3(64), 0, stream, d_recs, n, d_out, pitch, d_table, nwit,
                         (uint32_t *)nullptr, (int32_t *)nullptr, (const uint32_t *)d_aux, 1u);
    }
    return (int)hipGetLastError();
  }
  if (variant > B3W_VARIANT_REGIONFILL_LIGHT) return -1;
  // bodies that start at the same offset into a 128-byte line are `stride` apart (WaveBodies); a wave takes W of them
  const uint32_t pm = (uint32_t)(pitch >> 5) & 3u, stride = (pitch & 31) ? 1u : pm == 0 ? 1u : pm == 2 ? 2u : 4u;
#define B3W_GRID(WV) dim3((n + stride * WV - 1) / (stride * WV) * stride)
  static const uint32_t lds_pad = getenv("B3W_LDS_PAD") ? (uint32_t)atoi(getenv("B3W_LDS_PAD")) : 0;   // experiment: occupancy limiter
  if (kind == B3W_KIND_COMP) {
#define B3W_LAUNCH_COMP(WV, NTV)                                                                         \
  hipLaunchKernelGGL((b3w_compression_kernel<WV, NTV, 0>), B3W_GRID(WV), dim3(64), lds_pad, stream,        \
                     d_recs, n, d_out, pitch, d_table, nwit, d_pub, d_status, stride)
    if (variant >= B3W_VARIANT_SLICED && variant < B3W_VARIANT_SWEEP) {     // one body per wave, its tiles shared out to `slices` waves
      const uint32_t slices = (uint32_t)(variant - B3W_VARIANT_SLICED);
      if (slices < 2 || slices > 64) return -1;
      hipLaunchKernelGGL((b3w_compression_kernel<1, false, 0, true>), dim3(B3W_GRID(1).x * slices), dim3(64), 0, stream, d_recs, n, d_out,
                         pitch, d_table, nwit, d_pub, d_status, stride, slices);
      return (int)hipGetLastError();
    }
    switch (variant) {
      case 0: B3W_LAUNCH_COMP(4, false); break;
      case 1: B3W_LAUNCH_COMP(1, false); break;
      case 2: B3W_LAUNCH_COMP(2, false); break;
      case 3: B3W_LAUNCH_COMP(8, false); break;
      case 4: B3W_LAUNCH_COMP(4, true); break;
      case 5: B3W_LAUNCH_COMP(2, true); break;
      case 6: B3W_LAUNCH_COMP(1, true); break;
      case 7: B3W_LAUNCH_COMP(16, false); break;
      case 8:   // W = 8 with two workgroups per CU (24 KB of unused dynamic LDS): about 4096 body streams in flight chip-wide,
                // the count a 4096-witness batch has by itself; large batches 6.86 -> 7.30 TB/s (tools/ubench/occupancy.py)
        hipLaunchKernelGGL((b3w_compression_kernel<8, false, 0>), B3W_GRID(8), dim3(64), lds_pad ? lds_pad : 24576u, stream, d_recs, n, d_out,
                           pitch, d_table, nwit, d_pub, d_status, stride);
        break;
      default: return -1;
    }
#undef B3W_LAUNCH_COMP
    return (int)hipGetLastError();
  }
  if (kind == B3W_KIND_NOVA_O2 || kind == B3W_KIND_NOVA_O1) {
    if (!d_aux) return -3;
#define B3W_LAUNCH_NOVA(KV, WV)                                                                           \
  hipLaunchKernelGGL((b3w_nova_kernel<KV, WV, false, 0>), B3W_GRID(WV), dim3(64), lds_pad, stream,          \
                     d_recs, n, d_out, pitch, d_table, nwit, d_pub, d_status, (const uint32_t *)d_aux, stride)
#define B3W_LAUNCH_NOVA_P(KV, WV, G)                                                                            \
  hipLaunchKernelGGL((b3w_nova_kernel<KV, WV, false, 0, false, true>), dim3(B3W_GRID(WV).x < (G) ? B3W_GRID(WV).x : (G)), dim3(64), lds_pad, stream, \
                     d_recs, n, d_out, pitch, d_table, nwit, d_pub, d_status, (const uint32_t *)d_aux, stride)
    if (variant >= B3W_VARIANT_SLICED && variant < B3W_VARIANT_SWEEP) {
      const uint32_t slices = (uint32_t)(variant - B3W_VARIANT_SLICED);
      if (slices < 2 || slices > 64) return -1;
      if (kind == B3W_KIND_NOVA_O2)
        hipLaunchKernelGGL((b3w_nova_kernel<B3W_KIND_NOVA_O2, 1, false, 0, true>), dim3(B3W_GRID(1).x * slices), dim3(64), 0, stream, d_recs, n,
                           d_out, pitch, d_table, nwit, d_pub, d_status, (const uint32_t *)d_aux, stride, slices);
      else
        hipLaunchKernelGGL((b3w_nova_kernel<B3W_KIND_NOVA_O1, 1, false, 0, true>), dim3(B3W_GRID(1).x * slices), dim3(64), 0, stream, d_recs, n,
                           d_out, pitch, d_table, nwit, d_pub, d_status, (const uint32_t *)d_aux, stride, slices);
      return (int)hipGetLastError();
    }
    if (kind == B3W_KIND_NOVA_O2) {
      switch (variant) {
        case 0: B3W_LAUNCH_NOVA(B3W_KIND_NOVA_O2, 2); break;
        case 1: B3W_LAUNCH_NOVA(B3W_KIND_NOVA_O2, 1); break;
        case 2: B3W_LAUNCH_NOVA(B3W_KIND_NOVA_O2, 4); break;
        case 3: B3W_LAUNCH_NOVA(B3W_KIND_NOVA_O2, 8); break;
        // 4: W = 8 on a PERSISTENT grid of 512 waves (two per CU) that take the groups of 8 bodies in turn: for batches of tens of
        // thousands of steps (65 536: 7.13 against 7.04 TB/s; 16 384: 6.91 against 6.99 — profiles/r06/variant_scan_persistent.log, where the
        // same grid with 4 or 2 bodies a wave, the store-only sweep's best shape, is issue-bound at 5.7 with the kernel's real work)
        case 4: B3W_LAUNCH_NOVA_P(B3W_KIND_NOVA_O2, 8, 512u); break;
        default: return -1;
      }
    } else {
      switch (variant) {
        case 0: B3W_LAUNCH_NOVA(B3W_KIND_NOVA_O1, 2); break;
        case 1: B3W_LAUNCH_NOVA(B3W_KIND_NOVA_O1, 1); break;
        default: return -1;
      }
    }
#undef B3W_LAUNCH_NOVA
#undef B3W_LAUNCH_NOVA_P
    return (int)hipGetLastError();
  }
  return -2;
}
