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
//           of the image each 32-byte slot holds.  Lane pair (2j,2j+1) owns slot 32g+j of group g
//           and each lane stores one 16-byte half, so every wave store instruction is one fully
//           coalesced 1 KiB segment (global_store_dwordx4 x 64 lanes).
//
// The work is integer/bit expansion bound by HBM writes (770 976 B written for 112 B read per
// compression witness); there is no contraction, so no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "b3w_atoms.h"
#include "b3w_kernels.h"

namespace {

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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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

// U = groups per software-pipeline stage: the slot-table words of the next U groups are loaded
// before the U*W stores of the current ones are issued, so a wave waits on memory once per U KiB*W.
template <int W, int WORDS, bool WIDE, bool NT>
__device__ __forceinline__ void expand(const uint32_t *lds, const uint32_t *__restrict__ table, uint32_t nwit,
                                       uint8_t *__restrict__ out, uint64_t pitch, uint32_t wit0, uint32_t n,
                                       const uint32_t *okmask /* per-w LDS flags or null */) {
  constexpr int U = 4;
  const int lane = threadIdx.x;
  const uint32_t ngroups = (nwit + 31) >> 5, full = nwit >> 5;
  const uint32_t nact = n - wit0 < (uint32_t)W ? n - wit0 : (uint32_t)W;
  uint64_t woff[W];                                    // byte offset of this lane's 16 B in group g of witness w
#pragma unroll
  for (int w = 0; w < W; ++w) woff[w] = (uint64_t)(wit0 + w) * pitch + (uint32_t)lane * 16u;
  const uint32_t *tp = table + (lane >> 1);            // entry of this lane's slot in group 0
  uint32_t g = 0;
  if (nact == (uint32_t)W && !okmask) {
    uint32_t cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = tp[u * 32];   // table is padded by U groups past ngroups
    for (; g + U <= full; g += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) nxt[u] = tp[(g + U + u) * 32];
#pragma unroll
      for (int u = 0; u < U; ++u) emit_group<W, WORDS, WIDE, NT, true>(lds, cur[u], out, woff, u * 1024u, true, nact, nullptr);
#pragma unroll
      for (int w = 0; w < W; ++w) woff[w] += U * 1024u;
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
  } 
  for (; g < ngroups; ++g) {                            // ragged waves, and the last (partial) groups
    const uint32_t slot = g * 32 + (lane >> 1);
    emit_group<W, WORDS, WIDE, NT, false>(lds, tp[g * 32], out, woff, 0, slot < nwit, nact, okmask);
#pragma unroll
    for (int w = 0; w < W; ++w) woff[w] += 1024u;
  }
}

// ------------------------------------------------------------------ compression circuit
template <int W, bool NT>
__global__ __launch_bounds__(64) void b3w_compression_kernel(const uint32_t *__restrict__ recs, uint32_t n,
                                                             uint8_t *__restrict__ out, uint64_t pitch,
                                                             const uint32_t *__restrict__ table, uint32_t nwit,
                                                             uint32_t *__restrict__ pub, int32_t *__restrict__ status) {
  constexpr int WORDS = B3W_LDS_WORDS_COMP;
  __shared__ __attribute__((aligned(16))) uint32_t lds[W * WORDS + 4];   // +4: expand reads src+1 unconditionally
  const int lane = threadIdx.x;
  const uint32_t wit0 = blockIdx.x * W;
  // stage the 28-word records of this wave's witnesses into atoms H M T B D (image words 1..28)
  for (int i = lane; i < W * 28; i += 64) {
    const int w = i / 28, j = i - w * 28;
    if (wit0 + w < n) lds[w * WORDS + B3W_A_H + j] = recs[(uint64_t)(wit0 + w) * 28 + j];
  }
  __syncthreads();
  {
    const int w = lane >> 2, col = lane & 3;
    if (w < W && wit0 + w < n) {
      trace_compression(lds + w * WORDS, col, pub ? pub + (uint64_t)(wit0 + w) * 16 : nullptr);
      if (status && col == 0) status[wit0 + w] = 0;     // canonical u32 inputs cannot fail an assert
    }
  }
  __syncthreads();
  expand<W, WORDS, false, NT>(lds, table, nwit, out, pitch, wit0, n, nullptr);
}

}  // namespace

// ------------------------------------------------------------------ launch
extern "C" int b3w_launch_batch(int kind, int variant, const uint32_t *d_recs, uint32_t n, uint8_t *d_out,
                                uint64_t pitch, const uint32_t *d_table, uint32_t nwit, uint32_t *d_pub,
                                int32_t *d_status, const void *d_aux, hipStream_t stream) {
  (void)d_aux;
  if (n == 0) return 0;
  if (kind == B3W_KIND_COMP) {
#define B3W_LAUNCH_COMP(WV, NTV)                                                                         \
  hipLaunchKernelGGL((b3w_compression_kernel<WV, NTV>), dim3((n + WV - 1) / WV), dim3(64), 0, stream,   \
                     d_recs, n, d_out, pitch, d_table, nwit, d_pub, d_status)
    switch (variant) {
      case 0: B3W_LAUNCH_COMP(4, false); break;
      case 1: B3W_LAUNCH_COMP(1, false); break;
      case 2: B3W_LAUNCH_COMP(2, false); break;
      case 3: B3W_LAUNCH_COMP(8, false); break;
      case 4: B3W_LAUNCH_COMP(4, true); break;
      case 5: B3W_LAUNCH_COMP(2, true); break;
      case 6: B3W_LAUNCH_COMP(1, true); break;
      case 7: B3W_LAUNCH_COMP(16, false); break;
      default: return -1;
    }
#undef B3W_LAUNCH_COMP
    return (int)hipGetLastError();
  }
  return -2;
}
