// b3w_plan.hip — step-input planner for chained ("nova fold") mode, on the device.
//
// The reference folds ONE chunk path per proof, sequentially: every step's inputs come from the
// previous step's outputs (rust_fold/src/main.rs:166-179; Blake3BlockCompressCircuit::format_input /
// update_for_step, rust_fold/src/blake3_circuit.rs:160-289; sibling CVs from bao in
// rust_fold/src/blake3_hash.rs:17-93).  The chaining value is plain BLAKE3, so a native pre-pass
// produces the input record of EVERY step up front and all step witnesses become independent
// (SURVEY.md §3.3): that is what these kernels do, for all chunks of a preimage at once.
//
//   b3w_plan_leaf_kernel    one thread per 1 KiB chunk: walks its <= 16 blocks, emits one nova step
//                           record per block (n_blocks, block_count, h = running CV, chunk index,
//                           depths, message words, b) and the chunk's chaining value
//   b3w_plan_merge_kernel   one tree level: parent CV = compress(IV, left || right, PARENT [| ROOT])
//   b3w_plan_paths_kernel   one thread per chunk: the parent steps of its path, bottom up, for ANY chunk count, planned
//                           the way the reference's driver plans them (h = the chain's running value, m[0..7] = the
//                           PathNode's sibling, depth; rust_fold/src/blake3_hash.rs:58-84, blake3_circuit.rs:230-245)
//
// Record layout = the batch input format of the nova kernels (b3wit.h).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include "b3w_kernels.h"

namespace {

__device__ __forceinline__ uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

#define B3_G(a, b, c, d, x, y)                                       \
  v[a] = v[a] + v[b] + (x); v[d] = rotr(v[d] ^ v[a], 16);            \
  v[c] = v[c] + v[d];       v[b] = rotr(v[b] ^ v[c], 12);            \
  v[a] = v[a] + v[b] + (y); v[d] = rotr(v[d] ^ v[a], 8);             \
  v[c] = v[c] + v[d];       v[b] = rotr(v[b] ^ v[c], 7);

// plain BLAKE3 compression, first 8 output words (BLAKE3 spec 2.2; the circuit's Blake3Compression
// computes the same function, circuits/blake3_compression.circom:171-228)
__device__ void blake3_cv(const uint32_t h[8], const uint32_t m_in[16], uint32_t t0, uint32_t t1, uint32_t b, uint32_t d,
                          uint32_t out[8]) {
  uint32_t v[16], m[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = h[i];
  v[8] = 0x6A09E667u; v[9] = 0xBB67AE85u; v[10] = 0x3C6EF372u; v[11] = 0xA54FF53Au;
  v[12] = t0; v[13] = t1; v[14] = b; v[15] = d;
#pragma unroll
  for (int i = 0; i < 16; ++i) m[i] = m_in[i];
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    B3_G(0, 4, 8, 12, m[0], m[1]) B3_G(1, 5, 9, 13, m[2], m[3]) B3_G(2, 6, 10, 14, m[4], m[5]) B3_G(3, 7, 11, 15, m[6], m[7])
    B3_G(0, 5, 10, 15, m[8], m[9]) B3_G(1, 6, 11, 12, m[10], m[11]) B3_G(2, 7, 8, 13, m[12], m[13]) B3_G(3, 4, 9, 14, m[14], m[15])
    const uint32_t t[16] = {m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]};
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = t[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = v[i] ^ v[i + 8];
}

__device__ __forceinline__ void iv(uint32_t h[8]) {
  h[0] = 0x6A09E667u; h[1] = 0xBB67AE85u; h[2] = 0x3C6EF372u; h[3] = 0xA54FF53Au;
  h[4] = 0x510E527Fu; h[5] = 0x9B05688Cu; h[6] = 0x1F83D9ABu; h[7] = 0x5BE0CD19u;
}

// number of parent nodes above chunk c in BLAKE3's tree over n chunks (left subtree = largest power
// of two strictly below n)
__host__ __device__ inline uint32_t path_len(uint64_t c, uint64_t n) {
  uint32_t p = 0;
  while (n > 1) {
    uint64_t k = 1;
    while (k * 2 < n) k *= 2;
    if (c < k) n = k; else { c -= k; n -= k; }
    p++;
  }
  return p;
}

// ---- the leaf planner, four lanes per chunk (r05) ------------------------------------------------------------------
// A chunk's 16 blocks chain through the chaining value: 16 compressions one after the other, 33 us with one thread per chunk — in
// front of the first witness kernel of every pass (7 % of a rank's share of a 1 MiB pass at 8 ranks).  Four lanes share a
// compression the way the witness kernels' TRACE phase does: lane `col` of a quad holds column col of the state, the diagonal
// step is the column step after a quad rotate (DPP), the block's 16 message words lie in LDS and every lane picks the two a G needs
// by the round's schedule.  Same records, a third of the latency.
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ uint32_t plan_quad_perm(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, false);
}
__device__ __forceinline__ void plan_g(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, uint32_t x, uint32_t y) {
  a = a + b + x; d = rotr(d ^ a, 16);
  c = c + d;     b = rotr(b ^ c, 12);
  a = a + b + y; d = rotr(d ^ a, 8);
  c = c + d;     b = rotr(b ^ c, 7);
}
// message schedule of round r, 4 bits per entry: round r uses m[PERM_r[j]] in place of m[j]
__host__ __device__ constexpr uint64_t plan_sched(int r) {
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
template <int R>
__device__ __forceinline__ void plan_quad_round(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, const uint32_t *M, int col) {
  constexpr uint64_t P = plan_sched(R);
  plan_g(a, b, c, d, M[(P >> (8 * col)) & 15], M[(P >> (8 * col + 4)) & 15]);
  b = plan_quad_perm<1, 2, 3, 0>(b); c = plan_quad_perm<2, 3, 0, 1>(c); d = plan_quad_perm<3, 0, 1, 2>(d);
  plan_g(a, b, c, d, M[(P >> (32 + 8 * col)) & 15], M[(P >> (36 + 8 * col)) & 15]);
  b = plan_quad_perm<3, 0, 1, 2>(b); c = plan_quad_perm<2, 3, 0, 1>(c); d = plan_quad_perm<1, 2, 3, 0>(d);
}
// lane col: in h_lo = h[col], h_hi = h[4 + col]; out the same words of the compression's first eight output words
__device__ __forceinline__ void plan_quad_cv(uint32_t &h_lo, uint32_t &h_hi, const uint32_t *M, int col, uint32_t t0, uint32_t t1, uint32_t b, uint32_t dflag) {
  const uint32_t IVc = col == 0 ? 0x6A09E667u : col == 1 ? 0xBB67AE85u : col == 2 ? 0x3C6EF372u : 0xA54FF53Au;
  uint32_t a = h_lo, bb = h_hi, c = IVc, d = col == 0 ? t0 : col == 1 ? t1 : col == 2 ? b : dflag;
  plan_quad_round<0>(a, bb, c, d, M, col); plan_quad_round<1>(a, bb, c, d, M, col); plan_quad_round<2>(a, bb, c, d, M, col);
  plan_quad_round<3>(a, bb, c, d, M, col); plan_quad_round<4>(a, bb, c, d, M, col); plan_quad_round<5>(a, bb, c, d, M, col);
  plan_quad_round<6>(a, bb, c, d, M, col);
  h_lo = a ^ c;
  h_hi = bb ^ d;
}

__global__ __launch_bounds__(64) void b3w_plan_leaf_quad_kernel(const uint8_t *__restrict__ pre /* at chunk first_chunk */, uint64_t total_len, uint64_t first_chunk,
                                                                uint32_t nlocal, uint64_t nchunks, uint32_t *__restrict__ recs, uint32_t *__restrict__ chunk_cv) {
  __shared__ uint32_t Ms[16][16];                             // the current block's message words, per chunk of the workgroup
  const uint32_t q = threadIdx.x >> 2, col = threadIdx.x & 3u;
  const uint32_t i = blockIdx.x * 16u + q;
  const bool live = i < nlocal;                               // (a dead quad walks chunk 0 of the workgroup's range with stores masked: the DPP moves want whole quads)
  const uint32_t ii = live ? i : blockIdx.x * 16u;
  const uint64_t c = first_chunk + ii;
  const uint64_t off = c * 1024;
  const uint32_t bytes = (uint32_t)(total_len - off < 1024 ? total_len - off : 1024);
  const uint32_t n_blocks = bytes ? (bytes + 63) / 64 : 1;
  const uint32_t P = path_len(c, nchunks);
  const uint8_t *src = pre + (uint64_t)ii * 1024;
  uint32_t *rec = recs + (uint64_t)ii * 16 * 32;
  uint32_t *M = Ms[q];
  uint32_t h_lo = col == 0 ? 0x6A09E667u : col == 1 ? 0xBB67AE85u : col == 2 ? 0x3C6EF372u : 0xA54FF53Au;
  uint32_t h_hi = col == 0 ? 0x510E527Fu : col == 1 ? 0x9B05688Cu : col == 2 ? 0x1F83D9ABu : 0x5BE0CD19u;
  for (uint32_t j = 0; j < n_blocks; ++j) {                   // (uniform over the quad; quads of a wave with fewer blocks idle)
    const uint32_t bb = bytes - j * 64 < 64 ? bytes - j * 64 : 64;
    uint32_t m4[4];
    if (bb == 64 && ((uintptr_t)src & 15) == 0) {
      const uint4 v = *reinterpret_cast<const uint4 *>(src + j * 64 + col * 16);
      m4[0] = v.x; m4[1] = v.y; m4[2] = v.z; m4[3] = v.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t w = 0;
        for (int x = 0; x < 4; ++x) { const uint32_t p = (col * 4 + k) * 4 + x; if (p < bb) w |= (uint32_t)src[j * 64 + p] << (8 * x); }
        m4[k] = w;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) M[col * 4 + k] = m4[k];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wave: the quad's words are in LDS)
    if (live) {
      uint32_t *r = rec + j * 32;
      if (col == 0) { r[0] = n_blocks; r[1] = j; r[10] = (uint32_t)c; r[11] = (uint32_t)(c >> 32); }
      if (col == 1) { r[12] = P + 1; r[13] = P + 1; r[14] = P; r[31] = bb; }      // leaf_depth, total_depth, depth (blake3_circuit.rs:83-110)
      r[2 + col] = h_lo;
      r[6 + col] = h_hi;
#pragma unroll
      for (int k = 0; k < 4; ++k) r[15 + col * 4 + k] = m4[k];
    }
    // flags as Blake3GetFlag assigns them (circuits/blake3_nova.circom:122-167)
    const uint32_t last = j == n_blocks - 1;
    const uint32_t d = (j == 0 ? 1u : 0u) | (last ? 2u : 0u) | ((last && P == 0) ? 8u : 0u);
    plan_quad_cv(h_lo, h_hi, M, (int)col, (uint32_t)c, (uint32_t)(c >> 32), bb, d);
    __builtin_amdgcn_wave_barrier();                          // (every lane has read this block's words before the next block's overwrite them)
  }
  if (live) { chunk_cv[(uint64_t)i * 8 + col] = h_lo; chunk_cv[(uint64_t)i * 8 + 4 + col] = h_hi; }
}

__global__ __launch_bounds__(64) void b3w_plan_leaf_kernel(const uint8_t *__restrict__ pre /* at chunk first_chunk */,
                                                           uint64_t total_len, uint64_t first_chunk, uint32_t nlocal,
                                                           uint64_t nchunks, uint32_t *__restrict__ recs,
                                                           uint32_t *__restrict__ chunk_cv) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= nlocal) return;
  const uint64_t c = first_chunk + i;
  const uint64_t off = c * 1024;
  const uint32_t bytes = (uint32_t)(total_len - off < 1024 ? total_len - off : 1024);
  const uint32_t n_blocks = bytes ? (bytes + 63) / 64 : 1;
  const uint32_t P = path_len(c, nchunks);
  const uint8_t *src = pre + (uint64_t)i * 1024;
  uint32_t *rec = recs + (uint64_t)i * 16 * 32;          // chunks before the last are full: 16 steps each
  uint32_t h[8];
  iv(h);
  for (uint32_t j = 0; j < n_blocks; ++j) {
    const uint32_t bb = bytes - j * 64 < 64 ? bytes - j * 64 : 64;
    uint32_t m[16];
    if (bb == 64 && ((uintptr_t)src & 3) == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) m[k] = reinterpret_cast<const uint32_t *>(src + j * 64)[k];
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        uint32_t w = 0;
        for (int q = 0; q < 4; ++q) { const uint32_t p = k * 4 + q; if (p < bb) w |= (uint32_t)src[j * 64 + p] << (8 * q); }
        m[k] = w;
      }
    }
    uint32_t *r = rec + j * 32;
    r[0] = n_blocks; r[1] = j;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[2 + k] = h[k];
    r[10] = (uint32_t)c; r[11] = (uint32_t)(c >> 32);
    r[12] = P + 1; r[13] = P + 1; r[14] = P;              // leaf_depth, total_depth, depth (blake3_circuit.rs:83-110)
#pragma unroll
    for (int k = 0; k < 16; ++k) r[15 + k] = m[k];
    r[31] = bb;
    // flags as Blake3GetFlag assigns them (circuits/blake3_nova.circom:122-167)
    const uint32_t last = j == n_blocks - 1;
    const uint32_t d = (j == 0 ? 1u : 0u) | (last ? 2u : 0u) | ((last && P == 0) ? 8u : 0u);
    uint32_t o[8];
    blake3_cv(h, m, (uint32_t)c, (uint32_t)(c >> 32), bb, d, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = o[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) chunk_cv[(uint64_t)i * 8 + k] = h[k];
}

// parents[i] = compress(IV, child[2i] || child[2i+1], t = 0, b = 64, PARENT | (root ? ROOT : 0))
__global__ __launch_bounds__(64) void b3w_plan_merge_kernel(const uint32_t *__restrict__ left, const uint32_t *__restrict__ right,
                                                            uint32_t stride_words, uint64_t npairs, uint32_t root,
                                                            uint32_t *__restrict__ parents) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= npairs) return;
  uint32_t h[8], m[16], o[8];
  iv(h);
#pragma unroll
  for (int k = 0; k < 8; ++k) { m[k] = left[i * stride_words + k]; m[8 + k] = right[i * stride_words + k]; }
  blake3_cv(h, m, 0, 0, 64, 4u | (root ? 8u : 0u), o);
#pragma unroll
  for (int k = 0; k < 8; ++k) parents[i * 8 + k] = o[k];
}

// ---- parent steps of every chunk path, any chunk count ----------------------------------------------------------
// BLAKE3's tree over n chunks (left subtree = the largest power of two below the count) is a right spine of complete
// subtrees: seg[0] (the largest, chunks 0 ..), seg[1], ... seg[last]; suffix[i] = the node over seg[i] .. seg[last]
// (suffix[0] = the root, suffix[last] = seg[last], suffix[i] = parent(seg[i], suffix[i+1])).  The level arrays of
// b3w_chain_tree_device hold every complete subtree (level t, node index), its scratch chain the suffix nodes.
//
// The step circuit takes left/right at height g from bit g of chunk_idx (Blake3GetDownLeftPath,
// circuits/blake3_nova.circom:47-84), and the reference's driver picks the PathNode's sibling by the same bit
// (blake3_hash.rs:63-78: bit clear -> the node's RIGHT child CV, bit set -> its LEFT child CV).  That is the leaf's true
// sibling exactly when the leaf's real path spells the low bits of its index — always in a complete tree; in an
// incomplete one only for some leaves (those of seg[0], and of later segments whose position happens to agree).  For
// the others the reference hands the circuit the node's other child, i.e. the path child's own CV, and the fold ends in a
// value that is not BLAKE3(input) (tests/golden/incomplete_trees.nova_vesta.json: the reference WASM driven that way).
// This kernel reproduces the reference's records for every leaf — running value computed the way the circuit does —
// and b3w_plan_path_provable says which paths end in the root.
struct B3wSpine {
  uint32_t nseg;                 // segments on the spine (1 = complete tree)
  uint32_t level[64];            // seg i = complete subtree of 2^level[i] chunks
  uint32_t plen[64];             // path length of its chunks
  uint64_t lo[64];               // first chunk
  uint64_t seg_off[64];          // word offset of the segment's CV in the levels buffer
  uint64_t suf_off[64];          // word offset of suffix[i]'s CV (i >= 1)
  uint64_t row_base[64];         // parent-step row of chunk lo[i]'s first parent step (rows of chunk 0 start at 0)
};

__host__ inline uint64_t level_off_words(uint64_t n, uint32_t t) {
  uint64_t off = 0;
  for (uint32_t l = 0; l < t; ++l) off += (n >> l) * 8;
  return off;
}

__host__ inline B3wSpine spine_of(uint64_t n) {
  B3wSpine sp{};
  // pairwise levels: an odd node out at level l is a complete subtree of 2^l chunks that waits (a "carry")
  uint32_t carry_level[64];
  uint64_t carry_node[64];
  uint32_t nc = 0, l = 0;
  uint64_t count = n;
  while (count > 1) {
    if (count & 1) { carry_level[nc] = l; carry_node[nc] = count - 1; nc++; }
    count >>= 1;
    l++;
  }
  sp.nseg = nc + 1;
  sp.level[0] = l; sp.lo[0] = 0; sp.seg_off[0] = level_off_words(n, l);
  for (uint32_t i = 0; i < nc; ++i) {                     // root-down order = decreasing size = reverse carry order
    const uint32_t k = nc - 1 - i;
    sp.level[1 + i] = carry_level[k];
    sp.lo[1 + i] = carry_node[k] << carry_level[k];
    sp.seg_off[1 + i] = level_off_words(n, carry_level[k]) + carry_node[k] * 8;
  }
  const uint32_t last = nc;
  const uint64_t scratch = 2 * n * 8;                     // b3w_chain_tree_device: suffix[last - i] = scratch[i - 1], i = 1 .. last - 1
  for (uint32_t i = 1; i < last; ++i) sp.suf_off[i] = scratch + (uint64_t)(last - i - 1) * 8;
  sp.suf_off[last] = sp.seg_off[last];
  uint64_t row = 0;
  for (uint32_t i = 0; i <= last; ++i) {
    sp.plen[i] = sp.level[i] + (last == 0 ? 0 : (i == last ? last : i + 1));
    sp.row_base[i] = row;
    row += (uint64_t)sp.plen[i] << sp.level[i];
  }
  return sp;
}

__host__ __device__ inline uint32_t seg_of(const B3wSpine &sp, uint64_t c) {
  uint32_t s = 0;
  while (s + 1 < sp.nseg && c >= sp.lo[s + 1]) s++;
  return s;
}

// the parent steps of chunk c's path, bottom up (one thread)
__device__ __forceinline__ void plan_path(const uint32_t *levels, uint64_t nchunks, const B3wSpine &sp, uint64_t c, uint32_t last_chunk_blocks,
                                          uint64_t row0, uint32_t *recs) {
  const uint32_t s = seg_of(sp, c), last = sp.nseg - 1, t = sp.level[s], plen = sp.plen[s];
  const uint32_t n_blocks = (c == nchunks - 1) ? last_chunk_blocks : 16;
  uint32_t *r = recs + (sp.row_base[s] + (c - sp.lo[s]) * plen - row0) * 32;
  uint32_t h[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) h[k] = levels[c * 8 + k];
  uint64_t off = 0, cnt = nchunks;                        // word offset of level g
  for (uint32_t g = 0; g < plen; ++g, r += 32) {
    const uint32_t *own, *sib;
    bool true_left;
    if (g < t) {                                          // inside the complete subtree
      const uint64_t node = c >> g;
      own = levels + off + node * 8;
      sib = levels + off + (node ^ 1) * 8;
      true_left = (node & 1) == 0;
      off += cnt * 8;
      cnt >>= 1;
    } else {                                              // on the spine
      const uint32_t k = g - t;
      if (s == last) {                                    // suffix[last - k] is the right child of suffix[last - k - 1]
        own = levels + sp.suf_off[last - k];
        sib = levels + sp.seg_off[last - k - 1];
        true_left = false;
      } else if (k == 0) {                                // seg[s] is the left child of suffix[s]
        own = levels + sp.seg_off[s];
        sib = levels + sp.suf_off[s + 1];
        true_left = true;
      } else {                                            // suffix[s - k + 1] is the right child of suffix[s - k]
        own = levels + sp.suf_off[s - k + 1];
        sib = levels + sp.seg_off[s - k];
        true_left = false;
      }
    }
    const bool bit_left = ((c >> g) & 1) == 0;
    const uint32_t *m8 = bit_left == true_left ? sib : own;      // blake3_hash.rs:63-78 (the PathNode's "other" child by the bit)
    const uint32_t depth = plen - 1 - g;
    r[0] = n_blocks; r[1] = n_blocks;                     // block_count stays at n_blocks on parent steps (blake3_nova.circom:251)
#pragma unroll
    for (int k = 0; k < 8; ++k) r[2 + k] = h[k];
    r[10] = (uint32_t)c; r[11] = (uint32_t)(c >> 32);
    r[12] = plen + 1; r[13] = plen + 1; r[14] = depth;
    uint32_t m[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const uint32_t w = m8[k]; r[15 + k] = w; r[23 + k] = 0; m[bit_left ? 8 + k : k] = w; }   // sibling CV, then zeros (blake3_circuit.rs:230-245)
    r[31] = 64;
    // the next step's h = this step's h_out: compress(IV, h || sibling or sibling || h, PARENT [| ROOT at depth 0])
#pragma unroll
    for (int k = 0; k < 8; ++k) m[bit_left ? k : 8 + k] = h[k];
    uint32_t ivv[8], o[8];
    iv(ivv);
    blake3_cv(ivv, m, 0, 0, 64, 4u | (depth == 0 ? 8u : 0u), o);
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = o[k];
  }
}

__global__ __launch_bounds__(64) void b3w_plan_paths_kernel(const uint32_t *__restrict__ levels, uint64_t nchunks, B3wSpine sp,
                                                            uint64_t first_chunk, uint32_t nlocal, uint32_t last_chunk_blocks,
                                                            uint64_t row0, uint32_t *__restrict__ recs) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= nlocal) return;
  plan_path(levels, nchunks, sp, first_chunk + i, last_chunk_blocks, row0, recs);
}

// ---- the upper tree in ONE launch (r05) -------------------------------------------------------------------------
// b3w_chain_tree_device used to launch one merge kernel per level (10 dependent launches for the 1 024 chunks of a 1 MiB preimage,
// plus one per carry): on a rank's share of a sharded 1 MiB pass that chain of launches was as long as the rank's leaf witness
// kernel it is supposed to hide under (profiles/r04/chain_scaling_model_1mib.json).  From the level that has at most
// B3W_TREE_NODES nodes on, ONE workgroup does every remaining level: the level lives in LDS (node i = words [8 i, 8 i + 8)), each
// round thread p merges nodes 2 p and 2 p + 1 into node p (registers between the two barriers), writes it to the level array in HBM
// as well (the path planner reads every level), an odd node out is a carry; thread 0 then folds the carries into the right spine
// and the root.  PLAN: the same workgroup goes on to plan the parent steps of the rank's chunks (one thread per chunk, after an
// agent-scope fence: the level arrays it reads were written by other waves of this workgroup) — taken when the rank has few
// chunks; with many, 16 waves on one CU are slower than the path kernel's workgroups spread over the chip.
constexpr uint32_t B3W_TREE_NODES = 1024;

template <bool PLAN>
__global__ __launch_bounds__(1024) void b3w_plan_tree_kernel(uint32_t *__restrict__ levels, uint64_t n, uint32_t l0, uint32_t *__restrict__ root,
                                                             B3wSpine sp, uint64_t first_chunk, uint32_t nlocal, uint32_t last_chunk_blocks, uint64_t row0,
                                                             uint32_t *__restrict__ recs) {
  __shared__ __attribute__((aligned(16))) uint32_t node[B3W_TREE_NODES * 8];
  __shared__ uint32_t carry[64 * 8];                        // carries this workgroup met, increasing level
  const uint32_t t = threadIdx.x;
  uint64_t off = 0;                                         // word offset of level l
  for (uint32_t l = 0; l < l0; ++l) off += (n >> l) * 8;
  uint64_t cnt = n >> l0;                                   // <= B3W_TREE_NODES
  for (uint64_t i = t; i < cnt * 8; i += 1024) node[i] = levels[off + i];
  // carries of the levels below l0 (merged by earlier launches): they wait in HBM
  uint32_t nlow = 0;
  for (uint32_t l = 0; l < l0; ++l) nlow += ((n >> l) & 1) && (n >> l) > 1 ? 1u : 0u;
  __syncthreads();
  uint32_t nc = 0;                                          // carries met here (uniform)
  while (cnt > 1) {
    const uint64_t pairs = cnt >> 1;
    const bool odd = (cnt & 1) != 0;
    const bool is_root = cnt == 2 && nlow + nc == 0;
    uint32_t o[8];
    if (t < pairs) {
      uint32_t h[8], m[16];
      iv(h);
#pragma unroll
      for (int k = 0; k < 16; ++k) m[k] = node[(uint64_t)t * 16 + k];
      blake3_cv(h, m, 0, 0, 64, 4u | (is_root ? 8u : 0u), o);
    }
    uint32_t cv[8];
    if (odd && t == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) cv[k] = node[(cnt - 1) * 8 + k];
    }
    __syncthreads();                                        // every pair is in registers: the level may be overwritten
    if (t < pairs) {
      uint32_t *dst = is_root ? root : levels + off + cnt * 8 + (uint64_t)t * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) { node[(uint64_t)t * 8 + k] = o[k]; dst[k] = o[k]; }
    }
    if (odd && t == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) carry[nc * 8 + k] = cv[k];
    }
    if (odd) nc++;
    __syncthreads();
    off += cnt * 8;
    cnt = pairs;
  }
  if (t == 0 && nlow + nc) {
    // root = P(main, P(carry_k, ... P(carry_2, carry_1))) — BLAKE3's right-leaning chain of complete subtrees, smallest first;
    // the chain's nodes go to the scratch behind the levels (suffix nodes: spine_of, b3w_plan_paths_kernel)
    uint32_t h[8], m[16], o[8], right[8];
    uint32_t *scratch = levels + 2 * n * 8;
    uint32_t seen = 0;
    uint64_t lo = 0;
    for (uint32_t l = 0; seen < nlow + nc; ++l) {
      const uint64_t c = n >> l;
      const bool has = (c & 1) && c > 1;
      if (has) {
        uint32_t cvv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) cvv[k] = l < l0 ? levels[lo + (c - 1) * 8 + k] : carry[(seen - nlow) * 8 + k];
        if (seen == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) right[k] = cvv[k];
        } else {
          iv(h);
#pragma unroll
          for (int k = 0; k < 8; ++k) { m[k] = cvv[k]; m[8 + k] = right[k]; }
          blake3_cv(h, m, 0, 0, 64, 4u, o);
#pragma unroll
          for (int k = 0; k < 8; ++k) { right[k] = o[k]; scratch[(seen - 1) * 8 + k] = o[k]; }
        }
        seen++;
      }
      lo += c * 8;
    }
    iv(h);
#pragma unroll
    for (int k = 0; k < 8; ++k) { m[k] = node[k]; m[8 + k] = right[k]; }
    blake3_cv(h, m, 0, 0, 64, 4u | 8u, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) root[k] = o[k];
  }
  if (PLAN) {
    __threadfence();                                        // agent scope: level arrays written by other waves, read below through L1 / L2
    __syncthreads();                                        // (r06: a workgroup-scope fence here changes nothing — 0.498 against 0.501 ms for rank 0's pass at 8 ranks)
    for (uint32_t i = t; i < nlocal; i += 1024) plan_path(levels, n, sp, first_chunk + i, last_chunk_blocks, row0, recs);
  }
}

}  // namespace

// ---- the fold's exchange: h_out of every step (public words 2 .. 9 of the 15 per step) -----------------------------------
namespace {
// pack: rows [row0, row0 + count) of the public-output array -> count x 8 contiguous words (what goes over the wire)
__global__ __launch_bounds__(256) void b3w_plan_pack_hout_kernel(const uint32_t *__restrict__ pub, uint64_t row0, uint64_t count,
                                                                 uint32_t *__restrict__ dst) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;          // one word per thread: coalesced stores
  if (i >= count * 8) return;
  dst[i] = pub[(row0 + (i >> 3)) * 15 + 2 + (i & 7)];
}

// unpack: the gathered per-rank blocks (each `block_words` long: leaf part at 0, parent part at `par_off` words, both padded to
// the largest shard) -> two dense arrays in global step order.  tab[r] = {leaf dst row, leaf rows, parent dst row, parent rows}.
__global__ __launch_bounds__(256) void b3w_plan_unpack_hout_kernel(const uint32_t *__restrict__ gathered, uint64_t block_words, uint64_t par_off,
                                                                   const uint64_t *__restrict__ tab, uint32_t *__restrict__ leaf_all,
                                                                   uint32_t *__restrict__ par_all) {
  const uint32_t r = blockIdx.y;
  const uint64_t l0 = tab[4 * r], ln = tab[4 * r + 1], p0 = tab[4 * r + 2], pn = tab[4 * r + 3];
  const uint32_t *src = gathered + (uint64_t)r * block_words;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < (ln + pn) * 8; i += (uint64_t)gridDim.x * 256) {
    if (i < ln * 8) { if (leaf_all) leaf_all[l0 * 8 + i] = src[i]; }
    else if (par_all) par_all[p0 * 8 + (i - ln * 8)] = src[par_off + (i - ln * 8)];
  }
}
}  // namespace

extern "C" int b3w_launch_pack_hout(const uint32_t *d_pub, uint64_t row0, uint64_t count, uint32_t *d_dst, hipStream_t stream) {
  if (!count) return 0;
  hipLaunchKernelGGL(b3w_plan_pack_hout_kernel, dim3((uint32_t)((count * 8 + 255) / 256)), dim3(256), 0, stream, d_pub, row0, count, d_dst);
  return (int)hipGetLastError();
}

extern "C" int b3w_launch_unpack_hout(const uint32_t *d_gathered, uint64_t block_words, uint64_t par_off, const uint64_t *d_tab, uint32_t nranks,
                                      uint64_t max_rows, uint32_t *d_leaf_all, uint32_t *d_par_all, hipStream_t stream) {
  if (!nranks || !max_rows) return 0;
  const uint64_t want = (max_rows * 8 + 255) / 256;
  hipLaunchKernelGGL(b3w_plan_unpack_hout_kernel, dim3((uint32_t)(want < 4096 ? want : 4096), nranks), dim3(256), 0, stream, d_gathered,
                     block_words, par_off, d_tab, d_leaf_all, d_par_all);
  return (int)hipGetLastError();
}

extern "C" uint32_t b3w_plan_path_len(uint64_t chunk, uint64_t nchunks) { return path_len(chunk, nchunks); }

extern "C" int b3w_launch_plan_leaves(const uint8_t *d_pre, uint64_t total_len, uint64_t first_chunk, uint32_t nlocal,
                                      uint64_t nchunks, uint32_t *d_recs, uint32_t *d_chunk_cv, hipStream_t stream) {
  if (!nlocal) return 0;
  // four lanes per chunk while that leaves the chip room (a third of the latency); one thread per chunk for preimages of 64 MiB and more,
  // where the planner is a throughput kernel (B3W_PLAN_QUAD=0 / 1: measurements)
  static const int env_quad = getenv("B3W_PLAN_QUAD") ? atoi(getenv("B3W_PLAN_QUAD")) : -1;
  const bool quad = env_quad >= 0 ? env_quad != 0 : nlocal <= 65536u;
  if (quad)
    hipLaunchKernelGGL(b3w_plan_leaf_quad_kernel, dim3((nlocal + 15) / 16), dim3(64), 0, stream, d_pre, total_len, first_chunk, nlocal, nchunks, d_recs, d_chunk_cv);
  else
    hipLaunchKernelGGL(b3w_plan_leaf_kernel, dim3((nlocal + 63) / 64), dim3(64), 0, stream, d_pre, total_len, first_chunk, nlocal,
                       nchunks, d_recs, d_chunk_cv);
  return (int)hipGetLastError();
}

extern "C" int b3w_launch_plan_merge(const uint32_t *d_left, const uint32_t *d_right, uint32_t stride_words, uint64_t npairs,
                                     uint32_t root, uint32_t *d_parents, hipStream_t stream) {
  if (!npairs) return 0;
  hipLaunchKernelGGL(b3w_plan_merge_kernel, dim3((uint32_t)((npairs + 63) / 64)), dim3(64), 0, stream, d_left, d_right,
                     stride_words, npairs, root, d_parents);
  return (int)hipGetLastError();
}

extern "C" int b3w_launch_plan_parents(const uint32_t *d_levels, uint64_t nchunks, uint64_t first_chunk, uint32_t nlocal,
                                       uint32_t last_chunk_blocks, uint32_t *d_recs, hipStream_t stream) {
  if (!nlocal || nchunks < 2) return 0;
  const B3wSpine sp = spine_of(nchunks);
  const uint32_t s0 = seg_of(sp, first_chunk);
  const uint64_t row0 = sp.row_base[s0] + (first_chunk - sp.lo[s0]) * sp.plen[s0];
  hipLaunchKernelGGL(b3w_plan_paths_kernel, dim3((nlocal + 63) / 64), dim3(64), 0, stream, d_levels, nchunks, sp, first_chunk, nlocal,
                     last_chunk_blocks, row0, d_recs);
  return (int)hipGetLastError();
}

// Every level of the tree from level `l0` (at most B3W_TREE_NODES nodes: b3w_plan_tree_first_level) up to the root, carries and
// spine included, in one launch; plan_nlocal > 0: the parent steps of chunks [first_chunk, + plan_nlocal) in the same launch.
extern "C" uint32_t b3w_plan_tree_first_level(uint64_t nchunks) {
  uint32_t l = 0;
  while ((nchunks >> l) > B3W_TREE_NODES) l++;
  return l;
}
extern "C" int b3w_launch_plan_tree(uint32_t *d_levels, uint64_t nchunks, uint32_t l0, uint32_t *d_root, uint64_t first_chunk, uint32_t plan_nlocal,
                                    uint32_t last_chunk_blocks, uint32_t *d_recs, hipStream_t stream) {
  if (nchunks < 2 || (nchunks >> l0) > B3W_TREE_NODES || (nchunks >> l0) < 1) return (int)hipErrorInvalidValue;
  const B3wSpine sp = spine_of(nchunks);
  if (plan_nlocal) {
    const uint32_t s0 = seg_of(sp, first_chunk);
    const uint64_t row0 = sp.row_base[s0] + (first_chunk - sp.lo[s0]) * sp.plen[s0];
    hipLaunchKernelGGL(b3w_plan_tree_kernel<true>, dim3(1), dim3(1024), 0, stream, d_levels, nchunks, l0, d_root, sp, first_chunk, plan_nlocal,
                       last_chunk_blocks, row0, d_recs);
  } else {
    hipLaunchKernelGGL(b3w_plan_tree_kernel<false>, dim3(1), dim3(1024), 0, stream, d_levels, nchunks, l0, d_root, sp, (uint64_t)0, 0u, 0u, (uint64_t)0,
                       (uint32_t *)nullptr);
  }
  return (int)hipGetLastError();
}

// parent-step row (counted from chunk 0's first parent step) of `chunk`'s first parent step; chunk == nchunks: the total
extern "C" uint64_t b3w_plan_parent_row(uint64_t chunk, uint64_t nchunks) {
  if (nchunks < 2) return 0;
  const B3wSpine sp = spine_of(nchunks);
  if (chunk >= nchunks) return sp.row_base[sp.nseg - 1] + ((uint64_t)sp.plen[sp.nseg - 1] << sp.level[sp.nseg - 1]);
  const uint32_t s = seg_of(sp, chunk);
  return sp.row_base[s] + (chunk - sp.lo[s]) * sp.plen[s];
}

// does the chunk's real path through the tree spell the low bits of its index (then, and only then, the reference's
// fold of that path ends in BLAKE3(input))?
extern "C" int b3w_plan_path_provable(uint64_t chunk, uint64_t nchunks) {
  uint32_t dirs[64], p = 0;                                // true directions, root first: 1 = left
  uint64_t c = chunk, n = nchunks;
  while (n > 1) {
    uint64_t k = 1;
    while (k * 2 < n) k *= 2;
    if (c < k) { n = k; dirs[p++] = 1; } else { c -= k; n -= k; dirs[p++] = 0; }
  }
  for (uint32_t i = 0; i < p; ++i) {
    const uint32_t g = p - 1 - i;                          // height of the node the direction is taken at
    if ((((chunk >> g) & 1) == 0) != (dirs[i] == 1)) return 0;
  }
  return 1;
}
