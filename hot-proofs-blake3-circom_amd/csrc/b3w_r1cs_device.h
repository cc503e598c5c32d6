// b3w_r1cs_device.h — device code the formulations of the constraint check share (b3w_r1cs.hip: gather, lean pair, stream;
// b3w_r1cs_walk.hip: the walk kernel): 256-bit field arithmetic, a row by the general road, the deferred rows of a (body, tile),
// the 8-byte element form, the loads and the barrier of the streaming kernels.  Everything is static to its translation unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "b3w_kernels.h"

namespace {
typedef uint32_t b3w_u32x4 __attribute__((ext_vector_type(4)));      // (what a 16-byte global load returns: ldg16)

struct Fe { uint32_t l[8]; };

__device__ __forceinline__ bool fe_geq(const Fe &a, const uint32_t p[8]) {
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    if (a.l[i] != p[i]) return a.l[i] > p[i];
  }
  return true;
}

__device__ __forceinline__ void fe_sub_p(Fe &a, const uint32_t p[8]) {
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)a.l[i] - p[i] - br;
    a.l[i] = (uint32_t)t;
    br = (t >> 63) & 1;
  }
}

// a = a + b mod p (a, b < p)
__device__ __forceinline__ void fe_add(Fe &a, const Fe &b, const uint32_t p[8]) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)a.l[i] + b.l[i] + c;
    a.l[i] = (uint32_t)t;
    c = t >> 32;
  }
  if (c || fe_geq(a, p)) fe_sub_p(a, p);
}

// a = a - b mod p (a, b < p)
__device__ __forceinline__ void fe_sub(Fe &a, const Fe &b, const uint32_t p[8]) {
  uint64_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)a.l[i] - b.l[i] - br;
    a.l[i] = (uint32_t)t;
    br = (t >> 63) & 1;
  }
  if (br) {
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint64_t t = (uint64_t)a.l[i] + p[i] + c;
      a.l[i] = (uint32_t)t;
      c = t >> 32;
    }
  }
}

__device__ __forceinline__ bool fe_is_zero(const Fe &a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) o |= a.l[i];
  return o == 0;
}

// a * b / 2^256 mod p (a, b < p): coarsely integrated operand scanning
__device__ __forceinline__ Fe mont_mul(const Fe &a, const Fe &b, const B3wField &F) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint64_t s = (uint64_t)a.l[j] * b.l[i] + t[j] + c;
      t[j] = (uint32_t)s;
      c = s >> 32;
    }
    uint64_t s = (uint64_t)t[8] + c;
    t[8] = (uint32_t)s;
    t[9] = (uint32_t)(s >> 32);
    const uint32_t m = t[0] * F.inv;
    c = ((uint64_t)m * F.p[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) {
      s = (uint64_t)m * F.p[j] + t[j] + c;
      t[j - 1] = (uint32_t)s;
      c = s >> 32;
    }
    s = (uint64_t)t[8] + c;
    t[7] = (uint32_t)s;
    t[8] = t[9] + (uint32_t)(s >> 32);
  }
  Fe r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = t[i];
  if (t[8] || fe_geq(r, F.p)) fe_sub_p(r, F.p);
  return r;
}

// one element of a body; *wild is set when it is not a canonical representative (>= p)
__device__ __forceinline__ Fe load_z(const uint8_t *body, uint32_t wire, const B3wField &F, bool *wild) {
  const uint4 *q = reinterpret_cast<const uint4 *>(body + (size_t)wire * 32);
  const uint4 lo = q[0], hi = q[1];
  Fe z;
  z.l[0] = lo.x; z.l[1] = lo.y; z.l[2] = lo.z; z.l[3] = lo.w;
  z.l[4] = hi.x; z.l[5] = hi.y; z.l[6] = hi.z; z.l[7] = hi.w;
  if (fe_geq(z, F.p)) {
    *wild = true;
    do fe_sub_p(z, F.p); while (fe_geq(z, F.p));         // keep the arithmetic below in range all the same
  }
  return z;
}

__device__ __forceinline__ Fe load_fe(const uint32_t *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  const uint4 lo = q[0], hi = q[1];
  Fe r;
  r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w;
  r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
  return r;
}

// 0, 1 or 2 = "something else"
__device__ __forceinline__ int small01(const Fe &a) {
  uint32_t hi = 0;
#pragma unroll
  for (int i = 1; i < 8; i++) hi |= a.l[i];
  return (hi | (a.l[0] >> 1)) ? 2 : (int)a.l[0];
}

// <row, z>: `n` terms starting at `off`; term = wire | coefficient id (0: +1, 1: -1, else index into the tables:
// coefs[16 * cid ..] = the coefficient, then the coefficient * 2^256 mod p)
__device__ __forceinline__ Fe dot(const uint8_t *body, const uint32_t *wires, const uint16_t *cids, const uint32_t *coefs,
                                  uint32_t off, uint32_t n, const B3wField &F, bool *wild, uint32_t start = 0, uint32_t step = 1) {
  Fe acc;
#pragma unroll
  for (int i = 0; i < 8; i++) acc.l[i] = 0;
  for (uint32_t k = start; k < n; k += step) {               // (start, step: a lane's share when a wave splits a long row)
    const uint32_t w = wires[off + k];
    const uint32_t cid = cids[off + k];
    const Fe z = load_z(body, w, F, wild);
    const int zs = small01(z);
    if (zs == 0) continue;                               // coef * 0
    if (cid == 0) fe_add(acc, z, F.p);
    else if (cid == 1) fe_sub(acc, z, F.p);
    else if (zs == 1) {                                  // coef * 1
      const Fe cf = load_fe(coefs + (size_t)cid * 16);
      fe_add(acc, cf, F.p);
    } else {
      const Fe cf = load_fe(coefs + (size_t)cid * 16 + 8);
      const Fe t = mont_mul(cf, z, F);                   // (coef * R) * z / R
      fe_add(acc, t, F.p);
    }
  }
  return acc;
}

// a (< p) as a small signed number: a = k or a = p - k with k < 2^32?
__device__ __forceinline__ bool small_signed(const Fe &a, const uint32_t p[8], uint32_t *k, bool *neg) {
  uint32_t hi = 0;
#pragma unroll
  for (int i = 1; i < 8; i++) hi |= a.l[i];
  if (!hi) { *k = a.l[0]; *neg = false; return true; }
  uint32_t d0 = 0;
  uint64_t br = 0;
  hi = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)p[i] - a.l[i] - br;
    if (i == 0) d0 = (uint32_t)t; else hi |= (uint32_t)t;
    br = (t >> 63) & 1;
  }
  *k = d0; *neg = true;
  return hi == 0;
}
// (+-k) * w = c mod p, for w, c < p, without a field multiplication: k * w -+ c is an integer s in (-p, (k + 1) p), and s = 0 mod p
// iff s = q p for the one q < 2^32 with q = s / p mod 2^32 (p is odd).  (The 134 rows of a nova step's 67 IsZero gadgets are
// "in * inv = 1 - out" and "in * out = 0" with in = depth - i, a small signed number, and inv a full field element: all of them
// deferred, and two Montgomery products each — 600 vector instructions — without this.)
__device__ __forceinline__ bool small_product_is(uint32_t k, bool neg, const Fe &w, const Fe &c, const B3wField &F) {
  uint32_t sgn[9];
  uint64_t cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)k * w.l[i] + cy;
    sgn[i] = (uint32_t)t;
    cy = t >> 32;
  }
  sgn[8] = (uint32_t)cy;
  if (neg) {                                               // s = k w + c
    cy = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint64_t t = (uint64_t)sgn[i] + c.l[i] + cy;
      sgn[i] = (uint32_t)t;
      cy = t >> 32;
    }
    sgn[8] += (uint32_t)cy;                                // (k w + c < 2^32 p: nine limbs hold it)
  } else {                                                 // s = k w - c
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint64_t t = (uint64_t)sgn[i] - c.l[i] - br;
      sgn[i] = (uint32_t)t;
      br = (t >> 63) & 1;
    }
    if (br > sgn[8]) return false;                         // s < 0 (and > -p): no multiple of p
    sgn[8] -= (uint32_t)br;
  }
  const uint32_t q = sgn[0] * (0u - F.inv);                // F.inv = -1 / p mod 2^32
  uint32_t diff = 0;
  cy = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const uint64_t t = (uint64_t)q * F.p[i] + cy;
    diff |= (uint32_t)t ^ sgn[i];
    cy = t >> 32;
  }
  diff |= (uint32_t)cy ^ sgn[8];
  return diff == 0;
}

// A z * B z = C z for the three sums of a row?  (linear: the row has no A or no B terms: 0 * B - C = 0)
__device__ __forceinline__ bool row_violated(const Fe &az, const Fe &bz, const Fe &cz, bool linear, const B3wField &F) {
  if (linear) return !fe_is_zero(cz);
  const int as = small01(az), bs = small01(bz);
  Fe ab;
  if (as == 0 || bs == 0) {                                                          // 0 * x
#pragma unroll
    for (int i = 0; i < 8; i++) ab.l[i] = 0;
  } else if (as == 1) ab = bz;                                                       // 1 * x
  else if (bs == 1) ab = az;
  else {
    uint32_t k = 0;
    bool neg = false;
    if (small_signed(az, F.p, &k, &neg)) return !small_product_is(k, neg, bz, cz, F);
    if (small_signed(bz, F.p, &k, &neg)) return !small_product_is(k, neg, az, cz, F);
    Fe r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.l[i] = F.r2[i];
    ab = mont_mul(mont_mul(az, r2, F), bz, F);                                       // (az * R) * bz / R = az * bz
  }
  Fe diff = ab;
  fe_sub(diff, cz, F.p);
  return !fe_is_zero(diff);
}

// one row of the gather formulation over one body: violated?  (d = first term, terms in A, B, C)
__device__ __forceinline__ bool gather_row(const uint8_t *body, const uint4 d, const uint32_t *wires, const uint16_t *cids,
                                           const uint32_t *coefR, const B3wField &F) {
  bool wild = false;
  const Fe cz = dot(body, wires, cids, coefR, d.x + d.y + d.z, d.w, F, &wild);
  // (a linear row's other part is still read, for the canonical-form check)
  const Fe az = dot(body, wires, cids, coefR, d.x, d.y, F, &wild);
  const Fe bz = dot(body, wires, cids, coefR, d.x + d.y, d.z, F, &wild);
  return row_violated(az, bz, cz, d.y == 0 || d.z == 0, F) || wild;                  // an element >= p is no witness value, whatever it is congruent to
}

// the same with the row's terms dealt to the 64 lanes of the wave (all lanes call it with the same d; all get the verdict)
__device__ __forceinline__ Fe wave_sum(Fe v, const uint32_t p[8]) {
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) {
    Fe o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.l[i] = (uint32_t)__shfl_xor((int)v.l[i], sh);
    fe_add(v, o, p);
  }
  return v;
}
// The row's three sums in ONE pass over its terms (A | B | C lie one behind the other in the term list): a lane takes every 64th
// term, whichever part it falls into.  Element and coefficient are loaded side by side (dot() asks for the coefficient only once it
// has seen the element: one more dependent load), and the next term's loads are in flight while this one is multiplied — a nova
// step's always-deferred row (66 + 67 terms with field-sized coefficients) is three rounds of two load levels, not four of three.
__device__ __forceinline__ bool gather_row_wave(const uint8_t *body, const uint4 d, const uint32_t *wires, const uint16_t *cids,
                                                const uint32_t *coefR, const B3wField &F) {
  bool wild = false;
  const uint32_t lane = threadIdx.x & 63u, n = d.y + d.z + d.w;
  Fe az, bz, cz;
#pragma unroll
  for (int i = 0; i < 8; i++) az.l[i] = bz.l[i] = cz.l[i] = 0;
  uint32_t cid = 0;
  uint4 zlo = make_uint4(0, 0, 0, 0), zhi = zlo;
  Fe cf;
  auto issue = [&](uint32_t k) {                            // (k < n)
    const uint32_t w = wires[d.x + k];
    cid = cids[d.x + k];
    const uint4 *q = reinterpret_cast<const uint4 *>(body + (size_t)w * 32);
    zlo = q[0]; zhi = q[1];
    cf = load_fe(coefR + (size_t)cid * 16 + 8);             // (coef * R; entries 0 and 1 — plus and minus one — are not used through it)
  };
  if (lane < n) issue(lane);
  for (uint32_t k = lane; k < n; k += 64u) {
    Fe z;
    z.l[0] = zlo.x; z.l[1] = zlo.y; z.l[2] = zlo.z; z.l[3] = zlo.w;
    z.l[4] = zhi.x; z.l[5] = zhi.y; z.l[6] = zhi.z; z.l[7] = zhi.w;
    const uint32_t c = cid;
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
    if (k < d.y) fe_add(az, t, F.p);
    else if (k < d.y + d.z) fe_add(bz, t, F.p);
    else fe_add(cz, t, F.p);
  }
  az = wave_sum(az, F.p);
  bz = wave_sum(bz, F.p);
  if (d.w) cz = wave_sum(cz, F.p);                          // (wave-uniform)
  return row_violated(az, bz, cz, d.y == 0 || d.z == 0, F) || __ballot(wild) != 0;
}

__device__ __forceinline__ unsigned long long lean_pack(const uint4 lo, const uint4 hi) {
  const uint32_t wide = lo.z | lo.w | hi.x | hi.y | hi.z | hi.w | (lo.y & 0x80000000u);
  return wide ? 0x8000000000000000ull : (unsigned long long)lo.x | (unsigned long long)lo.y << 32;
}

// a = {elements 0 ... 31: low halves | high halves}, b = {elements 32 ... 63 likewise}  ->  a = low halves of 0 ... 63, b = high halves
__device__ __forceinline__ void halves_apart(uint32_t &a, uint32_t &b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);           // lanes 32 ... 63 of a <-> lanes 0 ... 31 of b
  a = r[0]; b = r[1];
}
// 16 bytes per lane into registers: base + off, non-temporal (read once) or not
template <bool NT>
__device__ __forceinline__ uint4 ldg16(const uint8_t *base /* wave-uniform */, uint32_t off /* per lane */) {
  const b3w_u32x4 *p = reinterpret_cast<const b3w_u32x4 *>(base + off);
  b3w_u32x4 v;
  if constexpr (NT) v = __builtin_nontemporal_load(p);
  else v = *p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
// a part sum {low, high} = sext(low) + sext(high) * 2^52, as a 128-bit two's complement number in two halves
__device__ __forceinline__ void part_sum(unsigned long long s_lo, unsigned long long s_hi, unsigned long long &lo, long long &hi) {
  lo = s_lo + (s_hi << 52);
  hi = ((long long)s_lo >> 63) + ((long long)s_hi >> 12) + (lo < s_lo ? 1ll : 0ll);
}
// workgroup barrier that leaves vector-memory operations (the fetches) in flight: LDS traffic retired, then s_barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the rows the lean kernel left: one WAVE per (body, tile), almost all of which leave on their first load.  A wave, not the
// lean kernel's four: a row of a flagged tile is a chain of dependent loads (term, element, coefficient), and sixteen single-wave
// workgroups fit a CU where four-wave ones fit four (354 -> 217 us per 4 096 nova bodies).  A lane takes a row; a LONG row (each
// derived nova system has one of 66 ... 133 terms whose coefficients, 2^70 and more, are no small integers: always deferred, and
// 200 us of dependent loads on one lane) is dealt to all 64 lanes instead.
#define B3W_R1CS_DEFERRED_TILES 1u
// the deferred rows of one (body, tile), by one wave: its scratch block says which (word 0: 0 = none; sparse — the stream and walk
// kernels' blocks: bit w = mask word w was stored and is not zero; not sparse — the lean kernel's: every mask word was stored)
// (SYS: B3wR1csSystem, or a kernel's own view of it with the members used here — tiles, ntiles, g_rows, g_wires, g_cids, coefs, row_k,
// row_id: a kernel argument of twenty pointers is twenty pairs of scalar registers the compiler keeps alive)
template <class SYS>
__device__ __forceinline__ bool deferred_tile(const uint8_t *body, const uint32_t b, const uint32_t tile, const SYS &S,
                                              const unsigned long long *__restrict__ scratch, const uint32_t block_words, const B3wField &F, const bool sparse,
                                              uint32_t &nbad, uint32_t &low, const uint32_t wave = 0u, const uint32_t nwaves = 1u) {      // (wave w of nwaves takes mask words w, w + nwaves, ...)
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long *block = scratch + ((size_t)b * S.ntiles + tile) * block_words;
  const uint4 td = reinterpret_cast<const uint4 *>(S.tiles)[tile];
  const uint32_t words = (td.y + 63u) >> 6;              // word w of the first kernel = rows first + 64 * w + lane (at most 64 words: 4 096 rows)
  const unsigned long long head = block[0];
  if (head == 0ull) return false;                        // (wave-uniform: one address)
  const unsigned long long mine_w = lane < words && (!sparse || ((head >> lane) & 1ull)) ? block[1 + lane] : 0ull;
  if (__ballot(mine_w != 0ull) == 0ull) return false;
  for (uint32_t wi = wave; wi < words; wi += nwaves) {
    const unsigned long long mask = __shfl(mine_w, (int)wi);
    if (mask == 0) continue;                             // (wave-uniform)
    const bool mine = (mask >> lane) & 1ull;
    const uint32_t r = td.x + 64u * wi + lane;    // (< td.x + td.y for a marked lane: only such lanes set a bit)
    uint4 d = make_uint4(0, 0, 0, 0);
    if (mine) d = reinterpret_cast<const uint4 *>(S.g_rows)[S.row_k[r]];
    const bool is_long = mine && d.y + d.z + d.w > 24u;
    if (mine && !is_long && gather_row(body, d, S.g_wires, S.g_cids, S.coefs, F)) { nbad++; low = min(low, S.row_id[r]); }
    unsigned long long longs = __ballot(is_long);
    while (longs) {                                      // (wave-uniform)
      const int L = __ffsll((long long)longs) - 1;
      longs &= longs - 1ull;
      const uint4 dl = make_uint4((uint32_t)__shfl((int)d.x, L), (uint32_t)__shfl((int)d.y, L), (uint32_t)__shfl((int)d.z, L),
                                  (uint32_t)__shfl((int)d.w, L));
      const bool bad = gather_row_wave(body, dl, S.g_wires, S.g_cids, S.coefs, F);
      if ((int)lane == L && bad) { nbad++; low = min(low, S.row_id[r]); }
    }
  }
  return true;
}
__device__ __forceinline__ void deferred_report(uint32_t nbad, uint32_t low, const uint32_t b, uint32_t *__restrict__ violations, uint32_t *__restrict__ first) {
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) {
    nbad += (uint32_t)__shfl_xor((int)nbad, sh);
    low = min(low, (uint32_t)__shfl_xor((int)low, sh));
  }
  if ((threadIdx.x & 63u) == 0 && nbad) {
    atomicAdd(&violations[b], nbad);
    if (first) atomicMin(&first[b], low);
  }
}
}  // namespace
