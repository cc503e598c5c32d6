// b3w_exact.hip — device path for ONE witness with arbitrary field-element inputs.
//
// The batch kernels (b3w_kernels.hip) cover the canonical domain (every input word < 2^32).  The
// reference's WASM also accepts values outside it in some positions (m[i] = -1,
// 2^32 <= m[i] < 2^34, huge n_blocks / block_count / total_depth, parent steps that mask h, m[8..15]
// and chunk_idx away ...) and rejects others with "Assert Failed".  b3w_calc_witness routes such
// inputs here: every signal is evaluated as a 256-bit field element with circom semantics (`>>` and
// `&` act on the canonical representative), asserts are checked in the circuit's execution order,
// and the body is written straight from the atom array through the (atom, bit) slot table.
//
// One 64-lane wave per witness; latency, not throughput, is what matters on this path:
//   lanes 0..63 : the 67 IsZero inverses (binary extended GCD, no multiplier needed)
//   lane 0      : control logic and the 112 half-G's, state and message kept in LDS
//   lanes 0..63 : expand
// Follows circuits/blake3_common.circom:142-203, circuits/blake3_compression.circom:72-228,
// circuits/blake3_nova.circom:13-267 and circomlib 2.0.5 IsZero/LessThan/Num2Bits/gates.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "b3w_atoms.h"
#include "b3w_kernels.h"

namespace {

struct F256 { uint32_t l[8]; };

__device__ __forceinline__ F256 f_small(uint64_t x) {
  F256 r;
  r.l[0] = (uint32_t)x; r.l[1] = (uint32_t)(x >> 32);
#pragma unroll
  for (int i = 2; i < 8; ++i) r.l[i] = 0;
  return r;
}
__device__ __forceinline__ uint32_t f_add_raw(F256 &r, const F256 &a, const F256 &b) {
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { c += (uint64_t)a.l[i] + b.l[i]; r.l[i] = (uint32_t)c; c >>= 32; }
  return (uint32_t)c;
}
__device__ __forceinline__ uint32_t f_sub_raw(F256 &r, const F256 &a, const F256 &b) {
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t d = (uint64_t)a.l[i] - b.l[i] - br;
    r.l[i] = (uint32_t)d;
    br = (uint32_t)(d >> 63);
  }
  return br;
}
__device__ __forceinline__ bool f_ge(const F256 &a, const F256 &b) {
  F256 t;
  return f_sub_raw(t, a, b) == 0;
}
__device__ __forceinline__ bool f_is_zero(const F256 &a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.l[i];
  return o == 0;
}
__device__ __forceinline__ bool f_eq(const F256 &a, const F256 &b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.l[i] ^ b.l[i];
  return o == 0;
}
__device__ __forceinline__ F256 f_add(const F256 &a, const F256 &b, const F256 &P) {
  F256 r;
  const uint32_t c = f_add_raw(r, a, b);
  if (c || f_ge(r, P)) { F256 t; f_sub_raw(t, r, P); return t; }
  return r;
}
__device__ __forceinline__ F256 f_sub(const F256 &a, const F256 &b, const F256 &P) {
  F256 r;
  if (f_sub_raw(r, a, b)) { F256 t; f_add_raw(t, r, P); return t; }
  return r;
}
__device__ __forceinline__ void f_shr1(F256 &a, uint32_t top) {
#pragma unroll
  for (int i = 0; i < 7; ++i) a.l[i] = (a.l[i] >> 1) | (a.l[i + 1] << 31);
  a.l[7] = (a.l[7] >> 1) | (top << 31);
}
// x/2 mod p (p odd)
__device__ __forceinline__ void f_half(F256 &x, const F256 &P) {
  if (x.l[0] & 1) { const uint32_t c = f_add_raw(x, x, P); f_shr1(x, c); }
  else f_shr1(x, 0);
}
// value < 2^n ?
__device__ __forceinline__ bool f_fits(const F256 &a, int n) {
  uint32_t o = 0;
  const int limb = n >> 5, sh = n & 31;
  if (sh) o |= a.l[limb] >> sh;
  for (int i = limb + (sh ? 1 : 0); i < 8; ++i) o |= a.l[i];
  return o == 0;
}
__device__ __forceinline__ uint32_t f_bit(const F256 &a, int i) { return (a.l[i >> 5] >> (i & 31)) & 1u; }

// inv <-- in != 0 ? 1/in : 0 (circomlib IsZero): binary extended GCD on (a, p)
__device__ F256 f_inv(const F256 &a, const F256 &P) {
  if (f_is_zero(a)) return a;
  F256 u = a, v = P, x1 = f_small(1), x2 = f_small(0);
  const F256 one = f_small(1);
  while (!f_eq(u, one) && !f_eq(v, one)) {
    while ((u.l[0] & 1) == 0) { f_shr1(u, 0); f_half(x1, P); }
    while ((v.l[0] & 1) == 0) { f_shr1(v, 0); f_half(x2, P); }
    if (f_ge(u, v)) { F256 t; f_sub_raw(t, u, v); u = t; x1 = f_sub(x1, x2, P); }
    else { F256 t; f_sub_raw(t, v, u); v = t; x2 = f_sub(x2, x1, P); }
  }
  return f_eq(u, one) ? x1 : x2;
}

__device__ __forceinline__ F256 at_get(const uint32_t *at, int atom) {
  F256 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = at[atom * 8 + i];
  return r;
}
__device__ __forceinline__ void at_put(uint32_t *at, int atom, const F256 &v) {
#pragma unroll
  for (int i = 0; i < 8; ++i) at[atom * 8 + i] = v.l[i];
}
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

// assert sites (status detail): code in bits 0..7, round/g/half in 8.. (compression sites)
#define SITE(code, r, g, hf) ((uint32_t)(code) | ((uint32_t)(r) << 8) | ((uint32_t)(g) << 12) | ((uint32_t)(hf) << 16))

// Blake3Compression on atoms H M T B D (already in `at`); v[] and msg[] scratch in LDS.  Lane 0 only.
// Returns 0 or the site of the first failed assert, in the circuit's execution order
// (HalfFunG: add1 -> rxor2.tb -> add3 -> rxor4.tb, circuits/blake3_compression.circom:89-94).
__device__ uint32_t exact_compression(uint32_t *at, uint32_t *vbuf /* 32 x 8 words */, const F256 &P) {
  const uint32_t IVW[4] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au};
  const uint8_t sigma[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
  const uint8_t gidx[8][4] = {{0, 4, 8, 12}, {1, 5, 9, 13}, {2, 6, 10, 14}, {3, 7, 11, 15},
                              {0, 5, 10, 15}, {1, 6, 11, 12}, {2, 7, 8, 13}, {3, 4, 9, 14}};
  uint32_t *v = vbuf, *msg = vbuf + 16 * 8;
  at_put(at, B3W_A_ONE, f_small(1));
  for (int i = 0; i < 8; ++i) at_put(v, i, at_get(at, B3W_A_H + i));
  for (int i = 0; i < 4; ++i) at_put(v, 8 + i, f_small(IVW[i]));
  for (int i = 0; i < 4; ++i) at_put(v, 12 + i, at_get(at, B3W_A_T + i));      // T0 T1 B D
  uint8_t perm[16];
  for (int i = 0; i < 16; ++i) { perm[i] = (uint8_t)i; at_put(msg, i, at_get(at, B3W_A_M + i)); }
  for (int r = 0; r < 7; ++r) {
    for (int g = 0; g < 8; ++g) {
      const int a = gidx[g][0], b = gidx[g][1], c = gidx[g][2], d = gidx[g][3];
      for (int hf = 0; hf < 2; ++hf) {
        const int R1 = hf ? 8 : 16, R2 = hf ? 7 : 12;
        const int base = B3W_A_HG + 8 * ((r * 8 + g) * 2 + hf);
        const F256 va = at_get(v, a), vb = at_get(v, b), vc = at_get(v, c), vd = at_get(v, d);
        const F256 s1 = f_add(f_add(va, vb, P), at_get(msg, perm[2 * g + hf]), P);
        if (!f_fits(s1, 34)) return SITE(1, r, g, hf);                     // Bits34 :201
        const uint32_t A = s1.l[0];
        if (!f_fits(vd, 32)) return SITE(2, r, g, hf);                     // ToBits (rxor2.tb) :153
        const uint32_t DI = vd.l[0], D2 = rotr32(DI ^ A, R1);
        const F256 s3 = f_add(vc, f_small(D2), P);
        if (!f_fits(s3, 33)) return SITE(3, r, g, hf);                     // Bits33 :176
        const uint32_t C = s3.l[0];
        if (!f_fits(vb, 32)) return SITE(4, r, g, hf);                     // ToBits (rxor4.tb) :153
        const uint32_t BI = vb.l[0], B4 = rotr32(BI ^ C, R2);
        at_put(at, base + B3W_HG_S1, s1); at_put(at, base + B3W_HG_A, f_small(A));
        at_put(at, base + B3W_HG_S3, s3); at_put(at, base + B3W_HG_C, f_small(C));
        at_put(at, base + B3W_HG_D2, f_small(D2)); at_put(at, base + B3W_HG_DI, f_small(DI));
        at_put(at, base + B3W_HG_B4, f_small(B4)); at_put(at, base + B3W_HG_BI, f_small(BI));
        at_put(v, a, f_small(A)); at_put(v, b, f_small(B4)); at_put(v, c, f_small(C)); at_put(v, d, f_small(D2));
      }
    }
    uint8_t np[16];
    for (int j = 0; j < 16; ++j) np[j] = perm[sigma[j]];
    for (int j = 0; j < 16; ++j) perm[j] = np[j];
  }
  for (int k = 0; k < 16; ++k) {                                           // outXor :213-227
    const F256 x = at_get(v, k), y = k < 8 ? at_get(v, k + 8) : at_get(at, B3W_A_H + k - 8);
    if (!f_fits(x, 32)) return SITE(5, 7, k & 7, k >> 3);
    if (!f_fits(y, 32)) return SITE(6, 7, k & 7, k >> 3);
    at_put(at, B3W_A_O + k, f_small(x.l[0] ^ y.l[0]));
  }
  return 0;
}

template <bool NOVA>
__global__ __launch_bounds__(64) void b3w_exact_kernel(const uint32_t *__restrict__ inputs /* nin x 8 words */,
                                                       const uint32_t *__restrict__ prime,
                                                       const uint32_t *__restrict__ table /* atom | bit<<16 (0xFFFF whole) */,
                                                       uint32_t nwit, uint8_t *__restrict__ out,
                                                       uint32_t *__restrict__ status /* [0] status, [1] site */) {
  __shared__ uint32_t at[B3W_N_NOVA_ATOMS * 8];
  __shared__ uint32_t vbuf[32 * 8];
  __shared__ uint32_t fail;
  const int lane = threadIdx.x;
  F256 P;
#pragma unroll
  for (int i = 0; i < 8; ++i) P.l[i] = prime[i];
  for (int i = lane; i < B3W_N_NOVA_ATOMS * 8; i += 64) at[i] = 0;
  if (lane == 0) fail = 0;
  __syncthreads();
  const int nin = NOVA ? 32 : 28, first = NOVA ? B3W_A_NV : B3W_A_H;
  for (int i = lane; i < nin * 8; i += 64) at[first * 8 + i] = inputs[i];
  __syncthreads();

  if (NOVA) {
    uint32_t *nv = at + B3W_A_NV * 8;
    const F256 zero = f_small(0), one = f_small(1);
    // the 67 IsZero gadgets, one per lane-job
    for (int j = lane; j < 67; j += 64) {
      const F256 depth = at_get(nv, NV_DEPTH);
      F256 in, in1 = zero;
      int flag_atom, inv_atom, isz_atom, in1_atom = -1;
      if (j == 0) { in = f_sub(zero, depth, P); flag_atom = NV_IS_ROOT; inv_atom = NV_ROOT_INV; isz_atom = NV_ROOT_ISZ_IN; }
      else if (j == 1) { in = f_sub(zero, at_get(nv, NV_BLOCK_COUNT), P); flag_atom = NV_E0; inv_atom = NV_E0_INV; isz_atom = NV_E0_ISZ_IN; }
      else if (j == 2) { in1 = f_sub(at_get(nv, NV_N_BLOCKS), one, P); in = f_sub(in1, at_get(nv, NV_BLOCK_COUNT), P);
                         flag_atom = NV_E1; inv_atom = NV_E1_INV; isz_atom = NV_E1_ISZ_IN; in1_atom = NV_E1_IN1; }
      else { const int i = j - 3; in1 = f_sub(at_get(nv, NV_TOTAL_DEPTH), f_small((uint64_t)i + 2), P); in = f_sub(in1, depth, P);
             flag_atom = NV_EQ_OUT + i; inv_atom = NV_EQ_INV + i; isz_atom = NV_EQ_ISZ_IN + i; in1_atom = NV_EQ_IN1 + i; }
      at_put(nv, inv_atom, f_inv(in, P));
      at_put(nv, flag_atom, f_small(f_is_zero(in) ? 1 : 0));
      at_put(nv, isz_atom, in);
      if (in1_atom >= 0) at_put(nv, in1_atom, in1);
    }
    __syncthreads();
    if (lane == 0) {
      uint32_t site = 0;
      const F256 depth = at_get(nv, NV_DEPTH), leaf_depth = at_get(nv, NV_LEAF_DEPTH);
      const F256 cil = at_get(nv, NV_CIL), cih = at_get(nv, NV_CIH);
      // Blake3NovaTreePath_CheckDepth (:13-45)
      const F256 cp_in1 = f_sub(leaf_depth, one, P);
      const F256 cp = f_sub(f_add(depth, f_small(256), P), cp_in1, P);
      const F256 ed_in1 = f_add(depth, one, P);
      const F256 ed = f_sub(f_add(leaf_depth, f_small(256), P), ed_in1, P);
      at_put(nv, NV_CP_IN1, cp_in1); at_put(nv, NV_CP_N2B_IN, cp);
      at_put(nv, NV_ED_IN1, ed_in1); at_put(nv, NV_ED_N2B_IN, ed);
      uint32_t parent = 0;
      if (!f_fits(cp, 9)) site = SITE(10, 0, 0, 0);                       // check_parent Num2Bits(9)
      else {
        parent = 1u - f_bit(cp, 8);
        if (!f_fits(ed, 9)) site = SITE(11, 0, 0, 0);                     // exceed_depth Num2Bits(9)
        else if (f_bit(ed, 8) == 0) site = SITE(12, 0, 0, 0);             // exceed_depth.out === 0 (line 38)
      }
      if (!site) {
        const uint32_t is_root = at_get(nv, NV_IS_ROOT).l[0], e0 = at_get(nv, NV_E0).l[0], e1 = at_get(nv, NV_E1).l[0];
        at_put(nv, NV_IS_PARENT, f_small(parent));
        at_put(nv, NV_ED_OUT, zero);
        at_put(nv, NV_NOT_ROOT, f_small(1 - is_root));
        at_put(nv, NV_NOT_PARENT, f_small(1 - parent));
        const uint32_t last = e1 & (1 - parent), first_ = e0 & (1 - parent), ur_tmp = parent | e1, ur_flag = ur_tmp & is_root;
        at_put(nv, NV_IS_LAST_BLOCK, f_small(last)); at_put(nv, NV_FIRST, f_small(first_));
        at_put(nv, NV_UR_TMP, f_small(ur_tmp)); at_put(nv, NV_UR_FLAG, f_small(ur_flag));
        const uint32_t dflag = first_ + 2 * last + 8 * ur_flag + 4 * parent;
        // chunk_idx = low + high * 2^32 (32 modular doublings)
        F256 hi = cih;
        for (int i = 0; i < 32; ++i) hi = f_add(hi, hi, P);
        const F256 chunk_idx = f_add(cil, hi, P);
        at_put(nv, NV_CHUNK_IDX, chunk_idx);
        if (!f_fits(chunk_idx, 65)) site = SITE(13, 0, 0, 0);             // down_left_path Num2Bits(65)
        else {
          F256 bad = zero;
          for (int i = 0; i < 64; ++i) {
            if (at_get(nv, NV_EQ_OUT + i).l[0] && f_bit(chunk_idx, i) == 0) bad = f_add(bad, one, P);
            at_put(nv, NV_BIT_AT_DEPTH + i, bad);
          }
          const F256 dl = parent ? bad : one;                               // (1-parent) + parent*bit_at_depth[63]
          at_put(nv, NV_DL, dl);
          if (!(f_is_zero(dl) || f_eq(dl, one))) site = SITE(14, 0, 0, 0); // out*(1-out) === 0
          else {
            const uint32_t dlb = dl.l[0];
            const uint32_t IV8[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
            for (int i = 0; i < 16; ++i) {                                  // Blake3GetFinal_m (:86-120)
              const F256 hsel = at_get(nv, NV_H + (i & 7)), msel = at_get(nv, NV_M + (i & 7)), mi = at_get(nv, NV_M + i);
              const uint32_t take_h = i < 8 ? dlb : 1 - dlb;
              const F256 td = take_h ? hsel : zero;
              const F256 mp = f_add(take_h ? zero : msel, td, P);
              const F256 tp = parent ? mp : zero;
              at_put(nv, NV_TMP_DOWN + i, td); at_put(nv, NV_M_IS_PARENT + i, mp); at_put(nv, NV_TMP_IS_PAR + i, tp);
              at_put(at, B3W_A_M + i, f_add(parent ? zero : mi, tp, P));
            }
            for (int i = 0; i < 8; ++i) {                                   // h_compression (:229-233)
              const F256 tiv = parent ? f_small(IV8[i]) : zero;
              at_put(nv, NV_TMPIV + i, tiv);
              at_put(at, B3W_A_H + i, f_add(parent ? zero : at_get(nv, NV_H + i), tiv, P));
            }
            at_put(at, B3W_A_T, parent ? zero : cil);
            at_put(at, B3W_A_T + 1, parent ? zero : cih);
            at_put(at, B3W_A_B, at_get(nv, NV_B));
            at_put(at, B3W_A_D, f_small(dflag));
            site = exact_compression(at, vbuf, P);
            if (!site) {
              at_put(nv, NV_BLOCK_COUNT_OUT, f_add(at_get(nv, NV_BLOCK_COUNT), f_small(1 - parent), P));
              const uint32_t cdd = last | parent, decr = cdd & (1 - is_root);
              at_put(nv, NV_CDD_OUT, f_small(cdd)); at_put(nv, NV_DECR_DEPTH, f_small(decr));
              at_put(nv, NV_DEPTH_OUT, f_sub(depth, f_small(decr), P));
            }
          }
        }
      }
      fail = site;
    }
  } else {
    if (lane == 0) fail = exact_compression(at, vbuf, P);
  }
  __syncthreads();
  const uint32_t site = fail;
  if (lane == 0) { status[0] = site ? 4u : 0u; status[1] = site; }
  if (site) return;
  for (uint32_t s = lane; s < nwit; s += 64) {
    const uint32_t e = table[s], atom = e & 0xFFFFu, bit = e >> 16;
    uint4 lo, hi;
    if (bit == 0xFFFFu) {
      const uint32_t *a = at + atom * 8;
      lo = make_uint4(a[0], a[1], a[2], a[3]);
      hi = make_uint4(a[4], a[5], a[6], a[7]);
    } else {
      lo = make_uint4((at[atom * 8 + (bit >> 5)] >> (bit & 31)) & 1u, 0, 0, 0);
      hi = make_uint4(0, 0, 0, 0);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(out + (uint64_t)s * 32);
    dst[0] = lo;
    dst[1] = hi;
  }
}

}  // namespace

extern "C" int b3w_launch_exact(int nova, const uint32_t *d_inputs, const uint32_t *d_prime, const uint32_t *d_table,
                                uint32_t nwit, uint8_t *d_out, uint32_t *d_status, hipStream_t stream) {
  if (nova) hipLaunchKernelGGL(b3w_exact_kernel<true>, dim3(1), dim3(64), 0, stream, d_inputs, d_prime, d_table, nwit, d_out, d_status);
  else hipLaunchKernelGGL(b3w_exact_kernel<false>, dim3(1), dim3(64), 0, stream, d_inputs, d_prime, d_table, nwit, d_out, d_status);
  return (int)hipGetLastError();
}
