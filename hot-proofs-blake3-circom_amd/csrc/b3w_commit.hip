// b3w_commit.hip — on-device consumer #2: Pedersen commitments of witness bodies where they lie in HBM.
//
// What arecibo does with a step witness right after `synthesize` (rust_fold/src/main.rs:166-179
// -> RecursiveSNARK::prove_step -> commit to W): C = sum_i w_i * G_i over the group whose scalar field is the
// circuit's field — BN254 G1 for the bn128 builds, the Vesta curve for the --prime vesta build (y^2 = x^3 + b, a = 0).
// The generators are the caller's (arecibo derives its commitment key from a label; pass that key), so the result
// is comparable with the prover's own commitment; tests check it against an independent big-integer implementation.
//
// The witness is almost all bits: of 24 093 compression slots ~23 500 hold 0 or 1, the rest 32/34-bit words (nova:
// plus 67 256-bit inverses).  So the multi-scalar multiplication is turned into "add precomputed points":
//   set-up   every slot is cut into its bits ("virtual slots": slot, bit k; V = 53 k compression, 58 k nova O2) with
//            the point 2^k * G_slot each (b3w_commit_setup_kernel); W = 12 consecutive virtual slots form a WINDOW whose
//            4 095 non-empty subset sums are tabulated (b3w_commit_window_kernel: 4.5 k windows x 4 095 affine points = 1.2 GB; or W = 16:
//            3.3 k windows x 65 535 points = 14 GB);
//   commit   32 or 64 lanes per witness: they stream the body once and pack its virtual-slot bits into LDS (records mode:
//            the same bits from the witness's 3.7-11 KB TRACE image — no body at all); lane t then owns
//            windows t, t + T, ...: it skips ahead to its next NON-ZERO window and the wave does one mixed (XYZZ +
//            affine) addition with the tabulated point — no doublings, one addition per W slots, no zero work in lock
//            step; an LDS tree shared by the workgroup's witnesses adds the partial sums; a second kernel normalises them, one thread per witness
//            (Fermat inversion), and stores the affine points.
// Arithmetic: 256-bit Montgomery, modulus passed at run time.  The per-slot doubling chains of the set-up use the textbook
// CIOS on eight 32-bit limbs; the table, commit and normalise kernels use nine 29-bit limbs (radix 2^261, lazy reduction: see the
// F9 section below), in which a limb product is a single v_mad_u64_u32.  Complete handling of the exceptional cases
// (infinity, P + P, P - P) so that related generators cannot break it.
// Domain: bodies of the batch kernels (every bit slot holds 0 or 1, words fit their slot's width); a slot holding
// anything else (e.g. a body of the exact kernel with a 254-bit input) flags the witness (status 103) instead of
// producing a wrong commitment.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "b3w_kernels.h"
#include "b3w_atoms.h"

namespace {

struct Fp { uint32_t l[8]; };
struct Jac { Fp X, Y, Z; };          // Z == 0: the point at infinity

__device__ __forceinline__ bool fp_is_zero(const Fp &a) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.l[i];
  return o == 0;
}
__device__ __forceinline__ bool fp_eq(const Fp &a, const Fp &b) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) o |= a.l[i] ^ b.l[i];
  return o == 0;
}
__device__ __forceinline__ Fp fp_zero() { Fp r; for (int i = 0; i < 8; ++i) r.l[i] = 0; return r; }

// r = a - p if a >= p (a < 2p, `hi` = the ninth limb of a)
__device__ __forceinline__ Fp fp_reduce_once(const Fp &a, uint32_t hi, const B3wCurve &C) {
  Fp d;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t t = (uint64_t)a.l[i] - C.p[i] - br;
    d.l[i] = (uint32_t)t;
    br = (uint32_t)(t >> 63);
  }
  const bool ge = hi != 0 || br == 0;      // a >= p
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = ge ? d.l[i] : a.l[i];
  return r;
}
__device__ __forceinline__ Fp fp_add(const Fp &a, const Fp &b, const B3wCurve &C) {
  Fp s;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t t = (uint64_t)a.l[i] + b.l[i] + c;
    s.l[i] = (uint32_t)t;
    c = (uint32_t)(t >> 32);
  }
  return fp_reduce_once(s, c, C);
}
__device__ __forceinline__ Fp fp_sub(const Fp &a, const Fp &b, const B3wCurve &C) {
  Fp d;
  uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t t = (uint64_t)a.l[i] - b.l[i] - br;
    d.l[i] = (uint32_t)t;
    br = (uint32_t)(t >> 63);
  }
  uint32_t c = 0;                          // + p when it went negative
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint64_t t = (uint64_t)d.l[i] + (br ? C.p[i] : 0u) + c;
    d.l[i] = (uint32_t)t;
    c = (uint32_t)(t >> 32);
  }
  return d;
}
__device__ __forceinline__ Fp fp_dbl(const Fp &a, const B3wCurve &C) { return fp_add(a, a, C); }

// Montgomery product a * b / 2^256 mod p (CIOS)
__device__ __forceinline__ Fp fp_mul(const Fp &a, const Fp &b, const B3wCurve &C) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c += (uint64_t)a.l[j] * b.l[i] + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (uint32_t)c;
    t[9] = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * C.inv;
    c = (uint64_t)m * C.p[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      c += (uint64_t)m * C.p[j] + t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (uint32_t)c;
    t[8] = t[9] + (uint32_t)(c >> 32);
  }
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = t[i];
  return fp_reduce_once(r, t[8], C);
}
// Montgomery square: the 28 cross products once, doubled, plus the 8 squares (36 limb products, not 64), then the
// reduction of the 512-bit product.  1.17x the rate of fp_mul(a, a) on gfx950 (tools/ubench/fpmul_peak.hip).
__device__ __forceinline__ Fp fp_sqr(const Fp &a, const B3wCurve &C) {
  uint32_t t[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = i + 1; j < 8; ++j) {
      c += (uint64_t)a.l[i] * a.l[j] + t[i + j];
      t[i + j] = (uint32_t)c;
      c >>= 32;
    }
    t[i + 8] = (uint32_t)c;
  }
  uint32_t top = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const uint32_t n = (t[i] << 1) | top;
    top = t[i] >> 31;
    t[i] = n;
  }
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.l[i] * a.l[i] + t[2 * i];
    t[2 * i] = (uint32_t)c;
    c >>= 32;
    c += t[2 * i + 1];
    t[2 * i + 1] = (uint32_t)c;
    c >>= 32;
  }
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t m = t[i] * C.inv;
    uint64_t d = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      d += (uint64_t)m * C.p[j] + t[i + j];
      t[i + j] = (uint32_t)d;
      d >>= 32;
    }
    d += (uint64_t)t[i + 8] + carry;
    t[i + 8] = (uint32_t)d;
    carry = (uint32_t)(d >> 32);
  }
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = t[i + 8];
  return fp_reduce_once(r, carry, C);
}

// a^(p-2): Fermat inversion (a != 0)
__device__ Fp fp_inv(const Fp &a, const B3wCurve &C) {
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = C.one[i];
  for (int i = 255; i >= 0; --i) {
    r = fp_sqr(r, C);
    if ((C.pm2[i >> 5] >> (i & 31)) & 1) r = fp_mul(r, a, C);
  }
  return r;
}

__device__ __forceinline__ Jac jac_infinity() { Jac r; r.X = fp_zero(); r.Y = fp_zero(); r.Z = fp_zero(); return r; }

// 2P, a = 0 (dbl-2009-l)
__device__ __forceinline__ Jac jac_dbl(const Jac &P, const B3wCurve &C) {
  if (fp_is_zero(P.Z) || fp_is_zero(P.Y)) return jac_infinity();
  const Fp A = fp_sqr(P.X, C), B = fp_sqr(P.Y, C), Cc = fp_sqr(B, C);
  Fp D = fp_sub(fp_sub(fp_sqr(fp_add(P.X, B, C), C), A, C), Cc, C);
  D = fp_dbl(D, C);
  const Fp E = fp_add(fp_dbl(A, C), A, C), F = fp_sqr(E, C);
  Jac R;
  R.X = fp_sub(F, fp_dbl(D, C), C);
  Fp c8 = fp_dbl(fp_dbl(fp_dbl(Cc, C), C), C);
  R.Y = fp_sub(fp_mul(E, fp_sub(D, R.X, C), C), c8, C);
  R.Z = fp_dbl(fp_mul(P.Y, P.Z, C), C);
  return R;
}

// P + (x2, y2, 1) (madd-2007-bl) with the exceptional cases
__device__ __forceinline__ Jac jac_madd(const Jac &P, const Fp &x2, const Fp &y2, const B3wCurve &C) {
  if (fp_is_zero(x2) && fp_is_zero(y2)) return P;        // (0, 0) is not on these curves (b != 0): the table's infinity
  if (fp_is_zero(P.Z)) {
    Jac R; R.X = x2; R.Y = y2;
#pragma unroll
    for (int i = 0; i < 8; ++i) R.Z.l[i] = C.one[i];
    return R;
  }
  const Fp Z1Z1 = fp_sqr(P.Z, C), U2 = fp_mul(x2, Z1Z1, C), S2 = fp_mul(fp_mul(y2, P.Z, C), Z1Z1, C);
  const Fp H = fp_sub(U2, P.X, C), rr = fp_sub(S2, P.Y, C);
  if (fp_is_zero(H)) {
    if (!fp_is_zero(rr)) return jac_infinity();        // P + (-P)
    Jac Q; Q.X = x2; Q.Y = y2;
#pragma unroll
    for (int i = 0; i < 8; ++i) Q.Z.l[i] = C.one[i];
    return jac_dbl(Q, C);                               // P + P
  }
  const Fp HH = fp_sqr(H, C), I = fp_dbl(fp_dbl(HH, C), C), J = fp_mul(H, I, C), r = fp_dbl(rr, C), V = fp_mul(P.X, I, C);
  Jac R;
  R.X = fp_sub(fp_sub(fp_sqr(r, C), J, C), fp_dbl(V, C), C);
  R.Y = fp_sub(fp_mul(r, fp_sub(V, R.X, C), C), fp_dbl(fp_mul(P.Y, J, C), C), C);
  R.Z = fp_sub(fp_sub(fp_sqr(fp_add(P.Z, H, C), C), Z1Z1, C), HH, C);
  return R;
}

// P + Q (add-2007-bl) with the exceptional cases
__device__ __forceinline__ Jac jac_add(const Jac &P, const Jac &Q, const B3wCurve &C) {
  if (fp_is_zero(P.Z)) return Q;
  if (fp_is_zero(Q.Z)) return P;
  const Fp Z1Z1 = fp_sqr(P.Z, C), Z2Z2 = fp_sqr(Q.Z, C);
  const Fp U1 = fp_mul(P.X, Z2Z2, C), U2 = fp_mul(Q.X, Z1Z1, C);
  const Fp S1 = fp_mul(fp_mul(P.Y, Q.Z, C), Z2Z2, C), S2 = fp_mul(fp_mul(Q.Y, P.Z, C), Z1Z1, C);
  const Fp H = fp_sub(U2, U1, C), rr = fp_sub(S2, S1, C);
  if (fp_is_zero(H)) return fp_is_zero(rr) ? jac_dbl(P, C) : jac_infinity();
  const Fp I = fp_sqr(fp_dbl(H, C), C), J = fp_mul(H, I, C), r = fp_dbl(rr, C), V = fp_mul(U1, I, C);
  Jac R;
  R.X = fp_sub(fp_sub(fp_sqr(r, C), J, C), fp_dbl(V, C), C);
  R.Y = fp_sub(fp_mul(r, fp_sub(V, R.X, C), C), fp_dbl(fp_mul(S1, J, C), C), C);
  R.Z = fp_mul(fp_sub(fp_sub(fp_sqr(fp_add(P.Z, Q.Z, C), C), Z1Z1, C), Z2Z2, C), H, C);
  return R;
}

__device__ __forceinline__ Fp load_fp(const uint32_t *p) {
  const uint4 a = reinterpret_cast<const uint4 *>(p)[0], b = reinterpret_cast<const uint4 *>(p)[1];
  Fp r;
  r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
  return r;
}
__device__ __forceinline__ void store_fp(uint32_t *p, const Fp &v) {
  reinterpret_cast<uint4 *>(p)[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
  reinterpret_cast<uint4 *>(p)[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// Jacobian (Montgomery) -> affine (Montgomery); infinity -> (0, 0)
__device__ void jac_to_affine(const Jac &P, Fp &x, Fp &y, const B3wCurve &C) {
  if (fp_is_zero(P.Z)) { x = fp_zero(); y = fp_zero(); return; }
  const Fp zi = fp_inv(P.Z, C), zi2 = fp_sqr(zi, C);
  x = fp_mul(P.X, zi2, C);
  y = fp_mul(P.Y, fp_mul(zi2, zi, C), C);
}

// =====================================================================================================================
// The commit kernel's own arithmetic: the same field in NINE 29-BIT LIMBS, Montgomery radix 2^261.
// Every column of a 9 x 9 limb product (plus the reduction's) fits one 64-bit accumulator, so a limb product is ONE
// v_mad_u64_u32 whose addend is the accumulator itself: no carry moves (the 8 x 32-bit CIOS above spends three quarters
// of its instructions on them).  172 G multiplications/s against 93 (tools/ubench/fpmul29_peak.hip).
// Values are LAZY: a value is any representative below 2^261 (>= 64p); limbs may exceed 29 bits between operations.
// Each routine states what it needs and what it gives:  "tidy" = limbs 0..7 below 2^29;  "< kp" = the value's bound.
//   mul29 / sqr29   limb bounds la * lb < 1.52 * 2^60; a < Ap, b < Bp  ->  tidy, < (AB/64 + 1)p
//   cn29            any limbs  ->  tidy, same value
//   red29           any limbs, value < 2^261  ->  tidy, < 2p   (one-word Barrett quotient, never too large, at most 1 short)
//   a + sp4 - b     b tidy < 2p  ->  value a + 4p - b, limbs < a's + 2^30   (sp4 = 4p with every limb lifted by 2^29)
#define M29 0x1FFFFFFFu
struct F9 { uint32_t l[9]; };
struct J9 { F9 X, Y, ZZ, ZZZ; bool inf; };  // the point (X / ZZ, Y / ZZZ), ZZ^3 = ZZZ^2 ("XYZZ" coordinates: a mixed addition is
                                             // 8M + 2S, one squaring less than Jacobian); between additions all four are tidy < 2p
struct B3wCurve9 {
  uint32_t p[9];        // modulus
  uint32_t sp4[9];      // 4p, limb i lifted by 2^29 and lowered by the 1 it lent to limb i - 1
  uint32_t one[9];      // 2^261 mod p
  uint32_t inv;         // -p^-1 mod 2^29
  uint32_t mu;          // floor(2^269 / p) (or 1 less)
  uint32_t kp0[3];      // limb 0 of 3p, 4p, 5p: the filter in front of the exact "H = 0 mod p" test
  __device__ __forceinline__ uint32_t P(int i) const { return p[i]; }
  __device__ __forceinline__ uint32_t INV() const { return inv; }
};
// The Vesta base field with its modulus as compile-time constants: p = 2^254 + (126 bits) has limb 0 = 1, limbs 5..7 = 0
// and limb 8 = 2^22, so a third of the reduction's multiplications fold away (same layout, chosen by the launcher
// when the key's modulus is this one).
struct B3wCurve9Vesta : B3wCurve9 {
  __host__ __device__ static constexpr uint32_t P(int i) {
    return i == 0 ? 0x1u : i == 1 ? 0x9698768u : i == 2 ? 0x133e46e6u : i == 3 ? 0xd31f812u : i == 4 ? 0x224u : i == 8 ? 0x400000u : 0u;
  }
  __host__ __device__ static constexpr uint32_t INV() { return M29; }
};

template <class CV>
__device__ __forceinline__ F9 mul29(const F9 &a, const F9 &b, const CV &C) {
  uint64_t acc = 0;
  uint32_t m[9];
  F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * C.P(k - i);
    m[k] = ((uint32_t)acc * C.INV()) & M29;
    acc += (uint64_t)m[k] * C.P(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * C.P(k - i);
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
// (a b + c d) / 2^261: two products, ONE reduction.  Column bound: 9 (la lb + lc ld) + 9 * 2^58 < 2^64, i.e.
// la lb + lc ld < 1.52 * 2^60 (e.g. 2^29 x 1.5 * 2^30 and 2^30 x 2^29); a < Ap ... d < Dp  ->  tidy, < ((AB + CD)/64 + 1)p
template <class CV>
__device__ __forceinline__ F9 muladd29(const F9 &a, const F9 &b, const F9 &c, const F9 &d, const CV &C) {
  uint64_t acc = 0;
  uint32_t m[9];
  F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc += (uint64_t)a.l[i] * b.l[k - i]; acc += (uint64_t)c.l[i] * d.l[k - i]; }
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * C.P(k - i);
    m[k] = ((uint32_t)acc * C.INV()) & M29;
    acc += (uint64_t)m[k] * C.P(0);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) { acc += (uint64_t)a.l[i] * b.l[k - i]; acc += (uint64_t)c.l[i] * d.l[k - i]; }
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * C.P(k - i);
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
template <class CV>
__device__ __forceinline__ F9 sqr29(const F9 &a, const CV &C) {      // limbs < 2^30
  uint64_t acc = 0;
  uint32_t m[9], d[9];
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
#pragma unroll
    for (int i = (k > 8 ? k - 8 : 0); 2 * i < k; ++i) acc += (uint64_t)d[i] * a.l[k - i];
    if (!(k & 1)) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
    if (k < 9) {
#pragma unroll
      for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * C.P(k - i);
      m[k] = ((uint32_t)acc * C.INV()) & M29;
      acc += (uint64_t)m[k] * C.P(0);
    } else {
#pragma unroll
      for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * C.P(k - i);
      r.l[k - 9] = (uint32_t)acc & M29;
    }
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
__device__ __forceinline__ F9 cn29(F9 a) {
#pragma unroll
  for (int i = 0; i < 8; ++i) { a.l[i + 1] += a.l[i] >> 29; a.l[i] &= M29; }
  return a;
}
template <class CV>
__device__ __forceinline__ F9 red29(const F9 &a, const CV &C) {
  const uint32_t top = (a.l[8] + (a.l[7] >> 29)) >> 13;            // value / 2^245, rounded down (< 2^16)
  const uint32_t q = (top * C.mu) >> 24;                           // floor(value / p) or 1 less
  int64_t acc = 0;
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    acc += (int64_t)(uint64_t)a.l[i];
    acc -= (int64_t)((uint64_t)q * C.P(i));
    r.l[i] = i < 8 ? (uint32_t)acc & M29 : (uint32_t)acc;
    acc >>= 29;
  }
  return r;
}
__device__ __forceinline__ F9 add29(const F9 &a, const F9 &b) {
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
  return r;
}
template <class CV>
__device__ __forceinline__ F9 sub29(const F9 &a, const F9 &b, const CV &C) {     // a + 4p - b   (b tidy < 2p)
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + (C.sp4[i] - b.l[i]);
  return r;
}
template <class CV>
__device__ __forceinline__ F9 neg29(const F9 &b, const CV &C) {                  // 4p - b
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = C.sp4[i] - b.l[i];
  return r;
}
__device__ __forceinline__ F9 shl29(const F9 &a, int s) {
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] << s;
  return r;
}
// tidy a == k * p ?
template <class CV>
__device__ __forceinline__ bool is_kp29(const F9 &a, uint32_t k, const CV &C) {
  uint32_t c = 0, o = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint32_t t = k * C.P(i) + c;
    o |= (i < 8 ? t & M29 : t) ^ a.l[i];
    c = t >> 29;
  }
  return o == 0;
}
// the safe forms (tidy < 2p in, tidy < 2p out) for the rare paths and the tree
template <class CV>
__device__ __forceinline__ F9 s_add(const F9 &a, const F9 &b, const CV &C) { return red29(add29(a, b), C); }
template <class CV>
__device__ __forceinline__ F9 s_sub(const F9 &a, const F9 &b, const CV &C) { return red29(sub29(a, b, C), C); }
template <class CV>
__device__ __forceinline__ F9 s_dbl(const F9 &a, const CV &C) { return red29(shl29(a, 1), C); }
template <class CV>
__device__ __forceinline__ bool s_is_zero(const F9 &a, const CV &C) { return is_kp29(a, 0, C) || is_kp29(a, 1, C); }
template <class CV>
__device__ __forceinline__ F9 one29(const CV &C) {
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.l[i] = C.one[i];
  return r;
}

// a^(p-2) (Fermat), a tidy < 2p and != 0 mod p; pm2 = p - 2 in eight 32-bit words
template <class CV>
__device__ __forceinline__ F9 inv29(const F9 &a, const uint32_t *pm2, const CV &C) {
  F9 r = one29(C);
#pragma unroll 1
  for (int i = 255; i >= 0; --i) {
    r = sqr29(r, C);
    if ((pm2[i >> 5] >> (i & 31)) & 1) r = mul29(r, a, C);
  }
  return r;
}

__device__ __forceinline__ F9 to29(const Fp &a) {                   // eight 32-bit words -> nine 29-bit limbs
  F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int bit = 29 * k, w = bit >> 5, sh = bit & 31;
    uint64_t v = a.l[w];
    if (w + 1 < 8) v |= (uint64_t)a.l[w + 1] << 32;
    r.l[k] = (uint32_t)(v >> sh) & M29;
  }
  return r;
}
__device__ __forceinline__ Fp from29(const F9 &a, uint32_t &hi) {     // tidy -> eight words + the bits above 2^256
  uint32_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int bit = 29 * k, w = bit >> 5, sh = bit & 31;
    const uint64_t v = (uint64_t)a.l[k] << sh;
    t[w] |= (uint32_t)v;
    if (w + 1 < 9) t[w + 1] |= (uint32_t)(v >> 32);
  }
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = t[i];
  hi = t[8];
  return r;
}

__device__ __forceinline__ J9 j9_infinity() {
  J9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.X.l[i] = r.Y.l[i] = r.ZZ.l[i] = r.ZZZ.l[i] = 0;
  r.inf = true;
  return r;
}
// P = 2P (dbl-2008-s-1, a = 0), safe forms
template <class CV>
__device__ __forceinline__ void j9_dbl(J9 &P, const CV &C) {
  if (P.inf) return;
  if (s_is_zero(P.Y, C)) { P.inf = true; return; }
  const F9 U = s_dbl(P.Y, C), V = sqr29(U, C), W = mul29(U, V, C), S = mul29(P.X, V, C);
  const F9 XX = sqr29(P.X, C), M = s_add(s_dbl(XX, C), XX, C);
  const F9 X3 = s_sub(sqr29(M, C), s_dbl(S, C), C);
  const F9 Y3 = s_sub(mul29(M, s_sub(S, X3, C), C), mul29(W, P.Y, C), C);
  P.ZZ = mul29(V, P.ZZ, C);
  P.ZZZ = mul29(W, P.ZZZ, C);
  P.X = X3; P.Y = Y3;
}
// P += (x2, y2): madd-2008-s with lazy values; x2, y2 tidy < p; (0, 0) (the table's infinity) is filtered by the caller.
// Bounds: inputs X1, Y1, ZZ1, ZZZ1 tidy < 2p  ->  outputs the same.
template <class CV>
__device__ __forceinline__ void j9_madd(J9 &P, const F9 &x2, const F9 &y2, const CV &C) {
  if (P.inf) {
    P.X = x2; P.Y = y2; P.ZZ = one29(C); P.ZZZ = one29(C); P.inf = false;
    return;
  }
  const F9 U2 = mul29(x2, P.ZZ, C), S2 = mul29(y2, P.ZZZ, C);      // < 2p
  const F9 Pd = cn29(sub29(U2, P.X, C));                           // tidy, in (2p, 6p): = 0 mod p iff 3p, 4p or 5p
  const F9 R = cn29(sub29(S2, P.Y, C));                            // likewise
  if (Pd.l[0] == C.kp0[0] || Pd.l[0] == C.kp0[1] || Pd.l[0] == C.kp0[2]) {
    if (is_kp29(Pd, 3, C) || is_kp29(Pd, 4, C) || is_kp29(Pd, 5, C)) {
      if (!(is_kp29(R, 3, C) || is_kp29(R, 4, C) || is_kp29(R, 5, C))) { P.inf = true; return; }     // P + (-P)
      P.X = x2; P.Y = y2; P.ZZ = one29(C); P.ZZZ = one29(C);
      j9_dbl(P, C);                                                                                  // P + P
      return;
    }
  }
  const F9 PP = sqr29(Pd, C);                                      // < 2p   (36 / 64 + 1)
  const F9 PPP = mul29(Pd, PP, C), Q = mul29(P.X, PP, C);          // < 2p
  const F9 X3 = red29(add29(add29(sqr29(R, C), neg29(PPP, C)), shl29(neg29(Q, C), 1)), C);    // R^2 - PPP - 2Q  (< 14p before)
  P.Y = muladd29(R, sub29(Q, X3, C), neg29(P.Y, C), PPP, C);       // R (Q - X3) + (4p - Y1) PPP: (6 * 5.1 + 4 * 1.2)/64 + 1 < 2p
  P.X = X3;
  P.ZZ = mul29(P.ZZ, PP, C);
  P.ZZZ = mul29(P.ZZZ, PPP, C);
}
// P += Q (add-2008-s), safe forms
template <class CV>
__device__ __forceinline__ void j9_add(J9 &P, const J9 &Q, const CV &C) {
  if (Q.inf) return;
  if (P.inf) { P.X = Q.X; P.Y = Q.Y; P.ZZ = Q.ZZ; P.ZZZ = Q.ZZZ; P.inf = false; return; }
  const F9 U1 = mul29(P.X, Q.ZZ, C), U2 = mul29(Q.X, P.ZZ, C);
  const F9 S1 = mul29(P.Y, Q.ZZZ, C), S2 = mul29(Q.Y, P.ZZZ, C);
  const F9 Pd = s_sub(U2, U1, C), R = s_sub(S2, S1, C);
  if (s_is_zero(Pd, C)) {
    if (s_is_zero(R, C)) j9_dbl(P, C); else P.inf = true;
    return;
  }
  const F9 PP = sqr29(Pd, C), PPP = mul29(Pd, PP, C), Qq = mul29(U1, PP, C);
  const F9 X3 = s_sub(s_sub(sqr29(R, C), PPP, C), s_dbl(Qq, C), C);
  P.Y = s_sub(mul29(R, s_sub(Qq, X3, C), C), mul29(S1, PPP, C), C);
  P.ZZ = mul29(mul29(P.ZZ, Q.ZZ, C), PP, C);
  P.ZZZ = mul29(mul29(P.ZZZ, Q.ZZZ, C), PPP, C);
  P.X = X3;
}

// ---- set-up: points[first_v[s] + k] = 2^k * G_s (affine, Montgomery radix 2^261, eight 32-bit words) for every committed slot s.
// One workgroup per slot, thread t takes bits t, t + 64, ...: t doublings to get there (a single thread walking all 256
// bits of a field-element slot, with an inversion per bit, was the whole key set-up time of the 12-bit tables).
__global__ __launch_bounds__(64) void b3w_commit_setup_kernel(const uint32_t *__restrict__ gens /* nslots x 16 words, standard form */,
                                                              const uint32_t *__restrict__ first_v, const uint32_t *__restrict__ nbits,
                                                              uint32_t nslots, uint32_t *__restrict__ points, B3wCurve C) {
  const uint32_t s = blockIdx.x, t = threadIdx.x;
  if (s >= nslots) return;
  const uint32_t nb = nbits[s], v0 = first_v[s];
  if (t >= nb) return;
  Fp r2;
#pragma unroll
  for (int i = 0; i < 8; ++i) r2.l[i] = C.r2[i];
  Jac P;
  P.X = fp_mul(load_fp(gens + (uint64_t)s * 16), r2, C);
  P.Y = fp_mul(load_fp(gens + (uint64_t)s * 16 + 8), r2, C);
#pragma unroll
  for (int i = 0; i < 8; ++i) P.Z.l[i] = C.one[i];
  Fp c32;                                                 // 32 in Montgomery form: the other kernels' radix is 2^261
#pragma unroll
  for (int k = 0; k < 8; ++k) c32.l[k] = C.one[k];
#pragma unroll 1
  for (int k = 0; k < 5; ++k) c32 = fp_dbl(c32, C);
#pragma unroll 1
  for (uint32_t i = 0; i < t; ++i) P = jac_dbl(P, C);
#pragma unroll 1
  for (uint32_t k = t; k < nb; k += 64) {
    Fp x, y;
    if (k == 0) { x = P.X; y = P.Y; } else jac_to_affine(P, x, y, C);
    store_fp(points + (uint64_t)(v0 + k) * 16, fp_mul(x, c32, C));
    store_fp(points + (uint64_t)(v0 + k) * 16 + 8, fp_mul(y, c32, C));
    if (k + 64 < nb)
#pragma unroll 1
      for (int i = 0; i < 64; ++i) P = jac_dbl(P, C);
  }
}

// ---- set-up 1b (O2 nova circuits): invtab[(j * 2 + neg) * nk + mag - 1] = (+-1 / mag mod r) * G of the slot that holds the inverse of
// IsZero gadget j, for mag = 1 ... nk (affine, radix 2^261, eight 32-bit words; (0, 0) where that slot is not committed).  A step has 67
// such slots — 256 virtual bit slots, i.e. sixteen 16-bit windows each, 39 % of what a folded O2 key commits — but their values
// are 1 / k for a SMALL signed k the step's inputs determine (k = -depth, -block_count, n_blocks - 1 - block_count,
// total_depth - i - 2 - depth: circuits/blake3_nova.circom:19-23,65-72,136-144), the same 2 048-entry table of inverses the
// witness kernels use: records mode adds ONE tabulated point per gadget instead of sixteen.
// One thread per entry: 255 doublings and the additions of the scalar's bits (set-up only: 8 ms for 67 x 2 047 entries).
__global__ __launch_bounds__(64) void b3w_commit_invtab_kernel(const uint32_t *__restrict__ gens /* nslots x 16 words, standard form */,
                                                               const uint32_t *__restrict__ inv_slot /* 67: committed slot index or ~0 */,
                                                               const uint32_t *__restrict__ inverses /* 8 words per magnitude, standard form; entry 0 unused */,
                                                               uint32_t njobs, uint32_t nk, uint32_t *__restrict__ invtab, B3wCurve C) {
  const uint64_t gid = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (gid >= (uint64_t)njobs * nk) return;
  const uint32_t j = (uint32_t)(gid / nk), mag = (uint32_t)(gid % nk) + 1u;
  uint32_t *o = invtab + ((uint64_t)j * 2 * nk + mag - 1u) * 16, *on = o + (uint64_t)nk * 16;        // 1 / mag, -1 / mag
  const uint32_t s = inv_slot[j];
  if (s == 0xFFFFFFFFu) { store_fp(o, fp_zero()); store_fp(o + 8, fp_zero()); store_fp(on, fp_zero()); store_fp(on + 8, fp_zero()); return; }
  Fp r2;
#pragma unroll
  for (int i = 0; i < 8; ++i) r2.l[i] = C.r2[i];
  const Fp gx = fp_mul(load_fp(gens + (uint64_t)s * 16), r2, C), gy = fp_mul(load_fp(gens + (uint64_t)s * 16 + 8), r2, C);
  const uint32_t *sc = inverses + 8ull * mag;
  Jac R = jac_infinity();
#pragma unroll 1
  for (int b = 255; b >= 0; --b) {
    R = jac_dbl(R, C);
    if ((sc[b >> 5] >> (b & 31)) & 1u) R = jac_madd(R, gx, gy, C);
  }
  Fp x, y, c32;
  jac_to_affine(R, x, y, C);
#pragma unroll
  for (int k = 0; k < 8; ++k) c32.l[k] = C.one[k];
#pragma unroll 1
  for (int k = 0; k < 5; ++k) c32 = fp_dbl(c32, C);         // 32 in Montgomery form: the commit kernel's radix is 2^261
  const Fp xo = fp_mul(x, c32, C);
  store_fp(o, xo);
  store_fp(o + 8, fp_mul(y, c32, C));
  store_fp(on, xo);
  store_fp(on + 8, fp_mul(fp_sub(fp_zero(), y, C), c32, C));
}

// ---- set-up 2: table[win * (2^W - 1) + m - 1] = sum of the window's virtual-slot points selected by the bits of m
// (affine, radix 2^261, eight 32-bit words; infinity = (0, 0)).  A thread walks K = 4 consecutive entries in Gray-code
// order — the first from scratch, each next one is the previous +/- one point — and the four share one inversion
// (Montgomery's trick), so an entry costs ~3 additions and a quarter of an inversion instead of W/2 additions and a
// whole one.  The thread's projective results and prefix products wait in LDS ([entry][limb][thread]: conflict-free).
#define B3W_WINDOW_K 4
template <int W>
__global__ __launch_bounds__(64) void b3w_commit_window_kernel(const uint32_t *__restrict__ points, uint32_t nwin,
                                                               uint32_t *__restrict__ table, B3wCurve C, B3wCurve9 C9) {
  constexpr int K = B3W_WINDOW_K;
  constexpr uint32_t CHUNKS = (1u << W) / K;                  // per window
  __shared__ uint32_t lds[5 * K * 9 * 64];                    // X, Y, ZZ, ZZZ, prefix product of the ZZ * ZZZ
  const uint32_t tid = threadIdx.x;
  const uint64_t gid = (uint64_t)blockIdx.x * 64 + tid;
  if (gid >= (uint64_t)nwin * CHUNKS) return;                 // no barriers below: every thread uses its own LDS column
  const uint32_t win = (uint32_t)(gid / CHUNKS), g0 = (uint32_t)(gid % CHUNKS) * K;
  auto at = [&](int what, int e, int l) -> uint32_t & { return lds[((what * K + e) * 9 + l) * 64 + tid]; };
  auto point = [&](uint32_t b, bool negate, F9 &x, F9 &y) {
    const uint32_t *pt = points + (uint64_t)(win * W + b) * 16;
    Fp py = load_fp(pt + 8);
    if (negate) py = fp_sub(fp_zero(), py, C);
    x = to29(load_fp(pt)); y = to29(py);
  };
  J9 acc = j9_infinity();
  const uint32_t m0 = g0 ^ (g0 >> 1);
#pragma unroll 1
  for (uint32_t b = 0; b < (uint32_t)W; ++b) {
    if (!((m0 >> b) & 1)) continue;
    F9 x, y;
    point(b, false, x, y);
    j9_madd(acc, x, y, C9);
  }
#pragma unroll 1
  for (int e = 0; e < K; ++e) {
    if (e > 0) {                                              // Gray code: entry g differs from g - 1 in bit ctz(g)
      const uint32_t g = g0 + e, m = g ^ (g >> 1), b = (uint32_t)__builtin_ctz(g);
      F9 x, y;
      point(b, !((m >> b) & 1), x, y);
      j9_madd(acc, x, y, C9);
    }
    const F9 A = acc.inf ? one29(C9) : mul29(acc.ZZ, acc.ZZZ, C9);      // what has to be inverted: 1/ZZ = ZZZ / A, 1/ZZZ = ZZ / A
#pragma unroll
    for (int l = 0; l < 9; ++l) {
      at(0, e, l) = acc.X.l[l]; at(1, e, l) = acc.Y.l[l];
      at(2, e, l) = acc.inf ? 0u : acc.ZZ.l[l]; at(3, e, l) = acc.inf ? 0u : acc.ZZZ.l[l];
    }
    F9 c;                                                     // prefix product of the (finite) As
    if (e == 0) c = A;
    else {
#pragma unroll
      for (int l = 0; l < 9; ++l) c.l[l] = at(4, e - 1, l);
      c = mul29(c, A, C9);
    }
#pragma unroll
    for (int l = 0; l < 9; ++l) at(4, e, l) = c.l[l];
  }
  F9 inv;
#pragma unroll
  for (int l = 0; l < 9; ++l) inv.l[l] = at(4, K - 1, l);
  inv = inv29(inv, C.pm2, C9);                                // 1 / (A_0 ... A_{K-1})
#pragma unroll 1
  for (int e = K - 1; e >= 0; --e) {
    F9 X, Y, ZZ, ZZZ, ia = inv;
    uint32_t z = 0;
#pragma unroll
    for (int l = 0; l < 9; ++l) { X.l[l] = at(0, e, l); Y.l[l] = at(1, e, l); ZZ.l[l] = at(2, e, l); ZZZ.l[l] = at(3, e, l); z |= ZZ.l[l]; }
    if (z) {                                                  // finite: it took part in the product
      if (e > 0) {
        F9 c;
#pragma unroll
        for (int l = 0; l < 9; ++l) c.l[l] = at(4, e - 1, l);
        ia = mul29(inv, c, C9);                               // 1 / A_e
      }
      inv = mul29(inv, mul29(ZZ, ZZZ, C9), C9);
    }
    const uint32_t g = g0 + e, m = g ^ (g >> 1);
    if (m == 0) continue;                                     // the empty subset has no entry
    Fp x = fp_zero(), y = fp_zero();
    if (z) {
      uint32_t hi;
      x = from29(mul29(X, mul29(ia, ZZZ, C9), C9), hi); x = fp_reduce_once(x, hi, C);
      y = from29(mul29(Y, mul29(ia, ZZ, C9), C9), hi); y = fp_reduce_once(y, hi, C);
    }
    uint32_t *o = table + ((uint64_t)win * B3W_COMMIT_ENTRIES(W) + m - 1) * 16;
    store_fp(o, x);
    store_fp(o + 8, y);
  }
}

// ---- commit: T lanes per witness.
//   phase 1  the lanes stream the body once (coalesced, 32 bytes per lane and step) and assemble the witness's
//            virtual-slot bits as one bit string in LDS (bit v = virtual slot v; 6.7 KB for the compression circuit);
//            the domain check lives here, where every word of an element is at hand anyway.  This phase waits on HBM,
//            the next one on the ALU: waves of different workgroups in different phases share a SIMD and overlap.
//            (Gathering the bits window by window — 12-16 dependent scattered loads each — cost 43 % of the kernel;
//            as a separate streaming kernel in front, 30 %.)
//   phase 2  lane t adds the tabulated points of the witness's non-zero windows t, t + T, ...
//   phase 3  LDS tree over the T partial sums (in the bit string's place).
template <int T, int WPB, int W, class CV>        // threads per witness, witnesses per workgroup (T * WPB threads), window width, field
__global__ __launch_bounds__(T * WPB) __attribute__((amdgpu_waves_per_eu(3, 3))) void b3w_commit_kernel(const uint8_t *__restrict__ bodies, uint32_t n, uint64_t pitch,
                                                         uint32_t first_slot, uint32_t nslots,
                                                         const uint32_t *__restrict__ slotdesc /* per committed slot: first virtual slot | width code << 24 */,
                                                         const uint32_t *__restrict__ images /* or null: TRACE images, word j of witness w at [j * img_row + w] */,
                                                         uint32_t img_row, const uint2 *__restrict__ runs, uint32_t nruns,
                                                         uint32_t region_words /* LDS words per witness: max(bit string, 36 T) */,
                                                         const uint32_t *__restrict__ table /* radix 2^261 */, uint32_t nwin,
                                                         uint32_t *__restrict__ sums /* n x B3W_COMMIT_SUM_WORDS: X Y ZZ ZZZ in 29-bit limbs */,
                                                         int32_t *__restrict__ status,
                                                         const uint32_t *__restrict__ invtab /* or null: the O2 nova circuits' inverse points, b3w_commit_invtab_kernel */,
                                                         uint32_t inv_nk, const uint32_t *__restrict__ invmeta /* bodies mode: gadget slots, input slots, first virtual slots */,
                                                         const uint32_t *__restrict__ aux /* bodies mode: the prime, then 1 / k as scalars */,
                                                         unsigned long long *__restrict__ adds /* or null: += mixed additions of phase 2 (b3w_commit_key_counts) */, CV C) {
  extern __shared__ uint32_t lds[];
  __shared__ uint32_t bad[WPB];
  // invtab: per IsZero gadget of the step 0 = nothing to add (k = 0: the inverse is 0; or a rejected step), +-mag = the tabulated
  // point of 1 / k (both signs are in the table), INV_WINDOWS = k beyond the table: the slot's bits go through the windows like any other slot's
  constexpr int32_t INV_WINDOWS = (int32_t)0x80000000;
  __shared__ int32_t kinv[WPB][B3W_NOVA_ISZERO];
  const uint32_t sub = threadIdx.x / T, t = threadIdx.x % T;       // which witness of the workgroup, lane within it
  const uint32_t w = blockIdx.x * WPB + sub;
  const bool live = w < n;
  uint32_t *packed = lds + sub * region_words;
  for (uint32_t i = t; i < region_words; i += T) packed[i] = 0;
  if (t == 0) bad[sub] = 0;
  if (images && invtab) {
    const uint32_t *img = images + (live ? w : 0);
    const bool ok = live && img[(uint64_t)B3W_LDS_OKWORD * img_row] != 0;
    for (uint32_t j = t; j < B3W_NOVA_ISZERO; j += T) {        // the gadgets' arguments, as the TRACE phase of the witness kernel derives them
      const int64_t depth = img[(uint64_t)(B3W_LDS_NV + NV_DEPTH) * img_row];
      int64_t k;
      if (j == 0) k = -depth;
      else if (j == 1) k = -(int64_t)img[(uint64_t)(B3W_LDS_NV + NV_BLOCK_COUNT) * img_row];
      else if (j == 2) k = (int64_t)img[(uint64_t)(B3W_LDS_NV + NV_N_BLOCKS) * img_row] - 1 - (int64_t)img[(uint64_t)(B3W_LDS_NV + NV_BLOCK_COUNT) * img_row];
      else k = (int64_t)img[(uint64_t)(B3W_LDS_NV + NV_TOTAL_DEPTH) * img_row] - (int64_t)(j - 3) - 2 - depth;
      const uint64_t mag = k < 0 ? (uint64_t)(-k) : (uint64_t)k;
      kinv[sub][j] = !ok || mag == 0 ? 0 : mag > inv_nk ? INV_WINDOWS : k < 0 ? -(int32_t)mag : (int32_t)mag;
    }
  } else if (!images) {
    // bodies mode: the same arguments from the body's four input slots — and the point is taken only if the body's inverse slot
    // IS +-1 / k (the context's scalar table), whatever the rest of the body says; everything else goes through the windows
    const uint4 *wbody = reinterpret_cast<const uint4 *>(bodies + (uint64_t)(live ? w : 0) * pitch);
    uint32_t inw[4] = {0, 0, 0, 0};
    bool words = live && invtab != nullptr;
    if (words)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t sidx = invmeta[B3W_NOVA_ISZERO + q];
        const uint4 a = wbody[(uint64_t)sidx * 2], b = wbody[(uint64_t)sidx * 2 + 1];
        inw[q] = a.x;
        words = words && !(a.y | a.z | a.w | b.x | b.y | b.z | b.w);
      }
    for (uint32_t j = t; j < B3W_NOVA_ISZERO; j += T) {
      int32_t res = INV_WINDOWS;
      const uint32_t sl = words ? invmeta[j] : 0xFFFFFFFFu;
      const int64_t depth = inw[3];
      const int64_t k = j == 0 ? -depth : j == 1 ? -(int64_t)inw[1] : j == 2 ? (int64_t)inw[0] - 1 - (int64_t)inw[1] : (int64_t)inw[2] - (int64_t)(j - 3) - 2 - depth;
      const uint64_t mag = k < 0 ? (uint64_t)(-k) : (uint64_t)k;
      if (sl != 0xFFFFFFFFu && mag <= inv_nk) {
        const uint4 va = wbody[((uint64_t)first_slot + sl) * 2], vb = wbody[((uint64_t)first_slot + sl) * 2 + 1];
        const uint32_t v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
        uint32_t diff = 0;
        if (mag == 0) {                                        // IsZero(0): the inverse slot holds 0 — nothing to add
#pragma unroll
          for (int i = 0; i < 8; i++) diff |= v[i];
          if (!diff) res = 0;
        } else {
          const uint32_t *e = aux + 16 + 8 * mag;              // 1 / mag; a negative argument's inverse is p - 1 / mag
          uint64_t br = 0;
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const uint64_t d = (uint64_t)aux[i] - e[i] - br;
            br = (d >> 63) & 1;
            diff |= v[i] ^ (k < 0 ? (uint32_t)d : e[i]);
          }
          if (!diff) res = k < 0 ? -(int32_t)mag : (int32_t)mag;
        }
      }
      kinv[sub][j] = res;
    }
  }
  __syncthreads();
  if (images) {
    // records mode: the witness is an expansion of its TRACE image (3.7-11 KB) through the slot table, so its bits are
    // pieces of image words: run r = `len` bits of image word `src` from bit `sh` on, at bit v0 of the bit string
    const uint32_t *img = images + (live ? w : 0);
    const bool ok = live && img[(uint64_t)B3W_LDS_OKWORD * img_row] != 0;              // a rejected step leaves the string empty
    for (uint32_t r = t; r < nruns; r += T) {
      const uint2 e = runs[r];
      if (invtab) {                                          // an inverse whose point is tabulated stays out of the bit string
        const uint32_t src = e.y & 0xFFFFu;
        if (src >= B3W_LDS_WIDE && src < B3W_LDS_WIDE + 8u * B3W_NOVA_ISZERO && kinv[sub][(src - B3W_LDS_WIDE) >> 3] != INV_WINDOWS) continue;
      }
      const uint32_t len = (e.x >> 24) + 1u, x = ok ? img[(uint64_t)(e.y & 0xFFFFu) * img_row] >> (e.y >> 16) : 0u;
      const uint32_t piece = len == 32 ? x : x & ((1u << len) - 1u);
      if (piece) {
        const uint32_t v = e.x & 0xFFFFFFu;
        atomicOr(&packed[v >> 5], piece << (v & 31));
        if (v & 31) atomicOr(&packed[(v >> 5) + 1], piece >> (32 - (v & 31)));
      }
    }
  } else {
    const uint4 *body = reinterpret_cast<const uint4 *>(bodies + (uint64_t)(live ? w : 0) * pitch) + (uint64_t)first_slot * 2;
    auto put32 = [&](uint32_t v, uint32_t x) {
      if (!x) return;
      atomicOr(&packed[v >> 5], x << (v & 31));
      if (v & 31) atomicOr(&packed[(v >> 5) + 1], x >> (32 - (v & 31)));
    };
    const uint64_t mine = T == 64 ? ~0ull : (threadIdx.x & 32) ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;   // my witness's lanes of the wave
    bool wrong = false;
    constexpr int U = 8;                                        // steps in flight: 8 x (32 + 4) bytes per lane
#pragma unroll 1
    for (uint32_t s0 = 0; s0 < nslots; s0 += T * U) {
      uint4 a[U], b[U];
      uint32_t d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t sl = s0 + u * T + t;
        a[u] = make_uint4(0, 0, 0, 0); b[u] = a[u]; d[u] = 0xFFFFFFFFu;          // code 255: not a slot
        if (live && sl < nslots) { a[u] = body[(uint64_t)sl * 2]; b[u] = body[(uint64_t)sl * 2 + 1]; d[u] = slotdesc[sl]; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t v0 = d[u] & 0xFFFFFFu, code = d[u] >> 24, hi = b[u].x | b[u].y | b[u].z | b[u].w;
        const uint32_t vbase = __shfl(v0, 0, T);
        const uint64_t run = __ballot(code == 0 && v0 == vbase + t), ones = __ballot(a[u].x & 1);
        if ((run & mine) == mine) {                             // T consecutive bit slots: one ballot
          wrong |= a[u].x > 1 || (a[u].y | a[u].z | a[u].w | hi);
          if (t == 0) {
            const uint64_t mk = T == 64 ? ones : (threadIdx.x & 32) ? ones >> 32 : ones & 0xFFFFFFFFull;
            put32(vbase, (uint32_t)mk);
            if (T == 64) put32(vbase + 32, (uint32_t)(mk >> 32));
          }
        } else if (code == 0) { wrong |= a[u].x > 1 || (a[u].y | a[u].z | a[u].w | hi); put32(v0, a[u].x & 1); }
        else if (code == 1) { wrong |= (a[u].y | a[u].z | a[u].w | hi) != 0; put32(v0, a[u].x); }
        else if (code == 2) { wrong |= (a[u].z | a[u].w | hi) != 0; put32(v0, a[u].x); put32(v0 + 32, a[u].y); }
        else if (code == 3) { put32(v0, a[u].x); put32(v0 + 32, a[u].y); put32(v0 + 64, a[u].z); put32(v0 + 96, a[u].w);
                              put32(v0 + 128, b[u].x); put32(v0 + 160, b[u].y); put32(v0 + 192, b[u].z); put32(v0 + 224, b[u].w); }
        else if (code == 5) {                                  // a nova inverse slot (v0 = the gadget): tabulated, or eight words like any 256-bit slot
          if (kinv[sub][v0] == INV_WINDOWS) {
            const uint32_t vv = invmeta ? invmeta[B3W_NOVA_ISZERO + 4 + v0] : 0u;
            put32(vv, a[u].x); put32(vv + 32, a[u].y); put32(vv + 64, a[u].z); put32(vv + 96, a[u].w);
            put32(vv + 128, b[u].x); put32(vv + 160, b[u].y); put32(vv + 192, b[u].z); put32(vv + 224, b[u].w);
          }
        }
        else if (code >= 8 && code < 40) { wrong |= (a[u].y | a[u].z | a[u].w | hi) != 0; put32(v0, (a[u].x >> (code - 8)) & 1u); }   // folded key: one bit of a word
        // (code 4: a slot folded into others' generators — nothing to add)
      }
    }
    if (wrong) bad[sub] = 1;
  }
  __syncthreads();
  J9 acc = j9_infinity();
  uint32_t win = live ? t : nwin, nadd = 0;
  while (true) {
    // skip ahead to this lane's next window with a set bit
    uint32_t m = 0;
    while (win < nwin) {
      if (W == 16) m = (packed[win >> 1] >> ((win & 1) * 16)) & 0xFFFFu;
      else {
        const uint32_t bit = win * W;
        m = (uint32_t)(((((uint64_t)packed[(bit >> 5) + 1]) << 32 | packed[bit >> 5]) >> (bit & 31)) & ((1u << W) - 1u));
      }
      if (m) break;
      win += T;
    }
    const bool has = win < nwin;
    if (!__any(has)) break;
    if (has) {
      const uint32_t *pt = table + ((uint64_t)win * B3W_COMMIT_ENTRIES(W) + m - 1) * 16;
      const Fp x2 = load_fp(pt), y2 = load_fp(pt + 8);
      if (!(fp_is_zero(x2) && fp_is_zero(y2))) {           // (0, 0) is not on these curves (b != 0): the table's infinity
        j9_madd(acc, to29(x2), to29(y2), C);
        nadd++;
      }
      win += T;
    }
  }
  if (invtab && live)
    for (uint32_t j = t; j < B3W_NOVA_ISZERO; j += T) {        // one tabulated point per IsZero gadget (y negated for a negative argument)
      const int32_t kc = kinv[sub][j];
      if (kc == 0 || kc == INV_WINDOWS) continue;
      const uint32_t *pt = invtab + (((uint64_t)j * 2 + (kc < 0 ? 1u : 0u)) * inv_nk + (uint32_t)(kc < 0 ? -kc : kc) - 1u) * 16;
      const Fp x2 = load_fp(pt), y2 = load_fp(pt + 8);
      if (fp_is_zero(x2) && fp_is_zero(y2)) continue;          // (the gadget's slot is not committed)
      j9_madd(acc, to29(x2), to29(y2), C);
      nadd++;
    }
  if (adds) {                                                // (uniform: a statistics pass, b3w_commit_key_count)
    for (int o = 32; o >= 1; o >>= 1) nadd += __shfl_xor(nadd, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(adds, (unsigned long long)nadd);
  }
  __syncthreads();                                           // every lane is done with the bit string: the tree takes its place
  // LDS tree over each witness's T partial sums (infinity travels as ZZ = 0).  The additions of all WPB witnesses of a
  // level are dealt to the first threads of the workgroup, so whole waves drop out instead of running half empty
  // (one witness per wave spent 6 nearly empty wave-wide additions here: 10 % of the kernel).
  auto put = [&](uint32_t *dst, const J9 &a) {
#pragma unroll
    for (int i = 0; i < 9; ++i) { dst[i] = a.X.l[i]; dst[9 + i] = a.Y.l[i]; dst[18 + i] = a.inf ? 0u : a.ZZ.l[i]; dst[27 + i] = a.inf ? 0u : a.ZZZ.l[i]; }
  };
  auto get = [&](const uint32_t *src) {
    J9 a;
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { a.X.l[i] = src[i]; a.Y.l[i] = src[9 + i]; a.ZZ.l[i] = src[18 + i]; a.ZZZ.l[i] = src[27 + i]; z |= src[18 + i]; }
    a.inf = z == 0;
    return a;
  };
  put(packed + t * 36, acc);
  __syncthreads();
  for (uint32_t st = T / 2; st >= 1; st >>= 1) {
    if (threadIdx.x < WPB * st) {
      uint32_t *m = lds + (threadIdx.x / st) * region_words + (threadIdx.x % st) * 36;
      J9 a = get(m);
      j9_add(a, get(m + st * 36), C);
      put(m, a);
    }
    __syncthreads();
  }
  if (live) for (uint32_t i = t; i < 36; i += T) sums[(uint64_t)w * B3W_COMMIT_SUM_WORDS + i] = packed[i];   // normalised by the next kernel
  if (live && t == 0 && status) status[w] = bad[sub] ? 103 : 0;
}

// ---- normalise: one THREAD per witness (a Fermat inversion is 380 dependent multiplications: on thread 0 of the commit
// workgroup it took longer than the workgroup's whole share of additions); affine, standard form, 64 bytes per point
template <class CV>
__global__ __launch_bounds__(64) void b3w_commit_normalize_kernel(const uint32_t *__restrict__ sums, uint32_t n, uint8_t *__restrict__ out,
                                                                  B3wCurve C, CV C9) {
  const uint32_t w = blockIdx.x * 64 + threadIdx.x;
  if (w >= n) return;
  F9 co[4];                                                  // X, Y, ZZ, ZZZ (tidy < 2p)
  uint32_t z = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 9; ++i) co[c].l[i] = sums[(uint64_t)w * B3W_COMMIT_SUM_WORDS + c * 9 + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) z |= co[2].l[i];
  Fp x = fp_zero(), y = fp_zero();                           // infinity -> (0, 0)
  if (z) {
    const F9 ia = inv29(mul29(co[2], co[3], C9), C.pm2, C9);  // 1 / (ZZ * ZZZ): 1/ZZ = ZZZ * ia, 1/ZZZ = ZZ * ia
    F9 unit;
#pragma unroll
    for (int i = 0; i < 9; ++i) unit.l[i] = i == 0 ? 1u : 0u;
    const F9 xs = mul29(mul29(co[0], mul29(ia, co[3], C9), C9), unit, C9);    // out of Montgomery form: < 2p
    const F9 ys = mul29(mul29(co[1], mul29(ia, co[2], C9), C9), unit, C9);
    uint32_t hi;
    x = from29(xs, hi); x = fp_reduce_once(x, hi, C);
    y = from29(ys, hi); y = fp_reduce_once(y, hi, C);
  }
  uint32_t *o = reinterpret_cast<uint32_t *>(out + (uint64_t)w * 64);
  store_fp(o, x);
  store_fp(o + 8, y);
}

// ---- normalise MANY: a thread takes B3W_NORMALIZE_K consecutive witnesses and inverts the product of their ZZ * ZZZ once
// (Montgomery's trick, as the window kernel above): 95 + 14 multiplications a witness instead of 390, and — the point — launched once
// for all the witnesses of a pass it fills the machine, where the per-batch launch was a chain of dependent multiplications on one
// wave per CU.  The sums are read twice (144 bytes a witness).
#define B3W_NORMALIZE_K 4
template <class CV>
__global__ __launch_bounds__(64) void b3w_commit_normalize_many_kernel(const uint32_t *__restrict__ sums, uint64_t n, uint8_t *__restrict__ out,
                                                                       B3wCurve C, CV C9) {
  constexpr int K = B3W_NORMALIZE_K;
  const uint64_t w0 = ((uint64_t)blockIdx.x * 64 + threadIdx.x) * K;
  if (w0 >= n) return;
  auto coord = [&](uint64_t w, int c) {
    F9 v;
#pragma unroll
    for (int i = 0; i < 9; ++i) v.l[i] = sums[w * B3W_COMMIT_SUM_WORDS + c * 9 + i];
    return v;
  };
  auto is_zero = [](const F9 &v) { uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) z |= v.l[i];
    return z == 0; };
  F9 pre[K];                                                 // prefix products of the finite witnesses' ZZ * ZZZ (infinity: the product so far)
#pragma unroll
  for (int e = 0; e < K; ++e) {
    const uint64_t w = w0 + e;
    F9 a = one29(C9);
    if (w < n) {
      const F9 zz = coord(w, 2);
      if (!is_zero(zz)) a = mul29(zz, coord(w, 3), C9);
    }
    pre[e] = e == 0 ? a : mul29(pre[e - 1], a, C9);
  }
  F9 inv = inv29(pre[K - 1], C.pm2, C9);                      // 1 / (A_0 ... A_{K-1})
  F9 unit;
#pragma unroll
  for (int i = 0; i < 9; ++i) unit.l[i] = i == 0 ? 1u : 0u;
#pragma unroll
  for (int e = K - 1; e >= 0; --e) {
    const uint64_t w = w0 + e;
    if (w >= n) continue;                                      // (its A was one: nothing to take out of `inv`)
    const F9 zz = coord(w, 2);
    Fp x = fp_zero(), y = fp_zero();                           // infinity -> (0, 0)
    if (!is_zero(zz)) {
      const F9 zzz = coord(w, 3);
      const F9 ia = e > 0 ? mul29(inv, pre[e - 1], C9) : inv;  // 1 / A_e
      inv = mul29(inv, mul29(zz, zzz, C9), C9);
      const F9 xs = mul29(mul29(coord(w, 0), mul29(ia, zzz, C9), C9), unit, C9);    // 1 / ZZ = ZZZ / A; out of Montgomery form: < 2p
      const F9 ys = mul29(mul29(coord(w, 1), mul29(ia, zz, C9), C9), unit, C9);     // 1 / ZZZ = ZZ / A
      uint32_t hi;
      x = from29(xs, hi); x = fp_reduce_once(x, hi, C);
      y = from29(ys, hi); y = fp_reduce_once(y, hi, C);
    }
    uint32_t *o = reinterpret_cast<uint32_t *>(out + w * 64);
    store_fp(o, x);
    store_fp(o + 8, y);
  }
}

// host: the 29-bit constants of a curve
void u288_split29(const uint32_t w[9], uint32_t out[9]) {
  for (int k = 0; k < 9; ++k) {
    const int bit = 29 * k, i = bit >> 5, sh = bit & 31;
    uint64_t v = w[i];
    if (i + 1 < 9) v |= (uint64_t)w[i + 1] << 32;
    out[k] = (uint32_t)(v >> sh) & (k < 8 ? M29 : 0xFFFFFFFFu);
  }
}
bool u256_geq_h(const uint32_t a[8], const uint32_t b[8]) {
  for (int i = 7; i >= 0; --i) if (a[i] != b[i]) return a[i] > b[i];
  return true;
}
void u256_dbl_mod_h(uint32_t a[8], const uint32_t p[8]) {
  uint32_t c = 0;
  for (int i = 0; i < 8; ++i) { const uint32_t n = (a[i] << 1) | c; c = a[i] >> 31; a[i] = n; }
  if (c || u256_geq_h(a, p)) { uint64_t br = 0; for (int i = 0; i < 8; ++i) { const uint64_t t = (uint64_t)a[i] - p[i] - br; a[i] = (uint32_t)t; br = (t >> 63) & 1; } }
}
B3wCurve9 make_curve9(const B3wCurve &C) {
  B3wCurve9 D{};
  uint32_t w[9];
  for (int i = 0; i < 8; ++i) w[i] = C.p[i];
  w[8] = 0;
  u288_split29(w, D.p);
  uint32_t p4[9];                                              // 4p
  for (int i = 0; i < 9; ++i) p4[i] = (i < 8 ? C.p[i] << 2 : 0u) | (i > 0 ? C.p[i - 1] >> 30 : 0u);
  u288_split29(p4, D.sp4);
  for (int i = 0; i < 8; ++i) D.sp4[i] += (1u << 29) - (i > 0 ? 1u : 0u);
  D.sp4[8] -= 1u;
  uint32_t x[8];
  for (int i = 0; i < 8; ++i) x[i] = C.one[i];
  for (int k = 0; k < 5; ++k) u256_dbl_mod_h(x, C.p);           // 2^261 mod p
  for (int i = 0; i < 8; ++i) w[i] = x[i];
  w[8] = 0;
  u288_split29(w, D.one);
  D.inv = C.inv & M29;
  const uint64_t phi = ((uint64_t)C.p[7] << 32) | C.p[6];      // p >> 192
  D.mu = (uint32_t)((((unsigned __int128)1) << 77) / ((unsigned __int128)phi + 1));
  for (uint32_t k = 3; k <= 5; ++k) D.kp0[k - 3] = (k * D.p[0]) & M29;
  return D;
}

}  // namespace

extern "C" int b3w_launch_commit_setup(const uint32_t *d_gens, const uint32_t *d_first_v, const uint32_t *d_nbits, uint32_t nslots,
                                       uint32_t *d_points, const B3wCurve *curve, hipStream_t stream) {
  if (!nslots) return 0;
  hipLaunchKernelGGL(b3w_commit_setup_kernel, dim3(nslots), dim3(64), 0, stream, d_gens, d_first_v, d_nbits, nslots, d_points, *curve);
  return (int)hipGetLastError();
}

extern "C" int b3w_launch_commit_windows(const uint32_t *d_points, uint32_t nwin, uint32_t window, uint32_t *d_table, const B3wCurve *curve, hipStream_t stream) {
  if (!nwin) return 0;
  if (!B3W_COMMIT_WINDOW_OK(window)) return (int)hipErrorInvalidValue;
  const uint64_t total = (uint64_t)nwin * ((1u << window) / B3W_WINDOW_K);      // threads: B3W_WINDOW_K entries each
  const dim3 grid((uint32_t)((total + 63) / 64));
  const B3wCurve9 c9 = make_curve9(*curve);
  if (window == B3W_COMMIT_WINDOW_XL)
    hipLaunchKernelGGL(b3w_commit_window_kernel<B3W_COMMIT_WINDOW_XL>, grid, dim3(64), 0, stream, d_points, nwin, d_table, *curve, c9);
  else if (window == B3W_COMMIT_WINDOW_LARGE)
    hipLaunchKernelGGL(b3w_commit_window_kernel<B3W_COMMIT_WINDOW_LARGE>, grid, dim3(64), 0, stream, d_points, nwin, d_table, *curve, c9);
  else
    hipLaunchKernelGGL(b3w_commit_window_kernel<B3W_COMMIT_WINDOW_SMALL>, grid, dim3(64), 0, stream, d_points, nwin, d_table, *curve, c9);
  return (int)hipGetLastError();
}

extern "C" int b3w_launch_commit(const uint8_t *d_bodies, uint32_t n, uint64_t pitch, uint32_t first_slot, uint32_t nslots,
                                 const uint32_t *d_slotdesc, const uint32_t *d_images /* or null */, uint32_t img_row, const uint32_t *d_runs,
                                 uint32_t nruns, const uint32_t *d_table, uint32_t nwin, uint32_t window,
                                 uint32_t *d_sums /* n * B3W_COMMIT_SUM_WORDS scratch */, uint8_t *d_out, int32_t *d_status,
                                 const uint32_t *d_invtab, uint32_t inv_nk, const uint32_t *d_invmeta, const uint32_t *d_aux,
                                 unsigned long long *d_adds, const B3wCurve *curve, hipStream_t stream) {
  if (!n) return 0;
  if (!d_images && !(d_invtab && d_invmeta && d_aux)) { d_invtab = nullptr; d_invmeta = nullptr; }      // (bodies mode needs all three)
  if (!B3W_COMMIT_WINDOW_OK(window)) return (int)hipErrorInvalidValue;
  const uint32_t bits_words = (nwin * window + 31) / 32 + 2;   // one packed witness in LDS (nova O1: 13.5 KB)
  if (bits_words * 4 > 32 * 1024) return (int)hipErrorInvalidValue;
  const B3wCurve9 c9 = make_curve9(*curve);
  // lanes per witness: 32 (two witnesses per wave) for small batches of the compression circuit, 64 for large ones and for
  // the longer nova witnesses (measured at both window widths).  Virtual slots: 53 k compression, 58 k nova O2, 108 k nova O1.
  static const int env_tpw = getenv("B3W_COMMIT_THREADS") ? atoi(getenv("B3W_COMMIT_THREADS")) : 0;
  const int tpw = window > B3W_COMMIT_WINDOW_LARGE ? 64 : env_tpw ? env_tpw : ((uint64_t)nwin * window > 55200 || n >= 8192) ? 64 : 32;
  bool vesta = true;                                           // the modulus with compile-time limbs?
  for (int i = 0; i < 9; ++i) vesta = vesta && c9.p[i] == B3wCurve9Vesta::P(i);
  vesta = vesta && c9.inv == B3wCurve9Vesta::INV() && tpw == 64;
  B3wCurve9Vesta c9v;
  static_cast<B3wCurve9 &>(c9v) = c9;
#define B3W_COMMIT_LAUNCH(T, WPB, W, CV, cv)                                                                              \
  {                                                                                                                       \
    const uint32_t region = bits_words > 36u * T ? bits_words : 36u * T;                                                  \
    if ((size_t)region * WPB * 4 > 64 * 1024) return (int)hipErrorInvalidValue;   /* LDS of one workgroup (nova O1: 54 KB) */ \
    hipLaunchKernelGGL((b3w_commit_kernel<T, WPB, W, CV>), dim3((n + WPB - 1) / WPB), dim3(T * WPB), region * WPB * 4, stream, d_bodies, n, \
                       pitch, first_slot, nslots, d_slotdesc, d_images, img_row, reinterpret_cast<const uint2 *>(d_runs), nruns, region, d_table, nwin, \
                       d_sums, d_status, d_invtab, inv_nk, d_invmeta, d_aux, d_adds, cv);                                         \
  }
  if (window == B3W_COMMIT_WINDOW_XL) {                        // (the wide table: 64 lanes a witness only)
    if (vesta) B3W_COMMIT_LAUNCH(64, 4, B3W_COMMIT_WINDOW_XL, B3wCurve9Vesta, c9v)
    else B3W_COMMIT_LAUNCH(64, 4, B3W_COMMIT_WINDOW_XL, B3wCurve9, c9)
  } else if (window == B3W_COMMIT_WINDOW_LARGE) {
    if (vesta) B3W_COMMIT_LAUNCH(64, 4, B3W_COMMIT_WINDOW_LARGE, B3wCurve9Vesta, c9v)
    else if (tpw == 64) B3W_COMMIT_LAUNCH(64, 4, B3W_COMMIT_WINDOW_LARGE, B3wCurve9, c9)
    else B3W_COMMIT_LAUNCH(32, 2, B3W_COMMIT_WINDOW_LARGE, B3wCurve9, c9)
  } else {
    if (vesta) B3W_COMMIT_LAUNCH(64, 4, B3W_COMMIT_WINDOW_SMALL, B3wCurve9Vesta, c9v)
    else if (tpw == 64) B3W_COMMIT_LAUNCH(64, 4, B3W_COMMIT_WINDOW_SMALL, B3wCurve9, c9)
    else B3W_COMMIT_LAUNCH(32, 2, B3W_COMMIT_WINDOW_SMALL, B3wCurve9, c9)
  }
#undef B3W_COMMIT_LAUNCH
  if (!d_out) return (int)hipGetLastError();                  // (the sums are normalised later: b3w_launch_commit_normalize)
  if (vesta) hipLaunchKernelGGL(b3w_commit_normalize_kernel<B3wCurve9Vesta>, dim3((n + 63) / 64), dim3(64), 0, stream, d_sums, n, d_out, *curve, c9v);
  else hipLaunchKernelGGL(b3w_commit_normalize_kernel<B3wCurve9>, dim3((n + 63) / 64), dim3(64), 0, stream, d_sums, n, d_out, *curve, c9);
  return (int)hipGetLastError();
}

extern "C" int b3w_launch_commit_normalize(const uint32_t *d_sums, uint64_t n, uint8_t *d_out, const B3wCurve *curve, hipStream_t stream) {
  if (!n) return 0;
  const B3wCurve9 c9 = make_curve9(*curve);
  const uint64_t threads = (n + B3W_NORMALIZE_K - 1) / B3W_NORMALIZE_K;
  hipLaunchKernelGGL(b3w_commit_normalize_many_kernel<B3wCurve9>, dim3((uint32_t)((threads + 63) / 64)), dim3(64), 0, stream, d_sums, n, d_out, *curve, c9);
  return (int)hipGetLastError();
}

extern "C" int b3w_launch_commit_invtab(const uint32_t *d_gens, const uint32_t *d_inv_slot, const uint32_t *d_inverses, uint32_t njobs, uint32_t nk,
                                        uint32_t *d_invtab, const B3wCurve *curve, hipStream_t stream) {
  if (!njobs || !nk) return 0;
  const uint64_t total = (uint64_t)njobs * nk;
  hipLaunchKernelGGL(b3w_commit_invtab_kernel, dim3((uint32_t)((total + 63) / 64)), dim3(64), 0, stream, d_gens, d_inv_slot, d_inverses, njobs, nk, d_invtab,
                     *curve);
  return (int)hipGetLastError();
}
