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
//            4 095 non-empty subset sums are tabulated (b3w_commit_window_kernel: 4.5 k windows x 4 095 affine points = 1.2 GB);
//   commit   32 lanes per witness (two witnesses per wave); lane t owns windows t, t + 32, ...; it gathers the W bits of
//            a window from the body, skips ahead to its next NON-ZERO window, and then the whole wave does one mixed
//            Jacobian + affine addition with the tabulated point — no doublings, one addition per W slots, no
//            zero work in lock step; an LDS tree adds the 32 partial sums; a second kernel normalises them, one
//            thread per witness (Fermat inversion), and stores the affine points.
// Arithmetic: 256-bit Montgomery (CIOS, eight 32-bit limbs, modulus passed at run time), complete handling of the
// exceptional cases (infinity, P + P, P - P) so that related generators cannot break it.
// Domain: bodies of the batch kernels (every bit slot holds 0 or 1, words fit their slot's width); a slot holding
// anything else (e.g. a body of the exact kernel with a 254-bit input) flags the witness (status 103) instead of
// producing a wrong commitment.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "b3w_kernels.h"

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

// ---- set-up: points[first_v[s] + k] = 2^k * G_s (Montgomery affine) for every committed slot s
__global__ __launch_bounds__(64) void b3w_commit_setup_kernel(const uint32_t *__restrict__ gens /* nslots x 16 words, standard form */,
                                                              const uint32_t *__restrict__ first_v, const uint32_t *__restrict__ nbits,
                                                              uint32_t nslots, uint32_t *__restrict__ points, B3wCurve C) {
  const uint32_t s = blockIdx.x * 64 + threadIdx.x;
  if (s >= nslots) return;
  Fp r2;
#pragma unroll
  for (int i = 0; i < 8; ++i) r2.l[i] = C.r2[i];
  Jac P;
  P.X = fp_mul(load_fp(gens + (uint64_t)s * 16), r2, C);
  P.Y = fp_mul(load_fp(gens + (uint64_t)s * 16 + 8), r2, C);
#pragma unroll
  for (int i = 0; i < 8; ++i) P.Z.l[i] = C.one[i];
  const uint32_t nb = nbits[s], v0 = first_v[s];
  for (uint32_t k = 0; k < nb; ++k) {
    Fp x, y;
    if (k == 0) { x = P.X; y = P.Y; } else jac_to_affine(P, x, y, C);
    store_fp(points + (uint64_t)(v0 + k) * 16, x);
    store_fp(points + (uint64_t)(v0 + k) * 16 + 8, y);
    if (k + 1 < nb) P = jac_dbl(P, C);
  }
}

// ---- set-up 2: table[win * (2^W - 1) + m - 1] = sum of the window's virtual-slot points selected by the bits of m
__global__ __launch_bounds__(64) void b3w_commit_window_kernel(const uint32_t *__restrict__ points, uint32_t nwin,
                                                               uint32_t *__restrict__ table, B3wCurve C) {
  const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= (uint64_t)nwin * B3W_COMMIT_ENTRIES) return;
  const uint32_t win = (uint32_t)(i / B3W_COMMIT_ENTRIES), m = (uint32_t)(i % B3W_COMMIT_ENTRIES) + 1;
  Jac acc = jac_infinity();
#pragma unroll 1
  for (int j = 0; j < B3W_COMMIT_WINDOW; ++j) {
    if (!((m >> j) & 1)) continue;
    const uint32_t *pt = points + (uint64_t)(win * B3W_COMMIT_WINDOW + j) * 16;
    acc = jac_madd(acc, load_fp(pt), load_fp(pt + 8), C);
  }
  Fp x, y;
  jac_to_affine(acc, x, y, C);                            // infinity -> (0, 0)
  store_fp(table + i * 16, x);
  store_fp(table + i * 16 + 8, y);
}

// ---- commit: one workgroup per witness
template <int T, int WPB>        // threads per witness, witnesses per workgroup (T * WPB threads)
__global__ __launch_bounds__(T * WPB) void b3w_commit_kernel(const uint8_t *__restrict__ bodies, uint32_t n, uint64_t pitch,
                                                         const uint32_t *__restrict__ vslots /* B3W_COMMIT_WINDOW per window: slot | bit << 19 | single << 27 | width words << 28 */,
                                                         const uint32_t *__restrict__ table, uint32_t nwin, uint32_t *__restrict__ sums /* n x 24 words: X Y Z */,
                                                         int32_t *__restrict__ status, B3wCurve C) {
  __shared__ __attribute__((aligned(16))) uint32_t red[T * WPB * 24];
  __shared__ uint32_t bad[WPB];
  const uint32_t sub = threadIdx.x / T, t = threadIdx.x % T;       // which witness of the workgroup, lane within it
  const uint32_t w = blockIdx.x * WPB + sub;
  const bool live = w < n;
  if (t == 0) bad[sub] = 0;
  __syncthreads();
  const uint32_t *body = reinterpret_cast<const uint32_t *>(bodies + (uint64_t)(live ? w : 0) * pitch);
  Jac acc = jac_infinity();
  uint32_t win = live ? t : nwin;
  while (true) {
    // skip ahead to this lane's next window with a set bit (four virtual slots at a time: short live ranges)
    uint32_t m = 0;
    while (win < nwin) {
#pragma unroll
      for (int j = 0; j < B3W_COMMIT_WINDOW; j += 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(vslots + (uint64_t)win * B3W_COMMIT_WINDOW + j);
        const uint32_t e[4] = {q.x, q.y, q.z, q.w};
        uint32_t word[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) word[i] = body[(uint64_t)(e[i] & 0x7FFFFu) * 8 + (((e[i] >> 19) & 0xFFu) >> 5)];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (((e[i] >> 27) & 1u) && word[i] > 1) bad[sub] = 1;   // a bit slot must hold 0 or 1
          const uint32_t words = e[i] >> 28;                 // on the first virtual slot of a 32- or 64-bit slot: its width in words
          if (words) {                                       // (rare: 1.4 % of the virtual slots) the rest of the element must be 0
            const uint32_t *el = body + (uint64_t)(e[i] & 0x7FFFFu) * 8;
            uint32_t rest = 0;
            for (uint32_t k = words; k < 8; ++k) rest |= el[k];
            if (rest) bad[sub] = 1;
          }
          m |= ((word[i] >> ((e[i] >> 19) & 31u)) & 1u) << (j + i);
        }
      }
      if (m) break;
      win += T;
    }
    const bool has = win < nwin;
    if (!__any(has)) break;
    if (has) {
      const uint32_t *pt = table + ((uint64_t)win * B3W_COMMIT_ENTRIES + m - 1) * 16;
      acc = jac_madd(acc, load_fp(pt), load_fp(pt + 8), C);
      win += T;
    }
  }
  // LDS tree over each witness's T partial sums
  uint32_t *mine = red + (sub * T + t) * 24;
  store_fp(mine, acc.X); store_fp(mine + 8, acc.Y); store_fp(mine + 16, acc.Z);
  __syncthreads();
  for (uint32_t st = T / 2; st >= 1; st >>= 1) {
    if (t < st) {
      Jac a, b;
      const uint32_t *other = mine + st * 24;
      a.X = load_fp(mine); a.Y = load_fp(mine + 8); a.Z = load_fp(mine + 16);
      b.X = load_fp(other); b.Y = load_fp(other + 8); b.Z = load_fp(other + 16);
      a = jac_add(a, b, C);
      store_fp(mine, a.X); store_fp(mine + 8, a.Y); store_fp(mine + 16, a.Z);
    }
    __syncthreads();
  }
  if (live && t < 24) sums[(uint64_t)w * 24 + t] = red[sub * T * 24 + t];   // the Jacobian sum; normalised by the next kernel
  if (live && t == 0 && status) status[w] = bad[sub] ? 103 : 0;
}

// ---- normalise: one THREAD per witness (a Fermat inversion is 380 dependent multiplications: on thread 0 of the commit
// workgroup it took longer than the workgroup's whole share of additions)
__global__ __launch_bounds__(64) void b3w_commit_normalize_kernel(const uint32_t *__restrict__ sums, uint32_t n, uint8_t *__restrict__ out, B3wCurve C) {
  const uint32_t w = blockIdx.x * 64 + threadIdx.x;
  if (w >= n) return;
  Jac R;
  R.X = load_fp(sums + (uint64_t)w * 24); R.Y = load_fp(sums + (uint64_t)w * 24 + 8); R.Z = load_fp(sums + (uint64_t)w * 24 + 16);
  Fp x, y, one_std;
  jac_to_affine(R, x, y, C);
#pragma unroll
  for (int i = 0; i < 8; ++i) one_std.l[i] = i == 0 ? 1u : 0u;
  x = fp_mul(x, one_std, C);                                 // out of Montgomery form
  y = fp_mul(y, one_std, C);
  uint32_t *o = reinterpret_cast<uint32_t *>(out + (uint64_t)w * 64);
  store_fp(o, x);
  store_fp(o + 8, y);
}

}  // namespace

extern "C" int b3w_launch_commit_setup(const uint32_t *d_gens, const uint32_t *d_first_v, const uint32_t *d_nbits, uint32_t nslots,
                                       uint32_t *d_points, const B3wCurve *curve, hipStream_t stream) {
  if (!nslots) return 0;
  hipLaunchKernelGGL(b3w_commit_setup_kernel, dim3((nslots + 63) / 64), dim3(64), 0, stream, d_gens, d_first_v, d_nbits, nslots, d_points, *curve);
  return (int)hipGetLastError();
}

extern "C" int b3w_launch_commit_windows(const uint32_t *d_points, uint32_t nwin, uint32_t *d_table, const B3wCurve *curve, hipStream_t stream) {
  if (!nwin) return 0;
  const uint64_t total = (uint64_t)nwin * B3W_COMMIT_ENTRIES;
  hipLaunchKernelGGL(b3w_commit_window_kernel, dim3((uint32_t)((total + 63) / 64)), dim3(64), 0, stream, d_points, nwin, d_table, *curve);
  return (int)hipGetLastError();
}

extern "C" int b3w_launch_commit(const uint8_t *d_bodies, uint32_t n, uint64_t pitch, const uint32_t *d_vslots, const uint32_t *d_table,
                                 uint32_t nwin, uint32_t *d_sums /* n * 24 words scratch */, uint8_t *d_out, int32_t *d_status,
                                 const B3wCurve *curve, hipStream_t stream) {
  if (!n) return 0;
  // threads per witness: 32 (two witnesses per wave) for the compression circuit, 64 for the longer nova witnesses (measured)
  static const int env_tpw = getenv("B3W_COMMIT_THREADS") ? atoi(getenv("B3W_COMMIT_THREADS")) : 0;
  const int tpw = env_tpw ? env_tpw : nwin > 4600 ? 64 : 32;      // windows: 4 455 compression, 4 853 nova O2, 8 981 nova O1
#define B3W_COMMIT_LAUNCH(T, WPB)                                                                                         \
  hipLaunchKernelGGL((b3w_commit_kernel<T, WPB>), dim3((n + WPB - 1) / WPB), dim3(T * WPB), 0, stream, d_bodies, n, pitch, d_vslots, d_table, \
                     nwin, d_sums, d_status, *curve)
  if (tpw == 256) B3W_COMMIT_LAUNCH(256, 1);
  else if (tpw == 128) B3W_COMMIT_LAUNCH(128, 1);
  else if (tpw == 64) B3W_COMMIT_LAUNCH(64, 1);
  else if (tpw == 16) B3W_COMMIT_LAUNCH(16, 4);
  else B3W_COMMIT_LAUNCH(32, 2);                       // two witnesses per wave
#undef B3W_COMMIT_LAUNCH
  hipLaunchKernelGGL(b3w_commit_normalize_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, d_sums, n, d_out, *curve);
  return (int)hipGetLastError();
}
