// fpmul_peak.hip — how many 256-bit Montgomery multiplications per second does the chip do with this code?
// (the ALU ceiling of csrc/b3w_commit.hip)   build: hipcc --offload-arch=gfx950 -O3 -I../../hot-proofs-blake3-circom_amd/csrc -o fpmul_peak fpmul_peak.hip
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <time.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
struct B3wCurve { uint32_t p[8], r2[8], one[8], pm2[8]; uint32_t inv; };
struct Fp { uint32_t l[8]; };
__device__ __forceinline__ Fp fp_reduce_once(const Fp &a, uint32_t hi, const B3wCurve &C) {
  Fp d; uint32_t br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const uint64_t t = (uint64_t)a.l[i] - C.p[i] - br; d.l[i] = (uint32_t)t; br = (uint32_t)(t >> 63); }
  const bool ge = hi != 0 || br == 0;
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = ge ? d.l[i] : a.l[i];
  return r;
}
__device__ __forceinline__ Fp fp_mul(const Fp &a, const Fp &b, const B3wCurve &C) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { c += (uint64_t)a.l[j] * b.l[i] + t[j]; t[j] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[8] = (uint32_t)c; t[9] = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * C.inv;
    c = (uint64_t)m * C.p[0] + t[0]; c >>= 32;
#pragma unroll
    for (int j = 1; j < 8; ++j) { c += (uint64_t)m * C.p[j] + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[7] = (uint32_t)c; t[8] = t[9] + (uint32_t)(c >> 32);
  }
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = t[i];
  return fp_reduce_once(r, t[8], C);
}

// ---- the same field in nine 29-bit limbs: every column sum fits a 64-bit accumulator, so a limb product is ONE
// v_mad_u64_u32 with the accumulator as its addend (no carry moves).  Montgomery radix 2^261; limbs of the inputs may be
// lazy (< 2^30), the result is < 2p with normalised limbs.
struct F9 { uint32_t l[9]; };
struct Curve9 { uint32_t p[9]; uint32_t inv; };
#define M29 0x1FFFFFFFu
__device__ __forceinline__ F9 mul29(const F9 &a, const F9 &b, const Curve9 &C) {
  uint64_t acc = 0;
  uint32_t m[9];
  F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * C.p[k - i];
    m[k] = ((uint32_t)acc * C.inv) & M29;
    acc += (uint64_t)m[k] * C.p[0];
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
    for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * C.p[k - i];
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
// the same with the multiply-adds pinned (inline asm): no reassociation into partial sums by the compiler
__device__ __forceinline__ void madacc(uint64_t &acc, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ F9 mul29asm(const F9 &a, const F9 &b, const Curve9 &C) {
  uint64_t acc = 0;
  uint32_t m[9];
  F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) madacc(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; ++i) madacc(acc, m[i], C.p[k - i]);
    m[k] = ((uint32_t)acc * C.inv) & M29;
    madacc(acc, m[k], C.p[0]);
    acc >>= 29;
  }
#pragma unroll
  for (int k = 9; k < 17; ++k) {
#pragma unroll
    for (int i = k - 8; i < 9; ++i) madacc(acc, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 8; i < 9; ++i) madacc(acc, m[i], C.p[k - i]);
    r.l[k - 9] = (uint32_t)acc & M29;
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
__device__ __forceinline__ F9 sqr29(const F9 &a, const Curve9 &C) {
  uint64_t acc = 0;
  uint32_t m[9], d[9];
  F9 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) d[i] = a.l[i] << 1;          // lazy limbs < 2^30 -> < 2^31
#pragma unroll
  for (int k = 0; k < 17; ++k) {
#pragma unroll
    for (int i = (k > 8 ? k - 8 : 0); 2 * i < k; ++i) acc += (uint64_t)d[i] * a.l[k - i];
    if (!(k & 1)) acc += (uint64_t)a.l[k / 2] * a.l[k / 2];
    if (k < 9) {
#pragma unroll
      for (int i = 0; i < k; ++i) acc += (uint64_t)m[i] * C.p[k - i];
      m[k] = ((uint32_t)acc * C.inv) & M29;
      acc += (uint64_t)m[k] * C.p[0];
    } else {
#pragma unroll
      for (int i = k - 8; i < 9; ++i) acc += (uint64_t)m[i] * C.p[k - i];
      r.l[k - 9] = (uint32_t)acc & M29;
    }
    acc >>= 29;
  }
  r.l[8] = (uint32_t)acc;
  return r;
}
__device__ __forceinline__ F9 to9(const Fp &a) {
  F9 r;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int bit = 29 * k, w = bit >> 5, s = bit & 31;
    uint64_t v = a.l[w];
    if (w + 1 < 8) v |= (uint64_t)a.l[w + 1] << 32;
    r.l[k] = (uint32_t)(v >> s) & M29;
  }
  return r;
}
__device__ __forceinline__ Fp from9(const F9 &a, uint32_t &hi) {        // limbs normalised (a.l[8] may be wide)
  uint32_t t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int bit = 29 * k, w = bit >> 5, s = bit & 31;
    const uint64_t v = (uint64_t)a.l[k] << s;
    t[w] |= (uint32_t)v;
    if (w + 1 < 9) t[w + 1] |= (uint32_t)(v >> 32);
  }
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = t[i];
  hi = t[8];
  return r;
}
__device__ __forceinline__ Fp fp_add(const Fp &a, const Fp &b, const B3wCurve &C) {
  Fp s; uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const uint64_t t = (uint64_t)a.l[i] + b.l[i] + c; s.l[i] = (uint32_t)t; c = (uint32_t)(t >> 32); }
  return fp_reduce_once(s, c, C);
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k32(uint32_t *out, uint32_t iters, B3wCurve C) {
  Fp a[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) a[c].l[i] = threadIdx.x * 977u + i * 131u + c * 7u + blockIdx.x;
  for (uint32_t it = 0; it < iters; ++it)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = fp_mul(a[c], a[(c + 1) % CHAINS], C);
  uint32_t x = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) x ^= a[c].l[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k29asm(uint32_t *out, uint32_t iters, Curve9 C) {
  F9 a[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 9; ++i) a[c].l[i] = (threadIdx.x * 977u + i * 131u + c * 7u + blockIdx.x) & M29;
  for (uint32_t it = 0; it < iters; ++it)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = mul29asm(a[c], a[(c + 1) % CHAINS], C);
  uint32_t x = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 9; ++i) x ^= a[c].l[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int CHAINS, bool SQ>
__global__ __launch_bounds__(256) void k29(uint32_t *out, uint32_t iters, Curve9 C) {
  F9 a[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 9; ++i) a[c].l[i] = (threadIdx.x * 977u + i * 131u + c * 7u + blockIdx.x) & M29;
  for (uint32_t it = 0; it < iters; ++it)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = SQ ? sqr29(a[c], C) : mul29(a[c], a[(c + 1) % CHAINS], C);
  uint32_t x = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 9; ++i) x ^= a[c].l[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
// 32 * mul29(a, b) == fp_mul(a, b)  (radix 2^261 against 2^256), and sqr29(a) == mul29(a, a)
__global__ void kcheck(uint32_t *out, B3wCurve C, Curve9 C9) {
  Fp a, b;
  uint32_t x = threadIdx.x * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i) { x = x * 1664525u + 1013904223u; a.l[i] = x; x = x * 1664525u + 1013904223u; b.l[i] = x; }
  a.l[7] &= 0x0FFFFFFFu; b.l[7] &= 0x0FFFFFFFu;           // < 2^252 < p
  if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) { a.l[i] = C.p[i]; }   // edge: p - 1 and all-ones limbs
  if (threadIdx.x == 0) a.l[0] -= 1;
  const Fp want = fp_mul(a, b, C);
  const F9 g9 = mul29(to9(a), to9(b), C9);
  uint32_t hi;
  Fp g = from9(g9, hi);
  g = fp_reduce_once(g, hi, C);
  for (int i = 0; i < 5; ++i) g = fp_add(g, g, C);
  bool ok = true;
  for (int i = 0; i < 8; ++i) ok &= g.l[i] == want.l[i];
  const F9 s9 = sqr29(to9(a), C9), m9 = mul29(to9(a), to9(a), C9);
  for (int i = 0; i < 9; ++i) ok &= s9.l[i] == m9.l[i];
  out[threadIdx.x] = ok;
}
int main(int argc, char **argv) {
  B3wCurve C{};
  const uint64_t q[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  memcpy(C.p, q, 32); C.inv = 0xe4866389u;   // -q^-1 mod 2^32 of BN254 q
  Curve9 C9{};
  for (int k = 0; k < 9; ++k) {
    const int bit = 29 * k, w = bit >> 5, s = bit & 31;
    uint64_t v = C.p[w]; if (w + 1 < 8) v |= (uint64_t)C.p[w + 1] << 32;
    C9.p[k] = (uint32_t)(v >> s) & M29;
  }
  C9.inv = C.inv & M29;
  uint32_t *out; hipMalloc((void **)&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, auto launch, double muls) {
    launch(); hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-36s %8.2f ms  %7.1f G field mul/s\n", name, ms, muls / ms / 1e6);
  };
  const uint32_t iters = 2000, grid = 4096;
  if (argc > 2 && !strcmp(argv[1], "loop")) {
    // ./fpmul29_peak loop <seconds> [chains = 2]: the multiplication loop alone for that long, its rate second by second — the workload
    // of a power measurement (tools/ubench/power_cases.py: is the "VALU ceiling" the commit kernel is priced against the chip's
    // arithmetic or its power limit?)
    const double secs = atof(argv[2]);
    const int chains = argc > 3 ? atoi(argv[3]) : 2;
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
      timespec ta, tb; clock_gettime(CLOCK_MONOTONIC, &ta);
      int n = 0;
      do {
        if (chains == 1) hipLaunchKernelGGL((k29<1, false>), dim3(grid), dim3(256), 0, 0, out, iters, C9);
        else hipLaunchKernelGGL((k29<2, false>), dim3(grid), dim3(256), 0, 0, out, iters, C9);
        hipDeviceSynchronize(); n++;
        clock_gettime(CLOCK_MONOTONIC, &tb);
      } while ((tb.tv_sec - ta.tv_sec) + (tb.tv_nsec - ta.tv_nsec) * 1e-9 < 1.0);
      const double dt = (tb.tv_sec - ta.tv_sec) + (tb.tv_nsec - ta.tv_nsec) * 1e-9;
      printf("%d chain(s): %.1f G field mul/s\n", chains, (double)grid * 256 * iters * chains * n / dt / 1e9);
      fflush(stdout);
      if ((tb.tv_sec - t0.tv_sec) + (tb.tv_nsec - t0.tv_nsec) * 1e-9 >= secs) break;
    }
    return 0;
  }
  run("8 x 32-bit CIOS, 2 chains", [&] { hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(256), 0, 0, out, iters, C); }, (double)grid * 256 * iters * 2);
  run("9 x 29-bit mul, 1 chain", [&] { hipLaunchKernelGGL((k29<1, false>), dim3(grid), dim3(256), 0, 0, out, iters, C9); }, (double)grid * 256 * iters);
  run("9 x 29-bit mul, 2 chains", [&] { hipLaunchKernelGGL((k29<2, false>), dim3(grid), dim3(256), 0, 0, out, iters, C9); }, (double)grid * 256 * iters * 2);
  run("9 x 29-bit mul, pinned mads, 2 chains", [&] { hipLaunchKernelGGL((k29asm<2>), dim3(grid), dim3(256), 0, 0, out, iters, C9); }, (double)grid * 256 * iters * 2);
  {
    uint32_t h1[256], h2[256];
    hipLaunchKernelGGL((k29<2, false>), dim3(1), dim3(256), 0, 0, out, 50u, C9); hipMemcpy(h1, out, 1024, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL((k29asm<2>), dim3(1), dim3(256), 0, 0, out, 50u, C9); hipMemcpy(h2, out, 1024, hipMemcpyDeviceToHost);
    int same = 1; for (int i = 0; i < 256; i++) same &= h1[i] == h2[i];
    printf("pinned == compiler-scheduled over 50 iterations: %s\n", same ? "yes" : "NO");
  }
  run("9 x 29-bit sqr, 2 chains", [&] { hipLaunchKernelGGL((k29<2, true>), dim3(grid), dim3(256), 0, 0, out, iters, C9); }, (double)grid * 256 * iters * 2);
  hipLaunchKernelGGL(kcheck, dim3(1), dim3(256), 0, 0, out, C, C9);
  uint32_t h[256]; hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  int ok = 1; for (int i = 0; i < 256; i++) ok &= h[i] == 1;
  printf("32 * mul29(a, b) == fp_mul(a, b) and sqr29 == mul29(a, a) on 256 lanes: %s\n", ok ? "yes" : "NO");
  return 0;
}
