// fpmul_peak.hip — how many 256-bit Montgomery multiplications per second does the chip do with this code?
// (the ALU ceiling of csrc/b3w_commit.hip)   build: hipcc --offload-arch=gfx950 -O3 -I../../hot-proofs-blake3-circom_amd/csrc -o fpmul_peak fpmul_peak.hip
#include <hip/hip_runtime.h>
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
// dedicated squaring: 36 limb products for a*a (cross terms doubled) + the same Montgomery reduction (SOS: product then reduce)
__device__ __forceinline__ Fp fp_sqr(const Fp &a, const B3wCurve &C) {
  uint32_t t[17];
#pragma unroll
  for (int i = 0; i < 17; ++i) t[i] = 0;
  // cross products a[i]*a[j], i < j
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
#pragma unroll
    for (int j = i + 1; j < 8; ++j) { c += (uint64_t)a.l[i] * a.l[j] + t[i + j]; t[i + j] = (uint32_t)c; c >>= 32; }
    t[i + 8] = (uint32_t)c;
  }
  // double, add the squares
  uint32_t top = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { const uint32_t n = (t[i] << 1) | top; top = t[i] >> 31; t[i] = n; }
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a.l[i] * a.l[i] + t[2 * i];
    t[2 * i] = (uint32_t)c; c >>= 32;
    c += t[2 * i + 1];
    t[2 * i + 1] = (uint32_t)c; c >>= 32;
  }
  // Montgomery reduction of the 512-bit product
  uint32_t carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint32_t m = t[i] * C.inv;
    uint64_t d = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { d += (uint64_t)m * C.p[j] + t[i + j]; t[i + j] = (uint32_t)d; d >>= 32; }
    d += (uint64_t)t[i + 8] + carry;
    t[i + 8] = (uint32_t)d;
    carry = (uint32_t)(d >> 32);
  }
  Fp r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.l[i] = t[i + 8];
  return fp_reduce_once(r, carry, C);
}
template <int CHAINS>
__global__ __launch_bounds__(256) void ksq(uint32_t *out, uint32_t iters, B3wCurve C) {
  Fp a[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) a[c].l[i] = threadIdx.x * 977u + i * 131u + c * 7u + blockIdx.x;
  for (uint32_t it = 0; it < iters; ++it)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = fp_sqr(a[c], C);
  uint32_t x = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) x ^= a[c].l[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void kmulsq(uint32_t *out, uint32_t iters, B3wCurve C) {     // squaring through fp_mul, for comparison
  Fp a[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) a[c].l[i] = threadIdx.x * 977u + i * 131u + c * 7u + blockIdx.x;
  for (uint32_t it = 0; it < iters; ++it)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = fp_mul(a[c], a[c], C);
  uint32_t x = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) x ^= a[c].l[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t iters, B3wCurve C) {
  Fp a[CHAINS], b;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) a[c].l[i] = threadIdx.x * 977u + i * 131u + c * 7u + blockIdx.x;
  for (int i = 0; i < 8; ++i) b.l[i] = 0x12345u * (i + 1) + threadIdx.x;
  for (uint32_t it = 0; it < iters; ++it)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = fp_mul(a[c], b, C);
  uint32_t x = 0;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 8; ++i) x ^= a[c].l[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
int main() {
  B3wCurve C{};
  const uint64_t q[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  memcpy(C.p, q, 32); C.inv = 0xe4866389u;   // -q^-1 mod 2^32 of BN254 q
  uint32_t *out; hipMalloc((void **)&out, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char *name, auto launch, double muls) {
    launch(); hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.2f ms  %7.1f G field mul/s\n", name, ms, muls / ms / 1e6);
  };
  const uint32_t iters = 2000, grid = 4096;
  run("1 chain/lane, 256 thr", [&] { hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters, C); }, (double)grid * 256 * iters);
  run("2 chains/lane", [&] { hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, iters, C); }, (double)grid * 256 * iters * 2);
  run("4 chains/lane", [&] { hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, iters, C); }, (double)grid * 256 * iters * 4);
  run("squaring via fp_mul, 2 chains", [&] { hipLaunchKernelGGL(kmulsq<2>, dim3(grid), dim3(256), 0, 0, out, iters, C); }, (double)grid * 256 * iters * 2);
  run("dedicated fp_sqr, 2 chains", [&] { hipLaunchKernelGGL(ksq<2>, dim3(grid), dim3(256), 0, 0, out, iters, C); }, (double)grid * 256 * iters * 2);
  // same values?
  uint32_t h1[256], h2[256];
  hipLaunchKernelGGL(kmulsq<1>, dim3(1), dim3(256), 0, 0, out, 50u, C); hipMemcpy(h1, out, 1024, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(ksq<1>, dim3(1), dim3(256), 0, 0, out, 50u, C); hipMemcpy(h2, out, 1024, hipMemcpyDeviceToHost);
  int same = 1; for (int i = 0; i < 256; i++) same &= h1[i] == h2[i];
  printf("fp_sqr == fp_mul(a, a) over 50 iterations on 256 lanes: %s\n", same ? "yes" : "NO");
  return 0;
}
