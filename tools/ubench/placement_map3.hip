// placement_map3.hip — do body streams spread over several 32 GiB regions write faster?
//   T1  locate a region boundary: plain 4096-body windows, start every 0.5 GiB
//   T2  R-way split: body i lives at lo + (i % R) * 32 GiB + (i / R) * pitch, for arbitrary lo
//   T3  a 25 GB batch (32768 bodies) centred on the boundary: blocks in body order vs alternating halves
// build: hipcc --offload-arch=gfx950 -O3 -o placement_map3 placement_map3.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// body i at out + (i % R) * rstride + (i / R) * pitch;  HALVES: block j even -> bodies from the lower half, odd -> upper half
template <int W, bool HALVES>
__global__ __launch_bounds__(64) void k_fused(uint8_t *out, uint64_t pitch, uint32_t full, uint32_t n, uint32_t R, uint64_t rstride) {
  uint32_t blk = blockIdx.x;
  if (HALVES) blk = (blk >> 1) + (blk & 1) * (gridDim.x >> 1);
  const uint32_t wit0 = blk * W, lane = threadIdx.x;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  uint64_t base[W];
#pragma unroll
  for (int w = 0; w < W; ++w) { const uint32_t i = wit0 + w; base[w] = (uint64_t)(i % R) * rstride + (uint64_t)(i / R) * pitch + lane * 16; }
  for (uint32_t g = 0; g < full; ++g)
#pragma unroll
    for (int w = 0; w < W; ++w) *reinterpret_cast<u32x4 *>(out + base[w] + (uint64_t)g * 1024) = v;
}
__global__ __launch_bounds__(256) void k_fill(u32x4 *out, uint64_t n16) {
  u32x4 v = {1, 0, 0, 0};
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) out[i] = v;
}
static hipEvent_t e0, e1;
template <class F>
static double timeit(F launch, int it = 10) {
  for (int i = 0; i < 2; i++) launch();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < it; i++) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / it;
}
int main() {
  const uint32_t nwit = 24093, full = nwit >> 5;
  const uint64_t body = 32ull * nwit, GiB = 1ull << 30;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t big = 136 * GiB;
  uint8_t *p;
  CK(hipMalloc((void **)&p, big));
  printf("base va %p\n", (void *)p);
  const uint32_t n = 4096;
  auto rate = [&](double ms, uint32_t nn) { return nn * (double)body / 1e6 / ms; };
  // T1
  double best = 0; uint64_t first_good = 0, last_good = 0; bool seen = false;
  for (uint64_t off = 0; off <= 40 * GiB; off += GiB / 2) {
    const double a = timeit([&] { hipLaunchKernelGGL((k_fused<4, false>), dim3(n / 4), dim3(64), 0, 0, p + off, body, full, n, 1u, 0ull); }, 6);
    const double r = rate(a, n);
    if (r > 6200) { if (!seen) { first_good = off; seen = true; } if (off - first_good < 8 * GiB) last_good = off; }
    if (r > best) best = r;
    printf("T1 off=%5.1f GiB %5.0f\n", off / (double)GiB, r);
  }
  const uint64_t centre = seen ? (first_good + last_good) / 2 + (uint64_t)(n * body / 2) : 32 * GiB;   // estimated boundary
  printf("T1 boundary estimate: %.2f GiB (good starts %.1f..%.1f)\n", centre / (double)GiB, first_good / (double)GiB, last_good / (double)GiB);
  fflush(stdout);
  // T2
  for (uint32_t R = 1; R <= 4; ++R)
    for (uint64_t lo = 0; lo <= 24 * GiB; lo += 6 * GiB) {
      const double a = timeit([&] { hipLaunchKernelGGL((k_fused<4, false>), dim3(n / 4), dim3(64), 0, 0, p + lo, body, full, n, R, 32 * GiB); });
      printf("T2 R=%u lo=%2llu GiB  %5.0f\n", R, (unsigned long long)(lo / GiB), rate(a, n));
    }
  // also other region strides for R=2: is 32 GiB special?
  for (uint64_t rs : {4 * GiB, 8 * GiB, 16 * GiB, 24 * GiB, 32 * GiB, 40 * GiB, 48 * GiB, 64 * GiB}) {
    const double a = timeit([&] { hipLaunchKernelGGL((k_fused<4, false>), dim3(n / 4), dim3(64), 0, 0, p + 3 * GiB, body, full, n, 2u, rs); });
    printf("T2b R=2 lo=3 GiB stride=%2llu GiB  %5.0f\n", (unsigned long long)(rs / GiB), rate(a, n));
  }
  fflush(stdout);
  // T3: 32768 bodies centred on the boundary
  {
    const uint32_t nb = 32768;
    const uint64_t span = nb * body;
    if (centre > span / 2 && centre + span / 2 < big) {
      uint8_t *q = p + ((centre - span / 2) & ~4095ull);
      const double a = timeit([&] { hipLaunchKernelGGL((k_fused<4, false>), dim3(nb / 4), dim3(64), 0, 0, q, body, full, nb, 1u, 0ull); }, 4);
      const double h = timeit([&] { hipLaunchKernelGGL((k_fused<4, true>), dim3(nb / 4), dim3(64), 0, 0, q, body, full, nb, 1u, 0ull); }, 4);
      const double f = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, (u32x4 *)q, span / 16); }, 4);
      printf("T3 32768 bodies centred: in order %5.0f   alternating halves %5.0f   fill %5.0f\n", rate(a, nb), rate(h, nb), rate(f, nb));
      uint8_t *q0 = p + 2 * GiB;   // not centred
      const double a0 = timeit([&] { hipLaunchKernelGGL((k_fused<4, false>), dim3(nb / 4), dim3(64), 0, 0, q0, body, full, nb, 1u, 0ull); }, 4);
      const double h0 = timeit([&] { hipLaunchKernelGGL((k_fused<4, true>), dim3(nb / 4), dim3(64), 0, 0, q0, body, full, nb, 1u, 0ull); }, 4);
      printf("T3 32768 bodies at +2 GiB:  in order %5.0f   alternating halves %5.0f\n", rate(a0, nb), rate(h0, nb));
    }
  }
  return 0;
}
