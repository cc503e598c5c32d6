// placement_map2.hip — fine map of the body-per-wave store rate over one large allocation.
// build: hipcc --offload-arch=gfx950 -O3 -o placement_map2 placement_map2.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int W>
__global__ __launch_bounds__(64) void k_fused(uint8_t *out, uint64_t pitch, uint32_t full, uint32_t n, const uint32_t *jit) {
  const uint32_t wit0 = blockIdx.x * W, lane = threadIdx.x;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  uint64_t base[W];
#pragma unroll
  for (int w = 0; w < W; ++w) base[w] = (uint64_t)(wit0 + w) * pitch + (jit ? (uint64_t)jit[wit0 + w] * 32 : 0) + lane * 16;
  for (uint32_t g = 0; g < full; ++g)
#pragma unroll
    for (int w = 0; w < W; ++w) *reinterpret_cast<u32x4 *>(out + base[w] + (uint64_t)g * 1024) = v;
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
  const uint32_t nwit = 24093;
  const uint64_t body = 32ull * nwit;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t GiB = 1ull << 30, big = 72 * GiB;
  uint8_t *p;
  CK(hipMalloc((void **)&p, big));
  printf("base va %p\n", (void *)p);
  // jitter table: random extra offset per body (0..8191 slots) to break the arithmetic progression
  uint32_t *hj = (uint32_t *)malloc(4096 * 4), *dj;
  srand(1);
  for (int i = 0; i < 4096; i++) hj[i] = rand() % 8192;
  CK(hipMalloc((void **)&dj, 4096 * 4));
  CK(hipMemcpy(dj, hj, 4096 * 4, hipMemcpyHostToDevice));
  // A: n=4096 windows (3.16 GB) every 1 GiB
  printf("# A: 4096 bodies W=4, window start every 1 GiB: rate GB/s (plain / jittered pitch+256KiB)\n");
  for (uint64_t off = 0; off + 5 * GiB <= big; off += GiB) {
    const uint32_t n = 4096;
    const double a = timeit([&] { hipLaunchKernelGGL((k_fused<4>), dim3(n / 4), dim3(64), 0, 0, p + off, body, nwit >> 5, n, (const uint32_t *)nullptr); });
    const double j = timeit([&] { hipLaunchKernelGGL((k_fused<4>), dim3(n / 4), dim3(64), 0, 0, p + off, body + 262144, nwit >> 5, n, (const uint32_t *)dj); });
    printf("A off=%3llu GiB  %5.0f  %5.0f\n", (unsigned long long)(off / GiB), n * (double)body / 1e6 / a, n * (double)body / 1e6 / j);
    fflush(stdout);
  }
  // B: n=1024 windows (790 MB) every 256 MiB over the first 24 GiB
  printf("# B: 1024 bodies W=1, window start every 256 MiB\n");
  for (uint64_t off = 0; off + 2 * GiB <= 24 * GiB; off += GiB / 4) {
    const uint32_t n = 1024;
    const double a = timeit([&] { hipLaunchKernelGGL((k_fused<1>), dim3(n), dim3(64), 0, 0, p + off, body, nwit >> 5, n, (const uint32_t *)nullptr); });
    printf("B off=%6.2f GiB  %5.0f\n", off / (double)GiB, n * (double)body / 1e6 / a);
    fflush(stdout);
  }
  return 0;
}
