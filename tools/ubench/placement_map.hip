// placement_map.hip — where does the "buffer-dependent" store rate of the body-per-wave pattern come from?
//   E1  one large allocation, 3.16 GB windows at increasing offsets: is the rate positional?
//   E2  separate hipMalloc buffers vs hipMemCreate/hipMemMap (explicit physical chunks)
//   E4  burst order: U consecutive 1 KiB groups of ONE body before moving to the next body of the wave
// Pure stores; build: hipcc --offload-arch=gfx950 -O3 -o placement_map placement_map.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// fused-kernel pattern: wave per block, W bodies per wave; U groups of one body back to back
template <int W, int U>
__global__ __launch_bounds__(64) void k_fused(uint8_t *out, uint64_t pitch, uint32_t full, uint32_t n) {
  const uint32_t wit0 = blockIdx.x * W, lane = threadIdx.x;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  uint32_t g = 0;
  for (; g + U <= full; g += U)
#pragma unroll
    for (int w = 0; w < W; ++w)
#pragma unroll
      for (int u = 0; u < U; ++u)
        *reinterpret_cast<u32x4 *>(out + (uint64_t)(wit0 + w) * pitch + (uint64_t)(g + u) * 1024 + lane * 16) = v;
  for (; g < full; ++g)
#pragma unroll
    for (int w = 0; w < W; ++w) *reinterpret_cast<u32x4 *>(out + (uint64_t)(wit0 + w) * pitch + (uint64_t)g * 1024 + lane * 16) = v;
}
__global__ __launch_bounds__(256) void k_fill(u32x4 *out, uint64_t n16) {
  u32x4 v = {1, 0, 0, 0};
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) out[i] = v;
}

static hipEvent_t e0, e1;
template <class F>
static double timeit(F launch, int it = 20) {
  for (int i = 0; i < 3; i++) launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < it; i++) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / it;
}

int main() {
  const uint32_t n = 4096, nwit = 24093;
  const uint64_t body = 32ull * nwit, bytes = (uint64_t)n * body;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto rate = [&](double ms) { return bytes / 1e9 / (ms * 1e-3); };
  auto probe = [&](const char *tag, uint8_t *buf) {
    const double f1 = timeit([&] { hipLaunchKernelGGL((k_fused<4, 1>), dim3(n / 4), dim3(64), 0, 0, buf, body, nwit >> 5, n); });
    const double f4 = timeit([&] { hipLaunchKernelGGL((k_fused<4, 4>), dim3(n / 4), dim3(64), 0, 0, buf, body, nwit >> 5, n); });
    const double f8 = timeit([&] { hipLaunchKernelGGL((k_fused<4, 8>), dim3(n / 4), dim3(64), 0, 0, buf, body, nwit >> 5, n); });
    const double g4 = timeit([&] { hipLaunchKernelGGL((k_fused<2, 4>), dim3(n / 2), dim3(64), 0, 0, buf, body, nwit >> 5, n); });
    const double h4 = timeit([&] { hipLaunchKernelGGL((k_fused<1, 4>), dim3(n), dim3(64), 0, 0, buf, body, nwit >> 5, n); });
    const double fl = timeit([&] { hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16); });
    printf("%-26s va=%p  W4U1 %5.0f  W4U4 %5.0f  W4U8 %5.0f  W2U4 %5.0f  W1U4 %5.0f  fill %5.0f GB/s\n", tag, (void *)buf, rate(f1),
           rate(f4), rate(f8), rate(g4), rate(h4), rate(fl));
    fflush(stdout);
  };
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
  // E2a: separate buffers
  uint8_t *bufs[6];
  for (int b = 0; b < 6; b++) CK(hipMalloc((void **)&bufs[b], bytes + (1 << 22)));
  for (int b = 0; b < 6; b++) { char t[32]; snprintf(t, sizeof t, "hipMalloc #%d", b); probe(t, bufs[b]); }
  for (int b = 0; b < 6; b++) CK(hipFree(bufs[b]));
  // E1: one big allocation
  {
    const uint64_t win = 3ull << 30, big = 10 * win + (1ull << 30);
    uint8_t *p;
    CK(hipMalloc((void **)&p, big));
    for (int k = 0; k < 10; k++) { char t[32]; snprintf(t, sizeof t, "big+%2d GiB", 3 * k); probe(t, p + k * win); }
    CK(hipFree(p));
  }
  // E2b: virtual memory API, explicit physical chunks
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("vmm granularity min %zu recommended %zu\n", gmin, grec);
    for (int rep = 0; rep < 3; rep++) {
      const size_t g = grec ? grec : gmin, sz = (bytes + (1 << 22) + g - 1) / g * g;
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, sz, &prop, 0));
      void *va = nullptr;
      CK(hipMemAddressReserve(&va, sz, 1ull << 30, nullptr, 0));
      CK(hipMemMap(va, sz, 0, h, 0));
      hipMemAccessDesc acc = {};
      acc.location = prop.location;
      acc.flags = hipMemAccessFlagsProtReadWrite;
      CK(hipMemSetAccess(va, sz, &acc, 1));
      char t[32]; snprintf(t, sizeof t, "vmm #%d", rep);
      probe(t, (uint8_t *)va);
      // keep mapped so the next one lands elsewhere
    }
  }
  return 0;
}
