// mixed_alloc.hip — prototype: a contiguous VA buffer whose 128 MiB pieces alternate between two HBM ranks.
// HIP virtual-memory API: physical handles are created in sequence (with unmapped spacers) until a group is found
// whose split-store probe against the first group is fast (= other rank); the two groups are then mapped
// interleaved into one VA range and everything else is released.
// build: hipcc --offload-arch=gfx950 -O3 -o mixed_alloc mixed_alloc.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <chrono>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int W>
__global__ __launch_bounds__(64) void k_split(uint8_t *lo, uint8_t *hi, uint64_t pitch, uint32_t full, uint32_t n) {
  const uint32_t wit0 = blockIdx.x * W, lane = threadIdx.x;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  uint8_t *base[W];
#pragma unroll
  for (int w = 0; w < W; ++w) { const uint32_t i = wit0 + w; base[w] = ((i & 1) ? hi : lo) + (uint64_t)(i >> 1) * pitch + lane * 16; }
  for (uint32_t g = 0; g < full; ++g)
#pragma unroll
    for (int w = 0; w < W; ++w) *reinterpret_cast<u32x4 *>(base[w] + (uint64_t)g * 1024) = v;
}
static hipEvent_t e0, e1;
template <class F>
static double timeit(F launch, int it = 4) {
  launch();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < it; i++) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / it;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const uint32_t nwit = 24093, full = nwit >> 5, n = 4096;
  const uint64_t body = 32ull * nwit, GiB = 1ull << 30, MiB = 1ull << 20;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t S = n * body, C = 128 * MiB;
  const uint32_t nch = (uint32_t)((S + C - 1) / C), mA = (nch + 1) / 2, mB = nch / 2;     // chunks from rank A / rank B
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("free %.2f GiB; buffer %.2f GB = %u chunks of 128 MiB (A %u, B %u)\n", fr / (double)GiB, S / 1e9, nch, mA, mB);
  const double t0 = now();
  void *vaA = nullptr, *vaB = nullptr, *vaF = nullptr;
  CK(hipMemAddressReserve(&vaA, (size_t)mA * C, 2 * MiB, nullptr, 0));
  CK(hipMemAddressReserve(&vaB, (size_t)mA * C, 2 * MiB, nullptr, 0));
  CK(hipMemAddressReserve(&vaF, (size_t)nch * C, 2 * MiB, nullptr, 0));
  auto make_group = [&](std::vector<hipMemGenericAllocationHandle_t> &g, uint32_t m, void *va) -> int {
    g.resize(m);
    for (uint32_t i = 0; i < m; i++) {
      CK(hipMemCreate(&g[i], C, &prop, 0));
      CK(hipMemMap((uint8_t *)va + (size_t)i * C, C, 0, g[i], 0));
    }
    CK(hipMemSetAccess(va, (size_t)m * C, &acc, 1));
    return 0;
  };
  auto drop_group = [&](std::vector<hipMemGenericAllocationHandle_t> &g, void *va) -> int {
    for (size_t i = 0; i < g.size(); i++) { CK(hipMemUnmap((uint8_t *)va + i * C, C)); CK(hipMemRelease(g[i])); }
    g.clear();
    return 0;
  };
  auto rate = [&](double ms) { return n * (double)body / 1e6 / ms; };
  auto probe = [&](uint8_t *a, uint8_t *b) { return rate(timeit([&] { hipLaunchKernelGGL((k_split<4>), dim3(n / 4), dim3(64), 0, 0, a, b, body, full, n); })); };
  std::vector<hipMemGenericAllocationHandle_t> gA, gB, spacers;
  if (make_group(gA, mA, vaA)) return 1;
  // same-rank reference rate: both halves inside group A is impossible (too small), so: one more group right after it
  if (make_group(gB, mA, vaB)) return 1;
  const double r_same = probe((uint8_t *)vaA, (uint8_t *)vaB);
  printf("adjacent groups: %.0f GB/s  (t=%.3f s)\n", r_same, now() - t0);
  bool found = r_same > 6300;
  const uint64_t SP = 8 * GiB;
  std::vector<hipMemGenericAllocationHandle_t> failed;
  for (int k = 0; !found && k < 30; k++) {
    // keep the failed candidate allocated (else its holes are what the next candidate gets back)
    for (size_t i = 0; i < gB.size(); i++) { CK(hipMemUnmap((uint8_t *)vaB + i * C, C)); failed.push_back(gB[i]); }
    gB.clear();
    size_t f2 = 0;
    CK(hipMemGetInfo(&f2, &tot));
    if (f2 < SP + 8 * GiB) { printf("out of room\n"); break; }
    hipMemGenericAllocationHandle_t sp;
    CK(hipMemCreate(&sp, SP, &prop, 0));
    spacers.push_back(sp);
    size_t f3 = 0;
    CK(hipMemGetInfo(&f3, &tot));
    if (make_group(gB, mA, vaB)) return 1;
    const double r = probe((uint8_t *)vaA, (uint8_t *)vaB);
    printf("after %2d spacers (free %.1f -> %.1f GiB): %.0f GB/s (t=%.3f s)\n", k + 1, f2 / (double)GiB, f3 / (double)GiB, r, now() - t0);
    found = r > 1.15 * r_same;
  }
  for (auto h : spacers) CK(hipMemRelease(h));
  for (auto h : failed) CK(hipMemRelease(h));
  // final mapping: unmap the temporaries, interleave A0 B0 A1 B1 ...
  for (uint32_t i = 0; i < mA; i++) CK(hipMemUnmap((uint8_t *)vaA + (size_t)i * C, C));
  for (uint32_t i = 0; i < gB.size(); i++) CK(hipMemUnmap((uint8_t *)vaB + (size_t)i * C, C));
  uint32_t ia = 0, ib = 0;
  for (uint32_t i = 0; i < nch; i++) {
    hipMemGenericAllocationHandle_t h = ((i & 1) && ib < mB) ? gB[ib++] : gA[ia < mA ? ia++ : 0];
    CK(hipMemMap((uint8_t *)vaF + (size_t)i * C, C, 0, h, 0));
  }
  CK(hipMemSetAccess(vaF, (size_t)nch * C, &acc, 1));
  for (uint32_t i = mB; i < gB.size(); i++) CK(hipMemRelease(gB[i]));      // unused tail of group B
  CK(hipMemAddressFree(vaA, (size_t)mA * C));
  CK(hipMemAddressFree(vaB, (size_t)mA * C));
  CK(hipMemGetInfo(&fr, &tot));
  printf("set-up %.3f s, found=%d, free now %.2f GiB\n", now() - t0, (int)found, fr / (double)GiB);
  // the plain contiguous pattern on the interleaved buffer
  uint8_t *F = (uint8_t *)vaF;
  const double rf = rate(timeit([&] { hipLaunchKernelGGL((k_split<4>), dim3(n / 4), dim3(64), 0, 0, F, F + body, 2 * body, full, n); }, 10));
  printf("contiguous bodies on the interleaved buffer: %.0f GB/s\n", rf);
  // sanity: copies see one linear buffer
  std::vector<uint8_t> h(3 * C);
  CK(hipMemcpy(h.data(), F + C / 2, 2 * C, hipMemcpyDeviceToHost));
  uint64_t ones = 0;
  for (size_t i = 0; i < 2 * C; i += 32) ones += h[i];
  printf("D2H across chunk borders ok, ones=%llu of %llu slots\n", (unsigned long long)ones, (unsigned long long)(2 * C / 32));
  uint8_t *plain;
  CK(hipMalloc((void **)&plain, S));
  const double rp = rate(timeit([&] { hipLaunchKernelGGL((k_split<4>), dim3(n / 4), dim3(64), 0, 0, plain, plain + body, 2 * body, full, n); }, 10));
  printf("same pattern on a plain hipMalloc buffer: %.0f GB/s\n", rp);
  return 0;
}
