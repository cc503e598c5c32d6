// place_cost.hip — what the steps of a placement search cost on this ROCm stack (r05): hipMemCreate by size, map + access,
// the split-store probe, hipMalloc / hipFree of a batch-sized buffer.  Decides how b3w_placement.hip walks towards the second
// class of HBM: 256 MiB handles one by one (r01-r04) or large unprobed "skip" handles with a probed handle after each.
//   hipcc --offload-arch=gfx950 -O2 -o place_cost place_cost.hip && ./place_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ __launch_bounds__(64) void probe(uint8_t *lo, uint8_t *hi, uint64_t pitch, uint32_t groups) {
  const uint32_t i = blockIdx.x, lane = threadIdx.x;
  const u32x4 v = {0, 0, 0, 0};
  uint8_t *base = ((i & 1) ? hi : lo) + (uint64_t)(i >> 1) * pitch + lane * 16;
  for (uint32_t g = 0; g < groups; ++g) *reinterpret_cast<u32x4 *>(base + (uint64_t)g * 1024) = v;
}

int main() {
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const uint64_t MiB = 1ull << 20, GiB = 1ull << 30;
  void *va = nullptr;
  CK(hipMemAddressReserve(&va, 1ull << 40, 2 * MiB, nullptr, 0));
  uint8_t *arena = (uint8_t *)va;
  uint64_t used = 0;
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("free %.1f GiB of %.1f GiB\n", fr / (double)GiB, tot / (double)GiB);
  // 1. hipMemCreate by size (fresh memory), then release
  for (uint64_t sz : {256 * MiB, 1 * GiB, 4 * GiB, 16 * GiB}) {
    std::vector<hipMemGenericAllocationHandle_t> hs;
    const int reps = sz <= GiB ? 16 : 3;
    double t0 = now();
    for (int i = 0; i < reps; i++) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, sz, &prop, 0)); hs.push_back(h); }
    double t1 = now();
    // map + access of the first
    CK(hipMemMap(arena + used, sz, 0, hs[0], 0));
    double t2 = now();
    CK(hipMemSetAccess(arena + used, sz, &acc, 1));
    double t3 = now();
    CK(hipMemUnmap(arena + used, sz));
    double t4 = now();
    used += sz;
    for (auto h : hs) CK(hipMemRelease(h));
    double t5 = now();
    printf("hipMemCreate %6llu MiB: %8.3f ms each (%d calls) | map %.3f ms  setaccess %.3f ms  unmap %.3f ms | release %.3f ms each\n",
           (unsigned long long)(sz / MiB), (t1 - t0) * 1e3 / reps, reps, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3 / reps);
  }
  // 2. the probe: two mapped 256 MiB handles, 1 + 3 launches as the library does
  {
    hipMemGenericAllocationHandle_t a, b;
    CK(hipMemCreate(&a, 256 * MiB, &prop, 0));
    CK(hipMemCreate(&b, 256 * MiB, &prop, 0));
    uint8_t *pa = arena + used; used += 256 * MiB;
    uint8_t *pb = arena + used; used += 256 * MiB;
    CK(hipMemMap(pa, 256 * MiB, 0, a, 0)); CK(hipMemSetAccess(pa, 256 * MiB, &acc, 1));
    CK(hipMemMap(pb, 256 * MiB, 0, b, 0)); CK(hipMemSetAccess(pb, 256 * MiB, &acc, 1));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
      double t0 = now();
      hipLaunchKernelGGL(probe, dim3(512), dim3(64), 0, 0, pa, pb, 768 * 1024ull, 768u);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 3; i++) hipLaunchKernelGGL(probe, dim3(512), dim3(64), 0, 0, pa, pb, 768 * 1024ull, 768u);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      double t1 = now();
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("probe (1 + 3 launches of 402 MB): host %.3f ms, device %.3f ms for 3 = %.0f GB/s\n", (t1 - t0) * 1e3, ms, 3 * 512 * 768 * 1024.0 / 1e6 / ms);
    }
    CK(hipMemUnmap(pa, 256 * MiB)); CK(hipMemUnmap(pb, 256 * MiB));
    CK(hipMemRelease(a)); CK(hipMemRelease(b));
  }
  // 3. hipMalloc / hipFree of a config-2 batch buffer (3.16 GB) and of 12 GB
  for (uint64_t sz : {3158ull * 1000 * 1000, 12ull * GiB}) {
    for (int rep = 0; rep < 3; rep++) {
      void *p = nullptr;
      double t0 = now();
      CK(hipMalloc(&p, sz));
      double t1 = now();
      CK(hipMemset(p, 0, 256));
      CK(hipDeviceSynchronize());
      double t2 = now();
      CK(hipFree(p));
      double t3 = now();
      printf("hipMalloc %.2f GB: %.3f ms, first touch %.3f ms, hipFree %.3f ms\n", sz / 1e9, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
    }
  }
  // 4. a long walk: 256 MiB handles until 96 GiB, time per 8 GiB
  {
    std::vector<hipMemGenericAllocationHandle_t> hs;
    double t0 = now();
    for (int i = 0; i < 384; i++) {
      hipMemGenericAllocationHandle_t h;
      if (hipMemCreate(&h, 256 * MiB, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
      hs.push_back(h);
      if ((i + 1) % 64 == 0) { double t = now(); printf("walk: %3d handles of 256 MiB created, %.3f s so far\n", i + 1, t - t0); }
    }
    double t1 = now();
    for (auto h : hs) CK(hipMemRelease(h));
    printf("walk: released %zu handles in %.3f s\n", hs.size(), now() - t1);
  }
  return 0;
}
