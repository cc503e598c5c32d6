// mixed_alloc2.hip — prototype 2: build a VA-contiguous buffer from physical handles of two different HBM "classes".
// Handles (H bytes each) are created one by one, each classified by a split-store probe against one representative per
// class found so far; stop when two classes can each cover half the buffer; map them alternately; release the rest.
// build: hipcc --offload-arch=gfx950 -O3 -o mixed_alloc2 mixed_alloc2.hip ; run: ./mixed_alloc2 <H MiB> <buffer MB> <probe n>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <chrono>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
// n bodies, even ones from lo, odd ones from hi, one wave per body
__global__ __launch_bounds__(64) void k_probe(uint8_t *lo, uint8_t *hi, uint64_t pitch, uint32_t full) {
  const uint32_t i = blockIdx.x, lane = threadIdx.x;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  uint8_t *base = ((i & 1) ? hi : lo) + (uint64_t)(i >> 1) * pitch + lane * 16;
  for (uint32_t g = 0; g < full; ++g) *reinterpret_cast<u32x4 *>(base + (uint64_t)g * 1024) = v;
}
// W bodies per wave (stride ST apart), U consecutive 1 KiB groups of one body back to back
template <int W, int U>
__global__ __launch_bounds__(64) void k_family(uint8_t *out, uint64_t pitch, uint32_t full, uint32_t n, uint32_t st) {
  const uint32_t b = blockIdx.x, lane = threadIdx.x;
  const uint32_t first = (b / st) * (st * W) + b % st;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  uint8_t *base[W];
#pragma unroll
  for (int w = 0; w < W; ++w) base[w] = out + (uint64_t)(first + w * st) * pitch + lane * 16;
  if (first + (W - 1) * st >= n) return;
  uint32_t g = 0;
  for (; g + U <= full; g += U)
#pragma unroll
    for (int w = 0; w < W; ++w)
#pragma unroll
      for (int u = 0; u < U; ++u) *reinterpret_cast<u32x4 *>(base[w] + (uint64_t)(g + u) * 1024) = v;
}
template <int W>
__global__ __launch_bounds__(64) void k_fused(uint8_t *out, uint64_t pitch, uint32_t full, uint32_t n) {
  const uint32_t wit0 = blockIdx.x * W, lane = threadIdx.x;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  for (uint32_t g = 0; g < full; ++g)
#pragma unroll
    for (int w = 0; w < W; ++w) *reinterpret_cast<u32x4 *>(out + (uint64_t)(wit0 + w) * pitch + (uint64_t)g * 1024 + lane * 16) = v;
}
static hipEvent_t e0, e1;
template <class F>
static double timeit(F launch, int it = 3) {
  launch();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < it; i++) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / it;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const uint32_t nwit = 24093, full = nwit >> 5;
  const uint64_t body = 32ull * nwit, MiB = 1ull << 20, GiB = 1ull << 30;
  const uint64_t H = (argc > 1 ? atoll(argv[1]) : 512) * MiB;
  const uint64_t S = (argc > 2 ? atoll(argv[2]) : 3159) * 1000000ull;
  const uint32_t pn = argc > 3 ? atoi(argv[3]) : 1024;          // probe bodies (half per side)
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const uint32_t nh = (uint32_t)((S + H - 1) / H), need = (nh + 1) / 2;
  const uint32_t MAXH = (uint32_t)(160 * GiB / H);
  if ((uint64_t)(pn / 2) * body > H) { printf("probe does not fit a handle\n"); return 1; }
  printf("H=%llu MiB, buffer %.2f GB = %u handles (%u per class), probe n=%u\n", (unsigned long long)(H / MiB), S / 1e9, nh, need, pn);
  const double t0 = now();
  void *scr = nullptr;
  CK(hipMemAddressReserve(&scr, (size_t)MAXH * H, 1ull << 21, nullptr, 0));
  uint8_t *V = (uint8_t *)scr;
  std::vector<hipMemGenericAllocationHandle_t> h;
  std::vector<int> cls;
  std::vector<int> reps, count;
  auto prate = [&](uint8_t *a, uint8_t *b) {
    return pn * (double)body / 1e6 / timeit([&] { hipLaunchKernelGGL(k_probe, dim3(pn), dim3(64), 0, 0, a, b, body, full); });
  };
  double r_same = 0;
  int cx = -1, cy = -1;
  double t_create = 0, t_probe = 0;
  for (uint32_t i = 0; i < MAXH; i++) {
    double t1 = now();
    hipMemGenericAllocationHandle_t hh;
    if (hipMemCreate(&hh, H, &prop, 0) != hipSuccess) { printf("hipMemCreate failed at %u\n", i); break; }
    h.push_back(hh);
    CK(hipMemMap(V + (size_t)i * H, H, 0, hh, 0));
    CK(hipMemSetAccess(V + (size_t)i * H, H, &acc, 1));
    t_create += now() - t1; t1 = now();
    if (i == 0) {
      (void)prate(V, V);
      r_same = prate(V, V);             // both halves on the same addresses of one handle: certainly one class
      printf("same-handle rate %.0f\n", r_same);
    }
    int c = -1;
    double seen[3] = {0, 0, 0};
    for (size_t k = 0; k < reps.size(); k++) {
      const double r = prate(V + (size_t)reps[k] * H, V + (size_t)i * H);
      seen[k] = r;
      if (r < 6100) { c = (int)k; break; }
    }
    if (c < 0) {
      if (reps.size() < 3) { c = (int)reps.size(); reps.push_back((int)i); count.push_back(0); }
      else c = 9;   // mixed / unclassifiable
    }
    cls.push_back(c);
    if (c < 3) count[c]++;
    t_probe += now() - t1;
    for (size_t a = 0; a < count.size() && cx < 0; a++)
      for (size_t b = a + 1; b < count.size(); b++)
        if (count[a] >= (int)need && count[b] >= (int)need) { cx = (int)a; cy = (int)b; break; }
    if (cx >= 0) break;
  }
  printf("last probe rates: %.0f %.0f %.0f\nclasses in creation order: ", 0.0, 0.0, 0.0);
  for (int c : cls) printf("%d", c);
  printf("\ncreated %zu handles (%.1f GiB) in %.3f s (create+map %.3f, probes %.3f); classes chosen %d,%d\n", h.size(),
         h.size() * (double)H / GiB, now() - t0, t_create, t_probe, cx, cy);
  if (cx < 0) { printf("no second class found\n"); return 0; }
  // final mapping: last `need` handles of each class, alternating
  std::vector<int> X, Y;
  for (int i = (int)h.size() - 1; i >= 0; i--) { if (cls[i] == cx && X.size() < need) X.push_back(i); if (cls[i] == cy && Y.size() < need) Y.push_back(i); }
  void *fin = nullptr;
  CK(hipMemAddressReserve(&fin, (size_t)nh * H, 1ull << 21, nullptr, 0));
  uint8_t *F = (uint8_t *)fin;
  std::vector<char> used(h.size(), 0);
  for (size_t i = 0; i < h.size(); i++) CK(hipMemUnmap(V + i * H, H));
  for (uint32_t s = 0; s < nh; s++) {
    const int idx = (s & 1) ? Y[s / 2] : X[s / 2];
    used[idx] = 1;
    CK(hipMemMap(F + (size_t)s * H, H, 0, h[idx], 0));
  }
  CK(hipMemSetAccess(F, (size_t)nh * H, &acc, 1));
  for (size_t i = 0; i < h.size(); i++) if (!used[i]) CK(hipMemRelease(h[i]));
  CK(hipMemAddressFree(scr, (size_t)MAXH * H));
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("total set-up %.3f s; free now %.2f GiB\n", now() - t0, fr / (double)GiB);
  const uint32_t n = (uint32_t)(S / body) & ~3u;
  const double ms = timeit([&] { hipLaunchKernelGGL((k_fused<4>), dim3(n / 4), dim3(64), 0, 0, F, body, full, n); }, 6);
  printf("fused pattern, %u contiguous bodies in natural order on the mixed buffer: %.0f GB/s\n", n, n * (double)body / 1e6 / ms);
  {
    auto fam = [&](const char *name, auto launch) {
      const double t = timeit(launch, 8);
      printf("  %-44s %7.0f GB/s\n", name, n * (double)body / 1e6 / t);
    };
    const uint64_t ap = 771072;   // 128-byte aligned pitch
    const uint32_t na = (uint32_t)(S / ap) & ~31u;
    fam("W4 U1 contiguous pitch", [&] { hipLaunchKernelGGL((k_family<4, 1>), dim3(n / 4), dim3(64), 0, 0, F, body, full, n, 1u); });
    fam("W4 U1 contiguous pitch, stride-4 bodies", [&] { hipLaunchKernelGGL((k_family<4, 1>), dim3(n / 4), dim3(64), 0, 0, F, body, full, n, 4u); });
    fam("W4 U1 aligned pitch", [&] { hipLaunchKernelGGL((k_family<4, 1>), dim3(na / 4), dim3(64), 0, 0, F, ap, full, na, 1u); });
    fam("W4 U2 aligned pitch", [&] { hipLaunchKernelGGL((k_family<4, 2>), dim3(na / 4), dim3(64), 0, 0, F, ap, full, na, 1u); });
    fam("W4 U4 aligned pitch", [&] { hipLaunchKernelGGL((k_family<4, 4>), dim3(na / 4), dim3(64), 0, 0, F, ap, full, na, 1u); });
    fam("W8 U1 aligned pitch", [&] { hipLaunchKernelGGL((k_family<8, 1>), dim3(na / 8), dim3(64), 0, 0, F, ap, full, na, 1u); });
    fam("W8 U2 aligned pitch", [&] { hipLaunchKernelGGL((k_family<8, 2>), dim3(na / 8), dim3(64), 0, 0, F, ap, full, na, 1u); });
    fam("W2 U1 aligned pitch", [&] { hipLaunchKernelGGL((k_family<2, 1>), dim3(na / 2), dim3(64), 0, 0, F, ap, full, na, 1u); });
    fam("W2 U4 aligned pitch", [&] { hipLaunchKernelGGL((k_family<2, 4>), dim3(na / 2), dim3(64), 0, 0, F, ap, full, na, 1u); });
    fam("W16 U1 aligned pitch", [&] { hipLaunchKernelGGL((k_family<16, 1>), dim3(na / 16), dim3(64), 0, 0, F, ap, full, na, 1u); });
    fam("W1 U1 aligned pitch", [&] { hipLaunchKernelGGL((k_family<1, 1>), dim3(na), dim3(64), 0, 0, F, ap, full, na, 1u); });
  }
  uint8_t *plain;
  CK(hipMalloc((void **)&plain, (size_t)n * body));
  const double mp = timeit([&] { hipLaunchKernelGGL((k_fused<4>), dim3(n / 4), dim3(64), 0, 0, plain, body, full, n); }, 6);
  printf("same on a plain hipMalloc buffer: %.0f GB/s\n", n * (double)body / 1e6 / mp);
  return 0;
}
