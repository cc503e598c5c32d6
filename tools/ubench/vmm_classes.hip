// vmm_classes.hip — which "class" (fast when mixed, slow when alone) do sequentially created physical handles fall in,
// and does the class follow the physical handle or the virtual address it is mapped at?
// build: hipcc --offload-arch=gfx950 -O3 -o vmm_classes vmm_classes.hip
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
  const uint32_t nwit = 24093, full = nwit >> 5, n = 4096;
  const uint64_t body = 32ull * nwit, GiB = 1ull << 30;
  const uint64_t H = (argc > 1 ? atoll(argv[1]) : 2) * GiB;     // handle size
  const int NH = argc > 2 ? atoi(argv[2]) : 64;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  void *va = nullptr;
  CK(hipMemAddressReserve(&va, (size_t)NH * H, 1ull << 21, nullptr, 0));
  printf("va %p, %d handles of %llu GiB\n", va, NH, (unsigned long long)(H / GiB));
  std::vector<hipMemGenericAllocationHandle_t> h(NH);
  double t0 = now();
  for (int i = 0; i < NH; i++) CK(hipMemCreate(&h[i], H, &prop, 0));
  printf("create: %.3f s\n", now() - t0); t0 = now();
  for (int i = 0; i < NH; i++) CK(hipMemMap((uint8_t *)va + (size_t)i * H, H, 0, h[i], 0));
  CK(hipMemSetAccess(va, (size_t)NH * H, &acc, 1));
  printf("map: %.3f s\n", now() - t0); t0 = now();
  auto rate = [&](double ms) { return n * (double)body / 1e6 / ms; };
  auto probe = [&](uint8_t *a, uint8_t *b) { return rate(timeit([&] { hipLaunchKernelGGL((k_split<4>), dim3(n / 4), dim3(64), 0, 0, a, b, body, full, n); })); };
  uint8_t *V = (uint8_t *)va;
  printf("first touch probe: %.0f\n", probe(V, V + H));
  printf("touch: %.3f s\n", now() - t0); t0 = now();
  // classes by union-find style: class id = index of first handle it is "slow" with
  std::vector<int> cls(NH, -1);
  std::vector<int> reps;
  for (int i = 0; i < NH; i++) {
    for (int r : reps) { if (probe(V + (size_t)r * H, V + (size_t)i * H) < 6000) { cls[i] = cls[r]; break; } }
    if (cls[i] < 0) { cls[i] = (int)reps.size(); reps.push_back(i); }
  }
  printf("classes of handles in creation order (mapped at va + i*H):\n");
  for (int i = 0; i < NH; i++) printf("%d", cls[i]);
  printf("\nprobes: %.3f s\n", now() - t0);
  // remap in reverse VA order: handle i at slot NH-1-i
  for (int i = 0; i < NH; i++) CK(hipMemUnmap(V + (size_t)i * H, H));
  for (int i = 0; i < NH; i++) CK(hipMemMap(V + (size_t)(NH - 1 - i) * H, H, 0, h[i], 0));
  CK(hipMemSetAccess(va, (size_t)NH * H, &acc, 1));
  std::vector<int> cls2(NH, -1);
  for (int s = 0; s < NH; s++) {            // s = VA slot; compare against the slots now holding the old representatives
    for (int r : reps) { if (probe(V + (size_t)(NH - 1 - r) * H, V + (size_t)s * H) < 6000) { cls2[s] = cls[r]; break; } }
  }
  printf("classes by VA slot after reversing the mapping (slot s holds handle NH-1-s):\n");
  for (int s = 0; s < NH; s++) printf("%c", cls2[s] < 0 ? '?' : '0' + cls2[s]);
  printf("\n");
  // within one handle: contiguous window at the seam between slot s and s+1 (plain layout): fast iff classes differ
  printf("seam windows (contiguous bodies centred on the slot border): ");
  for (int s = 0; s + 1 < NH && s < 24; s++) {
    uint8_t *c = V + (size_t)(s + 1) * H - ((n / 2) * body & ~4095ull);
    printf("%.0f ", rate(timeit([&] { hipLaunchKernelGGL((k_split<4>), dim3(n / 4), dim3(64), 0, 0, c, c + body, 2 * body, full, n); })));
  }
  printf("\n");
  return 0;
}
