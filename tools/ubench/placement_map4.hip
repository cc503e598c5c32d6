// placement_map4.hip — region structure of (nearly) the whole HBM through one linear allocation:
// body-per-wave store pattern (4096 x 770 976 B windows) every 1 GiB; a window is fast iff it straddles a boundary.
// build: hipcc --offload-arch=gfx950 -O3 -o placement_map4 placement_map4.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int W>
__global__ __launch_bounds__(64) void k_fused(uint8_t *lo, uint8_t *hi, uint64_t pitch, uint32_t full, uint32_t n) {
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
int main() {
  const uint32_t nwit = 24093, full = nwit >> 5, n = 4096;
  const uint64_t body = 32ull * nwit, GiB = 1ull << 30;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  const uint64_t big = (fr / GiB - 6) * GiB;
  uint8_t *p;
  CK(hipMalloc((void **)&p, big));
  printf("free %.2f GiB total %.2f GiB; allocation %llu GiB at %p\n", fr / (double)GiB, tot / (double)GiB, (unsigned long long)(big / GiB), (void *)p);
  auto rate = [&](double ms) { return n * (double)body / 1e6 / ms; };
  // contiguous window: lo = start, hi = start + body (bodies alternate lo/hi with pitch 2*body => plain contiguous layout)
  std::vector<double> r;
  for (uint64_t off = 0; off + 4 * GiB <= big; off += GiB) {
    const double a = timeit([&] { hipLaunchKernelGGL((k_fused<4>), dim3(n / 4), dim3(64), 0, 0, p + off, p + off + body, 2 * body, full, n); });
    r.push_back(rate(a));
  }
  printf("# window start (GiB): rate; '*' = fast\n");
  for (size_t i = 0; i < r.size(); i++) { printf("%3zu:%5.0f%s ", i, r[i], r[i] > 6000 ? "*" : " "); if (i % 8 == 7) printf("\n"); }
  printf("\n");
  // pairwise: two half-sets at offsets 1 + 16*j GiB
  std::vector<uint64_t> pts;
  for (uint64_t o = GiB; o + 3 * GiB <= big; o += 16 * GiB) pts.push_back(o);
  printf("# pair matrix: half the bodies at row offset, half at column offset (GiB)\n      ");
  for (uint64_t c : pts) printf("%6llu", (unsigned long long)(c / GiB));
  printf("\n");
  for (uint64_t a : pts) {
    printf("%6llu", (unsigned long long)(a / GiB));
    for (uint64_t b : pts) {
      if (b <= a) { printf("     ."); continue; }
      const double t = timeit([&] { hipLaunchKernelGGL((k_fused<4>), dim3(n / 4), dim3(64), 0, 0, p + a, p + b, body, full, n); }, 3);
      printf("%6.0f", rate(t));
    }
    printf("\n");
  }
  return 0;
}
