// store_rotated.hip — body-owning waves (as in the fused kernel) but each wave walks its body's 4 KiB
// tiles in *class order*, class = absolute tile index mod 256, rotated by the wave's id, so that at any
// moment every class (HBM channel group) is written by about one wave.  Pure stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: sequential order (fused kernel today); 1: class-rotated; 2: class-rotated, start offset = global wave id
template <int MODE, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_body(uint8_t *out, uint64_t body, uint32_t n, uint32_t rounds) {
  const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  const uint32_t gw = blockIdx.x * WPB + wave;               // global wave id
  const uint32_t nwaves = gridDim.x * WPB;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t w = r * nwaves + gw;
    if (w >= n) break;
    const uint64_t a0 = (uint64_t)out + (uint64_t)w * body, a1 = a0 + body;   // byte range of the body
    const uint64_t t0 = a0 >> 12, t1 = (a1 + 4095) >> 12;                     // absolute tiles [t0, t1)
    if (MODE == 0) {
      for (uint64_t t = t0; t < t1; ++t) {
        const uint64_t lo = t << 12 > a0 ? t << 12 : a0, hi = (t + 1) << 12 < a1 ? (t + 1) << 12 : a1;
        for (uint64_t p = lo + lane * 16; p < hi; p += 1024) *reinterpret_cast<u32x4 *>(p) = v;
      }
    } else {
      const uint32_t start = MODE == 1 ? blockIdx.x : gw;
      for (uint32_t s = 0; s < 256; ++s) {
        const uint32_t c = (start + s) & 255;
        // the tile of this body whose class is c (at most one: bodies span < 256 tiles)
        uint64_t t = (t0 & ~255ull) | c;
        if (t < t0) t += 256;
        if (t >= t1) continue;
        const uint64_t lo = t << 12 > a0 ? t << 12 : a0, hi = (t + 1) << 12 < a1 ? (t + 1) << 12 : a1;
        for (uint64_t p = lo + lane * 16; p < hi; p += 1024) *reinterpret_cast<u32x4 *>(p) = v;
      }
    }
  }
}
int main() {
  const uint32_t n = 4096, nwit = 24093;
  const uint64_t body = 32ull * nwit;
  uint8_t *buf;
  CK(hipMalloc((void **)&buf, (uint64_t)n * body + (1 << 22)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto launch) {
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0, 0);
    const int it = 20;
    for (int i = 0; i < it; i++) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %7.3f ms %7.0f GB/s\n", name, ms / it, n * (double)body / 1e9 / (ms / it * 1e-3));
  };
#define RUN(MODE, WPB, GRID) timeit("mode=" #MODE " waves/block=" #WPB " grid=" #GRID, [&] { \
    hipLaunchKernelGGL((k_body<MODE, WPB>), dim3(GRID), dim3(64 * WPB), 0, 0, buf, body, n, (n + GRID * WPB - 1) / (GRID * WPB)); });
  for (int rep = 0; rep < 2; rep++) {
    RUN(0, 4, 256) RUN(1, 4, 256) RUN(2, 4, 256)
    RUN(0, 1, 256) RUN(2, 1, 256)
    RUN(0, 4, 1024) RUN(1, 4, 1024) RUN(2, 4, 1024)
    RUN(0, 8, 256) RUN(2, 8, 256)
    RUN(0, 16, 256) RUN(2, 16, 256)
  }
  return 0;
}
