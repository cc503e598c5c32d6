// store_ceiling2.hip — which store flavour / grid shape reaches the hipMemset rate (6.5 TB/s)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__device__ __forceinline__ void st(void *p, u32x4 v) {
  if (MODE == 0) *reinterpret_cast<u32x4 *>(p) = v;
  else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  else if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

// wave-per-block body pattern (as the witness kernel), W bodies per wave
template <int W, int MODE>
__global__ __launch_bounds__(64) void k_pattern(uint8_t *out, uint64_t pitch, uint32_t nwit) {
  const uint32_t wit0 = blockIdx.x * W, lane = threadIdx.x;
  const uint32_t full = nwit >> 5;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  for (uint32_t g = 0; g < full; ++g) {
#pragma unroll
    for (int w = 0; w < W; ++w) st<MODE>(out + (uint64_t)(wit0 + w) * pitch + (uint64_t)g * 1024 + lane * 16, v);
  }
}
template <int MODE, int TPB>
__global__ __launch_bounds__(TPB) void k_fill(u32x4 *out, uint64_t n16) {
  u32x4 v = {1, 0, 0, 0};
  for (uint64_t i = blockIdx.x * (uint64_t)TPB + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * TPB) st<MODE>(out + i, v);
}
// persistent waves sweeping the buffer in big contiguous tiles: block b takes tiles b, b+G, ...
template <int MODE, int TPB, int TILE_KB>
__global__ __launch_bounds__(TPB) void k_tiles(uint8_t *out, uint64_t bytes) {
  const uint64_t tile = (uint64_t)TILE_KB * 1024;
  u32x4 v = {1, 0, 0, 0};
  for (uint64_t t = blockIdx.x; t * tile < bytes; t += gridDim.x) {
    uint8_t *base = out + t * tile;
    for (uint32_t o = threadIdx.x * 16; o < tile; o += TPB * 16) st<MODE>(base + o, v);
  }
}

int main() {
  const uint32_t n = 4096, nwit = 24093;
  const uint64_t body = 32ull * nwit;
  uint8_t *buf;
  const uint64_t bytes = (uint64_t)n * 771072;
  CK(hipMalloc((void **)&buf, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char *name, auto launch, double gb) {
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0, 0);
    const int it = 20;
    for (int i = 0; i < it; i++) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.3f ms  %8.1f GB/s\n", name, ms / it, gb / (ms / it * 1e-3));
  };
  const double gbody = (double)n * (nwit >> 5) * 1024 / 1e9;
  const uint64_t n16 = (uint64_t)n * body / 16;
  const double gfill = n16 * 16 / 1e9;
#define PAT(MODE, label) timeit("pattern W=4 " label, [&] { hipLaunchKernelGGL((k_pattern<4, MODE>), dim3(n / 4), dim3(64), 0, 0, buf, body, nwit); }, gbody);
  PAT(0, "plain") PAT(1, "sc0") PAT(2, "sc1") PAT(3, "sc0 sc1") PAT(4, "nt") PAT(5, "sc0 sc1 nt")
  for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
    char nm[96];
    snprintf(nm, sizeof nm, "grid-stride fill %dx256 plain", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_fill<0, 256>), dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, n16); }, gfill);
  }
  for (int grid : {256, 1024, 4096}) {
    char nm[96];
    snprintf(nm, sizeof nm, "grid-stride fill %dx1024 plain", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_fill<0, 1024>), dim3(grid), dim3(1024), 0, 0, (u32x4 *)buf, n16); }, gfill);
    snprintf(nm, sizeof nm, "grid-stride fill %dx256 sc0 sc1", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_fill<3, 256>), dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, n16); }, gfill);
    snprintf(nm, sizeof nm, "grid-stride fill %dx256 nt", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_fill<4, 256>), dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, n16); }, gfill);
  }
  for (int grid : {256, 512, 1024, 2048}) {
    char nm[96];
    snprintf(nm, sizeof nm, "tiles 64KB %dx256 plain", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_tiles<0, 256, 64>), dim3(grid), dim3(256), 0, 0, buf, n16 * 16); }, gfill);
    snprintf(nm, sizeof nm, "tiles 4KB %dx256 plain", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_tiles<0, 256, 4>), dim3(grid), dim3(256), 0, 0, buf, n16 * 16); }, gfill);
    snprintf(nm, sizeof nm, "tiles 1024KB %dx256 plain", grid);
    timeit(nm, [&] { hipLaunchKernelGGL((k_tiles<0, 256, 1024>), dim3(grid), dim3(256), 0, 0, buf, n16 * 16); }, gfill);
  }
  timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(buf, 1, n16 * 16, 0); }, gfill);
  return 0;
}
