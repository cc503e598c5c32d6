// store_ceiling.hip — what HBM write rate does the witness kernel's store pattern allow, without any
// of its compute?  Build-box experiment (gpurun), not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o store_ceiling store_ceiling.hip && ./store_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// same addressing as b3w expand(): one wave per block, W bodies per wave, 1 KiB per store
template <int W>
__global__ __launch_bounds__(64) void k_pattern(uint8_t *out, uint64_t pitch, uint32_t nwit, uint32_t n) {
  const uint32_t wit0 = blockIdx.x * W, lane = threadIdx.x;
  const uint32_t full = nwit >> 5;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  for (uint32_t g = 0; g < full; ++g) {
#pragma unroll
    for (int w = 0; w < W; ++w)
      *reinterpret_cast<u32x4 *>(out + (uint64_t)(wit0 + w) * pitch + (uint64_t)g * 1024 + lane * 16) = v;
  }
}
// blocks of 256 threads: 4 waves share the W bodies, wave j takes groups g = 4i + j
template <int W>
__global__ __launch_bounds__(256) void k_pattern256(uint8_t *out, uint64_t pitch, uint32_t nwit, uint32_t n) {
  const uint32_t wit0 = blockIdx.x * W, tid = threadIdx.x;
  const uint32_t full = nwit >> 5;
  u32x4 v = {tid & 1 ? 0u : 1u, 0, 0, 0};
  for (uint32_t g = 0; g + 4 <= full; g += 4) {
#pragma unroll
    for (int w = 0; w < W; ++w)
      *reinterpret_cast<u32x4 *>(out + (uint64_t)(wit0 + w) * pitch + (uint64_t)g * 1024 + tid * 16) = v;
  }
}
// classic grid-stride fill
__global__ __launch_bounds__(256) void k_fill(u32x4 *out, uint64_t n16) {
  u32x4 v = {1, 0, 0, 0};
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) out[i] = v;
}
// each block owns a contiguous chunk
__global__ __launch_bounds__(256) void k_fill_chunk(u32x4 *out, uint64_t n16) {
  u32x4 v = {1, 0, 0, 0};
  const uint64_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const uint64_t s = blockIdx.x * per, e = s + per < n16 ? s + per : n16;
  for (uint64_t i = s + threadIdx.x; i < e; i += 256) out[i] = v;
}

int main() {
  const uint32_t n = 4096, nwit = 24093;
  const uint64_t body = 32ull * nwit, pitchA = 771072;
  uint8_t *buf;
  const uint64_t bytes = (uint64_t)n * pitchA;
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
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms / it, gb / (ms / it * 1e-3));
  };
  const double gbody = (double)n * (nwit >> 5) * 1024 / 1e9;
  timeit("pattern W=4 wave/block contiguous pitch", [&] { hipLaunchKernelGGL(k_pattern<4>, dim3(n / 4), dim3(64), 0, 0, buf, body, nwit, n); }, gbody);
  timeit("pattern W=4 wave/block 128B-aligned pitch", [&] { hipLaunchKernelGGL(k_pattern<4>, dim3(n / 4), dim3(64), 0, 0, buf, pitchA, nwit, n); }, gbody);
  timeit("pattern W=1 wave/block contiguous", [&] { hipLaunchKernelGGL(k_pattern<1>, dim3(n), dim3(64), 0, 0, buf, body, nwit, n); }, gbody);
  timeit("pattern W=1 wave/block aligned", [&] { hipLaunchKernelGGL(k_pattern<1>, dim3(n), dim3(64), 0, 0, buf, pitchA, nwit, n); }, gbody);
  timeit("pattern W=16 wave/block contiguous", [&] { hipLaunchKernelGGL(k_pattern<16>, dim3(n / 16), dim3(64), 0, 0, buf, body, nwit, n); }, gbody);
  timeit("pattern256 W=4 (4 waves share bodies) contig", [&] { hipLaunchKernelGGL(k_pattern256<4>, dim3(n / 4), dim3(256), 0, 0, buf, body, nwit, n); }, gbody);
  timeit("pattern256 W=4 aligned", [&] { hipLaunchKernelGGL(k_pattern256<4>, dim3(n / 4), dim3(256), 0, 0, buf, pitchA, nwit, n); }, gbody);
  timeit("pattern256 W=1 aligned", [&] { hipLaunchKernelGGL(k_pattern256<1>, dim3(n), dim3(256), 0, 0, buf, pitchA, nwit, n); }, gbody);
  const uint64_t n16 = (uint64_t)n * body / 16;
  timeit("grid-stride fill 2048x256", [&] { hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (u32x4 *)buf, n16); }, n16 * 16 / 1e9);
  timeit("grid-stride fill 8192x256", [&] { hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (u32x4 *)buf, n16); }, n16 * 16 / 1e9);
  timeit("chunk fill 2048x256", [&] { hipLaunchKernelGGL(k_fill_chunk, dim3(2048), dim3(256), 0, 0, (u32x4 *)buf, n16); }, n16 * 16 / 1e9);
  timeit("chunk fill 16384x256", [&] { hipLaunchKernelGGL(k_fill_chunk, dim3(16384), dim3(256), 0, 0, (u32x4 *)buf, n16); }, n16 * 16 / 1e9);
  timeit("hipMemsetAsync", [&] { hipMemsetAsync(buf, 1, n16 * 16, 0); }, n16 * 16 / 1e9);
  return 0;
}
