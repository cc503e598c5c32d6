// store_patterns.hip — which body-writing patterns are robust across allocations?  (pure stores)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// A: fused-kernel pattern: wave per block, W bodies per wave, all blocks resident
template <int W>
__global__ __launch_bounds__(64) void k_fused(uint8_t *out, uint64_t pitch, uint32_t full, uint32_t n) {
  const uint32_t wit0 = blockIdx.x * W, lane = threadIdx.x;
  u32x4 v = {lane & 1 ? 0u : 1u, 0, 0, 0};
  for (uint32_t g = 0; g < full; ++g)
#pragma unroll
    for (int w = 0; w < W; ++w) *reinterpret_cast<u32x4 *>(out + (uint64_t)(wit0 + w) * pitch + (uint64_t)g * 1024 + lane * 16) = v;
}
// B: cooperative: TPB-thread block writes ONE body at a time contiguously, bodies b, b+G, b+2G, ...
template <int TPB>
__global__ __launch_bounds__(TPB) void k_coop(uint8_t *out, uint64_t pitch, uint32_t body, uint32_t n) {
  u32x4 v = {threadIdx.x & 1 ? 0u : 1u, 0, 0, 0};
  for (uint32_t w = blockIdx.x; w < n; w += gridDim.x) {
    uint8_t *p = out + (uint64_t)w * pitch;
    for (uint32_t o = threadIdx.x * 16; o < body; o += TPB * 16) *reinterpret_cast<u32x4 *>(p + o) = v;
  }
}
// C: cooperative, block takes a contiguous run of bodies [b*R, b*R+R) then strides
template <int TPB>
__global__ __launch_bounds__(TPB) void k_coop_run(uint8_t *out, uint64_t pitch, uint32_t body, uint32_t n, uint32_t R) {
  u32x4 v = {threadIdx.x & 1 ? 0u : 1u, 0, 0, 0};
  for (uint32_t w0 = blockIdx.x * R; w0 < n; w0 += gridDim.x * R)
    for (uint32_t w = w0; w < w0 + R && w < n; ++w) {
      uint8_t *p = out + (uint64_t)w * pitch;
      for (uint32_t o = threadIdx.x * 16; o < body; o += TPB * 16) *reinterpret_cast<u32x4 *>(p + o) = v;
    }
}
// D: memset shape
__global__ __launch_bounds__(256) void k_fill(u32x4 *out, uint64_t n16) {
  u32x4 v = {1, 0, 0, 0};
  for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) out[i] = v;
}

int main() {
  const uint32_t n = 4096, nwit = 24093;
  const uint64_t body = 32ull * nwit;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint8_t *bufs[4];
  for (int b = 0; b < 4; b++) CK(hipMalloc((void **)&bufs[b], (uint64_t)n * body + (1 << 22)));
  for (int b = 0; b < 4; b++) {
    uint8_t *buf = bufs[b];
    auto timeit = [&](const char *name, auto launch) {
      for (int i = 0; i < 3; i++) launch();
      hipEventRecord(e0, 0);
      const int it = 20;
      for (int i = 0; i < it; i++) launch();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("buf%d %-34s %7.3f ms %7.0f GB/s\n", b, name, ms / it, n * (double)body / 1e9 / (ms / it * 1e-3));
    };
    timeit("fused W=4 (4096 streams)", [&] { hipLaunchKernelGGL(k_fused<4>, dim3(n / 4), dim3(64), 0, 0, buf, body, nwit >> 5, n); });
    timeit("coop 256x256", [&] { hipLaunchKernelGGL(k_coop<256>, dim3(256), dim3(256), 0, 0, buf, body, (uint32_t)body, n); });
    timeit("coop 512x256", [&] { hipLaunchKernelGGL(k_coop<256>, dim3(512), dim3(256), 0, 0, buf, body, (uint32_t)body, n); });
    timeit("coop 1024x256", [&] { hipLaunchKernelGGL(k_coop<256>, dim3(1024), dim3(256), 0, 0, buf, body, (uint32_t)body, n); });
    timeit("coop 256x512", [&] { hipLaunchKernelGGL(k_coop<512>, dim3(256), dim3(512), 0, 0, buf, body, (uint32_t)body, n); });
    timeit("coop 256x1024", [&] { hipLaunchKernelGGL(k_coop<1024>, dim3(256), dim3(1024), 0, 0, buf, body, (uint32_t)body, n); });
    timeit("coop 1024x64", [&] { hipLaunchKernelGGL(k_coop<64>, dim3(1024), dim3(64), 0, 0, buf, body, (uint32_t)body, n); });
    timeit("coop 2048x64", [&] { hipLaunchKernelGGL(k_coop<64>, dim3(2048), dim3(64), 0, 0, buf, body, (uint32_t)body, n); });
    timeit("coop_run16 256x256", [&] { hipLaunchKernelGGL(k_coop_run<256>, dim3(256), dim3(256), 0, 0, buf, body, (uint32_t)body, n, 16u); });
    timeit("memset-shape 256x256", [&] { hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, (u32x4 *)buf, (uint64_t)n * body / 16); });
  }
  return 0;
}
