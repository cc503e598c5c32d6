// store_shapes2.hip — does the fill shape survive "one 32-byte slot per lane" (two 16-byte stores at a
// 32-byte lane stride) and 128-thread workgroups?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// each thread writes BYTES contiguous bytes as BYTES/16 stores; block tile = TPB*BYTES; blocks sweep tiles b, b+G, ...
template <int TPB, int BYTES>
__global__ __launch_bounds__(TPB) void k_sweep(uint8_t *out, uint64_t total) {
  const uint64_t tile = (uint64_t)TPB * BYTES;
  u32x4 v = {1, 0, 0, 0}, z = {0, 0, 0, 0};
  for (uint64_t t = blockIdx.x; (t + 1) * tile <= total; t += gridDim.x) {
    uint8_t *p = out + t * tile + threadIdx.x * BYTES;
#pragma unroll
    for (int i = 0; i < BYTES / 16; ++i) *reinterpret_cast<u32x4 *>(p + 16 * i) = i ? z : v;
  }
}
int main() {
  const uint64_t total = 4096ull * 770976;
  uint8_t *buf;
  CK(hipMalloc((void **)&buf, total + (1 << 22)));
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
    printf("%-46s %7.3f ms %7.0f GB/s\n", name, ms / it, total / 1e9 / (ms / it * 1e-3));
  };
#define RUN(TPB, BYTES, G) timeit("tpb=" #TPB " bytes/thread=" #BYTES " grid=" #G, [&] { hipLaunchKernelGGL((k_sweep<TPB, BYTES>), dim3(G), dim3(TPB), 0, 0, buf, total); });
  for (int rep = 0; rep < 2; rep++) {
    RUN(256, 16, 256) RUN(128, 32, 256) RUN(256, 32, 256) RUN(64, 64, 256) RUN(128, 16, 256) RUN(128, 16, 512) RUN(512, 16, 256) RUN(64, 32, 256)
    RUN(128, 32, 512) RUN(256, 32, 128) RUN(256, 16, 248) RUN(256, 16, 264)
  }
  return 0;
}
