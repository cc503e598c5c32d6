// store_shapes3.hip — can a workgroup hold MORE waves than one tile needs and stay on the fill shape?
// PAIRS wave-pairs per workgroup; pair q writes the workgroup's tiles k = q, q+PAIRS, ... (tile = b + 256k).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int PAIRS>
__global__ __launch_bounds__(128 * PAIRS) void k_pairs(uint8_t *out, uint64_t total) {
  const uint32_t q = threadIdx.x / 128, l = threadIdx.x % 128;
  u32x4 v = {1, 0, 0, 0}, z = {0, 0, 0, 0};
  for (uint64_t k = q;; k += PAIRS) {
    const uint64_t t = blockIdx.x + k * gridDim.x;
    if ((t + 1) * 4096 > total) break;
    uint8_t *p = out + t * 4096 + l * 32;
    *reinterpret_cast<u32x4 *>(p) = v;
    *reinterpret_cast<u32x4 *>(p + 16) = z;
  }
}
// pair q owns a contiguous quarter of the workgroup's tile sequence instead of interleaving
template <int PAIRS>
__global__ __launch_bounds__(128 * PAIRS) void k_pairs_split(uint8_t *out, uint64_t total) {
  const uint32_t q = threadIdx.x / 128, l = threadIdx.x % 128;
  const uint64_t ntiles = total / 4096, per_block = ntiles / gridDim.x, per_pair = per_block / PAIRS;
  u32x4 v = {1, 0, 0, 0}, z = {0, 0, 0, 0};
  for (uint64_t k = q * per_pair; k < (q + 1) * per_pair; ++k) {
    uint8_t *p = out + (blockIdx.x + k * gridDim.x) * 4096 + l * 32;
    *reinterpret_cast<u32x4 *>(p) = v;
    *reinterpret_cast<u32x4 *>(p + 16) = z;
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
    printf("%-40s %7.3f ms %7.0f GB/s\n", name, ms / it, total / 1e9 / (ms / it * 1e-3));
  };
  for (int rep = 0; rep < 2; rep++) {
    timeit("pairs=1 (128 thr)", [&] { hipLaunchKernelGGL(k_pairs<1>, dim3(256), dim3(128), 0, 0, buf, total); });
    timeit("pairs=2 interleaved (256 thr)", [&] { hipLaunchKernelGGL(k_pairs<2>, dim3(256), dim3(256), 0, 0, buf, total); });
    timeit("pairs=4 interleaved (512 thr)", [&] { hipLaunchKernelGGL(k_pairs<4>, dim3(256), dim3(512), 0, 0, buf, total); });
    timeit("pairs=8 interleaved (1024 thr)", [&] { hipLaunchKernelGGL(k_pairs<8>, dim3(256), dim3(1024), 0, 0, buf, total); });
    timeit("pairs=2 split halves", [&] { hipLaunchKernelGGL(k_pairs_split<2>, dim3(256), dim3(256), 0, 0, buf, total); });
    timeit("pairs=4 split quarters", [&] { hipLaunchKernelGGL(k_pairs_split<4>, dim3(256), dim3(512), 0, 0, buf, total); });
  }
  return 0;
}
