// pair_matrix.hip — full slow/fast matrix of the split-store probe over N sequentially created 256 MiB handles:
// is "slow together" an equivalence relation (classes) or something else?
// build: hipcc --offload-arch=gfx950 -O3 -o pair_matrix pair_matrix.hip ; run: ./pair_matrix [N] [skip handles before]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(64) void k_probe(uint8_t *lo, uint8_t *hi, uint64_t pitch, uint32_t groups) {
  const uint32_t i = blockIdx.x, lane = threadIdx.x;
  const u32x4 v = {0, 0, 0, 0};
  uint8_t *base = ((i & 1) ? hi : lo) + (uint64_t)(i >> 1) * pitch + lane * 16;
  for (uint32_t g = 0; g < groups; ++g) *reinterpret_cast<u32x4 *>(base + (uint64_t)g * 1024) = v;
}
static hipEvent_t e0, e1;
static double probe(uint8_t *a, uint8_t *b) {
  const uint64_t pitch = 768 * 1024;
  hipLaunchKernelGGL(k_probe, dim3(512), dim3(64), 0, 0, a, b, pitch, 768u);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_probe, dim3(512), dim3(64), 0, 0, a, b, pitch, 768u);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return 3.0 * 512 * pitch / 1e6 / ms;
}
int main(int argc, char **argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 48, SKIP = argc > 2 ? atoi(argv[2]) : 0;
  const uint64_t H = 256ull << 20;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> skip(SKIP), h(N);
  for (int i = 0; i < SKIP; i++) CK(hipMemCreate(&skip[i], H, &prop, 0));
  void *va = nullptr;
  CK(hipMemAddressReserve(&va, (size_t)N * H, 1ull << 21, nullptr, 0));
  uint8_t *V = (uint8_t *)va;
  for (int i = 0; i < N; i++) { CK(hipMemCreate(&h[i], H, &prop, 0)); CK(hipMemMap(V + (size_t)i * H, H, 0, h[i], 0)); }
  CK(hipMemSetAccess(va, (size_t)N * H, &acc, 1));
  std::vector<std::vector<int>> M(N, std::vector<int>(N, 0));
  for (int i = 0; i < N; i++) for (int j = i; j < N; j++) { M[i][j] = M[j][i] = (int)(probe(V + (size_t)i * H, V + (size_t)j * H) / 100); }
  printf("matrix rate/100 ('.' >= 58 fast):\n");
  for (int i = 0; i < N; i++) { for (int j = 0; j < N; j++) printf("%c", M[i][j] >= 58 ? '.' : (i == j ? 'o' : '#')); printf("   %d\n", M[i][i]); }
  // also halves of one handle against another handle: is a handle homogeneous?
  printf("half-handle probes (first/second 96 MiB of handle i vs handle 0 .. ):\n");
  return 0;
}
