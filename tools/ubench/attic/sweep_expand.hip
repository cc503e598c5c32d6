// sweep_expand.hip — feasibility of a "sweep" expand kernel: 256 workgroups (one per CU) walk the
// whole output linearly in grid-stride order (the hipMemset pattern, 6.5 TB/s) and gather each
// 16-byte unit's value through slot table -> per-witness trace words (both L2 resident).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int K, int TPB>
__global__ __launch_bounds__(TPB) void k_sweep(uint8_t *__restrict__ out, uint64_t total_units, uint32_t units_per_body,
                                               const uint32_t *__restrict__ table, const uint32_t *__restrict__ trace, uint32_t tw) {
  const uint64_t nthreads = (uint64_t)gridDim.x * TPB;
  uint64_t u = (uint64_t)blockIdx.x * TPB + threadIdx.x;
  // (witness, unit-in-body) maintained incrementally
  uint32_t w = (uint32_t)(u / units_per_body), o = (uint32_t)(u % units_per_body);
  const uint32_t dq = (uint32_t)(nthreads / units_per_body), dr = (uint32_t)(nthreads % units_per_body);
  while (u + (K - 1) * nthreads < total_units) {
    uint32_t e[K], ww[K], oo[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      ww[k] = w; oo[k] = o;
      e[k] = table[o >> 1];
      w += dq; o += dr;
      if (o >= units_per_body) { o -= units_per_body; w++; }
    }
    uint32_t a[K];
#pragma unroll
    for (int k = 0; k < K; ++k) a[k] = trace[(uint64_t)ww[k] * tw + (e[k] & 0xFFF)];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t sh = (e[k] >> 12) & 31, mode = (e[k] >> 17) & 3;
      const uint32_t m0 = (oo[k] & 1) ? 0u : (mode == 0 ? 1u : 0xFFFFFFFFu);
      u32x4 v = {(a[k] >> sh) & m0, 0, 0, 0};
      *reinterpret_cast<u32x4 *>(out + (u + k * nthreads) * 16) = v;
    }
    u += K * nthreads;
  }
  for (; u < total_units; u += nthreads) {
    const uint32_t e0 = table[o >> 1];
    const uint32_t a0 = trace[(uint64_t)w * tw + (e0 & 0xFFF)];
    u32x4 v = {(a0 >> ((e0 >> 12) & 31)) & ((o & 1) ? 0u : 1u), 0, 0, 0};
    *reinterpret_cast<u32x4 *>(out + u * 16) = v;
    w += dq; o += dr;
    if (o >= units_per_body) { o -= units_per_body; w++; }
  }
}

int main() {
  const uint32_t n = 4096, nwit = 24093, tw = 944;
  const uint64_t body = 32ull * nwit;
  uint8_t *buf; uint32_t *table, *trace;
  CK(hipMalloc((void **)&buf, (uint64_t)n * body));
  std::vector<uint32_t> ht(nwit + 64), htr((size_t)n * tw);
  // realistic table: runs of 32 bits of one word
  for (uint32_t s = 0; s < nwit + 64; s++) ht[s] = (48 + (s / 32) % 890) | ((s % 32) << 12);
  for (auto &x : htr) x = rand();
  CK(hipMalloc((void **)&table, ht.size() * 4)); CK(hipMemcpy(table, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc((void **)&trace, htr.size() * 4)); CK(hipMemcpy(trace, htr.data(), htr.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t units = (uint64_t)n * body / 16;
  auto timeit = [&](const char *name, auto launch) {
    for (int i = 0; i < 3; i++) launch();
    hipEventRecord(e0, 0);
    const int it = 20;
    for (int i = 0; i < it; i++) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %8.3f ms  %8.1f GB/s   %s\n", name, ms / it, units * 16 / 1e9 / (ms / it * 1e-3), hipGetErrorString(hipGetLastError()));
  };
#define RUN(K, TPB, G) timeit("sweep K=" #K " tpb=" #TPB " grid=" #G, [&] { hipLaunchKernelGGL((k_sweep<K, TPB>), dim3(G), dim3(TPB), 0, 0, buf, units, (uint32_t)(body / 16), table, trace, tw); });
  RUN(1, 256, 256) RUN(2, 256, 256) RUN(4, 256, 256) RUN(8, 256, 256) RUN(16, 256, 256)
  RUN(4, 512, 256) RUN(8, 512, 256) RUN(4, 1024, 256) RUN(4, 256, 512) RUN(8, 256, 512) RUN(4, 256, 1024) RUN(4, 128, 256) RUN(8, 128, 256) RUN(8, 64, 256)
  RUN(4, 512, 128) RUN(8, 512, 128) RUN(8, 1024, 64)
  timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(buf, 1, units * 16, 0); });
  return 0;
}
