// read_ceiling.hip — how fast can this chip READ a large buffer with the walk kernel's access pattern and nothing else?
// Persistent 512-thread workgroups, each walking its own contiguous share in 32 KiB units: every wave issues four 1 KiB
// global_load_dwordx4 per unit (its two groups of 64 elements, low and high halves), AHEAD units ahead of their use, with or
// without a workgroup barrier per unit, with plain or non-temporal loads.  The loaded words are only XORed together.
// The number is the ceiling the constraint check's HBM roofline fraction should be read against (profiles/r04/read_ceiling.log).
//   build: hipcc --offload-arch=gfx950 -O3 -o read_ceiling read_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const uint8_t *p) {
  const u32x4 *q = reinterpret_cast<const u32x4 *>(p);
  if constexpr (NT) return __builtin_nontemporal_load(q);
  else return *q;
}

template <int AHEAD, bool NT, bool BARRIER>
__global__ __launch_bounds__(512) void read_kernel(const uint8_t *__restrict__ buf, uint64_t units, uint32_t *__restrict__ sink, uint32_t pace = 0) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
  u32x4 r[AHEAD][4];
  u32x4 acc = {0, 0, 0, 0};
  auto fetch = [&](uint64_t u, int slot) {
    const uint8_t *base = buf + u * 32768ull;
#pragma unroll
    for (int q = 0; q < 4; q++) r[slot][q] = ld<NT>(base + ((wave + 8u * (q >> 1)) * 64u + (q & 1) * 32u + (lane & 31u)) * 32u + (lane >> 5) * 16u);
  };
#pragma unroll
  for (int a = 0; a < AHEAD; a++) fetch(u0 + a < u1 ? u0 + a : u1 - 1, a);
  for (uint64_t u = u0; u < u1; u += AHEAD) {
#pragma unroll
    for (int a = 0; a < AHEAD; a++) {
#pragma unroll
      for (int q = 0; q < 4; q++) acc ^= r[a][q];
      const uint64_t nxt = u + a + AHEAD;
      fetch(nxt < u1 ? nxt : u1 - 1, a);
      for (uint32_t z = 0; z < pace; ++z) __builtin_amdgcn_s_sleep(1);      // PACED (round 6: the write side has a cliff in its pace; the read side?)
      if (BARRIER) __syncthreads();
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int AHEAD, bool NT, bool BARRIER>
static void run(const char *name, const uint8_t *d, uint64_t bytes, uint32_t *sink, int grid, uint32_t pace = 0) {
  const uint64_t units = bytes / 32768;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((read_kernel<AHEAD, NT, BARRIER>), dim3(grid), dim3(512), 0, 0, d, units, sink, pace);
  hipEventRecord(e0);
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL((read_kernel<AHEAD, NT, BARRIER>), dim3(grid), dim3(512), 0, 0, d, units, sink, pace);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s grid %4d pace %2u: %7.3f ms per pass = %6.2f TB/s\n", name, grid, pace, ms / 5, (double)units * 32768 / (ms / 5 * 1e-3) / 1e12);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const uint64_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 12ull) << 30;
  uint8_t *d = nullptr;
  uint32_t *sink = nullptr;
  if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(d, 1, bytes);
  hipDeviceSynchronize();
  printf("reading %llu GiB in 32 KiB units, 512-thread persistent workgroups\n", (unsigned long long)(bytes >> 30));
  for (int grid : {256, 512, 768, 1024}) {
    run<1, true, true>("1 unit ahead, nt loads, barrier per unit", d, bytes, sink, grid);
    run<2, true, true>("2 units ahead, nt loads, barrier per unit", d, bytes, sink, grid);
    run<2, false, true>("2 units ahead, plain loads, barrier per unit", d, bytes, sink, grid);
    run<2, true, false>("2 units ahead, nt loads, no barrier", d, bytes, sink, grid);
    run<3, true, true>("3 units ahead, nt loads, barrier per unit", d, bytes, sink, grid);
    run<4, true, false>("4 units ahead, nt loads, no barrier", d, bytes, sink, grid);
  }
  // round 6: the same readers PACED (s_sleep 1 per unit and wave): does a read stream have the cliff the fill-ordered stores have?
  if (argc > 2)
    for (int grid : {512, 768})
      for (uint32_t pace : {0u, 1u, 2u, 3u, 4u, 6u, 8u, 12u}) run<3, true, true>("3 units ahead, nt loads, barrier, PACED", d, bytes, sink, grid, pace);
  return 0;
}
