// tilefill.hip — can a fill-shaped store loop (256 workgroups, workgroup b owns the 4 KiB tiles b + 256k,
// 4 waves x 16 B per lane per tile) carry the witness work itself?  Per tile every lane reads a 16-bit slot
// table entry and an image word from LDS, shapes 16 bytes, stores them; every STEP tiles one wave (rotating)
// recomputes STEP images (7 rounds of half-G's on a quad, 56 ds_write_b128 per lane) into the other LDS
// buffer, then the workgroup synchronises.  No correctness, only the instruction mix and the rate.
// build: hipcc --offload-arch=gfx950 -O3 -o tilefill tilefill.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NW = 24093, IMG = 944, TABW = 12048;
__device__ __forceinline__ uint32_t rotr32(uint32_t x, int r) { return __builtin_rotateright32(x, r); }

__device__ __forceinline__ void half_g(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, uint32_t m, uint32_t *Lk, int R1, int R2) {
  const uint64_t s1 = (uint64_t)a + b + m;
  const uint32_t A = (uint32_t)s1, D2 = rotr32(d ^ A, R1);
  const uint64_t s3 = (uint64_t)c + D2;
  const uint32_t C = (uint32_t)s3, B4 = rotr32(b ^ C, R2);
  *reinterpret_cast<uint4 *>(Lk) = make_uint4(A, (uint32_t)(s1 >> 32), C, (uint32_t)(s3 >> 32));
  *reinterpret_cast<uint4 *>(Lk + 4) = make_uint4(D2, d, B4, b);
  a = A; d = D2; c = C; b = B4;
}

// MODE bit0: producer on; bit1: barrier per step on; bit2: table+image reads on
template <int STEP, int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_tile(uint8_t *out, uint32_t ntiles) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint16_t *tab = reinterpret_cast<uint16_t *>(lds);
  uint32_t *img = lds + TABW;                                      // [2][STEP][IMG]
  const uint32_t tid = threadIdx.x, wave = tid / 64, lane = tid % 64;
  for (uint32_t i = tid; i < NW; i += 64 * WAVES) tab[i] = (uint16_t)(((i / 32) * 5 % IMG) | ((i % 32) << 10));
  for (uint32_t i = tid; i < 2 * STEP * IMG; i += 64 * WAVES) img[i] = i * 2654435761u;
  __syncthreads();
  constexpr uint32_t BPL = 4096 / (64 * WAVES);                     // bytes per lane per tile (16 or 32)
  for (uint32_t step = 0;; ++step) {
    const uint32_t k0 = step * STEP;
    if (blockIdx.x + 256u * k0 >= ntiles) break;
    uint32_t *cur = img + (step & 1) * STEP * IMG, *nxt = img + ((step + 1) & 1) * STEP * IMG;
    if ((MODE & 1) && wave == (step % WAVES)) {
      const uint32_t q = lane >> 2, col = lane & 3;
      if (q < STEP) {
        uint32_t *L = nxt + q * IMG;
        uint32_t a = L[1 + col] + step, b = L[5 + col], c = 0x6A09E667u + col, d = L[25 + col];
        for (int r = 0; r < 7; ++r) {
          uint32_t *Lk = L + 45 + 16 * (r * 8 + col);
          half_g(a, b, c, d, L[9 + ((r * 3 + col * 2) & 15)], Lk, 16, 12);
          half_g(a, b, c, d, L[9 + ((r * 5 + col * 2 + 1) & 15)], Lk + 8, 8, 7);
          b = __shfl_xor(b, 1); c = __shfl_xor(c, 2); d = __shfl_xor(d, 3);
          uint32_t *Lj = L + 45 + 16 * (r * 8 + 4 + col);
          half_g(a, b, c, d, L[9 + ((r * 7 + col * 2) & 15)], Lj, 16, 12);
          half_g(a, b, c, d, L[9 + ((r * 11 + col * 2 + 1) & 15)], Lj + 8, 8, 7);
          b = __shfl_xor(b, 3); c = __shfl_xor(c, 2); d = __shfl_xor(d, 1);
        }
        L[29 + col] = a ^ c;
      }
    }
#pragma unroll
    for (int k = 0; k < STEP; ++k) {
      const uint32_t t = blockIdx.x + 256u * (k0 + k);
      if (t < ntiles) {
        uint32_t slot = (t * 128u) % NW + (tid * BPL) / 32;
        if (slot >= NW) slot -= NW;
        u32x4 v = {1, 0, 0, 0};
        if (MODE & 4) {
          const uint32_t e = tab[slot];
          const uint32_t *L = cur + k * IMG + (e & 1023u);
          v.x = (L[0] >> ((e >> 10) & 31u)) & 1u;
          v.y = L[1] & ((e >> 15) ? 0xFFFFFFFFu : 0u);
        }
        uint8_t *p = out + (uint64_t)t * 4096 + tid * BPL;
        if (BPL == 16) { if (lane & 1) v.x = v.y = 0; *reinterpret_cast<u32x4 *>(p) = v; }
        else { u32x4 z = {0, 0, 0, 0}; *reinterpret_cast<u32x4 *>(p) = v; *reinterpret_cast<u32x4 *>(p + 16) = z; }
      }
    }
    if (MODE & 2) __syncthreads();
  }
}

static hipEvent_t e0, e1;
template <class F>
static double timeit(F launch, int it = 20) {
  for (int i = 0; i < 3; i++) launch();
  hipEventRecord(e0, 0);
  for (int i = 0; i < it; i++) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / it;
}

template <int STEP, int MODE, int WAVES>
static int run(uint8_t *buf, uint64_t bytes, int bi) {
  const size_t smem = (size_t)(TABW + 2 * STEP * IMG) * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile<STEP, MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const uint32_t ntiles = (uint32_t)(bytes / 4096);
  const double ms = timeit([&] { hipLaunchKernelGGL((k_tile<STEP, MODE, WAVES>), dim3(256), dim3(64 * WAVES), smem, 0, buf, ntiles); });
  CK(hipGetLastError());
  printf("buf%d step=%2d waves=%d producer=%d barrier=%d ldsreads=%d   %7.3f ms %7.0f GB/s\n", bi, STEP, WAVES, MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, ms,
         bytes / 1e9 / (ms * 1e-3));
  fflush(stdout);
  return 0;
}

int main() {
  const uint64_t bytes = 4096ull * 770976;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint8_t *bufs[2];
  for (int b = 0; b < 2; b++) CK(hipMalloc((void **)&bufs[b], bytes + (1 << 22)));
  for (int b = 0; b < 2; b++) {
    uint8_t *buf = bufs[b];
    run<12, 0, 4>(buf, bytes, b);
    run<12, 4, 4>(buf, bytes, b);
    run<12, 6, 4>(buf, bytes, b);
    run<12, 7, 4>(buf, bytes, b);
    run<8, 7, 4>(buf, bytes, b);
    run<4, 7, 4>(buf, bytes, b);
    run<12, 0, 2>(buf, bytes, b);
    run<12, 4, 2>(buf, bytes, b);
    run<12, 6, 2>(buf, bytes, b);
    run<12, 7, 2>(buf, bytes, b);
    run<8, 7, 2>(buf, bytes, b);
  }
  return 0;
}
