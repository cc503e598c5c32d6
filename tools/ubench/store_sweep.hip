// store_sweep.hip — what does this chip's HBM take from STORES, by shape?  Store-only kernels (16 bytes per lane, whole
// 128-byte lines, no loads, no LDS) over the same body buffer in every shape a witness kernel could write in, on a plain
// hipMalloc buffer and on a placed one (b3w_bodies_alloc) — the sweep behind roofline.store_ceiling (round 6: the r05 probe
// shapes were slower than the witness kernel they were meant to bound).
//
//   S<W>        body streams: one wave per W bodies, 1 KiB per body and step (the fused kernels' EXPAND shape)
//   P<W>x<G>    the same with a persistent grid: G single-wave workgroups take groups of W bodies in turn
//   F<G>x<T>    fill: G workgroups of T threads, tile = 16 T bytes, tile t of workgroup b = b + G k (the runtime's fill: F256x256)
//   L<s>        sliced: wave (body, slice) stores 1/s of a body's 1 KiB tiles, consecutive waves = consecutive slices of a body
//   C<G>x<K>    chunked fill: G single-wave workgroups, wave b stores the K KiB chunks b, b + G, ... of the linear buffer
//   X<G>x<K>    the same, XCD-aware: workgroup b -> chunk sequence so that the 8 XCDs (round-robin dispatch) share a window
//
// build: hipcc --offload-arch=gfx950 -O3 -o store_sweep store_sweep.hip -L../../hot-proofs-blake3-circom_amd -lb3wit -Wl,-rpath,'$ORIGIN/../../hot-proofs-blake3-circom_amd'
// run:   ./store_sweep [n=4096] [pitch=770976] [iters=6]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../include/b3wit.h"
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W>
__global__ __launch_bounds__(64) void k_streams(uint8_t *out, uint64_t pitch, uint32_t n, uint32_t tiles) {
  const uint32_t b0 = blockIdx.x * W, lane = threadIdx.x;
  const u32x4 v = {lane, blockIdx.x, 0, 0};
  uint8_t *base[W];
#pragma unroll
  for (int w = 0; w < W; ++w) base[w] = out + (uint64_t)(b0 + w < n ? b0 + w : n - 1) * pitch + lane * 16;
  for (uint32_t g = 0; g < tiles; ++g) {
#pragma unroll
    for (int w = 0; w < W; ++w) *reinterpret_cast<u32x4 *>(base[w] + (uint64_t)g * 1024) = v;
  }
}
template <int W>
__global__ __launch_bounds__(64) void k_persist(uint8_t *out, uint64_t pitch, uint32_t n, uint32_t tiles) {
  const uint32_t lane = threadIdx.x;
  const u32x4 v = {lane, blockIdx.x, 0, 0};
  for (uint32_t b0 = blockIdx.x * W; b0 < n; b0 += gridDim.x * W) {
    uint8_t *base[W];
#pragma unroll
    for (int w = 0; w < W; ++w) base[w] = out + (uint64_t)(b0 + w < n ? b0 + w : n - 1) * pitch + lane * 16;
    for (uint32_t g = 0; g < tiles; ++g) {
#pragma unroll
      for (int w = 0; w < W; ++w) *reinterpret_cast<u32x4 *>(base[w] + (uint64_t)g * 1024) = v;
    }
  }
}
__global__ void k_fill(uint8_t *out, uint64_t bytes) {
  const u32x4 v = {threadIdx.x, blockIdx.x, 0, 0};
  const uint64_t tile = (uint64_t)blockDim.x * 16;
  for (uint64_t t = blockIdx.x; (t + 1) * tile <= bytes; t += gridDim.x) *reinterpret_cast<u32x4 *>(out + t * tile + threadIdx.x * 16) = v;
}
__global__ __launch_bounds__(64) void k_sliced(uint8_t *out, uint64_t pitch, uint32_t n, uint32_t tiles, uint32_t slices) {
  const uint32_t body = blockIdx.x / slices, slice = blockIdx.x % slices, lane = threadIdx.x;
  const uint32_t per = ((tiles + slices - 1) / slices + 3) & ~3u;
  const uint32_t k0 = slice * per < tiles ? slice * per : tiles, k1 = k0 + per < tiles ? k0 + per : tiles;
  const u32x4 v = {lane, blockIdx.x, 0, 0};
  uint8_t *base = out + (uint64_t)body * pitch + lane * 16;
  for (uint32_t k = k0; k < k1; ++k) *reinterpret_cast<u32x4 *>(base + (uint64_t)k * 1024) = v;
}
// chunked fill; XCD: workgroup b runs on XCD b % 8 — chunk order (b % 8) + 8 * (b / 8) is the identity, so "XCD-aware" here means the
// opposite deal: XCD x owns the x-th EIGHTH of every window of G chunks (its L2 sees one contiguous piece)
template <bool XCD>
__global__ __launch_bounds__(64) void k_chunk(uint8_t *out, uint64_t bytes, uint32_t kib) {
  const uint32_t lane = threadIdx.x, G = gridDim.x;
  const uint32_t b = XCD ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
  const u32x4 v = {lane, blockIdx.x, 0, 0};
  const uint64_t chunk = (uint64_t)kib * 1024, nchunks = bytes / chunk;
  for (uint64_t c = b; c < nchunks; c += G) {
    uint8_t *p = out + c * chunk + lane * 16;
    for (uint32_t k = 0; k < kib; ++k) *reinterpret_cast<u32x4 *>(p + (uint64_t)k * 1024) = v;
  }
}

struct Shape { char name[24]; int kind, a, b; };

static double run(const Shape &s, uint8_t *buf, uint32_t n, uint64_t pitch, uint32_t body, int iters, hipStream_t st) {
  const uint32_t tiles = body / 1024;
  const uint64_t bytes = (uint64_t)n * pitch;
  uint64_t per_pass = (uint64_t)n * tiles * 1024;
  auto launch = [&] {
    switch (s.kind) {
      case 0:
        switch (s.a) {
          case 1: hipLaunchKernelGGL(k_streams<1>, dim3(n), dim3(64), 0, st, buf, pitch, n, tiles); break;
          case 2: hipLaunchKernelGGL(k_streams<2>, dim3((n + 1) / 2), dim3(64), 0, st, buf, pitch, n, tiles); break;
          case 4: hipLaunchKernelGGL(k_streams<4>, dim3((n + 3) / 4), dim3(64), 0, st, buf, pitch, n, tiles); break;
          case 8: hipLaunchKernelGGL(k_streams<8>, dim3((n + 7) / 8), dim3(64), 0, st, buf, pitch, n, tiles); break;
          default: hipLaunchKernelGGL(k_streams<16>, dim3((n + 15) / 16), dim3(64), 0, st, buf, pitch, n, tiles); break;
        }
        break;
      case 1:
        switch (s.a) {
          case 1: hipLaunchKernelGGL(k_persist<1>, dim3(s.b), dim3(64), 0, st, buf, pitch, n, tiles); break;
          case 2: hipLaunchKernelGGL(k_persist<2>, dim3(s.b), dim3(64), 0, st, buf, pitch, n, tiles); break;
          case 4: hipLaunchKernelGGL(k_persist<4>, dim3(s.b), dim3(64), 0, st, buf, pitch, n, tiles); break;
          default: hipLaunchKernelGGL(k_persist<8>, dim3(s.b), dim3(64), 0, st, buf, pitch, n, tiles); break;
        }
        break;
      case 2: hipLaunchKernelGGL(k_fill, dim3(s.a), dim3(s.b), 0, st, buf, bytes); break;
      case 3: hipLaunchKernelGGL(k_sliced, dim3(n * (uint32_t)s.a), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.a); break;
      case 4: hipLaunchKernelGGL(k_chunk<false>, dim3(s.a), dim3(64), 0, st, buf, bytes, (uint32_t)s.b); break;
      default: hipLaunchKernelGGL(k_chunk<true>, dim3(s.a), dim3(64), 0, st, buf, bytes, (uint32_t)s.b); break;
    }
  };
  if (s.kind == 2) per_pass = bytes / ((uint64_t)s.b * 16) * ((uint64_t)s.b * 16);
  if (s.kind >= 4) per_pass = bytes / ((uint64_t)s.b * 1024) * ((uint64_t)s.b * 1024);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  std::vector<float> ms(iters);
  for (int i = 0; i < iters; i++) {
    CK(hipEventRecord(e0, st)); launch(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms[i], e0, e1));
  }
  CK(hipGetLastError());
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  std::sort(ms.begin(), ms.end());
  return (double)per_pass / 1e9 / ms[iters / 2];          // TB/s by the median pass
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoi(argv[1]) : 4096;
  const uint64_t pitch = argc > 2 ? (uint64_t)atoll(argv[2]) : 770976;
  const int iters = argc > 3 ? atoi(argv[3]) : 6;
  const bool quick = argc > 4 && !strcmp(argv[4], "quick");
  const uint32_t body = (uint32_t)pitch;
  const uint64_t bytes = (uint64_t)n * pitch;
  std::vector<Shape> shapes;
  auto add = [&](const char *fmt, int kind, int a, int b) { Shape s; snprintf(s.name, sizeof s.name, fmt, a, b); s.kind = kind; s.a = a; s.b = b; shapes.push_back(s); };
  for (int w : {1, 2, 4, 8, 16}) add("S%d", 0, w, 0);
  for (int w : {4, 8}) for (int g : {256, 512, 1024, 2048}) if (!quick || g >= 512) add("P%dx%d", 1, w, g);
  for (int g : {256, 512, 1024, 2048}) for (int t : {256, 512, 1024}) if (!quick || t == 256) add("F%dx%d", 2, g, t);
  for (int s : {4, 8, 16, 32, 64}) add("L%d", 3, s, 0);
  for (int g : {512, 1024, 2048, 4096, 8192}) for (int k : {1, 4, 16, 48, 192}) if (!quick || (k != 1 && k != 192)) add("C%dx%d", 4, g, k);
  for (int g : {2048, 4096}) for (int k : {4, 16, 48}) add("X%dx%d", 5, g, k);
  b3w_ctx *ctx = nullptr;
  if (b3w_create(0 /* compression */, 0, &ctx)) { printf("b3w_create failed\n"); return 1; }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  void *placed = nullptr, *plain = nullptr;
  int32_t label = -1;
  if (b3w_bodies_alloc(ctx, bytes, &placed, &label)) { printf("b3w_bodies_alloc failed\n"); return 1; }
  CK(hipMalloc(&plain, bytes));
  printf("store_sweep: n %u pitch %llu (%.2f GB), median of %d passes, TB/s; placed buffer label %d (1 = mixed, 2 = interleaved, 0 = plain)\n",
         n, (unsigned long long)pitch, bytes / 1e9, iters, label);
  printf("%-12s %8s %8s\n", "shape", "plain", "placed");
  double best[2] = {0, 0};
  const char *bestn[2] = {"", ""};
  for (const Shape &s : shapes) {
    const double a = run(s, (uint8_t *)plain, n, pitch, body, iters, st), b = run(s, (uint8_t *)placed, n, pitch, body, iters, st);
    printf("%-12s %8.3f %8.3f\n", s.name, a, b);
    fflush(stdout);
    if (a > best[0]) { best[0] = a; bestn[0] = s.name; }
    if (b > best[1]) { best[1] = b; bestn[1] = s.name; }
  }
  printf("best: plain %s %.3f TB/s (%.3f of 8), placed %s %.3f TB/s (%.3f of 8)\n", bestn[0], best[0], best[0] / 8, bestn[1], best[1], best[1] / 8);
  CK(hipFree(plain));
  b3w_bodies_free(ctx, placed);
  b3w_destroy(ctx);
  return 0;
}
