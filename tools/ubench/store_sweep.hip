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
extern "C" int b3w_place_alloc(int device, uint64_t bytes, int want_mixed, void **out, int *mixed, float *rates);
extern "C" int b3w_place_free(void *ptr);
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// pace: dependent vector-ALU instructions per 16-byte store (the witness kernel has ~6 and an LDS read): a store-only kernel without
// any issues its stores faster than the memory system drains them, and that is NOT the fastest way to fill HBM (round 6)
__device__ __forceinline__ void pace_valu(u32x4 &v, uint32_t k) {
  uint32_t x = v.z;
  for (uint32_t i = 0; i < k; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x));
  v.z = x;
}
template <int W>
__global__ __launch_bounds__(64) void k_streams(uint8_t *out, uint64_t pitch, uint32_t n, uint32_t tiles, uint32_t pace = 0) {
  const uint32_t b0 = blockIdx.x * W, lane = threadIdx.x;
  u32x4 v = {lane, blockIdx.x, 0, 0};
  uint8_t *base[W];
#pragma unroll
  for (int w = 0; w < W; ++w) base[w] = out + (uint64_t)(b0 + w < n ? b0 + w : n - 1) * pitch + lane * 16;
  for (uint32_t g = 0; g < tiles; ++g) {
#pragma unroll
    for (int w = 0; w < W; ++w) { pace_valu(v, pace); *reinterpret_cast<u32x4 *>(base[w] + (uint64_t)g * 1024) = v; }
  }
}
template <int W>
__global__ __launch_bounds__(64) void k_persist(uint8_t *out, uint64_t pitch, uint32_t n, uint32_t tiles, uint32_t pace = 0) {
  const uint32_t lane = threadIdx.x;
  u32x4 v = {lane, blockIdx.x, 0, 0};
  for (uint32_t b0 = blockIdx.x * W; b0 < n; b0 += gridDim.x * W) {
    uint8_t *base[W];
#pragma unroll
    for (int w = 0; w < W; ++w) base[w] = out + (uint64_t)(b0 + w < n ? b0 + w : n - 1) * pitch + lane * 16;
    for (uint32_t g = 0; g < tiles; ++g) {
#pragma unroll
      for (int w = 0; w < W; ++w) { pace_valu(v, pace); *reinterpret_cast<u32x4 *>(base[w] + (uint64_t)g * 1024) = v; }
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
__global__ __launch_bounds__(64) void k_chunk(uint8_t *out, uint64_t bytes, uint32_t kib, uint32_t pace) {
  const uint32_t lane = threadIdx.x, G = gridDim.x;
  const uint32_t b = XCD ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
  u32x4 v = {lane, blockIdx.x, 0, 0};
  const uint64_t chunk = (uint64_t)kib * 1024, nchunks = bytes / chunk;
  for (uint64_t c = b; c < nchunks; c += G) {
    uint8_t *p = out + c * chunk + lane * 16;
    for (uint32_t k = 0; k < kib; ++k) { pace_valu(v, pace); *reinterpret_cast<u32x4 *>(p + (uint64_t)k * 1024) = v; }
  }
}
// the fill shape with a paced multi-wave workgroup: G workgroups of T threads, tile = 16 T bytes
__global__ void k_fill_paced(uint8_t *out, uint64_t bytes, uint32_t pace) {
  u32x4 v = {threadIdx.x, blockIdx.x, 0, 0};
  const uint64_t tile = (uint64_t)blockDim.x * 16;
  for (uint64_t t = blockIdx.x; (t + 1) * tile <= bytes; t += gridDim.x) { pace_valu(v, pace); *reinterpret_cast<u32x4 *>(out + t * tile + threadIdx.x * 16) = v; }
}

// fill-shaped workgroups (4 waves, one 4 KiB tile per step) that STAY in one place for R steps instead of jumping 1 MiB every step:
//   A<G>x<R>  XCD-interleaved: the eight workgroups b, b+1, .. b+7 (one per XCD, round-robin dispatch) share a region of 8 R tiles, workgroup
//             b taking the tiles = b (mod 8) of it: every XCD keeps writing 4 KiB blocks of ITS residue class, like the fill shape
//   B<G>x<R>  plain chunks: workgroup b takes R consecutive tiles (a contiguous 4 R KiB chunk), then the chunk G further on
// (a witness kernel in this shape needs one TRACE per ~R tiles instead of one per tile: R = 24 is an eighth of a body)
template <bool XCD>
__global__ __launch_bounds__(256) void k_stay(uint8_t *out, uint64_t bytes, uint32_t R, uint32_t pace) {
  u32x4 v = {threadIdx.x, blockIdx.x, 0, 0};
  const uint64_t ntiles = bytes / 4096;
  const uint32_t G = gridDim.x, b = blockIdx.x;
  if (XCD) {
    const uint32_t x = b % 8, g = b / 8, groups = G / 8;
    for (uint64_t base = (uint64_t)g * 8 * R; base < ntiles; base += (uint64_t)groups * 8 * R)
      for (uint32_t r = 0; r < R; ++r) {
        const uint64_t t = base + 8ull * r + x;
        if (t < ntiles) { pace_valu(v, pace); *reinterpret_cast<u32x4 *>(out + t * 4096 + threadIdx.x * 16) = v; }
      }
  } else {
    for (uint64_t base = (uint64_t)b * R; base < ntiles; base += (uint64_t)G * R)
      for (uint32_t r = 0; r < R; ++r) {
        const uint64_t t = base + r;
        if (t < ntiles) { pace_valu(v, pace); *reinterpret_cast<u32x4 *>(out + t * 4096 + threadIdx.x * 16) = v; }
      }
  }
}
// the fill shape's addresses from single-wave workgroups: 1024 waves, wave i = 8 (4 r + j) + x stores KiB j of tile 8 r + x of every
// 1 MiB window (so XCD x — round-robin dispatch — writes the tiles = x (mod 8), as in F256x256)
__global__ __launch_bounds__(64) void k_mimic(uint8_t *out, uint64_t bytes, uint32_t pace) {
  u32x4 v = {threadIdx.x, blockIdx.x, 0, 0};
  const uint32_t i = blockIdx.x, x = i % 8, q = i / 8, j = q % 4, r = q / 4;
  const uint64_t nwin = bytes >> 20;
  for (uint64_t w = 0; w < nwin; ++w) { pace_valu(v, pace); *reinterpret_cast<u32x4 *>(out + (w << 20) + (uint64_t)(8 * r + x) * 4096 + j * 1024 + threadIdx.x * 16) = v; }
}

// G<m>: BODY-major fill: 256 workgroups of 4 waves in groups of 8 m; a group takes whole bodies in turn, its workgroup u (XCD u % 8) the
// body's absolute 4 KiB blocks = u (mod 8 m) — 256 / (8 m) bodies in flight chip-wide, each written through a contiguous, moving
// window of 8 m blocks; a witness kernel in this shape needs ONE trace per body and workgroup (G32 would be the fill shape itself)
__global__ __launch_bounds__(256) void k_bodyfill(uint8_t *out, uint64_t pitch, uint32_t n, uint32_t body, uint32_t m8, uint32_t pace) {
  u32x4 v = {threadIdx.x, blockIdx.x, 0, 0};
  const uint32_t u = blockIdx.x % m8, grp = blockIdx.x / m8, ngrp = gridDim.x / m8;
  const uint64_t base = reinterpret_cast<uint64_t>(out);
  for (uint32_t w = grp; w < n; w += ngrp) {
    const uint64_t b0 = base + (uint64_t)w * pitch, b1 = b0 + body;
    const uint64_t first = b0 >> 12, last = (b1 - 1) >> 12;
    for (uint64_t blk = first + ((u + m8 - (uint32_t)(first % m8)) % m8); blk <= last; blk += m8) {
      const uint64_t a = (blk << 12) + threadIdx.x * 16;
      pace_valu(v, pace);
      if (a >= b0 && a + 16 <= b1) *reinterpret_cast<u32x4 *>(a) = v;
    }
  }
}

struct Shape { char name[24]; int kind, a, b, pace; };

static double run(const Shape &s, uint8_t *buf, uint32_t n, uint64_t pitch, uint32_t body, int iters, hipStream_t st) {
  const uint32_t tiles = body / 1024;
  const uint64_t bytes = (uint64_t)n * pitch;
  uint64_t per_pass = (uint64_t)n * tiles * 1024;
  auto launch = [&] {
    switch (s.kind) {
      case 0:
        switch (s.a) {
          case 1: hipLaunchKernelGGL(k_streams<1>, dim3(n), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.pace); break;
          case 2: hipLaunchKernelGGL(k_streams<2>, dim3((n + 1) / 2), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.pace); break;
          case 4: hipLaunchKernelGGL(k_streams<4>, dim3((n + 3) / 4), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.pace); break;
          case 8: hipLaunchKernelGGL(k_streams<8>, dim3((n + 7) / 8), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.pace); break;
          default: hipLaunchKernelGGL(k_streams<16>, dim3((n + 15) / 16), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.pace); break;
        }
        break;
      case 1:
        switch (s.a) {
          case 1: hipLaunchKernelGGL(k_persist<1>, dim3(s.b), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.pace); break;
          case 2: hipLaunchKernelGGL(k_persist<2>, dim3(s.b), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.pace); break;
          case 4: hipLaunchKernelGGL(k_persist<4>, dim3(s.b), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.pace); break;
          default: hipLaunchKernelGGL(k_persist<8>, dim3(s.b), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.pace); break;
        }
        break;
      case 2: if (s.pace) hipLaunchKernelGGL(k_fill_paced, dim3(s.a), dim3(s.b), 0, st, buf, bytes, (uint32_t)s.pace); else hipLaunchKernelGGL(k_fill, dim3(s.a), dim3(s.b), 0, st, buf, bytes); break;
      case 3: hipLaunchKernelGGL(k_sliced, dim3(n * (uint32_t)s.a), dim3(64), 0, st, buf, pitch, n, tiles, (uint32_t)s.a); break;
      case 4: hipLaunchKernelGGL(k_chunk<false>, dim3(s.a), dim3(64), 0, st, buf, bytes, (uint32_t)s.b, (uint32_t)s.pace); break;
      case 5: hipLaunchKernelGGL(k_chunk<true>, dim3(s.a), dim3(64), 0, st, buf, bytes, (uint32_t)s.b, (uint32_t)s.pace); break;
      case 6: hipLaunchKernelGGL(k_stay<true>, dim3(s.a), dim3(256), 0, st, buf, bytes, (uint32_t)s.b, (uint32_t)s.pace); break;
      case 7: hipLaunchKernelGGL(k_stay<false>, dim3(s.a), dim3(256), 0, st, buf, bytes, (uint32_t)s.b, (uint32_t)s.pace); break;
      case 8: hipLaunchKernelGGL(k_mimic, dim3(1024), dim3(64), 0, st, buf, bytes, (uint32_t)s.pace); break;
      default: hipLaunchKernelGGL(k_bodyfill, dim3(s.b), dim3(256), 0, st, buf, pitch, n, body, (uint32_t)(8 * s.a), (uint32_t)s.pace); break;
    }
  };
  if (s.kind == 2) per_pass = bytes / ((uint64_t)s.b * 16) * ((uint64_t)s.b * 16);
  if (s.kind == 4 || s.kind == 5) per_pass = bytes / ((uint64_t)s.b * 1024) * ((uint64_t)s.b * 1024);
  if (s.kind == 6 || s.kind == 7) per_pass = bytes / 4096 * 4096;
  if (s.kind == 8) per_pass = (bytes >> 20) << 20;
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
  auto add = [&](const char *fmt, int kind, int a, int b, int pace = 0) {
    Shape s; snprintf(s.name, sizeof s.name, fmt, a, b); s.kind = kind; s.a = a; s.b = b; s.pace = pace;
    if (pace) snprintf(s.name + strlen(s.name), sizeof s.name - strlen(s.name), "v%d", pace);
    shapes.push_back(s);
  };
  const bool big = n > 16384;
  for (int w : {1, 2, 4, 8, 16}) add("S%d", 0, w, 0);
  for (int w : {4, 8}) for (int pc : {4, 8, 16, 32}) add("S%d", 0, w, 0, pc);
  for (int w : {4, 8}) for (int g : {256, 512, 1024, 2048}) if (!quick || g >= 512) add("P%dx%d", 1, w, g);
  for (int w : {4, 8}) for (int g : {512, 1024}) for (int pc : {4, 8, 16}) add("P%dx%d", 1, w, g, pc);
  for (int g : {256, 512, 1024, 2048}) for (int t : {256, 512, 1024}) if (!quick || t == 256) add("F%dx%d", 2, g, t);
  for (int pc : {4, 8, 16, 32}) add("F%dx%d", 2, 256, 256, pc);
  for (int s : {4, 8, 16, 32, 64}) if (!big) add("L%d", 3, s, 0);
  for (int g : {512, 1024, 2048, 4096, 8192}) for (int k : {1, 4, 16, 48, 192}) if (!quick || (k != 1 && k != 192)) add("C%dx%d", 4, g, k);
  for (int g : {2048, 8192}) for (int k : {16, 48}) for (int pc : {4, 8, 16}) add("C%dx%d", 4, g, k, pc);
  for (int g : {2048, 4096}) for (int k : {4, 16, 48}) add("X%dx%d", 5, g, k);
  for (int g : {256, 512}) for (int r : {1, 4, 8, 24, 47, 94, 188}) add("A%dx%d", 6, g, r);
  for (int r : {8, 24}) for (int pc : {4, 8}) add("A%dx%d", 6, 256, r, pc);
  for (int g : {256, 512}) for (int r : {1, 4, 8, 24, 47, 188}) add("B%dx%d", 7, g, r);
  add("M1024", 8, 0, 0); add("M1024", 8, 0, 0, 4);
  for (int m : {1, 2, 4, 8, 16, 32}) add("G%dx%d", 9, m, 256);
  for (int m : {2, 4, 8}) for (int pc : {2, 4, 8}) add("G%dx%d", 9, m, 256, pc);
  for (int m : {2, 4, 8}) add("G%dx%d", 9, m, 512);
  if (getenv("SWEEP_ONLY")) {                     // e.g. SWEEP_ONLY=AFM: the shapes whose name starts with one of these letters
    std::vector<Shape> keep;
    for (const Shape &s : shapes) if (strchr(getenv("SWEEP_ONLY"), s.name[0])) keep.push_back(s);
    shapes.swap(keep);
  }
  b3w_ctx *ctx = nullptr;
  if (b3w_create(0 /* compression */, 0, &ctx)) { printf("b3w_create failed\n"); return 1; }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  void *placed = nullptr, *plain = nullptr, *single = nullptr;
  int32_t label = -1;
  int one = 0;
  if (b3w_bodies_alloc(ctx, bytes, &placed, &label)) { printf("b3w_bodies_alloc failed\n"); return 1; }
  CK(hipMalloc(&plain, bytes));
  // a ONE-class buffer (b3w_place_alloc mode 2): what a plain hipMalloc is on an unlucky day, made on purpose
  if (b3w_place_alloc(0, bytes, 2, &single, &one, nullptr) != 0) single = nullptr;
  printf("store_sweep: n %u pitch %llu (%.2f GB), median of %d passes, TB/s; placed buffer label %d (1 = mixed, 2 = interleaved, 0 = plain)\n",
         n, (unsigned long long)pitch, bytes / 1e9, iters, label);
  printf("%-14s %8s %8s %8s\n", "shape", "hipMalloc", "1-class", "placed");
  double best[3] = {0, 0, 0};
  const char *bestn[3] = {"", "", ""};
  for (const Shape &s : shapes) {
    const double r[3] = {run(s, (uint8_t *)plain, n, pitch, body, iters, st), single ? run(s, (uint8_t *)single, n, pitch, body, iters, st) : 0.0,
                         run(s, (uint8_t *)placed, n, pitch, body, iters, st)};
    printf("%-14s %8.3f %8.3f %8.3f\n", s.name, r[0], r[1], r[2]);
    fflush(stdout);
    for (int i = 0; i < 3; i++) if (r[i] > best[i]) { best[i] = r[i]; bestn[i] = s.name; }
  }
  printf("best: hipMalloc %s %.3f TB/s (%.3f of 8), one class %s %.3f (%.3f), placed %s %.3f (%.3f)\n", bestn[0], best[0], best[0] / 8, bestn[1], best[1], best[1] / 8,
         bestn[2], best[2], best[2] / 8);
  if (single) b3w_place_free(single);
  CK(hipFree(plain));
  b3w_bodies_free(ctx, placed);
  b3w_destroy(ctx);
  return 0;
}
