// b3w_placement.hip — where witness bodies live in HBM.
//
// Measured on MI355X (tools/ubench/placement_map*.hip, vmm_classes.hip, pair_matrix.hip; logs under
// profiles/r01/placement/): the 288 GB of HBM3E fall into three "classes" of physical memory (about 96 GB each, in
// physical segments of 8-64 GiB; presumably the three ranks of the 12-high stacks).  A store pattern made of
// thousands of independent streams — one witness body per stream, what the witness kernels emit — runs at about
// 5.4 TB/s while all streams land in ONE class and at about 7.0 TB/s when they are split over two; a plain
// hipMalloc buffer almost always sits in one.  "Slow together" is an equivalence relation on fresh memory
// (pair_matrix.log: clean blocks).
//
// Physical memory cannot be chosen through HIP, but it can be classified.  Per device this file keeps a small
// POOL of 256 MiB physical handles (HIP virtual-memory API) with a label each:
//   * two reference handles stay allocated and mapped for the life of the process: h0 = the first handle ever
//     created, f0 = the first later handle that a timed split-store probe shows to be FAST together with h0;
//   * every other handle is probed against both (0.3 ms each): slow with h0 / fast with f0 -> label A (h0's class),
//     fast / slow -> B (f0's class), fast / fast -> C (the third class), slow / slow -> M ("mixed": the driver
//     assembled the handle from fragments of two classes, which happens once memory has been freed and reused;
//     such handles are not used and are given back at the end of the search).
// b3w_place_alloc takes labelled handles from the pool (creating and labelling new ones while it lacks them — the
// driver hands out fresh memory linearly, so a search walks 15-30 GiB on average until both sides are covered),
// maps them ALTERNATELY (A or C, then B or C, ...) into one contiguous virtual range, checks every seam of that
// order with the same probe and repairs slow seams from the spare handles.  Bodies written in natural order then
// always straddle two classes; nothing changes for the kernels or for consumers of the buffer (one linear device
// range).  Handles left over from a search and the handles of a freed buffer go back to the pool (at most 12 GiB per label), so a
// second buffer continues on the same clean, already labelled memory instead of on recycled fragments.
// When no second class shows up (other hardware, memory nearly full) the buffer is simply handles in creation order
// — correct, only slower — and the placement is reported as "plain".
//
// VIRTUAL ADDRESSES ARE NEVER REUSED.  On this ROCm stack a virtual address that has been unmapped and is mapped to
// another physical handle keeps serving the OLD pages (tools/ubench/vmm_remap.hip: writes through the new mapping
// land in the previous handle, also after hipDeviceSynchronize and after hipMemAddressFree + a new reservation that
// returns the same range).  So each device gets one 32 TiB reservation (instant, costs nothing) from which every
// mapping takes the next free 256 MiB slots; nothing in it is ever mapped twice and it is never freed.  A handle is
// mapped once for probing when it is created and stays there until it is released (a second, simultaneous mapping
// in a buffer is fine).  A buffer of B bytes therefore uses up B bytes of address space for good: thousands of
// 12 GB buffers per process; when the range is used up allocations fall back to plain hipMalloc.
#include <hip/hip_runtime.h>
#include "b3w_capture.h"
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <chrono>
#include <mutex>

#include "b3w_kernels.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// probe: stream i writes PROBE_PITCH consecutive bytes, even streams from lo, odd ones from hi, one wave per stream
__global__ __launch_bounds__(64) void b3w_store_probe_kernel(uint8_t *lo, uint8_t *hi, uint64_t pitch, uint32_t groups) {
  const uint32_t i = blockIdx.x, lane = threadIdx.x;
  const u32x4 v = {0, 0, 0, 0};
  uint8_t *base = ((i & 1) ? hi : lo) + (uint64_t)(i >> 1) * pitch + lane * 16;
  for (uint32_t g = 0; g < groups; ++g) *reinterpret_cast<u32x4 *>(base + (uint64_t)g * 1024) = v;
}

// Pure-store ceilings of a body buffer (b3w_place_store_rate): what the memory system takes from stores alone, in the two shapes the
// witness kernels write in.  STREAMS: one wave per W bodies, 1 KiB per body and step, as the fused kernels' EXPAND phase (no trace, no
// slot table, no LDS: the stores and their addresses only).  FILL: 256 workgroups of 256 threads walking 4 KiB tiles b, b + 256, ...
// (the runtime's own fill shape, what the sweep kernels imitate).
template <int W>
__global__ __launch_bounds__(64) void b3w_store_streams_kernel(uint8_t *out, uint64_t pitch, uint32_t n, uint32_t tiles, uint32_t pause = 0) {
  const uint32_t b0 = blockIdx.x * W, lane = threadIdx.x;
  const u32x4 v = {lane, blockIdx.x, 0, 0};
  uint8_t *base[W];
#pragma unroll
  for (int w = 0; w < W; ++w) base[w] = out + (uint64_t)(b0 + w < n ? b0 + w : n - 1) * pitch + lane * 16;
  for (uint32_t g = 0; g < tiles; ++g) {
#pragma unroll
    for (int w = 0; w < W; ++w) *reinterpret_cast<u32x4 *>(base[w] + (uint64_t)g * 1024) = v;
    // (measurements: a writer that leaves the memory system a little air — `pause` x 64 cycles of s_sleep per W KiB stored)
    for (uint32_t k = 0; k < pause; ++k) __builtin_amdgcn_s_sleep(1);
  }
}
// a writer with a FIXED small footprint: `gridDim.x` single-wave workgroups (a few per CU) that take groups of W bodies in turn.
// `pace` = dependent vector-ALU instructions in front of every 16-byte store: a store-only kernel without any issues its stores faster than
// the memory system drains them, and that is NOT the fastest way to fill HBM (round 6, tools/ubench/store_sweep.hip: 512 waves x 4 bodies
// at pace 4 store 7.29 TB/s into a placed buffer, the same without pacing 6.96 — and the witness kernel, which has ~6 instructions and an
// LDS read per store by itself, 7.18: the r05 probes were SLOWER than the kernel they were meant to bound)
__device__ __forceinline__ void store_pace(u32x4 &v, uint32_t k) {
  uint32_t x = v.z;
  for (uint32_t i = 0; i < k; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x));
  v.z = x;
}
template <int W>
__global__ __launch_bounds__(64) void b3w_store_persistent_kernel(uint8_t *out, uint64_t pitch, uint32_t n, uint32_t tiles, uint32_t pace = 0) {
  const uint32_t lane = threadIdx.x;
  u32x4 v = {lane, blockIdx.x, 0, 0};
  for (uint32_t b0 = blockIdx.x * W; b0 < n; b0 += gridDim.x * W) {
    uint8_t *base[W];
#pragma unroll
    for (int w = 0; w < W; ++w) base[w] = out + (uint64_t)(b0 + w < n ? b0 + w : n - 1) * pitch + lane * 16;
    for (uint32_t g = 0; g < tiles; ++g) {
#pragma unroll
      for (int w = 0; w < W; ++w) { store_pace(v, pace); *reinterpret_cast<u32x4 *>(base[w] + (uint64_t)g * 1024) = v; }
    }
  }
}
// A writer that takes as little of a SIMD's register file as a wave can: one body stream per wave, 6 VGPRs used (the allocation granule is 8),
// on a persistent grid.  CLOB > 0 makes the SAME code ALLOCATE CLOB + 1 registers (an empty asm that names v<CLOB> as clobbered): what a
// neighbour's registers cost the kernel beside it (round 6: does a commit kernel of 3 x 168 registers a SIMD lose a whole wave to a writer
// that needs 16?  tools/ubench/overlap_commit_probe.py)
template <int CLOB>
__global__ __launch_bounds__(64) void b3w_store_tiny_kernel(uint8_t *out, uint64_t pitch, uint32_t n, uint32_t tiles) {
  if (CLOB == 15) asm volatile("" ::: "v15");
  if (CLOB == 31) asm volatile("" ::: "v31");
  if (CLOB == 63) asm volatile("" ::: "v63");
  const u32x4 v = {0, 0, 0, 0};
  for (uint32_t b = blockIdx.x; b < n; b += gridDim.x) {
    uint8_t *p = out + (uint64_t)b * pitch + threadIdx.x * 16;
    for (uint32_t g = 0; g < tiles; ++g, p += 1024) *reinterpret_cast<u32x4 *>(p) = v;
  }
}
__global__ __launch_bounds__(256) void b3w_store_fill_kernel(uint8_t *out, uint64_t bytes) {
  const u32x4 v = {threadIdx.x, blockIdx.x, 0, 0};
  for (uint64_t t = blockIdx.x; (t + 1) * 4096 <= bytes; t += gridDim.x) *reinterpret_cast<u32x4 *>(out + t * 4096 + threadIdx.x * 16) = v;
}

// The fill-ordered witness kernel's store order with nothing else (b3w_regionfill_kernel): absolute 128 KiB regions of 32 blocks, 32 groups
// of eight 4-wave workgroups (one per XCD) on 32 consecutive regions, workgroup x storing the blocks x, x + 8, x + 16, x + 24 of its region —
// one contiguous 4 MiB window chip-wide — and every wave waiting behind every store (that kernel's storing waves have ~300 clocks of table
// and image reads per store).  The rate is a cliff in the pace: a wave that stores faster than its share of the memory drains is held up at
// random, the 256 workgroups drift and the window frays (6.0-6.5 TB/s); paced just below that it runs at 6.95-7.27 TB/s on ANY buffer, placed
// or not (tools/ubench/store_region_scan.py, profiles/r06/store_region_scan*.log).
// PACE is compiled in (straight-line code: a run-time pacing loop or switch costs more than the steps it is to provide):
// PACE / 16 x `s_sleep 1` (64 clocks each) and one `s_nop` of PACE % 16 wait states behind every store.
template <int PACE>
__global__ __launch_bounds__(256) void b3w_store_regionfill_kernel(uint8_t *out, uint64_t bytes) {
  const uint32_t grp = blockIdx.x >> 3, x = blockIdx.x & 7;
  u32x4 v = {threadIdx.x, blockIdx.x, 0, 0};
  uint8_t *first = reinterpret_cast<uint8_t *>((reinterpret_cast<uintptr_t>(out) + 0x1ffff) & ~(uintptr_t)0x1ffff);   // (whole regions only)
  const uint64_t regions = (uint64_t)(out + bytes - first) >> 17;
  for (uint64_t r = grp; r < regions; r += 32) {
    uint8_t *p = first + (r << 17) + x * 4096 + threadIdx.x * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      *reinterpret_cast<u32x4 *>(p + k * 32768) = v;
      if (PACE < 1000) {
#pragma unroll
        for (int i = 0; i < PACE / 16; ++i) __builtin_amdgcn_s_sleep(1);
        if (PACE % 16) asm volatile("s_nop %0" :: "n"(PACE % 16 ? PACE % 16 - 1 : 0));
      } else {                                              // PACE - 1000 dependent vector-ALU instructions instead (no 64-clock grid)
#pragma unroll
        for (int i = 0; i < PACE - 1000; ++i) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v.z));
      }
    }
  }
}
constexpr int REGION_PACES[] = {0, 64, 80, 84, 88, 92, 96, 1032, 1036, 1038, 1039, 1040, 1041, 1042, 1044, 1048};
static bool region_pace_known(int pace) {
  for (int p : REGION_PACES) if (p == pace) return true;
  return false;
}
template <int I = 0>
static void launch_region_store(int pace, uint8_t *buf, uint64_t bytes, hipStream_t stream) {
  if constexpr (I < (int)(sizeof(REGION_PACES) / sizeof(int))) {
    if (pace == REGION_PACES[I]) hipLaunchKernelGGL(b3w_store_regionfill_kernel<REGION_PACES[I]>, dim3(256), dim3(256), 0, stream, buf, bytes);
    else launch_region_store<I + 1>(pace, buf, bytes, stream);
  }
}

constexpr uint64_t MiB = 1ull << 20, GiB = 1ull << 30;
constexpr uint64_t HANDLE = 256 * MiB;             // physical granule of a placed buffer
constexpr uint32_t PROBE_STREAMS = 512;            // 256 per side
constexpr uint64_t PROBE_PITCH = 768 * 1024;       // 256 streams x 768 KiB = 192 MiB <= HANDLE
constexpr double CONTRAST = 1.18;                  // fast / slow ratio that counts as "another class" (measured: 1.25-1.3)
constexpr uint64_t POOL_CAP_DEFAULT = 4 * GiB;     // labelled handles kept for later buffers, per label (A, B, C): 12 GiB in all
constexpr uint64_t SEARCH_CAP_MAX = 160 * GiB;     // new handles a search may walk beyond the buffer itself, at the very most
constexpr uint64_t ARENA = 32ull << 40;            // virtual address range per device (less if refused), bump-allocated, never reused

enum : uint8_t { LA = 0, LB = 1, LC = 2, LM = 3 };

struct Cand { hipMemGenericAllocationHandle_t h; uint8_t label; uint32_t slot; };   // slot: its own probe mapping in the arena

struct Placed {
  void *va = nullptr;
  size_t va_bytes = 0;
  std::vector<Cand> pieces;                         // in VA order
  int device = 0;
  int mixed = 0;
};

struct Pool {
  int device = -1;
  bool refs = false;                                // h0 and f0 exist, thr is calibrated
  bool hopeless = false;                            // a walk of 64 GiB or more found no second class: do not walk again
  hipMemGenericAllocationHandle_t h0{}, f0{};
  double thr = 0, lo = 0, hi = 0;
  uint8_t *arena = nullptr;                         // arena_bytes of VA
  uint64_t arena_bytes = 0;
  uint32_t next_slot = 0;                           // next never-used 256 MiB slot of it
  uint32_t slot_h0 = 0, slot_f0 = 0;
  std::vector<Cand> spare;                          // labelled, each still mapped in its probe slot
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t stream = nullptr;                     // the probes' own stream: other work of the process does not sit between the events
  uint64_t handles_created = 0;
  uint64_t search_handles = 0;                      // ... of them by searches for a second class
  double search_seconds = 0;                        // wall time inside b3w_place_alloc calls that wanted a mixed buffer
  uint32_t search_timeouts = 0;                     // searches ended by the time limit
  double t_create = 0, t_map = 0, t_probe = 0, t_release = 0;   // seconds inside hipMemCreate / hipMemMap + SetAccess / probes / unmap + release
  uint64_t n_probe = 0, n_release = 0;
  hipMemAllocationProp prop{};
  hipMemAccessDesc acc{};
};

// Knobs (b3w_place_configure / B3W_PLACE_SEARCH_GIB, B3W_PLACE_POOL_GIB): how much memory a search may touch
// transiently and how much labelled memory stays pooled — co-resident allocators (torch, RCCL) cannot see either.
struct Knobs {
  int64_t search_gib = -1;     // < 0: SEARCH_CAP_MAX (and never more than half of what is free beyond the buffer: b3w_place_alloc)
  int64_t pool_gib = -1;       // < 0: POOL_CAP_DEFAULT per label
  double search_s = 5.0;       // a search that has not found a second class after this many seconds ends: the buffer is plain
  Knobs() {
    if (const char *e = getenv("B3W_PLACE_SEARCH_GIB")) search_gib = atoll(e);
    if (const char *e = getenv("B3W_PLACE_POOL_GIB")) pool_gib = atoll(e);
    if (const char *e = getenv("B3W_PLACE_SEARCH_S")) search_s = atof(e);
  }
};
Knobs &knobs() { static Knobs k; return k; }
uint64_t pool_cap_per_label() { return knobs().pool_gib < 0 ? POOL_CAP_DEFAULT : (uint64_t)knobs().pool_gib * GiB / 3; }
uint64_t search_cap(uint64_t own) {
  // r01-r04: 16 x the buffer, at least 24 GiB — 48 GiB for a config-2 batch, while the first class border of a fresh device lies up
  // to 96 GiB in (a class is about 96 GB; BENCH_r04: 94.5 GiB walked), so an integrator's first buffer on a fresh box came out plain
  // and only bench.py, which raised the knob, got a placed one.  What bounds a search now is TIME (search_s: a walk costs 2-35 ms per
  // GiB, all of it the driver's — tools/ubench/place_cost.hip, profiles/r05/place_cost.log) and half of the free memory.
  (void)own;
  if (knobs().search_gib >= 0) return std::min<uint64_t>((uint64_t)knobs().search_gib * GiB, SEARCH_CAP_MAX);
  return SEARCH_CAP_MAX;
}

std::mutex &mtx() { static std::mutex m; return m; }
std::vector<Pool *> &pools() { static std::vector<Pool *> p; return p; }
std::vector<Placed *> &registry() { static std::vector<Placed *> r; return r; }

Pool *pool_for(int device) {
  for (Pool *p : pools()) if (p->device == device) return p;
  Pool *p = new Pool;
  p->device = device;
  p->prop.type = hipMemAllocationTypePinned;
  p->prop.location.type = hipMemLocationTypeDevice;
  p->prop.location.id = device;
  p->acc.location = p->prop.location;
  p->acc.flags = hipMemAccessFlagsProtReadWrite;
  void *va = nullptr;
  // B3W_PLACE_ARENA_GIB: a smaller address range (tests of what happens when it is used up: buffers come from hipMalloc, labelled plain)
  const uint64_t want = getenv("B3W_PLACE_ARENA_GIB") ? std::max<uint64_t>(2, (uint64_t)atoll(getenv("B3W_PLACE_ARENA_GIB"))) * GiB : ARENA;
  if (want < ARENA) {
    if (hipMemAddressReserve(&va, want, 2 * MiB, nullptr, 0) == hipSuccess) p->arena_bytes = want;
    else { (void)hipGetLastError(); va = nullptr; }
  }
  for (uint64_t sz = ARENA; sz >= (1ull << 40) && !va; sz >>= 1) {
    if (hipMemAddressReserve(&va, sz, 2 * MiB, nullptr, 0) == hipSuccess) p->arena_bytes = sz;
    else { (void)hipGetLastError(); va = nullptr; }
  }
  if (!va) { delete p; return nullptr; }
  p->arena = static_cast<uint8_t *>(va);
  (void)hipEventCreate(&p->e0);
  (void)hipEventCreate(&p->e1);
  if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); p->stream = nullptr; }
  pools().push_back(p);
  return p;
}

// a failure in here only ends a search; the sticky HIP error must not leak into the caller's next launch check
bool create_handle(Pool *p, hipMemGenericAllocationHandle_t *h, uint64_t bytes = HANDLE) {
  const double t0 = wall_s();
  const hipError_t e = hipMemCreate(h, bytes, &p->prop, 0);
  p->t_create += wall_s() - t0;
  if (e == hipSuccess) { p->handles_created += bytes / HANDLE; return true; }
  (void)hipGetLastError();
  return false;
}
uint8_t *slot_addr(Pool *p, uint32_t slot) { return p->arena + (size_t)slot * HANDLE; }
uint32_t slots_left(Pool *p) { return (uint32_t)(p->arena_bytes / HANDLE) - p->next_slot; }
// map a handle at the next never-used slot; false when the arena is used up or the driver refuses
bool map_new(Pool *p, Cand &c) {
  if (!slots_left(p)) return false;
  struct Clock { double &acc; double t0 = wall_s(); ~Clock() { acc += wall_s() - t0; } } clock{p->t_map};
  uint8_t *a = slot_addr(p, p->next_slot);
  if (hipMemMap(a, HANDLE, 0, c.h, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
  c.slot = p->next_slot++;                           // used up whatever happens next
  if (hipMemSetAccess(a, HANDLE, &p->acc, 1) != hipSuccess) { (void)hipMemUnmap(a, HANDLE); (void)hipGetLastError(); return false; }
  return true;
}
void release(Pool *p, Cand &c) {
  const double t0 = wall_s();
  (void)hipMemUnmap(slot_addr(p, c.slot), HANDLE);
  (void)hipMemRelease(c.h);
  p->t_release += wall_s() - t0;
  p->n_release++;
}

double probe_slots(Pool *p, uint32_t sa, uint32_t sb, int reps = 3) {
  struct Clock { double &acc; double t0 = wall_s(); ~Clock() { acc += wall_s() - t0; } } clock{p->t_probe};
  p->n_probe++;
  uint8_t *a = slot_addr(p, sa), *b = slot_addr(p, sb);
  const uint32_t groups = (uint32_t)(PROBE_PITCH / 1024);
  hipLaunchKernelGGL(b3w_store_probe_kernel, dim3(PROBE_STREAMS), dim3(64), 0, p->stream, a, b, PROBE_PITCH, groups);
  (void)hipEventRecord(p->e0, p->stream);
  for (int i = 0; i < reps; i++)
    hipLaunchKernelGGL(b3w_store_probe_kernel, dim3(PROBE_STREAMS), dim3(64), 0, p->stream, a, b, PROBE_PITCH, groups);
  (void)hipEventRecord(p->e1, p->stream);
  if (hipEventSynchronize(p->e1) != hipSuccess) { (void)hipGetLastError(); return 0; }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, p->e0, p->e1);
  return ms > 0 ? (double)reps * PROBE_STREAMS * PROBE_PITCH / 1e6 / ms : 0;       // GB/s
}

// label of the handle mapped in `slot` against the two references
uint8_t label_slot(Pool *p, uint32_t slot) {
  const bool slow0 = probe_slots(p, p->slot_h0, slot) < p->thr, slow1 = probe_slots(p, p->slot_f0, slot) < p->thr;
  return slow0 && !slow1 ? LA : !slow0 && slow1 ? LB : !slow0 && !slow1 ? LC : LM;
}

// First use on a device: walk fresh handles until one is clearly fast with h0 and the next one agrees; fix the
// references and the threshold.  Handles seen on the way are labelled and returned in `got`.  `budget` = handles
// this search may still create.
bool calibrate(Pool *p, uint32_t &budget, std::vector<Cand> &got, double deadline, bool &timed_out) {
  Cand r{};
  if (!budget || !create_handle(p, &r.h)) return false;
  budget--;
  if (!map_new(p, r)) { (void)hipMemRelease(r.h); return false; }
  for (int i = 0; i < 6; i++) (void)probe_slots(p, r.slot, r.slot, 4);             // clocks up before anything is timed
  std::vector<Cand> seen;
  std::vector<double> r0;
  int cand = -1;
  while (budget) {
    if (wall_s() > deadline) { timed_out = true; break; }
    Cand c{};
    if (!create_handle(p, &c.h)) break;
    budget--;
    if (!map_new(p, c)) { (void)hipMemRelease(c.h); break; }
    seen.push_back(c);
    r0.push_back(probe_slots(p, r.slot, c.slot));
    double lo = r0[0], hi = r0[0];
    for (double x : r0) { lo = std::min(lo, x); hi = std::max(hi, x); }
    if (lo > 0 && hi >= CONTRAST * lo) {
      const double thr = sqrt(lo * hi);
      // two consecutive handles that are fast with h0 (not a timing outlier) AND slow with each other: both lie in one
      // other class.  A handle that straddles a class border is fast with h0 and still fairly fast with its pure
      // neighbour; as the second reference it would label a whole class "third" and both sides of the buffer would be
      // filled from it (seen on a box whose first border fell into the second handle).
      // (On memory recycled from earlier processes the classes do not come in long runs: the partner may lie a few
      // handles back, so the newest handle is tried against the last few that were fast with h0, nearest first.)
      const size_t n = r0.size();
      if (n >= 2 && r0[n - 1] >= thr) {
        int tried = 0;
        for (size_t j = n - 1; j-- > 0 && tried < 4 && cand < 0;) {
          if (r0[j] < thr) continue;
          tried++;
          if (probe_slots(p, seen[j].slot, seen[n - 1].slot) < thr) cand = (int)n - 1;
        }
        if (cand >= 0) { p->lo = lo; p->hi = hi; p->thr = thr; break; }
      }
    }
  }
  if (cand < 0) {                                     // one class as far as the budget reaches
    r.label = LM;
    got.push_back(r);
    for (Cand &c : seen) { c.label = LM; got.push_back(c); }
    p->hopeless = !timed_out && (uint64_t)seen.size() * HANDLE >= 64 * GiB;   // a short walk (little free memory right now, the time limit) may be retried
    return false;
  }
  p->h0 = r.h; p->slot_h0 = r.slot;
  p->f0 = seen[cand].h; p->slot_f0 = seen[cand].slot;
  p->refs = true;
  for (size_t i = 0; i < seen.size(); i++) {
    if ((int)i == cand) continue;
    seen[i].label = label_slot(p, seen[i].slot);
    got.push_back(seen[i]);
  }
  return true;
}

void give_back(Pool *p, std::vector<Cand> &v) {           // to the pool while it has room for that label, else to the driver
  uint32_t held[4] = {0, 0, 0, 0};
  for (const Cand &c : p->spare) held[c.label]++;
  for (Cand &c : v) {
    if (p->refs && c.label != LM && (uint64_t)(held[c.label] + 1) * HANDLE <= pool_cap_per_label()) { p->spare.push_back(c); held[c.label]++; }
    else release(p, c);
  }
  v.clear();
}

}  // namespace

// Returns 0 and a device pointer of at least `bytes` bytes; *mixed = 1 when the buffer alternates two memory
// classes, 0 when it is plain.  want_mixed = 0 skips the search (plain VMM buffer).  Negative = hipError_t.
extern "C" int b3w_place_alloc(int device, uint64_t bytes, int want_mixed, void **out, int *mixed, float *rates /* [2] slow, fast or null */) {
  if (!out || !bytes) return -(int)hipErrorInvalidValue;
  *out = nullptr;
  if (mixed) *mixed = 0;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return -(int)e;
  std::lock_guard<std::mutex> guard(mtx());
  Pool *p = pool_for(device);
  if (!p) return -(int)hipErrorOutOfMemory;
  const uint32_t nh = (uint32_t)((bytes + HANDLE - 1) / HANDLE), need1 = (nh + 1) / 2, need2 = nh - need1;
  size_t fr = 0, tot = 0;
  if ((e = hipMemGetInfo(&fr, &tot)) != hipSuccess) return -(int)e;
  const uint64_t own = (uint64_t)nh * HANDLE, pooled = (uint64_t)p->spare.size() * HANDLE;
  if (fr + pooled < own + GiB) return -(int)hipErrorOutOfMemory;
  // search budget in NEW handles: the buffer itself plus at most half of what is free beyond it (other processes may
  // be doing the same on this device), bounded by the search knob — and never more than the address range still has
  uint32_t budget = (uint32_t)((std::min<uint64_t>(own, fr) + std::min<uint64_t>((fr > own ? fr - own : 0) / 2, search_cap(own))) / HANDLE);
  if (want_mixed) (void)hipDeviceSynchronize();               // the probes time stores: nothing else should be running
  if (slots_left(p) < 2 * nh + 8) return -(int)hipErrorOutOfMemory;          // address range used up: caller falls back
  budget = std::min<uint32_t>(budget, slots_left(p) - nh - 4);

  // Candidates by label.  The buffer alternates two SIDES; a side may mix two classes (its pieces are never neighbours),
  // but no class may appear on both sides: the sides are one class against the other two.
  std::vector<Cand> cls[3], rejects, extra;
  auto sort_in = [&](const Cand &c) { (c.label < 3 ? cls[c.label] : rejects).push_back(c); };
  // which class stands alone (and on which side) so that both sides can be covered; -1 = not yet
  auto split = [&](bool &alone_is_side1) {
    const size_t total = cls[0].size() + cls[1].size() + cls[2].size();
    for (int x = 0; x < 3; x++) {
      const size_t own = cls[x].size(), rest = total - own;
      if (own >= need1 && rest >= need2) { alone_is_side1 = true; return x; }
      if (own >= need2 && rest >= need1) { alone_is_side1 = false; return x; }
    }
    return -1;
  };
  bool found = false, alone1 = true, single_class = false;
  int alone = -1;
  std::vector<Cand> got;
  got.swap(p->spare);                                          // labelled earlier, still mapped in their probe slots
  const double t_begin = wall_s(), deadline = t_begin + (knobs().search_s > 0 ? knobs().search_s : 1e9);
  const uint64_t created_before = p->handles_created;
  bool timed_out = false;
  struct Account {                                             // whatever way the call ends
    Pool *p; double t0; uint64_t h0; bool on; bool *to;
    ~Account() { if (on) { p->search_seconds += wall_s() - t0; p->search_handles += p->handles_created - h0; if (*to) p->search_timeouts++; } }
  } account{p, t_begin, created_before, want_mixed != 0, &timed_out};
  double t_cal = 0, t_walk = 0, t_seams = 0;                 // B3W_PLACE_DEBUG: where a search's time goes
  if (want_mixed && !p->hopeless) {
    const double tc0 = wall_s();
    if (!p->refs) (void)calibrate(p, budget, got, deadline, timed_out);
    t_cal = wall_s() - tc0;
    for (const Cand &c : got) sort_in(c);
    got.clear();
    const double tw0 = wall_s();
    while (p->refs && (alone = split(alone1)) < 0 && budget) {
      if (wall_s() > deadline) { timed_out = true; break; }
      Cand c{};
      if (!create_handle(p, &c.h)) break;
      budget--;
      if (!map_new(p, c)) { (void)hipMemRelease(c.h); break; }
      c.label = label_slot(p, c.slot);
      sort_in(c);
    }
    found = p->refs && (alone = split(alone1)) >= 0;
    // want_mixed == 2 (harness, B3W_PLACEMENT=single): every piece from ONE class — the buffer a plain hipMalloc is on an unlucky day,
    // made on purpose, so that the store shapes meant for caller-owned buffers can be developed and measured on any box
    if (want_mixed == 2 && p->refs) {
      auto biggest = [&] { int b = 0; for (int x = 1; x < 3; x++) if (cls[x].size() > cls[b].size()) b = x; return b; };
      while (cls[biggest()].size() < nh && budget && wall_s() <= deadline) {
        Cand c{};
        if (!create_handle(p, &c.h)) break;
        budget--;
        if (!map_new(p, c)) { (void)hipMemRelease(c.h); break; }
        c.label = label_slot(p, c.slot);
        sort_in(c);
      }
      const int b = biggest();
      if (cls[b].size() >= nh) {
        std::vector<Cand> keep(cls[b].end() - nh, cls[b].end());
        cls[b].resize(cls[b].size() - nh);
        for (auto *v : {&cls[0], &cls[1], &cls[2]}) { extra.insert(extra.end(), v->begin(), v->end()); v->clear(); }
        cls[b] = keep;                                         // the plain branch below takes these nh pieces first
        single_class = true;
      }
      found = false;
    }
    t_walk = wall_s() - tw0;
  }
  for (const Cand &c : got) sort_in(c);                        // (search skipped)
  // the order of the pieces
  std::vector<Cand> order;
  uint32_t slow_seams = 0;
  const double ts0 = wall_s();
  if (found) {
    std::vector<Cand> side1, side2;                            // most recently created first
    std::vector<Cand> &sa = alone1 ? side1 : side2, &sr = alone1 ? side2 : side1;
    const uint32_t na = alone1 ? need1 : need2, nr = alone1 ? need2 : need1;
    for (size_t i = cls[alone].size(); i-- > 0;) (sa.size() < na ? sa : extra).push_back(cls[alone][i]);
    for (int x = 0; x < 3; x++) {
      if (x == alone) continue;
      for (size_t i = cls[x].size(); i-- > 0;) (sr.size() < nr ? sr : extra).push_back(cls[x][i]);
    }
    for (auto &v : cls) v.clear();
    for (uint32_t s = 0; s < nh; s++) order.push_back((s & 1) ? side2[s / 2] : side1[s / 2]);
    // check every seam of that order directly; where two neighbours are slow together try other candidates
    for (uint32_t s = 0; s + 1 < nh; s++) {
      bool ok = probe_slots(p, order[s].slot, order[s + 1].slot) >= p->thr;
      for (size_t t = 0; t < extra.size() && t < 6 && !ok; t++)
        if (probe_slots(p, order[s].slot, extra[t].slot) >= p->thr) { std::swap(order[s + 1], extra[t]); ok = true; }
      if (!ok) slow_seams++;
    }
    if (slow_seams > nh / 8) {
      // the labels did not hold (impure references or recycled, fragmented handles): build the chain greedily from
      // direct probes instead — next piece = any unused candidate that is fast with the previous piece
      std::vector<Cand> cands = order;
      cands.insert(cands.end(), extra.begin(), extra.end());
      std::vector<char> used(cands.size(), 0);
      std::vector<Cand> chain;
      chain.push_back(cands[0]);
      used[0] = 1;
      uint32_t slow2 = 0;
      size_t scan = 1;
      while (chain.size() < nh) {
        int pick = -1, fallback = -1, tries = 0;
        for (size_t k = 0; k < cands.size() && tries < 12; k++) {
          const size_t i = (scan + k) % cands.size();
          if (used[i]) continue;
          if (fallback < 0) fallback = (int)i;
          tries++;
          if (probe_slots(p, chain.back().slot, cands[i].slot) >= p->thr) { pick = (int)i; break; }
        }
        if (pick < 0) { pick = fallback; slow2++; }
        if (pick < 0) break;
        used[pick] = 1;
        chain.push_back(cands[pick]);
        scan = (size_t)pick + 1;
      }
      if (chain.size() == nh && slow2 < slow_seams) {
        order = chain;
        extra.clear();
        for (size_t i = 0; i < cands.size(); i++) if (!used[i]) extra.push_back(cands[i]);
        slow_seams = slow2;
      }
      if (slow_seams > nh / 8) found = false;            // not worth calling it mixed
    }
  } else {
    // plain: whatever handles there are, then new ones
    std::vector<Cand> all;
    for (auto *v : {&cls[0], &cls[1], &cls[2], &rejects, &extra}) { all.insert(all.end(), v->begin(), v->end()); v->clear(); }
    for (const Cand &c : all) (order.size() < nh ? order : extra).push_back(c);
    while (order.size() < nh) {
      Cand c{};
      c.label = LM;
      if (!create_handle(p, &c.h)) break;
      if (!map_new(p, c)) { (void)hipMemRelease(c.h); break; }
      order.push_back(c);
    }
  }
  t_seams = wall_s() - ts0;
  if (timed_out && getenv("B3W_PLACE_DEBUG"))
    fprintf(stderr, "b3w_place_alloc: no second class of memory within %.0f s (B3W_PLACE_SEARCH_S): this buffer is plain\n", knobs().search_s);
  if (getenv("B3W_PLACE_DEBUG")) {
    fprintf(stderr, "b3w_place_alloc: %u pieces, found=%d, thr=%.0f (lo %.0f hi %.0f), slow seams left=%u, spare %zu, rejects %zu, slots used %u\n  ", nh,
            (int)found, p->thr, p->lo, p->hi, slow_seams, extra.size(), rejects.size(), p->next_slot);
    for (const Cand &c : order) fputc("ABCM"[c.label], stderr);
    fprintf(stderr, "\n  this call: calibrate %.3f s, walk %.3f s, order + seams %.3f s; pool totals: hipMemCreate %.3f s (%llu handles), map %.3f s, %llu probes %.3f s, "
            "%llu releases %.3f s\n", t_cal, t_walk, t_seams, p->t_create, (unsigned long long)p->handles_created, p->t_map, (unsigned long long)p->n_probe, p->t_probe,
            (unsigned long long)p->n_release, p->t_release);
  }
  int rc = order.size() < nh ? -(int)hipErrorOutOfMemory : 0;
  Placed *pl = nullptr;
  if (rc == 0) {
    // the buffer: the next nh never-used slots of the arena, second mapping of every piece
    uint8_t *fin = slot_addr(p, p->next_slot);
    uint32_t mapped = 0;
    for (; mapped < nh; mapped++) {
      e = hipMemMap(fin + (size_t)mapped * HANDLE, HANDLE, 0, order[mapped].h, 0);
      if (e != hipSuccess) { rc = -(int)e; break; }
    }
    p->next_slot += nh;                                  // used up whatever happens next
    if (rc == 0 && (e = hipMemSetAccess(fin, (size_t)nh * HANDLE, &p->acc, 1)) != hipSuccess) rc = -(int)e;
    if (rc != 0) {
      for (uint32_t s = 0; s < mapped; s++) (void)hipMemUnmap(fin + (size_t)s * HANDLE, HANDLE);
      (void)hipGetLastError();
    } else {
      pl = new Placed;
      pl->va = fin;
      pl->va_bytes = (size_t)nh * HANDLE;
      pl->pieces = order;
      pl->device = device;
      pl->mixed = found ? 1 : single_class ? -1 : 0;           // (-1: one class on purpose; reported as plain)
    }
  }
  if (rc != 0) give_back(p, order);
  give_back(p, extra);
  for (Cand &c : rejects) release(p, c);
  if (rc != 0) return rc;
  registry().push_back(pl);
  *out = pl->va;
  if (mixed) *mixed = pl->mixed > 0 ? pl->mixed : 0;
  if (rates) { rates[0] = (float)p->lo; rates[1] = (float)p->hi; }
  return 0;
}

// What the allocator knows about the memory `ptr` points into: 1 = inside a buffer of alternating classes ("mixed" pieces), 0 = anything
// else — a plain hipMalloc / torch buffer, a one-class buffer, a foreign pointer.  The default launch policy asks (b3w_ctx.cpp).
extern "C" int b3w_place_is_mixed(const void *ptr) {
  std::lock_guard<std::mutex> guard(mtx());
  const uint8_t *p = static_cast<const uint8_t *>(ptr);
  for (const Placed *pl : registry())
    if (p >= static_cast<const uint8_t *>(pl->va) && p < static_cast<const uint8_t *>(pl->va) + pl->va_bytes) return pl->mixed > 0 ? 1 : 0;
  return 0;
}

// 0 = freed; 1 = not a pointer handed out by b3w_place_alloc
extern "C" int b3w_place_free(void *ptr) {
  std::lock_guard<std::mutex> guard(mtx());
  Placed *pl = nullptr;
  std::vector<Placed *> &r = registry();
  for (size_t i = 0; i < r.size() && !pl; i++)
    if (r[i]->va == ptr) { pl = r[i]; r.erase(r.begin() + i); }
  if (!pl) return 1;
  (void)hipSetDevice(pl->device);
  {
    B3wCaptureRelaxed relaxed;                               // (b3w_capture.h: hipDeviceSynchronize is refused while ANY stream captures)
    Pool *pw = pool_for(pl->device);                         // nothing may still write what is unmapped next
    const hipStream_t mine[1] = {pw ? pw->stream : nullptr};
    (void)b3w_device_wait(mine, pw && pw->stream ? 1 : 0);   // (without a spare and without streams: hipDeviceSynchronize; b3w_capture.h)
  }
  for (size_t s = 0; s < pl->pieces.size(); s++) (void)hipMemUnmap(static_cast<uint8_t *>(pl->va) + s * HANDLE, HANDLE);   // the range is not used again
  Pool *p = pool_for(pl->device);
  if (p) give_back(p, pl->pieces);                       // labelled pieces: kept for the next buffer while there is room
  delete pl;
  return 0;
}

extern "C" void b3w_place_configure(int64_t search_gib, int64_t pool_gib) {
  std::lock_guard<std::mutex> guard(mtx());
  if (search_gib >= 0) knobs().search_gib = search_gib;
  if (pool_gib >= 0) knobs().pool_gib = pool_gib;
}

extern "C" void b3w_place_search_limit(double seconds) {      // <= 0: no time limit
  std::lock_guard<std::mutex> guard(mtx());
  knobs().search_s = seconds;
}

// out: seconds inside searching b3w_place_alloc calls, GiB of new physical memory those calls created, searches ended by the
// time limit, the time limit in seconds
extern "C" void b3w_place_search_stats(int device, double out[4]) {
  std::lock_guard<std::mutex> guard(mtx());
  out[0] = out[1] = out[2] = 0;
  out[3] = knobs().search_s;
  for (Pool *p : pools()) {
    if (p->device != device) continue;
    out[0] = p->search_seconds;
    out[1] = (double)p->search_handles * (double)HANDLE / (double)GiB;
    out[2] = p->search_timeouts;
  }
}

// out: arena bytes, arena bytes used up (never reused), pooled bytes, bytes of live placed buffers, live placed buffers,
// physical handles created so far
extern "C" void b3w_place_stats(int device, uint64_t out[6]) {
  std::lock_guard<std::mutex> guard(mtx());
  for (int i = 0; i < 6; i++) out[i] = 0;
  for (Pool *p : pools()) {
    if (p->device != device) continue;
    out[0] = p->arena_bytes;
    out[1] = (uint64_t)p->next_slot * HANDLE;
    out[2] = (uint64_t)p->spare.size() * HANDLE;
    out[5] = p->handles_created;
  }
  for (Placed *pl : registry()) if (pl->device == device) { out[3] += pl->va_bytes; out[4]++; }
}

// release the pooled handles of every device (the references stay)
extern "C" void b3w_place_trim(void) {
  std::lock_guard<std::mutex> guard(mtx());
  for (Pool *p : pools()) {
    (void)hipSetDevice(p->device);
    for (Cand &c : p->spare) release(p, c);
    p->spare.clear();
  }
}

// out: seconds inside hipMemCreate, hipMemMap + hipMemSetAccess, the store probes, hipMemUnmap + hipMemRelease — since the process began
extern "C" void b3w_place_cost_breakdown(int device, double out[4]) {
  std::lock_guard<std::mutex> guard(mtx());
  out[0] = out[1] = out[2] = out[3] = 0;
  for (Pool *p : pools()) {
    if (p->device != device) continue;
    out[0] = p->t_create; out[1] = p->t_map; out[2] = p->t_probe; out[3] = p->t_release;
  }
}

// (measurements, tools/ubench/gather_beside_writer.py) the commit kernels' memory behaviour without their arithmetic's details: every lane
// reads a random 64-byte entry of a table of `entries` entries and then runs `valu` dependent 32-bit multiply-adds on it, `iters` times;
// 768 threads a CU as the commit kernel.  Does an L2-sized table keep such a kernel fed beside a writer that saturates HBM?
__global__ __launch_bounds__(256) void b3w_gather_valu_kernel(const uint4 *__restrict__ table, uint32_t entries, uint32_t iters, uint32_t valu, uint32_t *__restrict__ sink) {
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u, acc = 0;
  // (the shader clock this kernel runs at: cycles / (100 MHz ticks) around the loop, by one lane of a workgroup in the middle of the grid)
  const bool stamp = blockIdx.x == gridDim.x / 2u && threadIdx.x == 0;
  const unsigned long long c0 = stamp ? __builtin_amdgcn_s_memtime() : 0ull, r0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0ull;
  for (uint32_t i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    const uint4 *e = table + (size_t)((x >> 4) % entries) * 4;
    const uint4 a = e[0], b = e[1], c = e[2], d = e[3];
    uint32_t v = a.x ^ b.y ^ c.z ^ d.w ^ x;
    for (uint32_t k = 0; k < valu; ++k) v = v * 2246822519u + (v >> 15) + k;
    acc += v;
  }
  if (stamp) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    reinterpret_cast<unsigned long long *>(sink)[1] = c1 - c0;
    reinterpret_cast<unsigned long long *>(sink)[2] = r1 - r0;
  }
  if (acc == 0x9E3779B9u) sink[0] = acc;                      // (keeps the loop)
}
extern "C" int b3w_place_gather_launch(const void *table, uint64_t table_bytes, uint32_t iters, uint32_t valu, uint32_t grid, uint32_t *sink, hipStream_t stream) {
  if (!table || table_bytes < 64 || !grid) return -(int)hipErrorInvalidValue;
  hipLaunchKernelGGL(b3w_gather_valu_kernel, dim3(grid), dim3(256), 0, stream, static_cast<const uint4 *>(table), (uint32_t)(table_bytes / 64), iters, valu, sink);
  return -(int)hipGetLastError();
}

// one pure-store pass, enqueued on `stream` and not waited for (tools/ubench/overlap_commit_probe.py: a writer with no LDS and few
// registers beside another kernel)
extern "C" int b3w_place_store_launch(uint8_t *buf, uint64_t pitch, uint32_t n, uint32_t body_bytes, int shape, hipStream_t stream) {
  if (!buf || !n || body_bytes < 1024 || pitch < body_bytes) return -(int)hipErrorInvalidValue;
  const uint32_t tiles = body_bytes / 1024;
  // 300 + k / 400 + k / 500 + k / 600 + k: the minimal-register writer, k waves per CU, allocating 8 / 16 / 32 / 64 VGPRs
  if (shape >= 600) hipLaunchKernelGGL(b3w_store_tiny_kernel<63>, dim3(256u * (uint32_t)(shape - 600)), dim3(64), 0, stream, buf, pitch, n, tiles);
  else if (shape >= 500) hipLaunchKernelGGL(b3w_store_tiny_kernel<31>, dim3(256u * (uint32_t)(shape - 500)), dim3(64), 0, stream, buf, pitch, n, tiles);
  else if (shape >= 400) hipLaunchKernelGGL(b3w_store_tiny_kernel<15>, dim3(256u * (uint32_t)(shape - 400)), dim3(64), 0, stream, buf, pitch, n, tiles);
  else if (shape >= 300) hipLaunchKernelGGL(b3w_store_tiny_kernel<0>, dim3(256u * (uint32_t)(shape - 300)), dim3(64), 0, stream, buf, pitch, n, tiles);
  else if (shape >= 200) hipLaunchKernelGGL(b3w_store_persistent_kernel<8>, dim3(256u * (uint32_t)(shape - 200)), dim3(64), 0, stream, buf, pitch, n, tiles, 0u);   // (shape - 200) waves per CU
  else if (shape >= 100) hipLaunchKernelGGL(b3w_store_streams_kernel<8>, dim3((n + 7) / 8), dim3(64), 0, stream, buf, pitch, n, tiles, (uint32_t)(shape - 100));   // paused
  else if (shape == 0) hipLaunchKernelGGL(b3w_store_streams_kernel<4>, dim3((n + 3) / 4), dim3(64), 0, stream, buf, pitch, n, tiles, 0u);
  else if (shape == 1) hipLaunchKernelGGL(b3w_store_streams_kernel<8>, dim3((n + 7) / 8), dim3(64), 0, stream, buf, pitch, n, tiles, 0u);
  else hipLaunchKernelGGL(b3w_store_fill_kernel, dim3(256), dim3(256), 0, stream, buf, (uint64_t)n * pitch);
  return -(int)hipGetLastError();
}

// GB/s of `iters` pure-store passes over [buf, buf + n * pitch) on `stream` (HIP events; 2 untimed passes first).
// shape 0: body streams, one wave per 4 bodies; 1: per 8 bodies; 2: the fill shape; 3 / 4: PACED persistent body streams — 512 single-wave
// workgroups (two per CU) taking groups of 4 / 8 bodies in turn, 4 vector-ALU instructions in front of every store (the best store-only
// shapes of round 6's sweep on a placed buffer); 5: 8 bodies per wave, paced, one wave per group; 6 / 7: the fill-ordered kernel's store order
// over the whole 128 KiB regions of the buffer, paced by s_sleep (5 or 6 x 64 clocks, + s_nop, per store) / by 38 - 44 dependent vector-ALU
// instructions per store: the best of five / six compiled-in paces each (700 + p: ONE pace of REGION_PACES, for the scan).  Negative = hipError_t.
extern "C" int b3w_place_store_rate(uint8_t *buf, uint64_t pitch, uint32_t n, uint32_t body_bytes, int shape, uint32_t iters, hipStream_t stream, double *gbs) {
  if (shape == 6 || shape == 7) {
    // the paced fill order is a cliff in its pace, and where the cliff stands moves a little with the buffer (a 49 GB one is 2-4 % slower
    // than a 3 GB one in every shape): the best of the compiled-in paces around it
    static const int sleeps[] = {80, 84, 88, 92, 96}, valus[] = {1038, 1039, 1040, 1041, 1042, 1044};
    double best = 0;
    for (int pace : std::vector<int>(shape == 6 ? std::begin(sleeps) : std::begin(valus), shape == 6 ? std::end(sleeps) : std::end(valus))) {
      double g = 0;
      const int rc = b3w_place_store_rate(buf, pitch, n, body_bytes, 700 + pace, iters, stream, &g);
      if (rc) return rc;
      best = g > best ? g : best;
    }
    if (gbs) *gbs = best;
    return gbs ? 0 : -(int)hipErrorInvalidValue;
  }
  if (!buf || !n || !iters || !gbs || body_bytes < 1024 || pitch < body_bytes || shape < 0 || (shape > 7 && (shape < 700 || shape > 1899))) return -(int)hipErrorInvalidValue;
  const uint32_t tiles = body_bytes / 1024;
  const bool region = shape == 6 || shape == 7 || shape >= 700;
  const uint64_t whole = (uint64_t)n * pitch, lead = (0x20000 - (reinterpret_cast<uintptr_t>(buf) & 0x1ffff)) & 0x1ffff;
  if (region && whole < lead + 0x20000) return -(int)hipErrorInvalidValue;
  if (shape >= 700 && !region_pace_known(shape - 700)) return -(int)hipErrorInvalidValue;
  const uint64_t per_pass = region ? ((whole - lead) >> 17) << 17 : shape == 2 ? ((uint64_t)n * pitch / 4096) * 4096 : (uint64_t)n * tiles * 1024;
  auto launch = [&] {
    if (shape == 0) hipLaunchKernelGGL(b3w_store_streams_kernel<4>, dim3((n + 3) / 4), dim3(64), 0, stream, buf, pitch, n, tiles, 0u);
    else if (shape == 1) hipLaunchKernelGGL(b3w_store_streams_kernel<8>, dim3((n + 7) / 8), dim3(64), 0, stream, buf, pitch, n, tiles, 0u);
    else if (shape == 3) hipLaunchKernelGGL(b3w_store_persistent_kernel<4>, dim3(512), dim3(64), 0, stream, buf, pitch, n, tiles, 4u);
    else if (shape == 4) hipLaunchKernelGGL(b3w_store_persistent_kernel<8>, dim3(512), dim3(64), 0, stream, buf, pitch, n, tiles, 4u);
    else if (shape == 5) hipLaunchKernelGGL(b3w_store_persistent_kernel<8>, dim3((n + 7) / 8), dim3(64), 0, stream, buf, pitch, n, tiles, 4u);
    else if (region) launch_region_store(shape - 700, buf, whole, stream);
    else hipLaunchKernelGGL(b3w_store_fill_kernel, dim3(256), dim3(256), 0, stream, buf, (uint64_t)n * pitch);
  };
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) { launch(); launch(); e = hipEventRecord(e0, stream); }
  for (uint32_t i = 0; i < iters && e == hipSuccess; i++) launch();
  if (e == hipSuccess) e = hipEventRecord(e1, stream);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e == hipSuccess) e = hipGetLastError();
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (e != hipSuccess) return -(int)e;
  *gbs = ms > 0 ? (double)per_pass * iters / 1e6 / ms : 0;
  return 0;
}
