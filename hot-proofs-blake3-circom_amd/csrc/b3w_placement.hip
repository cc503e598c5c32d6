// b3w_placement.hip — where witness bodies live in HBM.
//
// Measured on MI355X (tools/ubench/placement_map*.hip, vmm_classes.hip, profiles/r01/placement/): the 288 GB of
// HBM3E fall into three "classes" of physical memory (about 96 GB each, in physical segments of 8-64 GiB;
// presumably the three ranks of the 12-high stacks).  A store pattern made of thousands of independent streams —
// one witness body per stream, what the witness kernels emit — runs at about 5.4 TB/s while all streams land in
// ONE class and at about 7.0 TB/s when they are split over two; a plain hipMalloc buffer almost always sits in one.
//
// b3w_place_alloc therefore builds the body buffer with the HIP virtual-memory API: physical handles of
// B3W_PLACE_HANDLE bytes are created one after the other (the driver hands out physical memory linearly), each is
// classified by timing a short split-store probe against the first handle ("same class" = slow, "other class" =
// fast), until both groups can cover half the buffer; the two groups are then mapped ALTERNATELY into one
// contiguous virtual range and every other handle is released.  Bodies written in natural order then always
// straddle both classes; nothing changes for the kernels or for consumers of the buffer (one linear device
// range).  When no second class shows up (other hardware, memory nearly full) the buffer is simply the first
// handles in creation order — correct, only slower — and the placement is reported as "plain".
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <mutex>

#include "b3w_kernels.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// probe: stream i writes `bytes` consecutive bytes, even streams from lo, odd ones from hi, one wave per stream
__global__ __launch_bounds__(64) void b3w_store_probe_kernel(uint8_t *lo, uint8_t *hi, uint64_t pitch, uint32_t groups) {
  const uint32_t i = blockIdx.x, lane = threadIdx.x;
  const u32x4 v = {0, 0, 0, 0};
  uint8_t *base = ((i & 1) ? hi : lo) + (uint64_t)(i >> 1) * pitch + lane * 16;
  for (uint32_t g = 0; g < groups; ++g) *reinterpret_cast<u32x4 *>(base + (uint64_t)g * 1024) = v;
}

constexpr uint64_t MiB = 1ull << 20, GiB = 1ull << 30;
constexpr uint64_t HANDLE = 256 * MiB;             // physical granule of a placed buffer
constexpr uint32_t PROBE_STREAMS = 512;            // 256 per side
constexpr uint64_t PROBE_PITCH = 768 * 1024;       // 256 streams x 768 KiB = 192 MiB <= HANDLE
constexpr double CONTRAST = 1.12;                  // fast / slow ratio that counts as "another class"

struct Placed {
  void *va = nullptr;
  size_t va_bytes = 0;
  std::vector<hipMemGenericAllocationHandle_t> handles;   // in VA order
  int mixed = 0;
};

std::vector<Placed *> &registry() { static std::vector<Placed *> r; return r; }
std::mutex &registry_mutex() { static std::mutex m; return m; }

double probe_rate(uint8_t *a, uint8_t *b, hipEvent_t e0, hipEvent_t e1) {
  const uint32_t groups = (uint32_t)(PROBE_PITCH / 1024);
  hipLaunchKernelGGL(b3w_store_probe_kernel, dim3(PROBE_STREAMS), dim3(64), 0, 0, a, b, PROBE_PITCH, groups);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 3; i++)
    hipLaunchKernelGGL(b3w_store_probe_kernel, dim3(PROBE_STREAMS), dim3(64), 0, 0, a, b, PROBE_PITCH, groups);
  (void)hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return 0;
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms > 0 ? 3.0 * PROBE_STREAMS * PROBE_PITCH / 1e6 / ms : 0;       // GB/s
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
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const uint32_t nh = (uint32_t)((bytes + HANDLE - 1) / HANDLE), need = (nh + 1) / 2;
  size_t fr = 0, tot = 0;
  if ((e = hipMemGetInfo(&fr, &tot)) != hipSuccess) return -(int)e;
  if (fr < (uint64_t)nh * HANDLE + GiB) return -(int)hipErrorOutOfMemory;
  // search budget: the buffer itself plus at most half of what is free beyond it (other processes may be doing the
  // same on this device), never more than 160 GiB
  const uint64_t own = (uint64_t)nh * HANDLE;
  const uint64_t budget = own + std::min<uint64_t>((fr - own) / 2, 160 * GiB);
  const uint32_t maxh = want_mixed ? (uint32_t)(budget / HANDLE) : nh;
  if (maxh < nh) return -(int)hipErrorOutOfMemory;

  void *scr = nullptr;
  if ((e = hipMemAddressReserve(&scr, (size_t)maxh * HANDLE, 2 * MiB, nullptr, 0)) != hipSuccess) return -(int)e;
  uint8_t *V = static_cast<uint8_t *>(scr);
  std::vector<hipMemGenericAllocationHandle_t> h;
  std::vector<double> rate;                          // rate[i] = probe(handle 0, handle i)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  std::vector<uint32_t> same, other;                 // handle indices by class relative to handle 0
  double lo = 0, hi = 0;
  bool found = false;
  for (uint32_t i = 0; i < maxh; i++) {
    hipMemGenericAllocationHandle_t hh;
    // a failure here only ends the search; the sticky HIP error must not leak into the caller's next launch check
    if (hipMemCreate(&hh, HANDLE, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
    if (hipMemMap(V + (size_t)i * HANDLE, HANDLE, 0, hh, 0) != hipSuccess) { (void)hipMemRelease(hh); (void)hipGetLastError(); break; }
    if (hipMemSetAccess(V + (size_t)i * HANDLE, HANDLE, &acc, 1) != hipSuccess) {
      (void)hipMemUnmap(V + (size_t)i * HANDLE, HANDLE);
      (void)hipMemRelease(hh);
      (void)hipGetLastError();
      break;
    }
    h.push_back(hh);
    if (!want_mixed) { rate.push_back(0); continue; }
    rate.push_back(i ? probe_rate(V, V + (size_t)i * HANDLE, e0, e1) : 0);
    if (i < 1) continue;
    lo = hi = rate[1];
    for (uint32_t k = 1; k <= i; k++) { lo = std::min(lo, rate[k]); hi = std::max(hi, rate[k]); }
    if (lo <= 0 || hi < CONTRAST * lo) continue;     // one class so far
    const double thr = sqrt(lo * hi);
    same.assign(1, 0u);
    other.clear();
    for (uint32_t k = 1; k <= i; k++) (rate[k] < thr ? same : other).push_back(k);
    if (same.size() >= need && other.size() >= nh - need) { found = true; break; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  int rc = 0;
  Placed *p = nullptr;
  std::vector<char> used(h.size(), 0);
  if (h.size() < nh) rc = -(int)hipErrorOutOfMemory;
  if (rc == 0) {
    // VA order of the final buffer: alternate the most recently created handles of both groups (plain: creation order)
    std::vector<uint32_t> order;
    if (found) {
      for (uint32_t s = 0; s < nh; s++) {
        std::vector<uint32_t> &g = (s & 1) ? other : same;
        order.push_back(g[g.size() - 1 - s / 2]);
      }
    } else {
      for (uint32_t s = 0; s < nh; s++) order.push_back(s);
    }
    for (size_t i = 0; i < h.size(); i++) (void)hipMemUnmap(V + i * HANDLE, HANDLE);
    void *fin = nullptr;
    e = hipMemAddressReserve(&fin, (size_t)nh * HANDLE, 2 * MiB, nullptr, 0);
    if (e != hipSuccess) rc = -(int)e;
    else {
      p = new Placed;
      p->va = fin;
      p->va_bytes = (size_t)nh * HANDLE;
      p->mixed = found ? 1 : 0;
      for (uint32_t s = 0; s < nh && rc == 0; s++) {
        e = hipMemMap(static_cast<uint8_t *>(fin) + (size_t)s * HANDLE, HANDLE, 0, h[order[s]], 0);
        if (e != hipSuccess) { rc = -(int)e; break; }
        p->handles.push_back(h[order[s]]);
        used[order[s]] = 1;
      }
      if (rc == 0 && (e = hipMemSetAccess(fin, (size_t)nh * HANDLE, &acc, 1)) != hipSuccess) rc = -(int)e;
      if (rc != 0) {
        for (size_t s = 0; s < p->handles.size(); s++) (void)hipMemUnmap(static_cast<uint8_t *>(fin) + s * HANDLE, HANDLE);
        (void)hipMemAddressFree(fin, (size_t)nh * HANDLE);
        std::fill(used.begin(), used.end(), 0);
        delete p;
        p = nullptr;
      }
    }
  } else {
    for (size_t i = 0; i < h.size(); i++) (void)hipMemUnmap(V + i * HANDLE, HANDLE);
  }
  for (size_t i = 0; i < h.size(); i++) if (!used[i]) (void)hipMemRelease(h[i]);
  (void)hipMemAddressFree(scr, (size_t)maxh * HANDLE);
  if (rc != 0) return rc;
  {
    std::lock_guard<std::mutex> g(registry_mutex());
    registry().push_back(p);
  }
  *out = p->va;
  if (mixed) *mixed = p->mixed;
  if (rates) { rates[0] = (float)lo; rates[1] = (float)hi; }
  return 0;
}

// 0 = freed; 1 = not a pointer handed out by b3w_place_alloc
extern "C" int b3w_place_free(void *ptr) {
  Placed *p = nullptr;
  {
    std::lock_guard<std::mutex> g(registry_mutex());
    std::vector<Placed *> &r = registry();
    for (size_t i = 0; i < r.size() && !p; i++)
      if (r[i]->va == ptr) { p = r[i]; r.erase(r.begin() + i); }
  }
  if (!p) return 1;
  {
    (void)hipDeviceSynchronize();
    for (size_t s = 0; s < p->handles.size(); s++) {
      (void)hipMemUnmap(static_cast<uint8_t *>(p->va) + s * HANDLE, HANDLE);
      (void)hipMemRelease(p->handles[s]);
    }
    (void)hipMemAddressFree(p->va, p->va_bytes);
    delete p;
  }
  return 0;
}
