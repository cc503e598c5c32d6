// b3w_capture.h — releasing objects while somebody's stream capture is open.
// A capture in the GLOBAL mode (the default of hipStreamBeginCapture and of torch.cuda.graph) forbids every thread the calls that
// wait for or allocate on the device: hipFree, hipMalloc, hipStreamSynchronize of ANY stream, hipDeviceSynchronize return
// hipErrorStreamCaptureUnsupported AND invalidate the capture (tools/ubench/capture_free_probe.hip).  A destroy or free of this
// library may run at any time — Python's cyclic collector finalises a Context in the middle of somebody's capture (it happened:
// profiles/r05/gpu_suite_abort_in_capture.log) — so the release paths put the calling thread into the RELAXED mode for their
// duration: hipFree, hipMalloc, stream and event waits are then allowed and leave the capture alone.  hipDeviceSynchronize is refused
// even then; what stands in for it is b3w_device_wait below.
#pragma once
#include <hip/hip_runtime_api.h>

struct B3wCaptureRelaxed {
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  bool swapped;
  B3wCaptureRelaxed() { swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; if (!swapped) (void)hipGetLastError(); }
  ~B3wCaptureRelaxed() { if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode); }
  B3wCaptureRelaxed(const B3wCaptureRelaxed &) = delete;
  B3wCaptureRelaxed &operator=(const B3wCaptureRelaxed &) = delete;
};

// Wait for everything the current device has been given so far, also while another stream is capturing (call it under a
// B3wCaptureRelaxed): hipFree waits for the work of every stream before it releases — a few bytes allocated and freed are a device-wide
// wait that a capture does not forbid (tools/ubench/capture_sync_probe.hip: 3.00 s behind a 3 s kernel on another stream, with and
// without a capture open, the capture valid afterwards).  That wait is the RUNTIME's behaviour, not a promise of the API, and an
// allocation on a release path can fail, so:
//   * the 256 bytes that are freed were allocated EARLIER — one spare per device, replaced after each wait while memory lasts — and
//     the release path itself needs no allocation to succeed;
//   * without a spare (memory used up), the caller's own streams are waited for one by one — what actually touches the memory about
//     to be recycled; hipDeviceSynchronize, which a capture refuses and which invalidates it, is the LAST resort (no streams given);
//   * the outcome is returned and a failure is said on stderr: the caller goes on releasing (a destructor cannot refuse), but not silently.
#include <stdio.h>
#include <mutex>

static inline void **b3w_wait_spare_slot(int dev) {
  static void *spare[64] = {};                               // (one per device ordinal; guarded by b3w_wait_mutex)
  return &spare[dev & 63];
}
static inline std::mutex &b3w_wait_mutex() { static std::mutex m; return m; }

// called where allocations are welcome (context creation): makes sure the current device has its spare
static inline void b3w_device_wait_prepare() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
  std::lock_guard<std::mutex> lock(b3w_wait_mutex());
  void **slot = b3w_wait_spare_slot(dev);
  if (!*slot && hipMalloc(slot, 256) != hipSuccess) { *slot = nullptr; (void)hipGetLastError(); }
}

static inline hipError_t b3w_device_wait(const hipStream_t *streams = nullptr, int nstreams = 0) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  void *p = nullptr;
  {
    std::lock_guard<std::mutex> lock(b3w_wait_mutex());
    void **slot = b3w_wait_spare_slot(dev);
    p = *slot;
    *slot = nullptr;
  }
  if (!p && hipMalloc(&p, 256) != hipSuccess) { p = nullptr; (void)hipGetLastError(); }
  if (p) {
    const hipError_t e = hipFree(p);                         // the device-wide wait
    {
      std::lock_guard<std::mutex> lock(b3w_wait_mutex());
      void **slot = b3w_wait_spare_slot(dev);
      if (!*slot && hipMalloc(slot, 256) != hipSuccess) { *slot = nullptr; (void)hipGetLastError(); }
    }
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
  }
  hipError_t e = hipSuccess;
  if (nstreams > 0) {
    for (int i = 0; i < nstreams; i++) {
      if (!streams[i]) continue;
      const hipError_t ei = hipStreamSynchronize(streams[i]);
      if (ei != hipSuccess) { e = ei; (void)hipGetLastError(); }
    }
  } else {
    e = hipDeviceSynchronize();
    if (e != hipSuccess) (void)hipGetLastError();
  }
  if (e != hipSuccess) fprintf(stderr, "b3wit: could not wait for the device before releasing memory (%s); releasing anyway\n", hipGetErrorString(e));
  return e;
}
