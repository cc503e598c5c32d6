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
// without a capture open, the capture valid afterwards).
static inline void b3w_device_wait() {
  void *p = nullptr;
  if (hipMalloc(&p, 256) == hipSuccess && p) { (void)hipFree(p); return; }
  (void)hipGetLastError();
  (void)hipDeviceSynchronize();
}
