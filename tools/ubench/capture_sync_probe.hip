// capture_sync_probe.hip — is "hipMalloc + hipFree of a few bytes" a device-wide wait on this ROCm (hipFree waits for every stream's work), and does
// it stay one — and stay legal — while another stream is capturing, under hipThreadExchangeStreamCaptureMode(relaxed)?  hipDeviceSynchronize is
// refused during a capture even then (capture_free_probe.hip) and invalidates the capture.
//   hipcc --offload-arch=gfx950 -O2 -w -o capture_sync_probe capture_sync_probe.hip && ./capture_sync_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void spin(long long cycles, int *out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) { }
  if (out) out[0] = 1;
}
static const char *E(hipError_t e) { return hipGetErrorName(e); }
int main() {
  for (int capturing = 0; capturing < 2; capturing++) {
    hipStream_t busy, cap;
    hipStreamCreateWithFlags(&busy, hipStreamNonBlocking); hipStreamCreateWithFlags(&cap, hipStreamNonBlocking);
    int *d = nullptr; hipMalloc(&d, 4); hipMemset(d, 0, 4); hipDeviceSynchronize();
    hipGraph_t g = nullptr;
    if (capturing) { printf("[capture open on another stream] begin: %s\n", E(hipStreamBeginCapture(cap, hipStreamCaptureModeGlobal))); hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, cap, 1000, nullptr); }
    else printf("[no capture]\n");
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    hipThreadExchangeStreamCaptureMode(&mode);                 // (everything below under the guard: a query of ANOTHER stream is refused in global mode too)
    hipLaunchKernelGGL(spin, dim3(1), dim3(1), 0, busy, 100000000ll * 3, d);      // wall_clock64 ticks at 100 MHz: 3 s
    printf("  the busy stream right after the launch: %s\n", E(hipStreamQuery(busy)));
    void *p = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e1 = hipMalloc(&p, 256), e2 = hipFree(p);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("  malloc + free under the relaxed guard: %s %s in %.2f s; the busy stream right after: %s\n", E(e1), E(e2), dt, E(hipStreamQuery(busy)));
    hipThreadExchangeStreamCaptureMode(&mode);
    if (capturing) printf("  end capture: %s\n", E(hipStreamEndCapture(cap, &g)));
    hipStreamSynchronize(busy);
    (void)hipGetLastError();
  }
  return 0;
}
