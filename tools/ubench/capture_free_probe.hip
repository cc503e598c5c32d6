// capture_free_probe.hip — what does this ROCm do with hipFree / hipMemRelease / hipStreamDestroy / hipEventDestroy while ANOTHER stream
// of the same thread is capturing in global mode, with and without hipThreadExchangeStreamCaptureMode(relaxed) around the call?
//   hipcc --offload-arch=gfx950 -O2 -o capture_free_probe capture_free_probe.hip && ./capture_free_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(int *p) { p[threadIdx.x] += 1; }
static const char *E(hipError_t e) { return hipGetErrorName(e); }
int main() {
  for (int relaxed = 0; relaxed < 2; relaxed++) {
    int *a = nullptr, *b = nullptr;
    hipStream_t s, t;
    hipEvent_t ev;
    hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20);
    hipStreamCreate(&s); hipStreamCreate(&t); hipEventCreate(&ev);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, a);
    hipStreamSynchronize(s);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    printf("[%s] begin capture: %s\n", relaxed ? "relaxed guard" : "no guard", E(e));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, a);
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    if (relaxed) printf("  exchange -> relaxed: %s\n", E(hipThreadExchangeStreamCaptureMode(&mode)));
    printf("  hipFree(other buffer): %s\n", E(hipFree(b)));
    printf("  hipEventDestroy: %s\n", E(hipEventDestroy(ev)));
    printf("  hipStreamSynchronize(other stream): %s\n", E(hipStreamSynchronize(t)));
    if (getenv("PROBE_DEVICE_SYNC")) { printf("  hipDeviceSynchronize: "); fflush(stdout); printf("%s\n", E(hipDeviceSynchronize())); }
    {                                                          // the placement allocator's pieces: VMM handles
      hipMemAllocationProp prop = {};
      prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
      size_t gran = 0;
      hipMemGenericAllocationHandle_t h;
      void *va = nullptr;
      hipError_t e1 = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
      hipError_t e2 = hipMemCreate(&h, gran, &prop, 0);
      hipError_t e3 = hipMemAddressReserve(&va, gran, 0, nullptr, 0);
      hipError_t e4 = hipMemMap(va, gran, 0, h, 0);
      hipError_t e5 = hipMemUnmap(va, gran);
      hipError_t e6 = hipMemRelease(h);
      hipError_t e7 = hipMemAddressFree(va, gran);
      printf("  VMM granularity/create/reserve/map/unmap/release/addressfree: %s %s %s %s %s %s %s\n", E(e1), E(e2), E(e3), E(e4), E(e5), E(e6), E(e7));
    }
    printf("  hipStreamDestroy(other stream): %s\n", E(hipStreamDestroy(t)));
    int *c = nullptr;
    printf("  hipMalloc: %s\n", E(hipMalloc(&c, 1 << 20)));
    if (relaxed) printf("  exchange back: %s\n", E(hipThreadExchangeStreamCaptureMode(&mode)));
    e = hipStreamEndCapture(s, &g);
    printf("  end capture: %s (graph %p)\n", E(e), (void *)g);
    (void)hipGetLastError();
    if (g) hipGraphDestroy(g);
    hipStreamDestroy(s);
    hipFree(a);
    if (c) hipFree(c);
    hipDeviceSynchronize();
    (void)hipGetLastError();
  }
  return 0;
}
