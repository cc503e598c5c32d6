// vmm_va.hip — how much virtual address space can be reserved at once, how fast, and can one handle be mapped twice?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  for (uint64_t tib : {1ull, 4ull, 16ull, 64ull}) {
    void *va = nullptr;
    double t0 = now();
    hipError_t e = hipMemAddressReserve(&va, tib << 40, 1 << 21, nullptr, 0);
    printf("reserve %llu TiB: %s va=%p (%.3f ms)\n", (unsigned long long)tib, hipGetErrorString(e), va, (now() - t0) * 1e3);
    if (e == hipSuccess) {
      // map something at the very end of it
      hipMemAllocationProp prop = {};
      prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
      hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
      hipMemGenericAllocationHandle_t h;
      const size_t H = 256ull << 20;
      hipError_t e1 = hipMemCreate(&h, H, &prop, 0);
      uint8_t *end = (uint8_t *)va + (tib << 40) - H;
      hipError_t e2 = hipMemMap(end, H, 0, h, 0);
      hipError_t e3 = hipMemSetAccess(end, H, &acc, 1);
      hipError_t e4 = hipMemset(end, 1, H);
      hipError_t e5 = hipDeviceSynchronize();
      // second mapping of the same handle
      hipError_t e6 = hipMemMap(va, H, 0, h, 0);
      hipError_t e7 = e6 == hipSuccess ? hipMemSetAccess(va, H, &acc, 1) : e6;
      uint8_t b = 0;
      hipError_t e8 = e7 == hipSuccess ? hipMemcpy(&b, va, 1, hipMemcpyDeviceToHost) : e7;
      printf("   create %d map@end %d access %d memset %d sync %d | second mapping: map %d access %d read %d value %d\n", e1, e2, e3, e4, e5, e6, e7, e8, (int)b);
      (void)hipGetLastError();
      (void)hipMemUnmap(end, H);
      if (e6 == hipSuccess) (void)hipMemUnmap(va, H);
      (void)hipMemRelease(h);
      (void)hipMemAddressFree(va, tib << 40);
    }
    (void)hipGetLastError();
  }
  return 0;
}
