// vmm_remap.hip — does a virtual address that is unmapped and mapped to ANOTHER physical handle serve the new pages?
// X is written with 1s at VA1, unmapped; Y mapped at VA1 and written with 2s; then X and Y are mapped at fresh VAs and read.
// expected: X holds 1s, Y holds 2s.  Variants: with hipDeviceSynchronize / a dummy launch between unmap and map.
// build: hipcc --offload-arch=gfx950 -O3 -o vmm_remap vmm_remap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k_fill(uint32_t *p, uint32_t v, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_count(const uint32_t *p, uint32_t v, size_t n, unsigned long long *cnt) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] == v;
  atomicAdd(cnt, c);
}
int main() {
  const size_t H = 256ull << 20, N = H / 4;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  unsigned long long *cnt; CK(hipMalloc((void **)&cnt, 8));
  auto count = [&](void *p, uint32_t v) { unsigned long long h = 0; (void)hipMemset(cnt, 0, 8); hipLaunchKernelGGL(k_count, dim3(1024), dim3(256), 0, 0, (const uint32_t *)p, v, N, cnt); (void)hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost); return h; };
  for (int variant = 0; variant < 4; variant++) {
    hipMemGenericAllocationHandle_t X, Y;
    CK(hipMemCreate(&X, H, &prop, 0)); CK(hipMemCreate(&Y, H, &prop, 0));
    void *va1 = nullptr, *va2 = nullptr, *va3 = nullptr;
    CK(hipMemAddressReserve(&va1, H, 1 << 21, nullptr, 0));
    CK(hipMemMap(va1, H, 0, X, 0)); CK(hipMemSetAccess(va1, H, &acc, 1));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (uint32_t *)va1, 1u, N);
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(va1, H));
    if (variant == 1) CK(hipDeviceSynchronize());
    if (variant == 2) { hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, 0, (uint32_t *)cnt, 0u, (size_t)2); CK(hipDeviceSynchronize()); }
    if (variant == 3) { CK(hipMemAddressFree(va1, H)); CK(hipMemAddressReserve(&va1, H, 1 << 21, nullptr, 0)); }
    CK(hipMemMap(va1, H, 0, Y, 0)); CK(hipMemSetAccess(va1, H, &acc, 1));
    hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, (uint32_t *)va1, 2u, N);
    CK(hipDeviceSynchronize());
    const unsigned long long via_va1 = count(va1, 2);
    CK(hipMemAddressReserve(&va2, H, 1 << 21, nullptr, 0)); CK(hipMemAddressReserve(&va3, H, 1 << 21, nullptr, 0));
    CK(hipMemMap(va2, H, 0, X, 0)); CK(hipMemSetAccess(va2, H, &acc, 1));
    CK(hipMemUnmap(va1, H));
    CK(hipMemMap(va3, H, 0, Y, 0)); CK(hipMemSetAccess(va3, H, &acc, 1));
    printf("variant %d (va1=%p): read back through va1: %llu twos | X holds %llu ones, %llu twos | Y holds %llu twos  (of %zu)\n", variant, va1, via_va1,
           count(va2, 1), count(va2, 2), count(va3, 2), N);
    CK(hipMemUnmap(va2, H)); CK(hipMemUnmap(va3, H));
    CK(hipMemRelease(X)); CK(hipMemRelease(Y));
  }
  return 0;
}
