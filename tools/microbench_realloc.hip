// microbench_realloc.hip — what does re-allocating memory this process released cost (bench.py's configurations follow each other in
// one process: config 3's 138 GB of tables land on pages the headline graph just freed), by allocator?
// hipcc --offload-arch=gfx950 -O3 tools/microbench_realloc.hip -o /tmp/mbr && /tmp/mbr [GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_fill(uint4 *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}
static double fill(void *p, size_t bytes) {
  const double t = now();
  hipLaunchKernelGGL(k_fill, dim3(256 * 32), dim3(256), 0, 0, (uint4 *)p, bytes / 16);
  CK(hipDeviceSynchronize());
  return (now() - t) * 1e3;
}
int main(int argc, char **argv) {
  const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 64) << 30;
  CK(hipSetDevice(0));
  for (int rep = 0; rep < 3; ++rep) {
    void *a = nullptr;
    double t = now(); CK(hipMalloc(&a, bytes)); const double ta = (now() - t) * 1e3;
    const double f1 = fill(a, bytes), f2 = fill(a, bytes);
    t = now(); CK(hipFree(a)); const double tf = (now() - t) * 1e3;
    printf("hipMalloc #%d: %.1f ms (%.1f ms/GB), fill %.1f / %.1f ms, hipFree %.1f ms\n", rep, ta, ta / (bytes / 1e9), f1, f2, tf);
  }
  hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  for (int rep = 0; rep < 3; ++rep) {
    void *va = nullptr; hipMemGenericAllocationHandle_t hd;
    double t = now();
    CK(hipMemAddressReserve(&va, bytes, (size_t)2 << 20, nullptr, 0));
    CK(hipMemCreate(&hd, bytes, &prop, 0));
    CK(hipMemMap(va, bytes, 0, hd, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    const double ta = (now() - t) * 1e3;
    const double f1 = fill(va, bytes), f2 = fill(va, bytes);
    t = now();
    CK(hipMemUnmap(va, bytes)); CK(hipMemRelease(hd)); CK(hipMemAddressFree(va, bytes));
    printf("VMM one handle #%d: %.1f ms (%.1f ms/GB), fill %.1f / %.1f ms, release %.1f ms\n", rep, ta, ta / (bytes / 1e9), f1, f2, (now() - t) * 1e3);
  }
  for (int rep = 0; rep < 2; ++rep) {
    void *a = nullptr;
    double t = now(); CK(hipMalloc(&a, bytes)); const double ta = (now() - t) * 1e3;
    const double f1 = fill(a, bytes);
    CK(hipFree(a));
    printf("hipMalloc again #%d: %.1f ms (%.1f ms/GB), fill %.1f ms\n", rep, ta, ta / (bytes / 1e9), f1);
  }
  return 0;
}
