// microbench_vmm.hip — VERDICT r02 item 6(a) / item 4: does an explicit virtual-memory placement (hipMemAddressReserve /
// hipMemCreate / hipMemMap at the largest granule the driver offers, VA aligned to 1 GiB) change (1) what an allocation COSTS
// (hipMalloc was measured at ~35 ms/GB: most of the cold start of the biased configs after the table build itself) and
// (2) the random-gather request rate of a 32 GiB table (the headline kernel's bound)?
//
// hipcc --offload-arch=gfx950 -O3 tools/microbench_vmm.hip -o /tmp/mbv && /tmp/mbv [GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int int4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CKV(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return -1.0; } } while (0)

__device__ __host__ inline uint64_t mix(uint64_t x) { x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; return x; }
__global__ void k_fill(int4v *t, size_t n16) {
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < n16; r += (size_t)gridDim.x * blockDim.x) {
    const uint64_t x = mix(r);
    int4v v; v.x = (int)(uint32_t)x; v.y = (int)(uint32_t)(x >> 32); v.z = 1; v.w = 0;
    t[r] = v;
  }
}
__global__ __launch_bounds__(256) void k_chase(const int4v *__restrict__ t, size_t n16, int hops, uint64_t *out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  uint64_t cur = mix(i) % n16, acc = 0;
  for (int h = 0; h < hops; ++h) {
    const int4v a = __builtin_nontemporal_load(t + cur);
    cur = (((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x) % n16; acc += (uint32_t)a.z;
  }
  out[i] = cur + acc;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double gather_rate(void *p, size_t bytes) {
  const size_t n16 = bytes / 16;
  for (int rep = 0; rep < 2; ++rep) {       // first touch vs second pass: is any allocation cost deferred to the first access?
    const double t0 = now();
    hipLaunchKernelGGL(k_fill, dim3(256 * 32), dim3(256), 0, 0, (int4v *)p, n16);
    CKV(hipDeviceSynchronize());
    printf("  fill pass %d: %.1f ms\n", rep, (now() - t0) * 1e3);
  }
  const size_t lanes = (size_t)16 << 20; const int hops = 64;
  uint64_t *out = nullptr; CKV(hipMalloc((void **)&out, lanes * 8));
  double best = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t a, b; CKV(hipEventCreate(&a)); CKV(hipEventCreate(&b));
    CKV(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k_chase, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, (const int4v *)p, n16, hops, out);
    CKV(hipEventRecord(b, 0)); CKV(hipEventSynchronize(b));
    float ms = 0; CKV(hipEventElapsedTime(&ms, a, b));
    best = std::max(best, (double)lanes * hops / (ms * 1e-3) / 1e9);
  }
  (void)hipFree(out);
  return best;
}

int main(int argc, char **argv) {
  const size_t gib = argc > 1 ? (size_t)atoll(argv[1]) : 32;
  const size_t bytes = gib << 30;
  CK(hipSetDevice(0));
  CK(hipFree(0));
  // ---- 1. plain hipMalloc, twice (is a second allocation of freed memory cheaper?) ----
  for (int rep = 0; rep < 2; ++rep) {
    void *p = nullptr;
    double t0 = now(); CK(hipMalloc(&p, bytes)); double t1 = now();
    printf("hipMalloc(%zu GiB) #%d: %.1f ms (%.1f ms/GB), VA %p (offset in 2 MiB: %zu, in 1 GiB: %zu)\n", gib, rep, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / (bytes / 1e9), p,
           (size_t)((uintptr_t)p & ((2u << 20) - 1)), (size_t)((uintptr_t)p & (((size_t)1 << 30) - 1)));
    printf("  random 16-byte dependent gathers over it: %.1f G requests/s\n", gather_rate(p, bytes));
    t0 = now(); CK(hipFree(p)); t1 = now();
    printf("  hipFree: %.1f ms\n", (t1 - t0) * 1e3);
  }
  // ---- 2. stream-ordered pool ----
  {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
    uint64_t thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
    for (int rep = 0; rep < 2; ++rep) {
      void *p = nullptr;
      double t0 = now(); CK(hipMallocAsync(&p, bytes, st)); CK(hipStreamSynchronize(st)); double t1 = now();
      printf("hipMallocAsync(%zu GiB) #%d: %.1f ms\n", gib, rep, (t1 - t0) * 1e3);
      t0 = now(); CK(hipFreeAsync(p, st)); CK(hipStreamSynchronize(st)); t1 = now();
      printf("  hipFreeAsync: %.1f ms\n", (t1 - t0) * 1e3);
    }
    uint64_t zero = 0; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &zero));
    CK(hipMemPoolTrimTo(pool, 0));
    CK(hipStreamDestroy(st));
  }
  // ---- 3. explicit VMM: reserve (1 GiB aligned) + create + map ----
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("VMM granularity: minimum %zu B, recommended %zu B\n", gmin, grec);
  for (size_t chunk : {(size_t)0 /* one handle */, (size_t)1 << 30, (size_t)2 << 20}) {
    const size_t csz = chunk ? chunk : bytes;
    if (csz % gmin) { printf("chunk %zu not a multiple of the granularity\n", csz); continue; }
    const size_t n_chunks = bytes / csz;
    if (n_chunks > 20000) { printf("chunk %zu: too many handles, skipped\n", csz); continue; }
    void *va = nullptr;
    double t0 = now();
    CK(hipMemAddressReserve(&va, bytes, (size_t)1 << 30, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> hs(n_chunks);
    for (size_t i = 0; i < n_chunks; ++i) CK(hipMemCreate(&hs[i], csz, &prop, 0));
    double t1 = now();
    for (size_t i = 0; i < n_chunks; ++i) CK(hipMemMap((char *)va + i * csz, csz, 0, hs[i], 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    double t2 = now();
    printf("VMM %zu x %zu MiB: create %.1f ms, map + access %.1f ms (total %.1f ms/GB), VA %p (offset in 1 GiB: %zu)\n", n_chunks, csz >> 20, (t1 - t0) * 1e3, (t2 - t1) * 1e3,
           (t2 - t0) * 1e3 / (bytes / 1e9), va, (size_t)((uintptr_t)va & (((size_t)1 << 30) - 1)));
    printf("  random 16-byte dependent gathers over it: %.1f G requests/s\n", gather_rate(va, bytes));
    t0 = now();
    CK(hipMemUnmap(va, bytes));
    for (size_t i = 0; i < n_chunks; ++i) CK(hipMemRelease(hs[i]));
    CK(hipMemAddressFree(va, bytes));
    printf("  unmap + release: %.1f ms\n", (now() - t0) * 1e3);
  }
  // ---- 4. one handle per buffer, allocate -> touch -> free -> allocate again (what a per-buffer VMM allocator would do) ----
  for (int rep = 0; rep < 3; ++rep) {
    void *va = nullptr; hipMemGenericAllocationHandle_t hd;
    double t0 = now();
    CK(hipMemAddressReserve(&va, bytes, (size_t)2 << 20, nullptr, 0));
    CK(hipMemCreate(&hd, bytes, &prop, 0));
    CK(hipMemMap(va, bytes, 0, hd, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    double t1 = now();
    printf("VMM single handle #%d: %.1f ms\n", rep, (t1 - t0) * 1e3);
    const double t2 = now();
    hipLaunchKernelGGL(k_fill, dim3(256 * 32), dim3(256), 0, 0, (int4v *)va, bytes / 16);
    CK(hipDeviceSynchronize());
    printf("  first fill: %.1f ms\n", (now() - t2) * 1e3);
    t0 = now();
    CK(hipMemUnmap(va, bytes)); CK(hipMemRelease(hd)); CK(hipMemAddressFree(va, bytes));
    printf("  unmap + release: %.1f ms\n", (now() - t0) * 1e3);
  }
  return 0;
}
