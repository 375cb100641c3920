// Scratch: does the stream-ordered allocator keep pages across free / allocate inside one process (release threshold = max), and can it
// serve DIFFERENT sizes from what it kept?  hipcc --offload-arch=gfx950 -O2 tools/microbench_pool.hip -o /tmp/mb_pool && /tmp/mb_pool [GiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char *p, size_t n) { size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4096; if (i < n) p[i] = 1; }
int main(int argc, char **argv) {
  const size_t G = (size_t)1 << 30, big = (size_t)(argc > 1 ? atoi(argv[1]) : 160) * G;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
  uint64_t thr = UINT64_MAX; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
  auto A = [&](size_t n, const char *what) -> void * {
    void *p = nullptr; double t0 = now();
    hipError_t e = hipMallocAsync(&p, n, st); hipStreamSynchronize(st);
    printf("%-44s %7.1f GiB: %8.1f ms %s\n", what, (double)n / G, now() - t0, e == hipSuccess ? "" : hipGetErrorString(e)); fflush(stdout);
    if (e == hipSuccess) { hipLaunchKernelGGL(touch, dim3((unsigned)((n / 4096 + 255) / 256)), dim3(256), 0, st, (char *)p, n); hipStreamSynchronize(st); }
    return e == hipSuccess ? p : nullptr;
  };
  auto F = [&](void *p) { double t0 = now(); hipFreeAsync(p, st); hipStreamSynchronize(st); printf("  free %.1f ms\n", now() - t0); };
  // plain hipMalloc for reference
  { void *p; double t0 = now(); CK(hipMalloc(&p, big)); printf("hipMalloc #0 %.1f ms\n", now() - t0); t0 = now(); CK(hipFree(p)); printf("hipFree %.1f ms\n", now() - t0);
    t0 = now(); CK(hipMalloc(&p, big)); printf("hipMalloc #1 (same size, after free) %.1f ms\n", now() - t0); CK(hipFree(p)); }
  void *a = A(big, "mallocAsync first");
  F(a);
  void *b = A(big, "mallocAsync again, same size");
  F(b);
  void *c = A(big * 3 / 4, "mallocAsync 3/4 of it");
  void *d = A(big / 8, "mallocAsync 1/8 beside it");
  F(c); F(d);
  void *e = A(big + 8 * G, "mallocAsync bigger than anything kept");
  if (e) F(e);
  size_t fr, tot; hipMemGetInfo(&fr, &tot); printf("hipMemGetInfo free %.1f GiB of %.1f while the pool holds its pages\n", (double)fr / G, (double)tot / G);
  uint64_t resv = 0, used = 0; hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &resv); hipMemPoolGetAttribute(pool, hipMemPoolAttrUsedMemCurrent, &used);
  printf("pool reserved %.1f GiB used %.1f GiB\n", (double)resv / G, (double)used / G);
  { void *p; double t0 = now(); hipError_t er = hipMalloc(&p, big); printf("hipMalloc beside the full pool: %.1f ms %s\n", now() - t0, hipGetErrorString(er)); if (er == hipSuccess) hipFree(p); }
  double t0 = now(); CK(hipMemPoolTrimTo(pool, 0)); printf("trim to 0: %.1f ms\n", now() - t0);
  hipMemGetInfo(&fr, &tot); printf("free after trim %.1f GiB\n", (double)fr / G);
  return 0;
}
