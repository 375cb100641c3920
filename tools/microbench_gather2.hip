// microbench_gather2.hip — where does the random-gather ceiling of MI355X come from?  Sweeps the table size (L2-resident
// -> Infinity-Cache-resident -> HBM) for dependent chains of random 16-byte reads at 16-byte stride, with 1 or 2
// independent chains per lane.  hipcc --offload-arch=gfx950 -O3 tools/microbench_gather2.hip -o /tmp/mg2 && /tmp/mg2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int int4v __attribute__((ext_vector_type(4)));
__global__ void k_fill(uint64_t *t, size_t n_rec) {
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < n_rec; r += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = r * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    t[r * 2] = x % n_rec; t[r * 2 + 1] = r;
  }
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k_chase(const uint64_t *__restrict__ t, size_t n_rec, int hops, uint64_t *out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  uint64_t cur[CHAINS], acc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) cur[c] = ((i * CHAINS + c) * 0x9E3779B97F4A7C15ull) % n_rec;
  for (int h = 0; h < hops; ++h) {
    int4v a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = *reinterpret_cast<const int4v *>(t + cur[c] * 2);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) { cur[c] = ((uint64_t)(uint32_t)a[c].y << 32) | (uint32_t)a[c].x; acc += (uint32_t)a[c].z; }
  }
  uint64_t s = acc;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += cur[c];
  out[i] = s;
}
int main(int argc, char **argv) {
  const int hops = 81;
  const size_t n_thr = 32ull << 20;
  uint64_t *out; hipMalloc(&out, n_thr * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t sizes_mb[] = {2, 16, 64, 192, 1024, 8192, 32768};
  for (size_t mb : sizes_mb) {
    size_t n_rec = mb * (1ull << 20) / 16;
    uint64_t *t; if (hipMalloc(&t, n_rec * 16) != hipSuccess) { printf("alloc %zu MB failed\n", mb); continue; }
    k_fill<<<8192, 256>>>(t, n_rec); hipDeviceSynchronize();
    for (int v = 0; v < 2; ++v) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (v == 0) k_chase<1><<<n_thr / 256, 256>>>(t, n_rec, hops, out);
        if (v == 1) k_chase<2><<<n_thr / 512, 256>>>(t, n_rec, hops, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("table %6zu MiB, %d chain(s)/lane: %.2f ms -> %.1f G random 16-B reads/s\n", mb, v + 1, best, (double)n_thr * hops / best / 1e6);
    }
    hipFree(t);
  }
  return 0;
}
