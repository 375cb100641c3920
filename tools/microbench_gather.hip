// microbench_gather.hip — ceiling of the first-order walk's access pattern on MI355X: every lane chases a chain of
// dependent, uniformly random, 32-byte-aligned 32-byte reads (L1-bypassing) over a table far larger than
// L2 + Infinity Cache.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench_gather.hip -o /tmp/mg && /tmp/mg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int int4v __attribute__((ext_vector_type(4)));
__global__ void k_fill(uint64_t *t, size_t n_rec) {   // record r: word0 = next record index (a random permutation-ish hash)
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < n_rec; r += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = r * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    t[r * 4] = x % n_rec; t[r * 4 + 1] = r; t[r * 4 + 2] = 0; t[r * 4 + 3] = 0;
  }
}
template <bool NT, int RECB>
__global__ __launch_bounds__(256) void k_chase(const uint64_t *__restrict__ t, size_t n_rec, int hops, uint64_t *out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  uint64_t cur = (i * 0x9E3779B97F4A7C15ull) % n_rec, acc = 0;
  for (int h = 0; h < hops; ++h) {
    const int4v *p = reinterpret_cast<const int4v *>(t + cur * 4);
    int4v a = NT ? __builtin_nontemporal_load(p) : *p;
    uint64_t nxt = ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x;
    if (RECB == 32) { int4v b = NT ? __builtin_nontemporal_load(p + 1) : p[1]; acc += (uint32_t)b.x; }
    acc += (uint32_t)a.z;
    cur = nxt + (acc & 0);   // dependent on both halves
  }
  out[i] = cur + acc;
}
int main(int argc, char **argv) {
  size_t gb = argc > 1 ? atol(argv[1]) : 64; int hops = 81;
  size_t n_rec = gb * (1ull << 30) / 32, n_thr = 32ull << 20;
  uint64_t *t, *out; hipMalloc(&t, n_rec * 32); hipMalloc(&out, n_thr * 8);
  k_fill<<<8192, 256>>>(t, n_rec); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 4; ++v) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (v == 0) k_chase<true, 32><<<n_thr / 256, 256>>>(t, n_rec, hops, out);
      if (v == 1) k_chase<false, 32><<<n_thr / 256, 256>>>(t, n_rec, hops, out);
      if (v == 2) k_chase<true, 16><<<n_thr / 256, 256>>>(t, n_rec, hops, out);
      if (v == 3) k_chase<false, 16><<<n_thr / 256, 256>>>(t, n_rec, hops, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double acc = (double)n_thr * hops;
    printf("table %zu GiB, %s loads, %d B used per record: %.2f ms -> %.2f G random records/s (%.0f GB/s useful, %.2f TB/s at 64 B/sector)\n",
           gb, (v & 1) ? "default" : "nontemporal", v < 2 ? 32 : 16, best, acc / best / 1e6, acc * (v < 2 ? 32 : 16) / best / 1e6, acc * 64 / best / 1e9);
  }
  return 0;
}
