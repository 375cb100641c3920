// microbench_translation.hip — is the random-gather ceiling of MI355X an address-translation (UTCL1/UTCL2) limit?
//
//   A  "translation only": dependent chains over 16384 lines (1 MiB of data: L2-resident on every XCD) that are SPREAD
//      over n_pages pages at a given page stride inside a 32 GiB allocation.  The data-cache behaviour is the same for
//      every row of the sweep (all L2 hits); only the number of distinct pages (TLB reach) changes.  If translation were
//      the limit of the 32 GiB gather, the rate would collapse towards ~50 G/s as n_pages * stride approaches 32 GiB.
//   B  uniformly random reads over the whole 32 GiB at three request sizes: 16 B per lane (one 64-byte sector request per
//      lane), 64 B per 4 lanes, 128 B per 8 lanes (one full line per group) — requests/s vs bytes/s.
//   C  streaming read of the 32 GiB (coalesced 16-byte loads): bytes/s and the 128-byte-line request rate it implies.
//
// hipcc --offload-arch=gfx950 -O3 tools/microbench_translation.hip -o /tmp/mbt && /tmp/mbt
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef int int4v __attribute__((ext_vector_type(4)));

__device__ __host__ inline uint64_t mix(uint64_t x) {
  x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  return x;
}
__device__ __host__ inline size_t slot_addr16(uint64_t s, uint64_t n_pages, uint64_t page_stride) {   // in 16-byte units
  return (size_t)(((s % n_pages) * page_stride + (s / n_pages) * 64) >> 4);
}
__global__ void k_fill_spread(int4v *t, uint64_t n_lines, uint64_t n_pages, uint64_t page_stride) {
  for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < n_lines; s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t nx = mix(s + 1) % n_lines;
    int4v v; v.x = (int)(uint32_t)nx; v.y = (int)(uint32_t)(nx >> 32); v.z = 1; v.w = 0;
    t[slot_addr16(s, n_pages, page_stride)] = v;
  }
}
__global__ __launch_bounds__(256) void k_chase_spread(const int4v *__restrict__ t, uint64_t n_lines, uint64_t n_pages,
                                                      uint64_t page_stride, int hops, uint64_t *out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  uint64_t cur = mix(i) % n_lines, acc = 0;
  for (int h = 0; h < hops; ++h) {
    const int4v a = t[slot_addr16(cur, n_pages, page_stride)];
    cur = ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x; acc += (uint32_t)a.z;
  }
  out[i] = cur + acc;
}
__global__ void k_fill_rand(int4v *t, size_t n16) {
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < n16; r += (size_t)gridDim.x * blockDim.x) {
    const uint64_t x = mix(r);
    int4v v; v.x = (int)(uint32_t)x; v.y = (int)(uint32_t)(x >> 32); v.z = 1; v.w = 0;
    t[r] = v;
  }
}
// GROUP lanes read GROUP*16 contiguous bytes at a random (GROUP*16)-aligned address; the next address comes from lane 0
// of the group's record (dependent chain per group).
template <int GROUP>
__global__ __launch_bounds__(256) void k_chase_rand(const int4v *__restrict__ t, size_t n16, int hops, uint64_t *out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t grp = i / GROUP; const int sub = (int)(i % GROUP);
  const size_t n_units = n16 / GROUP;
  uint64_t cur = mix(grp) % n_units, acc = 0;
  for (int h = 0; h < hops; ++h) {
    const int4v a = t[cur * GROUP + sub];
    uint64_t nx = ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x;
    if (GROUP > 1) {   // everybody follows sub-lane 0's link
      const int src = (int)(threadIdx.x & 63) & ~(GROUP - 1);
      const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)nx, src), hi = (uint32_t)__shfl((int)(uint32_t)(nx >> 32), src);
      nx = ((uint64_t)hi << 32) | lo;
    }
    cur = nx % n_units; acc += (uint32_t)a.z;
  }
  out[i] = cur + acc;
}
__global__ __launch_bounds__(256) void k_stream(const int4v *__restrict__ t, size_t n16, uint64_t *out) {
  uint64_t acc = 0;
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < n16; r += (size_t)gridDim.x * blockDim.x) {
    const int4v a = __builtin_nontemporal_load(t + r);
    acc += (uint32_t)a.x + (uint32_t)a.z;
  }
  out[blockIdx.x * (size_t)blockDim.x + threadIdx.x] = acc;
}

template <class F>
static float best_of(int reps, hipEvent_t e0, hipEvent_t e1, F f) {
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t bytes = 32ull << 30, n16 = bytes / 16;
  int4v *t; if (hipMalloc(&t, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  const size_t n_thr = 16ull << 20;
  uint64_t *out; hipMalloc(&out, n_thr * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("table %p (offset inside a 2 MiB page: %zu, inside 1 GiB: %zu)\n", (void *)t, (size_t)((uintptr_t)t & ((2u << 20) - 1)),
         (size_t)((uintptr_t)t & ((1ull << 30) - 1)));
  const int hops = 64;
  // ---- A: translation only ----
  printf("\nA. 16384 L2-resident lines spread over n_pages pages (same cache behaviour, growing translation footprint)\n");
  const uint64_t n_lines = 16384;
  const uint64_t strides[] = {4096, 65536, 2ull << 20, 1ull << 30};
  for (uint64_t stride : strides) {
    for (uint64_t n_pages = 1; n_pages <= n_lines; n_pages *= 4) {
      if (n_pages * stride > bytes) break;
      if ((n_lines / n_pages) * 64 > stride) continue;   // the lines of one page must fit its stride
      hipMemset(t, 0, 64);   // (no need to clear the whole table: only the slots are read)
      k_fill_spread<<<64, 256>>>(t, n_lines, n_pages, stride); hipDeviceSynchronize();
      const float ms = best_of(3, e0, e1, [&] { k_chase_spread<<<n_thr / 256, 256>>>(t, n_lines, n_pages, stride, hops, out); });
      printf("  stride %10llu B  pages %6llu  span %8.1f MiB : %7.2f ms -> %6.1f G reads/s\n", (unsigned long long)stride,
             (unsigned long long)n_pages, (double)n_pages * stride / 1048576.0, ms, (double)n_thr * hops / ms / 1e6);
    }
  }
  // ---- B: random over 32 GiB at three request sizes ----
  printf("\nB. uniformly random dependent reads over 32 GiB\n");
  k_fill_rand<<<8192, 256>>>(t, n16); hipDeviceSynchronize();
  {
    float ms = best_of(3, e0, e1, [&] { k_chase_rand<1><<<n_thr / 256, 256>>>(t, n16, hops, out); });
    double req = (double)n_thr * hops / ms / 1e6;
    printf("  16 B per lane  (64-B sector requests): %7.2f ms -> %6.1f G requests/s, %6.2f TB/s of sectors\n", ms, req, req * 64 / 1e3);
    ms = best_of(3, e0, e1, [&] { k_chase_rand<4><<<n_thr / 256, 256>>>(t, n16, hops, out); });
    req = (double)n_thr / 4 * hops / ms / 1e6;
    printf("  64 B per 4 lanes (one sector per group): %7.2f ms -> %6.1f G requests/s, %6.2f TB/s\n", ms, req, req * 64 / 1e3);
    ms = best_of(3, e0, e1, [&] { k_chase_rand<8><<<n_thr / 256, 256>>>(t, n16, hops, out); });
    req = (double)n_thr / 8 * hops / ms / 1e6;
    printf("  128 B per 8 lanes (one line per group):  %7.2f ms -> %6.1f G requests/s, %6.2f TB/s\n", ms, req, req * 128 / 1e3);
  }
  // ---- C: streaming ----
  printf("\nC. streaming read of 32 GiB\n");
  {
    const float ms = best_of(3, e0, e1, [&] { k_stream<<<256 * 32, 256>>>(t, n16, out); });
    const double tbs = (double)bytes / ms / 1e9;
    printf("  %7.2f ms -> %5.2f TB/s = %5.1f G 128-B line requests/s = %5.1f G 64-B sectors/s\n", ms, tbs, tbs * 1e3 / 128, tbs * 1e3 / 64);
  }
  return 0;
}
