// probe.hip — measurement hooks behind bench.py's `roofline` object (gfx950).  Not on the walk's path.
//
//   srw_probe_request_rate : the ceiling the walk kernels are held against — dependent, uniformly random 16-byte reads over a table
//                            of the walk's own size, one chain per lane (the access pattern of k_walk_first_order: one record per step),
//                            measured on THIS box in about a second instead of being quoted from an earlier round's microbenchmark
//                            (tools/microbench_gather2.hip, profiles/r02_translation_and_request_rate.md: ~50 G L2-miss requests/s);
//   srw_result_scan_sums   : the fixed denominator of SURVEY §8(d): what the REFERENCE's algorithm reads for the walk that was just
//                            done — sum over the steps of deg(curr) (RandomSample.sample scans N(curr), RandomSample.scala:12-25) and,
//                            for second-order steps, of deg(prev) (the `exists` over N(prev), :27-44) — counted from the finished paths,
//                            whatever kernel produced them.
#include "engine.h"
#include "device_common.h"
#include "wave_primitives.h"

namespace srw {
namespace {
typedef int int4v __attribute__((ext_vector_type(4)));

__global__ void k_probe_fill(uint64_t *t, size_t n_rec) {
  for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < n_rec; r += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = r * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    t[r * 2] = x % n_rec; t[r * 2 + 1] = r;
  }
}
__global__ __launch_bounds__(256) void k_probe_chase(const uint64_t *__restrict__ t, size_t n_rec, int hops, uint64_t *out) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  uint64_t cur = (i * 0x9E3779B97F4A7C15ull) % n_rec, acc = 0;
  for (int h = 0; h < hops; ++h) {
    const int4v a = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(t + cur * 2));
    cur = ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x; acc += (uint32_t)a.z;
  }
  out[i] = acc + cur;
}

// one thread per walker: sum of deg(path[s - 1]) over the steps s = 1 .. len - 1, and of deg(path[s - 2]) over s >= 2
__global__ __launch_bounds__(256) void k_scan_sums(GraphView g, const int32_t *__restrict__ paths, const int32_t *__restrict__ lens,
                                                   int64_t n_walkers, int64_t stride, unsigned long long *out /* [3] */) {
  unsigned long long dc = 0, dp = 0, st = 0;
  for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n_walkers; w += (int64_t)gridDim.x * blockDim.x) {
    const int32_t len = lens[w];
    const int32_t *p = paths + w * stride;
    int32_t d_prev = 0;
    for (int32_t s = 1; s < len; ++s) {
      const int64_t slot = g.orig_id ? -1 : (int64_t)p[s - 1] - g.vmin;        // (compacted ids: the caller passes rank paths, see below)
      const int32_t d = (slot >= 0 && slot < g.n_slots) ? g.rows[slot].deg : 0;
      dc += (unsigned long long)d;
      if (s >= 2) dp += (unsigned long long)d_prev;
      d_prev = d;
      ++st;
    }
  }
  dc = wave_sum_u64(dc); dp = wave_sum_u64(dp); st = wave_sum_u64(st);
  if (lane_id() == 0) { atomicAdd(&out[0], dc); atomicAdd(&out[1], dp); atomicAdd(&out[2], st); }
}
}  // namespace

void probe_request_rate(srw_handle *h, int64_t table_bytes, double *reads_per_s, double *table_gib) {
  hipStream_t st = h->stream;
  size_t free_b = 0, total_b = 0;
  SRW_HIP(hipMemGetInfo(&free_b, &total_b));
  size_t want = table_bytes > 0 ? (size_t)table_bytes : (size_t)32 << 30;
  const size_t room = free_b > ((size_t)6 << 30) ? free_b - ((size_t)4 << 30) : free_b / 2;
  if (want > room) want = room;
  const size_t n_rec = want / 16;
  if (n_rec < 1024) throw Error(SRW_ERR_NOMEM, "no room for the request-rate probe's table");
  const size_t n_thr = (size_t)16 << 20;
  const int hops = 81;
  DevBuf<uint64_t> tab, out;
  tab.alloc(n_rec * 2); out.alloc(n_thr);
  hipLaunchKernelGGL(k_probe_fill, dim3(8192), dim3(256), 0, st, tab.p, n_rec);
  SRW_HIP(hipGetLastError());
  hipEvent_t e0, e1;
  SRW_HIP(hipEventCreate(&e0)); SRW_HIP(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {                 // rep 0 warms the TLBs
    SRW_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(k_probe_chase, dim3((unsigned)(n_thr / 256)), dim3(256), 0, st, tab.p, n_rec, hops, out.p);
    SRW_HIP(hipEventRecord(e1, st));
    SRW_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SRW_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  SRW_HIP(hipGetLastError());
  *reads_per_s = (double)n_thr * hops / ((double)best * 1e-3);
  if (table_gib) *table_gib = (double)(n_rec * 16) / (double)((size_t)1 << 30);
}

void result_scan_sums(srw_handle *h, int64_t *out3) {
  if (!h->res.valid) throw Error(SRW_ERR_INVALID, "no walk result on this handle");
  if (h->g.compact) throw Error(SRW_ERR_INVALID, "srw_result_scan_sums: not available on a compacted id space");
  hipStream_t st = h->stream;
  DevBuf<unsigned long long> acc; acc.alloc(3);
  SRW_HIP(hipMemsetAsync(acc.p, 0, 24, st));
  if (h->res.n_walkers > 0) {
    hipLaunchKernelGGL(k_scan_sums, dim3(h->n_cus * 8), dim3(256), 0, st, h->g.view(), h->res.paths.p, h->res.lens.p, h->res.n_walkers,
                       (int64_t)h->res.stride, acc.p);
    SRW_HIP(hipGetLastError());
  }
  unsigned long long v[3];
  SRW_HIP(hipMemcpyAsync(v, acc.p, 24, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  out3[0] = (int64_t)v[0]; out3[1] = (int64_t)v[1]; out3[2] = (int64_t)v[2];
}
}  // namespace srw
