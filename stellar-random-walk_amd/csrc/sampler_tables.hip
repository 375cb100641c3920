// sampler_tables.hip — first-order exact-CDF + guide tables, built on the device (gfx950).
//
// For p == q == 1 computeSecondOrderWeights returns its input bit for bit (x / 1.0f == x,
// M/algorithm/RandomSample.scala:33-39), so every step samples the RAW neighbor list and the reference's
// running CDF  acc_k  (RandomSample.scala:18-20) is a fixed function of the vertex.  It is precomputed here
// with the reference's own operation order — S by the certified-exact / sequential sum, acc by the
// sequential chain of wave_primitives.h — and stored next to the neighbor id.  The guide table makes the
// "first k with acc_k >= p" search O(1) expected instead of O(deg) without changing its answer:
//   guide[j] = first k with acc_k >= ceil(j * 2^24 / deg) * 2^-24
// and any draw p whose bucket floor(floor(p * 2^24) * deg / 2^24) is j satisfies p >= that threshold, so the
// reference's first crossing index is >= guide[j]; acc is non-decreasing on a regular row (all w >= 0,
// 0 < S < inf), so scanning forward from guide[j] finds exactly that index.
// Rows with a negative / NaN weight or a non-positive / non-finite sum are flagged ROW_IRREGULAR and
// sampled by the literal sequential scan.
#include "engine.h"
#include "sampling.h"

namespace srw {
namespace {

constexpr int SMALL_DEG = 32;

__device__ inline bool weight_regular(float w) { return w >= 0.0f; }  // false for NaN and negatives

// deg <= SMALL_DEG: one lane per row, literal sequential evaluation.
__global__ void k_fo_small(Row *rows, const Ent *__restrict__ ent, FoEnt *__restrict__ fo, int64_t n_slots) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x) {
    Row r = rows[v];
    if (r.deg <= 0 || r.deg > SMALL_DEG) continue;
    const Ent *row = ent + r.off;
    FoEnt *out = fo + r.off;
    double sum = 0.0;
    bool irr = false;
    for (int32_t k = 0; k < r.deg; ++k) {
      float w = row[k].w;
      sum = sum + (double)w;
      irr |= !weight_regular(w);
    }
    irr |= !(sum > 0.0) || isinf(sum);
    double acc = 0.0;
    for (int32_t k = 0; k < r.deg; ++k) {
      Ent e = row[k];
      acc += (double)e.w / sum;
      FoEnt f; f.cdf = acc; f.id = e.id; f.guide = 0; f.noff = 0; f.ndeg = 0; f.nflags = 0;
      out[k] = f;
    }
    if (irr) rows[v].flags = r.flags | ROW_IRREGULAR;
  }
}

// deg > SMALL_DEG: one wave per row.
__global__ void k_fo_large(Row *rows, const Ent *__restrict__ ent, FoEnt *__restrict__ fo, int64_t n_slots) {
  const int lane = lane_id();
  int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t v = wave; v < n_slots; v += n_waves) {
    Row r = rows[v];
    if (r.deg <= SMALL_DEG) continue;
    const Ent *row = ent + r.off;
    FoEnt *out = fo + r.off;
    double part = 0.0;
    SumCert cert;
    bool irr = false;
    for (int32_t base = 0; base < r.deg; base += 64) {
      int32_t k = base + lane;
      if (k < r.deg) {
        float w = row[k].w;
        part += (double)w;
        cert.add(w);
        irr |= !weight_regular(w);
      }
    }
    int emin = wave_min_i32(cert.emin), emax = wave_max_i32(cert.emax);
    bool bad = __any(cert.bad);
    irr = __any(irr);
    double S;
    if (sum_is_exact(emin, emax, bad, r.deg)) {
      S = wave_sum_f64(part);
    } else {
      S = 0.0;
      for (int32_t base = 0; base < r.deg; base += 64) {
        int32_t k = base + lane;
        double wd = (k < r.deg) ? (double)row[k].w : 0.0;
        int cnt = min(64, r.deg - base);
        for (int i = 0; i < cnt; ++i) S = S + readlane_f64(wd, i);
      }
    }
    irr |= !(S > 0.0) || isinf(S);
    double acc = 0.0;
    for (int32_t base = 0; base < r.deg; base += 64) {
      int32_t k = base + lane;
      Ent e; e.id = 0; e.w = 0.0f;
      if (k < r.deg) e = row[k];
      double d = (k < r.deg) ? (double)e.w / S : 0.0;
      int cnt = min(64, r.deg - base);
      double mine = 0.0;
      for (int i = 0; i < cnt; ++i) {
        acc = acc + readlane_f64(d, i);
        if (lane == i) mine = acc;
      }
      if (k < r.deg) { FoEnt f; f.cdf = mine; f.id = e.id; f.guide = 0; f.noff = 0; f.ndeg = 0; f.nflags = 0; out[k] = f; }
    }
    if (irr && lane == 0) rows[v].flags = r.flags | ROW_IRREGULAR;
  }
}

__device__ inline double bucket_threshold(uint32_t j, uint32_t deg) {
  uint64_t m = (((uint64_t)j << 24) + deg - 1) / deg;  // ceil(j * 2^24 / deg)
  return (double)m * (1.0 / 16777216.0);               // exact
}

__global__ void k_guide_small(const Row *__restrict__ rows, FoEnt *__restrict__ fo, int64_t n_slots) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x) {
    Row r = rows[v];
    if (r.deg <= 0 || r.deg > SMALL_DEG || (r.flags & ROW_IRREGULAR)) continue;
    FoEnt *row = fo + r.off;
    int32_t k = 0;
    for (int32_t j = 0; j < r.deg; ++j) {   // thresholds and cdf both non-decreasing: two-pointer merge
      double t = bucket_threshold((uint32_t)j, (uint32_t)r.deg);
      while (k < r.deg && !(row[k].cdf >= t)) ++k;
      row[j].guide = k;
    }
  }
}

__global__ void k_guide_large(const Row *__restrict__ rows, FoEnt *__restrict__ fo, int64_t n_slots) {
  const int lane = lane_id();
  int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t v = wave; v < n_slots; v += n_waves) {
    Row r = rows[v];
    if (r.deg <= SMALL_DEG || (r.flags & ROW_IRREGULAR)) continue;
    FoEnt *row = fo + r.off;
    for (int32_t j = lane; j < r.deg; j += 64) {
      double t = bucket_threshold((uint32_t)j, (uint32_t)r.deg);
      int32_t lo = 0, hi = r.deg;
      while (lo < hi) {
        int32_t mid = lo + ((hi - lo) >> 1);
        if (row[mid].cdf >= t) hi = mid; else lo = mid + 1;
      }
      row[j].guide = lo;
    }
  }
}

// Link pass: copy the row descriptor of every neighbor into its record (after the irregular flags are final).
__global__ void k_fo_link(const Row *__restrict__ rows, FoEnt *__restrict__ fo, int64_t n_entries, int32_t vmin,
                          int64_t n_slots) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_entries; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t s = (int64_t)fo[e].id - vmin;
    Row r; r.off = 0; r.deg = 0; r.flags = 0;
    if (s >= 0 && s < n_slots) r = rows[s];
    fo[e].noff = r.off; fo[e].ndeg = r.deg; fo[e].nflags = r.flags;
  }
}

}  // namespace

void build_first_order_tables(srw_handle *h) {
  Graph &g = h->g;
  if (g.has_fo) return;
  hipStream_t st = h->stream;
  g.fo.alloc((size_t)g.n_entries);
  int64_t tb = (g.n_slots + 255) / 256;
  int gs = (int)std::min<int64_t>(std::max<int64_t>(tb, 1), 256 * 32);
  int64_t wb = (g.n_slots + 3) / 4;  // 4 waves per block
  int gl = (int)std::min<int64_t>(std::max<int64_t>(wb, 1), 256 * 64);
  hipLaunchKernelGGL(k_fo_small, dim3(gs), dim3(256), 0, st, g.rows.p, g.ent.p, g.fo.p, g.n_slots);
  hipLaunchKernelGGL(k_fo_large, dim3(gl), dim3(256), 0, st, g.rows.p, g.ent.p, g.fo.p, g.n_slots);
  hipLaunchKernelGGL(k_guide_small, dim3(gs), dim3(256), 0, st, g.rows.p, g.fo.p, g.n_slots);
  hipLaunchKernelGGL(k_guide_large, dim3(gl), dim3(256), 0, st, g.rows.p, g.fo.p, g.n_slots);
  if (g.n_entries > 0) {
    int ge = (int)std::min<int64_t>((g.n_entries + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_fo_link, dim3(ge), dim3(256), 0, st, g.rows.p, g.fo.p, g.n_entries, g.vmin, g.n_slots);
  }
  SRW_HIP(hipGetLastError());
  SRW_HIP(hipStreamSynchronize(st));
  g.has_fo = true;
}

}  // namespace srw
