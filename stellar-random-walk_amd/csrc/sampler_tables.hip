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
#include <cstring>

#include "engine.h"
#include "sampling.h"

namespace srw {
namespace {

constexpr int SMALL_DEG = 32;

__device__ inline bool weight_regular(float w) { return w >= 0.0f; }  // false for NaN and negatives

// Where the table kernels keep the running CDF and the guide entries: in the exact 32-byte records, or in slim
// temporaries (8-byte CDF + 4-byte guide) from which the compact 16-byte records are derived without ever
// materialising the exact table (build_first_order_tables, "compact first").
struct FoStore {
  FoEnt *fo;
  __device__ inline void put(int64_t e, double cdf, int32_t id) const {
    FoEnt f; f.cdf = cdf; f.id = id; f.guide = 0; f.noff = 0; f.ndeg = 0; f.nflags = 0; fo[e] = f;
  }
  __device__ inline double cdf(int64_t e) const { return fo[e].cdf; }
  __device__ inline void set_guide(int64_t e, int32_t g) const { fo[e].guide = g; }
  __device__ inline int32_t guide(int64_t e) const { return fo[e].guide; }
};
struct SlimStore {
  double *c; int32_t *g;
  __device__ inline void put(int64_t e, double cdf, int32_t) const { c[e] = cdf; }
  __device__ inline double cdf(int64_t e) const { return c[e]; }
  __device__ inline void set_guide(int64_t e, int32_t gg) const { g[e] = gg; }
  __device__ inline int32_t guide(int64_t e) const { return g[e]; }
};

// deg <= SMALL_DEG: one lane per row, literal sequential evaluation.
template <class ST>
__global__ void k_fo_small(Row *rows, const Ent *__restrict__ ent, ST st, int64_t n_slots) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x) {
    Row r = rows[v];
    if (r.deg <= 0 || r.deg > SMALL_DEG) continue;
    const Ent *row = ent + r.off;
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
      st.put(r.off + k, acc, e.id);
    }
    if (irr) rows[v].flags = r.flags | ROW_IRREGULAR;
  }
}

// deg > SMALL_DEG: one wave per row.
// Rows are handed out through a global counter (4 slots per grab, ascending id): a fixed wave stride would give
// wave w every id == w (mod #waves), and in RMAT the ids with few set low bits are ALL hubs (one wave then owns
// ~5 % of the graph).  Ascending order also starts the longest rows (low ids) first.
template <class ST>
__global__ void k_fo_large(Row *rows, const Ent *__restrict__ ent, ST st, int64_t n_slots,
                           unsigned long long *next_slot) {
  const int lane = lane_id();
  while (true) {
    unsigned long long grab = 0;
    if (lane == 0) grab = atomicAdd(next_slot, 4ull);
    grab = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab);
    if ((int64_t)grab >= n_slots) break;
   for (int64_t v = (int64_t)grab; v < (int64_t)grab + 4 && v < n_slots; ++v) {
    Row r = rows[v];
    if (r.deg <= SMALL_DEG) continue;
    const Ent *row = ent + r.off;
    double part = 0.0;
    SumCert cert;
    bool irr = false;
    for (int32_t base = 0; base < r.deg; base += 256) {     // 4 independent loads in flight per lane
      float wv[4]; bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int32_t k = base + u * 64 + lane;
        ok[u] = k < r.deg;
        wv[u] = row[ok[u] ? k : r.deg - 1].w;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) { part += (double)wv[u]; cert.add(wv[u]); irr |= !weight_regular(wv[u]); }
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
    // A hub row is walked by ONE wave (the chain is sequential by definition), so its loads are software-pipelined:
    // the 4 chunks of group g+1 are requested before the 4 chunks of group g are chained.
    Ent z; z.id = 0; z.w = 0.0f;
    Ent cur[4], nxt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { int32_t k = u * 64 + lane; cur[u] = k < r.deg ? row[k] : z; }
    for (int32_t base = 0; base < r.deg; base += 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { int64_t k = (int64_t)base + 256 + u * 64 + lane; nxt[u] = k < r.deg ? row[k] : z; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int32_t cb = base + u * 64;
        if (cb < r.deg) {
          const int32_t k = cb + lane;
          const Ent e = cur[u];
          const double d = (k < r.deg) ? (double)e.w / S : 0.0;
          const int cnt = min(64, r.deg - cb);
          double mine = 0.0;
          if (cnt == 64) {
#pragma unroll
            for (int i = 0; i < 64; ++i) {       // fully unrolled: readlanes hoist ahead of the dependent adds
              acc = acc + readlane_f64(d, i);
              if (lane == i) mine = acc;
            }
          } else {
            for (int i = 0; i < cnt; ++i) {
              acc = acc + readlane_f64(d, i);
              if (lane == i) mine = acc;
            }
          }
          if (k < r.deg) st.put(r.off + k, mine, e.id);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
    }
    if (irr && lane == 0) rows[v].flags = r.flags | ROW_IRREGULAR;
   }
  }
}

__device__ inline double bucket_threshold(uint32_t j, uint32_t deg) {
  uint64_t m = (((uint64_t)j << 24) + deg - 1) / deg;  // ceil(j * 2^24 / deg)
  return (double)m * (1.0 / 16777216.0);               // exact
}

template <class ST>
__global__ void k_guide_small(const Row *__restrict__ rows, ST st, int64_t n_slots) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x) {
    Row r = rows[v];
    if (r.deg <= 0 || r.deg > SMALL_DEG || (r.flags & ROW_IRREGULAR)) continue;
    int32_t k = 0;
    for (int32_t j = 0; j < r.deg; ++j) {   // thresholds and cdf both non-decreasing: two-pointer merge
      double t = bucket_threshold((uint32_t)j, (uint32_t)r.deg);
      while (k < r.deg && !(st.cdf(r.off + k) >= t)) ++k;
      st.set_guide(r.off + j, k);
    }
  }
}

// Guide entries of rows with deg > SMALL_DEG.  Rows are cut into work items of GUIDE_ITEM consecutive buckets so
// that a hub is spread over many waves; inside an item the answers are non-decreasing in j, so after the first
// chunk (plain binary search) every lane gallops from the previous chunk's last answer: ~log2(64) probes inside a
// couple of cache lines instead of log2(deg) probes across the whole row.
constexpr int GUIDE_ITEM = 8192;
struct GuideItem { uint32_t slot_lo, slot_hi_part; };   // {slot (low 32), slot (high 8) << 24 | part}

__global__ void k_guide_make_items(const Row *__restrict__ rows, int64_t n_slots, unsigned long long *counter,
                                   uint2 *__restrict__ items, unsigned long long cap) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x) {
    Row r = rows[v];
    if (r.deg <= SMALL_DEG || (r.flags & ROW_IRREGULAR)) continue;
    unsigned n = (unsigned)((r.deg + GUIDE_ITEM - 1) / GUIDE_ITEM);
    unsigned long long b = atomicAdd(counter, (unsigned long long)n);
    for (unsigned i = 0; i < n && b + i < cap; ++i) items[b + i] = make_uint2((uint32_t)v, ((uint32_t)(v >> 32) << 24) | i);
  }
}

template <class ST>
__device__ inline int32_t lower_bound_cdf(const ST &st, int64_t off, int32_t lo, int32_t hi, double t) {
  while (lo < hi) {
    int32_t mid = lo + ((hi - lo) >> 1);
    if (st.cdf(off + mid) >= t) hi = mid; else lo = mid + 1;
  }
  return lo;
}

template <class ST>
__global__ void k_guide_large(const Row *__restrict__ rows, ST st, const uint2 *__restrict__ items,
                              const unsigned long long *n_items_p) {
  const int lane = lane_id();
  const unsigned long long n_items = *n_items_p;
  unsigned long long wave = (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x) >> 6;
  const unsigned long long n_waves = ((unsigned long long)gridDim.x * blockDim.x) >> 6;
  for (unsigned long long it = wave; it < n_items; it += n_waves) {
    uint2 item = items[it];
    const int64_t v = (int64_t)item.x | ((int64_t)(item.y >> 24) << 32);
    const int32_t part = (int32_t)(item.y & 0xFFFFFFu);
    Row r = rows[v];
    const int32_t jb = part * GUIDE_ITEM, je = min(r.deg, jb + GUIDE_ITEM);
    int32_t start = -1;                                   // answer of the previous chunk's last lane
    for (int32_t j0 = jb; j0 < je; j0 += 64) {
      const int32_t j = j0 + lane;
      int32_t ans = r.deg;
      if (j < je) {
        const double t = bucket_threshold((uint32_t)j, (uint32_t)r.deg);
        if (start < 0) {
          ans = lower_bound_cdf(st, r.off, 0, r.deg, t);
        } else {                                           // gallop: answer >= start (monotone in j)
          int32_t lo = start, step = 1;
          while (lo + step < r.deg && !(st.cdf(r.off + lo + step - 1) >= t)) { lo += step; step <<= 1; }
          ans = lower_bound_cdf(st, r.off, lo, min(r.deg, lo + step), t);
        }
        st.set_guide(r.off + j, ans);
      }
      const int last = min(63, je - j0 - 1);
      start = __builtin_amdgcn_readlane(ans, last);
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

// ---- compact 16-byte records (lattice draws) derived from the CDF / guide store ------------------------------------
template <class ST>
__device__ inline void cfo_write(const ST &st, const Row *__restrict__ rows /* of the neighbors: the links */, const Ent *__restrict__ ent, int32_t vmin,
                                 int64_t n_slots, const Row &r, int32_t j, bool regular, CfoEnt *__restrict__ cfo,
                                 unsigned long long *escapes) {
  const int64_t e = r.off + j;
  const int32_t id = ent[e].id;
  const int64_t s = (int64_t)id - vmin;
  Row nr; nr.off = 0; nr.deg = 0; nr.flags = 0;
  if (s >= 0 && s < n_slots) nr = rows[s];
  bool esc = nr.deg > (int32_t)CFO_NDEG_MAX || nr.off >= ((int64_t)1 << 36);
  uint32_t cg = 0; uint32_t dbits = 0;
  if (regular) {
    const double sc = st.cdf(e) * 16777216.0;                     // exact scaling
    const uint32_t c = sc >= 16777215.0 ? 16777215u : (uint32_t)sc;   // min(floor, 2^24 - 1); cdf >= 0 on a regular row
    const int32_t delta = j - st.guide(e);
    const int32_t d12 = (delta < -2047 || delta > 2047) ? CFO_GD_SAT : delta;   // saturated: the pick bisects the row
    dbits = (uint32_t)d12 & 0xFFFu;
    cg = c | ((dbits & 0xFFu) << 24);
  }
  CfoEnt o;
  o.cg = cg; o.id = id;
  o.link = ((uint64_t)nr.off & CFO_NOFF_MASK) | ((uint64_t)(dbits >> 8) << 36) |
           ((uint64_t)(uint32_t)min(nr.deg, (int32_t)CFO_NDEG_MAX) << 40) | ((uint64_t)((nr.flags & ROW_IRREGULAR) != 0) << 63);
  cfo[e] = o;
  if (esc) atomicAdd(escapes, 1ull);
}
// every row gets records (irregular rows are never sampled through cg, but their links are used)
template <class ST>
__global__ void k_cfo_rows(const Row *__restrict__ rows, const Row *__restrict__ link_rows, const Ent *__restrict__ ent, ST st,
                           CfoEnt *__restrict__ cfo, int64_t n_slots, int32_t vmin, unsigned long long *escapes,
                           unsigned long long *next_slot) {
  const int lane = lane_id();
  while (true) {
    unsigned long long grab = 0;
    if (lane == 0) grab = atomicAdd(next_slot, 4ull);
    grab = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab);
    if ((int64_t)grab >= n_slots) break;
    for (int64_t v = (int64_t)grab; v < (int64_t)grab + 4 && v < n_slots; ++v) {
      const Row r = rows[v];
      const bool regular = !(r.flags & ROW_IRREGULAR);
      for (int32_t j = lane; j < r.deg; j += 64) cfo_write(st, link_rows, ent, vmin, n_slots, r, j, regular, cfo, escapes);
    }
  }
}
// exact records from the slim temporaries (only when the compact table had to be abandoned)
__global__ void k_fo_from_slim(const Row *__restrict__ rows, const Ent *__restrict__ ent, const double *__restrict__ c,
                               const int32_t *__restrict__ g, FoEnt *__restrict__ fo, int64_t n_entries, int32_t vmin,
                               int64_t n_slots) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_entries; e += (int64_t)gridDim.x * blockDim.x) {
    FoEnt f; f.cdf = c[e]; f.id = ent[e].id; f.guide = g[e];
    int64_t s = (int64_t)f.id - vmin;
    Row r; r.off = 0; r.deg = 0; r.flags = 0;
    if (s >= 0 && s < n_slots) r = rows[s];
    f.noff = r.off; f.ndeg = r.deg; f.nflags = r.flags;
    fo[e] = f;
  }
}

// ---- per-call exact prefix sums of the base weights fl(w / q) (general kernel, prefix-sum sampler) ---------------
// A row qualifies when every variant a candidate can take — fl(w/q), w, fl(w/p) — is finite and >= 0 and NO sum the
// samplers form can round: every variant is a multiple of g = 2^(emin - 23) (emin = smallest exponent of a nonzero
// variant), and every quantity the samplers form — a prefix of base weights, a sum of corrections |w' - fl(w/q)| <=
// max variant, their sum — is bounded in magnitude by 2 M with M = sum over the row of the LARGEST variant; so
// 2 M <= 2^53 g makes all of them exactly representable, in any order.  M is accumulated in f64 (relative error
// <= n 2^-53), hence one more bit of margin: M <= 2^51 g.  (The cruder bound n 2^(emax+1) for M turned hub rows away
// four times earlier — at config 3 every row beyond 524 288 entries, which then took the streaming scan.)
struct PqCert {
  int emin; bool bad; double mass;
  int glsb;   // lowest set bit position over all nonzero variants: every variant (hence every sum) is a multiple of 2^glsb
  float maxv; // largest variant of any candidate
  __device__ PqCert() : emin(1 << 20), bad(false), mass(0.0), glsb(1 << 20), maxv(0.0f) {}
  __device__ inline void add(float x) {
    uint32_t b = __float_as_uint(x);
    int ex = (int)((b >> 23) & 0xFFu);
    if (ex == 255 || (b >> 31 && (b & 0x7FFFFFFFu))) { bad = true; return; }
    if ((b & 0x7FFFFFFFu) == 0u) return;
    int e = ex ? ex - 127 : -126;
    emin = min(emin, e);
    const uint32_t mant = (b & 0x7FFFFFu) | (ex ? 0x800000u : 0u);
    glsb = min(glsb, e - 23 + (int)__builtin_ctz(mant));
  }
  __device__ inline void add_entry(float w, float p, float q) {
    const float a = div_exact(w, q), c = div_exact(w, p);
    add(w); add(a); add(c);
    const float mx = fmaxf(w, fmaxf(a, c));
    mass += (double)mx;      // NaN-free once !bad
    maxv = fmaxf(maxv, mx);
  }
};
// every sum over the row is a multiple of 2^glsb and at most 2 * mass: below 2^(24 + glsb) it has at most 24 significant
// bits and is exactly representable in binary32 (mass carries a relative error <= n 2^-53: one bit of margin)
__device__ inline bool pq_row_f32(int glsb, double mass) {
  if (glsb > (1 << 19) || glsb < -1000) return false;
  return mass < ldexp(1.0, 22 + glsb);
}
// Row::flags bits of the 16-bit table deltas: the largest chunk (2^(c + 5) candidates, c = 1 .. 7) whose mass — at most chunk size x the
// largest variant — stays below 65 536 units of 2^glsb, and the unit exponent itself
__device__ inline uint32_t pq_row_u16_bits(int glsb, float maxv) {
  if (glsb > 100 || glsb < -100 || !(maxv > 0.0f)) return 0u;
  const double upc = ldexp((double)maxv, -glsb);                 // units per candidate at most (an integer: maxv is a multiple of 2^glsb)
  uint32_t c = 0;
  while (c < 7u && upc * (double)(1u << (c + 6u)) <= 65535.0) ++c;
  return (c << ROW_U16_SHIFT) | (((uint32_t)(glsb + 128) & 0xFFu) << ROW_G_SHIFT);
}
__device__ inline bool pq_row_ok(int emin, bool bad, double mass) {
  if (bad) return false;
  if (emin > (1 << 19)) return false;                 // all-zero row: leave it to the literal sampler
  return mass <= ldexp(1.0, 51 + emin - 23);
}

__global__ void k_pq_small(Row *__restrict__ rows, const Ent *__restrict__ ent, double *__restrict__ pq,
                           uint8_t *__restrict__ ok, int64_t n_slots, float p, float q, unsigned long long *bad_rows) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x) {
    Row r = rows[v];
    ok[v] = 0;
    if (r.flags & ROW_PQ_BITS) { r.flags &= ~ROW_PQ_BITS; rows[v].flags = r.flags; }   // the flag mirrors ok[v] (one load less per step)
    if (r.deg <= 0 || r.deg > SMALL_DEG) continue;
    PqCert c;
    for (int32_t k = 0; k < r.deg; ++k) c.add_entry(ent[r.off + k].w, p, q);
    if (!pq_row_ok(c.emin, c.bad, c.mass)) { atomicAdd(bad_rows, 1ull); continue; }
    double acc = 0.0;
    if (pq) for (int32_t k = 0; k < r.deg; ++k) { acc += (double)div_exact(ent[r.off + k].w, q); pq[r.off + k] = acc; }
    ok[v] = 1;
    rows[v].flags = r.flags | ROW_PQ_OK | (pq_row_f32(c.glsb, c.mass) ? ROW_PQ_F32 : 0u) | pq_row_u16_bits(c.glsb, c.maxv);
  }
}

__global__ void k_pq_large(Row *__restrict__ rows, const Ent *__restrict__ ent, double *__restrict__ pq,
                           uint8_t *__restrict__ ok, int64_t n_slots, float p, float q, unsigned long long *next_slot) {
  const int lane = lane_id();
  while (true) {
    unsigned long long grab = 0;
    if (lane == 0) grab = atomicAdd(next_slot, 4ull);
    grab = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab);
    if ((int64_t)grab >= n_slots) break;
    for (int64_t v = (int64_t)grab; v < (int64_t)grab + 4 && v < n_slots; ++v) {
      Row r = rows[v];
      if (r.deg <= SMALL_DEG) continue;
      const Ent *row = ent + r.off;
      PqCert c;
      for (int32_t k = lane; k < r.deg; k += 64) c.add_entry(row[k].w, p, q);
      const int emin = wave_min_i32(c.emin), glsb = wave_min_i32(c.glsb);
      const bool bad = __any(c.bad);
      const double mass = wave_sum_f64(bad ? 0.0 : c.mass);
      const float maxv = __int_as_float(wave_max_i32(__float_as_int(bad ? 0.0f : c.maxv)));     // (non-negative floats order like their bit patterns)
      if (!pq_row_ok(emin, bad, mass)) { if (lane == 0) atomicAdd(next_slot + 1, 1ull); continue; }   // ok[v] stays 0 (set by k_pq_small)
      double carry = 0.0;                                            // exact sums: any order gives the same bits
      for (int32_t base = 0; base < r.deg; base += 64) {
        int32_t k = base + lane;
        double x = k < r.deg ? (double)div_exact(row[k].w, q) : 0.0;
        for (int o = 1; o < 64; o <<= 1) { double t = __shfl_up(x, o); if (lane >= o) x += t; }
        if (pq && k < r.deg) pq[r.off + k] = carry + x;
        carry += readlane_f64(x, 63);
      }
      if (lane == 0) { ok[v] = 1; rows[v].flags = r.flags | ROW_PQ_OK | (pq_row_f32(glsb, mass) ? ROW_PQ_F32 : 0u) | pq_row_u16_bits(glsb, maxv); }
    }
  }
}

}  // namespace

void build_pq_tables(srw_handle *h, float p, float q) {
  Graph &g = h->g;
  uint32_t pb, qb; memcpy(&pb, &p, 4); memcpy(&qb, &q, 4);
  if (g.has_pq && g.pq_pbits == pb && g.pq_qbits == qb) return;
  hipStream_t st = h->stream;
  g.has_pq = false;
  size_t free_b = 0, total_b = 0;
  SRW_HIP(hipMemGetInfo(&free_b, &total_b));
  // unit-weight graphs: the prefix sums have a closed form (device_common.h:PqRow) — no array, only the rows' certificates and flags
  const bool unit = !getenv("SRW_NO_UNIT_PQ") && graph_has_unit_weights(h);
  g.pq_unit = unit ? (double)(1.0f / q) : 0.0;
  const size_t need = (unit ? 0 : (size_t)g.n_entries * sizeof(double)) + (size_t)g.n_slots;
  if (!unit && g.pq.n < (size_t)g.n_entries && free_b < need + ((size_t)8 << 30)) return;   // optional structure: skip when tight
  if (unit) g.pq.release(); else g.pq.ensure((size_t)g.n_entries);
  g.pq_ok.ensure((size_t)g.n_slots);
  int gs = (int)std::min<int64_t>(std::max<int64_t>((g.n_slots + 255) / 256, 1), 256 * 32);
  DevBuf<unsigned long long> next_slot; next_slot.alloc(2);      // [0] cursor, [1] rows without the certificate
  SRW_HIP(hipMemsetAsync(next_slot.p, 0, 16, st));
  hipLaunchKernelGGL(k_pq_small, dim3(gs), dim3(256), 0, st, g.rows.p, g.ent.p, g.pq.p, g.pq_ok.p, g.n_slots, p, q, next_slot.p + 1);
  hipLaunchKernelGGL(k_pq_large, dim3(256 * 8), dim3(256), 0, st, g.rows.p, g.ent.p, g.pq.p, g.pq_ok.p, g.n_slots, p, q,
                     next_slot.p);
  SRW_HIP(hipGetLastError());
  unsigned long long nbad = 0;
  SRW_HIP(hipMemcpyAsync(&nbad, next_slot.p + 1, 8, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  g.pq_bad_rows = (int64_t)nbad;
  g.pq_pbits = pb; g.pq_qbits = qb; g.has_pq = true;
}

// CDF + guide into `st` (exact records or slim temporaries); leaves the guide work items in `items`.
template <class ST>
static void run_cdf_and_guide(srw_handle *h, ST st, DevBuf<uint2> &items, DevBuf<unsigned long long> &n_items) {
  Graph &g = h->g;
  hipStream_t stq = h->stream;
  int64_t tb = (g.n_slots + 255) / 256;
  int gs = (int)std::min<int64_t>(std::max<int64_t>(tb, 1), 256 * 32);
  DevBuf<unsigned long long> next_slot; next_slot.alloc(1);
  SRW_HIP(hipMemsetAsync(next_slot.p, 0, 8, stq));
  hipLaunchKernelGGL((k_fo_small<ST>), dim3(gs), dim3(256), 0, stq, g.rows.p, g.ent.p, st, g.n_slots);
  hipLaunchKernelGGL((k_fo_large<ST>), dim3(256 * 8), dim3(256), 0, stq, g.rows.p, g.ent.p, st, g.n_slots, next_slot.p);
  hipLaunchKernelGGL((k_guide_small<ST>), dim3(gs), dim3(256), 0, stq, g.rows.p, st, g.n_slots);
  const unsigned long long cap = (unsigned long long)g.n_entries / SMALL_DEG + 1024;
  items.alloc((size_t)cap); n_items.alloc(2);
  SRW_HIP(hipMemsetAsync(n_items.p, 0, 16, stq));
  hipLaunchKernelGGL(k_guide_make_items, dim3(gs), dim3(256), 0, stq, g.rows.p, g.n_slots, n_items.p, items.p, cap);
  hipLaunchKernelGGL((k_guide_large<ST>), dim3(256 * 16), dim3(256), 0, stq, g.rows.p, st, items.p, n_items.p);
  SRW_HIP(hipGetLastError());
  SRW_HIP(hipStreamSynchronize(stq));   // next_slot goes out of scope
}

template <class ST>
static unsigned long long run_cfo(srw_handle *h, ST st, unsigned long long *esc_word, const Row *link_rows = nullptr) {
  Graph &g = h->g;
  hipStream_t stq = h->stream;
  DevBuf<unsigned long long> next_slot; next_slot.alloc(1);
  SRW_HIP(hipMemsetAsync(next_slot.p, 0, 8, stq));
  SRW_HIP(hipMemsetAsync(esc_word, 0, 8, stq));
  hipLaunchKernelGGL((k_cfo_rows<ST>), dim3(256 * 8), dim3(256), 0, stq, g.rows.p, link_rows ? link_rows : g.rows.p, g.ent.p, st,
                     g.cfo.p, g.n_slots, g.vmin, esc_word, next_slot.p);
  unsigned long long n_esc = 0;
  SRW_HIP(hipMemcpyAsync(&n_esc, esc_word, 8, hipMemcpyDeviceToHost, stq));
  SRW_HIP(hipStreamSynchronize(stq));
  SRW_HIP(hipGetLastError());
  return n_esc;
}

// First-order tables.  want_exact = false (Philox draws): "compact first" — the running CDF and the guide entries go
// into slim temporaries (12 B/entry), the 16-byte lattice records are derived from them, and the 32-byte exact table
// is materialised only if some entry needs an escape.  want_exact = true (constant-r hook, SRW_WALK_NO_COMPACT,
// sharded handles): the exact table is built (and kept next to the compact one when both exist).
void build_first_order_tables(srw_handle *h, bool want_exact) {
  Graph &g = h->g;
  const bool sharded = h->cfg.world > 1;
  if (g.has_cfo_local && !sharded) { g.has_cfo = true; g.has_cfo_local = false; }   // world 1: the local records are the whole graph's
  if (want_exact ? g.has_fo : (g.has_cfo || sharded || g.cfo_rejected || g.n_entries == 0) && (g.has_cfo || g.has_fo)) return;
  hipStream_t st = h->stream;
  DevBuf<uint2> items; DevBuf<unsigned long long> n_items;
  if (!want_exact && g.has_fo) {           // exact table already there (an earlier constant-r call): derive the compact one
    n_items.alloc(2);
    g.cfo.alloc((size_t)g.n_entries);
    if (run_cfo(h, FoStore{g.fo.p}, n_items.p + 1) == 0) g.has_cfo = true;
    else { g.cfo.release(); g.cfo_rejected = true; }
    return;
  }
  if (!want_exact && !sharded && g.n_entries > 0 && !g.cfo_rejected) {
    DevBuf<double> c; DevBuf<int32_t> gd;
    c.alloc((size_t)g.n_entries); gd.alloc((size_t)g.n_entries);
    SlimStore slim{c.p, gd.p};
    run_cdf_and_guide(h, slim, items, n_items);
    g.cfo.alloc((size_t)g.n_entries);
    if (getenv("SRW_TIMING")) fprintf(stderr, "[timing] compact first-order table: %zu bytes at %p\n", (size_t)g.n_entries * sizeof(CfoEnt), (void *)g.cfo.p);
    if (run_cfo(h, slim, n_items.p + 1) == 0) { g.has_cfo = true; return; }
    // some entry needs an escape (large guide delta on weighted hubs, giant degree): exact records instead
    g.cfo.release(); g.cfo_rejected = true;
    g.fo.alloc((size_t)g.n_entries);
    int ge = (int)std::min<int64_t>((g.n_entries + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_fo_from_slim, dim3(ge), dim3(256), 0, st, g.rows.p, g.ent.p, c.p, gd.p, g.fo.p, g.n_entries, g.vmin,
                       g.n_slots);
    SRW_HIP(hipGetLastError());
    SRW_HIP(hipStreamSynchronize(st));
    g.has_fo = true;
    return;
  }
  g.fo.alloc((size_t)g.n_entries);
  FoStore fs{g.fo.p};
  run_cdf_and_guide(h, fs, items, n_items);
  if (g.n_entries > 0) {
    int ge = (int)std::min<int64_t>((g.n_entries + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_fo_link, dim3(ge), dim3(256), 0, st, g.rows.p, g.fo.p, g.n_entries, g.vmin, g.n_slots);
  }
  SRW_HIP(hipGetLastError());
  SRW_HIP(hipStreamSynchronize(st));
  g.has_fo = true;
}

// Compact records over the LOCAL rows of a sharded handle: the guide deltas, the lattice values c and the ids that the per-lane
// q == 1 step needs (k_sh_step_q1); the links name local rows only and are not used there.  false: an entry needs an escape.
bool build_local_cfo(srw_handle *h) {
  Graph &g = h->g;
  if (g.has_cfo || g.cfo_linked || g.has_cfo_local) return true;
  if (g.cfo_rejected || g.n_entries == 0) return false;
  build_first_order_tables(h, true);
  DevBuf<unsigned long long> esc; esc.alloc(1);
  g.cfo.alloc((size_t)g.n_entries);
  if (run_cfo(h, FoStore{g.fo.p}, esc.p) != 0) { g.cfo.release(); g.cfo_rejected = true; return false; }
  g.has_cfo_local = true;
  return true;
}

// ---- vertex-sharded walk: row descriptors across shards (srw_shard_rows_*) -------------------------------------------
// A record of the replicated first-order table carries the row descriptor of the neighbor it names, so a step never
// reads the row table.  On a shard the neighbor's row lives on ANOTHER shard: the shards exchange their row tables once
// (16 B per id slot; element-wise maximum — only the owner of a vertex has a non-zero descriptor for it), and the compact
// records are derived with links into the owners' tables; a walker then travels with the link of the vertex it stands on.
namespace {
__global__ void k_rows_max(unsigned long long *__restrict__ acc, const unsigned long long *__restrict__ other, int64_t n_words) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) {
    const long long a = (long long)acc[i], b = (long long)other[i];
    acc[i] = (unsigned long long)(a > b ? a : b);
  }
}
}  // namespace

void shard_rows_export(srw_handle *h, void *d_rows, int64_t n_slots) {
  Graph &g = h->g;
  if (!g.loaded) throw Error(SRW_ERR_INVALID, "no graph loaded");
  if (n_slots != g.n_slots) throw Error(SRW_ERR_INVALID, "row table size mismatch");
  build_first_order_tables(h, true);           // the irregular-row flags are final after the CDF pass
  SRW_HIP(hipMemcpyAsync(d_rows, g.rows.p, (size_t)n_slots * sizeof(Row), hipMemcpyDeviceToDevice, h->stream));
  SRW_HIP(hipStreamSynchronize(h->stream));
}

void shard_rows_merge(srw_handle *h, void *d_rows, const void *d_other, int64_t n_slots) {
  if (n_slots != h->g.n_slots) throw Error(SRW_ERR_INVALID, "row table size mismatch");
  const int64_t n_words = n_slots * 2;
  const int blocks = (int)std::min<int64_t>(std::max<int64_t>((n_words + 255) / 256, 1), 256 * 32);
  hipLaunchKernelGGL(k_rows_max, dim3(blocks), dim3(256), 0, h->stream, (unsigned long long *)d_rows, (const unsigned long long *)d_other, n_words);
  SRW_HIP(hipGetLastError());
  SRW_HIP(hipStreamSynchronize(h->stream));
}

// true: the compact linked table is in place (k_sh_step_cfo); false: some record needs an escape, nothing changed
bool shard_rows_commit(srw_handle *h, const void *d_rows_all, int64_t n_slots) {
  Graph &g = h->g;
  if (!g.loaded) throw Error(SRW_ERR_INVALID, "no graph loaded");
  if (n_slots != g.n_slots) throw Error(SRW_ERR_INVALID, "row table size mismatch");
  build_first_order_tables(h, true);
  shard_rows_release(h);
  if (g.has_cfo_local && h->cfg.world == 1) { g.has_cfo = true; g.has_cfo_local = false; }
  if (g.has_cfo_local) { g.cfo.release(); g.has_cfo_local = false; }      // the linked records replace the local ones
  g.rows_all.alloc((size_t)n_slots);
  SRW_HIP(hipMemcpyAsync(g.rows_all.p, d_rows_all, (size_t)n_slots * sizeof(Row), hipMemcpyDeviceToDevice, h->stream));
  if (g.n_entries == 0) { SRW_HIP(hipStreamSynchronize(h->stream)); g.cfo_linked = true; return true; }
  DevBuf<unsigned long long> esc; esc.alloc(1);
  if (g.has_cfo) {      // world == 1 after a replicated walk: the local links are the owners' links
    if (h->cfg.world != 1) throw Error(SRW_ERR_INVALID, "srw_shard_rows_commit: the handle already holds an unsharded compact table");
    SRW_HIP(hipStreamSynchronize(h->stream)); g.cfo_linked = true; return true;
  }
  g.cfo.alloc((size_t)g.n_entries);
  if (run_cfo(h, FoStore{g.fo.p}, esc.p, g.rows_all.p) != 0) { g.cfo.release(); g.rows_all.release(); return false; }
  g.cfo_linked = true;
  return true;
}

void shard_rows_release(srw_handle *h) {
  Graph &g = h->g;
  if (g.cfo_linked && !g.has_cfo) g.cfo.release();
  g.cfo_linked = false;
  g.rows_all.release();
}

}  // namespace srw
