// sampling.h — device samplers, bit-identical to M/algorithm/RandomSample.scala.
#pragma once
#include "wave_primitives.h"

namespace srw {

// ---- node2vec bias: RandomSample.computeSecondOrderWeights (:27-44) ------------------------------------
struct Bias {
  float p, q;
  int32_t prev;             // previous vertex id
  bool second_order;        // false on the first step (initFirstStep samples the raw weights, RandomWalk.scala:57)
  bool need_member;         // q != 1: membership in N(prev) changes the weight; q == 1: w / 1.0f == w either way
  const uint32_t *prev_sids;  // sorted (id - vmin) of N(prev)
  int32_t prev_deg;
  int32_t vmin;
};

__device__ inline bool sorted_contains(const uint32_t *a, int32_t n, uint32_t x) {
  int32_t lo = 0, hi = n;
  while (lo < hi) {
    int32_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo < n && a[lo] == x;
}

__device__ inline float biased_weight(const Bias &b, int32_t id, float w) {
  if (!b.second_order) return w;
  if (id == b.prev) return w / b.p;                 // :35  (checked first)
  if (!b.need_member) return w / b.q;               // q == 1.0f
  if (sorted_contains(b.prev_sids, b.prev_deg, (uint32_t)((int64_t)id - b.vmin))) return w;  // :37
  return w / b.q;                                   // :33
}

// ---- exact pick, one lane, fully sequential (irregular rows and tiny degrees) ---------------------------
// Literally RandomSample.sample on the (biased) row.
__device__ inline int32_t lane_pick_sequential(const Ent *row, int32_t deg, const Bias &b, float r) {
  double sum = 0.0;
  for (int32_t k = 0; k < deg; ++k) {
    Ent e = row[k];
    sum = sum + (double)biased_weight(b, e.id, e.w);
  }
  double p = (double)r, acc = 0.0;
  for (int32_t k = 0; k < deg; ++k) {
    Ent e = row[k];
    acc += (double)biased_weight(b, e.id, e.w) / sum;
    if (acc >= p) return k;
  }
  return 0;
}

// ---- exact pick through the precomputed first-order CDF + guide table (p == q == 1) --------------------
// fo[k].cdf is the reference's running `acc` after entry k, precomputed with the reference's own operation
// order, so "first k with cdf_k >= p" is the reference's answer.  The guide entry of bucket
// j = floor(m * deg / 2^24), m = floor(r * 2^24), is a proven lower bound of that k (sampler_tables.hip),
// so the forward scan from it returns the same k as the reference's scan from 0.
typedef int int4v __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ inline FoEnt load_fo(const FoEnt *p) {
  if (NT) {  // L1-bypassing 2 x 16-byte loads: the record is used once, do not pull its 128-B line into the TCP
    const int4v *q = reinterpret_cast<const int4v *>(p);
    int4v a = __builtin_nontemporal_load(q), b = __builtin_nontemporal_load(q + 1);
    FoEnt e;
    e.cdf = __longlong_as_double((long long)(((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x));
    e.id = a.z; e.guide = a.w;
    e.noff = (int64_t)(((uint64_t)(uint32_t)b.y << 32) | (uint32_t)b.x);
    e.ndeg = b.z; e.nflags = (uint32_t)b.w;
    return e;
  }
  return *p;
}

// Returns the chosen record (id + the neighbor's row descriptor); k_out = its position.
template <bool NT>
__device__ inline FoEnt fo_pick(const FoEnt *row, int32_t deg, float r, int32_t &k_out, unsigned &reads) {
  double p = (double)r;
  float rs = r * 16777216.0f;
  uint32_t m = (rs >= 16777215.0f) ? 16777215u : (rs > 0.0f ? (uint32_t)rs : 0u);
  uint32_t j = (uint32_t)(((uint64_t)m * (uint64_t)(uint32_t)deg) >> 24);
  FoEnt e = load_fo<NT>(row + j);
  reads = 1;
  int32_t k = e.guide;
  bool found = false;
  if (k != (int32_t)j && k < deg) { e = load_fo<NT>(row + k); ++reads; }
  if (k < deg) {
    while (true) {
      if (e.cdf >= p) { found = true; break; }
      ++k;
      if (k >= deg) break;
      e = load_fo<NT>(row + k); ++reads;
    }
  }
  if (!found) { k = 0; e = load_fo<NT>(row); ++reads; }   // edges.head fallback (:24)
  k_out = k;
  return e;
}

// ---- exact pick, one wave per walker (general p, q) -----------------------------------------------------
// All 64 lanes must call this with identical arguments.  Returns the chosen position (wave-uniform).
__device__ inline int32_t wave_pick(const Ent *row, int32_t deg, const Bias &b, float r, unsigned &fallback) {
  const int lane = lane_id();
  // pass 1: S = foldLeft(0.0)(_ + w')      (:14)
  double part = 0.0;
  SumCert cert;
  for (int32_t base = 0; base < deg; base += 64) {
    int32_t k = base + lane;
    if (k < deg) {
      Ent e = row[k];
      float w = biased_weight(b, e.id, e.w);
      part += (double)w;
      cert.add(w);
    }
  }
  int emin = wave_min_i32(cert.emin), emax = wave_max_i32(cert.emax);
  bool bad = __any(cert.bad);
  double S;
  if (sum_is_exact(emin, emax, bad, deg)) {
    S = wave_sum_f64(part);
  } else {
    fallback = 1;
    S = 0.0;
    for (int32_t base = 0; base < deg; base += 64) {
      int32_t k = base + lane;
      double wd = 0.0;
      if (k < deg) { Ent e = row[k]; wd = (double)biased_weight(b, e.id, e.w); }
      int cnt = min(64, deg - base);
      for (int i = 0; i < cnt; ++i) S = S + readlane_f64(wd, i);
    }
  }
  // pass 2: acc += w' / S ; first acc >= p   (:18-22)
  const double p = (double)r;
  double acc = 0.0;
  for (int32_t base = 0; base < deg; base += 64) {
    int32_t k = base + lane;
    double d = 0.0;
    if (k < deg) { Ent e = row[k]; d = (double)biased_weight(b, e.id, e.w) / S; }
    int cnt = min(64, deg - base);
    for (int i = 0; i < cnt; ++i) {
      acc = acc + readlane_f64(d, i);
      if (__builtin_amdgcn_readfirstlane((int)(acc >= p))) return base + i;  // acc is wave-uniform
    }
  }
  return 0;  // edges.head (:24)
}

}  // namespace srw
