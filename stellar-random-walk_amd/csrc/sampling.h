// sampling.h — device samplers, bit-identical to M/algorithm/RandomSample.scala.
#pragma once
#include "wave_primitives.h"

namespace srw {

// ---- node2vec bias: RandomSample.computeSecondOrderWeights (:27-44) ------------------------------------
struct Bias {
  uint32_t prev_hub = 0;   // 1 + ordinal of N(prev)'s bitmap over the id slots, 0 = none (GraphView::hub_bm)
  float p, q;
  int32_t prev;             // previous vertex id
  bool second_order;        // false on the first step (initFirstStep samples the raw weights, RandomWalk.scala:57)
  bool need_member;         // q != 1: membership in N(prev) changes the weight; q == 1: w / 1.0f == w either way
  const uint32_t *prev_sids;  // sorted (id - vmin) of N(prev)
  int32_t prev_deg;
  int32_t vmin;
};


__device__ inline Row uniform_row(Row r) {            // the row descriptor of a wave's walker is wave-uniform: keep it in SGPRs
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r.off), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)r.off >> 32));
  Row o; o.off = (int64_t)(((uint64_t)hi << 32) | lo); o.deg = __builtin_amdgcn_readfirstlane(r.deg);
  o.flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.flags);
  return o;
}
// ... and lane j's row descriptor (j wave-uniform): the per-lane rows of a grab of records (k_sh_step_tab)
__device__ inline Row lane_row(const Row &r, int j) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)r.off, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)r.off >> 32), j);
  Row o; o.off = (int64_t)(((uint64_t)hi << 32) | lo); o.deg = __builtin_amdgcn_readlane(r.deg, j);
  o.flags = (uint32_t)__builtin_amdgcn_readlane((int)r.flags, j);
  return o;
}

// x / d, evaluated as x * 2^-k when d = 2^k: both are the correctly rounded value of the SAME real number (2^-k is exact
// for |k| <= 126; FP32 denormals are on, .amdhsa_float_denorm_mode_32 3), so the bits are those of the reference's
// `weight / q` (RandomSample.scala:33,35) — without the ~11-instruction correctly-rounded f32 divide per candidate.  d is
// wave-uniform (p or q), so the test is scalar.  node2vec's usual p, q (0.25, 0.5, 2, 4) all qualify.
__device__ inline float div_exact(float x, float d) {
  const uint32_t bits = __float_as_uint(d), ex = bits >> 23;
  if ((bits & 0x007FFFFFu) == 0u && ex >= 1u && ex <= 253u) return x * __uint_as_float((254u - ex) << 23);
  return x / d;
}

// The biased weight of a candidate (RandomSample.scala:33-38) without a branch per class: the divisor (p for the return edge, 1 for a
// common neighbor — x / 1.0f is x —, q otherwise) is selected per lane, then ONE divide; with p and q both powers of two (the test is
// wave-uniform) one multiply by the selected 2^-k, as div_exact.  Same bits as the three-way if of biased_weight.
struct BiasDiv {
  float p, q, ip, iq; bool fast;
  __device__ BiasDiv(float p_, float q_) : p(p_), q(q_), ip(0.0f), iq(0.0f) {
    const uint32_t bp = __float_as_uint(p_), bq = __float_as_uint(q_), ep = bp >> 23, eq = bq >> 23;
    fast = (bp & 0x007FFFFFu) == 0u && ep >= 1u && ep <= 253u && (bq & 0x007FFFFFu) == 0u && eq >= 1u && eq <= 253u;
    if (fast) { ip = __uint_as_float((254u - ep) << 23); iq = __uint_as_float((254u - eq) << 23); }
  }
  __device__ inline float operator()(float w, bool is_prev, bool member) const {
    if (fast) return w * (is_prev ? ip : member ? 1.0f : iq);
    return w / (is_prev ? p : member ? 1.0f : q);
  }
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
  if (id == b.prev) return div_exact(w, b.p);       // :35  (checked first)
  if (!b.need_member) return div_exact(w, b.q);     // q == 1.0f
  if (sorted_contains(b.prev_sids, b.prev_deg, (uint32_t)((int64_t)id - b.vmin))) return w;  // :37
  return div_exact(w, b.q);                         // :33
}

// ---- exact pick, one lane, fully sequential (irregular rows and tiny degrees) ---------------------------
// Literally RandomSample.sample on the (biased) row.
// entry k of a row: the (id, w) pair, or — unit-weight graphs with GraphView::ids32 — the id alone (half the bytes per candidate)
__device__ inline Ent load_ent(const GraphView &g, const Ent *row, int32_t k) {
  if (g.ids32) { Ent e; e.id = g.ids32[(row - g.ent) + k]; e.w = 1.0f; return e; }
  return row[k];
}
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

template <bool NT>
__device__ inline CfoEnt load_cfo(const CfoEnt *p) {
  const int4v *q = reinterpret_cast<const int4v *>(p);
  int4v a = NT ? __builtin_nontemporal_load(q) : *q;
  CfoEnt e;
  e.cg = (uint32_t)a.x; e.id = a.y;
  e.link = ((uint64_t)(uint32_t)a.w << 32) | (uint32_t)a.z;
  return e;
}

// Lattice draw (p = m * 2^-24) through the compact table: same index as fo_pick, one 16-byte load per probe.
template <bool NT>
__device__ inline CfoEnt cfo_pick(const CfoEnt *row, int32_t deg, uint32_t m, unsigned &reads, int32_t &k_out) {
  const uint32_t j = (uint32_t)(((uint64_t)m * (uint64_t)(uint32_t)deg) >> 24);
  CfoEnt e = load_cfo<NT>(row + j);
  reads = 1;
  const int32_t gd = cfo_delta(e.cg, e.link);
  int32_t k;
  if (gd == CFO_GD_SAT) {                       // delta beyond 12 bits: first k with c_k >= m by bisection (c is monotone)
    int32_t lo = 0, hi = deg;
    while (lo < hi) {
      const int32_t mid = lo + ((hi - lo) >> 1);
      const CfoEnt t = load_cfo<NT>(row + mid); ++reads;
      if ((t.cg & 0xFFFFFFu) < m) lo = mid + 1; else hi = mid;
    }
    k = lo < deg ? lo : 0;                      // no crossing: edges.head (:24)
    e = load_cfo<NT>(row + k); ++reads;
    k_out = k;
    return e;
  }
  k = (int32_t)j - gd;
  if (gd) { e = load_cfo<NT>(row + k); ++reads; }
  while ((e.cg & 0xFFFFFFu) < m) {              // min(floor(cdf * 2^24), 2^24 - 1) < m  <=>  cdf < p
    ++k;
    if (k >= deg) { k = 0; e = load_cfo<NT>(row); ++reads; break; }   // edges.head fallback (:24)
    e = load_cfo<NT>(row + k); ++reads;
  }
  k_out = k;
  return e;
}
template <bool NT>
__device__ inline CfoEnt cfo_pick(const CfoEnt *row, int32_t deg, uint32_t m, unsigned &reads) {
  int32_t k;
  return cfo_pick<NT>(row, deg, m, reads, k);
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

// ---- certified parallel form -----------------------------------------------------------------------------
// Membership of the candidates in N(prev), resolved once per step into an LDS bitmap over the candidate
// positions ("reverse" marking: each element of the usually short N(prev) is looked up in the sorted N(curr)
// and its occurrences' input-order positions are marked), or per candidate by binary search when that is
// cheaper.  Both give the same booleans as prevNeighbors.exists(_._1 == dstId) (:37).
constexpr int BM_WORDS = 2048;               // per-wave LDS bitmap: 65536 candidate positions per segment
constexpr int BM_BITS = BM_WORDS * 32;

struct Member {
  int mode;            // 0: not needed, 1: binary search per candidate, 2: bitmap
  uint32_t *bm;        // LDS, BM_WORDS words, private to the wave
  int32_t seg_base;    // first candidate position covered by the bitmap
  const uint64_t *ehash = nullptr; uint64_t ehash_mask = 0;   // edge hash set (whole-graph handles): mode 1 probes it
  const uint32_t *hub = nullptr;   // this step's N(prev) bitmap over the id slots, if prev is a hub: mode 1 reads one bit
  const uint32_t *bf = nullptr; uint32_t bf_nw = 0;   // N(prev)'s neighbor-set filter (device_common.h:bf_hash): a negative needs no search
  unsigned long long res_bytes = 0;   // binned_resolve: bytes of the candidates it evaluated (entries + prefix sums), bench.py
#ifdef SRW_PHASE_TIMING
  unsigned long long t_fill = 0, t_pass1 = 0, t_pass2 = 0, t_prefix = 0, t_mark;
  unsigned long long t_a = 0, t_p1 = 0, t_p2 = 0, t_w = 0, t_fin = 0;
  unsigned long long t_w_lb = 0, t_w_ins = 0, t_w_la = 0, t_w_probe = 0, t_mark2;
  unsigned long long n_w = 0, n_w_elems = 0, n_w_windows = 0, n_p1 = 0, n_p1_elems = 0, n_binned = 0;
  unsigned long long t_strat[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_step0 = 0;   // wave time per SRW_STRAT_* (whole step)
#endif
};
#ifdef SRW_PHASE_TIMING
#define SRW_T0(m) ((m).t_mark = wall_clock64())
#define SRW_T1(m, f) ((m).f += wall_clock64() - (m).t_mark)
#define SRW_U0(m) ((m).t_mark2 = wall_clock64())
#define SRW_U1(m, f) ((m).f += wall_clock64() - (m).t_mark2)
#define SRW_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define SRW_T0(m)
#define SRW_T1(m, f)
#define SRW_U0(m)
#define SRW_U1(m, f)
#define SRW_DRAIN()
#endif

__device__ inline float biased_weight_m(const Bias &b, const Member &m, int32_t pos, int32_t id, float w) {
  if (!b.second_order) return w;
  if (id == b.prev) return div_exact(w, b.p);
  if (m.mode == 0) return div_exact(w, b.q);
  bool in;
  if (m.mode == 2) { uint32_t t = (uint32_t)(pos - m.seg_base); in = (m.bm[t >> 5] >> (t & 31)) & 1u; }
  else if (m.hub) { const uint32_t x = (uint32_t)((int64_t)id - b.vmin); in = (m.hub[x >> 5] >> (x & 31)) & 1u; }
  else if (m.ehash) in = edge_exists(m.ehash, m.ehash_mask, (uint32_t)((int64_t)b.prev - b.vmin), (uint32_t)((int64_t)id - b.vmin));
  else {
    const uint32_t x = (uint32_t)((int64_t)id - b.vmin);
    in = true;
    if (m.bf) { uint32_t word, mask; bf_hash(x, m.bf_nw, word, mask); in = (m.bf[word] & mask) == mask; }
    if (in) in = sorted_contains(b.prev_sids, b.prev_deg, x);
  }
  return in ? w : div_exact(w, b.q);
}

// ---- exact pick, one wave per walker (general p, q) -----------------------------------------------------
// Sequential-chain form: all 64 lanes call with identical arguments.  Returns the chosen position
// (wave-uniform).  S_known: pass the already-computed sum (fallback of wave_pick_scan) or NaN to compute it.
__device__ inline double wave_sum_exact_or_chain(const Ent *row, int32_t deg, const Bias &b, unsigned &fallback) {
  const int lane = lane_id();
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
  if (sum_is_exact(emin, emax, bad, deg)) return wave_sum_f64(part);
  fallback = 1;
  double S = 0.0;
  for (int32_t base = 0; base < deg; base += 64) {
    int32_t k = base + lane;
    double wd = 0.0;
    if (k < deg) { Ent e = row[k]; wd = (double)biased_weight(b, e.id, e.w); }
    int cnt = min(64, deg - base);
    for (int i = 0; i < cnt; ++i) S = S + readlane_f64(wd, i);
  }
  return S;
}

// The reference's chain acc = fl(acc + d_i), first acc >= p (:18-22) — evaluated 64 elements at a time, bit for bit.
// While acc stays inside one binade [2^e, 2^(e+1)) it is an integer N of units u = 2^(e-52), and adding d rounds to
//   N + D + t,   D = floor(d / u),   t = 1 / 0 as the remainder is above / below u/2, and on a tie t makes N + D + t even
// (round-half-even) — a map N -> N + c[N & 1] whose composition has the same form, so 64 of them are combined by a wave
// scan.  The first element that would leave the binade (N + D >= 2^53, or d >= 2^(e+1), or acc still zero) is added with a
// real f64 addition and the scan restarts behind it: ~1.1 scans + 0.25 single steps per 64 elements on the rows measured
// (tools/chain_model.c is the CPU model this was checked with: 120 000 rows, ties included).  Long rows are where the chain
// hurts: a draw on a CDF boundary of a 10^6-neighbor hub used to cost 10^6 dependent additions of one lane.
__device__ inline void chain_elem_map(double d, int e, unsigned long long &c0, unsigned long long &c1) {
  const unsigned long long BIG = 1ull << 53;               // "leaves the binade": handled by a real addition
  c0 = c1 = 0ull;
  if (d == 0.0) return;
  const unsigned long long bb = (unsigned long long)__double_as_longlong(d);
  const int ed = (int)((bb >> 52) & 0x7FFull);
  if (ed == 0 || ed == 0x7FF || (bb >> 63)) { c0 = c1 = BIG; return; }   // subnormal, infinite, NaN, negative
  const unsigned long long md = (bb & ((1ull << 52) - 1ull)) | (1ull << 52);
  const int shift = e - (ed - 1023);
  if (shift <= 0) { c0 = c1 = BIG; return; }
  if (shift >= 64) return;
  const unsigned long long D = md >> shift, rem = md & ((1ull << shift) - 1ull), half = 1ull << (shift - 1);
  if (rem > half) c0 = c1 = D + 1ull;
  else if (rem < half) c0 = c1 = D;
  else { c0 = D + (D & 1ull); c1 = D + ((D + 1ull) & 1ull); }
}
__device__ inline unsigned long long shfl_up_u64(unsigned long long v, int off) {
  const int lo = __shfl_up((int)(uint32_t)v, off), hi = __shfl_up((int)(uint32_t)(v >> 32), off);
  return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}

// N edge-hash probes per lane in lockstep: the probe loads of one round are independent, so N candidates cost the
// round trips of one.
template <int N>
__device__ inline void edge_exists_n(const uint64_t *tab, uint64_t mask, uint32_t row_slot, const uint32_t (&id_slot)[N],
                                     const bool (&want)[N], bool (&out)[N]) {
  uint64_t key[N], s[N]; bool act[N];
  bool any = false;
#pragma unroll
  for (int u = 0; u < N; ++u) {
    key[u] = ((uint64_t)row_slot << 32) | id_slot[u];
    s[u] = edge_hash(key[u], mask); act[u] = want[u]; out[u] = false; any |= act[u];
  }
  while (any) {
    uint64_t v[N];
#pragma unroll
    for (int u = 0; u < N; ++u) v[u] = act[u] ? tab[s[u]] : 0ull;
    any = false;
#pragma unroll
    for (int u = 0; u < N; ++u)
      if (act[u]) {
        if (v[u] == key[u]) { out[u] = true; act[u] = false; }
        else if (v[u] == 0xFFFFFFFFFFFFFFFFull) act[u] = false;
        else s[u] = (s[u] + 1) & mask;
        any |= act[u];
      }
  }
}

// Returned by the CHAIN = false instantiations of the certified samplers (the lean table kernels): the draw sits within
// rounding distance of a CDF boundary and only the exact chain can decide — the caller hands the step to a kernel that has it.
constexpr int32_t CHAIN_NEEDED = -2;

// ---- the chain, in pieces (shared by wave_chain_pick and the sharded walk's chain kernels, walk_kernels.hip) ------------
// One group of up to 64 quotients d (lane l holds element l, lanes >= cnt ignored) appended to the accumulator: the lane of
// the first element with acc >= p, or -1 (acc then holds the accumulator after the group).
__device__ inline int chain_group64(double &acc, double d, int cnt, double p) {
  const int lane = lane_id();
  int start = 0;
  while (start < cnt) {
    const unsigned long long ab = (unsigned long long)__double_as_longlong(acc);
    const int ea = (int)((ab >> 52) & 0x7FFull);
    int f = start;                                     // acc zero / subnormal / not finite: one plain addition
    if (!(ea == 0 || ea == 0x7FF || (ab >> 63))) {
      const int e = ea - 1023;
      const unsigned long long N0 = (ab & ((1ull << 52) - 1ull)) | (1ull << 52);
      unsigned long long c0 = 0ull, c1 = 0ull;
      if (lane >= start && lane < cnt) chain_elem_map(d, e, c0, c1);
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long f0 = shfl_up_u64(c0, off), f1 = shfl_up_u64(c1, off);
        if (lane >= off) {
          const unsigned long long g0 = c0, g1 = c1;
          c0 = f0 + ((f0 & 1ull) ? g1 : g0);
          c1 = f1 + (((f1 + 1ull) & 1ull) ? g1 : g0);
        }
      }
      const unsigned long long N = N0 + ((N0 & 1ull) ? c1 : c0);
      const bool mine = lane >= start && lane < cnt;
      const unsigned long long cross = __ballot(mine && N >= (1ull << 53));
      f = cross ? __ffsll((long long)cross) - 1 : -1;
      const int lim = f < 0 ? cnt : f;
      // inside the binade N < 2^53: the accumulator after my element, exactly
      const double a = __longlong_as_double((long long)(((unsigned long long)ea << 52) | (N & ((1ull << 52) - 1ull))));
      const unsigned long long hit = __ballot(lane >= start && lane < lim && a >= p);
      if (hit) return __ffsll((long long)hit) - 1;
      if (f < 0) { acc = readlane_f64(a, cnt - 1); return -1; }
      if (f > start) acc = readlane_f64(a, f - 1);
    }
    acc = acc + readlane_f64(d, f);
    if (acc >= p) return f;
    start = f + 1;
  }
  return -1;
}
// A whole round of NU * 64 quotients at once (lane l holds elements u * 64 + l; elements at or beyond n_valid hold 0.0).
// While the accumulator stays inside its binade and no element sits exactly on a rounding tie (c0 == c1: its increment does
// not depend on the accumulator's parity), the maps N -> N + c commute, so the round is ONE exact integer sum; and with
// every d >= 0 the accumulator is non-decreasing: if it is still below p after the round, no element of the round was the
// answer.  true: absorbed (acc updated).  false: a tie, a binade crossing, the answer's round, a negative / non-finite
// quotient — the caller evaluates the round group by group (chain_group64).
template <int NU>
__device__ inline bool chain_round_fast(double &acc, const double (&d)[NU], double p) {
  const unsigned long long ab = (unsigned long long)__double_as_longlong(acc);
  const int ea = (int)((ab >> 52) & 0x7FFull);
  if (ea == 0 || ea == 0x7FF || (ab >> 63)) return false;
  const unsigned long long N0 = (ab & ((1ull << 52) - 1ull)) | (1ull << 52);
  unsigned long long loc = 0ull; bool odd = false;
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    unsigned long long c0 = 0ull, c1 = 0ull;
    chain_elem_map(d[u], ea - 1023, c0, c1);             // (d == 0: c = 0)
    odd |= (c0 != c1) || (c0 >> 53);
    loc += c0;
  }
  if (__any(odd)) return false;
  const unsigned long long total = wave_sum_u64(loc);
  if (N0 + total >= (1ull << 53)) return false;
  const double a = __longlong_as_double((long long)(((unsigned long long)ea << 52) | ((N0 + total) & ((1ull << 52) - 1ull))));
  if (!(a < p)) return false;
  acc = a;
  return true;
}

// The biased quotients w'_k / S of four candidates per lane; mem (optional, mode 1) = the step's fast membership test: the
// chain runs over the WHOLE row up to the answer, and a draw on a CDF boundary of a hub row used to pay one binary search of
// N(prev) per candidate.  The four membership tests of a lane run in lockstep (independent loads; edge_exists' probe loop
// would serialise them).
__device__ inline void chain_quotients4(const Ent *row, int32_t deg, int32_t base4, const Bias &b, double S, const Member *mem, double (&d4)[4]) {
  const int lane = lane_id();
  Ent e4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int32_t k = base4 + u * 64 + lane; e4[u].id = 0; e4[u].w = 0.0f; if (k < deg) e4[u] = row[k]; }
  if (mem && mem->mode == 1 && b.second_order && b.need_member && (mem->hub || mem->ehash)) {
    uint32_t xs[4]; bool want[4], in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xs[u] = (uint32_t)((int64_t)e4[u].id - b.vmin);
      want[u] = base4 + u * 64 + lane < deg && e4[u].id != b.prev;
      in[u] = false;
    }
    if (mem->hub) {
      uint32_t wd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) wd[u] = want[u] ? mem->hub[xs[u] >> 5] : 0u;
#pragma unroll
      for (int u = 0; u < 4; ++u) in[u] = (wd[u] >> (xs[u] & 31)) & 1u;
    } else {
      edge_exists_n<4>(mem->ehash, mem->ehash_mask, (uint32_t)((int64_t)b.prev - b.vmin), xs, want, in);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      d4[u] = 0.0;
      if (base4 + u * 64 + lane < deg) {
        const float w = e4[u].w;
        d4[u] = (double)(e4[u].id == b.prev ? div_exact(w, b.p) : in[u] ? w : div_exact(w, b.q)) / S;
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int32_t k = base4 + u * 64 + lane;
      d4[u] = 0.0;
      if (k < deg) d4[u] = (double)(mem ? biased_weight_m(b, *mem, k, e4[u].id, e4[u].w) : biased_weight(b, e4[u].id, e4[u].w)) / S;
    }
  }
}

// 256 candidates per round: the four entry loads and the four membership probes of a lane are in flight together.
__device__ inline int32_t wave_chain_pick(const Ent *row, int32_t deg, const Bias &b, float r, double S, const Member *mem = nullptr) {
  const double p = (double)r;
  double acc = 0.0;                                      // wave-uniform
  for (int32_t base4 = 0; base4 < deg; base4 += 256) {
    double d4[4];
    chain_quotients4(row, deg, base4, b, S, mem, d4);
    if (chain_round_fast<4>(acc, d4, p)) continue;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int32_t base = base4 + u * 64;
      if (base >= deg) break;                            // wave-uniform
      const int f = chain_group64(acc, d4[u], min(64, deg - base), p);
      if (f >= 0) return base + f;
    }
  }
  return 0;  // edges.head (:24)
}

__device__ inline int32_t wave_pick(const Ent *row, int32_t deg, const Bias &b, float r, unsigned &fallback) {
  double S = wave_sum_exact_or_chain(row, deg, b, fallback);   // S = foldLeft(0.0)(_ + w')  (:14)
  return wave_chain_pick(row, deg, b, r, S);
}

// Mark the candidates of segment [seg_base, seg_base + seg_len) that occur in N(prev).
// Two strategies over the SORTED rows (both give the same bits):
//   * probe : each (distinct) element of N(prev) is binary-searched in N(curr)      ~ |N(prev)| * log|N(curr)| probes
//   * merge : N(curr) is cut into 64 contiguous slices, one per lane; each lane finds its start in N(prev) by one
//             binary search and then advances both sorted lists linearly              ~ (|N(curr)| + |N(prev)|) / 64 steps
// merge wins when both rows are large (hub -> hub steps, which dominate biased walks on power-law graphs).
__device__ inline void fill_member_bitmap(const Bias &b, Member &m, const uint32_t *curr_sids,
                                          const uint32_t *curr_sperm, int32_t deg, int32_t seg_base, int32_t seg_len) {
  const int lane = lane_id();
  m.seg_base = seg_base;
  const int words = (seg_len + 31) >> 5;
  for (int t = lane; t < words; t += 64) m.bm[t] = 0u;
  __builtin_amdgcn_wave_barrier();
  const int lc = 32 - __clz(deg | 1);
  const bool merge = (int64_t)deg + b.prev_deg < (int64_t)b.prev_deg * lc * 2;
  if (merge) {
    // Block-wise sorted intersection, one contiguous slice of N(curr) per lane: 4 ids of each list are held in
    // registers, all 16 pairs are compared, then the block with the smaller maximum is replaced by the next one,
    // which was requested one step earlier (software prefetch) — no per-element dependent load.
    const int32_t T = (((deg + 63) >> 6) + 3) & ~3;          // slice length, multiple of 4
    const int32_t c0 = lane * T, c1 = min(deg, c0 + T);
    if (c0 < c1) {
      const uint32_t first = curr_sids[c0];
      int32_t j = 0, hi = b.prev_deg;                       // lower_bound of my slice's first id in N(prev)
      while (j < hi) { int32_t mid = j + ((hi - j) >> 1); if (b.prev_sids[mid] < first) j = mid + 1; else hi = mid; }
      const uint32_t PAD = 0xFFFFFFFFu;
      auto load4 = [](const uint32_t *p, int32_t i, int32_t n, uint32_t out[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) out[u] = (i + u < n) ? p[i + u] : 0xFFFFFFFFu;
      };
      uint32_t a[4], bb[4], an[4], bn[4];
      int32_t ca = c0, jb = j;
      load4(curr_sids, ca, c1, a);          load4(b.prev_sids, jb, b.prev_deg, bb);
      load4(curr_sids, ca + 4, c1, an);     load4(b.prev_sids, jb + 4, b.prev_deg, bn);
      unsigned fa = 0;                      // match flags of the current a block
      while (true) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          fa |= ((a[u] == bb[0]) | (a[u] == bb[1]) | (a[u] == bb[2]) | (a[u] == bb[3])) ? (1u << u) : 0u;
        const uint32_t amax = a[3], bmax = bb[3];
        // on a tie only A advances: the next A block may start with more copies of that id (multi-edges) and must
        // still meet the current B block
        const bool adv_a = amax <= bmax, adv_b = !adv_a;
        if (adv_a) {
          if (fa) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (((fa >> u) & 1u) && a[u] != PAD) {
                const int32_t orig = (int32_t)curr_sperm[ca + u] - seg_base;
                if (orig >= 0 && orig < seg_len) atomicOr(&m.bm[orig >> 5], 1u << (orig & 31));
              }
          }
          fa = 0; ca += 4;
          if (ca >= c1) break;
#pragma unroll
          for (int u = 0; u < 4; ++u) a[u] = an[u];
          load4(curr_sids, ca + 4, c1, an);
        }
        if (adv_b) {
          jb += 4;
          if (jb >= b.prev_deg) {            // N(prev) exhausted: flush the flags gathered so far and stop
            if (fa) {
#pragma unroll
              for (int u = 0; u < 4; ++u)
                if (((fa >> u) & 1u) && a[u] != PAD) {
                  const int32_t orig = (int32_t)curr_sperm[ca + u] - seg_base;
                  if (orig >= 0 && orig < seg_len) atomicOr(&m.bm[orig >> 5], 1u << (orig & 31));
                }
            }
            break;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) bb[u] = bn[u];
          load4(b.prev_sids, jb + 4, b.prev_deg, bn);
        }
      }
    }
  } else {
    for (int32_t t = lane; t < b.prev_deg; t += 64) {
      uint32_t x = b.prev_sids[t];
      if (t > 0 && b.prev_sids[t - 1] == x) continue;            // duplicates of a multi-edge: once is enough
      int32_t lo = 0, hi = deg;
      while (lo < hi) { int32_t mid = lo + ((hi - lo) >> 1); if (curr_sids[mid] < x) lo = mid + 1; else hi = mid; }
      for (int32_t pos = lo; pos < deg && curr_sids[pos] == x; ++pos) {
        int32_t orig = (int32_t)curr_sperm[pos] - seg_base;
        if (orig >= 0 && orig < seg_len) atomicOr(&m.bm[orig >> 5], 1u << (orig & 31));
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// Inclusive scan over the 64 lanes with DPP moves (row_shr 1/2/4/8 inside each row of 16 lanes, then row_bcast:15 /
// row_bcast:31 across the rows): 6 x (2 v_mov_dpp + v_add_f64), no LDS traffic and no s_waitcnt — the __shfl_up form
// (2 ds_bpermute per step) was a visible share of the second-order kernel's 1 100 VALU + 1 000 SALU instructions per
// step.  Lanes without a source add 0.0 (the identity), so the result is the same sum in another order; every caller
// either works with exactly representable partial sums or carries a tolerance that covers any summation order.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_add_f64(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xF, false);
  return v + __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}
__device__ inline double wave_incl_scan_f64(double v) {
  v = dpp_add_f64<0x111, 0xF>(v);   // row_shr:1
  v = dpp_add_f64<0x112, 0xF>(v);   // row_shr:2
  v = dpp_add_f64<0x114, 0xF>(v);   // row_shr:4
  v = dpp_add_f64<0x118, 0xF>(v);   // row_shr:8
  v = dpp_add_f64<0x142, 0xA>(v);   // row_bcast:15 -> rows 1 and 3
  v = dpp_add_f64<0x143, 0xC>(v);   // row_bcast:31 -> rows 2 and 3
  return v;
}

// the wave's total by the same DPP steps (lane 63 of the scan) — for sums that are exact in any order (a row certificate holds) or
// whose order is not observable; wave_sum_f64's butterflies stay where the order is part of a result
__device__ inline double wave_total_f64(double v) { return readlane_f64(wave_incl_scan_f64(v), 63); }

// Certified parallel evaluation of RandomSample.sample on the biased row.
//   S: certified-exact parallel sum (else the sequential chain).
//   acc: ANY summation order of the same quotients d_k = w'_k / S differs from the reference's left-to-right
//   acc_k by at most 2*gamma_k*sum(d_i) (Higham, d_i >= 0); with tol_k = (k + 1) * 2^-51 * acc_k a candidate is a
//   CERTAIN miss if acc_k + tol_k < p and a CERTAIN hit if acc_k - tol_k >= p.  The reference's answer is the
//   first index that is not a miss; if the first index that is not a certain miss is a certain hit it IS that
//   answer, otherwise (a draw within rounding distance of a CDF boundary, an exact lattice tie, NaN) the step is
//   redone with the exact sequential chain.  Rows with a negative/NaN/Inf weight go straight to the chain.
__device__ inline int32_t wave_pick_scan(const GraphView &g, const Row &rc, const Bias &b, Member &m, float r,
                                         unsigned &fallback) {
  const int lane = lane_id();
  const Ent *row = g.ent + rc.off;
  const int32_t deg = rc.deg;
  const uint32_t *csids = g.sids + rc.off, *csperm = g.sperm + rc.off;
  // membership strategy
  m.mode = 0;
  m.hub = (b.need_member && b.prev_hub && g.hub_bm) ? g.hub_bm + (int64_t)(b.prev_hub - 1) * g.hub_words : nullptr;
  if (b.need_member) {
    // cost model (in binary-search probes): per-candidate search twice (two passes) vs reverse marking per segment
    int lp = 32 - __clz(b.prev_deg | 1), lc = 32 - __clz(deg | 1);
    int64_t nseg = ((int64_t)deg + BM_BITS - 1) / BM_BITS;
    const int64_t probe = (int64_t)b.prev_deg * lc, merged = ((int64_t)deg + b.prev_deg) / 2;
    int64_t direct = 2ll * deg * (m.hub ? 1 : m.ehash ? 2 : lp), reverse = 2ll * nseg * (probe < merged ? probe : merged) + deg / 16;
    m.mode = reverse < direct ? 2 : 1;
  }
  const int32_t seg_cap = (m.mode == 2) ? BM_BITS : 0x7FFFFFFF;
  // ---- pass 1: S ----
  double part = 0.0;
  SumCert cert;
  bool neg = false;
  for (int32_t sb = 0; sb < deg; sb += seg_cap) {
    int32_t sl = min(seg_cap, deg - sb);
    SRW_T0(m);
    if (m.mode == 2) fill_member_bitmap(b, m, csids, csperm, deg, sb, sl);
    SRW_T1(m, t_fill); SRW_T0(m);
    for (int32_t base = sb; base < sb + sl; base += 256) {
      int32_t k0 = base + lane * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int32_t k = k0 + i;
        if (k < sb + sl) {
          Ent e = row[k];
          float w = biased_weight_m(b, m, k, e.id, e.w);
          part += (double)w;
          cert.add(w);
          neg |= !(w >= 0.0f);
        }
      }
    }
    SRW_T1(m, t_pass1);
  }
  int emin = wave_min_i32(cert.emin), emax = wave_max_i32(cert.emax);
  bool bad = __any(cert.bad) || __any(neg);
  double S;
  SRW_T0(m);
  if (!bad && sum_is_exact(emin, emax, false, deg)) {
    S = wave_sum_f64(part);
  } else {
    unsigned f = 0;
    S = wave_sum_exact_or_chain(row, deg, b, f);
    fallback = 1;
    return wave_chain_pick(row, deg, b, r, S);
  }
  // ---- pass 2: certified scan ----
  const double p = (double)r;
  double carry = 0.0;
  const bool single_seg = deg <= seg_cap;
  for (int32_t sb = 0; sb < deg; sb += seg_cap) {
    int32_t sl = min(seg_cap, deg - sb);
    if (m.mode == 2 && !single_seg) fill_member_bitmap(b, m, csids, csperm, deg, sb, sl);
    for (int32_t base = sb; base < sb + sl; base += 256) {
      int32_t k0 = base + lane * 4;
      double l[4];
      double run = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int32_t k = k0 + i;
        double d = 0.0;
        if (k < sb + sl) { Ent e = row[k]; d = (double)biased_weight_m(b, m, k, e.id, e.w) / S; }
        run += d;
        l[i] = run;
      }
      double incl = wave_incl_scan_f64(run);
      double excl = __shfl_up(incl, 1);               // pure additions only: no (incl - run) cancellation
      if (lane == 0) excl = 0.0;
      double lane_base = carry + excl;
      int first = 4;          // first of my 4 candidates that is not a certain miss
      bool sure = false;
#pragma unroll
      for (int i = 3; i >= 0; --i) {
        int32_t k = k0 + i;
        double acc = lane_base + l[i];
        double tol = (double)(k + 1) * 0x1p-51 * acc;
        bool valid = k < sb + sl;
        bool miss = acc + tol < p;
        if (valid && !miss) { first = i; sure = (acc - tol >= p); }
      }
      unsigned long long cand = __ballot(first < 4);
      if (cand) {
        int fl = __ffsll((long long)cand) - 1;
        int fi = __builtin_amdgcn_readlane(first, fl);
        int fs = __builtin_amdgcn_readlane((int)sure, fl);
        SRW_T1(m, t_pass2);
        if (fs) return base + fl * 4 + fi;
        fallback = 1;                                     // within rounding distance of a boundary: exact chain
        return wave_chain_pick(row, deg, b, r, S);
      }
      carry += readlane_f64(incl, 63);
    }
  }
  return 0;  // edges.head (:24)
}

// ---- exact pick by SEARCH over exact prefix sums (general p, q; large rows) --------------------------------------
// Under the row certificate of sampler_tables.hip:k_pq_* every sum of candidate variants is exact, so
//   A'_k = sum_{i<=k} w'_i = PQ[k] + sum over the "special" positions <= k of (w'_i - fl(w_i/q))
// is exact, where PQ is the per-call prefix sum of the base weights fl(w/q) and the specials are the return edges
// (w' = fl(w/p)) and the members of N(prev) (w' = w) — found by looking the few elements of N(prev) up in the sorted
// N(curr).  S = A'_{deg-1} is the reference's sum bit for bit.  The reference's acc_k = sum of fl(w'_i/S) differs from
// X_k = A'_k / S by at most (k+2) u X_k (one rounding per quotient + the summation), so with
// tol_k = (k+8) 2^-51 X_k the first k that is not a certain miss (X_k + tol_k < p) is found by a 64-ary search
// (X, tol non-decreasing), and if it is a certain hit (X_k - tol_k >= p) it IS the reference's answer; otherwise the
// step is redone with the sequential chain.  Cost: O(|N(prev)| log deg + |specials| log_64 deg) instead of O(deg).
constexpr int SP_CAP = 512;   // specials kept in the wave's LDS scratch: pos[512] | corr[512] (f64) | counter

__device__ inline int32_t wave_pick_prefix(const GraphView &g, const Row &rc, int64_t curr_slot, const Bias &b,
                                           uint32_t *lds, float r, unsigned &fallback, unsigned &served) {
  if (!pq_ready(g) || !b.second_order) return -1;
  const int32_t deg = rc.deg;
  if (deg < 128 || !(rc.flags & ROW_PQ_OK)) return -1;
  const int lc = 32 - __clz(deg | 1);
  if (b.need_member && (int64_t)b.prev_deg * (lc + 2) > (int64_t)deg * 2) return -1;   // marking would cost more than streaming
  if (b.need_member && b.prev_deg > 2048) return -1;   // two large rows share too many neighbors for the LDS list
  const int lane = lane_id();
  uint32_t *sp_pos = lds;
  double *sp_corr = reinterpret_cast<double *>(lds + SP_CAP);
  uint32_t *counter = lds + 3 * SP_CAP;
  if (lane == 0) *counter = 0u;
  __builtin_amdgcn_wave_barrier();
  const Ent *row = g.ent + rc.off;
  const uint32_t *cs = g.sids + rc.off, *cp = g.sperm + rc.off;
  const uint32_t xprev = (uint32_t)((int64_t)b.prev - b.vmin);
  // (a) return edges: occurrences of prev in N(curr)
  {
    int32_t lo = 0, hi = deg;
    while (lo < hi) { int32_t mid = lo + ((hi - lo) >> 1); if (cs[mid] < xprev) lo = mid + 1; else hi = mid; }
    for (int32_t c = lo + lane; c < deg && cs[c] == xprev; c += 64) {
      const uint32_t orig = cp[c];
      const float w = row[orig].w;
      const uint32_t idx = atomicAdd(counter, 1u);
      if (idx < (uint32_t)SP_CAP) { sp_pos[idx] = orig; sp_corr[idx] = (double)div_exact(w, b.p) - (double)div_exact(w, b.q); }
    }
  }
  // (b) members of N(prev) (only matter when q != 1)
  if (b.need_member) {
    for (int32_t t = lane; t < b.prev_deg; t += 64) {
      const uint32_t x = b.prev_sids[t];
      if (x == xprev || (t > 0 && b.prev_sids[t - 1] == x)) continue;
      int32_t lo = 0, hi = deg;
      while (lo < hi) { int32_t mid = lo + ((hi - lo) >> 1); if (cs[mid] < x) lo = mid + 1; else hi = mid; }
      for (int32_t c = lo; c < deg && cs[c] == x; ++c) {
        const uint32_t orig = cp[c];
        const float w = row[orig].w;
        const uint32_t idx = atomicAdd(counter, 1u);
        if (idx < (uint32_t)SP_CAP) { sp_pos[idx] = orig; sp_corr[idx] = (double)w - (double)div_exact(w, b.q); }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  const uint32_t n_sp = *counter;
  if (n_sp > (uint32_t)SP_CAP) return -1;          // too many specials for the scratch: streaming scan instead
  double csum = 0.0;
  for (uint32_t j = lane; j < n_sp; j += 64) csum += sp_corr[j];
  csum = wave_sum_f64(csum);                       // exact under the certificate
  const PqRow PQ(g, rc.off);
  const double S = PQ[deg - 1] + csum;
  const double p = (double)r;
  auto not_miss = [&](int32_t k, bool &hit) {
    double a = PQ[k];
    for (uint32_t j = 0; j < n_sp; ++j) a += (sp_pos[j] <= (uint32_t)k) ? sp_corr[j] : 0.0;   // LDS broadcast reads
    const double X = a / S;
    const double tol = (double)(k + 8) * 0x1p-51 * X;
    hit = X - tol >= p;
    return !(X + tol < p);
  };
  int32_t lo = 0, hi = deg - 1;                    // the first not-certain-miss index, if any, lies in [lo, hi]
  bool hit = false;
  while (hi - lo >= 64) {
    const int64_t span = (int64_t)hi - lo;
    const int32_t k = lo + (int32_t)((span * (lane + 1)) >> 6);          // lane 63 probes hi
    bool h;
    const bool nm = not_miss(k, h);
    const unsigned long long m = __ballot(nm);
    if (!m) return 0;                               // even k = hi is a certain miss: no crossing -> edges.head
    const int f = __ffsll((long long)m) - 1;
    const int32_t kf = __builtin_amdgcn_readlane(k, f);
    const int32_t kprev = f ? __builtin_amdgcn_readlane(k, f - 1) : lo - 1;
    hi = kf; lo = kprev + 1;
  }
  {
    const int32_t k = lo + lane;
    bool h = false, nm = false;
    if (k <= hi) nm = not_miss(k, h);
    const unsigned long long m = __ballot(nm);
    if (!m) return 0;
    const int f = __ffsll((long long)m) - 1;
    hit = __builtin_amdgcn_readlane((int)h, f) != 0;
    const int32_t kf = lo + f;
    served = 1;
    if (hit) return kf;
  }
  fallback = 1;
  return wave_chain_pick(row, deg, b, r, S);
}

// ---- q == 1, ANY number of return edges (hub <-> hub multi-edges, self-loops of a hub: hundreds of them) ------------------
// The only biased candidates are the nr return edges, a RUN [so, so + nr) of curr's sorted row whose input-order positions
// (sperm) increase along the run (the rows are sorted by a stable sort).  Between two of them A'_k = PQ[k] + C with C constant, so
//   1. the run is walked 64 at a time with a wave scan of the corrections: the first return edge that is not a certain miss
//      bounds the answer to the interval behind its predecessor;
//   2. a 64-ary search over the exact prefix sums of that interval (constant correction) finds the first k that is not a
//      certain miss; a certain hit is the reference's answer, else the chain (CHAIN = false: CHAIN_NEEDED, S in *S_out).
// O(nr / 64 + log_64 deg) wave steps and no LDS, where wave_pick_prefix keeps at most SP_CAP specials and adds all of them up
// per probe, and the per-lane q1_pick walks the run once per prefix value.  -1: the row has no usable prefix sums.
template <bool CHAIN = true>
__device__ inline int32_t wave_pick_returns(const GraphView &g, const Row &rc, const Bias &b, int64_t so, int32_t nr, float r,
                                            unsigned &fallback, double *S_out = nullptr) {
  const int32_t deg = rc.deg;
  if (!pq_ready(g) || !(rc.flags & ROW_PQ_OK) || deg < 1) return -1;
  const int lane = lane_id();
  const PqRow PQ(g, rc.off);
  auto corr_of = [&](int64_t i) { const float w = g.sw[i]; return (double)div_exact(w, b.p) - (double)w; };
  double cs = 0.0;
  for (int32_t i = lane; i < nr; i += 64) cs += corr_of(so + i);
  cs = wave_sum_f64(cs);                                  // exact under the certificate, in any order
  const double S0 = PQ[deg - 1], S = S0 + cs;
  if (!(S > 0.0) || !(S0 > 0.0)) return -1;
  if (!CHAIN && S_out && g.dbg_chain_deg && deg >= g.dbg_chain_deg) { *S_out = S; return CHAIN_NEEDED; }
  const double pS = (double)r * S;
  auto not_miss = [&](int32_t kk, double num) { return !(num * (1.0 + (double)(kk + 8) * 0x1p-51) < pS); };
  // 1. the first return edge that is not a certain miss
  int32_t lo = 0, ans = -1;
  double C = 0.0, ans_num = 0.0, carry = 0.0;
  int32_t last_pos = -1;
  for (int32_t base = 0; base < nr; base += 64) {
    const int32_t i = base + lane;
    const bool valid = i < nr;
    const int32_t pos = valid ? (int32_t)g.sperm[so + i] : deg;
    const double c = valid ? corr_of(so + i) : 0.0;
    const double incl = wave_incl_scan_f64(c);
    const double num = valid ? PQ[pos] + (carry + incl) : 0.0;
    const unsigned long long m = __ballot(valid && not_miss(pos, num));
    if (m) {
      const int f = __ffsll((long long)m) - 1;
      ans = __builtin_amdgcn_readlane(pos, f); ans_num = readlane_f64(num, f);
      C = readlane_f64(carry + (incl - c), f);
      lo = f ? __builtin_amdgcn_readlane(pos, f - 1) + 1 : last_pos + 1;
      break;
    }
    const int nv = min(64, nr - base);
    carry += readlane_f64(incl, 63);
    last_pos = __builtin_amdgcn_readlane(pos, nv - 1);
  }
  if (ans < 0) { lo = last_pos + 1; C = carry; }
  // 2. the interval [lo, hi] in front of it (behind the last return edge when there is none): constant correction C
  int32_t hi = ans >= 0 ? ans - 1 : deg - 1;
  bool none = false;
  while (hi - lo >= 64) {
    const int64_t span = (int64_t)hi - lo;
    const int32_t k = lo + (int32_t)((span * (lane + 1)) >> 6);          // lane 63 probes hi
    const unsigned long long m = __ballot(not_miss(k, PQ[k] + C));
    if (!m) { none = true; break; }                  // even k = hi is a certain miss
    const int f = __ffsll((long long)m) - 1;
    const int32_t kf = __builtin_amdgcn_readlane(k, f);
    const int32_t kprev = f ? __builtin_amdgcn_readlane(k, f - 1) : lo - 1;
    hi = kf; lo = kprev + 1;
  }
  if (!none && lo <= hi) {
    const int32_t k = lo + lane;
    const double num = k <= hi ? PQ[k] + C : 0.0;
    const unsigned long long m = __ballot(k <= hi && not_miss(k, num));
    if (m) { const int f = __ffsll((long long)m) - 1; ans = lo + f; ans_num = readlane_f64(num, f); }
  }
  if (ans < 0) return 0;                             // no crossing: edges.head (:24)
  if (ans_num * (1.0 - (double)(ans + 8) * 0x1p-51) >= pS) return ans;     // a certain hit
  if (!CHAIN) { if (S_out) *S_out = S; return CHAIN_NEEDED; }
  fallback = 1;
  return wave_chain_pick(g.ent + rc.off, deg, b, r, S);
}

// ---- exact pick by search over exact prefix sums, ANY number of specials (q != 1; hub -> hub steps) ----------------
// Same mathematics as wave_pick_prefix, but the corrections of the specials are not kept as a list: each one is
// added (exactly — the row certificate makes every sum of variant differences exact in any order) into an LDS bin
// of its input-order position, bins[pos >> csh].  Then
//   A'_{end of chunk j} = PQ[end_j] + sum(bins[0..j])          (exact)
// locates, by a 64-ary search over the chunk ends with the same certified tolerance, the ONE chunk that holds the
// first not-certain-miss index; only that chunk (64 .. a few thousand candidates) is evaluated candidate by
// candidate.  No scan of N(curr): the step costs the membership work plus O(chunk).
// The members of N(prev) among the candidates are found by the cheapest of four strategies:
//   P1  every distinct id of N(prev) is searched in the sorted N(curr)         ~ |N(prev)| log |N(curr)| probes
//   P2  every candidate (input order) is probed in the edge hash set            ~ |N(curr)| probes
//   P3  prev is a hub with a neighbor-set bitmap over the id slots (GraphView::hub_bm): one bit read per candidate
//   W   sorted-chunk intersection: 1024 ids of N(prev) at a time are staged in LDS in sorted order, the candidates
//       come 256 at a time (16-byte loads; ids, positions and weights) and do a branch-free 10-level lower bound in
//       LDS, four per lane in lockstep; whichever list ends first in id order advances.  At most
//       |N(curr)| / 256 + |N(prev)| / 1024 rounds, independent of the id range.  (Tried before it: an id-window bitmap
//       — 6 us per window, 300 windows per step at RMAT-24 — and an LDS hash set per chunk — 3x the LDS operations.)
#ifndef SRW_W_MATCH_LOOP
#define SRW_W_MATCH_LOOP 1                    // sorted-chunk intersection: the matches of a lane in a uniform loop (0: four exec-masked blocks, as before round 5)
#endif
constexpr int BIN_CAP = 512;                  // f64 bins: 4 KB of the wave's LDS
constexpr int WIN_WORDS = 1280;               // scratch behind the bins: 5 KB (W stages 1024 sorted ids of N(prev) here)
constexpr int HCHUNK = 1024;                // ids of N(prev) staged in LDS per round
constexpr uint32_t HEMPTY = 0xFFFFFFFFu;
constexpr int BINNED_LDS_WORDS = 2 * BIN_CAP + WIN_WORDS;
struct __attribute__((packed, aligned(4))) U32x4 { uint32_t a, b, c, d; };   // 16-byte load at 4-byte alignment
struct __attribute__((packed, aligned(4))) F32x4 { float a, b, c, d; };

__device__ inline int32_t wave_lower_bound_u32(const uint32_t *a, int32_t n, uint32_t x) {   // wave-uniform, 64-ary
  const int lane = lane_id();
  int32_t lo = 0, hi = n;
  while (true) {
    const int32_t span = hi - lo;
    if (span <= 0) return lo;
    if (span <= 64) {
      const int32_t k = lo + lane;
      const bool ge = k < hi && a[k] >= x;
      const unsigned long long m = __ballot(ge);
      return m ? lo + (__ffsll((long long)m) - 1) : hi;
    }
    const int32_t k = lo + (int32_t)(((int64_t)span * lane) >> 6);
    const bool ge = a[k] >= x;
    const unsigned long long m = __ballot(ge);
    if (!m) { lo = __builtin_amdgcn_readlane(k, 63) + 1; continue; }
    const int f = __ffsll((long long)m) - 1;
    hi = __builtin_amdgcn_readlane(k, f);
    if (f) lo = __builtin_amdgcn_readlane(k, f - 1) + 1;
  }
}

// K wave-uniform lower bounds advanced in lockstep: the probes of one round are independent loads, so K searches cost
// the round trips of one.
struct LowerBound { const uint32_t *a; int32_t lo, hi; uint32_t x; bool done; };
template <int K>
__device__ inline void wave_lower_bound_multi(LowerBound (&s)[K]) {
  const int lane = lane_id();
  while (true) {
    bool act[K], ge[K];
    int32_t k[K];
    bool any = false;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      act[i] = false; ge[i] = false; k[i] = 0;
      if (s[i].done) continue;
      const int32_t span = s[i].hi - s[i].lo;
      if (span <= 0) { s[i].done = true; continue; }
      act[i] = true; any = true;
      k[i] = span <= 64 ? s[i].lo + lane : s[i].lo + (int32_t)(((int64_t)span * lane) >> 6);
      ge[i] = k[i] < s[i].hi && s[i].a[k[i]] >= s[i].x;
    }
    if (!any) break;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      if (!act[i]) continue;
      const int32_t span = s[i].hi - s[i].lo;
      const unsigned long long m = __ballot(ge[i]);
      if (span <= 64) { s[i].lo = m ? s[i].lo + (__ffsll((long long)m) - 1) : s[i].hi; s[i].done = true; continue; }
      if (!m) { s[i].lo = __builtin_amdgcn_readlane(k[i], 63) + 1; continue; }
      const int f = __ffsll((long long)m) - 1;
      s[i].hi = __builtin_amdgcn_readlane(k[i], f);
      if (f) s[i].lo = __builtin_amdgcn_readlane(k[i], f - 1) + 1;
    }
  }
}

// Chunking of a row's candidate positions: chunk j = positions [j << csh, (j + 1) << csh), at most `cap` chunks.
struct BinGeom { int csh; int32_t n_bins; };
// ---- geometry of the per-edge tables (edge_tables.hip), shared by the planner, the builder and the walk -------------------------
// A pair (prev -> curr) with dv candidates in N(curr) and du ids in N(prev) gets chunks of 2^csh candidates:
//   * the base rule: at least 2^min_sh candidates per chunk, at most `cap` chunks (bin_geometry);
//   * rows of at most cm_max candidates whose N(prev) is longer than cm_min_du (what the walk stages in LDS and searches
//     there for free) also get CHUNK MASKS: the pair's membership bits over the candidate positions, one 64-bit word per 64
//     positions — the located chunk is then evaluated without a single membership probe (profiles/r04_request_attribution.md:
//     the probes were 48 of the 62 HBM requests of an average step at config 3);
//   * pairs WITHOUT a mask whose N(prev) is longer than fine_min_du — every candidate of their located chunk costs one probe, one
//     HBM request — get finer tables: chunks of at least 2^fine_sh candidates, at most fine_cap of them (0: off).
// A table of more than 64 chunks is a 64-ary tree: level 0 = the n chunk prefixes, level 1 = every 64th of them (the last
// element of each block of 64), level 2 = every 64th of level 1 — a search reads one block of <= 64 values per level (<= 4 lines
// of floats) instead of striding over the whole table, and S = the last element of the top level.  Layout of a pair's block,
// each part padded to 64 bytes: [level 2][level 1][level 0][chunk masks].
struct PairGeom { int csh; int32_t n_bins; bool cmask; };
constexpr int32_t EB_CM_LIMIT = 16384;         // the build keeps the mask of one pair in 2 KB of the wave's LDS; longer rows: straight into the table block
__host__ __device__ inline BinGeom bin_geometry(int32_t deg, int min_sh, int cap);
__host__ __device__ inline PairGeom eb_pair_geometry(int32_t dv, int32_t du, const EbPolicy &P) {
  const BinGeom b = bin_geometry(dv, P.min_sh, P.cap);
  PairGeom g; g.csh = b.csh; g.n_bins = b.n_bins; g.cmask = false;
  // (a shorter N(prev) is staged in LDS by the walk — deg(prev) / 16 lines per step — unless the row is short enough for the mask to be the
  //  cheaper thing to keep: deg(curr) / 8 bytes of mask against deg(prev) x 4 bytes of staging traffic per visit, cm_ratio = deg(curr) / deg(prev))
  // (round 6: cm_max may exceed EB_CM_LIMIT — LONG masks for the pairs whose N(prev) is too long for the LDS staging, the hub -> hub pairs whose
  //  located chunks otherwise probe prev's bitmap once per candidate: half of config 3's memory requests; the ratio rule stays with the short rows)
  if (dv <= P.cm_max && b.csh >= 6 && (du > P.cm_min_du || (dv <= EB_CM_LIMIT && du > 32 && (int64_t)dv <= (int64_t)P.cm_ratio * du))) g.cmask = true;
  else if (P.fine_cap > 0 && du > P.fine_min_du && P.fine_sh >= 6) {
    const BinGeom f = bin_geometry(dv, P.fine_sh, P.fine_cap);
    if (f.csh < b.csh) { g.csh = f.csh; g.n_bins = f.n_bins; }
  }
  return g;
}
__host__ __device__ inline uint32_t eb_prefix_units(bool f32, int32_t n_bins) { return f32 ? (uint32_t)((n_bins + 15) >> 4) : (uint32_t)((n_bins + 7) >> 3); }
__host__ __device__ inline uint32_t eb_cmask_units(int32_t deg) { return (uint32_t)((((deg + 63) >> 6) + 7) >> 3); }
// 16-bit level 0 (EbPolicy::u16): where the row's flags say that a chunk of this pair's size weighs less than 65 536 units of 2^G
// (sampler_tables.hip:pq_row_u16_bits), level 0 holds the chunks' own masses as u16 multiples of 2^G instead of the absolute prefixes —
// half (a quarter, for f64 rows) of the bytes of the level that IS the table; the search adds a block of 64 up with a wave scan on top
// of the absolute prefix that the level above (or nothing, for the first block) provides.  Levels 1 and 2 stay absolute.
__host__ __device__ inline bool eb_pair_u16(uint32_t row_flags, int csh, const EbPolicy &P) {
  const int c = (int)((row_flags >> ROW_U16_SHIFT) & 7u);
  return P.u16 && c > 0 && csh >= 6 && csh <= c + 5;
}
__host__ __device__ inline double eb_row_unit(uint32_t row_flags) {        // 2^G
  const int G = (int)((row_flags >> ROW_G_SHIFT) & 0xFFu) - 128;
  union { unsigned long long u; double d; } x; x.u = (unsigned long long)(1023 + G) << 52;     // (|G| <= 100: a normal double)
  return x.d;
}
struct EbLayout { uint32_t l2_off, l1_off, l0_off, cm_off, units; int32_t n1, n2; };   // offsets in 64-byte units from the pair's block
__host__ __device__ inline EbLayout eb_layout(bool f32, int32_t n_bins, bool cmask, int32_t dv, bool u16 = false) {
  EbLayout L;
  L.n1 = n_bins > 64 ? (n_bins + 63) >> 6 : 0;
  L.n2 = L.n1 > 64 ? (L.n1 + 63) >> 6 : 0;          // (at most 64: tables have at most 2^18 chunks)
  L.l2_off = 0u;
  L.l1_off = L.n2 ? eb_prefix_units(f32, L.n2) : 0u;
  L.l0_off = L.l1_off + (L.n1 ? eb_prefix_units(f32, L.n1) : 0u);
  L.cm_off = L.l0_off + (u16 ? (uint32_t)((n_bins + 31) >> 5) : eb_prefix_units(f32, n_bins));
  L.units = L.cm_off + (cmask ? eb_cmask_units(dv) : 0u);
  return L;
}
constexpr int32_t EB_FINE_CAP_LIMIT = 32768;     // chunks of a table at most (the build keeps one f64 per chunk and wave in an HBM scratch beyond BIN_CAP)
__host__ __device__ inline BinGeom bin_geometry(int32_t deg, int min_sh, int cap) {
  // smallest csh >= min_sh with ceil(deg / 2^csh) <= cap, i.e. cap * 2^csh >= deg, in closed form (the walk computes this per table step
  // on the scalar unit: profiles/r04_valu_issue.md).  With c0 = bit_length(deg - 1) - bit_length(cap): no c < c0 can do
  // (cap * 2^c < 2^(bit_length(deg - 1) - 1) <= deg - 1) and c0 + 1 always does (cap * 2^(c0 + 1) >= 2^bit_length(deg - 1) >= deg).
  BinGeom g; g.csh = min_sh;
  if (deg > 1 && cap > 0) {
    const int c0 = (32 - __builtin_clz((unsigned)(deg - 1))) - (32 - __builtin_clz((unsigned)cap));
    if (c0 > g.csh) g.csh = c0;
    if (((int64_t)cap << g.csh) < (int64_t)deg) ++g.csh;
  }
  g.n_bins = (int32_t)(((int64_t)deg + ((int64_t)1 << g.csh) - 1) >> g.csh);
  return g;
}

// Rough cost (wave-cycles) of finding N(prev) ∩ N(curr) with the cheapest membership strategy — the model the binned
// search chooses its strategy with, and the one the per-edge tables (edge_tables.hip) are prioritised by.
struct BinnedCost { int64_t c1, c2, cw; };
__host__ __device__ inline int bit_length_i32(int32_t x) { int l = 0; while (l < 31 && ((int32_t)1 << l) <= x) ++l; return l; }   // 32 - clz(x | 1)
__host__ __device__ inline BinnedCost binned_cost(int32_t deg, int32_t m, bool hubbits, bool ehash) {
  const int lc = bit_length_i32(deg | 1), lp = bit_length_i32(m | 1);
  BinnedCost c;
  // a dependent probe chain ~ 10 cycles per level per element (5 with two in lockstep)
  // per candidate: P3 = one bit read (a random sector, like the hash probe, but no hashing and no probe loop)
  c.c1 = (int64_t)m * lc * 5; c.c2 = (int64_t)deg * (hubbits ? 3 : ehash ? 6 : lp) * 10;
  c.cw = (int64_t)deg * 2 + (int64_t)m / 2 + 3000;      // ~2000 cycles per round of 256 / 1024 ids
  return c;
}

// Fills the wave's LDS bins with the EXACT inclusive prefix, by chunk, of the corrections of the specials of this
// (prev, curr) pair: bins[j] = sum over positions k < ((j + 1) << csh) of (w'_k - fl(w_k / q)).
// tune: 0 = automatic strategy, 1 = P1, 2 = P2, 3 = W, 4 = P3 if prev has a bitmap (tests force each one)
// PF: rounds of candidates the sorted-chunk intersection keeps in flight ahead of the one it works on (12 VGPRs each); P1K: binary
// searches per lane in lockstep in P1.  Both are experiment parameters: 2 / 8 did not move the table build (profiles/r03_eb_build.md)
// GB (edge_tables.hip, tables of more chunks than BIN_CAP): the bins are `gbins`, a slice of an HBM scratch owned by this wave
// HC: ids of N(prev) staged per chunk (a power of two >= 1 024; the table build stages 2 048: half the chunk advances of a long N(prev))
template <int PF = 1, int P1K = 2, bool GB = false, int HC = HCHUNK>
__device__ inline void binned_fill(const GraphView &g, const Row &rc, const Bias &b, uint32_t *lds, int tune,
                                   const BinGeom geo, Member &tm, unsigned long long &alg_bytes, unsigned &strat_used,
                                   uint32_t *mbits = nullptr /* edge_tables.hip: LDS bitmap over curr's positions, bit k = candidate k is in N(prev) */,
                                   double *gbins = nullptr, bool mbits_in_hbm = false /* long masks: mbits points into the pair's table block */) {
  const int32_t deg = uni(rc.deg);                    // (one wave, one pair: scalar state — wave_primitives.h:uni)
  const int lane = lane_id();
  double *bins = GB ? gbins : reinterpret_cast<double *>(lds);
  uint32_t *win = lds + 2 * BIN_CAP;
  SRW_T0(tm);
  const int csh = uni(geo.csh);
  const int32_t n_bins = uni(geo.n_bins);
  const int64_t roff = uni(rc.off);
  for (int t = lane; t < n_bins; t += 64) bins[t] = 0.0;
  if (mbits) for (int t = lane; t < ((deg + 31) >> 5); t += 64) mbits[t] = 0u;
  if (GB || mbits_in_hbm) __threadfence();            // the zeros reach L2 before the first atomic does
  __builtin_amdgcn_wave_barrier();
  const Ent *row = g.ent + roff;
  const uint32_t *cs = g.sids + roff, *cp = g.sperm + roff;
  const float *csw = g.sw + roff;
  const uint32_t *B = uni(b.prev_sids);
  const int32_t m = uni(b.prev_deg);
  const uint32_t xprev = uni((uint32_t)((int64_t)b.prev - b.vmin));
  const float p_ = b.p, q_ = b.q;
  // strategy for (b) first, so that every wave-uniform lower bound this step needs is searched in one lockstep pass
  int strat = tune;
  const uint32_t prev_hub = uni(b.prev_hub);
  const uint32_t *hubbits = (prev_hub && g.hub_bm) ? g.hub_bm + (int64_t)(prev_hub - 1) * g.hub_words : nullptr;
  if (strat == 4 && !hubbits) strat = 0;               // forced P3 without a bitmap: automatic choice
  uint32_t lo_id = 1u, hi_id = 0u;
  if (m > 0 && (strat == 0 || strat == 3)) {
    lo_id = uni(max(cs[0], B[0])); hi_id = uni(min(cs[deg - 1], B[m - 1]));
    if (strat == 0) {
      BinnedCost bc = binned_cost(deg, m, hubbits != nullptr, g.ehash != nullptr);
      if (P1K > 2) bc.c1 = bc.c1 * 4 / P1K;                      // the searches with P1K (not two) in lockstep; measured flat between 2 / P1K and 4 / P1K
      strat = (bc.cw < bc.c1 && bc.cw < bc.c2) ? 3 : (bc.c1 <= bc.c2 ? 1 : (hubbits ? 4 : 2));
    }
  }
  int32_t ret_lo = deg, pa = 0, pb = 0;
  if (strat == 4 && m > 0) {
    // P3: prev is a hub with a neighbor-set bitmap over the id slots.  One streaming pass over N(curr) in INPUT order
    // (entries as they are, no sorted structure): a candidate is a member iff its bit is set — one L2-resident read
    // each, four in flight per lane — and the return edges are recognised on the way.
    alg_bytes += 12ull * (unsigned long long)deg;        // 8-byte entries + one bitmap word per candidate
    for (int32_t k0 = 0; k0 < deg; k0 += 256) {
      Ent e[4]; uint32_t wd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int32_t k = k0 + u * 64 + lane; e[u].id = b.prev; e[u].w = 0.0f; if (k < deg) e[u] = row[k]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const uint32_t x = (uint32_t)((int64_t)e[u].id - b.vmin); wd[u] = hubbits[x >> 5]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int32_t k = k0 + u * 64 + lane;
        if (k >= deg) continue;
        const uint32_t x = (uint32_t)((int64_t)e[u].id - b.vmin);
        if (e[u].id == b.prev) atomicAdd(&bins[k >> csh], (double)div_exact(e[u].w, p_) - (double)div_exact(e[u].w, q_));
        else if ((wd[u] >> (x & 31)) & 1u) {
          atomicAdd(&bins[k >> csh], (double)e[u].w - (double)div_exact(e[u].w, q_));
          if (mbits) atomicOr(&mbits[k >> 5], 1u << (k & 31));
        }
      }
    }
  } else if (strat == 3 && lo_id <= hi_id) {
    LowerBound lb[3] = {{cs, 0, deg, xprev, false}, {cs, 0, deg, lo_id, false}, {B, 0, m, lo_id, false}};
    wave_lower_bound_multi<3>(lb);
    ret_lo = lb[0].lo; pa = lb[1].lo; pb = lb[2].lo;
  } else {
    ret_lo = wave_lower_bound_u32(cs, deg, xprev);
  }
  // (a) return edges: occurrences of prev in N(curr)
  for (int32_t c = ret_lo + lane; c < deg && cs[c] == xprev; c += 64)
    atomicAdd(&bins[cp[c] >> csh], (double)div_exact(csw[c], p_) - (double)div_exact(csw[c], q_));
  SRW_T1(tm, t_a); SRW_T0(tm);
  // (b) members of N(prev)
  if (m > 0 && strat != 4) {
#ifdef SRW_PHASE_TIMING
    tm.n_binned += 1;
    if (strat == 1) { tm.n_p1 += 1; tm.n_p1_elems += m; }
#endif
    if (strat == 1) alg_bytes += 8ull * (unsigned long long)m;   // the ids looked up + the entry each one lands on (probes above it: cache)
    else if (strat == 2) alg_bytes += 16ull * (unsigned long long)deg;                                            // entries + hash slots
    else alg_bytes += 12ull * (unsigned long long)(deg - pa) + 4ull * (unsigned long long)(m - pb);              // both sorted rows
    if (strat == 1) {
      // P1K searches per lane in lockstep: P1K loads in flight on the dependent probe chain
      for (int32_t t0 = lane; t0 < m; t0 += 64 * P1K) {
        uint32_t x[P1K]; bool act[P1K]; int32_t lo[P1K], hi[P1K];
        bool any = false;
#pragma unroll
        for (int i = 0; i < P1K; ++i) {
          const int32_t tt = t0 + 64 * i;
          act[i] = tt < m;
          x[i] = act[i] ? B[tt] : 0u;
          if (act[i] && (x[i] == xprev || (tt > 0 && B[tt - 1] == x[i]))) act[i] = false;
          lo[i] = 0; hi[i] = act[i] ? deg : 0;
          any |= act[i];
        }
        while (any) {
          uint32_t v[P1K]; int32_t mid[P1K];
#pragma unroll
          for (int i = 0; i < P1K; ++i) { mid[i] = lo[i] + ((hi[i] - lo[i]) >> 1); v[i] = lo[i] < hi[i] ? cs[mid[i]] : 0u; }
          any = false;
#pragma unroll
          for (int i = 0; i < P1K; ++i) {
            if (lo[i] < hi[i]) { if (v[i] < x[i]) lo[i] = mid[i] + 1; else hi[i] = mid[i]; }
            any |= lo[i] < hi[i];
          }
        }
#pragma unroll
        for (int i = 0; i < P1K; ++i)
          if (act[i])
            for (int32_t c = lo[i]; c < deg && cs[c] == x[i]; ++c) {
              const uint32_t orig = cp[c];
              const float w = csw[c];
              atomicAdd(&bins[orig >> csh], (double)w - (double)div_exact(w, q_));
              if (mbits) atomicOr(&mbits[orig >> 5], 1u << (orig & 31));
            }
      }
    } else if (strat == 2) {
      for (int32_t k = lane; k < deg; k += 64) {
        const Ent e = row[k];
        if (e.id == b.prev) continue;
        const uint32_t xs = (uint32_t)((int64_t)e.id - b.vmin);
        if (g.ehash ? edge_exists(g.ehash, g.ehash_mask, xprev, xs) : sorted_contains(B, m, xs)) {
          atomicAdd(&bins[k >> csh], (double)e.w - (double)div_exact(e.w, q_));
          if (mbits) atomicOr(&mbits[k >> 5], 1u << (k & 31));
        }
      }
    } else {
      if (lo_id <= hi_id) {
        // Sorted-chunk intersection.  1024 ids of N(prev) are staged in LDS in sorted order (16-byte loads, padded with
        // 0xFFFFFFFF); the candidates come 256 at a time (4 per lane, with their input-order positions and weights)
        // and every candidate not above the staged chunk's last id does a branch-free 10-level lower bound in LDS,
        // four searches per lane in lockstep.  Whichever list ends first in id order advances: at most
        // |N(curr)| / 256 + |N(prev)| / 1024 rounds, independent of the id range.
        uint32_t *bch = win;                                 // 1024 words of the 1280-word region
        uint32_t AI[4], AC[4], NI[PF][4], NC[PF][4];
        float AW[4], NW[PF][4];
        auto load_a = [&](int32_t pos, uint32_t v[4], uint32_t c[4], float w[4]) {   // lane holds 4 consecutive entries
          const int32_t i0 = pos + 4 * lane;
          if (i0 + 3 < deg) {
            const U32x4 q = *reinterpret_cast<const U32x4 *>(cs + i0);
            const U32x4 r4 = *reinterpret_cast<const U32x4 *>(cp + i0);
            const F32x4 w4 = *reinterpret_cast<const F32x4 *>(csw + i0);
            v[0] = q.a; v[1] = q.b; v[2] = q.c; v[3] = q.d;
            c[0] = r4.a; c[1] = r4.b; c[2] = r4.c; c[3] = r4.d;
            w[0] = w4.a; w[1] = w4.b; w[2] = w4.c; w[3] = w4.d;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const bool ok = i0 + j < deg;
              v[j] = ok ? cs[i0 + j] : HEMPTY; c[j] = ok ? cp[i0 + j] : 0u; w[j] = ok ? csw[i0 + j] : 0.0f;
            }
          }
        };
        auto stage_b = [&](int32_t pos) {                       // B[pos .. pos + HC) -> LDS, sorted, padded
#pragma unroll
          for (int u = 0; u < HC / 256; ++u) {
            const int32_t i0 = pos + 256 * u + 4 * lane;
            uint4 q;
            if (i0 + 3 < m) { const U32x4 t = *reinterpret_cast<const U32x4 *>(B + i0); q = make_uint4(t.a, t.b, t.c, t.d); }
            else q = make_uint4(i0 < m ? B[i0] : HEMPTY, i0 + 1 < m ? B[i0 + 1] : HEMPTY, i0 + 2 < m ? B[i0 + 2] : HEMPTY, HEMPTY);
            reinterpret_cast<uint4 *>(bch)[64 * u + lane] = q;
          }
        };
        stage_b(pb);
        load_a(pa, AI, AC, AW);
#pragma unroll
        for (int sgi = 0; sgi < PF; ++sgi) load_a(pa + 256 * (sgi + 1), NI[sgi], NC[sgi], NW[sgi]);
        __builtin_amdgcn_wave_barrier();
        uint32_t handled = 0; bool have_handled = false;     // candidates <= handled met every id of N(prev) they could equal
#ifdef SRW_PHASE_TIMING
        tm.n_w += 1; tm.n_w_elems += (unsigned long long)(deg - pa) + (m - pb);
#endif
        while (true) {
#ifdef SRW_PHASE_TIMING
          tm.n_w_windows += 1;
#endif
          const int32_t nb = (m - pb) < HC ? (m - pb) : HC;
          const uint32_t bmax = bch[nb - 1];                   // uniform LDS read
          // my four candidates against the staged chunk
          bool want[4]; uint32_t pos[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            want[j] = AI[j] <= bmax && AI[j] != xprev && !(have_handled && AI[j] <= handled);   // padding never <= bmax
            pos[j] = 0u;
          }
          SRW_U0(tm);
#pragma unroll
          for (int step = HC / 2; step >= 1; step >>= 1) {
            uint32_t probe[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) probe[j] = bch[pos[j] + step - 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) if (probe[j] < AI[j]) pos[j] += step;
          }
          SRW_U1(tm, t_pass1);
#if SRW_W_MATCH_LOOP
          // the matches of a lane's four candidates, one per pass of a wave-uniform loop: with ~10 % of the candidates matching, the
          // fullest lane holds two of them, so two passes of the correction code run instead of four exec-masked copies of it
          {
            uint32_t hit[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) hit[j] = bch[pos[j]];
            uint32_t mm = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) mm |= (want[j] && hit[j] == AI[j]) ? (1u << j) : 0u;
            while (__any(mm != 0u)) {
              if (mm) {
                const int j = __ffs((int)mm) - 1;
                mm &= mm - 1u;
                const uint32_t ac = j == 0 ? AC[0] : j == 1 ? AC[1] : j == 2 ? AC[2] : AC[3];
                const float aw = j == 0 ? AW[0] : j == 1 ? AW[1] : j == 2 ? AW[2] : AW[3];
                atomicAdd(&bins[ac >> csh], (double)aw - (double)div_exact(aw, q_));
                if (mbits) atomicOr(&mbits[ac >> 5], 1u << (ac & 31));
              }
            }
          }
#else
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (want[j] && bch[pos[j]] == AI[j]) {
              atomicAdd(&bins[AC[j] >> csh], (double)AW[j] - (double)div_exact(AW[j], q_));
              if (mbits) atomicOr(&mbits[AC[j] >> 5], 1u << (AC[j] & 31));
            }
#endif
          // advance the list that ends first
          const int32_t na = (deg - pa) < 256 ? (deg - pa) : 256;
          const int jl = (na - 1) & 3;
          const uint32_t alast = jl == 0 ? AI[0] : jl == 1 ? AI[1] : jl == 2 ? AI[2] : AI[3];
          const uint32_t amax = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl((int)alast, (na - 1) >> 2));
          if (amax <= bmax) {
            pa += 256;
            if (pa >= deg) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) { AI[j] = NI[0][j]; AC[j] = NC[0][j]; AW[j] = NW[0][j]; }
#pragma unroll
            for (int sgi = 0; sgi + 1 < PF; ++sgi) {
#pragma unroll
              for (int j = 0; j < 4; ++j) { NI[sgi][j] = NI[sgi + 1][j]; NC[sgi][j] = NC[sgi + 1][j]; NW[sgi][j] = NW[sgi + 1][j]; }
            }
            load_a(pa + 256 * PF, NI[PF - 1], NC[PF - 1], NW[PF - 1]);
          } else {
            handled = bmax; have_handled = true;
            pb += HC;
            if (pb >= m || bmax >= hi_id) break;
            __builtin_amdgcn_wave_barrier();
            stage_b(pb);
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
#ifdef SRW_PHASE_TIMING
  if (strat == 1) SRW_T1(tm, t_p1); else if (strat == 2) SRW_T1(tm, t_p2); else SRW_T1(tm, t_w);
  SRW_T0(tm);
#endif
  // inclusive prefix over the bins, in place (exact additions)
  if (GB) {
    __threadfence();                                  // every atomic of this wave has been performed
    double carry = 0.0;
    for (int32_t base = 0; base < n_bins; base += 64) {
      const int32_t j = base + lane;
      const double v = j < n_bins ? __hip_atomic_load(bins + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
      const double incl = carry + wave_incl_scan_f64(v);
      if (j < n_bins) __hip_atomic_store(bins + j, incl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      carry = readlane_f64(incl, 63);
    }
    __threadfence();
  } else {
    constexpr int PER = BIN_CAP / 64;
    double loc[PER];
    double run = 0.0;
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int j = lane * PER + i; run += (j < n_bins) ? bins[j] : 0.0; loc[i] = run; }
    const double incl = wave_incl_scan_f64(run);
    double excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 0.0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < PER; ++i) { const int j = lane * PER + i; if (j < n_bins) bins[j] = excl + loc[i]; }
    __builtin_amdgcn_wave_barrier();
  }
  strat_used = (unsigned)strat;
}

// Given the exact inclusive chunk prefixes of the corrections: 64-ary search over the chunk ends for the ONE chunk
// holding the first not-certain-miss index, then that chunk candidate by candidate.  -1: not applicable.
//   ABS = false: `bins` is the wave's LDS right after binned_fill, bins[j] = corrections up to the end of chunk j
//   ABS = true : `bins` is the (prev -> curr) edge's precomputed table in HBM (edge_tables.hip), which stores the
//                complete numerator A'_end(j) = PQ[end_j] + corrections — the search then touches nothing but the
//                table's 4 lines (64 scattered PQ reads otherwise), and inside the chunk
//                A'_k = A'_end(jc-1) + (PQ[k] - PQ[k0-1]) + corrections of [k0, k]  (all exact under the certificate).
// The table steps are bound by the number of memory requests they issue, so the membership probes of the chunk's
// candidates — one random sector each — are avoided whenever the data allows it: a chunk whose corrections sum to
// exactly zero holds neither a member nor a return edge (q > 1: all corrections are >= 0 ... the general case checks
// |sum| only when all corrections of the pair have one sign, see `one_sign`), and a short N(prev) (<= 1024 ids) is
// staged once in LDS and searched there.  Loads are batched: (1) S + the first search round, (2) 256 candidates'
// entries and prefix sums (4 per lane), (3) their probes; the chosen candidate's id comes from the registers of the
// lane that held it (id_out).
// candidates evaluated per lane and round of the located chunk (4: one round per 256-candidate chunk; 2: the second half
// of a 256-candidate chunk is only read when the answer is not in the first — the kernel is bound by memory requests as
// much as by latency, profiles/r02c_general_kernel_c3.md)
#ifndef SRW_RESOLVE_PER_LANE
// round 2 (one monolithic kernel, 4 waves/SIMD), config 3: 2 -> 182 M steps/s, 4 -> 169 M, 1 -> 168 M.  Round 3 (lean table kernel at 6-7
// waves/SIMD, chunks of 64 where they fit): 1 -> 7.05e8 against 6.72e8 at config 3, 3.61e8 against 3.46e8 at config 5's stand-in (s57, s58)
#define SRW_RESOLVE_PER_LANE 1
#endif
template <bool ABS, bool BF = false, bool CHAIN = true>
__device__ inline int32_t binned_resolve(const GraphView &g, const Row &rc, const Bias &b, const double *bins,
                                         const BinGeom geo, float r, unsigned &fallback, unsigned &served, Member &tm,
                                         int32_t &id_out, uint32_t *stage, double *S_out = nullptr,
                                         const unsigned long long *cmask = nullptr /* the pair's chunk masks (edge_tables.hip): bit k = candidate k is in N(prev) */) {
  constexpr int PL = SRW_RESOLVE_PER_LANE;
  const int lane = lane_id();
  // (one wave, one row, one pair: everything below that does not depend on the lane is scalar work — wave_primitives.h:uni)
  const int32_t deg = uni(rc.deg);
  const int csh = uni(geo.csh);
  const int32_t n_bins = uni(geo.n_bins);
  const uint32_t rflags = uni(rc.flags);
  const int64_t roff = uni(rc.off);
  const Ent *row = g.ent + roff;
  const uint32_t *B = uni(b.prev_sids);
  const int32_t m = uni(b.prev_deg);
  const uint32_t xprev = uni((uint32_t)((int64_t)b.prev - b.vmin));
  const float p_ = b.p, q_ = b.q;
  const uint32_t prev_hub = uni(b.prev_hub);
  const uint32_t *hubbits = (prev_hub && g.hub_bm) ? g.hub_bm + (int64_t)(prev_hub - 1) * g.hub_words : nullptr;
  const PqRow PQ(g, roff);
  auto chunk_end = [&](int32_t j) { const int64_t e = (((int64_t)j + 1) << csh) - 1; return (int32_t)(e < deg ? e : deg - 1); };
  // tables of rows whose every sum is binary32-exact are stored as floats (edge_tables.hip); LDS bins are always f64
  const bool f32t = ABS && g.ebp.f32 && (rflags & ROW_PQ_F32);
  // A short N(prev) is staged in LDS and searched there by the located chunk's candidates.  On a table step the copy is started
  // NOW, with direct-to-LDS loads (global_load_lds: no registers, no wait here), so that it travels together with the table's
  // first block instead of costing a dependent round trip once the chunk is known.  (Speculative: a chunk that turns out to hold
  // no special does not need it.)
  bool pre_staged = false;
#ifdef SRW_PRESTAGE            // measured and not kept (profiles/r04_table_kernel.md): the speculative copy costs requests when the chunk holds no special, -5 %
  if (ABS && !cmask && stage && !hubbits && m > 0 && m <= 1024) {
    int P2 = 1; while (P2 < m) P2 <<= 1;
    for (int32_t t0 = 0; t0 < m; t0 += 64)
      if (t0 + lane < m)
        __builtin_amdgcn_global_load_lds(B + t0 + lane, (__attribute__((address_space(3))) void *)(stage + t0), 4, 0, 0);
    for (int32_t t = m + lane; t < P2; t += 64) stage[t] = 0xFFFFFFFFu;
    pre_staged = true;
  }
#endif
  const double p = (double)r;
  // Certified compares without a divide.  The reference's acc_k = sum of fl(w'_i / S) differs from num / S (num exact)
  // by at most (k + 2) u num / S.  With t = (k + 8) 2^-51 = 4 (k + 8) u:
  //   fl(num (1 + t)) <  fl(p S)  =>  num (1 + t)(1 - 3u) < p S  =>  (num / S)(1 + (k + 2) u) < p   : a CERTAIN miss
  //   fl(num (1 - t)) >= fl(p S)  =>  num (1 - t)(1 + 3u) >= p S =>  (num / S)(1 - (k + 2) u) >= p  : a CERTAIN hit
  // (1 +- t is exact in f64; each product rounds once).  NaN compares false both ways -> "not a certain miss, not a
  // certain hit" -> the sequential chain, as before.
  double S = 0.0, pS = 0.0;
  auto not_miss = [&](int32_t k, double num) { return !(num * (1.0 + (double)(k + 8) * 0x1p-51) < pS); };
  auto sure_hit = [&](int32_t k, double num) { return num * (1.0 - (double)(k + 8) * 0x1p-51) >= pS; };
  int32_t jc;
  double b_prev = 0.0, b_this = 0.0;
  if constexpr (ABS) {
    // the pair's table in HBM: a 64-ary tree over the chunk prefixes (eb_layout), one block of <= 64 values per level;
    // S is the last element of the top level
    const bool u16t = eb_pair_u16(rflags, csh, g.ebp);
    const double unit = u16t ? eb_row_unit(rflags) : 0.0;
#ifndef SRW_LAYOUT_ON_SALU         // (the layout on the vector unit — the offsets only feed per-lane addresses: 577 against 591 ms at config 3, profiles/r04_table_kernel_ab_runs.txt)
    EbLayout lay = eb_layout(f32t, on_vector(n_bins), false, 0, u16t);
    lay.n1 = uni(lay.n1); lay.n2 = uni(lay.n2);
#else
    const EbLayout lay = eb_layout(f32t, n_bins, false, 0, u16t);
#endif
    const int nlev = lay.n2 ? 3 : lay.n1 ? 2 : 1;
    int32_t blk = 0;
    double prev_val = 0.0;                            // prefix just before the block being searched
    jc = 0;
    // (three copies with L a constant: what selects the level's offset / length / shift by L disappears — scalar instructions, the busier
    //  unit: 591 against 604 ms at config 3, profiles/r04_table_kernel_ab_runs.txt)
#ifdef SRW_EB_ROLLED_LEVELS
    for (int L = nlev - 1; L >= 0; --L) {
#else
#pragma unroll
    for (int L = 2; L >= 0; --L) {
      if (L >= nlev) continue;
#endif
      const uint32_t off = L == 2 ? lay.l2_off : L == 1 ? lay.l1_off : lay.l0_off;
      const int32_t cnt = L == 2 ? lay.n2 : L == 1 ? lay.n1 : n_bins;
      const int32_t i = blk * 64 + lane;
      const bool in_r = i < cnt;
      double v = 0.0;
      if (L == 0 && u16t) {                           // the chunks' own masses, 16 bits each: prefix = what precedes the block + a wave scan
        const double d = in_r ? (double)reinterpret_cast<const unsigned short *>(bins + (size_t)off * 8)[i] * unit : 0.0;
        v = prev_val + wave_incl_scan_f64(d);         // (exact: every partial sum is an exact multiple of 2^G under the row certificate)
      } else if (in_r) v = f32t ? (double)reinterpret_cast<const float *>(bins + (size_t)off * 8)[i] : bins[(size_t)off * 8 + i];
      if (L == nlev - 1) {
        S = readlane_f64(v, cnt - 1);               // (the top level has at most 64 elements)
        if (!(S > 0.0)) return -1;
        if (!CHAIN && S_out && g.dbg_chain_deg && deg >= g.dbg_chain_deg) { *S_out = S; return CHAIN_NEEDED; }   // tests: the chain kernels on every long row
        pS = p * S;
      }
      const int64_t je = (((int64_t)i + 1) << (6 * L)) - 1;      // the chunk this element is the prefix of
      const int32_t j = (int32_t)(je < n_bins ? je : n_bins - 1);
      const unsigned long long mm = __ballot(in_r && not_miss(chunk_end(j), v));
      if (!mm) { id_out = row[0].id; return 0; }      // even the last candidate is a certain miss -> edges.head
      const int f = __ffsll((long long)mm) - 1;
      if (f) prev_val = readlane_f64(v, f - 1);
      if (L == 0) { jc = blk * 64 + f; b_this = readlane_f64(v, f); b_prev = prev_val; }
      else blk = blk * 64 + f;
    }
  } else {
    // the wave's LDS bins right after binned_fill: round trip 1 = everything the search needs first
    int32_t lo = 0, hi = n_bins - 1;                 // the chunk of the first not-certain-miss index lies in [lo, hi]
    int32_t j1 = hi >= 64 ? (int32_t)(((int64_t)hi * (lane + 1)) >> 6) : (lane <= hi ? lane : hi);   // lane 63 probes hi
    const double b_last = bins[n_bins - 1];
    double b_j = bins[j1];
    const double pq_last = PQ[deg - 1];
    double pq_j = PQ[chunk_end(j1)];
    S = pq_last + b_last;
    if (!(S > 0.0)) return -1;
    if (!CHAIN && S_out && g.dbg_chain_deg && deg >= g.dbg_chain_deg) { *S_out = S; return CHAIN_NEEDED; }   // tests: the chain kernels on every long row
    pS = p * S;
    while (hi - lo >= 64) {
      const unsigned long long mm = __ballot(not_miss(chunk_end(j1), pq_j + b_j));
      if (!mm) { id_out = row[0].id; return 0; }      // even the last candidate is a certain miss -> edges.head
      const int f = __ffsll((long long)mm) - 1;
      const int32_t jf = __builtin_amdgcn_readlane(j1, f);
      const int32_t jprev = f ? __builtin_amdgcn_readlane(j1, f - 1) : lo - 1;
      hi = jf; lo = jprev + 1;
      const int64_t span = (int64_t)hi - lo;
      j1 = span >= 64 ? lo + (int32_t)((span * (lane + 1)) >> 6) : (lo + lane <= hi ? lo + lane : hi);
      b_j = bins[j1];
      pq_j = PQ[chunk_end(j1)];
    }
    {
      const bool nm = lo + lane <= hi && not_miss(chunk_end(j1), pq_j + b_j);
      const unsigned long long mm = __ballot(nm);
      if (!mm) { id_out = row[0].id; return 0; }
      jc = lo + (__ffsll((long long)mm) - 1);
    }
    b_prev = jc ? bins[jc - 1] : 0.0; b_this = bins[jc];
  }
  // candidate-by-candidate evaluation of chunk jc, 256 candidates per round
  const int32_t k0 = (int32_t)((int64_t)jc << csh), k1 = chunk_end(jc);
  // The entries (and mask words) of a round are requested one round AHEAD: the first round's together with the reads of PQ below,
  // round r + 1's before round r's probes are waited for — a located chunk of several rounds (hub rows: deg / cap candidates) costs
  // one dependent round trip per round instead of two.
  Ent e_nx[PL]; unsigned long long mw_nx[PL];
  auto fetch_round = [&](int32_t base) {
#pragma unroll
    for (int u = 0; u < PL; ++u) {
      const int32_t k = base + u * 64 + lane;
      e_nx[u].id = b.prev; e_nx[u].w = 0.0f; mw_nx[u] = 0ull;
      if (k <= k1) e_nx[u] = load_ent(g, row, k);
      if (cmask && base + u * 64 <= k1) mw_nx[u] = cmask[(base >> 6) + u];
    }
  };
  fetch_round(k0);
  // exact value of the numerator just before the chunk, and the exact sum of the chunk's corrections
  // (inside the chunk the base prefix sums PQ[k] - PQ[k0-1] are not loaded: under the row certificate every partial sum of
  // the base weights fl(w / q) is exact in any order, so the wave scan that adds up the corrections adds them up as well —
  // 8 bytes per candidate fewer to request)
  double carry, chunk_corr = 1.0;
  if (ABS) {
    carry = b_prev;                                          // A'_{k0-1}
    // (with the pair's chunk masks nothing asks whether the chunk holds a special: two reads of PQ fewer)
    if (!cmask) chunk_corr = (b_this - b_prev) - (PQ[k1] - (k0 ? PQ[k0 - 1] : 0.0));
  } else {
    carry = b_prev + (k0 ? PQ[k0 - 1] : 0.0);                // corrections before the chunk + base weights before the chunk
    chunk_corr = b_this - b_prev;
  }
  // q > 1 and p <= q: every correction (w - w/q for a member, w/p - w/q for a return edge) is >= 0; q < 1 and p >= q:
  // every one is <= 0.  Then "the chunk's corrections sum to exactly 0" means "no special in the chunk".
  const BiasDiv bdiv(p_, q_);
  const bool one_sign = (q_ > 1.0f && p_ <= q_) || (q_ < 1.0f && p_ >= q_);
  const bool no_specials = one_sign && !cmask && chunk_corr == 0.0;
  // a short N(prev): staged in LDS once (sorted, padded to a power of two), searched there
  int stage_levels = 0;
  if (!no_specials && !cmask && stage && !hubbits && m > 0 && m <= 1024) {
    stage_levels = m > 1 ? 32 - __builtin_clz((unsigned)(m - 1)) : 0;       // ceil(log2 m): the padded length is a power of two
    const int P2 = 1 << stage_levels;
    if (pre_staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the copy started before the table search has landed
    else for (int32_t t = lane; t < P2; t += 64) stage[t] = t < m ? B[t] : 0xFFFFFFFFu;
    if (stage_levels == 0) stage_levels = -1;               // m == 1: one compare, no search level
    __builtin_amdgcn_wave_barrier();
  }
  // a long N(prev) without a bitmap: its neighbor-set filter (device_common.h:bf_hash), if the graph has them
  const uint32_t *bf = nullptr; uint32_t bf_nw = 0;
  if (BF && !no_specials && !cmask && !stage_levels && !hubbits && g.bf_off && m >= BF_MIN_DEG) {
    const uint32_t bo = g.bf_off[xprev];
    if (bo != BF_NONE) { bf = g.bf_bits + bo; bf_nw = bf_words(m); }
  }
  served = 1;
  // membership for the exact chain, should the draw sit on a CDF boundary: the same tests, candidate by candidate
  Member cm; cm.mode = 1; cm.bm = nullptr; cm.seg_base = 0; cm.hub = hubbits; cm.ehash = g.ehash; cm.ehash_mask = g.ehash_mask;
  if (CHAIN && !hubbits && !g.ehash && g.bf_off && m >= BF_MIN_DEG) {
    const uint32_t bo = g.bf_off[xprev];
    if (bo != BF_NONE) { cm.bf = g.bf_bits + bo; cm.bf_nw = bf_words(m); }
  }
#ifdef SRW_PHASE_TIMING
  tm.n_binned += 1;                                             // resolved chunks ...
  tm.n_w_elems += (unsigned long long)(k1 - k0 + 1);             // ... their candidates
  if (no_specials) tm.n_w += 1; else if (stage_levels) tm.n_p1 += 1; else if (hubbits) tm.n_w_windows += 1; else tm.n_p1_elems += 1;
#endif
  for (int32_t base = k0; base <= k1; base += 64 * PL) {
#ifdef SRW_PHASE_TIMING
    tm.t_fin += 1;                                              // rounds of 64 * PL candidates
#endif
#ifndef SRW_PREFETCH_ROUNDS
    if (base > k0) fetch_round(base);
#endif
    Ent e[PL]; bool valid[PL], in[PL], want[PL]; uint32_t xs[PL]; unsigned long long mwc[PL];
    tm.res_bytes += 8ull * (unsigned long long)((k1 - base + 1) < 64 * PL ? (k1 - base + 1) : 64 * PL);
#pragma unroll
    for (int u = 0; u < PL; ++u) {
      const int32_t k = base + u * 64 + lane;
      valid[u] = k <= k1;
      e[u] = e_nx[u]; mwc[u] = mw_nx[u];
    }
#ifdef SRW_PREFETCH_ROUNDS      // measured and not kept (profiles/r04_table_kernel.md): a round requested ahead is wasted when the answer is in the current one, -9 %
    if (base + 64 * PL <= k1) fetch_round(base + 64 * PL);
#endif
#pragma unroll
    for (int u = 0; u < PL; ++u) {
      xs[u] = (uint32_t)((int64_t)e[u].id - b.vmin); in[u] = false;
      want[u] = !no_specials && valid[u] && e[u].id != b.prev;
    }
    if (no_specials) {
    } else if (cmask) {                               // the membership of these 64 candidates was precomputed with the table: no probe
#pragma unroll
      for (int u = 0; u < PL; ++u) {
        in[u] = want[u] && ((mwc[u] >> lane) & 1ull);
      }
    } else if (stage_levels) {
      uint32_t pos[PL];
#pragma unroll
      for (int u = 0; u < PL; ++u) pos[u] = 0u;
      for (int st = stage_levels > 0 ? (1 << (stage_levels - 1)) : 0; st >= 1; st >>= 1) {
        uint32_t probe[PL];
#pragma unroll
        for (int u = 0; u < PL; ++u) probe[u] = stage[pos[u] + st - 1];
#pragma unroll
        for (int u = 0; u < PL; ++u) if (probe[u] < xs[u]) pos[u] += st;
      }
#pragma unroll
      for (int u = 0; u < PL; ++u) in[u] = want[u] && stage[pos[u]] == xs[u];
    } else if (hubbits) {
      uint32_t wd[PL];
#pragma unroll
      for (int u = 0; u < PL; ++u) wd[u] = want[u] ? hubbits[xs[u] >> 5] : 0u;
#pragma unroll
      for (int u = 0; u < PL; ++u) in[u] = (wd[u] >> (xs[u] & 31)) & 1u;
    } else if (BF && bf) {                            // long N(prev), no bitmap: the row's filter first, the exact test on a positive
#pragma unroll
      for (int u = 0; u < PL; ++u) {
        in[u] = false;
        if (want[u]) {
          uint32_t word, mask;
          bf_hash(xs[u], bf_nw, word, mask);
          if ((bf[word] & mask) == mask)
            in[u] = g.ehash ? edge_exists(g.ehash, g.ehash_mask, xprev, xs[u]) : sorted_contains(B, m, xs[u]);
        }
      }
    } else if (g.ehash) {
      edge_exists_n<PL>(g.ehash, g.ehash_mask, xprev, xs, want, in);
    } else {
#pragma unroll
      for (int u = 0; u < PL; ++u) in[u] = want[u] && sorted_contains(B, m, xs[u]);
    }
#pragma unroll
    for (int u = 0; u < PL; ++u) {
      const int32_t k = base + u * 64 + lane;
      if (base + u * 64 > k1) break;                 // wave-uniform
      // the candidate's variant w' (= base weight fl(w / q) + the correction the tables hold, exact under the row certificate)
      const double term = valid[u] ? (double)bdiv(e[u].w, !no_specials && e[u].id == b.prev, in[u]) : 0.0;
      const double incl = wave_incl_scan_f64(term);
      const double num = carry + incl;
      const bool nm = valid[u] && not_miss(k, num);
      const bool hit = sure_hit(k, num);
      const unsigned long long mm = __ballot(nm);
      if (mm) {
        const int f = __ffsll((long long)mm) - 1;
        if (__builtin_amdgcn_readlane((int)hit, f)) { id_out = __builtin_amdgcn_readlane(e[u].id, f); return base + u * 64 + f; }
        if (!CHAIN) { if (S_out) *S_out = S; return CHAIN_NEEDED; }
        fallback = 1;
        const int32_t kk = wave_chain_pick(row, deg, b, r, S, &cm);
        id_out = row[kk].id;
        return kk;
      }
      carry += readlane_f64(incl, 63);
    }
  }
  if (!CHAIN) { if (S_out) *S_out = S; return CHAIN_NEEDED; }
  fallback = 1;
  const int32_t kk = wave_chain_pick(row, deg, b, r, S, &cm);
  id_out = row[kk].id;
  return kk;
}

// ---- first step of a walk (initFirstStep, RandomWalk.scala:51-66): RandomSample.sample on the RAW row -------------------
// Certified like the masked path: exact parallel sum under the certificate, exact prefix sums, divide-free compares;
// the sequential chain otherwise.  No membership, no LDS.
template <bool CHAIN = true>
__device__ inline int32_t wave_pick_first(const GraphView &g, const Row &rc, float r, unsigned &fallback, int32_t &id_out) {
  const int lane = lane_id();
  const Ent *row = g.ent + rc.off;
  const int32_t deg = rc.deg;
  Bias nb; nb.second_order = false; nb.need_member = false; nb.p = nb.q = 1.0f; nb.prev = 0; nb.prev_sids = nullptr; nb.prev_deg = 0; nb.vmin = g.vmin;
  double part = 0.0;
  SumCert cert;
  bool neg = false;
  if (g.ids32) {                                     // unit weights: S = deg, nothing to read
    if (lane == 0) part = (double)deg;
    cert.add(1.0f);
  } else
  for (int32_t base = 0; base < deg; base += 256) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int32_t k = base + u * 64 + lane;
      if (k < deg) { const float w = row[k].w; part += (double)w; cert.add(w); neg |= !(w >= 0.0f); }
    }
  }
  const int emin = wave_min_i32(cert.emin), emax = wave_max_i32(cert.emax);
  const bool bad = __any(cert.bad) || __any(neg);
  const double S = wave_total_f64(part);                   // (used only under the certificate below: exact in any order)
  if (bad || !sum_is_exact(emin, emax, false, deg) || !(S > 0.0)) {
    if (!CHAIN) return CHAIN_NEEDED;
    unsigned f = 0;
    const double Sc = wave_sum_exact_or_chain(row, deg, nb, f);
    fallback = 1;
    const int32_t kk = wave_chain_pick(row, deg, nb, r, Sc);
    id_out = row[kk].id;
    return kk;
  }
  const double pS = (double)r * S;
  double carry = 0.0;
  for (int32_t base = 0; base < deg; base += 64) {
    const int32_t k = base + lane;
    const bool valid = k < deg;
    Ent e; e.id = 0; e.w = 0.0f;
    if (valid) e = load_ent(g, row, k);
    const double incl = wave_incl_scan_f64((double)e.w);
    const double num = carry + incl;
    const double t = (double)(k + 8) * 0x1p-51;
    const bool nm = valid && !(num * (1.0 + t) < pS);
    const bool hit = num * (1.0 - t) >= pS;
    const unsigned long long mm = __ballot(nm);
    if (mm) {
      const int f = __ffsll((long long)mm) - 1;
      if (__builtin_amdgcn_readlane((int)hit, f)) { id_out = __builtin_amdgcn_readlane(e.id, f); return base + f; }
      if (!CHAIN) return CHAIN_NEEDED;
      fallback = 1;
      const int32_t kk = wave_chain_pick(row, deg, nb, r, S);
      id_out = row[kk].id;
      return kk;
    }
    carry += readlane_f64(incl, 63);
  }
  id_out = row[0].id;     // edges.head (:24)
  return 0;
}

// ---- rows of fewer than 256 candidates with a precomputed membership MASK of the (prev -> curr) pair -------------------
// (edge_tables.hip: bit k = "candidate k of N(curr) is in N(prev)".)  The whole row sits in registers (4 candidates per
// lane), so the step is: one round trip for the mask + the row, then RandomSample.sample's certified evaluation — the
// same certified evaluation as the table path (exact parallel S and prefix sums under the certificate, divide-free
// certain-miss / certain-hit compares, the sequential chain otherwise) without a single membership lookup.
constexpr int MASK_MAX_DEG = 256;       // rows up to 255 candidates: 4 per lane stay in registers
// NS: slots of 64 candidates the instantiation handles (rows up to 64 * NS candidates); NS = 1 is the straight-line form for the short rows
template <bool CHAIN = true, int NS = 4>
__device__ inline int32_t wave_pick_masked(const GraphView &g, const Row &rc, const Bias &b, uint32_t inline_mask,
                                           const uint32_t *words, float r, unsigned &fallback, int32_t &id_out) {
  const int lane = lane_id();
  const Ent *row = g.ent + rc.off;
  const int32_t deg = rc.deg;
  const int ni = NS == 1 ? 1 : (deg + 63) >> 6;     // <= NS
  float wv[NS]; int32_t idv[NS]; uint32_t mw[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    mw[i] = 0u;
    if (i < ni) mw[i] = words ? words[2 * i + (lane >> 5)] : ((i == 0 && lane < 32) ? inline_mask : 0u);
  }
  double part = 0.0;
  SumCert cert;
  bool neg = false;
  const BiasDiv bdiv(b.p, b.q);
  // A row that holds the per-call certificate (ROW_PQ_OK, sampler_tables.hip:k_pq_*: every variant w, fl(w/p), fl(w/q) of every candidate
  // finite and >= 0, every sum of variants exact in any order) needs no certificate of THIS step's weights: they are variants of that row.
  // (Round 6: ~60 of the mask step's ~250 vector instructions — the per-candidate exponent bookkeeping and two wave reductions.)
  const bool certified = uni((uint32_t)(rc.flags & ROW_PQ_OK)) != 0u;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int32_t k = i * 64 + lane;
    wv[i] = 0.0f; idv[i] = 0;
    if (i < ni && k < deg) {
      const Ent e = load_ent(g, row, k);
      const float w = bdiv(e.w, e.id == b.prev, ((mw[i] >> (lane & 31)) & 1u) != 0u);
      wv[i] = w; idv[i] = e.id;
      part += (double)w;
      if (!certified) { cert.add(w); neg |= !(w >= 0.0f); }
    }
  }
  bool uncertain = false;
  if (!certified) {
    const int emin = wave_min_i32(cert.emin), emax = wave_max_i32(cert.emax);
    uncertain = __any(cert.bad) || __any(neg) || !sum_is_exact(emin, emax, false, deg);
  }
  const double S_par = wave_total_f64(part);               // (used only under a certificate: exact in any order)
  if (uncertain || !(S_par > 0.0)) {     // (S = 0: the reference divides by zero -> chain)
    if (!CHAIN) return CHAIN_NEEDED;
    unsigned f = 0;
    const double Sc = wave_sum_exact_or_chain(row, deg, b, f);
    fallback = 1;
    const int32_t kk = wave_chain_pick(row, deg, b, r, Sc);
    id_out = row[kk].id;
    return kk;
  }
  // Under the certificate every partial sum of the w' is exact in any order, so A'_k = sum_{i <= k} w'_i is exact and
  // S = A'_{deg-1} is the reference's sum bit for bit; the reference's acc_k = sum of fl(w'_i / S) differs from A'_k / S
  // by at most (k + 2) u A'_k / S: the divide-free certified compares of binned_resolve apply.
  const double S = S_par;
  const double p = (double)r;
  const double pS = p * S;
  double carry = 0.0;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    if (i >= ni) break;
    const int32_t k = i * 64 + lane;
    const bool valid = k < deg;
    const double incl = wave_incl_scan_f64(valid ? (double)wv[i] : 0.0);
    const double num = carry + incl;
    const double t = (double)(k + 8) * 0x1p-51;
    const bool nm = valid && !(num * (1.0 + t) < pS);
    const bool hit = num * (1.0 - t) >= pS;
    const unsigned long long mm = __ballot(nm);
    if (mm) {
      const int f = __ffsll((long long)mm) - 1;
      if (__builtin_amdgcn_readlane((int)hit, f)) { id_out = __builtin_amdgcn_readlane(idv[i], f); return i * 64 + f; }
      if (!CHAIN) return CHAIN_NEEDED;
      fallback = 1;                                 // within rounding distance of a boundary: exact chain
      const int32_t kk = wave_chain_pick(row, deg, b, r, S);
      id_out = row[kk].id;
      return kk;
    }
    carry += readlane_f64(incl, 63);
  }
  id_out = __builtin_amdgcn_readlane(idv[0], 0);   // edges.head (:24)
  return 0;
}

// force_small: no minimum degree.  strat_used: 1 = P1, 2 = P2, 3 = W, 4 = P3 (0 when the search does not apply).
__device__ inline int32_t wave_pick_binned(const GraphView &g, const Row &rc, int64_t curr_slot, const Bias &b,
                                           uint32_t *lds, float r, unsigned &fallback, unsigned &served, int tune,
                                           bool force_small, Member &tm, unsigned long long &alg_bytes, unsigned &strat_used,
                                           int32_t &id_out) {
  strat_used = 0;
  if (!pq_ready(g) || !b.second_order || !b.need_member) return -1;
  const int32_t deg = rc.deg;
  if ((!force_small && deg < 128) || !(rc.flags & ROW_PQ_OK)) return -1;
  const BinGeom geo = bin_geometry(deg, 6, BIN_CAP);
  binned_fill(g, rc, b, lds, tune, geo, tm, alg_bytes, strat_used);
  return binned_resolve<false>(g, rc, b, reinterpret_cast<const double *>(lds), geo, r, fallback, served, tm, id_out, lds + 2 * BIN_CAP);
}

// Per-edge bias table (edge_tables.hip): the chunk prefixes of this (prev -> curr) pair were computed once per (p, q)
// by k_eb_build with the same binned_fill; the step is the search + one chunk, no intersection of the two rows.
#ifndef SRW_EB_BINS
#define SRW_EB_BINS 64
#endif
constexpr int EB_BINS = SRW_EB_BINS;           // default chunks per table: one lane each in the search.  GraphView::eb_cap (32 / 64 /
                                               // 128 / 256) is what the standing tables were built with: more chunks = a second search round, shorter chunks
template <bool BF = false, bool CHAIN = true>
__device__ inline int32_t wave_pick_edge_table(const GraphView &g, const Row &rc, const Bias &b, const double *table,
                                               float r, unsigned &fallback, unsigned &served, Member &tm, int32_t &id_out,
                                               uint32_t *stage /* 1024 words of the wave's LDS */, double *S_out = nullptr) {
  const int32_t dv = uni(rc.deg);
  const uint32_t rflags = uni(rc.flags);
  table = uni(table);
#ifndef SRW_GEOM_ON_SALU          // (the pair's geometry on the vector unit: 604-607 against 610-614 ms at config 3, profiles/r04_table_kernel_ab_runs.txt; -DSRW_GEOM_ON_SALU: as before)
  const PairGeom pgv = eb_pair_geometry(on_vector(dv), on_vector(uni(b.prev_deg)), g.ebp);
  PairGeom pg; pg.csh = uni(pgv.csh); pg.n_bins = uni(pgv.n_bins); pg.cmask = uni((int32_t)pgv.cmask) != 0;
#else
  const PairGeom pg = eb_pair_geometry(dv, uni(b.prev_deg), g.ebp);
#endif
  BinGeom geo; geo.csh = pg.csh; geo.n_bins = pg.n_bins;
  const unsigned long long *cmask = nullptr;
  if (pg.cmask) cmask = reinterpret_cast<const unsigned long long *>(table + (size_t)eb_layout(g.ebp.f32 && (rflags & ROW_PQ_F32), pg.n_bins, true, dv,
                                                                                              eb_pair_u16(rflags, pg.csh, g.ebp)).cm_off * 8);
  return binned_resolve<true, BF, CHAIN>(g, rc, b, table, geo, r, fallback, served, tm, id_out, stage, S_out, cmask);
}

}  // namespace srw
