// edge_tables.hip — per-edge bias tables for the second-order walk (gfx950).
//
// The reference recomputes computeSecondOrderWeights for every step (M/algorithm/RandomSample.scala:27-44): for the
// step prev -> curr -> ? that is an intersection N(prev) ∩ N(curr), O(min(deg) log) at best, and on a power-law graph
// the walk sits on hub -> hub pairs most of the time (≈40 K row elements per step at config 3).  The walk visits each
// directed edge about numWalks * 1.3 times, so the intersection of a pair is worth computing ONCE per (p, q):
//
//   table(prev -> curr)[j] = A'_end(j) = sum over positions k < (j + 1) << csh of N(curr) of w'_k               j < 64
//                          = PQ[end_j] + sum of (w'_k - fl(w_k / q))   (PQ: the row's exact prefix sums of fl(w / q))
//
// — the exact chunk prefixes of the corrections the binned search of sampling.h accumulates in LDS (same binned_fill,
// same certificate: every such sum is exact in any order), at most 64 chunks per pair (one lane each in the search),
// 512 B per pair (256 B where every prefix is exactly representable in binary32: ROW_PQ_F32).  A step over a pair with a table is then: 64 chunk ends compared in one wave instruction + ONE chunk
// evaluated candidate by candidate (sampling.h:binned_resolve) — the same arithmetic, hence the same bits, as the
// on-the-fly search.  Pairs are prioritised by the cost model of the on-the-fly strategies (binned_cost) and take what
// HBM is left after every other structure; everything else keeps the on-the-fly path.
//
// Rows of fewer than 256 candidates get a second, smaller kind of table instead: the membership MASK of the pair (bit k =
// "candidate k of N(curr) is in N(prev)", 4 .. 32 bytes; rows of at most 32 candidates keep it inline in the per-entry
// offset word).  The step then needs no membership lookup at all (sampling.h:wave_pick_masked) — q-independent, but
// rebuilt with the rest for simplicity.
//
// This is north_star's "per-edge p/q-biased tables built by a CDNA4 kernel that stages neighbor lists in LDS" in the
// only form that stays bit-identical to the reference's CDF inversion.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "engine.h"
#include "sampling.h"

namespace srw {
namespace {

constexpr int TPB = 256;
constexpr int GRAB_SLOTS = 4;      // row slots per cursor grab of the selection passes
constexpr int GRAB_ITEMS = 16;     // pairs per cursor grab of the build kernel (a single counter word saturates at ~88 atomics/us)

struct EbSel {          // which pairs get a table
  int32_t mask_max;     // rows of curr up to this many candidates: membership mask (0: none)
  int32_t min_deg;      // bins tables: shortest row of curr
  int64_t min_cost;     // bins tables: smallest priority (saved wave-cycles per 64 B of table, see eb_units) that still fits the budget
  int32_t has_ehash, has_hub;
  EbPolicy pol;         // bins tables: chunk sizes, chunk masks, finer tables of the pairs with a long N(prev) (sampling.h:eb_pair_geometry)
};
constexpr int INLINE_MAX_DEG = 32;   // masks of rows up to 32 candidates live in eb_off[e] itself

// Table of pair (u -> v): kind 0 = none, 1 = bins (64-byte units), 2 = mask (16-byte units), 3 = inline mask.
// cost_out: the pair's priority under the HBM budget (bins tables).
__device__ inline uint32_t eb_units(const Row &ru, const Row &rv, const EbSel &s, int64_t &cost_out, int &kind) {
  cost_out = 0; kind = 0;
  if (ru.deg <= 0 || rv.deg <= 0) return 0u;
  if (rv.deg <= s.mask_max) {
    if (rv.deg <= INLINE_MAX_DEG) { kind = 3; return 0u; }
    kind = 2;
    return (uint32_t)((((rv.deg + 31) >> 5) + 3) >> 2);
  }
  if (rv.deg < s.min_deg || !(rv.flags & ROW_PQ_OK)) return 0u;
  const BinnedCost c = binned_cost(rv.deg, ru.deg, s.has_hub && (ru.flags >> ROW_HUB_SHIFT) != 0u, s.has_ehash != 0);
  const int64_t cost = c.c1 < c.c2 ? (c.c1 < c.cw ? c.c1 : c.cw) : (c.c2 < c.cw ? c.c2 : c.cw);
  const PairGeom geo = eb_pair_geometry(rv.deg, ru.deg, s.pol);
  const uint32_t units = eb_layout(s.pol.f32 && (rv.flags & ROW_PQ_F32), geo.n_bins, geo.cmask, rv.deg, eb_pair_u16(rv.flags, geo.csh, s.pol)).units;
  // priority = wave-cycles a table saves per visit, per 64 bytes of table: an on-the-fly step costs its intersection
  // work plus ~20 us of dependent round trips whatever its size (measured: 38 .. 43 us per P1 / W step at config 3
  // against 10 .. 25 us per table step), so short cheap tables are worth as much per byte as the hub <-> hub ones
  cost_out = (cost + 49152) / (int64_t)units;
  if (cost_out < s.min_cost) return 0u;
  kind = 1;
  return units;
}

// Priority classes for the budget fit: 4 per octave from 2^11 up (class 1 .. 62; 0 unused, 63 = the masks).
__host__ __device__ inline int prio_class(int64_t x) {
  if (x < 2048) return 1;
  int e = 0; while ((x >> (e + 1)) != 0) ++e;          // floor(log2 x) >= 11
  const int c = (e - 11) * 4 + (int)((x >> (e - 2)) & 3) + 1;
  return c > 62 ? 62 : c;
}
__host__ __device__ inline int64_t prio_class_floor(int c) {   // smallest priority in class c (c >= 2)
  const int e = (c - 1) / 4 + 11, sub = (c - 1) & 3;
  return (int64_t)(4 + sub) << (e - 2);
}

__device__ inline int64_t grab_u64(unsigned long long *cursor, unsigned long long n) {
  unsigned long long grab = 0;
  if (lane_id() == 0) grab = atomicAdd(cursor, n);
  return (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab));
}

// Which pairs (u -> x) a handle builds tables for, and where their table words go:
//   SH = false (whole-graph handle, srw_walk): every entry k of u's row in input order; word -> eb_off[ru.off + k]
//   SH = true  (vertex-sharded handle, srw_shard_*): every DISTINCT neighbor x of u that this shard owns — enumerated from
//              the replicated membership structure (mrows / msids: sorted neighbor ids of the whole graph; at world 1 they
//              alias rows / sids), because prev's row lives on owner(prev) while the table describes N(curr) and belongs to
//              owner(curr); word -> pair hash (device_common.h:PairSlot), k = position in u's SORTED row.
struct ShardSel { int32_t rank, world; };
template <bool SH>
__device__ inline Row eb_urow(const GraphView &g, int64_t u) { return SH ? g.mrows[u] : g.rows[u]; }
template <bool SH>
__device__ inline bool eb_pair(const GraphView &g, const Row &ru, int32_t k, const ShardSel &ss, uint32_t &xs) {
  if (!SH) { xs = (uint32_t)((int64_t)g.ent[ru.off + k].id - g.vmin); return true; }
  xs = g.msids[ru.off + k];
  if (k > 0 && g.msids[ru.off + k - 1] == xs) return false;                     // multi-edge: one table per pair
  return ss.world == 1 || owner_of_tab((int32_t)((int64_t)xs + g.vmin), ss.world, g.owner_tab, g.vmin, g.n_slots) == ss.rank;
}

// The enumeration of the table passes (hist / rowsum / assign): position i of u's SORTED row.  A multi-edge (u -> x) x m — RMAT keeps
// its duplicates, and between two hubs m runs into the hundreds — is m entries of u's row and ONE pair: its table depends on the
// two rows only, so the first sorted occurrence (`rep`) gets the table and the work-list item, and the other m - 1 entries share
// its word (k_eb_dups).  Cost-weighted, config 3 has ~1.3x more entries than pairs into table rows (profiles/r05_table_plan.md).
//   k_item: what the work list records — the entry's input-order position (whole-graph handle) / the sorted position (shard).
template <bool SH>
__device__ inline bool eb_enum(const GraphView &g, const Row &ru, int32_t i, const ShardSel &ss, uint32_t &xs, int32_t &k_item, bool &rep) {
  if (SH) { rep = true; k_item = i; return eb_pair<true>(g, ru, i, ss, xs); }
  xs = g.sids[ru.off + i];
  rep = !(i > 0 && g.sids[ru.off + i - 1] == xs);
  k_item = (int32_t)g.sperm[ru.off + i];
  return true;
}

// pass 1: table bytes per cost class (class = bit length of the cost), so that the host can fit a threshold to the budget
// (hist[0][1]: pairs whose mask is inline — a shard's pair hash needs their count)
template <bool SH>
__global__ __launch_bounds__(TPB) void k_eb_hist(GraphView g, ShardSel ss, EbSel sel, unsigned long long *cursor,
                                                 unsigned long long *hist /* [64][2]: units, pairs */) {
  __shared__ unsigned long long lh[64][2];
  if (threadIdx.x < 128) lh[threadIdx.x >> 1][threadIdx.x & 1] = 0ull;
  __syncthreads();
  const int lane = lane_id();
  while (true) {
    const int64_t v0 = grab_u64(cursor, GRAB_SLOTS);
    if (v0 >= g.n_slots) break;
    for (int64_t u = v0; u < v0 + GRAB_SLOTS && u < g.n_slots; ++u) {
      const Row ru = eb_urow<SH>(g, u);
      for (int32_t k = lane; k < ru.deg; k += 64) {
        uint32_t xs; int32_t k_item; bool rep;
        if (!eb_enum<SH>(g, ru, k, ss, xs, k_item, rep) || !rep) continue;
        const Row rv = g.rows[xs];
        int64_t cost; int kind;
        const uint32_t un = eb_units(ru, rv, sel, cost, kind);
        if (kind == 1) {
          const int cls = prio_class(cost);                                  // 1 .. 62
          atomicAdd(&lh[cls][0], (unsigned long long)un);
          atomicAdd(&lh[cls][1], 1ull);
        } else if (kind == 2) {                                            // slot 63: the masks (all or nothing)
          atomicAdd(&lh[63][0], (unsigned long long)un);
          atomicAdd(&lh[63][1], 1ull);
        } else if (kind == 3) atomicAdd(&lh[0][1], 1ull);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 128 && lh[threadIdx.x >> 1][threadIdx.x & 1])
    atomicAdd(&hist[threadIdx.x], lh[threadIdx.x >> 1][threadIdx.x & 1]);
}

// pass 2: per row of prev, the units (bins: 64 B, masks: 16 B) and the number of its pairs that get a table in HBM
template <bool SH>
__global__ __launch_bounds__(TPB) void k_eb_rowsum(GraphView g, ShardSel ss, EbSel sel, unsigned long long *cursor,
                                                   unsigned long long *__restrict__ row_units,
                                                   unsigned long long *__restrict__ row_munits,
                                                   unsigned long long *__restrict__ row_pairs) {
  const int lane = lane_id();
  while (true) {
    const int64_t v0 = grab_u64(cursor, GRAB_SLOTS);
    if (v0 >= g.n_slots) break;
    for (int64_t u = v0; u < v0 + GRAB_SLOTS && u < g.n_slots; ++u) {
      const Row ru = eb_urow<SH>(g, u);
      unsigned long long un = 0, mu = 0, np = 0;
      for (int32_t k = lane; k < ru.deg; k += 64) {
        uint32_t xs; int32_t k_item; bool rep;
        if (!eb_enum<SH>(g, ru, k, ss, xs, k_item, rep) || !rep) continue;
        const Row rv = g.rows[xs];
        int64_t cost; int kind;
        const uint32_t x = eb_units(ru, rv, sel, cost, kind);
        if (kind == 1) un += x; else if (kind == 2) mu += x;
        np += (kind == 1 || kind == 2) ? 1u : 0u;
      }
      un = wave_sum_u64(un); mu = wave_sum_u64(mu); np = wave_sum_u64(np);
      if (lane == 0) { row_units[u] = un; row_munits[u] = mu; row_pairs[u] = np; }
    }
  }
}

// pass 3: table offsets per entry + the work list (row slot of prev, position inside the row).  Rows of curr up to 32
// candidates are not listed: k_eb_inline writes their masks into eb_off itself (SH: into the pair hash).
template <bool SH>
__global__ __launch_bounds__(TPB) void k_eb_assign(GraphView g, ShardSel ss, EbSel sel, unsigned long long *cursor,
                                                   const unsigned long long *__restrict__ row_units,
                                                   const unsigned long long *__restrict__ row_munits,
                                                   const unsigned long long *__restrict__ row_pairs,
                                                   uint32_t *__restrict__ eb_off, uint2 *__restrict__ items,
                                                   PairSlot *__restrict__ ph, uint32_t ph_buckets, uint32_t *__restrict__ item_off) {
  const int lane = lane_id();
  while (true) {
    const int64_t v0 = grab_u64(cursor, GRAB_SLOTS);
    if (v0 >= g.n_slots) break;
    for (int64_t u = v0; u < v0 + GRAB_SLOTS && u < g.n_slots; ++u) {
      const Row ru = eb_urow<SH>(g, u);
      unsigned long long ubase = row_units[u], mbase = row_munits[u], pbase = row_pairs[u];
      for (int32_t base = 0; base < ru.deg; base += 64) {
        const int32_t k = base + lane;                 // position in u's SORTED row (eb_enum)
        uint32_t x = 0, xs = 0; int kind = 0;
        int32_t k_item = 0; bool rep = false;
        bool mine = false;
        if (k < ru.deg && eb_enum<SH>(g, ru, k, ss, xs, k_item, rep)) {
          mine = true;
          const Row rv = g.rows[xs];
          int64_t cost;
          x = eb_units(ru, rv, sel, cost, kind);
        }
        const bool tab = rep && (kind == 1 || kind == 2);          // a multi-edge's other entries take their word from k_eb_dups
        uint32_t xb = (rep && kind == 1) ? x : 0u, xm = (rep && kind == 2) ? x : 0u, ib = xb, im = xm;
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t tb = (uint32_t)__shfl_up((int)ib, o), tm = (uint32_t)__shfl_up((int)im, o);
          if (lane >= o) { ib += tb; im += tm; }
        }
        const bool listed = tab;
        const unsigned long long has = __ballot(listed);
        const uint32_t word = !tab ? EB_NONE : kind == 1 ? (uint32_t)(ubase + ib - xb) : (uint32_t)(mbase + im - xm);
        if (!SH) { if (mine && (tab || kind == 0)) eb_off[ru.off + k_item] = word; }      // (kind 3: k_eb_inline writes the mask itself)
        else if (mine && listed) pair_insert(ph, ph_buckets, (uint32_t)u, xs, word);
        if (listed) {
          const unsigned long long idx = pbase + (unsigned long long)__popcll(has & ((1ull << lane) - 1ull));
          items[idx] = make_uint2((uint32_t)u, (uint32_t)k_item);
          if (SH) item_off[idx] = word;
        }
        ubase += (uint32_t)__builtin_amdgcn_readlane((int)ib, 63);
        mbase += (uint32_t)__builtin_amdgcn_readlane((int)im, 63);
        pbase += (unsigned long long)__popcll(has);
      }
    }
  }
}

// pass 3b (whole-graph handles): the other entries of a multi-edge take the word of the pair's first sorted occurrence
__global__ __launch_bounds__(TPB) void k_eb_dups(GraphView g, EbSel sel, unsigned long long *cursor, uint32_t *__restrict__ eb_off) {
  const int lane = lane_id();
  while (true) {
    const int64_t v0 = grab_u64(cursor, GRAB_SLOTS);
    if (v0 >= g.n_slots) break;
    for (int64_t u = v0; u < v0 + GRAB_SLOTS && u < g.n_slots; ++u) {
      const Row ru = g.rows[u];
      const uint32_t *sr = g.sids + ru.off;
      for (int32_t i = lane; i < ru.deg; i += 64) {
        const uint32_t xs = sr[i];
        if (i == 0 || sr[i - 1] != xs) continue;
        const Row rv = g.rows[xs];
        int64_t cost; int kind;
        (void)eb_units(ru, rv, sel, cost, kind);
        if (kind != 1 && kind != 2) continue;                  // no table (assign wrote EB_NONE) or an inline mask (k_eb_inline)
        int32_t lo = 0, hi = i;                                // first occurrence of xs in the sorted row
        while (lo < hi) { const int32_t mid = lo + ((hi - lo) >> 1); if (sr[mid] < xs) lo = mid + 1; else hi = mid; }
        eb_off[ru.off + g.sperm[ru.off + i]] = eb_off[ru.off + g.sperm[ru.off + lo]];
      }
    }
  }
}

// "x in N(u)?" for the mask builders: the hub bitmap of u, the edge hash set, or u's sorted row
__device__ inline bool eb_member(const GraphView &g, const Row &ru, uint32_t uslot, uint32_t xs) {
  const uint32_t hub = ru.flags >> ROW_HUB_SHIFT;
  if (hub && g.hub_bm) return (g.hub_bm[(int64_t)(hub - 1) * g.hub_words + (xs >> 5)] >> (xs & 31)) & 1u;
  if (g.ehash) return edge_exists(g.ehash, g.ehash_mask, uslot, xs);
  return sorted_contains(g.msids + ru.off, ru.deg, xs);     // (msids == sids on a whole-graph handle)
}

// inline masks: one LANE per pair (u -> v) with deg(v) <= 32
template <bool SH>
__global__ __launch_bounds__(TPB) void k_eb_inline(GraphView g, ShardSel ss, EbSel sel, unsigned long long *cursor, uint32_t *__restrict__ eb_off,
                                                   PairSlot *__restrict__ ph, uint32_t ph_buckets) {
  const int lane = lane_id();
  while (true) {
    const int64_t v0 = grab_u64(cursor, GRAB_SLOTS);
    if (v0 >= g.n_slots) break;
    for (int64_t u = v0; u < v0 + GRAB_SLOTS && u < g.n_slots; ++u) {
      const Row ru = eb_urow<SH>(g, u);
      for (int32_t k = lane; k < ru.deg; k += 64) {
        uint32_t xs;
        if (!eb_pair<SH>(g, ru, k, ss, xs)) continue;
        const Row rv = g.rows[xs];
        if (rv.deg <= 0 || rv.deg > sel.mask_max || rv.deg > INLINE_MAX_DEG) continue;
        uint32_t mask = 0u;
        for (int32_t c = 0; c < rv.deg; ++c)
          if (eb_member(g, ru, (uint32_t)u, (uint32_t)((int64_t)g.ent[rv.off + c].id - g.vmin))) mask |= 1u << c;
        if (SH) pair_insert(ph, ph_buckets, (uint32_t)u, xs, mask);
        else eb_off[ru.off + k] = mask;
      }
    }
  }
}

// pass 4: the tables.  One wave per pair: binned_fill exactly as a walk step over that pair would run it, then the
// prefix at every table chunk end goes to HBM.
// experiment switches (profiles/r03_eb_build.md: neither deeper prefetch nor wider lockstep searches move the build)
#ifndef SRW_EB_PREFETCH
#define SRW_EB_PREFETCH 1
#endif
#ifndef SRW_EB_P1K
#define SRW_EB_P1K 2
#endif
#ifndef SRW_EB_WAVES
#define SRW_EB_WAVES 4
#endif
#ifndef SRW_EB_HC
#define SRW_EB_HC 1024             // (2 048 ids per staged chunk halve the chunk advances but cost a wave per SIMD: 10.3 against 8.9 s at config 3, r05)
#endif
constexpr int EB_LDS_WORDS = 2 * BIN_CAP + SRW_EB_HC;      // binned_fill's bins + the staged ids of N(prev)
template <bool SH>
__global__ __launch_bounds__(TPB, (SRW_EB_HC > 1024 && SRW_EB_WAVES > 3) ? 3 : SRW_EB_WAVES) void k_eb_build(GraphView g, const uint2 *__restrict__ items, int64_t n_items, float p,
                                                     float q, int32_t mask_max, EbPolicy pol, const uint32_t *__restrict__ eb_off,
                                                     double *__restrict__ eb_bins, uint32_t *__restrict__ em_bits, unsigned long long *cursor,
                                                     unsigned long long *strat_count /* [8] */, int fill_tune, double *gscratch, int64_t gs_stride) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[TPB / 64][EB_LDS_WORDS];
  const int lane = lane_id();
  uint32_t *mine = lds[threadIdx.x >> 6];
  Member tm; tm.mode = 0; tm.bm = mine; tm.seg_base = 0; tm.ehash = g.ehash; tm.ehash_mask = g.ehash_mask;
  unsigned long long ns[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef SRW_PHASE_TIMING
  unsigned long long tt[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // wave time: [0] mask pairs, [1] item header, [2] P3, [3] P1, [4] P2, [5] W fill, [6] prefix sums + table write
#endif
  while (true) {
    const int64_t i0 = grab_u64(cursor, GRAB_ITEMS);
    if (i0 >= n_items) break;
    for (int64_t i = i0; i < i0 + GRAB_ITEMS && i < n_items; ++i) {
#ifdef SRW_PHASE_TIMING
      const unsigned long long t_it0 = wall_clock64();
#endif
      uint2 it = items[i];
      it.x = uni(it.x); it.y = uni(it.y);                 // (one wave, one pair: the pair's state is scalar — wave_primitives.h:uni)
      // SH: it.y = position in u's SORTED row of the membership structure, eb_off = the work list's own offsets (item_off)
      const Row ru = uniform_row(eb_urow<SH>(g, it.x));
      const int64_t e = ru.off + it.y;
      const Row rv = uniform_row(g.rows[SH ? (int64_t)g.msids[e] : (int64_t)g.ent[e].id - g.vmin]);
      const uint32_t tab_word = uni(eb_off[SH ? i : e]);
      Bias b;
      b.p = p; b.q = q; b.prev = (int32_t)((int64_t)it.x + g.vmin); b.second_order = true; b.need_member = true;
      b.prev_sids = g.msids + ru.off; b.prev_deg = ru.deg; b.vmin = g.vmin; b.prev_hub = ru.flags >> ROW_HUB_SHIFT;
      if (rv.deg <= mask_max) {                   // membership mask: 64 candidates per round, one probe each
        uint32_t *out = em_bits + (size_t)tab_word * 4;
        const int32_t n_words = ((((rv.deg + 31) >> 5) + 3) >> 2) << 2;
        // (a row of at most 255 candidates: its four rounds of 64 in lockstep — the entries, then the probes, each one round trip)
        uint32_t xs[4]; bool want[4], in[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int32_t c = r4 * 64 + lane;
          want[r4] = c < rv.deg; in[r4] = false;
          xs[r4] = want[r4] ? (uint32_t)((int64_t)g.ent[rv.off + c].id - g.vmin) : 0u;
        }
        const uint32_t hub = ru.flags >> ROW_HUB_SHIFT;
        if (hub && g.hub_bm) {
          uint32_t wd[4];
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) wd[r4] = want[r4] ? g.hub_bm[(int64_t)(hub - 1) * g.hub_words + (xs[r4] >> 5)] : 0u;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) in[r4] = (wd[r4] >> (xs[r4] & 31)) & 1u;
        } else if (g.ehash) {
          edge_exists_n<4>(g.ehash, g.ehash_mask, it.x, xs, want, in);
        } else {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) in[r4] = want[r4] && sorted_contains(g.msids + ru.off, ru.deg, xs[r4]);
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int32_t c0 = r4 * 64;
          const unsigned long long mm = __ballot(in[r4]);            // (every lane votes: outside the bounds check)
          if (c0 < n_words * 32 && lane == 0) { out[c0 >> 5] = (uint32_t)mm; if ((c0 >> 5) + 1 < n_words) out[(c0 >> 5) + 1] = (uint32_t)(mm >> 32); }
        }
        ns[0] += 1;
#ifdef SRW_PHASE_TIMING
        tt[0] += wall_clock64() - t_it0;
#endif
        continue;
      }
#ifdef SRW_PHASE_TIMING
      const unsigned long long t_hdr = wall_clock64();
      tt[1] += t_hdr - t_it0;
#endif
      const PairGeom pg = eb_pair_geometry(rv.deg, ru.deg, pol);
      BinGeom gc; gc.csh = pg.csh; gc.n_bins = pg.n_bins;
      // more chunks than the wave's LDS holds bins for (the finer tables of the pairs with a long N(prev)): the bins live in this
      // wave's slice of an HBM scratch (exact f64 atomics: every partial sum is exact in any order) and are filled at the table's own granularity
      const bool big = gc.n_bins > BIN_CAP;
      const BinGeom gf = (big || gc.csh < 6) ? gc : bin_geometry(rv.deg, 6, BIN_CAP);     // fill granularity: the walk's own
      double *gbins = big ? gscratch + ((int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6)) * gs_stride : nullptr;
      unsigned long long ab = 0; unsigned su = 0;
      // chunk masks: rows of at most EB_CM_LIMIT candidates have at most 256 fill bins, the upper half of the bins' LDS is free for the pair's
      // bits; a LONGER row's bits (round 6) are OR-ed straight into the pair's table block in HBM (binned_fill clears them first; the
      // block is this wave's alone)
      const double *out0 = eb_bins + (size_t)tab_word * 8;
      const bool long_mask = pg.cmask && rv.deg > EB_CM_LIMIT;
      uint32_t *mbits = !pg.cmask ? nullptr : !long_mask ? mine + BIN_CAP
                        : reinterpret_cast<uint32_t *>(const_cast<double *>(out0) + (size_t)eb_layout(pol.f32 && (rv.flags & ROW_PQ_F32), gc.n_bins, true, rv.deg, eb_pair_u16(rv.flags, gc.csh, pol)).cm_off * 8);
      if (long_mask && lane == 0 && (((rv.deg + 31) >> 5) & 1)) mbits[(rv.deg + 31) >> 5] = 0u;      // (the upper half of the last 64-bit word)
      if (big) binned_fill<SRW_EB_PREFETCH, SRW_EB_P1K, true, SRW_EB_HC>(g, rv, b, mine, fill_tune, gf, tm, ab, su, mbits, gbins);
      else binned_fill<SRW_EB_PREFETCH, SRW_EB_P1K, false, SRW_EB_HC>(g, rv, b, mine, fill_tune, gf, tm, ab, su, mbits, nullptr, long_mask);
      ns[su & 7] += 1;
#ifdef SRW_PHASE_TIMING
      const unsigned long long t_fill = wall_clock64();
      tt[2 + (su & 3)] += t_fill - t_hdr;              // su: 1 P1, 2 P2, 3 W, 4 P3 (-> slot 2)
#endif
      const double *bins = reinterpret_cast<const double *>(mine);
      double *out = eb_bins + (size_t)tab_word * 8;
      const int up = gc.csh - gf.csh;
      const PqRow PQ(g, rv.off);            // the table keeps the complete numerator A'_end(j) = PQ[end_j] + corrections
      const bool as_f32 = pol.f32 && (rv.flags & ROW_PQ_F32);      // every such sum is exactly representable in binary32
      const bool as_u16 = eb_pair_u16(rv.flags, gc.csh, pol);           // level 0 as the chunks' own masses, u16 multiples of the row's unit 2^G
      const double inv_unit = as_u16 ? 1.0 / eb_row_unit(rv.flags) : 0.0;
      const EbLayout lay = eb_layout(as_f32, gc.n_bins, pg.cmask, rv.deg, as_u16);
      // level 0 = the chunk prefixes; level 1 / 2 = the last element of every block of 64 of the level below (the walk's search tree)
      for (int L = 0; L < 3; ++L) {
        const int32_t cnt = L == 0 ? gc.n_bins : L == 1 ? lay.n1 : lay.n2;
        double *lo_ = out + (size_t)(L == 0 ? lay.l0_off : L == 1 ? lay.l1_off : lay.l2_off) * 8;
        for (int32_t t = lane; t < cnt; t += 64) {
          const int64_t je = (((int64_t)t + 1) << (6 * L)) - 1;
          const int64_t j = je < gc.n_bins ? je : gc.n_bins - 1;          // the chunk this element is the prefix of
          const int64_t fi = ((j + 1) << up) - 1;
          const int64_t ke = ((j + 1) << gc.csh) - 1;
          const int64_t fb = fi < gf.n_bins ? fi : gf.n_bins - 1;
          const double corr = big ? __hip_atomic_load(gbins + fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : bins[fb];
          const double a = PQ[ke < rv.deg ? ke : rv.deg - 1] + corr;
          if (L == 0 && as_u16) {                  // the chunk's own mass = this prefix - the previous one (exact), in units of 2^G
            double a0 = 0.0;
            if (j > 0) {
              const int64_t fp = (j << up) - 1, kp = (j << gc.csh) - 1;
              a0 = PQ[kp] + (big ? __hip_atomic_load(gbins + fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : bins[fp]);
            }
            const double du_ = (a - a0) * inv_unit;
            const unsigned short q16 = (unsigned short)(du_ >= 0.0 && du_ <= 65535.0 ? du_ : 0.0);
            if ((double)q16 != du_) atomicAdd(&strat_count[7], 1ull);         // must never happen (pq_row_u16_bits' bound)
            reinterpret_cast<unsigned short *>(lo_)[t] = q16;
          } else if (as_f32) {
            reinterpret_cast<float *>(lo_)[t] = (float)a;
            if ((double)(float)a != a) atomicAdd(&strat_count[7], 1ull);      // must never happen (pq_row_f32's bound)
          } else lo_[t] = a;
        }
      }
      if (mbits && !long_mask) {
        unsigned long long *mo = reinterpret_cast<unsigned long long *>(out + (size_t)lay.cm_off * 8);
        const int32_t n_mw = (rv.deg + 63) >> 6, n_w32 = (rv.deg + 31) >> 5;
        for (int32_t t = lane; t < n_mw; t += 64)
          mo[t] = (unsigned long long)mbits[2 * t] | ((2 * t + 1 < n_w32) ? ((unsigned long long)mbits[2 * t + 1] << 32) : 0ull);
      }
      __builtin_amdgcn_wave_barrier();          // the next fill clears the bins
#ifdef SRW_PHASE_TIMING
      tt[6] += wall_clock64() - t_fill;
#endif
    }
  }
#ifdef SRW_PHASE_TIMING
  tt[7] = tm.t_pass1;                                  // inside W: the 10-level lower bounds in LDS
  if (lane == 0)
    for (int i = 0; i < 8; ++i) if (tt[i]) atomicAdd(&strat_count[8 + i], tt[i] >> 10);
#endif
  if (lane == 0)
    for (int i = 0; i < 8; ++i) if (ns[i]) atomicAdd(&strat_count[i], ns[i]);
}

// The build in segments (vm_buf.h): tables are laid out in row order and so is the work list, so the work list splits where the
// tables cross a chunk boundary of the progressively mapped buffer.  Segment k = the rows that START in chunk k; seg_item[k] = its first
// work-list item, seg_end[k] = where its last row's tables end (64-byte units) — what must be mapped before it runs.
__global__ void k_eb_segments(const unsigned long long *__restrict__ row_units, const unsigned long long *__restrict__ row_pairs, int64_t n_slots,
                              unsigned long long chunk_units, int n_seg, unsigned long long total_units, unsigned long long n_listed,
                              unsigned long long *__restrict__ seg_item /* [n_seg + 1] */, unsigned long long *__restrict__ seg_end /* [n_seg] */) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k > n_seg) return;
  if (k == 0) { seg_item[0] = 0ull; return; }
  if (k == n_seg) { seg_item[k] = n_listed; seg_end[k - 1] = total_units; return; }
  const unsigned long long B = (unsigned long long)k * chunk_units;
  int64_t lo = 0, hi = n_slots;
  while (lo < hi) { const int64_t mid = lo + ((hi - lo) >> 1); if (row_units[mid] < B) lo = mid + 1; else hi = mid; }
  seg_item[k] = lo < n_slots ? row_pairs[lo] : n_listed;
  seg_end[k - 1] = lo < n_slots ? row_units[lo] : total_units;
}

// rev[e] for every entry e = (u -> v): where the return edges sit in N(v) (k_walk_q1) — the index, in v's SORTED row, of
// the first entry that leads back to u, and how many there are (multi-edges).  One lane per entry: lower bound of u in
// v's sorted row, then the run of equal ids.
__global__ __launch_bounds__(TPB) void k_rev_build(GraphView g, unsigned long long *cursor, RevEnt *__restrict__ rev) {
  const int lane = lane_id();
  while (true) {
    const int64_t v0 = grab_u64(cursor, GRAB_SLOTS);
    if (v0 >= g.n_slots) break;
    for (int64_t u = v0; u < v0 + GRAB_SLOTS && u < g.n_slots; ++u) {
      const Row ru = g.rows[u];
      for (int32_t k = lane; k < ru.deg; k += 64) {
        const Row rv = g.rows[(int64_t)g.ent[ru.off + k].id - g.vmin];
        const uint32_t *cs = g.sids + rv.off;
        int32_t lo = 0, hi = rv.deg;
        while (lo < hi) { const int32_t mid = lo + ((hi - lo) >> 1); if (cs[mid] < (uint32_t)u) lo = mid + 1; else hi = mid; }
        uint32_t cnt = 0;
        while (cnt < 255u && lo + (int32_t)cnt < rv.deg && cs[lo + cnt] == (uint32_t)u) ++cnt;
        RevEnt o; o.cl = cnt ? ((cnt << 24) | (uint32_t)lo) : REV_NONE; o.pos0 = 0u; o.w0 = 0.0f; o.pad = 0u;   // (rows have fewer than 2^23 entries: CFO_NDEG_MAX)
        if (cnt) { o.pos0 = g.sperm[rv.off + lo]; o.w0 = g.sw[rv.off + lo]; }
        rev[ru.off + k] = o;
      }
    }
  }
}

size_t env_gb(const char *name, size_t dflt_gb) {
  const char *e = getenv(name);
  if (!e || !*e) return dflt_gb << 30;
  return (size_t)(atof(e) * (double)((size_t)1 << 30));
}

}  // namespace

// smallest chunk of a table: 2^8 = 256 candidates (one round of the located chunk's evaluation); SRW_EB_MIN_SH: experiments
// The callers try 6 and 7 once the number of chunks is chosen and keep the finest whose complete set still fits (Graph::eb_min_sh_sel):
// the located chunk is what a table step streams and probes, and the rows of 256 .. 16 384 candidates — half of all steps at config 3 —
// have fewer than 64 chunks of 256 (config 3: 256 / 128 / 64 candidates 5.1 / 5.8 / 6.8e8 steps/s, 77 / 111 / 138 GB, s52).
static int eb_min_shift(const Graph &g) {
  const char *e = getenv("SRW_EB_MIN_SH");
  const int v = e && *e ? atoi(e) : g.eb_min_sh_sel;
  return v < 2 ? 2 : v > 12 ? 12 : v;
}
// Chunk masks: the bins table of a pair into a row of at most this many candidates is followed by the pair's membership mask over
// the candidate positions (one bit per candidate) — the located chunk's evaluation then needs no membership probe at all
// (profiles/r04_request_attribution.md: the probes were 48 of the 62 HBM requests of an average step at config 3).  Chunks must
// start on mask words (chunk shift >= 6).  SRW_EB_CM_MAX: experiments.
static int32_t eb_cm_select(const Graph &g, int mode, int min_sh) {
  if (mode || min_sh < 6) return 0;
  const char *e = getenv("SRW_EB_CM_MAX");
  const int32_t v = e && *e ? atoi(e) : g.eb_cm_sel;
  return v < MASK_MAX_DEG ? 0 : v;      // (beyond EB_CM_LIMIT: long masks, sampling.h:eb_pair_geometry)
}
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e && *e ? atoi(e) : dflt; }
// The geometry of the next set of tables: Graph::eb_min_sh_sel / eb_cm_sel / eb_fine_cap_sel are the planners' choices
// (prepare_tables, prepare_shard_tables); mode 1 (tests): chunks of 4 candidates, no masks, no finer tables.
static EbPolicy eb_policy(const Graph &g, int mode, int bins_cap) {
  EbPolicy P;
  P.min_sh = mode ? 2 : eb_min_shift(g); P.cap = bins_cap;
  P.cm_max = eb_cm_select(g, mode, P.min_sh);
  P.cm_min_du = std::max(0, env_int("SRW_EB_CM_MIN_DU", 1024));      // N(prev) up to 1 024 ids: staged in LDS by the walk, searched there
  P.cm_ratio = P.cm_max ? std::max(0, env_int("SRW_EB_CM_RATIO", g.eb_cm_ratio_sel)) : 0;
  P.fine_min_du = std::max(0, env_int("SRW_EB_FINE_MIN_DU", 1024));
  P.fine_sh = std::min(12, std::max(6, env_int("SRW_EB_FINE_SH", 6)));
  P.fine_cap = mode ? 0 : std::min(EB_FINE_CAP_LIMIT, std::max(0, env_int("SRW_EB_FINE_CAP", g.eb_fine_cap_sel)));
  if (P.fine_cap <= P.cap) P.fine_cap = 0;
  { const char *e = getenv("SRW_EB_NO_F32"); P.f32 = (e && *e == '1') ? 0 : 1; }
  { const char *e = getenv("SRW_EB_NO_U16"); P.u16 = (mode || (e && *e == '1')) ? 0 : 1; }
  return P;
}
// HBM a COMPLETE set of tables would take (every pair into a certified row + every mask + offsets + the work list):
// prepare_tables sizes the hub bitmaps with what is left beside it.  0: no tables possible.
size_t edge_tables_full_bytes(srw_handle *h, int mode, int bins_cap) {
  Graph &g = h->g;
  if (!g.has_pq || !g.has_member || g.n_entries <= 0) return 0;
  hipStream_t st = h->stream;
  EbSel sel;
  sel.mask_max = mode ? 0 : MASK_MAX_DEG - 1;
  sel.min_deg = mode ? 1 : MASK_MAX_DEG; sel.min_cost = 0;
  { const char *e = getenv("SRW_EB_NO_MASKS"); if (e && *e == '1') sel.mask_max = 0; }
  sel.has_ehash = 1; sel.has_hub = 1;               // (they only move priorities, not sizes)
  sel.pol = eb_policy(g, mode, bins_cap);
  DevBuf<unsigned long long> cursor, hist;
  cursor.alloc(1); hist.alloc(128);
  SRW_HIP(hipMemsetAsync(cursor.p, 0, 8, st));
  SRW_HIP(hipMemsetAsync(hist.p, 0, 128 * 8, st));
  hipLaunchKernelGGL((k_eb_hist<false>), dim3(h->n_cus * 8), dim3(TPB), 0, st, g.view(), ShardSel{0, 1}, sel, cursor.p, hist.p);
  SRW_HIP(hipGetLastError());
  unsigned long long hh[128];
  SRW_HIP(hipMemcpyAsync(hh, hist.p, sizeof(hh), hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  unsigned long long bytes = hh[126] * 16 + hh[127] * 8;
  for (int cls = 1; cls <= 62; ++cls) bytes += hh[cls * 2] * 64 + hh[cls * 2 + 1] * 8;
  return (size_t)bytes + (size_t)g.n_entries * 4 + (size_t)g.n_slots * 24;
}

void build_edge_tables(srw_handle *h, float p, float q, int mode, int bins_cap) {
  Graph &g = h->g;
  uint32_t pb, qb; memcpy(&pb, &p, 4); memcpy(&qb, &q, 4);
  if (bins_cap < 8) bins_cap = 8;
  if (bins_cap > BIN_CAP) bins_cap = BIN_CAP;
  if (const char *e = getenv("SRW_EB_FAIL_ABOVE"); e && *e && bins_cap > atoi(e))      // tests: prepare_tables' fallback
    throw Error(SRW_ERR_NOMEM, "simulated allocation failure of the per-edge tables (SRW_EB_FAIL_ABOVE)");
  { const char *e = getenv("SRW_EB_NO_F32"); const int want_f32 = (e && *e == '1') ? 0 : 1;
    if (g.has_eb && !g.eb_sharded && g.eb_pbits == pb && g.eb_qbits == qb && g.eb_mode == mode && g.ebp.f32 == want_f32 && g.ebp.cap == bins_cap) return; }
  hipStream_t st = h->stream;
  g.has_eb = false; g.eb_tables = 0; g.eb_bytes = 0; g.eb_build_ms = 0.0; g.eb_complete = false;
  g.eb_bins.release(); g.em_bits.release();
  if (!g.has_pq || !g.has_member || g.n_entries <= 0) return;
  const auto t0 = std::chrono::steady_clock::now();
  EbSel sel;
  // mode 1 (tests): bins tables for every certified row, chunks of 4 candidates, no masks
  sel.mask_max = mode ? 0 : MASK_MAX_DEG - 1;
  sel.min_deg = mode ? 1 : MASK_MAX_DEG;
  { const char *e = getenv("SRW_EB_MIN_COST"); sel.min_cost = (!mode && e && *e) ? atoll(e) : 0; }
  { const char *e = getenv("SRW_EB_NO_MASKS"); if (e && *e == '1') sel.mask_max = 0; }
  sel.pol = eb_policy(g, mode, bins_cap);
  g.ebp = sel.pol; g.eb_f32 = sel.pol.f32; g.eb_cap = bins_cap;
  sel.has_ehash = (g.has_ehash && g.use_ehash) ? 1 : 0; sel.has_hub = (g.has_hub && g.use_hub) ? 1 : 0;
  g.eb_min_sh = sel.pol.min_sh;
  // budget: what is free now minus the offsets and a reserve for the walk's own buffers
  size_t free_b = 0, total_b = 0;
  SRW_HIP(hipMemGetInfo(&free_b, &total_b));
  free_b += g.eb_off.n * sizeof(uint32_t);
  const size_t fixed = (size_t)g.n_entries * 4 + (size_t)g.n_slots * 24 + (getenv("SRW_EB_RESERVE_GB") ? env_gb("SRW_EB_RESERVE_GB", 24) : g.eb_reserve);
  if (free_b < fixed + ((size_t)64 << 20)) return;
  size_t budget = std::min(env_gb("SRW_EB_BUDGET_GB", g.eb_budget_gb), free_b - fixed);
  const int blocks = h->n_cus * 8;
  DevBuf<unsigned long long> cursor, hist, row_units, row_munits, row_pairs;
  cursor.alloc(1); hist.alloc(128);
  SRW_HIP(hipMemsetAsync(cursor.p, 0, 8, st));
  SRW_HIP(hipMemsetAsync(hist.p, 0, 128 * 8, st));
  hipLaunchKernelGGL((k_eb_hist<false>), dim3(blocks), dim3(TPB), 0, st, g.view(), ShardSel{0, 1}, sel, cursor.p, hist.p);
  SRW_HIP(hipGetLastError());
  unsigned long long hh[128];
  SRW_HIP(hipMemcpyAsync(hh, hist.p, sizeof(hh), hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  // the masks first (16 B units + 8 B per work-list item): cheap, and they serve every step into a short row
  unsigned long long munits = hh[126], mpairs = hh[127];
  if (munits * 16 + mpairs * 8 > budget || munits >= 0xFFFFFFF0ull) { sel.mask_max = 0; munits = 0; mpairs = 0; }
  else budget -= munits * 16 + mpairs * 8;
  // then the bins tables, highest priority classes first while they fit
  unsigned long long units = 0, pairs = 0;
  int cls = 63;
  while (cls > 1) {
    const unsigned long long nu = units + hh[(cls - 1) * 2], np = pairs + hh[(cls - 1) * 2 + 1];
    if (nu * 64 + np * 8 > budget || nu >= 0xFFFFFFF0ull) break;
    units = nu; pairs = np; --cls;
  }
  g.eb_complete = cls <= 1 && sel.mask_max > 0;      // nothing was cut by the budget (k_walk_tables may run)
  if (cls > 1) sel.min_cost = std::max<int64_t>(sel.min_cost, prio_class_floor(cls));
  if (pairs == 0) sel.min_deg = 0x7FFFFFFF;      // no bins tables at all
  if (pairs + mpairs == 0 && sel.mask_max == 0) return;
  row_units.alloc((size_t)g.n_slots + 1); row_munits.alloc((size_t)g.n_slots + 1); row_pairs.alloc((size_t)g.n_slots + 1);
  SRW_HIP(hipMemsetAsync(cursor.p, 0, 8, st));
  hipLaunchKernelGGL((k_eb_rowsum<false>), dim3(blocks), dim3(TPB), 0, st, g.view(), ShardSel{0, 1}, sel, cursor.p,
                     row_units.p, row_munits.p, row_pairs.p);
  SRW_HIP(hipGetLastError());
  {
    size_t tb = 0;
    SRW_HIP(rocprim::exclusive_scan(nullptr, tb, row_units.p, row_units.p, 0ull, (size_t)g.n_slots, rocprim::plus<unsigned long long>(), st));
    DevBuf<char> temp; temp.alloc(tb);
    SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, row_units.p, row_units.p, 0ull, (size_t)g.n_slots, rocprim::plus<unsigned long long>(), st));
    SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, row_munits.p, row_munits.p, 0ull, (size_t)g.n_slots, rocprim::plus<unsigned long long>(), st));
    SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, row_pairs.p, row_pairs.p, 0ull, (size_t)g.n_slots, rocprim::plus<unsigned long long>(), st));
    SRW_HIP(hipStreamSynchronize(st));
  }
  const unsigned long long all_pairs = pairs + mpairs;
  const auto t_alloc0 = std::chrono::steady_clock::now();
  g.eb_off.ensure((size_t)g.n_entries);
  g.eb_bins.alloc_progressive((size_t)units * 8, h->cfg.device);      // the pages arrive while the build fills them (vm_buf.h)
  g.em_bits.alloc((size_t)munits * 4);
  DevBuf<uint2> items; items.alloc((size_t)all_pairs);
  if (getenv("SRW_TIMING"))
    fprintf(stderr, "[edge tables] allocations (%.1f GB of tables, %.1f GB of masks, %.1f GB work list): %.0f ms; planning passes before them: %.0f ms\n",
            (double)units * 64 / 1e9, (double)munits * 16 / 1e9, (double)all_pairs * 8 / 1e9,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc0).count(),
            std::chrono::duration<double, std::milli>(t_alloc0 - t0).count());
  SRW_HIP(hipMemsetAsync(cursor.p, 0, 8, st));
  g.eb_sharded = false; g.ph.release(); g.ph_buckets = 0;
  hipLaunchKernelGGL((k_eb_assign<false>), dim3(blocks), dim3(TPB), 0, st, g.view(), ShardSel{0, 1}, sel, cursor.p,
                     row_units.p, row_munits.p, row_pairs.p, g.eb_off.p, items.p, (PairSlot *)nullptr, 0u, (uint32_t *)nullptr);
  SRW_HIP(hipGetLastError());
  // The multi-edge copies (k_eb_dups) and the inline masks (k_eb_inline) write eb_off words that the table build neither reads nor
  // writes (it reads the representatives' words, written by k_eb_assign above): they run on a side stream NEXT TO the build (0.4 s at config 3).
  DevBuf<unsigned long long> cursor3; cursor3.alloc(1);          // (declared before the guard: released after the side stream has drained)
  hipStream_t side = nullptr; hipEvent_t side_ev = nullptr;
  struct SideGuard { hipStream_t &s; hipEvent_t &e; ~SideGuard() { if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } if (e) (void)hipEventDestroy(e); } } side_guard{side, side_ev};
  SRW_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  SRW_HIP(hipEventCreateWithFlags(&side_ev, hipEventDisableTiming));
  SRW_HIP(hipEventRecord(side_ev, st));
  SRW_HIP(hipStreamWaitEvent(side, side_ev, 0));
  SRW_HIP(hipMemsetAsync(cursor3.p, 0, 8, side));
  hipLaunchKernelGGL(k_eb_dups, dim3(blocks), dim3(TPB), 0, side, g.view(), sel, cursor3.p, g.eb_off.p);
  SRW_HIP(hipGetLastError());
  g.has_eb = true; g.use_eb = true; g.eb_mask_max = sel.mask_max;
  GraphView gv = g.view();
  gv.eb_off = nullptr;                          // the builders never read the tables they are writing
  if (sel.mask_max > 0) {
    SRW_HIP(hipMemsetAsync(cursor3.p, 0, 8, side));
    hipLaunchKernelGGL((k_eb_inline<false>), dim3(blocks), dim3(TPB), 0, side, gv, ShardSel{0, 1}, sel, cursor3.p, g.eb_off.p, (PairSlot *)nullptr, 0u);
    SRW_HIP(hipGetLastError());
  }
  SRW_HIP(hipEventRecord(side_ev, side));       // (the handle's stream waits for it after the build: below)
  unsigned long long sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // tables of more chunks than the build's LDS holds bins for: one f64 per chunk and wave in an HBM scratch
  DevBuf<double> gscratch;
  const int64_t gs_stride = sel.pol.fine_cap > BIN_CAP ? (int64_t)sel.pol.fine_cap : 0;
  if (gs_stride) gscratch.alloc((size_t)blocks * (TPB / 64) * (size_t)gs_stride);
  double waited_ms = 0.0; int n_launches = 0;
  if (all_pairs) {
    const int fill_tune = getenv("SRW_EB_FILL_TUNE") ? atoi(getenv("SRW_EB_FILL_TUNE")) : 0;
    SRW_HIP(hipMemsetAsync(hist.p, 0, 16 * 8, st));
    // the segments of a progressively mapped buffer alternate between two streams (own cursor, own HBM-scratch bins): the tail of one
    // segment — its last, slowest pairs — overlaps the start of the next (41 launches in a row cost 0.3 - 0.6 s of tails at config 3)
    hipStream_t aux = nullptr; hipEvent_t ev = nullptr;
    DevBuf<unsigned long long> cursor2; DevBuf<double> gscratch2;
    struct AuxGuard { hipStream_t &s; hipEvent_t &e; ~AuxGuard() { if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } if (e) (void)hipEventDestroy(e); } } aux_guard{aux, ev};
    auto launch = [&](unsigned long long i0, unsigned long long n_it, int which = 0) {
      hipStream_t ls = which ? aux : st;
      unsigned long long *cur = which ? cursor2.p : cursor.p;
      SRW_HIP(hipMemsetAsync(cur, 0, 8, ls));
      hipLaunchKernelGGL((k_eb_build<false>), dim3(blocks), dim3(TPB), 0, ls, gv, items.p + i0, (int64_t)n_it, p, q, sel.mask_max,
                         sel.pol, g.eb_off.p, g.eb_bins.p, g.em_bits.p, cur, hist.p, fill_tune, which ? gscratch2.p : gscratch.p, gs_stride);
      SRW_HIP(hipGetLastError());
      ++n_launches;
    };
    if (!g.eb_bins.progressive()) launch(0ull, all_pairs);
    else {
      if (!getenv("SRW_EB_ONE_STREAM")) {
        cursor2.alloc(1);
        if (gs_stride) gscratch2.alloc((size_t)blocks * (TPB / 64) * (size_t)gs_stride);
        SRW_HIP(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
        SRW_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        SRW_HIP(hipEventRecord(ev, st));                       // (everything the build reads was written on the handle's stream)
        SRW_HIP(hipStreamWaitEvent(aux, ev, 0));
      }
      // one launch per chunk of the table buffer, each as soon as its rows' tables are backed by pages
      const unsigned long long chunk_units = (unsigned long long)(g.eb_bins.chunk_bytes() / 64);
      const int n_seg = (int)((units + chunk_units - 1) / chunk_units);
      DevBuf<unsigned long long> d_seg; d_seg.alloc((size_t)2 * n_seg + 2);
      hipLaunchKernelGGL(k_eb_segments, dim3((unsigned)((n_seg + 1 + 255) / 256)), dim3(256), 0, st, row_units.p, row_pairs.p, (int64_t)g.n_slots, chunk_units,
                         n_seg, units, all_pairs, d_seg.p, d_seg.p + n_seg + 1);
      SRW_HIP(hipGetLastError());
      std::vector<unsigned long long> seg((size_t)2 * n_seg + 2);
      SRW_HIP(hipMemcpyAsync(seg.data(), d_seg.p, seg.size() * 8, hipMemcpyDeviceToHost, st));
      SRW_HIP(hipStreamSynchronize(st));
      for (int k = 0; k < n_seg; ++k) {
        const unsigned long long i0 = seg[(size_t)k], i1 = seg[(size_t)k + 1];
        if (i1 <= i0) continue;
        const auto tw = std::chrono::steady_clock::now();
        g.eb_bins.wait_mapped((size_t)seg[(size_t)n_seg + 1 + k] * 64);
        waited_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
        launch(i0, i1 - i0, aux ? (n_launches & 1) : 0);
      }
      if (aux) { SRW_HIP(hipEventRecord(ev, aux)); SRW_HIP(hipStreamWaitEvent(st, ev, 0)); }       // the handle's stream goes on after both
      const auto tw = std::chrono::steady_clock::now();
      g.eb_bins.wait_all();
      waited_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
    }
    SRW_HIP(hipMemcpyAsync(sc, hist.p, sizeof(sc), hipMemcpyDeviceToHost, st));
  } else g.eb_bins.wait_all();
  SRW_HIP(hipStreamWaitEvent(st, side_ev, 0));
  SRW_HIP(hipStreamSynchronize(st));
  if (getenv("SRW_TIMING") && g.eb_bins.progressive())
    fprintf(stderr, "[edge tables] table buffer mapped in chunks of %.0f MiB while the build ran: %d launches, the host waited %.0f ms for pages\n",
            (double)g.eb_bins.chunk_bytes() / (double)((size_t)1 << 20), n_launches, waited_ms);
#ifdef SRW_PHASE_TIMING
  {
    unsigned long long tt[8];
    SRW_HIP(hipMemcpy(tt, hist.p + 8, sizeof(tt), hipMemcpyDeviceToHost));
    static const char *nm[8] = {"mask pairs", "item header", "fill P3", "fill P1", "fill P2", "fill W", "prefix + write", "(of W: LDS lower bounds)"};
    fprintf(stderr, "[eb_build phase] wave-ms:");
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.0f", nm[i], (double)tt[i] * 1024.0 / 100e3);
    fprintf(stderr, "\n");
  }
#endif
  if (sc[7]) throw Error(SRW_ERR_INVALID, "per-edge tables: a prefix sum of a ROW_PQ_F32 row was not exactly representable in binary32 (or a chunk mass not in 16 bits)");
  g.eb_pbits = pb; g.eb_qbits = qb; g.eb_mode = mode;
  g.eb_tables = (int64_t)all_pairs;
  g.eb_bytes = (int64_t)(units * 64 + munits * 16 + (unsigned long long)g.n_entries * 4);
  g.eb_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("SRW_TIMING"))
    fprintf(stderr, "[edge tables] chunks of >= %d candidates, chunk masks up to %d candidates (N(prev) > %d or deg(curr) <= %d deg(prev)), up to %d chunks for unmasked pairs with N(prev) > %d; ",
            1 << sel.pol.min_sh, sel.pol.cm_max, sel.pol.cm_min_du, sel.pol.cm_ratio, sel.pol.fine_cap ? sel.pol.fine_cap : sel.pol.cap, sel.pol.fine_min_du);
  if (getenv("SRW_TIMING"))
    fprintf(stderr, "[edge tables] up to %d chunks per table; %llu bins tables (%.2f GB, min priority %lld; fill: P1 %llu, P2 %llu, W %llu, P3 %llu) + %llu masks (%.2f GB) "
            "+ inline masks (%.2f GB of offsets), built in %.0f ms\n", sel.pol.cap, pairs, (double)units * 64 / 1e9, (long long)sel.min_cost, sc[1], sc[2],
            sc[3], sc[4], mpairs, (double)munits * 16 / 1e9, (double)g.n_entries * 4 / 1e9, g.eb_build_ms);
}

// ---- the same tables on a vertex-sharded handle ------------------------------------------------------------------------
// north_star's biased multi-GPU walk (directed, p = 4, q = .5 on 8 GPUs) replaces RandomWalk.scala:92-139 — the shuffle
// ships N(prev) with every walker and the receiving partition recomputes computeSecondOrderWeights (RandomSample.scala:
// 27-44).  Here owner(curr) holds the precomputed table of every pair (prev -> curr) into its own rows: 1 / world of the
// whole graph's set (config 5: 168 GB -> 21 GB per GPU at world 8), found through the pair hash because a walker arrives
// with (prev, curr) and nothing else.  The pairs are enumerated from the replicated membership structure (every shard sees
// every (u, sorted N(u)) and keeps those whose x it owns); N(prev) for the fill and for the located chunk's probes comes
// from the same structure.  A COMPLETE set at the finest resolution that fits, or none (the on-the-fly samplers).
namespace {
struct ShardTabPlan { unsigned long long units, pairs, munits, mpairs, inl; size_t bytes; uint32_t buckets; };

EbSel shard_sel(const Graph &g, int mode, int bins_cap) {
  EbSel sel;
  sel.mask_max = mode ? 0 : MASK_MAX_DEG - 1;
  sel.min_deg = mode ? 1 : MASK_MAX_DEG; sel.min_cost = 0;
  { const char *e = getenv("SRW_EB_NO_MASKS"); if (e && *e == '1') sel.mask_max = 0; }
  sel.has_ehash = 0; sel.has_hub = 1; sel.pol = eb_policy(g, mode, bins_cap);
  return sel;
}

bool shard_plan(srw_handle *h, const EbSel &sel, ShardTabPlan &pl) {
  Graph &g = h->g;
  hipStream_t st = h->stream;
  DevBuf<unsigned long long> cursor, hist;
  cursor.alloc(1); hist.alloc(128);
  SRW_HIP(hipMemsetAsync(cursor.p, 0, 8, st));
  SRW_HIP(hipMemsetAsync(hist.p, 0, 128 * 8, st));
  hipLaunchKernelGGL((k_eb_hist<true>), dim3(h->n_cus * 8), dim3(TPB), 0, st, g.view(), ShardSel{h->cfg.rank, h->cfg.world}, sel, cursor.p, hist.p);
  SRW_HIP(hipGetLastError());
  unsigned long long hh[128];
  SRW_HIP(hipMemcpyAsync(hh, hist.p, sizeof(hh), hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  pl.units = pl.pairs = 0;
  for (int cls = 1; cls <= 62; ++cls) { pl.units += hh[cls * 2]; pl.pairs += hh[cls * 2 + 1]; }
  pl.munits = hh[126]; pl.mpairs = hh[127]; pl.inl = hh[1];
  const unsigned long long all = pl.pairs + pl.mpairs + pl.inl;
  if (all == 0 || pl.units >= 0xFFFFFFF0ull || pl.munits >= 0xFFFFFFF0ull) return false;
  const unsigned long long nb = all * 20 / 44 + 16;                 // 4 slots per bucket at load <= 0.55
  if (nb >= 0xFFFFFFF0ull) return false;
  pl.buckets = (uint32_t)nb;
  pl.bytes = (size_t)(pl.units * 64 + pl.munits * 16 + (pl.pairs + pl.mpairs) * 12 + nb * 64) + (size_t)g.n_slots * 24;
  return true;
}

void build_shard_edge_tables(srw_handle *h, float p, float q, int mode, int bins_cap, const ShardTabPlan &pl, const EbSel &sel) {
  Graph &g = h->g;
  hipStream_t st = h->stream;
  const auto t0 = std::chrono::steady_clock::now();
  const ShardSel ss{h->cfg.rank, h->cfg.world};
  const int blocks = h->n_cus * 8;
  g.ebp = sel.pol; g.eb_f32 = sel.pol.f32; g.eb_cap = bins_cap; g.eb_min_sh = sel.pol.min_sh; g.eb_mask_max = sel.mask_max;
  DevBuf<unsigned long long> cursor, hist, row_units, row_munits, row_pairs;
  cursor.alloc(1); hist.alloc(8);
  row_units.alloc((size_t)g.n_slots + 1); row_munits.alloc((size_t)g.n_slots + 1); row_pairs.alloc((size_t)g.n_slots + 1);
  SRW_HIP(hipMemsetAsync(cursor.p, 0, 8, st));
  hipLaunchKernelGGL((k_eb_rowsum<true>), dim3(blocks), dim3(TPB), 0, st, g.view(), ss, sel, cursor.p, row_units.p, row_munits.p, row_pairs.p);
  SRW_HIP(hipGetLastError());
  {
    size_t tb = 0;
    SRW_HIP(rocprim::exclusive_scan(nullptr, tb, row_units.p, row_units.p, 0ull, (size_t)g.n_slots, rocprim::plus<unsigned long long>(), st));
    DevBuf<char> temp; temp.alloc(tb);
    SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, row_units.p, row_units.p, 0ull, (size_t)g.n_slots, rocprim::plus<unsigned long long>(), st));
    SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, row_munits.p, row_munits.p, 0ull, (size_t)g.n_slots, rocprim::plus<unsigned long long>(), st));
    SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, row_pairs.p, row_pairs.p, 0ull, (size_t)g.n_slots, rocprim::plus<unsigned long long>(), st));
    SRW_HIP(hipStreamSynchronize(st));
  }
  const unsigned long long all_pairs = pl.pairs + pl.mpairs;
  g.eb_bins.alloc_progressive((size_t)pl.units * 8, h->cfg.device);      // (vm_buf.h: the pages arrive while the build fills them)
  g.em_bits.alloc((size_t)pl.munits * 4);
  g.ph.alloc((size_t)pl.buckets * 4); g.ph_buckets = pl.buckets;
  SRW_HIP(hipMemsetAsync(g.ph.p, 0xFF, (size_t)pl.buckets * 4 * sizeof(PairSlot), st));
  DevBuf<uint2> items; DevBuf<uint32_t> item_off;
  items.alloc((size_t)all_pairs); item_off.alloc((size_t)all_pairs);
  GraphView gv = g.view();                      // (has_eb still false: the builders never read the tables they are writing)
  SRW_HIP(hipMemsetAsync(cursor.p, 0, 8, st));
  hipLaunchKernelGGL((k_eb_assign<true>), dim3(blocks), dim3(TPB), 0, st, gv, ss, sel, cursor.p, row_units.p, row_munits.p, row_pairs.p,
                     (uint32_t *)nullptr, items.p, g.ph.p, g.ph_buckets, item_off.p);
  SRW_HIP(hipGetLastError());
  if (sel.mask_max > 0) {
    SRW_HIP(hipMemsetAsync(cursor.p, 0, 8, st));
    hipLaunchKernelGGL((k_eb_inline<true>), dim3(blocks), dim3(TPB), 0, st, gv, ss, sel, cursor.p, (uint32_t *)nullptr, g.ph.p, g.ph_buckets);
    SRW_HIP(hipGetLastError());
  }
  unsigned long long sc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // tables of more chunks than the build's LDS holds bins for: one f64 per chunk and wave in an HBM scratch
  DevBuf<double> gscratch;
  const int64_t gs_stride = sel.pol.fine_cap > BIN_CAP ? (int64_t)sel.pol.fine_cap : 0;
  if (gs_stride) gscratch.alloc((size_t)blocks * (TPB / 64) * (size_t)gs_stride);
  if (all_pairs) {
    const int fill_tune = getenv("SRW_EB_FILL_TUNE") ? atoi(getenv("SRW_EB_FILL_TUNE")) : 0;
    SRW_HIP(hipMemsetAsync(hist.p, 0, 8 * 8, st));
    hipStream_t aux = nullptr; hipEvent_t ev = nullptr;       // (two streams for the segments, as build_edge_tables)
    DevBuf<unsigned long long> cursor2; DevBuf<double> gscratch2;
    struct AuxGuard { hipStream_t &s; hipEvent_t &e; ~AuxGuard() { if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } if (e) (void)hipEventDestroy(e); } } aux_guard{aux, ev};
    int n_launches = 0;
    auto launch = [&](unsigned long long i0, unsigned long long n_it, int which = 0) {
      hipStream_t ls = which ? aux : st;
      unsigned long long *cur = which ? cursor2.p : cursor.p;
      SRW_HIP(hipMemsetAsync(cur, 0, 8, ls));
      hipLaunchKernelGGL((k_eb_build<true>), dim3(blocks), dim3(TPB), 0, ls, gv, items.p + i0, (int64_t)n_it, p, q, sel.mask_max,
                         sel.pol, item_off.p + i0, g.eb_bins.p, g.em_bits.p, cur, hist.p, fill_tune, which ? gscratch2.p : gscratch.p, gs_stride);
      SRW_HIP(hipGetLastError());
      ++n_launches;
    };
    if (!g.eb_bins.progressive()) launch(0ull, all_pairs);
    else {                                           // one launch per chunk of the table buffer (build_edge_tables, k_eb_segments)
      if (!getenv("SRW_EB_ONE_STREAM")) {
        cursor2.alloc(1);
        if (gs_stride) gscratch2.alloc((size_t)blocks * (TPB / 64) * (size_t)gs_stride);
        SRW_HIP(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
        SRW_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        SRW_HIP(hipEventRecord(ev, st));
        SRW_HIP(hipStreamWaitEvent(aux, ev, 0));
      }
      const unsigned long long chunk_units = (unsigned long long)(g.eb_bins.chunk_bytes() / 64);
      const int n_seg = (int)((pl.units + chunk_units - 1) / chunk_units);
      DevBuf<unsigned long long> d_seg; d_seg.alloc((size_t)2 * n_seg + 2);
      SRW_HIP(hipMemsetAsync(row_pairs.p + g.n_slots, 0xFF, 8, st));          // (never read: the search stops below n_slots)
      hipLaunchKernelGGL(k_eb_segments, dim3((unsigned)((n_seg + 1 + 255) / 256)), dim3(256), 0, st, row_units.p, row_pairs.p, (int64_t)g.n_slots, chunk_units,
                         n_seg, pl.units, all_pairs, d_seg.p, d_seg.p + n_seg + 1);
      SRW_HIP(hipGetLastError());
      std::vector<unsigned long long> seg((size_t)2 * n_seg + 2);
      SRW_HIP(hipMemcpyAsync(seg.data(), d_seg.p, seg.size() * 8, hipMemcpyDeviceToHost, st));
      SRW_HIP(hipStreamSynchronize(st));
      for (int k = 0; k < n_seg; ++k) {
        const unsigned long long i0 = seg[(size_t)k], i1 = seg[(size_t)k + 1];
        if (i1 <= i0) continue;
        g.eb_bins.wait_mapped((size_t)seg[(size_t)n_seg + 1 + k] * 64);
        launch(i0, i1 - i0, aux ? (n_launches & 1) : 0);
      }
      if (aux) { SRW_HIP(hipEventRecord(ev, aux)); SRW_HIP(hipStreamWaitEvent(st, ev, 0)); }
      g.eb_bins.wait_all();
    }
    SRW_HIP(hipMemcpyAsync(sc, hist.p, sizeof(sc), hipMemcpyDeviceToHost, st));
  } else g.eb_bins.wait_all();
  SRW_HIP(hipStreamSynchronize(st));
  if (sc[7]) throw Error(SRW_ERR_INVALID, "per-edge tables: a prefix sum of a ROW_PQ_F32 row was not exactly representable in binary32 (or a chunk mass not in 16 bits)");
  uint32_t pb, qb; memcpy(&pb, &p, 4); memcpy(&qb, &q, 4);
  g.eb_pbits = pb; g.eb_qbits = qb; g.eb_mode = mode;
  g.eb_tables = (int64_t)all_pairs;
  g.eb_bytes = (int64_t)(pl.units * 64 + pl.munits * 16 + (unsigned long long)pl.buckets * 64);
  g.eb_complete = sel.mask_max > 0;
  g.eb_sharded = true; g.has_eb = true; g.use_eb = true;
  g.eb_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (getenv("SRW_TIMING"))
    fprintf(stderr, "[shard %d/%d edge tables] up to %d chunks per table; %llu bins tables (%.2f GB; fill: P1 %llu, P2 %llu, W %llu, P3 %llu) + %llu masks (%.2f GB) "
            "+ %llu inline masks; pair hash %.2f GB; built in %.0f ms\n", h->cfg.rank, h->cfg.world, sel.pol.cap, pl.pairs, (double)pl.units * 64 / 1e9, sc[1], sc[2],
            sc[3], sc[4], pl.mpairs, (double)pl.munits * 16 / 1e9, pl.inl, (double)pl.buckets * 64 / 1e9, g.eb_build_ms);
}
}  // namespace

// Called at every super-step of a q != 1 Mode R walk (cheap once the tables stand).  Needs build_membership + build_pq_tables.
void prepare_shard_tables(srw_handle *h, const srw_walk_params &P) {
  Graph &g = h->g;
  const bool want = P.q != 1.0f && g.has_pq && g.has_member && g.n_entries > 0 && !(P.flags & (SRW_WALK_NO_EDGE_TABLES | SRW_WALK_NO_BINNED)) &&
                    ((P.flags >> 12) & 7) == 0 && !getenv("SRW_SHARD_NO_TABLES");
  if (!want) { g.use_eb = false; return; }
  const int mode = (P.flags & SRW_WALK_EDGE_TABLES_ALL) ? 1 : 0;
  uint32_t pb, qb; memcpy(&pb, &P.p, 4); memcpy(&qb, &P.q, 4);
  if (g.has_eb && g.eb_sharded && g.eb_pbits == pb && g.eb_qbits == qb && g.eb_mode == mode) { g.use_eb = true; return; }
  // new (p, q) or first use: drop what stands, then size the new set against what is free
  g.has_eb = false; g.use_eb = false; g.eb_complete = false; g.eb_tables = 0; g.eb_bytes = 0;
  g.eb_bins.release(); g.em_bits.release(); g.ph.release(); g.ph_buckets = 0; g.eb_off.release();
  build_unit_ids(h);                                         // unit-weight graphs: 4-byte ids for the table steps
  // 1. the finest complete set that fits
  size_t free_b = 0, total_b = 0;
  SRW_HIP(hipMemGetInfo(&free_b, &total_b));
  if (g.hub_bm.n) {                                          // rebuilt below with what the tables leave: released NOW, so that the credit is real
    free_b += g.hub_bm.n * sizeof(uint32_t);                // (build_hub_bitmaps returns early on an unchanged (min_deg, budget) — it must not find the old ones standing)
    g.hub_bm.release(); g.has_hub = false; g.n_hubs = 0; g.hub_budget_cap = 0; g.hub_min_deg = 0;
  }
  const size_t reserve = env_gb("SRW_EB_RESERVE_GB", 16);
  const char *env_cap = getenv("SRW_EB_CHUNKS");
  ShardTabPlan pl; EbSel sel; int cap_sel = 0;
  g.eb_min_sh_sel = 8; g.eb_cm_sel = 0; g.eb_fine_cap_sel = 0; g.eb_cm_ratio_sel = 0;
  for (int cap : {256, 128, 64, 32}) {
    if (env_cap && *env_cap) cap = std::min(std::max(atoi(env_cap), 8), BIN_CAP);
    sel = shard_sel(g, mode, cap);
    if (shard_plan(h, sel, pl) && pl.bytes * (size_t)std::max(1, h->dev_share) + reserve < free_b && pl.bytes < env_gb("SRW_EB_BUDGET_GB", 200)) { cap_sel = cap; break; }   // (virtual shards of one device share its HBM)
    if (env_cap && *env_cap) break;
  }
  if (!cap_sel) {
    if (getenv("SRW_TIMING")) fprintf(stderr, "[shard %d/%d edge tables] no complete set fits: on-the-fly samplers\n", h->cfg.rank, h->cfg.world);
    return;
  }
  // ... and, at that number of chunks, the smallest chunks (64, then 128 candidates instead of 256) whose set still leaves the
  // edge hash and 16 GB of hub bitmaps their room
  if (!mode && !getenv("SRW_EB_MIN_SH")) {
    uint64_t ehs = 1024;
    while (ehs < (uint64_t)g.n_entries_global + (uint64_t)g.n_entries_global / 2) ehs <<= 1;
    const size_t room = reserve + (g.has_ehash ? 0 : (size_t)ehs * 8) + ((size_t)24 << 30);
    for (int sh : {6, 7}) {
      g.eb_min_sh_sel = sh;
      const EbSel s2 = shard_sel(g, mode, cap_sel);
      ShardTabPlan p2;
      if (shard_plan(h, s2, p2) && p2.bytes * (size_t)std::max(1, h->dev_share) + room < free_b && p2.bytes < env_gb("SRW_EB_BUDGET_GB", 200)) { sel = s2; pl = p2; break; }   // (virtual shards of one device share its HBM)
      g.eb_min_sh_sel = 8;
    }
  }
  // ... then the chunk masks of the rows up to 16 384 / 4 096 candidates and the finer tables of the unmasked pairs with a long N(prev)
  // (sampling.h:eb_pair_geometry), under the same condition
  if (!mode && g.eb_min_sh_sel >= 6) {
    uint64_t ehs = 1024;
    while (ehs < (uint64_t)g.n_entries_global + (uint64_t)g.n_entries_global / 2) ehs <<= 1;
    const size_t room = reserve + (g.has_ehash ? 0 : (size_t)ehs * 8) + ((size_t)24 << 30);
    auto fits = [&](EbSel &s2, ShardTabPlan &p2) {
      s2 = shard_sel(g, mode, cap_sel);
      return shard_plan(h, s2, p2) && p2.bytes * (size_t)std::max(1, h->dev_share) + room < free_b && p2.bytes < env_gb("SRW_EB_BUDGET_GB", 230);
    };
    if (!getenv("SRW_EB_CM_MAX"))
      for (int cm : {16384, 4096}) {
        g.eb_cm_sel = cm;
        EbSel s2; ShardTabPlan p2;
        if (fits(s2, p2)) { sel = s2; pl = p2; break; }
        g.eb_cm_sel = 0;
      }
    if (g.eb_cm_sel && !getenv("SRW_EB_CM_RATIO"))
      for (int ratio : {16, 4}) {
        g.eb_cm_ratio_sel = ratio;
        EbSel s2; ShardTabPlan p2;
        if (fits(s2, p2)) { sel = s2; pl = p2; break; }
        g.eb_cm_ratio_sel = 0;
      }
    // (finer tables beyond 512 chunks only for a job long enough to pay for their build: walk_kernels.hip:prepare_tables, srw_plan_walks)
    if (!getenv("SRW_EB_FINE_CAP"))
      for (int fc : {4096, 1024, 512}) {
        if (fc <= cap_sel) break;
        if (fc > BIN_CAP && std::max<int64_t>(h->planned_walks > 0 ? h->planned_walks : 10, P.num_walks) < 64) continue;   // (beyond the LDS bins: long jobs only)
        g.eb_fine_cap_sel = fc;
        EbSel s2; ShardTabPlan p2;
        if (fits(s2, p2)) { sel = s2; pl = p2; break; }
        g.eb_fine_cap_sel = 0;
      }
  }
  // 2. the located chunk's probes of a long N(prev), with what the tables leave (a shard of a sharded graph has room: its
  //    tables are 1 / world of the set): the edge hash set of the whole graph (one probe per candidate) or, when it does not
  //    fit, the long rows' neighbor-set filters in front of the sorted rows; neighbor-set bitmaps of the hubs take the rest
  size_t left = (free_b - pl.bytes - reserve) / (size_t)std::max(1, h->dev_share);
  uint64_t eh_slots = 1024;
  while (eh_slots < (uint64_t)g.n_entries_global + (uint64_t)g.n_entries_global / 2) eh_slots <<= 1;
  bool ehash = !(P.flags & SRW_WALK_NO_EDGE_HASH) && !getenv("SRW_EB_DROP_EHASH") && (g.has_ehash || left > eh_slots * 8 + ((size_t)8 << 30));
  auto optional = [&](const char *what, auto &&build) {      // an optional accelerator never fails a walk: another (virtual) shard may have taken the room since hipMemGetInfo
    try { build(); }
    catch (const Error &e) {
      if (e.code != SRW_ERR_NOMEM) throw;
      (void)hipGetLastError();
      if (getenv("SRW_TIMING")) fprintf(stderr, "[shard %d/%d edge tables] %s skipped: %s\n", h->cfg.rank, h->cfg.world, what, e.what());
    }
  };
  if (ehash && !g.has_ehash) { optional("edge hash", [&] { build_edge_hash(h); }); if (g.has_ehash) left -= eh_slots * 8; }
  ehash = ehash && g.has_ehash;
  if (!ehash && g.has_ehash) { g.ehash.release(); g.has_ehash = false; }
  g.use_ehash = ehash;
  if (!ehash && !getenv("SRW_NO_ROW_FILTERS")) optional("row filters", [&] { build_row_filters(h); });
  const bool hubs = !(P.flags & SRW_WALK_NO_HUB_BITMAPS);
  if (hubs) {
    size_t hub_cap = std::min<size_t>((size_t)96 << 30, left > ((size_t)4 << 30) ? left - ((size_t)4 << 30) : 0);
    if (const char *e = getenv("SRW_HUB_BUDGET_GB"); e && *e) hub_cap = (size_t)(atof(e) * (double)((size_t)1 << 30));
    optional("hub bitmaps", [&] { build_hub_bitmaps(h, ((P.flags >> 15) & 1) ? 1 : 1024, hub_cap); });
  }
  g.use_hub = hubs && g.has_hub;
  // 3. the tables (an optional accelerator never fails a walk)
  for (int attempt = 0; attempt < 2; ++attempt) {
    try { build_shard_edge_tables(h, P.p, P.q, mode, cap_sel, pl, sel); return; }
    catch (const Error &e) {
      const bool remap = e.code == SRW_ERR_HIP && vm_buf_broken().load() && attempt == 0 && std::string(e.what()).find("mapping the table buffer") != std::string::npos;
      if (e.code != SRW_ERR_NOMEM && !remap) throw;
      (void)hipGetLastError();
      (void)hipStreamSynchronize(h->stream);          // (segments of the build may be running over the chunks that did get mapped)
      g.eb_bins.release(); g.em_bits.release(); g.ph.release(); g.ph_buckets = 0; g.has_eb = false;
      if (remap) { attempt = -1; continue; }          // (vm_buf.h: a mapping call refused for another reason than memory — once more with one hipMalloc)
      if (attempt || cap_sel <= 32) break;
      cap_sel = 32; g.eb_min_sh_sel = 8; sel = shard_sel(g, mode, 32);
      if (!shard_plan(h, sel, pl)) break;
    }
  }
  if (getenv("SRW_TIMING")) fprintf(stderr, "[shard %d/%d edge tables] no complete set fits: on-the-fly samplers\n", h->cfg.rank, h->cfg.world);
}

// ---- return edges on a vertex-sharded handle (k_sh_step_q1: p != 1, q == 1) ------------------------------------------------
// rev[e] belongs to the entry e = (u -> x) of u's row and describes x's row: on a shard the two live on different ranks.  The
// record is therefore keyed by the pair, on owner(x), and comes straight from x's own sorted row: every distinct id u in it
// is a pair (u -> x) with a return edge — first sorted index, multiplicity, input-order position of the first occurrence.
namespace {
__global__ __launch_bounds__(TPB) void k_rev_hash(GraphView g, unsigned long long *cursor, unsigned long long *count, PairSlot *__restrict__ rh,
                                                  uint32_t rh_buckets) {
  const int lane = lane_id();
  unsigned long long n = 0;
  while (true) {
    const int64_t v0 = grab_u64(cursor, GRAB_SLOTS);
    if (v0 >= g.n_slots) break;
    for (int64_t x = v0; x < v0 + GRAB_SLOTS && x < g.n_slots; ++x) {
      const Row rx = g.rows[x];
      const uint32_t *cs = g.sids + rx.off;
      for (int32_t i = lane; i < rx.deg; i += 64) {
        const uint32_t u = cs[i];
        if (i > 0 && cs[i - 1] == u) continue;                       // the first occurrence speaks for the run
        ++n;
        if (rh) {
          uint32_t cnt = 1;
          while (cnt < 255u && i + (int32_t)cnt < rx.deg && cs[i + cnt] == u) ++cnt;
          pair_insert(rh, rh_buckets, u, (uint32_t)x, (cnt << 24) | (uint32_t)i, g.sperm[rx.off + i]);
        }
      }
    }
  }
  n = wave_sum_u64(n);
  if (lane == 0 && n && count) atomicAdd(count, n);
}
}  // namespace

void build_shard_rev_hash(srw_handle *h) {
  Graph &g = h->g;
  if (g.has_rh) return;
  build_membership(h);
  hipStream_t st = h->stream;
  DevBuf<unsigned long long> cur; cur.alloc(2);
  SRW_HIP(hipMemsetAsync(cur.p, 0, 16, st));
  hipLaunchKernelGGL(k_rev_hash, dim3(h->n_cus * 8), dim3(TPB), 0, st, g.view(), cur.p, cur.p + 1, (PairSlot *)nullptr, 0u);
  unsigned long long n = 0;
  SRW_HIP(hipMemcpyAsync(&n, cur.p + 1, 8, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  SRW_HIP(hipGetLastError());
  const unsigned long long nb = n * 20 / 44 + 16;
  if (nb >= 0xFFFFFFF0ull) throw Error(SRW_ERR_NOMEM, "return-edge hash: too many pairs");
  g.rh.alloc((size_t)nb * 4); g.rh_buckets = (uint32_t)nb;
  SRW_HIP(hipMemsetAsync(g.rh.p, 0xFF, (size_t)nb * 4 * sizeof(PairSlot), st));
  SRW_HIP(hipMemsetAsync(cur.p, 0, 16, st));
  hipLaunchKernelGGL(k_rev_hash, dim3(h->n_cus * 8), dim3(TPB), 0, st, g.view(), cur.p, (unsigned long long *)nullptr, g.rh.p, g.rh_buckets);
  SRW_HIP(hipGetLastError());
  SRW_HIP(hipStreamSynchronize(st));
  g.has_rh = true;
}

void build_rev_table(srw_handle *h) {
  Graph &g = h->g;
  if (g.has_rev) return;
  build_membership(h);
  hipStream_t st = h->stream;
  g.rev.alloc((size_t)std::max<int64_t>(g.n_entries, 1));
  DevBuf<unsigned long long> cursor; cursor.alloc(1);
  SRW_HIP(hipMemsetAsync(cursor.p, 0, 8, st));
  hipLaunchKernelGGL(k_rev_build, dim3(h->n_cus * 8), dim3(TPB), 0, st, g.view(), cursor.p, g.rev.p);
  SRW_HIP(hipGetLastError());
  SRW_HIP(hipStreamSynchronize(st));
  g.has_rev = true;
}

}  // namespace srw

// Host-only hook (include/stellar_rw.h): the chunk geometry the planner, the builder and the walk kernels share (sampling.h:
// eb_pair_geometry), for the CPU test that holds its closed form against the definition.
extern "C" int32_t srw_table_geometry(int32_t deg_curr, int32_t deg_prev, const int32_t *policy, int32_t *chunk_shift, int32_t *n_chunks,
                                      int32_t *masked) {
  if (!policy || !chunk_shift || !n_chunks || !masked || deg_curr < 1 || deg_prev < 0 || policy[1] < 1 || policy[0] < 0 || policy[0] > 30) return SRW_ERR_INVALID;
  srw::EbPolicy P;
  P.min_sh = policy[0]; P.cap = policy[1]; P.cm_max = policy[2]; P.cm_min_du = policy[3]; P.fine_min_du = policy[4]; P.fine_sh = policy[5];
  P.fine_cap = policy[6]; P.f32 = policy[7]; P.u16 = policy[8]; P.cm_ratio = policy[9];
  const srw::PairGeom g = srw::eb_pair_geometry(deg_curr, deg_prev, P);
  *chunk_shift = g.csh; *n_chunks = g.n_bins; *masked = g.cmask ? 1 : 0;
  return SRW_OK;
}
