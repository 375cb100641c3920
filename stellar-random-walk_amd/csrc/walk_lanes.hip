// walk_lanes.hip — the bit-exact second-order walk over the per-edge tables with ONE WALKER PER LANE.
// Replaces the inner loop of RandomWalk.randomWalk (M/algorithm/RandomWalk.scala:95-139) + RandomSample.secondOrderSample
// (M/algorithm/RandomSample.scala:27-62) for q != 1 on a whole-graph handle whose (prev -> curr) pairs all have a table / mask
// (edge_tables.hip).  Same contract as walk_kernels.hip:k_walk_tables: persistent waves take walkers from a cursor; a walker that
// meets a pair without a table, a row without a certificate or a draw within rounding distance of a CDF boundary is handed over
// untouched (its index goes to `todo`, a boundary draw on a table step also to the tie list of the chain kernels) and
// k_walk_general redoes it from its first step — the keyed RNG makes that the same path.
//
// Why lanes (profiles/r06_group_kernel.md, r06_lane_kernel.md): the vector work of a step is per ITEM — a block of 64 chunk prefixes or
// candidates evaluated by 64 lanes (or by 16 lanes, four each) costs 64 x (convert, bias, two certified compares) + a scan network,
// whatever sits behind the crossing included.  A lane that walks ITS block sequentially adds the exact values up in order, tests only
// what comes within rounding distance of the draw and stops at the crossing; the walker's bookkeeping (row fetch, geometry, draw,
// path) is one vector instruction for 64 walkers.  The arithmetic is the wave samplers' (sampling.h: exact numerators under the row
// certificate ROW_PQ_OK, divide-free certain-miss / certain-hit compares), so the picks are the same bit for bit.
// Steps a lane cannot take cheaply — first steps (the raw row's sum), rows of more than 64 candidates, located chunks of more than one
// round, a short N(prev) that wants staging in LDS, row filters, the edge hash — are SERVED: the wave takes them one at a time with the
// one-walker-per-wave samplers (wave_pick_first / wave_pick_masked / wave_pick_edge_table), the owner lane keeps the result.
#include <algorithm>

#include "group_sampling.h"
#include "walk_records.h"

namespace srw {
namespace {

// waves per SIMD: the kernel is bound by the latency of its served steps (one chain of dependent round trips per wave at a time), so
// occupancy pays even through spills — config 3, tables per lane: 4 waves (128 VGPRs) 662 ms, 5 (96) 607 ms, 6 (80 + 244 B of scratch) 587 ms
// per iteration (profiles/r06_lane_kernel.md)
#ifndef SRW_LANE_WAVES
#define SRW_LANE_WAVES 6
#endif
#define LTAB_ARGS() fresh_args<LaneArgs>()
struct LaneArgs { TabArgs t; int32_t mode, max_csh; }; // mode: bit 0 mask rows per lane, bit 1 table steps per lane (0: every step served); max_csh: located chunks of up to 2^max_csh candidates per lane

enum : int32_t { K_NONE = 0, K_SERVE_FIRST = 1, K_SERVE_MASK = 2, K_SERVE_TABLE = 3, K_LANE_ROW = 4, K_LANE_TABLE = 5 };
#ifndef SRW_LANE_NB
#define SRW_LANE_NB 2                 // 16-byte loads in flight per array and trip of a lane's candidate loops (x 4 candidates)
#endif
constexpr int LNB = SRW_LANE_NB;
constexpr int32_t LANE_ROW_MAX = 255;                 // whole rows a lane walks itself (two passes): every row with a membership mask

// A lane's loops are chains of dependent round trips unless the loads of a trip are issued together: candidates are taken NB x 4 at a
// time (NB 16-byte loads per array in flight), whole small levels at once.
// A run of candidates [k0, k1] of a row appended, in order, to the exact numerator `acc`: the first candidate that is not a certain
// miss (sampling.h:binned_resolve's compares).  members: bit ((k - kw) & 63) of the word of position k says "candidate k is in N(prev)";
// mw: the words of positions kw, kw + 64, ... (kw a multiple of 64 at or before k0; the run covers at most four words); hub: prev's
// neighbor-set bitmap to probe instead (or null).  Returns its position and sets id_out / hit, or -1 (acc then holds the whole run).
template <int NB>
__device__ inline int32_t lane_scan(const GraphView &g, int64_t roff, int32_t k0, int32_t k1, int32_t kw, const unsigned long long (&mw)[4],
                                    bool members, const uint32_t *hub, bool returns, bool biased, int32_t prev, const BiasDiv &bdiv,
                                    double &acc, double pS, int32_t &id_out, bool &hit) {
  // num < lo  =>  fl(num (1 + t_k)) < pS for every k <= k1: a certain miss without the two products
  const double lo = pS * (1.0 - ((double)(k1 + 8) * 0x1p-50 + 0x1p-50));
  int32_t res = -1;
  for (int32_t kb = k0; kb <= k1 && res < 0; kb += 4 * NB) {
    g16::Cand4 c[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) g16::load_cand4(g, roff, kb + 4 * b, k1, c[b]);
    uint32_t hb[NB][4];
    if (hub) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t) { hb[b][t] = 0u; if (c[b].valid[t] && c[b].id[t] != prev) hb[b][t] = hub[(uint32_t)((int64_t)c[b].id[t] - g.vmin) >> 5]; }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int32_t kq = kb + 4 * b;
      if (res < 0 && kq <= k1) {
        const int wsel = (kq - kw) >> 6;
        const unsigned long long word = wsel == 0 ? mw[0] : wsel == 1 ? mw[1] : wsel == 2 ? mw[2] : mw[3];
        float w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool is_prev = returns && c[b].id[t] == prev;
          bool in = false;
          if (hub) in = !is_prev && ((hb[b][t] >> ((uint32_t)((int64_t)c[b].id[t] - g.vmin) & 31u)) & 1u);
          else if (members) in = !is_prev && ((word >> ((kq + t - kw) & 63)) & 1ull);
          w[t] = !c[b].valid[t] ? 0.0f : biased ? bdiv(c[b].w[t], is_prev, in) : c[b].w[t];
        }
        // (the four together first: a block that stays below `lo` is four certain misses; one exit, every index a constant — a `return`
        //  per candidate is merged by the compiler into one with a variable index, and the candidates go through scratch)
        const double a0 = acc + (double)w[0], a1 = a0 + (double)w[1], a2 = a1 + (double)w[2], a3 = a2 + (double)w[3];
        if (!(a3 < lo)) {
          if (c[b].valid[3] && g16::not_miss(kq + 3, a3, pS)) { res = kq + 3; id_out = c[b].id[3]; hit = g16::sure_hit(kq + 3, a3, pS); }
          if (c[b].valid[2] && g16::not_miss(kq + 2, a2, pS)) { res = kq + 2; id_out = c[b].id[2]; hit = g16::sure_hit(kq + 2, a2, pS); }
          if (c[b].valid[1] && g16::not_miss(kq + 1, a1, pS)) { res = kq + 1; id_out = c[b].id[1]; hit = g16::sure_hit(kq + 1, a1, pS); }
          if (c[b].valid[0] && g16::not_miss(kq + 0, a0, pS)) { res = kq + 0; id_out = c[b].id[0]; hit = g16::sure_hit(kq + 0, a0, pS); }
        }
        acc = a3;
      }
    }
  }
  return res;
}

// A whole row of at most 255 candidates that holds the certificate (ROW_PQ_OK: every sum of variants is exact in any order): the first
// step of a walk (raw weights, RandomWalk.scala:51-66) or a second-order step whose pair has a membership mask — wave_pick_first /
// wave_pick_masked, sequentially: S in one pass, the scan in a second.  k >= 0, or CHAIN_NEEDED.
__device__ inline int32_t lane_pick_row(const GraphView &g, const Row &r, bool second, int32_t prev, const BiasDiv &bdiv, uint32_t eo, float u, int32_t &id_out) {
  unsigned long long mw[4] = {eo, 0ull, 0ull, 0ull};
  if (second && r.deg > 32) {
    const uint2 *wp = reinterpret_cast<const uint2 *>(g.em_bits + (size_t)eo * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i * 64 < r.deg) { const uint2 q = wp[i]; mw[i] = ((unsigned long long)q.y << 32) | q.x; }
  }
  double S = 0.0;
  if (!second && g.ids32) S = (double)r.deg;         // unit weights: nothing to read
  else {
    for (int32_t kb = 0; kb < r.deg; kb += 4 * LNB) {
      g16::Cand4 c[LNB];
#pragma unroll
      for (int b = 0; b < LNB; ++b) g16::load_cand4(g, r.off, kb + 4 * b, r.deg - 1, c[b]);
#pragma unroll
      for (int b = 0; b < LNB; ++b) {
        const int32_t kq = kb + 4 * b;
        const int wsel = kq >> 6;
        const unsigned long long word = wsel == 0 ? mw[0] : wsel == 1 ? mw[1] : wsel == 2 ? mw[2] : mw[3];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool is_prev = second && c[b].id[t] == prev;
          const float x = second ? bdiv(c[b].w[t], is_prev, !is_prev && ((word >> ((kq + t) & 63)) & 1ull)) : c[b].w[t];
          S += c[b].valid[t] ? (double)x : 0.0;
        }
      }
    }
  }
  if (!(S > 0.0)) return CHAIN_NEEDED;                // (S = 0: the reference divides by zero -> chain)
  const double pS = (double)u * S;
  double acc = 0.0;
  bool hit = false;
  const int32_t k = lane_scan<LNB>(g, r.off, 0, r.deg - 1, 0, mw, second, nullptr, second, second, prev, bdiv, acc, pS, id_out, hit);
  if (k < 0) { id_out = load_ent(g, g.ent + r.off, 0).id; return 0; }      // edges.head (:24)
  return hit ? k : CHAIN_NEEDED;
}

// A second-order step through the pair's table: binned_resolve<ABS>, sequentially — per level the block's absolute prefixes (a short
// block: all of it in one round trip; else a binary search: the compare is monotone, prefixes and tolerances both grow) or a run
// over the 16-bit deltas, 32 at a time, then the located chunk.
// k >= 0; -1 (non-positive sum); CHAIN_NEEDED (S_out = the row's sum); LANE_SERVE: the wave takes the step.
constexpr int32_t LANE_SERVE = -4;
__device__ inline int32_t lane_pick_table(const GraphView &g, const Row &rc, int32_t prev, const Row &rprev, float p_, float q_, uint32_t eo,
                                          float u, int max_csh, int32_t &id_out, double &S_out, uint32_t &res_bytes) {
  const int32_t deg = rc.deg, m = rprev.deg;
  const uint32_t rflags = rc.flags;
  const PairGeom pg = eb_pair_geometry(deg, m, g.ebp);
  const int csh = pg.csh;
  if (csh > max_csh) return LANE_SERVE;              // longer located chunks: the wave's rounds
  const uint32_t prev_hub = rprev.flags >> ROW_HUB_SHIFT;
  const uint32_t *hubbits = (prev_hub && g.hub_bm) ? g.hub_bm + (int64_t)(prev_hub - 1) * g.hub_words : nullptr;
  const bool one_sign = (q_ > 1.0f && p_ <= q_) || (q_ < 1.0f && p_ >= q_);
  if (!pg.cmask && !hubbits && !one_sign) return LANE_SERVE;
  const int32_t n_bins = pg.n_bins;
  const bool f32t = g.ebp.f32 && (rflags & ROW_PQ_F32);
  const bool u16t = eb_pair_u16(rflags, csh, g.ebp);
  const double unit = u16t ? eb_row_unit(rflags) : 0.0;
  const EbLayout lay = eb_layout(f32t, n_bins, pg.cmask, deg, u16t);
  const char *table = reinterpret_cast<const char *>(g.eb_bins) + (size_t)eo * 64;
  auto chunk_end = [&](int32_t j) { const int64_t e = (((int64_t)j + 1) << csh) - 1; return (int32_t)(e < deg ? e : deg - 1); };
  const int nlev = lay.n2 ? 3 : lay.n1 ? 2 : 1;
  double S = 0.0, pS = 0.0, prev_val = 0.0, b_prev = 0.0, b_this = 0.0;
  int32_t blk = 0, jc = 0;
  for (int it = 0; it < nlev; ++it) {
    const int L = nlev - 1 - it;
    const uint32_t off = L == 2 ? lay.l2_off : L == 1 ? lay.l1_off : lay.l0_off;
    const int32_t cnt = L == 2 ? lay.n2 : L == 1 ? lay.n1 : n_bins;
    const char *lvl = table + (size_t)off * 64;
    const int32_t i_first = blk * 64;
    const int32_t n_here = cnt - i_first < 64 ? cnt - i_first : 64;
    if (L == 0 && u16t) {
      const unsigned short *d = reinterpret_cast<const unsigned short *>(lvl) + i_first;
      if (it == 0) {                                 // the only level: S = the sum of all the deltas (cnt <= 64; the level is padded to 64 bytes)
        uint32_t tot = 0u;
        for (int32_t i = 0; i < n_here; i += 32) {
          U32x4 q[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) q[b] = *reinterpret_cast<const U32x4 *>(d + i + 8 * b);
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const uint32_t x[4] = {q[b].a, q[b].b, q[b].c, q[b].d};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int32_t e = i + 8 * b + 2 * t;
              tot += (e < n_here ? x[t] & 0xFFFFu : 0u) + (e + 1 < n_here ? x[t] >> 16 : 0u);
            }
          }
        }
        S = (double)tot * unit;
        if (!(S > 0.0)) { S_out = S; return -1; }
        if (g.dbg_chain_deg && deg >= g.dbg_chain_deg) { S_out = S; return CHAIN_NEEDED; }
        pS = (double)u * S;
      }
      // (every partial sum is an exact multiple of 2^G under the row certificate: integer units, one conversion per test; a pair of
      //  deltas that stays below the threshold in units is two certain misses)
      const double lo = pS * (1.0 - ((double)(deg + 8) * 0x1p-50 + 0x1p-50));
      const double thr_d = (lo - prev_val) / unit - 2.0;
      const uint32_t thr = thr_d > 0.0 ? (thr_d < 4294967040.0 ? (uint32_t)thr_d : 0xFFFFFF00u) : 0u;
      uint32_t au = 0u;
      int32_t found = -1;
      for (int32_t i = 0; i < n_here && found < 0; i += 32) {
        U32x4 q[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) q[b] = *reinterpret_cast<const U32x4 *>(d + i + 8 * b);      // (64 bytes of a level padded to 64 bytes)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint32_t x[4] = {q[b].a, q[b].b, q[b].c, q[b].d};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int32_t e = i + 8 * b + 2 * t;
            if (found >= 0 || e >= n_here) break;
            const uint32_t d0 = x[t] & 0xFFFFu, d1 = e + 1 < n_here ? x[t] >> 16 : 0u;
            if (au + d0 + d1 < thr) { au += d0 + d1; continue; }
            const uint32_t a0 = au + d0, a1 = a0 + d1;
            const double v0 = prev_val + (double)a0 * unit, v1 = prev_val + (double)a1 * unit, vb = prev_val + (double)au * unit;
            if (g16::not_miss(chunk_end(i_first + e), v0, pS)) { found = e; b_this = v0; b_prev = vb; }
            else if (e + 1 < n_here && g16::not_miss(chunk_end(i_first + e + 1), v1, pS)) { found = e + 1; b_this = v1; b_prev = v0; }
            au = a1;
          }
        }
      }
      if (found < 0) { id_out = load_ent(g, g.ent + rc.off, 0).id; S_out = S; return 0; }    // even the last candidate is a certain miss -> edges.head
      jc = i_first + found;
    } else {
      auto value = [&](int32_t i) { return f32t ? (double)reinterpret_cast<const float *>(lvl)[i_first + i] : reinterpret_cast<const double *>(lvl)[i_first + i]; };
      auto elem_miss = [&](int32_t i, double v) {
        const int64_t je = (((int64_t)(i_first + i) + 1) << (6 * L)) - 1;      // the chunk this element is the prefix of
        const int32_t j = (int32_t)(je < n_bins ? je : n_bins - 1);
        return !g16::not_miss(chunk_end(j), v, pS);
      };
      int32_t lo_i = 0, hi_i = n_here;               // first element that is not a certain miss
      double v_lo = prev_val, v_hi = 0.0;            // value just before lo_i, value at hi_i
      if (n_here <= 8) {                             // a short block (the upper levels of tables of up to 512 chunks): all of it at once
        double v[8];
        if (f32t) {
          const float4 q0 = *reinterpret_cast<const float4 *>(lvl + (size_t)i_first * 4), q1 = *reinterpret_cast<const float4 *>(lvl + (size_t)i_first * 4 + 16);
          v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;      // (a level is padded to 64 bytes)
        } else {
          const double *dp = reinterpret_cast<const double *>(lvl) + i_first;
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = dp[i];
        }
        if (it == 0) {
          const int e = cnt - 1;
          S = e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : e == 3 ? v[3] : e == 4 ? v[4] : e == 5 ? v[5] : e == 6 ? v[6] : v[7];
          if (!(S > 0.0)) { S_out = S; return -1; }
          if (g.dbg_chain_deg && deg >= g.dbg_chain_deg) { S_out = S; return CHAIN_NEEDED; }
          pS = (double)u * S;
        }
        lo_i = n_here;
#pragma unroll
        for (int i = 7; i >= 0; --i)
          if (i < n_here && !elem_miss(i, v[i])) { lo_i = i; v_hi = v[i]; v_lo = i ? v[i - 1] : prev_val; }
      } else {
        if (it == 0) {                               // (the top level has at most 64 elements)
          S = value(cnt - 1);
          if (!(S > 0.0)) { S_out = S; return -1; }
          if (g.dbg_chain_deg && deg >= g.dbg_chain_deg) { S_out = S; return CHAIN_NEEDED; }
          pS = (double)u * S;
        }
        while (lo_i < hi_i) {
          const int32_t mid = (lo_i + hi_i) >> 1;
          const double v = value(mid);
          if (!elem_miss(mid, v)) { hi_i = mid; v_hi = v; } else { lo_i = mid + 1; v_lo = v; }
        }
      }
      if (lo_i >= n_here) { id_out = load_ent(g, g.ent + rc.off, 0).id; S_out = S; return 0; }
      if (L == 0) { jc = i_first + lo_i; b_this = v_hi; b_prev = v_lo; }
      else { blk = i_first + lo_i; prev_val = v_lo; }
    }
  }
  S_out = S;
  // ---- the located chunk (at most 256 candidates: four mask words)
  const int32_t k0 = (int32_t)((int64_t)jc << csh), k1 = chunk_end(jc);
  const BiasDiv bdiv(p_, q_);
  unsigned long long mw[4] = {0ull, 0ull, 0ull, 0ull};
  bool members = false, returns = true;
  const uint32_t *hub = nullptr;
  if (pg.cmask) {
    const unsigned long long *cm = reinterpret_cast<const unsigned long long *>(table + (size_t)lay.cm_off * 64) + (k0 >> 6);
#pragma unroll
    for (int i = 0; i < 4; ++i) if (k0 + 64 * i <= k1) mw[i] = cm[i];
    members = true;
  } else {
    bool no_specials = false;
    if (one_sign) {                                  // "the chunk's corrections sum to exactly 0" means "no special in the chunk"
      const PqRow PQ(g, rc.off);
      no_specials = (b_this - b_prev) - (PQ[k1] - (k0 ? PQ[k0 - 1] : 0.0)) == 0.0;
    }
    if (no_specials) returns = false;
    else if (hubbits) hub = hubbits;
    else return LANE_SERVE;                          // a short N(prev) (LDS staging), the row filters, the edge hash: the wave
  }
  double acc = b_prev;                               // A'_{k0-1}
  bool hit = false;
  const int32_t k = hub ? lane_scan<2>(g, rc.off, k0, k1, k0, mw, members, hub, returns, true, prev, bdiv, acc, pS, id_out, hit)
                        : lane_scan<LNB>(g, rc.off, k0, k1, k0, mw, members, hub, returns, true, prev, bdiv, acc, pS, id_out, hit);
  {                                                  // (counted as the wave kernel counts it: the rounds of 64 up to the one that decided)
    const int32_t rounds = (((k >= 0 ? k : k1) - k0) >> 6) + 1, all = k1 - k0 + 1;
    res_bytes += 8u * (uint32_t)(all < 64 * rounds ? all : 64 * rounds);
  }
  if (k < 0 || !hit) return CHAIN_NEEDED;
  return k;
}

// The wave's totals live in LDS (ds_add): eight counters less in every lane's registers — the kernel's occupancy is set by its VGPRs
// (the wave samplers of the served steps need ~64 next to the lanes' own state).
enum { T_STEPS = 0, T_SRCH, T_DEAD, T_TAB, T_MASK, T_FIRST, T_SV_FIRST, T_SV_MASK, T_SV_TAB, T_N };

template <bool BF>
__global__ __launch_bounds__(TPB, SRW_LANE_WAVES) void k_walk_tables_lanes(LaneArgs a0) {
  __shared__ __attribute__((aligned(16))) uint32_t stage_all[TPB / 64][1024];
  __shared__ unsigned long long tot_all[TPB / 64][T_N];
  const int lane = lane_id();
  uint32_t *stage = stage_all[threadIdx.x >> 6];
  unsigned long long *tot = tot_all[threadIdx.x >> 6];
  if (lane < T_N) tot[lane] = 0ull;
  __builtin_amdgcn_wave_barrier();
  Member mem; mem.mode = 0; mem.bm = stage; mem.seg_base = 0;
  const int32_t L = a0.t.L, mode = a0.mode;
  const int64_t stride = (int64_t)L + 2;
  // the lane's walker: what is not here is recomputed where it is needed (the iteration and the source from wi, the length from s)
  bool active = false, exhausted = false;
  int64_t wi = 0, eprev = 0;
  int32_t s = 1, prev = 0, curr = 0;
  uint32_t iter = 0, ksrc = 0;
  Row rprev; rprev.off = 0; rprev.deg = 0; rprev.flags = 0;
  int32_t pb0 = -1, pb1 = -1, pb2 = -1, pb3 = -1;    // the path slots (s & ~3) .. (s | 3) of the walker: one 16-byte store per four steps
  uint32_t w_tab = 0, w_mask = 0, w_srch = 0;        // (a handed-over walker is not counted)
  while (true) {
    // ---- lanes without a walker take the next ones from the cursor (one atomic per wave)
    {
      const unsigned long long need = __ballot(!active && !exhausted);
      if (need) {
        const LaneArgs aw = LTAB_ARGS();
        unsigned long long grab = 0;
        if (lane == 0) grab = atomicAdd(aw.t.cursor, (unsigned long long)__popcll(need));
        const int64_t w0 = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab));
        if (!active && !exhausted) {
          wi = w0 + (int64_t)__popcll(need & ((1ull << lane) - 1ull));
          if (wi >= aw.t.n_walkers) exhausted = true;
          else {
            const int64_t it = wi / aw.t.n_verts, vi = wi - it * aw.t.n_verts;
            iter = (uint32_t)(aw.t.first_walk + it);
            const int32_t src = aw.t.verts[vi];
            ksrc = (uint32_t)rng_source(aw.t.g, src);
            s = 1; prev = src; curr = src; eprev = 0;
            rprev.off = 0; rprev.deg = 0; rprev.flags = 0;
            pb0 = src; pb1 = -1; pb2 = -1; pb3 = -1;
            w_tab = 0; w_mask = 0; w_srch = 0;
            active = true;
          }
        }
      }
    }
    if (!__ballot(active)) break;
    // ---- the row of curr, the pair word, the draw; what kind of step this is
    int32_t kind = K_NONE, k = -1, next = 0;
    bool finish = false;
    Row r; r.off = 0; r.deg = 0; r.flags = 0;
    uint32_t eo = EB_NONE;
    float u = 0.0f;
    double S_tie = 0.0;
    const bool second = s > 1;
    if (active) {
      const LaneArgs as = LTAB_ARGS();
      const GraphView &gs = as.t.g;
      const int64_t cslot = (int64_t)curr - gs.vmin;
      const bool in_range = cslot >= 0 && cslot < gs.n_slots;
      r = gs.rows[in_range ? cslot : 0];
      if (second) eo = gs.eb_off[eprev];
      if (!in_range) { r.off = 0; r.deg = 0; r.flags = 0; }
      if (r.deg == 0) { if (second) atomicAdd(&tot[T_DEAD], 1ull); finish = true; }
      else {
        u = draw_uniform(as.t.rng, iter, ksrc, (uint32_t)s);
        const bool lane_row = (mode & 1) && r.deg <= LANE_ROW_MAX && (r.flags & ROW_PQ_OK);
        if (!second) kind = (lane_row && r.deg < (1 << 29)) ? K_LANE_ROW : K_SERVE_FIRST;
        else if (r.deg <= gs.eb_mask_max && (r.deg <= 32 || eo != EB_NONE)) kind = lane_row ? K_LANE_ROW : K_SERVE_MASK;
        else if (r.deg > gs.eb_mask_max && eo != EB_NONE && (r.flags & ROW_PQ_OK)) kind = (mode & 2) ? K_LANE_TABLE : K_SERVE_TABLE;
        // (else: no table for this pair — k stays -1: the general kernel takes the walker)
      }
    }
    // ---- the steps a lane takes itself
    if (kind == K_LANE_ROW) {
      const LaneArgs am = LTAB_ARGS();
      const BiasDiv bdiv(am.t.p, am.t.q);
      k = lane_pick_row(fresh_graph(), r, second, prev, bdiv, eo, u, next);
      if (second) { w_mask += 1; w_srch += 8u * (uint32_t)r.deg + 4u * (uint32_t)((r.deg + 31) >> 5); }
    }
    if (kind == K_LANE_TABLE) {
      const LaneArgs at = LTAB_ARGS();
      uint32_t rb = 0;
      k = lane_pick_table(fresh_graph(), r, prev, rprev, at.t.p, at.t.q, eo, u, at.max_csh, next, S_tie, rb);
      if (k == LANE_SERVE) { kind = K_SERVE_TABLE; k = -1; }
      else { if (rb) atomicAdd(&tot[T_SRCH], (unsigned long long)rb); if (k >= 0) { w_tab += 1; w_srch += 8u * (uint32_t)EB_BINS; } }
    }
    // ---- the steps the wave serves, one at a time (the one-walker-per-wave samplers; the owner lane keeps the result)
    {
      unsigned long long pend = __ballot(kind == K_SERVE_FIRST || kind == K_SERVE_MASK || kind == K_SERVE_TABLE);
      while (pend) {
        const int j = __ffsll((long long)pend) - 1;
        pend &= pend - 1ull;
        const int32_t kj = __builtin_amdgcn_readlane(kind, j);
        const Row rj = lane_row(r, j), rpj = lane_row(rprev, j);
        const int32_t prev_j = __builtin_amdgcn_readlane(prev, j);
        const uint32_t eo_j = (uint32_t)__builtin_amdgcn_readlane((int)eo, j);
        const float u_j = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(u), j));
        const LaneArgs av = LTAB_ARGS();
        unsigned f = 0, sv = 0;
        int32_t kk = -1, nx = 0;
        double Sj = 0.0;
        uint32_t add_srch = 0, add_tab = 0, add_mask = 0;
        if (kj == K_SERVE_FIRST) {
          kk = uni(wave_pick_first<false>(fresh_graph(), rj, u_j, f, nx));
        } else {
          Bias b;
          b.p = av.t.p; b.q = av.t.q; b.prev = prev_j; b.second_order = true; b.need_member = true; b.vmin = av.t.g.vmin;
          b.prev_sids = av.t.g.sids + rpj.off; b.prev_deg = rpj.deg; b.prev_hub = rpj.flags >> ROW_HUB_SHIFT;
          if (kj == K_SERVE_MASK) {
            kk = uni(wave_pick_masked<false>(fresh_graph(), rj, b, eo_j, rj.deg > 32 ? av.t.g.em_bits + (size_t)eo_j * 4 : nullptr, u_j, f, nx));
            add_mask = 1; add_srch = 8u * (uint32_t)rj.deg + 4u * (uint32_t)((rj.deg + 31) >> 5);
          } else {
            kk = uni(wave_pick_edge_table<BF, false>(fresh_graph(), rj, b, av.t.g.eb_bins + (size_t)eo_j * 8, u_j, f, sv, mem, nx, stage, &Sj));
            if (kk >= 0) { add_tab = 1; add_srch = 8u * (uint32_t)EB_BINS; }
          }
        }
        nx = uni(nx);
        if (lane == j) { k = kk; next = nx; S_tie = Sj; w_srch += add_srch; w_tab += add_tab; w_mask += add_mask; }
        if ((mode & 8) && lane == 0) atomicAdd(&tot[kj == K_SERVE_FIRST ? T_SV_FIRST : kj == K_SERVE_MASK ? T_SV_MASK : T_SV_TAB], 1ull);
      }
    }
    // ---- the step's outcome
    if (active) {
      const LaneArgs ac = LTAB_ARGS();
      bool handed = false;
      int32_t tie_rec = -1;
      if (!finish) {
        if (k < 0) {
          handed = true;                             // no table / no certificate / a boundary draw: the general kernel takes the walker
          if (k == CHAIN_NEEDED && second && r.deg > ac.t.g.eb_mask_max && ac.t.tie.list) {   // a tie on a table step: its exact chain by the chain kernels
            const TieSink tie = ac.t.tie;
            const unsigned long long c = atomicAdd(tie.cur, 1ull);
            if (c < (unsigned long long)CHAIN_CAP) {
              tie_rec = (int32_t)c;
              const int64_t it = wi / ac.t.n_verts;
              WWalker wr; wr.lw = (int32_t)it; wr.src = ac.t.verts[wi - it * ac.t.n_verts]; wr.prev = prev; wr.curr = curr; tie.recs[c] = wr;
              ChainRec cr; cr.ri = (uint32_t)c; cr.pad = (uint32_t)s; cr.S = S_tie; tie.list[c] = cr;
              atomicAdd(tie.hdr, 1u);
            }
          }
        } else {
          const int sl = s & 3;
          pb0 = sl == 0 ? next : pb0; pb1 = sl == 1 ? next : pb1; pb2 = sl == 2 ? next : pb2; pb3 = sl == 3 ? next : pb3;
          if (sl == 3) {                             // (a full block of four: s <= L + 1 < stride)
            U32x4 v; v.a = (uint32_t)pb0; v.b = (uint32_t)pb1; v.c = (uint32_t)pb2; v.d = (uint32_t)pb3;
            *reinterpret_cast<U32x4 *>(ac.t.paths + wi * stride + (s & ~3)) = v;
            pb0 = pb1 = pb2 = pb3 = -1;
          }
          prev = curr; curr = next; rprev = r; eprev = r.off + k;
          ++s;
          if (s > L + 1) finish = true;
        }
      }
      if (handed) {
        const unsigned long long t = atomicAdd(ac.t.todo_n, 1ull);
        ac.t.todo[t] = (int32_t)wi;
        if (ac.t.tie.todo_tie) ac.t.tie.todo_tie[t] = tie_rec;
        atomicAdd(&ac.t.ctr->strat[SRW_STAT_HANDED_OVER], 1ull);
        active = false;
      } else if (finish) {
        const int32_t len = s;                        // (slots 0 .. s - 1 are written)
        int32_t *path = ac.t.paths + wi * stride;
        const int64_t b0 = (int64_t)(len & ~3);       // the block the walk ended in (its unused slots are the tail's -1), then the rest of the tail
        if (b0 < stride) path[b0] = pb0;
        if (b0 + 1 < stride) path[b0 + 1] = pb1;
        if (b0 + 2 < stride) path[b0 + 2] = pb2;
        if (b0 + 3 < stride) path[b0 + 3] = pb3;
        for (int64_t t = b0 + 4; t < stride; ++t) path[t] = -1;
        ac.t.lens[wi] = len;
        atomicAdd(&tot[T_STEPS], (unsigned long long)(len - 1));
        if (len > 1) atomicAdd(&tot[T_FIRST], 1ull);
        atomicAdd(&tot[T_SRCH], (unsigned long long)w_srch);
        if (w_tab) atomicAdd(&tot[T_TAB], (unsigned long long)w_tab);
        if (w_mask) atomicAdd(&tot[T_MASK], (unsigned long long)w_mask);
        active = false;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    DevCounters *ctr = LTAB_ARGS().t.ctr;
    const unsigned long long srch = tot[T_SRCH] + mem.res_bytes;      // (+ the served table steps' candidates: the wave's count)
    if (tot[T_STEPS]) atomicAdd(&ctr->steps, tot[T_STEPS]);
    if (tot[T_DEAD]) atomicAdd(&ctr->dead_ends, tot[T_DEAD]);
    if (tot[T_TAB]) { atomicAdd(&ctr->ent_reads, tot[T_TAB]); atomicAdd(&ctr->strat[SRW_STRAT_EDGE_TABLE], tot[T_TAB]); }
    if (srch) atomicAdd(&ctr->trials, srch);
    if (tot[T_MASK]) atomicAdd(&ctr->strat[SRW_STRAT_EDGE_MASK], tot[T_MASK]);
    if (tot[T_FIRST]) atomicAdd(&ctr->strat[SRW_STRAT_SCAN], tot[T_FIRST]);
    if (mode & 8) {                                  // experiments: the served steps by kind in the (otherwise unused) p1 / p2 / p3 counters
      atomicAdd(&ctr->strat[SRW_STRAT_P1], tot[T_SV_FIRST]); atomicAdd(&ctr->strat[SRW_STRAT_P2], tot[T_SV_MASK]); atomicAdd(&ctr->strat[SRW_STRAT_P3], tot[T_SV_TAB]);
    }
  }
}

}  // namespace

void launch_walk_tables_lanes(const TabArgs &ta, bool row_filters, int mode, int max_csh, int n_cus, hipStream_t st) {
  // persistent waves, 64 walkers each: enough blocks to fill every CU at the kernel's occupancy
  const int64_t waves = (ta.n_walkers + 63) / 64;
  const int64_t lb = std::max<int64_t>(1, std::min<int64_t>((waves * 64 + TPB - 1) / TPB, (int64_t)n_cus * SRW_LANE_WAVES * 2));
  LaneArgs la; la.t = ta; la.mode = mode; la.max_csh = std::min(std::max(max_csh, 0), 8);
  if (row_filters) hipLaunchKernelGGL((k_walk_tables_lanes<true>), dim3((unsigned)lb), dim3(TPB), 0, st, la);
  else hipLaunchKernelGGL((k_walk_tables_lanes<false>), dim3((unsigned)lb), dim3(TPB), 0, st, la);
}

}  // namespace srw
