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

#include "lane_sampling.h"
#include "walk_records.h"

namespace srw {
namespace {
using namespace lane;

// waves per SIMD: the kernel is bound by the latency of its served steps (one chain of dependent round trips per wave at a time), so
// occupancy pays even through spills — config 3, tables per lane: 4 waves (128 VGPRs) 662 ms, 5 (96) 607 ms, 6 (80 + 244 B of scratch) 587 ms
// per iteration (profiles/r06_lane_kernel.md)
#ifndef SRW_LANE_WAVES
#define SRW_LANE_WAVES 6
#endif
#define LTAB_ARGS() fresh_args<LaneArgs>()
struct LaneArgs { TabArgs t; int32_t mode, max_csh; }; // mode: bit 0 mask rows per lane, bit 1 table steps per lane (0: every step served); max_csh: located chunks of up to 2^max_csh candidates per lane

enum : int32_t { K_NONE = 0, K_SERVE_FIRST = 1, K_SERVE_MASK = 2, K_SERVE_TABLE = 3, K_LANE_ROW = 4, K_LANE_TABLE = 5 };
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
