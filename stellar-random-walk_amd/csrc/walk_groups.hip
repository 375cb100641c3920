// walk_groups.hip — the bit-exact second-order walk over the per-edge tables with ONE WALKER PER 16 LANES (four per wave).
// Replaces the inner loop of RandomWalk.randomWalk (M/algorithm/RandomWalk.scala:95-139) + RandomSample.secondOrderSample
// (M/algorithm/RandomSample.scala:27-62) for q != 1 on a whole-graph handle whose (prev -> curr) pairs all have a table / mask
// (edge_tables.hip).  Same contract as walk_kernels.hip:k_walk_tables, which it supersedes as the default: persistent waves take
// walkers from a cursor; a walker that meets a pair without a table, a row without a certificate or a draw within rounding distance
// of a CDF boundary is handed over untouched (its index goes to `todo`, a boundary draw on a table step also to the tie list of
// the chain kernels) and k_walk_general redoes it from its first step — the keyed RNG makes that the same path.
//
// Why groups (profiles/r06_group_kernel.md): the one-walker-per-wave kernel issues ~350 vector + ~420 scalar instructions per walk
// step and is bound by that, at 0.54 of the HBM request rate the same run measures.  Here a 64-lane instruction serves four
// walkers, the walker state is vector state (no scalar bookkeeping per step), a wave keeps four chains of dependent round trips
// in flight, and the path leaves as one 64-byte store per 16 steps instead of 16 four-byte stores.
#include <algorithm>

#include "group_sampling.h"
#include "walk_records.h"

namespace srw {
namespace {

#ifndef SRW_GROUP_WAVES
#define SRW_GROUP_WAVES 4            // waves per SIMD (x 4 walkers); 128 VGPRs
#endif
#define GTAB_ARGS() fresh_args<TabArgs>()
#define GFRESH_G() fresh_graph()

template <bool BF>
__global__ __launch_bounds__(TPB, SRW_GROUP_WAVES) void k_walk_tables_groups(TabArgs a0) {
  __shared__ __attribute__((aligned(16))) uint32_t stage_all[TPB / 64][1024];
  using namespace g16;
  const int lane = lane_id(), gl = lane & 15;
  uint32_t *stage = stage_all[threadIdx.x >> 6];
  const int32_t L = a0.L;
  const int64_t stride = (int64_t)L + 2;
  // per-group totals (replicated in the group's lanes; lane 0 of each group reports them)
  unsigned long long steps = 0, srch = 0, res_bytes = 0;
  uint32_t dead = 0, n_tab = 0, n_mask = 0, n_first = 0;
  // the group's walker (all group-uniform)
  bool active = false, exhausted = false;
  int64_t wi = 0, eprev = 0;
  int32_t s = 1, src = 0, prev = 0, curr = 0, len = 1, it_off = 0;
  uint32_t iter = 0, ksrc = 0;
  Row rprev; rprev.off = 0; rprev.deg = 0; rprev.flags = 0;
  float ub = 0.0f;                                   // the draws of 16 consecutive steps, lane gl: step (first of the batch) + gl
  int32_t pbuf = -1;                                 // 16 path slots, lane gl: slot (block base) + gl; -1 = unused
  uint32_t w_tab = 0, w_mask = 0, w_srch = 0;        // (a handed-over walker is not counted)
  while (true) {
    // ---- groups without a walker take the next ones from the cursor (one atomic per wave)
    {
      const unsigned long long need = __ballot(!active && !exhausted && gl == 0);
      if (need) {
        const TabArgs aw = GTAB_ARGS();
        unsigned long long grab = 0;
        if (lane == 0) grab = atomicAdd(aw.cursor, (unsigned long long)__popcll(need));
        const int64_t w0 = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab));
        if (!active && !exhausted) {
          wi = w0 + (int64_t)__popcll(need & ((1ull << (lane & 48)) - 1ull));
          if (wi >= aw.n_walkers) exhausted = true;
          else {
            const int64_t it = wi / aw.n_verts, vi = wi - it * aw.n_verts;
            it_off = (int32_t)it;
            iter = (uint32_t)(aw.first_walk + it);
            src = aw.verts[vi];
            ksrc = (uint32_t)rng_source(aw.g, src);
            s = 1; prev = src; curr = src; len = 1; eprev = 0;
            rprev.off = 0; rprev.deg = 0; rprev.flags = 0;
            pbuf = gl == 0 ? src : -1;
            w_tab = 0; w_mask = 0; w_srch = 0;
            active = true;
          }
        }
      }
    }
    if (!__ballot(active)) break;
    if (active) {
      // ---- one step of the group's walker
      const TabArgs as = GTAB_ARGS();
      const GraphView &gs = as.g;
      const bool second = s > 1;
      const int64_t cslot = (int64_t)curr - gs.vmin;
      const bool in_range = cslot >= 0 && cslot < gs.n_slots;
      Row r = gs.rows[in_range ? cslot : 0];
      uint32_t eo = EB_NONE;
      if (second) eo = gs.eb_off[eprev];
      if (!in_range) { r.off = 0; r.deg = 0; r.flags = 0; }
      bool finish = false, handed = false;
      int32_t tie_rec = -1;
      if (r.deg == 0) { dead += second ? 1u : 0u; finish = true; }
      else {
        const int di = (s - 1) & 15;
        if (di == 0) ub = draw_uniform(as.rng, iter, ksrc, (uint32_t)(s + gl));
        const float u = grp_get(ub, di);
        int32_t k = -1, next = 0;
        if (!second || (r.deg <= gs.eb_mask_max && (r.deg <= 32 || eo != EB_NONE))) {
          const BiasDiv bdiv(as.p, as.q);
          k = grp_pick_row(GFRESH_G(), r, second, prev, bdiv, eo, u, next);
          if (second) { w_mask += 1; w_srch += 8u * (uint32_t)r.deg + 4u * (uint32_t)((r.deg + 31) >> 5); }
        } else if (r.deg > gs.eb_mask_max && eo != EB_NONE && (r.flags & ROW_PQ_OK)) {
          double S_tie = 0.0;
          k = grp_pick_table<BF>(GFRESH_G(), r, prev, rprev, as.p, as.q, eo, u, stage, next, S_tie, res_bytes);
          if (k >= 0) { w_tab += 1; w_srch += 8u * (uint32_t)EB_BINS; }
          else if (k == CHAIN_NEEDED && as.tie.list) {   // a tie on a table step: its exact chain by the chain kernels (the whole GPU)
            if (gl == 0) {
              const TieSink tie = GTAB_ARGS().tie;
              const unsigned long long c = atomicAdd(tie.cur, 1ull);
              if (c < (unsigned long long)CHAIN_CAP) {
                tie_rec = (int32_t)c;
                WWalker wr; wr.lw = it_off; wr.src = src; wr.prev = prev; wr.curr = curr; tie.recs[c] = wr;
                ChainRec cr; cr.ri = (uint32_t)c; cr.pad = (uint32_t)s; cr.S = S_tie; tie.list[c] = cr;
                atomicAdd(tie.hdr, 1u);
              }
            }
          }
        }
        if (k < 0) handed = true;                    // no table / no certificate / a boundary draw: the general kernel takes the walker
        else {
          int32_t *path = GTAB_ARGS().paths + wi * stride;
          pbuf = gl == (s & 15) ? next : pbuf;
          if ((s & 15) == 15) { path[(s & ~15) + gl] = pbuf; pbuf = -1; }      // (a full block: s <= L + 1 < stride)
          prev = curr; curr = next; ++len; rprev = r; eprev = r.off + k;
          ++s;
          if (s > L + 1) finish = true;
        }
      }
      if (handed) {
        if (gl == 0) {
          const TabArgs ah = GTAB_ARGS();
          const unsigned long long t = atomicAdd(ah.todo_n, 1ull);
          ah.todo[t] = (int32_t)wi;
          if (ah.tie.todo_tie) ah.tie.todo_tie[t] = tie_rec;
          atomicAdd(&ah.ctr->strat[SRW_STAT_HANDED_OVER], 1ull);
        }
        active = false;
      } else if (finish) {
        const TabArgs af = GTAB_ARGS();
        int32_t *path = af.paths + wi * stride;
        const int64_t b0 = (int64_t)(len & ~15);      // the block the walk ended in (its unused lanes are the tail's -1), then the rest of the tail
        if (b0 + gl < stride) path[b0 + gl] = pbuf;
        for (int64_t t = b0 + 16 + gl; t < stride; t += 16) path[t] = -1;
        if (gl == 0) af.lens[wi] = len;
        steps += (unsigned long long)(len - 1); n_first += len > 1 ? 1u : 0u;
        srch += w_srch; n_tab += w_tab; n_mask += w_mask;
        active = false;
      }
    }
  }
  if (gl == 0) {
    DevCounters *ctr = GTAB_ARGS().ctr;
    srch += res_bytes;
    if (steps) atomicAdd(&ctr->steps, steps);
    if (dead) atomicAdd(&ctr->dead_ends, (unsigned long long)dead);
    if (n_tab) { atomicAdd(&ctr->ent_reads, (unsigned long long)n_tab); atomicAdd(&ctr->strat[SRW_STRAT_EDGE_TABLE], (unsigned long long)n_tab); }
    if (srch) atomicAdd(&ctr->trials, srch);
    if (n_mask) atomicAdd(&ctr->strat[SRW_STRAT_EDGE_MASK], (unsigned long long)n_mask);
    if (n_first) atomicAdd(&ctr->strat[SRW_STRAT_SCAN], (unsigned long long)n_first);
  }
}

}  // namespace

void launch_walk_tables_groups(const TabArgs &ta, bool row_filters, int n_cus, hipStream_t st) {
  // persistent waves, four walkers each: enough blocks to fill every CU at the kernel's occupancy
  const int64_t waves = (ta.n_walkers + 3) / 4;
  const int64_t lb = std::max<int64_t>(1, std::min<int64_t>((waves * 64 + TPB - 1) / TPB, (int64_t)n_cus * SRW_GROUP_WAVES * 2));
  if (row_filters) hipLaunchKernelGGL((k_walk_tables_groups<true>), dim3((unsigned)lb), dim3(TPB), 0, st, ta);
  else hipLaunchKernelGGL((k_walk_tables_groups<false>), dim3((unsigned)lb), dim3(TPB), 0, st, ta);
}

}  // namespace srw
