// walk_rounds.hip — the bit-exact second-order walk over the per-edge tables IN ROUNDS of two kernels.
// Replaces the inner loop of RandomWalk.randomWalk (M/algorithm/RandomWalk.scala:95-139) + RandomSample.secondOrderSample
// (M/algorithm/RandomSample.scala:27-62) for q != 1 on a whole-graph handle whose pairs all have a table / mask (edge_tables.hip);
// same contract as walk_kernels.hip:k_walk_tables (todo list, tie list, counters; k_walk_general redoes what is handed over).
//
// Why rounds (profiles/r06_lane_kernel.md): a lane takes 77 % of config 3's table steps itself at a third of the wave kernel's
// instructions, but the steps it cannot take — first steps, rows with a mask, long located chunks, staged N(prev) — need the 64-lane
// samplers, whose ~64 VGPRs next to the lanes' own state left the one-kernel form (walk_lanes.hip) 6 waves per SIMD and ONE chain of
// dependent round trips per wave for the served steps: bound by latency.  Here the two kinds of step live in two kernels:
//   k_round_lanes  one walker per LANE, no wave sampler in it: every walker of the list advances through the steps a lane can
//                  take and PARKS (80 B of state: the walker, the row of curr, the pair word, the draw) at the first step it cannot;
//   k_round_serve  one parked step per WAVE — k_walk_tables' step, one chain per wave at 8 waves per SIMD — then the walker goes
//                  back to the lanes' list.
// A round is one launch of each; every alive walker advances by at least one step per round, so at most walkLength + 1 rounds (the
// host reads the list length every few rounds and stops when it is zero).  Paths, lengths, counters as the other table kernels.
#include <algorithm>

#include "lane_sampling.h"
#include "walk_records.h"

namespace srw {
namespace {
using namespace lane;

#ifndef SRW_ROUND_LANE_WAVES
#define SRW_ROUND_LANE_WAVES 6
#endif
#ifndef SRW_ROUND_SERVE_WAVES
#define SRW_ROUND_SERVE_WAVES 8
#endif

// A walker between the kernels.  q0-q2: the walker (what k_round_lanes needs to go on); q3-q4: the step it is parked at (the row of
// curr, the pair word, the draw: k_round_serve starts at the sampler, not at a row fetch).  s == 0: a fresh walker (first step not taken).
struct alignas(16) WState {
  int64_t eprev, rp_off;                                   // q0
  int32_t rp_deg; uint32_t rp_flags; int32_t s, prev;      // q1
  int32_t curr; uint32_t w_tab, w_mask, w_srch;            // q2
  int64_t r_off; int32_t r_deg; uint32_t r_flags;          // q3
  uint32_t eo; float u; int32_t pad0, pad1;                // q4
};
static_assert(sizeof(WState) == 80, "WState is five 16-byte words");

struct RoundArgs {
  TabArgs t;
  WState *state;
  const int32_t *list_in; int32_t *list_out;               // walkers of this launch (null: the identity — every walker, fresh) | walkers for the other kernel
  const unsigned long long *n_in; unsigned long long *n_out, *cursor;
  int32_t max_csh;
};
#define RARGS() fresh_args<RoundArgs>()

enum { RT_STEPS = 0, RT_SRCH, RT_DEAD, RT_TAB, RT_MASK, RT_FIRST, RT_N };

__device__ inline void finish_walker(const TabArgs &t, int64_t wi, int32_t len, uint32_t w_tab, uint32_t w_mask, uint32_t w_srch, unsigned long long *tot) {
  t.lens[wi] = len;
  atomicAdd(&tot[RT_STEPS], (unsigned long long)(len - 1));
  if (len > 1) atomicAdd(&tot[RT_FIRST], 1ull);
  if (w_srch) atomicAdd(&tot[RT_SRCH], (unsigned long long)w_srch);
  if (w_tab) atomicAdd(&tot[RT_TAB], (unsigned long long)w_tab);
  if (w_mask) atomicAdd(&tot[RT_MASK], (unsigned long long)w_mask);
}
__device__ inline void hand_over(const TabArgs &t, int64_t wi, int32_t tie_rec) {
  const unsigned long long x = atomicAdd(t.todo_n, 1ull);
  t.todo[x] = (int32_t)wi;
  if (t.tie.todo_tie) t.tie.todo_tie[x] = tie_rec;
  atomicAdd(&t.ctr->strat[SRW_STAT_HANDED_OVER], 1ull);
}
__device__ inline int32_t record_tie(const TabArgs &t, int64_t wi, int32_t s, int32_t prev, int32_t curr, double S) {
  const TieSink tie = t.tie;
  const unsigned long long c = atomicAdd(tie.cur, 1ull);
  if (c >= (unsigned long long)CHAIN_CAP) return -1;
  const int64_t it = wi / t.n_verts;
  WWalker wr; wr.lw = (int32_t)it; wr.src = t.verts[wi - it * t.n_verts]; wr.prev = prev; wr.curr = curr; tie.recs[c] = wr;
  ChainRec cr; cr.ri = (uint32_t)c; cr.pad = (uint32_t)s; cr.S = S; tie.list[c] = cr;
  atomicAdd(tie.hdr, 1u);
  return (int32_t)c;
}
__device__ inline void flush_totals(DevCounters *ctr, const unsigned long long *tot, unsigned long long extra_srch) {
  const unsigned long long srch = tot[RT_SRCH] + extra_srch;
  if (tot[RT_STEPS]) atomicAdd(&ctr->steps, tot[RT_STEPS]);
  if (tot[RT_DEAD]) atomicAdd(&ctr->dead_ends, tot[RT_DEAD]);
  if (tot[RT_TAB]) { atomicAdd(&ctr->ent_reads, tot[RT_TAB]); atomicAdd(&ctr->strat[SRW_STRAT_EDGE_TABLE], tot[RT_TAB]); }
  if (srch) atomicAdd(&ctr->trials, srch);
  if (tot[RT_MASK]) atomicAdd(&ctr->strat[SRW_STRAT_EDGE_MASK], tot[RT_MASK]);
  if (tot[RT_FIRST]) atomicAdd(&ctr->strat[SRW_STRAT_SCAN], tot[RT_FIRST]);
}

// ---- the lanes: every walker of the list through the steps a lane can take; parked at the first one it cannot ---------------------------
__global__ __launch_bounds__(TPB, SRW_ROUND_LANE_WAVES) void k_round_lanes(RoundArgs a0) {
  __shared__ unsigned long long tot_all[TPB / 64][RT_N];
  const int lane = lane_id();
  unsigned long long *tot = tot_all[threadIdx.x >> 6];
  if (lane < RT_N) tot[lane] = 0ull;
  __builtin_amdgcn_wave_barrier();
  const int32_t L = a0.t.L;
  const int64_t stride = (int64_t)L + 2;
  const unsigned long long n_in = *a0.n_in;
  bool active = false, exhausted = false;
  int64_t wi = 0, eprev = 0;
  int32_t s = 1, prev = 0, curr = 0;
  uint32_t iter = 0, ksrc = 0;
  Row rprev; rprev.off = 0; rprev.deg = 0; rprev.flags = 0;
  int32_t pb0 = 0, pb1 = 0, pb2 = 0, pb3 = 0;        // the path slots (s & ~3) .. (s | 3) ...
  uint32_t pbm = 0u;                                 // ... and which of them THIS launch wrote (bit t: slot block + t)
  uint32_t w_tab = 0, w_mask = 0, w_srch = 0;
  while (true) {
    {   // lanes without a walker take the next ones of the list (one atomic per wave)
      const unsigned long long need = __ballot(!active && !exhausted);
      if (need) {
        const RoundArgs aw = RARGS();
        unsigned long long grab = 0;
        if (lane == 0) grab = atomicAdd(aw.cursor, (unsigned long long)__popcll(need));
        const unsigned long long g0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab);
        if (!active && !exhausted) {
          const unsigned long long li = g0 + (unsigned long long)__popcll(need & ((1ull << lane) - 1ull));
          if (li >= n_in) exhausted = true;
          else if ((wi = aw.list_in[li]) >= 0) {       // (-1: a walker the wave finished or handed over — the slot is a hole)
            const uint4 *sp = reinterpret_cast<const uint4 *>(aw.state + wi);
            const uint4 q0 = sp[0], q1 = sp[1], q2 = sp[2];
            eprev = (int64_t)(((uint64_t)q0.y << 32) | q0.x);
            rprev.off = (int64_t)(((uint64_t)q0.w << 32) | q0.z); rprev.deg = (int32_t)q1.x; rprev.flags = q1.y;
            s = (int32_t)q1.z; prev = (int32_t)q1.w; curr = (int32_t)q2.x; w_tab = q2.y; w_mask = q2.z; w_srch = q2.w;
            const int64_t it = wi / aw.t.n_verts, vi = wi - it * aw.t.n_verts;
            iter = (uint32_t)(aw.t.first_walk + it);
            ksrc = (uint32_t)rng_source(aw.t.g, aw.t.verts[vi]);
            pbm = 0u;
            active = true;
          }
        }
      }
    }
    if (!__ballot(active || !exhausted)) break;
    bool park = false;
    if (active) {
      const RoundArgs as = RARGS();
      const GraphView &gs = as.t.g;
      const int64_t cslot = (int64_t)curr - gs.vmin;
      const bool in_range = cslot >= 0 && cslot < gs.n_slots;
      Row r = gs.rows[in_range ? cslot : 0];
      const uint32_t eo = gs.eb_off[eprev];            // (s >= 2 here: the first step is always served)
      if (!in_range) { r.off = 0; r.deg = 0; r.flags = 0; }
      int32_t k = LANE_SERVE, next = 0;
      double S_tie = 0.0;
      float u = 0.0f;
      bool finish = false;
      if (r.deg == 0) { atomicAdd(&tot[RT_DEAD], 1ull); finish = true; }
      else {
        u = draw_uniform(as.t.rng, iter, ksrc, (uint32_t)s);
        if (r.deg > gs.eb_mask_max && eo != EB_NONE && (r.flags & ROW_PQ_OK)) {
          uint32_t rb = 0;
          k = lane_pick_table(fresh_graph(), r, prev, rprev, as.t.p, as.t.q, eo, u, as.max_csh, next, S_tie, rb);
          if (k != LANE_SERVE) { if (rb) atomicAdd(&tot[RT_SRCH], (unsigned long long)rb); if (k >= 0) { w_tab += 1; w_srch += 8u * (uint32_t)EB_BINS; } }
        } else if (!(r.deg <= gs.eb_mask_max && (r.deg <= 32 || eo != EB_NONE))) k = -1;      // no table for this pair: the general kernel takes the walker
      }
      int32_t *path = as.t.paths + wi * stride;
      auto flush_slots = [&]() {                       // the slots of the current block of four written by this launch
        const int32_t b0 = (s - 1) & ~3;               // (slot s - 1 is the last one written)
        if (pbm & 1u) path[b0] = pb0;
        if (pbm & 2u) path[b0 + 1] = pb1;
        if (pbm & 4u) path[b0 + 2] = pb2;
        if (pbm & 8u) path[b0 + 3] = pb3;
        pbm = 0u;
      };
      if (finish) {                                    // a dead end: slots 0 .. s - 1 are the path
        flush_slots();
        for (int64_t x = s; x < stride; ++x) path[x] = -1;
        finish_walker(as.t, wi, s, w_tab, w_mask, w_srch, tot);
        active = false;
      } else if (k == LANE_SERVE) {                    // park: the wave takes this step (k_round_serve), with the row, the pair word and the draw
        flush_slots();
        uint4 *sp = reinterpret_cast<uint4 *>(as.state + wi);
        uint4 q;
        q.x = (uint32_t)eprev; q.y = (uint32_t)((uint64_t)eprev >> 32); q.z = (uint32_t)rprev.off; q.w = (uint32_t)((uint64_t)rprev.off >> 32); sp[0] = q;
        q.x = (uint32_t)rprev.deg; q.y = rprev.flags; q.z = (uint32_t)s; q.w = (uint32_t)prev; sp[1] = q;
        q.x = (uint32_t)curr; q.y = w_tab; q.z = w_mask; q.w = w_srch; sp[2] = q;
        q.x = (uint32_t)r.off; q.y = (uint32_t)((uint64_t)r.off >> 32); q.z = (uint32_t)r.deg; q.w = r.flags; sp[3] = q;
        q.x = eo; q.y = __float_as_uint(u); q.z = 0u; q.w = 0u; sp[4] = q;
        park = true;
        active = false;
      } else if (k < 0) {                              // no table / a boundary draw: the general kernel takes the walker
        int32_t tie_rec = -1;
        if (k == CHAIN_NEEDED && as.t.tie.list) tie_rec = record_tie(as.t, wi, s, prev, curr, S_tie);
        hand_over(as.t, wi, tie_rec);
        active = false;
      } else {
        const int sl = s & 3;
        if (sl == 0) pb0 = next; else if (sl == 1) pb1 = next; else if (sl == 2) pb2 = next; else pb3 = next;      // (the previous block left with its last slot, below)
        pbm |= 1u << sl;
        prev = curr; curr = next; rprev = r; eprev = r.off + k;
        ++s;
        if (sl == 3) flush_slots();                    // (a block's last slot: what this launch wrote of it goes out)
        if (s > L + 1) {                               // the path is complete: L + 2 slots
          flush_slots();
          finish_walker(as.t, wi, s, w_tab, w_mask, w_srch, tot);
          active = false;
        }
      }
    }
    {   // the walkers parked in this trip join the wave's list: ONE atomic per wave (one per walker on one address serialises the GPU:
        // the first version of this file spent 8.4 s per iteration of config 3 on it)
      const unsigned long long pm = __ballot(park);
      if (pm) {
        const RoundArgs ap = RARGS();
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(ap.n_out, (unsigned long long)__popcll(pm));
        const unsigned long long b0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        if (park) ap.list_out[b0 + (unsigned long long)__popcll(pm & ((1ull << lane) - 1ull))] = (int32_t)wi;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) flush_totals(RARGS().t.ctr, tot, 0ull);
}

// ---- the wave: ONE parked step per wave (k_walk_tables' step), the walker back to the lanes' list ---------------------------------------------
template <bool BF>
__global__ __launch_bounds__(TPB, SRW_ROUND_SERVE_WAVES) void k_round_serve(RoundArgs a0) {
  __shared__ __attribute__((aligned(16))) uint32_t stage_all[TPB / 64][1024];
  __shared__ unsigned long long tot_all[TPB / 64][RT_N];
  const int lane = lane_id();
  uint32_t *stage = stage_all[threadIdx.x >> 6];
  unsigned long long *tot = tot_all[threadIdx.x >> 6];
  if (lane < RT_N) tot[lane] = 0ull;
  __builtin_amdgcn_wave_barrier();
  Member mem; mem.mode = 0; mem.bm = stage; mem.seg_base = 0;
  const int32_t L = a0.t.L;
  const int64_t stride = (int64_t)L + 2;
  const bool identity = a0.list_in == nullptr;         // the first launch of a walk: every walker, fresh
  const unsigned long long n_in = identity ? (unsigned long long)a0.t.n_walkers : *a0.n_in;
  // (no cursor, no append: wave w takes entries w, w + waves, ... — one step each, hundreds per wave — and answers in the SAME slot of the
  //  lanes' list, -1 where the walker finished or was handed over)
  const unsigned long long wave_id = (unsigned long long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6), n_waves = (unsigned long long)gridDim.x * (TPB / 64);
  if (wave_id == 0 && lane == 0) *a0.n_out = n_in;
  for (unsigned long long li = wave_id; li < n_in; li += n_waves) {
    const RoundArgs aw = RARGS();
    const int64_t wi = identity ? (int64_t)li : (int64_t)__builtin_amdgcn_readfirstlane(aw.list_in[li]);
    // the parked step (scalar loads: the state was written by an EARLIER launch)
    Row r, rprev;
    int32_t s, prev, curr;
    uint32_t eo, w_tab, w_mask, w_srch;
    float u;
    int32_t src = 0;
    if (identity) {
      const int64_t it = wi / aw.t.n_verts, vi = wi - it * aw.t.n_verts;
      src = __builtin_amdgcn_readfirstlane(aw.t.verts[vi]);
      s = 1; prev = src; curr = src; eo = EB_NONE; w_tab = w_mask = w_srch = 0u;
      rprev.off = 0; rprev.deg = 0; rprev.flags = 0;
      const int64_t cslot = (int64_t)curr - aw.t.g.vmin;
      const bool in_range = cslot >= 0 && cslot < aw.t.g.n_slots;
      r = uniform_row(aw.t.g.rows[in_range ? cslot : 0]);
      if (!in_range) { r.off = 0; r.deg = 0; r.flags = 0; }
      const uint32_t ksrc = (uint32_t)__builtin_amdgcn_readfirstlane(rng_source(aw.t.g, src));
      u = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(
            draw_uniform(aw.t.rng, (uint32_t)(aw.t.first_walk + it), (uint32_t)on_vector((int32_t)ksrc), 1u))));
      if (lane == 0) aw.t.paths[wi * stride] = src;
    } else {
      const WState st = aw.state[wi];
      rprev.off = uni(st.rp_off); rprev.deg = uni(st.rp_deg); rprev.flags = uni(st.rp_flags);
      s = uni(st.s); prev = uni(st.prev); curr = uni(st.curr); w_tab = uni(st.w_tab); w_mask = uni(st.w_mask); w_srch = uni(st.w_srch);
      r.off = uni(st.r_off); r.deg = uni(st.r_deg); r.flags = uni(st.r_flags);
      eo = uni(st.eo); u = __uint_as_float(uni(__float_as_uint(st.u)));
    }
    const bool second = s > 1;
    int32_t *path = aw.t.paths + wi * stride;
    if (r.deg == 0) {                                  // (only a fresh walker can stand on a row without candidates here)
      for (int64_t x = s + lane; x < stride; x += 64) path[x] = -1;
      if (lane == 0) { finish_walker(aw.t, wi, s, w_tab, w_mask, w_srch, tot); aw.list_out[li] = -1; }
      continue;
    }
    unsigned f = 0, sv = 0;
    int32_t k, next = 0, tie_rec = -1;
    double S_tie = 0.0;
    if (!second) {
      k = uni(wave_pick_first<false>(fresh_graph(), r, u, f, next));
    } else {
      const RoundArgs as = RARGS();
      const GraphView &gs = as.t.g;
      Bias b;
      b.p = as.t.p; b.q = as.t.q; b.prev = prev; b.second_order = true; b.need_member = true; b.vmin = gs.vmin;
      b.prev_sids = gs.sids + rprev.off; b.prev_deg = rprev.deg; b.prev_hub = rprev.flags >> ROW_HUB_SHIFT;
      if (r.deg <= gs.eb_mask_max && (r.deg <= 32 || eo != EB_NONE)) {
        k = uni(wave_pick_masked<false>(fresh_graph(), r, b, eo, r.deg > 32 ? gs.em_bits + (size_t)eo * 4 : nullptr, u, f, next));
        w_mask += 1; w_srch += 8u * (uint32_t)r.deg + 4u * (uint32_t)((r.deg + 31) >> 5);
      } else if (r.deg > gs.eb_mask_max && eo != EB_NONE && (r.flags & ROW_PQ_OK)) {
        k = uni(wave_pick_edge_table<BF, false>(fresh_graph(), r, b, gs.eb_bins + (size_t)eo * 8, u, f, sv, mem, next, stage, &S_tie));
        if (k >= 0) { w_tab += 1; w_srch += 8u * (uint32_t)EB_BINS; }
        else if (k == CHAIN_NEEDED && as.t.tie.list) {
          if (lane == 0) tie_rec = record_tie(RARGS().t, wi, s, prev, curr, S_tie);
        }
      } else k = -1;
      k = uni(k);
    }
    next = uni(next);
    const RoundArgs ao = RARGS();
    if (k < 0) {
      if (lane == 0) { hand_over(ao.t, wi, tie_rec); ao.list_out[li] = -1; }
      continue;
    }
    if (lane == 0) path[s] = next;
    if (s + 1 > L + 1) {                               // the path is complete
      if (lane == 0) { finish_walker(ao.t, wi, s + 1, w_tab, w_mask, w_srch, tot); ao.list_out[li] = -1; }
      continue;
    }
    if (lane == 0) {                                   // back to the lanes
      const int64_t eprev = r.off + k;
      uint4 *sp = reinterpret_cast<uint4 *>(ao.state + wi);
      uint4 q;
      q.x = (uint32_t)eprev; q.y = (uint32_t)((uint64_t)eprev >> 32); q.z = (uint32_t)r.off; q.w = (uint32_t)((uint64_t)r.off >> 32); sp[0] = q;
      q.x = (uint32_t)r.deg; q.y = r.flags; q.z = (uint32_t)(s + 1); q.w = (uint32_t)curr; sp[1] = q;
      q.x = (uint32_t)next; q.y = w_tab; q.z = w_mask; q.w = w_srch; sp[2] = q;
      ao.list_out[li] = (int32_t)wi;
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) flush_totals(RARGS().t.ctr, tot, mem.res_bytes);
}

}  // namespace

// ta.cursor / ta.todo / ta.tie as for the other table kernels.  Rounds until no walker is left (at most walk length + 1).
void launch_walk_tables_rounds(srw_handle *h, const TabArgs &ta, bool row_filters, int max_csh, hipStream_t st) {
  const int64_t n = ta.n_walkers;
  if (n <= 0) return;
  if (n > 0x7FFFFFFFll) throw Error(SRW_ERR_INVALID, "table walk in rounds: more than 2^31 walkers in one launch");
  h->round_state.ensure((size_t)n * sizeof(WState));
  h->round_list.ensure((size_t)n * 2);
  h->round_ctr.ensure(8);
  unsigned long long *ctr = h->round_ctr.p;            // [0] lanes' list length, [1] serve list length, [2] cursor
  RoundArgs ra;
  ra.t = ta; ra.state = reinterpret_cast<WState *>(h->round_state.p); ra.max_csh = std::min(std::max(max_csh, 0), 8);
  int32_t *list_lanes = h->round_list.p, *list_serve = h->round_list.p + n;
  const int64_t serve_blocks_max = (int64_t)h->n_cus * SRW_ROUND_SERVE_WAVES * 2, lane_blocks_max = (int64_t)h->n_cus * SRW_ROUND_LANE_WAVES * 2;
  SRW_HIP(hipMemsetAsync(ctr, 0, 8 * sizeof(unsigned long long), st));
  unsigned long long n_serve_host = (unsigned long long)n;
  for (int round = 0; round <= ta.L + 1; ++round) {
    // the wave serves the parked steps (round 0: the first step of every walker) ...
    ra.list_in = round == 0 ? nullptr : list_serve; ra.list_out = list_lanes; ra.n_in = ctr + 1; ra.n_out = ctr + 0; ra.cursor = ctr + 2;
    {
      const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(((int64_t)n_serve_host * 64 + TPB - 1) / TPB, serve_blocks_max));
      if (row_filters) hipLaunchKernelGGL((k_round_serve<true>), dim3((unsigned)blocks), dim3(TPB), 0, st, ra);
      else hipLaunchKernelGGL((k_round_serve<false>), dim3((unsigned)blocks), dim3(TPB), 0, st, ra);
    }
    SRW_HIP(hipMemsetAsync(ctr + 1, 0, 2 * sizeof(unsigned long long), st));      // the serve list is consumed; the cursor for the lanes
    // ... the lanes advance every walker that came back, and park it at its next step for the wave
    ra.list_in = list_lanes; ra.list_out = list_serve; ra.n_in = ctr + 0; ra.n_out = ctr + 1; ra.cursor = ctr + 2;
    {
      const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(((int64_t)n_serve_host + TPB - 1) / TPB, lane_blocks_max));
      hipLaunchKernelGGL(k_round_lanes, dim3((unsigned)blocks), dim3(TPB), 0, st, ra);
    }
    SRW_HIP(hipMemsetAsync(ctr + 0, 0, sizeof(unsigned long long), st));          // the lanes' list is consumed
    SRW_HIP(hipMemsetAsync(ctr + 2, 0, sizeof(unsigned long long), st));          // the cursor for the next serve
    if ((round & 3) == 3 || round >= ta.L) {           // how many walkers are parked: every fourth round (one small copy + a wait)
      SRW_HIP(hipMemcpyAsync(&n_serve_host, ctr + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      SRW_HIP(hipStreamSynchronize(st));
      if (n_serve_host == 0) break;
    }
  }
  SRW_HIP(hipGetLastError());
  if (n_serve_host != 0) throw Error(SRW_ERR_HIP, "table walk in rounds: walkers left after walk length + 2 rounds");
}

}  // namespace srw
