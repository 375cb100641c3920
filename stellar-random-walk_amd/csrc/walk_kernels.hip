// walk_kernels.hip — the walk itself (gfx950).  Replaces RandomWalk.initFirstStep + the super-step loop of
// RandomWalk.randomWalk (M/algorithm/RandomWalk.scala:51-66, 95-139).
//
//   k_walk_first_order   p == q == 1: one walker per LANE, all walk_length+1 steps in one launch, O(1)
//                        exact sampling through the CDF+guide records (16 B, one sector per probe), paths
//                        staged in LDS and flushed as 64-byte runs.  HBM-latency/sector bound gather.
//   k_walk_general       any p, q: one walker per WAVE, persistent waves taking walkers from a counter; per step the
//                        first sampler that applies (sampling.h): search over exact prefix sums (specials binned by
//                        position in LDS, sorted-row intersection by probes / edge hash / id-window bitmap),
//                        certified parallel scan of the streamed row, or the reference's sequential f64 chain.
//   k_walk_alias         Mode A: per-vertex alias tables + rejection, one walker per lane.
//   k_shard_step(_fo)    one super-step of the vertex-sharded multi-GPU path: same samplers, one step per record,
//                        survivors counted per destination in LDS, k_shard_offsets + k_shard_bucket group them by
//                        owner(next) for the RCCL all-to-all.
// No MFMA anywhere: integer/byte gather work bounded by HBM.
#include <algorithm>
#include <map>
#include <tuple>
#include <chrono>
#include <cstring>

#include "engine.h"
#include "sampling.h"
#include "walk_records.h"

namespace srw {
namespace {
constexpr int TILE = 16;  // path slots staged in LDS between flushes (64 B per walker per flush)

__device__ inline const Row *row_of(const GraphView &g, int32_t v) {
  int64_t s = (int64_t)v - g.vmin;
  if (s < 0 || s >= g.n_slots) return nullptr;
  return g.rows + s;
}

__device__ inline void flush_counters(DevCounters *ctr, unsigned long long steps, unsigned long long dead,
                                      unsigned long long degc, unsigned long long degp, unsigned long long reads,
                                      unsigned long long fb) {
  steps = wave_sum_u64(steps); dead = wave_sum_u64(dead); degc = wave_sum_u64(degc);
  degp = wave_sum_u64(degp); reads = wave_sum_u64(reads); fb = wave_sum_u64(fb);
  if (lane_id() == 0) {
    if (steps) atomicAdd(&ctr->steps, steps);
    if (dead) atomicAdd(&ctr->dead_ends, dead);
    if (degc) atomicAdd(&ctr->sum_deg_curr, degc);
    if (degp) atomicAdd(&ctr->sum_deg_prev, degp);
    if (reads) atomicAdd(&ctr->ent_reads, reads);
    if (fb) atomicAdd(&ctr->fallbacks, fb);
  }
}

// ---------------------------------------------------------------------------------------------------------
template <bool NT, int MINW, bool COMPACT>
__global__ __launch_bounds__(TPB, MINW) void k_walk_first_order(GraphView g, const int32_t *__restrict__ verts,
                                                          int64_t n_verts, int64_t n_walkers, int32_t L,
                                                          int32_t first_walk, RngSpec rng,
                                                          int32_t *__restrict__ paths, int32_t *__restrict__ lens,
                                                          DevCounters *ctr) {
  __shared__ int32_t tile[TPB / 64][64][TILE + 1];
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const int64_t wi = blockIdx.x * (int64_t)TPB + threadIdx.x;
  const int64_t wave_base = wi - lane;
  const int64_t stride = (int64_t)L + 2;
  bool alive = wi < n_walkers;
  uint32_t iter = 0; int32_t src = 0;
  Row r; r.off = 0; r.deg = 0; r.flags = 0;           // row descriptor of the current vertex
  if (alive) {
    int64_t it = wi / n_verts, vi = wi - it * n_verts;
    iter = (uint32_t)(first_walk + it);
    src = verts[vi];
    const Row *rp = row_of(g, src);
    if (rp) r = *rp;                                     // the only row-table access of the whole walk
  }
  int32_t len = 1;
  unsigned long long reads = 0, dead = 0, fb = 0;
  Bias nobias; nobias.second_order = false; nobias.need_member = false; nobias.p = nobias.q = 1.0f;
  nobias.prev = 0; nobias.prev_sids = nullptr; nobias.prev_deg = 0; nobias.vmin = g.vmin;
  tile[wv][lane][0] = src;
  src = rng_source(g, src);                            // from here on src only keys the Philox stream (compacted ids: the input's id)
  for (int32_t s = 1; s <= L + 1; ++s) {
    const int c = s & (TILE - 1);
    int32_t val = -1;
    if (alive) {
      if (r.deg == 0) {
        alive = false; if (s > 1) ++dead;                      // dead end, RandomWalk.scala:115-120 (acc2 counts the loop only)
      } else {
        if (COMPACT && !(r.flags & ROW_IRREGULAR)) {          // Philox draw on the 2^-24 lattice: 16-byte records
          const uint32_t m = walk_bits24(rng.seed, iter, (uint32_t)src, (uint32_t)s);
          unsigned rd;
          const CfoEnt e = cfo_pick<NT>(g.cfo + r.off, r.deg, m, rd); reads += rd;
          val = e.id; ++len;
          r.off = (int64_t)(e.link & CFO_NOFF_MASK); r.deg = (int32_t)((e.link >> 40) & 0x7FFFFFu);
          r.flags = (e.link >> 63) ? ROW_IRREGULAR : 0u;
        } else {
          float u = draw_uniform(rng, iter, (uint32_t)src, (uint32_t)s);
          FoEnt e;
          if (r.flags & ROW_IRREGULAR) {
            int32_t k = lane_pick_sequential(g.ent + r.off, r.deg, nobias, u);
            ++fb;
            if (COMPACT) {                                   // only the compact table exists: its links are valid for every row
              const CfoEnt ce = g.cfo[r.off + k];
              e.id = ce.id; e.noff = (int64_t)(ce.link & CFO_NOFF_MASK); e.ndeg = (int32_t)((ce.link >> 40) & 0x7FFFFFu);
              e.nflags = (ce.link >> 63) ? ROW_IRREGULAR : 0u;
            } else {
              e = g.fo[r.off + k];
            }
          } else {
            unsigned rd; int32_t k;
            e = fo_pick<NT>(g.fo + r.off, r.deg, u, k, rd); reads += rd;
          }
          val = e.id; ++len;
          r.off = e.noff; r.deg = e.ndeg; r.flags = e.nflags;   // the picked record carries the next row
        }
      }
    }
    tile[wv][lane][c] = val;
    if (c == TILE - 1 || s == L + 1) {
      const int ncols = c + 1;
      const int64_t base_slot = s - c;
      __syncthreads();   // (the tile is private to the wave; a wave-level barrier measured the same, 64.1 vs 64.4 ms)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        int row = rr * 4 + (lane >> 4), col = lane & 15;
        int64_t w = wave_base + row;
        if (col < ncols && w < n_walkers) paths[w * stride + base_slot + col] = tile[wv][row][col];
      }
      __syncthreads();
    }
  }
  if (wi < n_walkers) lens[wi] = len;
  flush_counters(ctr, (unsigned long long)(len - 1), dead, 0, 0, reads, fb);
}

// ---------------------------------------------------------------------------------------------------------
__device__ inline Bias make_bias(const GraphView &g, float p, float q, int32_t prev, bool second_order) {
  Bias b;
  b.p = p; b.q = q; b.prev = prev; b.second_order = second_order;
  b.need_member = second_order && (q != 1.0f);
  b.prev_sids = nullptr; b.prev_deg = 0; b.vmin = g.vmin;
  if (b.need_member) {   // N(prev) through the membership structure (replicated on every shard)
    int64_t s = (int64_t)prev - g.vmin;
    if (s >= 0 && s < g.n_slots) { Row r = g.mrows[s]; b.prev_sids = g.msids + r.off; b.prev_deg = r.deg; b.prev_hub = r.flags >> ROW_HUB_SHIFT; }
  }
  return b;
}

__global__ __launch_bounds__(TPB, 4) void k_walk_general(GraphView g, const int32_t *__restrict__ verts,
                                                      int64_t n_verts, int64_t n_walkers, int32_t L,
                                                      int32_t first_walk, RngSpec rng, float p, float q,
                                                      int32_t *__restrict__ paths, int32_t *__restrict__ lens,
                                                      DevCounters *ctr, unsigned long long *cursor, int32_t tune,
                                                      const int32_t *__restrict__ todo, const unsigned long long *todo_n,
                                                      const int32_t *__restrict__ todo_tie, const ChainRec *__restrict__ tie_list,
                                                      const SWalker *__restrict__ tie_out) {
  __shared__ __attribute__((aligned(16))) uint32_t bitmap[TPB / 64][BINNED_LDS_WORDS];
  const int lane = lane_id();
  Member mem; mem.mode = 0; mem.bm = bitmap[threadIdx.x >> 6]; mem.seg_base = 0;
  mem.ehash = g.ehash; mem.ehash_mask = g.ehash_mask;
#ifdef SRW_PHASE_TIMING
  const unsigned long long t_begin = wall_clock64();
#endif
  const int64_t stride = (int64_t)L + 2;
  unsigned long long steps = 0, degc = 0, degp = 0, fb = 0, dead = 0, fast = 0, srch = 0;
  unsigned long long n_strat[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // SRW_STRAT_*
  // Persistent waves: a walker costs anything from a few to millions of entry reads, and a block's LDS is only
  // released when its slowest wave ends — so every wave takes the next walker from a counter instead of owning one.
  while (true) {
    unsigned long long grab = 0;
    if (lane == 0) grab = atomicAdd(cursor, 1ull);
    int64_t wi = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab));
    int32_t tie_s = -1, tie_k = 0, tie_next = 0;     // the walker's step the chain kernels resolved (k_walk_tables' tie), if any
    if (todo) {                                      // only the walkers k_walk_tables handed over
      if (wi >= (int64_t)*todo_n) break;
      if (todo_tie) {
        const int32_t ci = todo_tie[wi];
        if (ci >= 0) {
          const SWalker o = tie_out[ci];
          if (o.kind == SK_WALKER_RET || o.kind == SK_RET) { tie_s = (int32_t)tie_list[ci].pad; tie_k = o.pad0; tie_next = o.curr; }
        }
      }
      wi = todo[wi];
    } else if (wi >= n_walkers) break;
    int64_t it = wi / n_verts, vi = wi - it * n_verts;
    const uint32_t iter = (uint32_t)(first_walk + it);
    const int32_t src = verts[vi];
    const uint32_t ksrc = (uint32_t)rng_source(g, src);   // Philox key: the input's id of the source
    int32_t *path = paths + wi * stride;
    if (lane == 0) path[0] = src;
    int32_t prev = src, curr = src, len = 1;
    Row rprev; rprev.off = 0; rprev.deg = 0; rprev.flags = 0;
    int64_t eprev = 0;                               // entry index of the edge (prev -> curr) the walker arrived by
    for (int32_t s = 1; s <= L + 1; ++s) {
      // the step's first round trip: the row of curr and the (prev -> curr) pair's table word, issued together
      const bool second = s > 1;
      const int64_t cslot = (int64_t)curr - g.vmin;
      const bool in_range = cslot >= 0 && cslot < g.n_slots;
      const bool want_tab = g.eb_off && second && q != 1.0f && !(tune & 16);
      Row r = g.rows[in_range ? cslot : 0];
      uint32_t eo = EB_NONE;
      if (want_tab) eo = g.eb_off[eprev];
      if (!in_range) { r.off = 0; r.deg = 0; r.flags = 0; }
      if (r.deg == 0) { dead += s > 1; break; }
      Bias b;                                          // N(prev) = last step's row (whole-graph handle)
      b.p = p; b.q = q; b.prev = prev; b.second_order = second; b.need_member = second && (q != 1.0f); b.vmin = g.vmin;
      b.prev_sids = g.sids + rprev.off; b.prev_deg = rprev.deg; b.prev_hub = rprev.flags >> ROW_HUB_SHIFT;
      float u = draw_uniform(rng, iter, ksrc, (uint32_t)s);
      unsigned f = 0, sv = 0;
#ifdef SRW_PHASE_TIMING
      mem.t_step0 = wall_clock64();
#endif
      SRW_T0(mem);
      int32_t k = -1, next = 0;
      bool binned_served = false, have_next = false;
      unsigned which = SRW_STRAT_SCAN;
      if (s == tie_s) { k = tie_k; next = tie_next; have_next = true; binned_served = true; which = SRW_STRAT_EDGE_TABLE; f = 1; n_strat[SRW_STAT_TIES_RESOLVED] += 1; }
      else if (want_tab) {
        if (r.deg <= g.eb_mask_max) {
          // membership mask of the pair (inline for rows up to 32 candidates): no lookup, the row sits in registers
          if (r.deg <= 32 || eo != EB_NONE) {
            k = wave_pick_masked(g, r, b, eo, r.deg > 32 ? g.em_bits + (size_t)eo * 4 : nullptr, u, f, next);
            have_next = true; binned_served = true; which = SRW_STRAT_EDGE_MASK;
            srch += 8ull * (unsigned long long)r.deg + 4ull * (unsigned long long)((r.deg + 31) >> 5);
          }
        } else if (eo != EB_NONE && (r.flags & ROW_PQ_OK)) {
          // chunk prefixes of the pair's corrections: search + one chunk, no intersection
          k = wave_pick_edge_table(g, r, b, g.eb_bins + (size_t)eo * 8, u, f, sv, mem, next, mem.bm + 2 * BIN_CAP);
          if (k >= 0) { have_next = true; binned_served = true; which = SRW_STRAT_EDGE_TABLE; srch += 8ull * EB_BINS; }
        }
      }
      // search over exact prefix sums: a short list of specials (return edges only) when q == 1, position bins else
      if (k < 0 && (!b.need_member || (tune & 16))) {
        k = wave_pick_prefix(g, r, cslot, b, mem.bm, u, f, sv);
        if (k >= 0) which = SRW_STRAT_PREFIX;
      }
      SRW_T1(mem, t_prefix);
      if (k < 0 && !(tune & 16)) {
        unsigned su = 0;
        k = wave_pick_binned(g, r, cslot, b, mem.bm, u, f, sv, tune & 7, (tune & 8) != 0, mem, srch, su, next);
        binned_served = k >= 0;                             // srch: bytes its membership strategy read (bench.py)
        if (k >= 0) { have_next = true; which = su == 1 ? SRW_STRAT_P1 : su == 2 ? SRW_STRAT_P2 : su == 4 ? SRW_STRAT_P3 : SRW_STRAT_W; }
      }
      if (k < 0) { k = wave_pick_scan(g, r, b, mem, u, f); degc += (unsigned long long)r.deg; }
      else { fast += sv; }
      n_strat[which] += 1; n_strat[SRW_STRAT_CHAIN] += f;
      if (!have_next) next = g.ent[r.off + k].id;
#ifdef SRW_PHASE_TIMING
      mem.t_strat[which] += wall_clock64() - mem.t_step0 + (unsigned long long)(next & 0);   // (next: the id load is part of the step)
#endif
      fb += f;
      if (b.need_member && !binned_served) degp += (unsigned long long)b.prev_deg;
      if (lane == 0) path[s] = next;
      prev = curr; curr = next; ++len; rprev = r; eprev = r.off + k;
    }
    for (int64_t t = len + lane; t < stride; t += 64) path[t] = -1;  // unused tail
    if (lane == 0) lens[wi] = len;
    steps += (unsigned long long)(len - 1);
  }
  if (lane == 0) {
    if (steps) atomicAdd(&ctr->steps, steps);
    if (dead) atomicAdd(&ctr->dead_ends, dead);
    if (degc) atomicAdd(&ctr->sum_deg_curr, degc);
    if (degp) atomicAdd(&ctr->sum_deg_prev, degp);
    if (fb) atomicAdd(&ctr->fallbacks, fb);
    if (fast) atomicAdd(&ctr->ent_reads, fast);      // general kernel: steps served by the prefix-sum search
    srch += mem.res_bytes;
    if (srch) atomicAdd(&ctr->trials, srch);         // ... and the bytes the binned ones' membership strategies read
    for (int i = 0; i < 12; ++i) if (n_strat[i]) atomicAdd(&ctr->strat[i], n_strat[i]);
#ifdef SRW_PHASE_TIMING
    const unsigned long long tv[10] = {wall_clock64() - t_begin, mem.t_prefix, mem.t_a, mem.t_p1, mem.t_p2, mem.t_w,
                                       mem.t_fin, mem.t_fill, mem.t_pass1, mem.t_pass2};
    for (int i = 0; i < 10; ++i) atomicAdd(&ctr->dbg[i], tv[i] >> 10);
    atomicAdd(&ctr->dbg[10], mem.n_w); atomicAdd(&ctr->dbg[11], mem.n_w_elems); atomicAdd(&ctr->dbg[12], mem.n_w_windows);
    atomicAdd(&ctr->dbg[13], mem.n_p1); atomicAdd(&ctr->dbg[14], mem.n_p1_elems); atomicAdd(&ctr->dbg[15], mem.n_binned);
    for (int i = 0; i < 12; ++i) atomicAdd(&ctr->dbg[24 + i], mem.t_strat[i] >> 10);
    atomicAdd(&ctr->dbg[16], mem.t_w_lb >> 10); atomicAdd(&ctr->dbg[17], mem.t_w_ins >> 10); atomicAdd(&ctr->dbg[18], mem.t_w_la >> 10); atomicAdd(&ctr->dbg[19], mem.t_w_probe >> 10);
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same walk for the walkers whose EVERY step finds a per-edge table (edge_tables.hip): first step on the raw row,
// then membership masks / chunk-prefix tables only.  Without the on-the-fly samplers the kernel needs a fraction of
// k_walk_general's registers (128 VGPRs + scratch there) and 4 KB of LDS per wave, so more waves hide the dependent
// round trips of a step.  A walker that meets a pair without a table is handed over untouched (its index goes to
// `todo`; k_walk_general redoes it from its first step: the keyed RNG makes that the same path).
// Waves per SIMD of the lean table kernels.  Round 2 (the exact chain still inlined): 4 waves/SIMD 221 M steps/s at config 3, 5
// 239 M, 6 235 M.  Round 3, with the chain out of these kernels (s25): config 3 (edge hash: request-bound) 5 -> 478 M, 6 -> 520 M,
// 7 -> 433 M; config 5's stand-in (row filters, no hash: latency-bound) 5 -> 290 M, 6 -> 326 M, 7 -> 348 M.  With one candidate per
// lane and round of the located chunk (SRW_RESOLVE_PER_LANE 1: fewer registers) and chunks of 64 (s59, A / B / C twice on one box):
// config 3 6 -> 705-719 M, 7 -> 771 M, 8 -> 690 M; config 5's stand-in 7 -> 363 M, 8 -> 387 M.  So by instantiation:
// Round 4, after the arguments left the SGPRs (TabArgs below: 65 VGPRs, no scratch at 7 waves; 64 VGPRs + 12 B at 8) and with SALU the
// busier unit (448 scalar against 302 vector instructions per step, profiles/r04_valu_issue.md): 8 waves 624 ms against 644 at config 3,
// 3 711 against 4 035 ms at config 5's stand-in (profiles/r04_table_kernel_ab_runs.txt, A / B / C twice on one box).
#ifndef SRW_LEAN_WAVES
#define SRW_LEAN_WAVES 8
#endif
// Round 4 (tree tables, 16-bit level 0, 4-byte ids; s125, one box): the row-filter instantiation at 8 waves/SIMD (64 VGPRs, 200 B of scratch
// per lane: 1.5 KB of spill writes per step reach the memory side at config 5's stand-in) 4 734 ms, 7 waves (72 VGPRs, 168 B) 4 637 ms, 6 waves 4 924 ms.
#ifndef SRW_LEAN_WAVES_BF
#define SRW_LEAN_WAVES_BF 8
#endif
// The lean table kernel takes its arguments as ONE struct and reads them again from the kernarg segment where a walker / a step
// needs them (device_common.h:fresh_args) instead of holding their ~130 dwords in SGPRs next to the walker's state.
// -DSRW_TAB_FRESH=0: the arguments stay where the compiler puts them (one copy for the whole kernel).
#ifndef SRW_TAB_FRESH
#define SRW_TAB_FRESH 1
#endif
#if SRW_TAB_FRESH
#define TAB_ARGS() fresh_args<TabArgs>()
#define GFRESH() fresh_graph()
#else
#define TAB_ARGS() a0
#define GFRESH() a0.g
#endif
template <bool BF>   // BF: the located chunk's probes of a long N(prev) go through the row filters (no edge hash; GraphView::bf_off)
__global__ __launch_bounds__(TPB, BF ? SRW_LEAN_WAVES_BF : SRW_LEAN_WAVES) void k_walk_tables(TabArgs a0) {
  __shared__ __attribute__((aligned(16))) uint32_t stage_all[TPB / 64][1024];
  const int lane = lane_id();
  uint32_t *stage = stage_all[threadIdx.x >> 6];
  Member mem; mem.mode = 0; mem.bm = stage; mem.seg_base = 0;
  const int32_t L = a0.L;
  const int64_t stride = (int64_t)L + 2;
  unsigned long long steps = 0, srch = 0;
  uint32_t fb = 0, dead = 0, fast = 0, n_tab = 0, n_mask = 0, n_first = 0;
  while (true) {
    unsigned long long grab = 0;
    const TabArgs aw = TAB_ARGS();                    // (what a walker's start needs)
    if (lane == 0) grab = atomicAdd(aw.cursor, 1ull);
    const int64_t wi = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab));
    if (wi >= aw.n_walkers) break;
    const int64_t it = wi / aw.n_verts, vi = wi - it * aw.n_verts;
    const uint32_t iter = (uint32_t)(aw.first_walk + it);
    const int32_t src = __builtin_amdgcn_readfirstlane(aw.verts[vi]);
    const uint32_t ksrc = (uint32_t)__builtin_amdgcn_readfirstlane(rng_source(aw.g, src));
    int32_t *path = aw.paths + wi * stride;
#ifdef SRW_PATH_BUF
    // the path of the walker, 64 slots at a time, in ONE register (lane t = slot base + t, -1 = unused): a step writes its lane
    // (v_writelane) instead of storing 4 bytes behind an exec mask, a block of 64 slots goes out with one coalesced store
    int32_t pbuf = -1;
    asm("v_writelane_b32 %0, %1, 0" : "+v"(pbuf) : "s"(src));
#else
    if (lane == 0) path[0] = src;
#endif
    int32_t prev = src, curr = src, len = 1;
    Row rprev; rprev.off = 0; rprev.deg = 0; rprev.flags = 0;
    int64_t eprev = 0;
    uint32_t w_fb = 0, w_dead = 0, w_fast = 0, w_srch = 0, w_tab = 0, w_mask = 0;    // (a handed-over walker is not counted here)
    bool handed_over = false;
    int32_t tie_rec = -1;
    WaveDraws draws;
    for (int32_t s = 1; s <= L + 1; ++s) {
      const bool second = s > 1;
      const TabArgs as = TAB_ARGS();                  // (vmin, n_slots, rows, eb_off, the mask geometry, p, q, the seed: what the top of a step needs)
      const GraphView &gs = as.g;
      const int64_t cslot = (int64_t)curr - gs.vmin;
      const bool in_range = cslot >= 0 && cslot < gs.n_slots;
      Row r = gs.rows[in_range ? cslot : 0];
      uint32_t eo = EB_NONE;
      if (second) eo = gs.eb_off[eprev];
      r = uniform_row(r); eo = (uint32_t)__builtin_amdgcn_readfirstlane((int)eo);
      if (!in_range) { r.off = 0; r.deg = 0; r.flags = 0; }
      if (r.deg == 0) { w_dead += s > 1; break; }
      const float u = draws.at(as.rng, iter, ksrc, s, 1);
      unsigned f = 0, sv = 0;
      int32_t k, next = 0;
      // (CHAIN = false: the exact chain is not in this kernel — a draw within rounding distance of a CDF boundary hands the
      //  walker over like a missing table; the chain's registers and scratch cost every step otherwise)
      if (!second) {
        k = uni(wave_pick_first<false>(GFRESH(), r, u, f, next));       // (uni: the pick is the wave's — a loop exit the compiler can see is uniform keeps the walker's state scalar)
        if (k < 0) { handed_over = true; break; }
      } else {
        Bias b;
        b.p = as.p; b.q = as.q; b.prev = prev; b.second_order = true; b.need_member = true; b.vmin = gs.vmin;
        b.prev_sids = gs.sids + rprev.off; b.prev_deg = rprev.deg; b.prev_hub = rprev.flags >> ROW_HUB_SHIFT;
        SRW_T0(mem);
        if (r.deg <= gs.eb_mask_max && (r.deg <= 32 || eo != EB_NONE)) {
#ifdef SRW_MASK1
          if (r.deg <= 64) k = uni(wave_pick_masked<false, 1>(GFRESH(), r, b, eo, r.deg > 32 ? gs.em_bits + (size_t)eo * 4 : nullptr, u, f, next));
          else
#endif
          k = uni(wave_pick_masked<false>(GFRESH(), r, b, eo, r.deg > 32 ? gs.em_bits + (size_t)eo * 4 : nullptr, u, f, next));
          w_mask += 1; w_srch += 8u * (uint32_t)r.deg + 4u * (uint32_t)((r.deg + 31) >> 5);
          SRW_T1(mem, t_a);
        } else if (r.deg > gs.eb_mask_max && eo != EB_NONE && (r.flags & ROW_PQ_OK)) {
          double S_tie = 0.0;
          k = uni(wave_pick_edge_table<BF, false>(GFRESH(), r, b, gs.eb_bins + (size_t)eo * 8, u, f, sv, mem, next, stage, &S_tie));
          if (k >= 0) { w_tab += 1; w_srch += 8u * EB_BINS; w_fast += sv; }
          else if (k == CHAIN_NEEDED && as.tie.list) {   // a tie on a table step: its exact chain by the chain kernels (the whole GPU)
            const TieSink tie = TAB_ARGS().tie;
            if (lane == 0) {
              const unsigned long long c = atomicAdd(tie.cur, 1ull);
              if (c < (unsigned long long)CHAIN_CAP) {
                tie_rec = (int32_t)c;
                WWalker wr; wr.lw = (int32_t)it; wr.src = src; wr.prev = prev; wr.curr = curr; tie.recs[c] = wr;
                ChainRec cr; cr.ri = (uint32_t)c; cr.pad = (uint32_t)s; cr.S = S_tie; tie.list[c] = cr;
                atomicAdd(tie.hdr, 1u);
              }
            }
          }
          SRW_T1(mem, t_p1);
#ifdef SRW_PHASE_TIMING
          if (rprev.deg > 1024) mem.t_p2 += wall_clock64() - mem.t_mark;      // ... of which steps with a long N(prev)
#endif
        } else k = -1;
        k = uni(k);                                        // (after the lane-0 region above: its join would make the pick look divergent)
        if (k < 0) { handed_over = true; break; }          // no table for this pair: the general kernel takes the walker
      }
      next = uni(next);
      w_fb += f;
#ifdef SRW_PATH_BUF
      asm("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(pbuf) : "s"(next), "s"(s & 63) : "m0");      // (one SGPR operand per instruction: the lane index goes through m0)
      if ((s & 63) == 63) { path[(s & ~63) + lane] = pbuf; pbuf = -1; }      // (a full block: s <= L + 1 < stride)
#else
      if (lane == 0) path[s] = next;
#endif
      prev = curr; curr = next; ++len; rprev = r; eprev = r.off + k;
    }
    if (handed_over) {
      const TabArgs ah = TAB_ARGS();
      if (lane == 0) {
        const unsigned long long t = atomicAdd(ah.todo_n, 1ull);
        ah.todo[t] = (int32_t)wi;
        if (ah.tie.todo_tie) ah.tie.todo_tie[t] = tie_rec;
        atomicAdd(&ah.ctr->strat[SRW_STAT_HANDED_OVER], 1ull);
      }
      continue;
    }
#ifdef SRW_PATH_BUF
    {                                                 // the block the walk ended in (its unused lanes are the tail's -1), then the rest of the tail
      const int64_t b0 = (int64_t)(len & ~63);
      if (b0 + lane < stride) path[b0 + lane] = pbuf;
      for (int64_t t = b0 + 64 + lane; t < stride; t += 64) path[t] = -1;
    }
#else
    for (int64_t t = len + lane; t < stride; t += 64) path[t] = -1;  // unused tail
#endif
    if (lane == 0) TAB_ARGS().lens[wi] = len;
    steps += (unsigned long long)(len - 1); n_first += len > 1 ? 1u : 0u;
    fb += w_fb; dead += w_dead; fast += w_fast; srch += w_srch; n_tab += w_tab; n_mask += w_mask;
  }
  if (lane == 0) {
    DevCounters *ctr = TAB_ARGS().ctr;
    srch += mem.res_bytes;
    if (steps) atomicAdd(&ctr->steps, steps);
    if (dead) atomicAdd(&ctr->dead_ends, (unsigned long long)dead);
    if (fb) { atomicAdd(&ctr->fallbacks, (unsigned long long)fb); atomicAdd(&ctr->strat[SRW_STRAT_CHAIN], (unsigned long long)fb); }
    if (fast) atomicAdd(&ctr->ent_reads, (unsigned long long)fast);
    if (srch) atomicAdd(&ctr->trials, srch);
    if (n_tab) atomicAdd(&ctr->strat[SRW_STRAT_EDGE_TABLE], (unsigned long long)n_tab);
    if (n_mask) atomicAdd(&ctr->strat[SRW_STRAT_EDGE_MASK], (unsigned long long)n_mask);
    if (n_first) atomicAdd(&ctr->strat[SRW_STRAT_SCAN], (unsigned long long)n_first);
#ifdef SRW_PHASE_TIMING
    // lean kernel: dbg[1] mask-step time, [2] table-step time, [3] ... with deg(prev) > 1024, [10..15] resolve statistics
    atomicAdd(&ctr->dbg[1], mem.t_a >> 10); atomicAdd(&ctr->dbg[2], mem.t_p1 >> 10); atomicAdd(&ctr->dbg[3], mem.t_p2 >> 10);
    atomicAdd(&ctr->dbg[10], mem.n_w); atomicAdd(&ctr->dbg[11], mem.n_w_elems); atomicAdd(&ctr->dbg[12], mem.n_w_windows);
    atomicAdd(&ctr->dbg[13], mem.n_p1); atomicAdd(&ctr->dbg[14], mem.n_p1_elems); atomicAdd(&ctr->dbg[15], mem.n_binned);
    atomicAdd(&ctr->dbg[16], mem.t_fin);
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------
// The second-order step of the q = 1 per-lane kernels (k_walk_q1, k_sh_step_q1): first k that is not a certain miss under the exact
// prefix sums PQ + the return edges' corrections.  rv / rv_pos0 / rv_w0: the pair's return-edge record (RevEnt: count << 24 | index in
// curr's sorted row, input-order position and weight of the first one).  0: picked (k, e = its compact record); 1: non-positive sum,
// 2: a draw within rounding distance of a CDF boundary — the caller hands the walker to the general sampler (or, with the row's
// sum in *S_out, to the chain kernels).
template <bool NT>
__device__ inline int q1_pick(const GraphView &g, const Row &r, const CfoEnt *crow, uint32_t rv, int32_t rv_pos0, float rv_w0, int32_t prev_id,
                              uint32_t m, float p, CfoEnt &e, int32_t &k, unsigned long long &reads, double *S_out = nullptr) {
  const PqRow PQ(g, r.off);
  int32_t rp[REV_MAX_RETURNS]; double rc[REV_MAX_RETURNS];      // return edges: input-order position, correction
  int nr = 0;
  double corr_all = 0.0;
  int64_t so = 0;                                   // first return edge in curr's sorted row
  if (rv != REV_NONE) {
    nr = (int)(rv >> 24);
    so = r.off + (int64_t)(rv & 0xFFFFFFu);
    if (nr >= 255) {                                 // the count saturated (hub <-> hub multi-edges): count the run of prev
      const uint32_t xprev = (uint32_t)((int64_t)prev_id - g.vmin);
      const int64_t row_end = r.off + r.deg;
      while (so + nr < row_end && g.sids[so + nr] == xprev) ++nr;
    }
#pragma unroll
    for (int i = 0; i < REV_MAX_RETURNS; ++i) {
      rp[i] = r.deg; rc[i] = 0.0;
      if (i < nr) {
        float w;
        if (i == 0) { rp[0] = rv_pos0; w = rv_w0; }             // the first return edge travels with rev[e]
        else { rp[i] = (int32_t)g.sperm[so + i]; w = g.sw[so + i]; }
        rc[i] = (double)div_exact(w, p) - (double)w; corr_all += rc[i];
      }
    }
    for (int i = REV_MAX_RETURNS; i < nr; ++i) { const float w = g.sw[so + i]; corr_all += (double)div_exact(w, p) - (double)w; }   // (small graphs: dozens of duplicates between hubs)
  } else {
#pragma unroll
    for (int i = 0; i < REV_MAX_RETURNS; ++i) { rp[i] = r.deg; rc[i] = 0.0; }
  }
  {
    const double S0 = PQ[r.deg - 1], S = S0 + corr_all;
    const double pS = (double)m * 0x1p-24 * S;
    auto corr_upto = [&](int32_t kk) {
      double a = 0.0;
#pragma unroll
      for (int i = 0; i < REV_MAX_RETURNS; ++i) a += (rp[i] <= kk) ? rc[i] : 0.0;       // exact under the certificate
      for (int i = REV_MAX_RETURNS; i < nr; ++i)
        if ((int32_t)g.sperm[so + i] <= kk) { const float w = g.sw[so + i]; a += (double)div_exact(w, p) - (double)w; }
      return a;
    };
    auto numer = [&](int32_t kk) { return PQ[kk] + corr_upto(kk); };
    auto not_miss = [&](int32_t kk, double num) { return !(num * (1.0 + (double)(kk + 8) * 0x1p-51) < pS); };
    // start position: the guide entry of the bucket the target falls into in UNBIASED units (any start is
    // correct, the loops below decide with the exact sums; a good one makes them O(1))
    auto guide_start = [&](double tau, bool &ok) {
      double f = tau / S0;
      f = f < 0.0 ? 0.0 : (f > 0.99999994 ? 0.99999994 : f);
      const uint32_t mm = (uint32_t)(f * 16777216.0);
      const uint32_t j = (uint32_t)(((uint64_t)mm * (uint64_t)(uint32_t)r.deg) >> 24);
      const CfoEnt ge = load_cfo<NT>(crow + j); ++reads;
      const int32_t gd = cfo_delta(ge.cg, ge.link);
      if (gd == CFO_GD_SAT) { ok = false; return 0; }      // no guide for this bucket: bisection below
      const int32_t st = (int32_t)j - gd;
      return st < 0 ? 0 : (st >= r.deg ? r.deg - 1 : st);
    };
    bool ok = S > 0.0 && S0 > 0.0;
    int32_t k0 = 0;
    if (ok) {
      k0 = guide_start(pS, ok);
      const double cb = ok ? corr_upto(k0) : 0.0;
      if (ok && cb != 0.0) {                           // past a return edge: its correction moves the answer
        int32_t first_r = r.deg;
#pragma unroll
        for (int i = 0; i < REV_MAX_RETURNS; ++i) first_r = rp[i] < first_r ? rp[i] : first_r;
        for (int i = REV_MAX_RETURNS; i < nr; ++i) { const int32_t q_ = (int32_t)g.sperm[so + i]; first_r = q_ < first_r ? q_ : first_r; }
        const int32_t k1 = guide_start(pS - cb, ok);
        k0 = k1 < first_r ? first_r : k1;
      }
    }
    const bool usable = S > 0.0 && S0 > 0.0;
    if (!usable) return 1;
    else if (S_out && g.dbg_chain_deg && r.deg >= g.dbg_chain_deg) { *S_out = S; return 2; }
    else {
      // first k that is not a certain miss (A' is non-decreasing, the tolerance grows with k: monotone) — a few
      // steps from the guide's start, else (saturated guide entry, many parallel return edges) by bisection
      int guard = 0;
      double nk = 0.0;
      if (ok) {
        nk = numer(k0);
        bool nm = not_miss(k0, nk);
        while (nm && k0 > 0 && guard < 12) {             // step back while the predecessor is not a certain miss either
          const double np = numer(k0 - 1);
          if (!not_miss(k0 - 1, np)) break;
          --k0; nk = np; ++guard;
        }
        while (!nm && guard < 12) {                      // step forward to the first not-certain-miss
          ++k0; ++guard;
          if (k0 >= r.deg) break;
          nk = numer(k0); nm = not_miss(k0, nk);
        }
        reads += (unsigned)guard + 1u;
      }
      if (!ok || guard >= 12) {
        int32_t lo = 0, hi = r.deg;
        while (lo < hi) {
          const int32_t mid = lo + ((hi - lo) >> 1);
          if (not_miss(mid, numer(mid))) hi = mid; else lo = mid + 1;
          ++reads;
        }
        k0 = lo;
        if (k0 < r.deg) nk = numer(k0);
      }
      if (k0 >= r.deg) { k = 0; e = load_cfo<NT>(crow); ++reads; }                    // no crossing: edges.head (:24)
      else if (nk * (1.0 - (double)(k0 + 8) * 0x1p-51) >= pS) { k = k0; e = load_cfo<NT>(crow + k); ++reads; }   // a certain hit
      else { if (S_out) *S_out = S; return 2; }   // a draw within rounding distance of a boundary: exact chain (S: the reference's sum, exact under the certificate)
    }
  }
  return 0;
}

// p != 1, q == 1: the only biased candidates are the return edges, so a step is a first-order step plus one
// correction — one walker per LANE, like k_walk_first_order (r01 ran this case one walker per wave: 4.6e8 steps/s).
// Per step, under the row certificate of sampler_tables.hip (every prefix sum exact):
//   A'_k = PQ[k] + sum over the return edges r_i <= k of c_i,  r_i from rev[e] (multi-edges: four in registers, more by loop), c_i = fl(w_i / p) - w_i,
//   S = PQ[deg-1] + sum of all c_i
// and the reference's acc_k = sum of fl(w'_i / S) differs from A'_k / S by at most (k + 2) u A'_k / S — the certified
// divide-free compares of binned_resolve.  The first k that is not a certain miss is located from the first-order guide
// table (a start position; the exact prefix sums decide; a saturated guide entry or a start more than a few positions
// off — rows with many parallel return edges — is replaced by a bisection) and must be a certain hit, else the walker is
// handed over to k_walk_general, as are walkers that meet an irregular row.
// The picked compact record carries the next row descriptor, as in k_walk_first_order.
template <bool NT>
__global__ __launch_bounds__(TPB) void k_walk_q1(GraphView g, const int32_t *__restrict__ verts, int64_t n_verts,
                                                 int64_t n_walkers, int32_t L, int32_t first_walk, RngSpec rng, float p,
                                                 int32_t *__restrict__ paths, int32_t *__restrict__ lens, DevCounters *ctr,
                                                 int32_t *__restrict__ todo, unsigned long long *todo_n, uint32_t max_ret) {
  __shared__ int32_t tile[TPB / 64][64][TILE + 1];
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const int64_t wi = blockIdx.x * (int64_t)TPB + threadIdx.x;
  const int64_t wave_base = wi - lane;
  const int64_t stride = (int64_t)L + 2;
  bool alive = wi < n_walkers, handed = false;
  uint32_t iter = 0; int32_t src = 0;
  Row r; r.off = 0; r.deg = 0; r.flags = 0;
  if (alive) {
    const int64_t it = wi / n_verts, vi = wi - it * n_verts;
    iter = (uint32_t)(first_walk + it);
    src = verts[vi];
    const Row *rp = row_of(g, src);
    if (rp) r = *rp;
  }
  int32_t len = 1;
  int64_t eprev = 0;
  int32_t prev_id = src, curr_id = src;
  unsigned long long reads = 0, dead = 0;
  tile[wv][lane][0] = src;
  src = rng_source(g, src);                            // from here on src only keys the Philox stream
  for (int32_t s = 1; s <= L + 1; ++s) {
    const int c = s & (TILE - 1);
    int32_t val = -1;
    bool big = false; uint32_t big_rv = 0u, big_m = 0u;
    if (alive) {
      if (r.deg == 0) {
        alive = false; if (s > 1) ++dead;
      } else if (r.flags & ROW_IRREGULAR) {
        alive = false; handed = true; atomicAdd(&ctr->why[0], 1ull);
      } else {
        const uint32_t m = walk_bits24(rng.seed, iter, (uint32_t)src, (uint32_t)s);
        const CfoEnt *crow = g.cfo + r.off;
        CfoEnt e; int32_t k = -1;
        if (s == 1) {                               // initFirstStep: raw weights = the first-order draw
          unsigned rd; e = cfo_pick<NT>(crow, r.deg, m, rd, k); reads += rd;
        } else {
          const int4v rv4 = NT ? __builtin_nontemporal_load(reinterpret_cast<const int4v *>(g.rev + eprev))
                               : *reinterpret_cast<const int4v *>(g.rev + eprev);
          if ((uint32_t)rv4.x != REV_NONE && ((uint32_t)rv4.x >> 24) > max_ret) { big = true; big_rv = (uint32_t)rv4.x; big_m = m; }
          else {
            const int why = q1_pick<NT>(g, r, crow, (uint32_t)rv4.x, rv4.y, __int_as_float(rv4.z), prev_id, m, p, e, k, reads);
            if (why) { alive = false; handed = true; atomicAdd(&ctr->why[why], 1ull); }
          }
        }
        if (alive && !big) {
          val = e.id; ++len;
          prev_id = curr_id; curr_id = val;
          eprev = r.off + k;
          r.off = (int64_t)(e.link & CFO_NOFF_MASK); r.deg = (int32_t)((e.link >> 40) & 0x7FFFFFu);
          r.flags = (e.link >> 63) ? ROW_IRREGULAR : 0u;
        }
      }
    }
    // Steps with many parallel return edges (a hub's self-loops, hub <-> hub multi-edges: up to hundreds): q1_pick walks the run once
    // per prefix value in ONE lane while 63 wait — the wave takes them one at a time instead (wave_pick_returns).
    unsigned long long mb = __ballot(big);
    while (mb) {
      const int l = __ffsll((long long)mb) - 1;
      mb &= mb - 1ull;
      Row rr;
      rr.off = (int64_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)r.off >> 32), l) << 32) |
                         (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)r.off, l));
      rr.deg = __builtin_amdgcn_readlane(r.deg, l); rr.flags = ROW_PQ_OK;          // (q1 runs only when every row holds the certificate)
      const int32_t pv = __builtin_amdgcn_readlane(prev_id, l);
      const uint32_t rvl = (uint32_t)__builtin_amdgcn_readlane((int)big_rv, l);
      const uint32_t ml = (uint32_t)__builtin_amdgcn_readlane((int)big_m, l);
      const int64_t so = rr.off + (int64_t)(rvl & 0xFFFFFFu);
      int32_t nr = (int32_t)(rvl >> 24);
      if (nr >= 255) {                                     // the count saturated: the run of prev in the sorted row
        const uint32_t xprev = (uint32_t)((int64_t)pv - g.vmin);
        nr = 0;
        for (int64_t cc = so;; cc += 64) {
          const unsigned long long mm = __ballot(cc + lane < rr.off + rr.deg && g.sids[cc + lane] == xprev);
          nr += __popcll(mm);
          if (mm != ~0ull) break;
        }
      }
      Bias b; b.p = p; b.q = 1.0f; b.prev = pv; b.second_order = true; b.need_member = false; b.vmin = g.vmin;
      b.prev_sids = nullptr; b.prev_deg = 0; b.prev_hub = 0;
      unsigned f = 0;
      const int32_t kk = wave_pick_returns<false>(g, rr, b, so, nr, (float)ml * (1.0f / 16777216.0f), f);
      if (lane == l) {
        big = false;
        if (kk < 0) { alive = false; handed = true; atomicAdd(&ctr->why[kk == CHAIN_NEEDED ? 2 : 1], 1ull); }
        else {
          const CfoEnt e = load_cfo<NT>(g.cfo + r.off + kk); ++reads;
          val = e.id; ++len;
          prev_id = curr_id; curr_id = val;
          eprev = r.off + kk;
          r.off = (int64_t)(e.link & CFO_NOFF_MASK); r.deg = (int32_t)((e.link >> 40) & 0x7FFFFFu);
          r.flags = (e.link >> 63) ? ROW_IRREGULAR : 0u;
        }
      }
    }
    tile[wv][lane][c] = val;
    if (c == TILE - 1 || s == L + 1) {
      const int ncols = c + 1;
      const int64_t base_slot = s - c;
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        int row = rr * 4 + (lane >> 4), col = lane & 15;
        int64_t w = wave_base + row;
        if (col < ncols && w < n_walkers) paths[w * stride + base_slot + col] = tile[wv][row][col];
      }
      __syncthreads();
    }
  }
  if (wi < n_walkers) {
    if (handed) { todo[atomicAdd(todo_n, 1ull)] = (int32_t)wi; atomicAdd(&ctr->strat[SRW_STAT_HANDED_OVER], 1ull); }   // k_walk_general redoes it from its first step
    else lens[wi] = len;
  }
  const unsigned long long my_steps = handed ? 0ull : (unsigned long long)(len - 1);
  flush_counters(ctr, my_steps, handed ? 0ull : dead, 0, 0, reads, 0);
  const unsigned long long tot = wave_sum_u64(my_steps);
  if (lane == 0 && tot) atomicAdd(&ctr->strat[SRW_STRAT_Q1_LANE], tot);
}

// ---------------------------------------------------------------------------------------------------------
// Mode A: per-vertex alias draw + rejection for the p/q bias (KnightKing-style), one walker per lane, each lane
// its own (step, trial) state machine so that lanes do not wait for each other's rejections.  Spec shared with
// oracle/srw_oracle.c:alias_pick — trial t of step s draws Philox(ctr = (iter, src, s, t), key = (seed, 0xA11A5)):
// slot j = ((x0:x1) * deg) >> 64, coin u2 = (x2 >> 8) 2^-24 keeps j or takes alias[j]; a second-order step accepts
// iff u3 * Q < bias, u3 = (x3 >> 8) 2^-24, Q = max(1, 1/q), bias = 1/p | 1 | 1/q; when 1/p > Q the excess
// (1/p - Q) * w of the return edge(s) is sampled by an appendix branch chosen by area (KnightKing's outlier folding).
// A trial needs only the first 16 bytes of the record (prob, alias, id, reverse weight); the link to the next row is
// fetched when the trial is ACCEPTED (same 64-byte sector: an L2 hit) — rejected trials return half the bytes.
template <bool NT>
__device__ inline AEnt load_al_head(const AEnt *p) {
  const int4v *q = reinterpret_cast<const int4v *>(p);
  const int4v a = NT ? __builtin_nontemporal_load(q) : *q;
  AEnt e;
  e.prob = __int_as_float(a.x); e.alias = a.y; e.id = a.z; e.wrev = __int_as_float(a.w);
  e.noff = 0; e.ndeg = 0; e.nflags = 0;
  return e;
}
template <bool NT>
__device__ inline void load_al_tail(const AEnt *p, AEnt &e) {
  const int4v *q = reinterpret_cast<const int4v *>(p) + 1;
  const int4v b = NT ? __builtin_nontemporal_load(q) : *q;
  e.noff = (int64_t)(((uint64_t)(uint32_t)b.y << 32) | (uint32_t)b.x);
  e.ndeg = b.z; e.nflags = (uint32_t)b.w;
}

template <bool NT>
__global__ __launch_bounds__(TPB, 6) void k_walk_alias(GraphView g, const int32_t *__restrict__ verts, int64_t n_verts,
                                                    int64_t n_walkers, int32_t L, int32_t first_walk, uint32_t seed,
                                                    float p, float q, int32_t *__restrict__ paths,
                                                    int32_t *__restrict__ lens, DevCounters *ctr) {
  // Lanes reject independently, so they are not in step: each lane stages its own 16 path slots in LDS and writes
  // them as one 64-byte run (16 scattered 4-byte stores per run cost 16 L2-miss-path requests instead of 1).
  __shared__ int32_t stage[TPB][TILE + 1];
  int32_t *buf = stage[threadIdx.x];
  const int64_t wi = blockIdx.x * (int64_t)TPB + threadIdx.x;
  unsigned long long reads = 0, dead = 0, fb = 0, trials = 0;
  int32_t len = 0;
  if (wi < n_walkers) {
    const int64_t stride = (int64_t)L + 2;
    const int64_t it = wi / n_verts, vi = wi - it * n_verts;
    const uint32_t iter = (uint32_t)(first_walk + it);
    const int32_t src = verts[vi];
    const uint32_t ksrc = (uint32_t)rng_source(g, src);   // Philox key: the input's id of the source
    int32_t *path = paths + wi * stride;
    Row rc; rc.off = 0; rc.deg = 0; rc.flags = 0;
    { const Row *rp0 = row_of(g, src); if (rp0) rc = *rp0; }
    Row rp = rc;
    int32_t curr = src, prev = src;
    buf[0] = src; len = 1;
    const float inv_p = 1.0f / p, inv_q = 1.0f / q;
    const float Q = inv_q > 1.0f ? inv_q : 1.0f;                  // envelope WITHOUT the return edge
    const bool biased_cfg = !(p == 1.0f && q == 1.0f);
    int32_t s = 1; uint32_t t = 0;
    // outlier folding state of the current step (valid while t > 0)
    bool fold = false; double fa = 0.0, ftot = 0.0, fw = 0.0; int32_t flo = -1, fhi = -1;
    float wprev_hint = __int_as_float(0x7FC00000);   // W_prev carried by the record that led here (NaN: unknown)
    while (s <= L + 1) {
      if (rc.deg == 0) { if (s > 1) ++dead; break; }
      const bool second = s > 1, biased = second && biased_cfg;
      uint32_t o[4];
      philox4x32_10(iter, ksrc, (uint32_t)s, t, seed, 0xA11A5u, o);
      AEnt e;
      const AEnt *ep = nullptr;                 // record whose link is still to be fetched
      bool accepted = true;
      if (rc.flags & ROW_ALIAS_IRREGULAR) {     // not alias-regular: the reference's CDF inversion, literally
        Bias b; b.p = p; b.q = q; b.prev = prev; b.second_order = second; b.need_member = second && q != 1.0f;
        b.prev_sids = g.sids + rp.off; b.prev_deg = rp.deg; b.vmin = g.vmin;
        float u = (float)(o[2] >> 8) * (1.0f / 16777216.0f);
        int32_t k = lane_pick_sequential(g.ent + rc.off, rc.deg, b, u);
        e = g.al[rc.off + k]; ++fb;
      } else {
        if (t == 0) {            // once per step: does the return edge stick out of the envelope?
          fold = false;
          if (biased && inv_p > Q) {
            flo = -1; fhi = -1;
            if (wprev_hint == wprev_hint) {
              fw = (double)wprev_hint;          // exact: precomputed at build time with the same f64 sum
            } else {
              // occurrences of prev in N(curr): equal range in the sorted row; weights in input order via sperm
              const uint32_t *cs = g.sids + rc.off;
              const uint32_t x = (uint32_t)((int64_t)prev - g.vmin);
              int32_t lo = 0, hi = rc.deg;
              while (lo < hi) { int32_t mid = lo + ((hi - lo) >> 1); if (cs[mid] < x) lo = mid + 1; else hi = mid; }
              flo = lo; fhi = lo; fw = 0.0;
              while (fhi < rc.deg && cs[fhi] == x) { fw += (double)g.ent[rc.off + g.sperm[rc.off + fhi]].w; ++fhi; }
            }
            if (fw > 0.0) {
              const double S = g.rsum[(int64_t)curr - g.vmin];
              fa = ((double)inv_p - (double)Q) * fw; ftot = (double)Q * S + fa; fold = true;
            }
          }
        }
        bool appendix = false;
        if (fold) {
          uint32_t y[4];
          philox4x32_10(iter, ksrc, (uint32_t)s, t, seed, 0xA11A6u, y);
          const float u5 = (float)(y[0] >> 8) * (1.0f / 16777216.0f);
          if ((double)u5 * ftot < fa) {       // appendix: return to prev; occurrence ~ w in input order
            appendix = true;
            if (flo < 0) {                      // the occurrences were not needed until now: find them
              const uint32_t *cs = g.sids + rc.off;
              const uint32_t x = (uint32_t)((int64_t)prev - g.vmin);
              int32_t lo = 0, hi = rc.deg;
              while (lo < hi) { int32_t mid = lo + ((hi - lo) >> 1); if (cs[mid] < x) lo = mid + 1; else hi = mid; }
              flo = lo; fhi = lo;
              while (fhi < rc.deg && cs[fhi] == x) ++fhi;
            }
            const float u6 = (float)(y[1] >> 8) * (1.0f / 16777216.0f);
            const double target = (double)u6 * fw;
            double cum = 0.0; int32_t pos = g.sperm[rc.off + flo];
            for (int32_t c = flo; c < fhi; ++c) {
              pos = (int32_t)g.sperm[rc.off + c];
              cum += (double)g.ent[rc.off + pos].w;
              if (cum >= target) break;
            }
            ep = g.al + rc.off + pos; e = load_al_head<NT>(ep); ++reads;
          }
        }
        ++trials;
        if (!appendix) {
          const uint64_t r64 = ((uint64_t)o[0] << 32) | o[1];
          const int64_t j = (int64_t)__umul64hi(r64, (uint64_t)(uint32_t)rc.deg);
          ep = g.al + rc.off + j; e = load_al_head<NT>(ep); ++reads;
          const float u2 = (float)(o[2] >> 8) * (1.0f / 16777216.0f);
          if (!(u2 < e.prob)) { ep = g.al + rc.off + e.alias; e = load_al_head<NT>(ep); ++reads; }
          if (biased) {
            const float u3 = (float)(o[3] >> 8) * (1.0f / 16777216.0f);
            const float thr = u3 * Q;
            if (e.id == prev) accepted = thr < inv_p;
            else if (thr < fminf(inv_q, 1.0f)) accepted = true;   // accepted whether or not x is in N(prev): no lookup
            else {
              // x in N(prev)?  On an undirected load this equals prev in N(x): probe the shorter sorted row
              bool in;
              if (g.ehash)                                  // one probe into the edge hash set: (prev -> x) exists?
                in = edge_exists(g.ehash, g.ehash_mask, (uint32_t)((int64_t)prev - g.vmin), (uint32_t)((int64_t)e.id - g.vmin));
              else {
                if (ep) { load_al_tail<NT>(ep, e); ep = nullptr; }     // the shorter-row choice needs the candidate's row
                if (g.symmetric && e.ndeg < rp.deg)
                  in = sorted_contains(g.sids + e.noff, e.ndeg, (uint32_t)((int64_t)prev - g.vmin));
                else
                  in = sorted_contains(g.sids + rp.off, rp.deg, (uint32_t)((int64_t)e.id - g.vmin));
              }
              accepted = thr < (in ? 1.0f : inv_q);
            }
            accepted = accepted || (t + 1u >= 65536u);
          }
        }
      }
      if (accepted) {
        if (ep) load_al_tail<NT>(ep, e);
        buf[s & (TILE - 1)] = e.id;
        if ((s & (TILE - 1)) == TILE - 1) {
          int32_t *dst = path + (s - (TILE - 1));
          if ((stride & 1) == 0) {                           // rows 8-byte aligned: 8-byte stores
#pragma unroll
            for (int c = 0; c < TILE; c += 2) {
              int2 v; v.x = buf[c]; v.y = buf[c + 1];
              *reinterpret_cast<int2 *>(dst + c) = v;
            }
          } else {                                           // back-to-back: the line is still in L2 when the last one lands
#pragma unroll
            for (int c = 0; c < TILE; ++c) dst[c] = buf[c];
          }
        }
        wprev_hint = e.wrev;
        prev = curr; rp = rc; curr = e.id;
        rc.off = e.noff; rc.deg = e.ndeg; rc.flags = e.nflags;
        ++s; ++len; t = 0;
      } else {
        ++t;
      }
    }
    for (int32_t t2 = len & ~(TILE - 1); t2 < len; ++t2) path[t2] = buf[t2 & (TILE - 1)];   // the partial last run
    for (int64_t t2 = len; t2 < stride; ++t2) path[t2] = -1;
    lens[wi] = len;
  }
  unsigned long long steps = len > 0 ? (unsigned long long)(len - 1) : 0ull;
  flush_counters(ctr, steps, dead, 0, 0, reads, fb);
  trials = wave_sum_u64(trials);
  if (lane_id() == 0 && trials) atomicAdd(&ctr->trials, trials);
}

// ---------------------------------------------------------------------------------------------------------
// Vertex-sharded walk (SURVEY §8e option 2; replaces transferWalkersToTheirPartitions, RandomWalk.scala:92-93,186-192,
// and UniformRandomWalk.prepareWalkersToTransfer, UniformRandomWalk.scala:103-112).
//
// A walker standing on v is processed by owner(v); its PATH lives on its home rank = owner(source).  What moves between
// ranks each super-step is fixed-size records in fixed-capacity CHUNKS, one chunk per (sender, receiver) pair:
//     chunk = { u32 n_walkers, n_rets, 0, 0 } | WWalker[cap_w] {lw, src, prev, curr} | WRet[cap_r] {lw, v}
// 16 + 8 = 24 bytes per walker-step on the wire (the reference ships the whole path so far and N(prev) with every walker,
// RandomWalk.scala:135).  lw = (local index of the source vertex on its home rank) * batch + (walk iteration inside the
// batch): the home rank's path row, and lw % batch is the RNG's iteration word.  Every sampled vertex goes home at once as
// an 8-byte return; its path slot is IMPLICIT: a return produced by super-step s belongs to slot s (the receiver applies
// the returns of the chunk it got after super-step s).  A return with the top bit of lw set is a death notice (the walker
// stopped before sampling slot s: its path has s entries): lens start at walk_length + 2 and only walkers that stop early
// are corrected.  On a linked p = q = 1 walk prev | curr << 32 is the row link of the vertex the walker stands on.
// (Round 2 carried the last three vertices inside a 32-byte walker and returned four slots at a time as 24 bytes: 38 bytes
// per walker-step — the exchange, not the kernels, bounded a shard on xGMI)
// A rank's receive buffer is `world` chunks (one per sender), its send side is `world` destination pointers — the
// local send buffer (one equal-split all_to_all_single moves it, distributed.py) or, inside one process, the peers'
// receive buffers themselves (xGMI peer stores, cluster.cpp).
// Everything is sized and counted on the device: NO host synchronisation per super-step; an overflowing chunk drops
// its surplus and raises a flag the host reads once per batch (the batch is then redone with more slack).
//   k_sh_seed    : the rank's own walkers, spread over the chunks of its receive buffer; path slot 0, lens = L + 2
//   k_sh_apply   : returns of the previous super-step -> their path slot; death notices -> lens
//   k_sh_step(_fo): sample every incoming walker once into `scratch` (kind says what the bucket kernel must emit), count
//                  the block's survivors per destination owner and the returns per home rank in LDS -> blk[b][2 * world]
//   k_sh_offsets : one block: scan of blk over the blocks -> every block's write cursors; chunk headers
//   k_sh_bucket  : re-reads the slice, writes walkers to chunk[owner(next)] and path returns to chunk[home(src)]
// The general kernel keeps one wave per record and the same samplers as k_walk_general (bit-identical paths for any
// world, asserted against the oracle).
constexpr int SHARD_MAX_WORLD = 64;
struct alignas(8) WRet { int32_t lw, v; };                         // on the wire: 8 bytes; lw top bit: death notice
// a record between the sampling kernel and the bucketing kernel (scratch, never on the wire): the forwarded walker
// (prev, curr), the vertex that goes home (v) and what to emit (kind)
constexpr int64_t SW_BYTES = 16, PR_BYTES = 8;
struct ShardIO {
  const char *recv;        // world chunks, one per sender
  int64_t chunk_bytes;
  int32_t cap_w, cap_r, world, rank, batch;
  // this rank's path staging (slot-major) and lengths: a return whose home is THIS rank is applied where it is produced
  // (no record): all of them at super-step 1 (every walker starts at home: n_local returns into one chunk otherwise —
  // world times the capacity an even spread needs), 1 / world of them later, every one at world 1
  int32_t *pt, *lens;
  int64_t n_rows;
};
__device__ inline void shard_return_home(const ShardIO &io, const SWalker &w, int32_t step) {
  if (w.kind == SK_DEAD) io.lens[w.lw] = step;
  else io.pt[(int64_t)step * io.n_rows + w.lw] = w.v;
}
struct ShardDst { char *p[SHARD_MAX_WORLD]; };   // where chunk (me -> d) is written
__device__ inline const uint32_t *chunk_hdr(const char *base, int64_t cb, int c) { return reinterpret_cast<const uint32_t *>(base + c * cb); }
__device__ inline const WWalker *chunk_walkers(const char *base, int64_t cb, int c) { return reinterpret_cast<const WWalker *>(base + c * cb + 16); }
__device__ inline const WRet *chunk_rets(const char *base, int64_t cb, int32_t cap_w, int c) {
  return reinterpret_cast<const WRet *>(base + c * cb + 16 + (int64_t)cap_w * SW_BYTES);
}

__device__ inline void block_flush_counters(DevCounters *ctr, unsigned long long *red, unsigned long long steps,
                                            unsigned long long dead, unsigned long long degc, unsigned long long degp,
                                            unsigned long long reads, unsigned long long fb) {
  // red: 6 words of LDS, zeroed before the block's work; one global atomic per counter per BLOCK
  steps = wave_sum_u64(steps); dead = wave_sum_u64(dead); degc = wave_sum_u64(degc);
  degp = wave_sum_u64(degp); reads = wave_sum_u64(reads); fb = wave_sum_u64(fb);
  if (lane_id() == 0) {
    if (steps) atomicAdd(&red[0], steps);
    if (dead) atomicAdd(&red[1], dead);
    if (degc) atomicAdd(&red[2], degc);
    if (degp) atomicAdd(&red[3], degp);
    if (reads) atomicAdd(&red[4], reads);
    if (fb) atomicAdd(&red[5], fb);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (red[0]) atomicAdd(&ctr->steps, red[0]);
    if (red[1]) atomicAdd(&ctr->dead_ends, red[1]);
    if (red[2]) atomicAdd(&ctr->sum_deg_curr, red[2]);
    if (red[3]) atomicAdd(&ctr->sum_deg_prev, red[3]);
    if (red[4]) atomicAdd(&ctr->ent_reads, red[4]);
    if (red[5]) atomicAdd(&ctr->fallbacks, red[5]);
  }
}

// prefix of the incoming walkers per chunk -> LDS pre[0 .. world]; returns the total
__device__ inline uint32_t shard_in_prefix(const ShardIO &io, uint32_t *pre) {
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int c = 0; c < io.world; ++c) { pre[c] = acc; acc += min(chunk_hdr(io.recv, io.chunk_bytes, c)[0], (uint32_t)io.cap_w); }
    pre[io.world] = acc;
  }
  __syncthreads();
  return pre[io.world];
}
__device__ inline SWalker shard_in_record(const ShardIO &io, const uint32_t *pre, uint32_t i) {
  int c = 0;
  while (c + 1 < io.world && i >= pre[c + 1]) ++c;
  const WWalker w = chunk_walkers(io.recv, io.chunk_bytes, c)[i - pre[c]];
  SWalker r; r.lw = w.lw; r.src = w.src; r.prev = w.prev; r.curr = w.curr; r.v = 0; r.kind = 0; r.pad0 = 0; r.pad1 = 0;
  return r;
}
// per-block slice of n records in units of `unit` records (TPB for the per-lane kernels, TPB / 64 for one wave per record)
__device__ inline void shard_slice(uint32_t n, uint32_t unit, uint32_t &lo, uint32_t &hi) {
  uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  per = (per + unit - 1) / unit * unit;
  const uint64_t l = (uint64_t)blockIdx.x * per, h = l + per;
  lo = (uint32_t)(l < n ? l : n); hi = (uint32_t)(h < n ? h : n);
}

// What happens to a walker that has just sampled `next` (or died): the scratch record the bucketing kernel turns into a
// forwarded walker (unless this was the last step) and the return that carries `next` home.
__device__ inline SWalker shard_advance(const SWalker &wk, int32_t step, int32_t next, bool last) {
  SWalker nw = wk;
  nw.prev = wk.curr; nw.curr = next; nw.v = next;
  nw.kind = last ? SK_RET : SK_WALKER_RET;
  return nw;
}
__device__ inline SWalker shard_dead(const SWalker &wk) { SWalker d = wk; d.kind = SK_DEAD; return d; }
__device__ inline WWalker shard_wire_of(const SWalker &w) { WWalker o; o.lw = w.lw; o.src = w.src; o.prev = w.prev; o.curr = w.curr; return o; }
__device__ inline WRet shard_ret_of(const SWalker &w) {
  WRet r;
  r.lw = w.kind == SK_DEAD ? (int32_t)((uint32_t)w.lw | 0x80000000u) : w.lw;
  r.v = w.kind == SK_DEAD ? 0 : w.v;
  return r;
}

// linked walkers (k_sh_step_cfo): prev | curr << 32 = the link of the vertex the walker stands on, laid out as CfoEnt::link
__device__ inline uint64_t shard_link_of(const Row &r) {
  return ((uint64_t)r.off & CFO_NOFF_MASK) | ((uint64_t)(uint32_t)min(r.deg, (int32_t)CFO_NDEG_MAX) << 40) |
         ((uint64_t)((r.flags & ROW_IRREGULAR) != 0) << 63);
}
// A shard's own seeds must fit the chunks of its receive buffer: n_local * batch / world per chunk against a capacity sized
// from nVertices / world^2 — skewed ownership (SRW_CFG_OWNER_FROM_PARTITIONS with fewer partitions than GPUs) breaks that,
// so the surplus is dropped and the overflow flag raised like everywhere else (the batch is redone with more slack).
__global__ void k_sh_seed(const int32_t *__restrict__ verts, int64_t n_local, ShardIO io, char *recv_w,
                          int32_t *__restrict__ paths, int32_t *__restrict__ lens, int64_t stride,
                          const Row *__restrict__ link_rows, int32_t vmin, uint32_t *overflow) {
  const int64_t n = n_local * io.batch;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t src = verts[i / io.batch];
    WWalker w; w.lw = (int32_t)i; w.src = src; w.prev = src; w.curr = src;
    if (link_rows) { const uint64_t l = shard_link_of(link_rows[(int64_t)src - vmin]); w.prev = (int32_t)(uint32_t)l; w.curr = (int32_t)(uint32_t)(l >> 32); }
    const int c = (int)(i % io.world);
    if (i / io.world < (int64_t)io.cap_w) reinterpret_cast<WWalker *>(recv_w + c * io.chunk_bytes + 16)[i / io.world] = w;
    paths[i] = src;                   // slot 0 of the slot-major staging [stride][n] (k_sh_apply)
    lens[i] = (int32_t)stride;        // full length unless a death notice says otherwise (k_sh_apply)
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < io.world) {
    const int c = (int)threadIdx.x;
    uint32_t *h = reinterpret_cast<uint32_t *>(recv_w + c * io.chunk_bytes);
    const int64_t mine = (n - c + io.world - 1) / io.world;
    if (mine > (int64_t)io.cap_w) atomicOr(overflow, 1u);
    h[0] = (uint32_t)(mine < (int64_t)io.cap_w ? mine : (int64_t)io.cap_w); h[1] = 0u; h[2] = 0u; h[3] = 0u;
  }
}

// returns of the previous super-step: each one is path slot `slot` of its walker; death notices set lens.
// The home rank stages its paths SLOT-MAJOR, pt[slot][row]: the 4-byte stores of one super-step — one per walker, in the
// order the returns arrive — then fall into one contiguous n_rows * 4 B row that the caches absorb (L2 + Infinity Cache) and
// write back as full lines; into the final [row][L + 2] matrix they were one partial 64-byte sector each, and k_sh_apply
// cost as much as the sampling kernel (s15: 1.94 ms vs 1.90 ms per super-step of 35.5 M walkers).  k_sh_transpose turns
// the staging into the final layout once per batch (two streaming passes over the batch's paths).
__global__ void k_sh_apply(ShardIO io, int32_t *__restrict__ pt, int32_t *__restrict__ lens, int64_t n_rows, int32_t slot) {
  for (int c = 0; c < io.world; ++c) {
    const uint32_t n = min(chunk_hdr(io.recv, io.chunk_bytes, c)[1], (uint32_t)io.cap_r);
    const WRet *r = chunk_rets(io.recv, io.chunk_bytes, io.cap_w, c);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
      const WRet x = r[i];
      if (x.lw < 0) lens[x.lw & 0x7FFFFFFF] = slot;             // death notice: the walker stopped with `slot` entries
      else pt[(int64_t)slot * n_rows + x.lw] = x.v;
    }
  }
}

// pt[slot][row] -> paths[row][slot], -1 beyond the row's length: tiles of 64 rows x 16 slots through LDS, 256-byte reads,
// 64-byte runs per row on the way out.
__global__ __launch_bounds__(TPB) void k_sh_transpose(const int32_t *__restrict__ pt, const int32_t *__restrict__ lens, int64_t n_rows,
                                                      int64_t stride, int32_t *__restrict__ paths) {
  __shared__ int32_t tile[16][64 + 1];
  const int t = threadIdx.x;
  for (int64_t w0 = (int64_t)blockIdx.x * 64; w0 < n_rows; w0 += (int64_t)gridDim.x * 64) {
    const int64_t wr = w0 + (t >> 2);
    const int32_t len = wr < n_rows ? lens[wr] : 0;
    for (int64_t s0 = 0; s0 < stride; s0 += 16) {
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int64_t sl = s0 + pass * 4 + (t >> 6), w = w0 + (t & 63);
        tile[pass * 4 + (t >> 6)][t & 63] = (sl < stride && w < n_rows) ? pt[sl * n_rows + w] : -1;
      }
      __syncthreads();
      if (wr < n_rows) {
        const int c0 = (t & 3) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int64_t sl = s0 + c0 + c;
          if (sl < stride) paths[wr * stride + sl] = sl < len ? tile[c0 + c][t >> 2] : -1;
        }
      }
      __syncthreads();
    }
  }
}

// todo != null: only the records listed there (those k_sh_step_tab found no table for); k_sh_scatter then buckets the
// whole scratch array, so no per-block counts are produced.
__global__ __launch_bounds__(TPB, 4) void k_sh_step(GraphView g, ShardIO io, int32_t first_walk, int32_t step, int32_t last,
                                                    RngSpec rng, float p, float q, SWalker *__restrict__ scratch,
                                                    uint32_t *__restrict__ blk, DevCounters *ctr,
                                                    const uint32_t *__restrict__ todo, const unsigned long long *todo_n) {
  __shared__ __attribute__((aligned(16))) uint32_t bitmap[TPB / 64][BINNED_LDS_WORDS];
  __shared__ uint32_t cnt[2 * SHARD_MAX_WORLD], pre[SHARD_MAX_WORLD + 1];
  __shared__ unsigned long long red[6];
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  if (threadIdx.x < 2 * SHARD_MAX_WORLD) cnt[threadIdx.x] = 0u;
  if (threadIdx.x < 6) red[threadIdx.x] = 0ull;
  const uint32_t n_in = shard_in_prefix(io, pre);          // contains the __syncthreads() cnt / red need
  Member mem; mem.mode = 0; mem.bm = bitmap[wv]; mem.seg_base = 0;
  unsigned long long steps = 0, dead = 0, degc = 0, degp = 0, fb = 0;
  uint32_t n_strat[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // SRW_STRAT_* 0 .. 7 (lane 0 counts)
  uint32_t lo, hi;
  shard_slice(n_in, TPB / 64, lo, hi);
  uint32_t t_step = TPB / 64;
  if (todo) { lo = blockIdx.x * (TPB / 64); hi = (uint32_t)*todo_n; t_step = gridDim.x * (TPB / 64); }
  for (uint32_t ti = lo + wv; ti < hi; ti += t_step) {     // one wave per record
    const uint32_t ri = todo ? todo[ti] : ti;
    const SWalker wk = shard_in_record(io, pre, ri);
    const Row *rp = row_of(g, wk.curr);
    Row r; r.off = 0; r.deg = 0; r.flags = 0;
    if (rp) r = *rp;
    if (r.deg == 0) {                                  // dead end (or a source without neighbors): tell the home rank the length
      if (lane == 0) {
        scratch[ri] = shard_dead(wk);
        const int32_t hm = owner_of_tab(wk.src, io.world, g.owner_tab, g.vmin, g.n_slots);
        if (hm != io.rank) atomicAdd(&cnt[SHARD_MAX_WORLD + hm], 1u);
      }
      if (step > 1) dead += (lane == 0);
      continue;
    }
    const uint32_t iter = (uint32_t)(first_walk + wk.lw % io.batch);
    Bias b = make_bias(g, p, q, wk.prev, step > 1);
    float u = draw_uniform(rng, iter, (uint32_t)rng_source(g, wk.src), (uint32_t)step);
    unsigned f = 0, sv = 0;
    int32_t k = -1, nid = 0;                         // same routing as k_walk_general (no per-edge tables on a shard)
    unsigned which = SRW_STRAT_SCAN;
    if (!b.need_member) { k = wave_pick_prefix(g, r, (int64_t)wk.curr - g.vmin, b, mem.bm, u, f, sv); if (k >= 0) which = SRW_STRAT_PREFIX; }
    else {
      unsigned long long ab = 0; unsigned su = 0;
      k = wave_pick_binned(g, r, (int64_t)wk.curr - g.vmin, b, mem.bm, u, f, sv, 0, false, mem, ab, su, nid);
      if (k >= 0) which = su == 1 ? SRW_STRAT_P1 : su == 2 ? SRW_STRAT_P2 : su == 4 ? SRW_STRAT_P3 : SRW_STRAT_W;
    }
    if (k < 0) k = wave_pick_scan(g, r, b, mem, u, f);
    if (lane == 0) { n_strat[which] += 1; n_strat[SRW_STRAT_CHAIN] += f; }
    const int32_t next = g.ent[r.off + k].id;
    if (lane == 0) {
      const SWalker nw = shard_advance(wk, step, next, last != 0);
      scratch[ri] = nw;
      if (nw.kind != SK_RET) atomicAdd(&cnt[owner_of_tab(next, io.world, g.owner_tab, g.vmin, g.n_slots)], 1u);
      { const int32_t hm = owner_of_tab(wk.src, io.world, g.owner_tab, g.vmin, g.n_slots);      // every sampled vertex goes home
        if (hm != io.rank) atomicAdd(&cnt[SHARD_MAX_WORLD + hm], 1u); }
      steps += 1; degc += (unsigned long long)r.deg; fb += f;
      if (b.need_member) degp += (unsigned long long)b.prev_deg;
    }
  }
  if (lane == 0)
    for (int i = 0; i < 8; ++i) if (n_strat[i]) atomicAdd(&ctr->strat[i], (unsigned long long)n_strat[i]);
  block_flush_counters(ctr, red, steps, dead, degc, degp, 0, fb);   // contains the __syncthreads() cnt needs
  if (!todo && (int)threadIdx.x < io.world) {
    blk[(int64_t)blockIdx.x * 2 * io.world + threadIdx.x] = cnt[threadIdx.x];
    blk[(int64_t)blockIdx.x * 2 * io.world + io.world + threadIdx.x] = cnt[SHARD_MAX_WORLD + threadIdx.x];
  }
}

// p = q = 1 on a shard: one record per lane through the precomputed CDF + guide table.
template <bool NT>
__global__ __launch_bounds__(TPB) void k_sh_step_fo(GraphView g, ShardIO io, int32_t first_walk, int32_t step, int32_t last,
                                                    RngSpec rng, SWalker *__restrict__ scratch, uint32_t *__restrict__ blk,
                                                    DevCounters *ctr) {
  __shared__ uint32_t cnt[2 * SHARD_MAX_WORLD], pre[SHARD_MAX_WORLD + 1];
  __shared__ unsigned long long red[6];
  if (threadIdx.x < 2 * SHARD_MAX_WORLD) cnt[threadIdx.x] = 0u;
  if (threadIdx.x < 6) red[threadIdx.x] = 0ull;
  const uint32_t n_in = shard_in_prefix(io, pre);
  unsigned long long steps = 0, dead = 0, reads = 0, fb = 0;
  uint32_t lo, hi;
  shard_slice(n_in, TPB, lo, hi);
  for (uint32_t base = lo; base < hi; base += TPB) {
    const uint32_t ri = base + threadIdx.x;
    int32_t o = -1, hm = -1;
    if (ri < hi) {
      const SWalker wk = shard_in_record(io, pre, ri);
      const Row *rp = row_of(g, wk.curr);
      Row r; r.off = 0; r.deg = 0; r.flags = 0;
      if (rp) r = *rp;
      if (r.deg == 0) {
        if (step > 1) ++dead;
        scratch[ri] = shard_dead(wk);
        hm = owner_of_tab(wk.src, io.world, g.owner_tab, g.vmin, g.n_slots);     // death notice to the home rank
        if (hm == io.rank) hm = -1;                                             // (applied in place by k_sh_bucket)
      } else {
        const uint32_t iter = (uint32_t)(first_walk + wk.lw % io.batch);
        float u = draw_uniform(rng, iter, (uint32_t)rng_source(g, wk.src), (uint32_t)step);
        int32_t next;
        if (r.flags & ROW_IRREGULAR) {
          Bias nb; nb.second_order = false; nb.need_member = false; nb.p = nb.q = 1.0f; nb.prev = 0;
          nb.prev_sids = nullptr; nb.prev_deg = 0; nb.vmin = g.vmin;
          next = g.ent[r.off + lane_pick_sequential(g.ent + r.off, r.deg, nb, u)].id; ++fb;
        } else {
          unsigned rd; int32_t k;
          FoEnt e = fo_pick<NT>(g.fo + r.off, r.deg, u, k, rd); reads += rd;
          next = e.id;
        }
        const SWalker nw = shard_advance(wk, step, next, last != 0);
        scratch[ri] = nw;
        ++steps;
        if (nw.kind != SK_RET) o = owner_of_tab(next, io.world, g.owner_tab, g.vmin, g.n_slots);
        hm = owner_of_tab(wk.src, io.world, g.owner_tab, g.vmin, g.n_slots);
        if (hm == io.rank) hm = -1;
      }
    }
    for (int32_t d = 0; d < io.world; ++d) {               // one LDS atomic per wave, destination and kind
      const unsigned long long m = __ballot(o == d), mh = __ballot(hm == d);
      if (lane_id() == 0) {
        if (m) atomicAdd(&cnt[d], (uint32_t)__popcll(m));
        if (mh) atomicAdd(&cnt[SHARD_MAX_WORLD + d], (uint32_t)__popcll(mh));
      }
    }
  }
  block_flush_counters(ctr, red, steps, dead, 0, 0, reads, fb);
  if ((int)threadIdx.x < io.world) {
    blk[(int64_t)blockIdx.x * 2 * io.world + threadIdx.x] = cnt[threadIdx.x];
    blk[(int64_t)blockIdx.x * 2 * io.world + io.world + threadIdx.x] = cnt[SHARD_MAX_WORLD + threadIdx.x];
  }
}

// p = q = 1 on a shard whose compact records carry links into the owners' tables (srw_shard_rows_*): sampling and
// bucketing in ONE pass.  A walker arrives with the link of the vertex it stands on (no row-table read), picks a 16-byte
// record like k_walk_first_order and leaves with that record's link.  Per tile of TPB * SH_R records: every lane samples
// its SH_R records into registers; the waves count their survivors per destination and their returns per home rank in
// LDS (one LDS atomic per wave and distinct destination); 2 * world threads move the block's counts onto the device-wide
// chunk cursors (one global atomic per tile, destination and kind); the lanes store their records straight into the
// destination chunks.  The last block to finish writes the chunk headers and clears the cursors for the next super-step.
#ifndef SRW_SH_R
#define SRW_SH_R 4
#endif
constexpr int SH_R = SRW_SH_R;      // records per lane and tile
constexpr int SH_CUR_DONE = 2 * SHARD_MAX_WORLD;      // cursors[0 .. 2 * MAX): walkers / returns per destination; [DONE]: finished blocks
template <bool NT>
__global__ __launch_bounds__(TPB) void k_sh_step_cfo(GraphView g, ShardIO io, int32_t first_walk, int32_t step, int32_t last,
                                                     RngSpec rng, uint32_t *__restrict__ cursors, ShardDst dst,
                                                     uint32_t *__restrict__ overflow, DevCounters *ctr) {
  __shared__ uint32_t cnt[2 * SHARD_MAX_WORLD], gbase[2 * SHARD_MAX_WORLD], pre[SHARD_MAX_WORLD + 1];
  __shared__ unsigned long long red[6];
  __shared__ uint32_t is_last;
  const int lane = lane_id();
  if (threadIdx.x < 2 * SHARD_MAX_WORLD) cnt[threadIdx.x] = 0u;
  if (threadIdx.x < 6) red[threadIdx.x] = 0ull;
  const uint32_t n_in = shard_in_prefix(io, pre);          // contains the __syncthreads() cnt / red need
  unsigned long long steps = 0, dead = 0, reads = 0, fb = 0;
  Bias nobias; nobias.second_order = false; nobias.need_member = false; nobias.p = nobias.q = 1.0f;
  nobias.prev = 0; nobias.prev_sids = nullptr; nobias.prev_deg = 0; nobias.vmin = g.vmin;
  uint32_t lo, hi;
  shard_slice(n_in, TPB * SH_R, lo, hi);
  for (uint32_t base = lo; base < hi; base += TPB * SH_R) {
    SWalker nw[SH_R];
    int32_t o[SH_R], hm[SH_R], kind[SH_R];
    uint32_t wpos[SH_R], rpos[SH_R];
#pragma unroll
    for (int r = 0; r < SH_R; ++r) {
      const uint32_t ri = base + (uint32_t)r * TPB + threadIdx.x;
      o[r] = -1; hm[r] = -1; kind[r] = SK_WALKER_RET; wpos[r] = 0; rpos[r] = 0;
      if (ri < hi) {
        const SWalker wk = shard_in_record(io, pre, ri);
        const uint64_t link = (uint64_t)(uint32_t)wk.prev | ((uint64_t)(uint32_t)wk.curr << 32);
        const int64_t off = (int64_t)(link & CFO_NOFF_MASK);
        const int32_t deg = (int32_t)((link >> 40) & 0x7FFFFFu);
        nw[r] = wk;
        if (deg == 0) {                                   // dead end (or a source without neighbors): death notice to the home rank
          if (step > 1) ++dead;
          kind[r] = SK_DEAD;
          hm[r] = owner_of_tab(wk.src, io.world, g.owner_tab, g.vmin, g.n_slots);
          if (hm[r] == io.rank) { hm[r] = -1; io.lens[wk.lw] = step; }
        } else {
          const uint32_t iter = (uint32_t)(first_walk + wk.lw % io.batch);
          CfoEnt e;
          if (!(link >> 63)) {
            const uint32_t m = walk_bits24(rng.seed, iter, (uint32_t)rng_source(g, wk.src), (uint32_t)step);
            unsigned rd;
            e = cfo_pick<NT>(g.cfo + off, deg, m, rd); reads += rd;
          } else {                                        // irregular row: the reference's scan, literally; the links are valid for every row
            const float u = draw_uniform(rng, iter, (uint32_t)rng_source(g, wk.src), (uint32_t)step);
            e = g.cfo[off + lane_pick_sequential(g.ent + off, deg, nobias, u)]; ++fb;
          }
          ++steps;
          const uint64_t nl = e.link & ~(0xFull << 36);
          nw[r].v = e.id; nw[r].prev = (int32_t)(uint32_t)nl; nw[r].curr = (int32_t)(uint32_t)(nl >> 32);   // forwarded: the link of the vertex it moves to
          if (last) kind[r] = SK_RET;
          else o[r] = owner_of_tab(e.id, io.world, g.owner_tab, g.vmin, g.n_slots);
          hm[r] = owner_of_tab(wk.src, io.world, g.owner_tab, g.vmin, g.n_slots);
          if (hm[r] == io.rank) { hm[r] = -1; io.pt[(int64_t)step * io.n_rows + wk.lw] = e.id; }
        }
      }
    }
    // positions inside the block's share of every chunk: wave-aggregated LDS atomics, one per distinct destination
#pragma unroll
    for (int r = 0; r < SH_R; ++r) {
      unsigned long long todo = __ballot(o[r] >= 0);
      while (todo) {
        const int d = __builtin_amdgcn_readlane(o[r], __ffsll((long long)todo) - 1);
        const unsigned long long m = __ballot(o[r] == d);
        uint32_t b0 = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) b0 = atomicAdd(&cnt[d], (uint32_t)__popcll(m));
        b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, leader);
        if (o[r] == d) wpos[r] = b0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
      }
      todo = __ballot(hm[r] >= 0);
      while (todo) {
        const int d = __builtin_amdgcn_readlane(hm[r], __ffsll((long long)todo) - 1);
        const unsigned long long m = __ballot(hm[r] == d);
        uint32_t b0 = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) b0 = atomicAdd(&cnt[SHARD_MAX_WORLD + d], (uint32_t)__popcll(m));
        b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, leader);
        if (hm[r] == d) rpos[r] = b0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * io.world) {
      const int d = (int)threadIdx.x < io.world ? (int)threadIdx.x : (int)threadIdx.x - io.world;
      const int idx = (int)threadIdx.x < io.world ? d : SHARD_MAX_WORLD + d;
      const uint32_t c = cnt[idx];
      cnt[idx] = 0u;
      uint32_t gb = 0;
      if (c) {
        gb = atomicAdd(&cursors[idx], c);
        if ((uint64_t)gb + c > (uint64_t)((int)threadIdx.x < io.world ? io.cap_w : io.cap_r)) atomicOr(overflow, 1u);
      }
      gbase[idx] = gb;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SH_R; ++r) {
      if (o[r] >= 0) {
        const uint32_t pos = gbase[o[r]] + wpos[r];
        if (pos < (uint32_t)io.cap_w) reinterpret_cast<WWalker *>(dst.p[o[r]] + 16)[pos] = shard_wire_of(nw[r]);
      }
      if (hm[r] >= 0) {
        const uint32_t pos = gbase[SHARD_MAX_WORLD + hm[r]] + rpos[r];
        if (pos < (uint32_t)io.cap_r) {
          SWalker t = nw[r]; t.kind = kind[r];
          reinterpret_cast<WRet *>(dst.p[hm[r]] + 16 + (int64_t)io.cap_w * SW_BYTES)[pos] = shard_ret_of(t);
        }
      }
    }
  }
  block_flush_counters(ctr, red, steps, dead, 0, 0, reads, fb);
  // the last block: chunk headers from the cursors, cursors cleared for the next super-step
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(&cursors[SH_CUR_DONE], 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (is_last) {
    __threadfence();
    if ((int)threadIdx.x < 2 * io.world) {
      const bool rets = (int)threadIdx.x >= io.world;
      const int d = rets ? (int)threadIdx.x - io.world : (int)threadIdx.x;
      const uint32_t total = atomicExch(&cursors[rets ? SHARD_MAX_WORLD + d : d], 0u);
      const uint32_t cap = (uint32_t)(rets ? io.cap_r : io.cap_w);
      reinterpret_cast<uint32_t *>(dst.p[d])[rets ? 1 : 0] = total < cap ? total : cap;
      atomicMax(&ctr->why[rets ? 1 : 0], (unsigned long long)total);      // the fullest chunk of the batch (run_shard_finish, SRW_TIMING)
    }
    if (threadIdx.x == 0) cursors[SH_CUR_DONE] = 0u;
  }
}

// blk[b][col] (counts) -> blk[b][col] (write cursor of block b inside chunk col's record array); the chunk headers get
// the totals (clamped to the capacity, overflow flagged).  One block; wave w handles columns w, w + nwaves, ...
__global__ void k_sh_offsets(uint32_t *__restrict__ blk, int32_t n_blocks, ShardIO io, ShardDst dst, uint32_t *overflow) {
  const int lane = lane_id(), wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int per = (n_blocks + 63) / 64, cols = 2 * io.world;
  for (int col = wv; col < cols; col += nw) {
    unsigned long long loc = 0;
    for (int i = 0; i < per; ++i) { const int b = lane * per + i; if (b < n_blocks) loc += blk[(int64_t)b * cols + col]; }
    unsigned long long incl = loc;
    for (int o = 1; o < 64; o <<= 1) { unsigned long long t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    unsigned long long run = incl - loc;
    for (int i = 0; i < per; ++i) {
      const int b = lane * per + i;
      if (b < n_blocks) { const uint32_t c = blk[(int64_t)b * cols + col]; blk[(int64_t)b * cols + col] = (uint32_t)run; run += c; }
    }
    const unsigned long long total = (unsigned long long)__shfl((long long)incl, 63);
    if (lane == 0) {
      const bool rets = col >= io.world;
      const int d = rets ? col - io.world : col;
      const uint32_t cap = (uint32_t)(rets ? io.cap_r : io.cap_w);
      uint32_t *h = reinterpret_cast<uint32_t *>(dst.p[d]);
      h[rets ? 1 : 0] = (uint32_t)(total < cap ? total : cap);
      if (total > cap) atomicOr(overflow, 1u);
    }
  }
}

__global__ __launch_bounds__(TPB) void k_sh_bucket(GraphView g, ShardIO io, int32_t unit, int32_t step,
                                                   const SWalker *__restrict__ recs, const uint32_t *__restrict__ blk, ShardDst dst) {
  __shared__ uint32_t cur[2 * SHARD_MAX_WORLD], pre[SHARD_MAX_WORLD + 1];
  const uint32_t n_in = shard_in_prefix(io, pre);
  if ((int)threadIdx.x < io.world) {
    cur[threadIdx.x] = blk[(int64_t)blockIdx.x * 2 * io.world + threadIdx.x];
    cur[SHARD_MAX_WORLD + threadIdx.x] = blk[(int64_t)blockIdx.x * 2 * io.world + io.world + threadIdx.x];
  }
  __syncthreads();
  const int lane = lane_id();
  uint32_t lo, hi;
  shard_slice(n_in, (uint32_t)unit, lo, hi);
  for (uint32_t base = lo; base < hi; base += TPB) {
    const uint32_t i = base + threadIdx.x;
    SWalker w; w.lw = 0; w.src = 0; w.prev = 0; w.curr = 0; w.v = 0; w.kind = SK_RET; w.pad0 = w.pad1 = 0;
    int32_t o = -1, hm = -1;
    if (i < hi) {
      w = recs[i];
      if (w.kind == SK_WALKER_RET) o = owner_of_tab(w.curr, io.world, g.owner_tab, g.vmin, g.n_slots);
      hm = owner_of_tab(w.src, io.world, g.owner_tab, g.vmin, g.n_slots);
      if (hm == io.rank) { hm = -1; shard_return_home(io, w, step); }
    }
    for (int32_t d = 0; d < io.world; ++d) {
      const unsigned long long m = __ballot(o == d);
      if (m) {
        uint32_t b0 = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) b0 = atomicAdd(&cur[d], (uint32_t)__popcll(m));
        b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, leader);
        const uint32_t pos = b0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (o == d && pos < (uint32_t)io.cap_w) reinterpret_cast<WWalker *>(dst.p[d] + 16)[pos] = shard_wire_of(w);
      }
      const unsigned long long mh = __ballot(hm == d);
      if (mh) {
        uint32_t b0 = 0;
        const int leader = __ffsll((long long)mh) - 1;
        if (lane == leader) b0 = atomicAdd(&cur[SHARD_MAX_WORLD + d], (uint32_t)__popcll(mh));
        b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, leader);
        const uint32_t pos = b0 + (uint32_t)__popcll(mh & ((1ull << lane) - 1ull));
        if (hm == d && pos < (uint32_t)io.cap_r)
          reinterpret_cast<WRet *>(dst.p[d] + 16 + (int64_t)io.cap_w * SW_BYTES)[pos] = shard_ret_of(w);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// q != 1 on a shard that holds the per-edge tables of the pairs into its own rows (edge_tables.hip:prepare_shard_tables):
// the lean table step of k_walk_tables, one super-step at a time.  One wave per incoming walker, persistent waves taking
// groups of records from a cursor (a step costs anything from one row of registers to a located chunk of a hub row).  The
// step's first round trip issues together: the row of curr (local), the membership row of prev (replicated) and the pair
// hash probe that yields the table word eb_off[e] would hold on a whole-graph handle.  Steps without a table (uncertified
// rows, test configurations) go to the todo list: k_sh_step redoes exactly those with the on-the-fly samplers.  The sampled
// records land in `scratch` in input order; k_sh_scatter buckets them.
// records per cursor grab (a single counter word saturates at ~88 atomics/us).  With the BATCH prologue of k_sh_step_tab a grab is also
// what one Philox evaluation / one round of record, row and pair-hash reads serves: 8 -> 1 221 ms, 16 -> 676, 32 -> 608, 64 -> 606 ms per
// iteration at config 3's shape, 3 809 / 3 756 / 3 741 ms at config 5's (profiles/r04_sharded_batch.md); the kernel takes fewer per grab
// when a super-step has fewer than 4 grabs per wave (small shards: the waves would not share the work evenly).
constexpr int SH_GRAB = 32;
struct alignas(16) ChainMeta { long long d_off; int32_t deg; uint32_t u_off; };   // first quotient in the scratch array, row length, first work unit
struct ChainUnits { double *usum; int32_t *ue; unsigned long long *utot; };       // per unit of 256 quotients: plain sum, guessed binade, integer increment
// (arguments as one struct, read again from the kernarg segment where a record needs them: k_walk_tables, device_common.h:fresh_args)
struct ShTabArgs {
  GraphView g; ShardIO io; int32_t first_walk, step, last; RngSpec rng; float p, q; SWalker *scratch;
  unsigned long long *cursor; uint32_t *todo; DevCounters *ctr; int32_t grab_n; ChainRec *chain;
};
#if SRW_TAB_FRESH
#define SH_TAB_ARGS() fresh_args<ShTabArgs>()
#else
#define SH_TAB_ARGS() a0
#endif
// BATCH (round 4, SRW_SH_BATCH=0 for the A / B): what a record needs before its first table read — the record itself, its
// Philox draw and the pair-hash probe — is fetched and computed for the WHOLE grab at once, lane l for record r0 + l (one Philox
// evaluation and one probe chain per 16 records instead of 16 wave-wide ones; the record's step then starts at the row reads),
// and every pick goes back through SGPRs (uni) so that the record loop is uniform in the compiler's eyes, as in k_walk_tables.
// BATCH == 2: also the row of curr and the membership row of prev per lane, and the sampled records collected per lane
// (v_writelane) and stored once per grab, coalesced.
#ifndef SRW_SH_LEAN_WAVES                 // (waves per SIMD of the sharded table step: the grab's per-lane state costs 36 B of scratch at 8)
#define SRW_SH_LEAN_WAVES SRW_LEAN_WAVES
#endif
#ifndef SRW_SH_LEAN_WAVES_BF
#define SRW_SH_LEAN_WAVES_BF SRW_LEAN_WAVES_BF
#endif
template <bool BF, int BATCH>
__global__ __launch_bounds__(TPB, BF ? SRW_SH_LEAN_WAVES_BF : SRW_SH_LEAN_WAVES) void k_sh_step_tab(ShTabArgs a0) {
  __shared__ __attribute__((aligned(16))) uint32_t stage_all[TPB / 64][1024];
  __shared__ uint32_t pre[SHARD_MAX_WORLD + 1];
  const int lane = lane_id();
  uint32_t *stage = stage_all[threadIdx.x >> 6];
  const uint32_t n_in = shard_in_prefix(a0.io, pre);
  Member mem; mem.mode = 0; mem.bm = stage; mem.seg_base = 0;
  const int32_t step = a0.step;
  const uint32_t n_waves4 = gridDim.x * (uint32_t)(TPB / 64) * 4u;
  const int32_t grab_n = (int32_t)uni(n_in / n_waves4 >= (uint32_t)a0.grab_n ? (uint32_t)a0.grab_n : (n_in / n_waves4 ? n_in / n_waves4 : 1u));
  const bool second = step > 1;
  unsigned long long srch = 0;
  uint32_t steps = 0, fb = 0, dead = 0, fast = 0, n_tab = 0, n_mask = 0, n_first = 0, n_todo = 0;
  while (true) {
    unsigned long long grab = 0;
    if (lane == 0) grab = atomicAdd(SH_TAB_ARGS().cursor, (unsigned long long)grab_n);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(grab > 0xFFFFFFFFull ? 0xFFFFFFFFull : grab));
    if (r0 >= n_in) break;
    const uint32_t r1 = r0 + (uint32_t)grab_n < n_in ? r0 + (uint32_t)grab_n : n_in;
    // (BATCH) lane l: record r0 + l, its draw, the table word of its pair (b_eo: bit 0 of b_found = the pair has one)
    SWalker bw; bw.lw = 0; bw.src = 0; bw.prev = 0; bw.curr = 0;
    float bu = 0.0f; uint32_t b_eo = EB_NONE, b_found = 0u;
    Row b_r, b_mr; b_r.off = 0; b_r.deg = 0; b_r.flags = 0; b_mr = b_r;
    int32_t b_next = 0, b_stat = 0;                   // (BATCH == 2) per lane: the sampled vertex, 1 = advance / 2 = dead / 0 = not this kernel's
    if constexpr (BATCH != 0) {
      const ShTabArgs ab = SH_TAB_ARGS();
      const GraphView &g = ab.g;
      const uint32_t rl = r0 + (uint32_t)lane < r1 ? r0 + (uint32_t)lane : r1 - 1u;      // (grab_n <= 64: run_shard_superstep)
      bw = shard_in_record(ab.io, pre, rl);
      const uint32_t iter = (uint32_t)(ab.first_walk + bw.lw % ab.io.batch);
      bu = draw_uniform(ab.rng, iter, (uint32_t)rng_source(g, bw.src), (uint32_t)step);
      const int64_t cslot = (int64_t)bw.curr - g.vmin, pslot = (int64_t)bw.prev - g.vmin;
      if (second && cslot >= 0 && cslot < g.n_slots && pslot >= 0 && pslot < g.n_slots) {
        uint32_t pad;
        if constexpr (BATCH == 2) b_mr = g.mrows[pslot];
        b_found = pair_lookup_lane(g.ph, g.ph_buckets, (uint32_t)pslot, (uint32_t)cslot, b_eo, pad) ? 1u : 0u;
      }
      if constexpr (BATCH == 2) { if (cslot >= 0 && cslot < g.n_slots) b_r = g.rows[cslot]; }
    }
    for (uint32_t ri = r0; ri < r1; ++ri) {
      const ShTabArgs ar = SH_TAB_ARGS();             // (what the top of a record needs; the samplers read the graph again where they start)
      const GraphView &g = ar.g;
      SWalker wk;
      if constexpr (BATCH != 0) {
        const int j = (int)(ri - r0);
        wk.lw = __builtin_amdgcn_readlane(bw.lw, j); wk.src = __builtin_amdgcn_readlane(bw.src, j);
        wk.prev = __builtin_amdgcn_readlane(bw.prev, j); wk.curr = __builtin_amdgcn_readlane(bw.curr, j);
        wk.v = 0; wk.kind = 0; wk.pad0 = 0; wk.pad1 = 0;
      } else {
        wk = shard_in_record(ar.io, pre, ri);
        wk.lw = __builtin_amdgcn_readfirstlane(wk.lw); wk.src = __builtin_amdgcn_readfirstlane(wk.src);
        wk.prev = __builtin_amdgcn_readfirstlane(wk.prev); wk.curr = __builtin_amdgcn_readfirstlane(wk.curr);
      }
      const int64_t cslot = (int64_t)wk.curr - g.vmin, pslot = (int64_t)wk.prev - g.vmin;
      const bool in_range = cslot >= 0 && cslot < g.n_slots;
      Row r, mr;
      uint32_t eo = EB_NONE; bool found = false;
      if constexpr (BATCH == 2) {                        // (zero rows where a slot is out of range: the prologue left them so)
        const int j = (int)(ri - r0);
        r = lane_row(b_r, j); mr = lane_row(b_mr, j);
        eo = (uint32_t)__builtin_amdgcn_readlane((int)b_eo, j); found = __builtin_amdgcn_readlane((int)b_found, j) != 0;
      } else {
        r = g.rows[in_range ? cslot : 0];
        mr.off = 0; mr.deg = 0; mr.flags = 0;
        if (second && in_range && pslot >= 0 && pslot < g.n_slots) {
          mr = g.mrows[pslot];
          if constexpr (BATCH != 0) {
            const int j = (int)(ri - r0);
            eo = (uint32_t)__builtin_amdgcn_readlane((int)b_eo, j); found = __builtin_amdgcn_readlane((int)b_found, j) != 0;
          } else found = pair_lookup_wave(g.ph, g.ph_buckets, (uint32_t)pslot, (uint32_t)cslot, eo);
          mr = uniform_row(mr);
        }
        r = uniform_row(r);
        if (!in_range) { r.off = 0; r.deg = 0; r.flags = 0; }
      }
      if (r.deg == 0) {                                  // dead end (or a source without neighbors): tell the home rank the length
        if constexpr (BATCH == 2) write_lane(b_stat, 2, (int)(ri - r0));
        else if (lane == 0) ar.scratch[ri] = shard_dead(wk);
        dead += second ? 1u : 0u;
        continue;
      }
      float u;
      if constexpr (BATCH != 0) u = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(bu), (int)(ri - r0)));
      else {
        const uint32_t iter = (uint32_t)(ar.first_walk + wk.lw % ar.io.batch);
        u = draw_uniform(ar.rng, iter, (uint32_t)__builtin_amdgcn_readfirstlane(rng_source(g, wk.src)), (uint32_t)step);
      }
      unsigned f = 0, sv = 0;
      int32_t k, next = 0;
      // CHAIN = false: a draw within rounding distance of a CDF boundary is not decided here.  On a table step the record
      // goes to the chain list with the reference's sum S (k_chain_*: the quotients of the whole row computed by the whole
      // GPU, then one sequential pass over them) — one wave running the chain over a hub row alone was the tail of every
      // other super-step; everywhere else (first steps, rows below 256 candidates) the general step takes the record.
      bool to_chain = false; double S_tie = 0.0;
      if (!second) {
        k = wave_pick_first<false>(GFRESH(), r, u, f, next);
        if constexpr (BATCH != 0) k = uni(k);
        n_first += k >= 0 ? 1u : 0u;
      } else {
        Bias b;
        b.p = ar.p; b.q = ar.q; b.prev = wk.prev; b.second_order = true; b.need_member = true; b.vmin = g.vmin;
        b.prev_sids = g.msids + mr.off; b.prev_deg = mr.deg; b.prev_hub = mr.flags >> ROW_HUB_SHIFT;
        if (r.deg <= g.eb_mask_max && found) {
          k = wave_pick_masked<false>(GFRESH(), r, b, eo, r.deg > 32 ? g.em_bits + (size_t)eo * 4 : nullptr, u, f, next);
          if constexpr (BATCH != 0) k = uni(k);
          if (k >= 0) { n_mask += 1; srch += 8ull * (unsigned long long)r.deg + 4ull * (unsigned long long)((r.deg + 31) >> 5); }
        } else if (r.deg > g.eb_mask_max && found && (r.flags & ROW_PQ_OK)) {
          k = wave_pick_edge_table<BF, false>(GFRESH(), r, b, g.eb_bins + (size_t)eo * 8, u, f, sv, mem, next, stage, &S_tie);
          if constexpr (BATCH != 0) k = uni(k);
          if (k >= 0) { n_tab += 1; srch += 8ull * EB_BINS; fast += sv; }
          to_chain = k == CHAIN_NEEDED;
        } else k = -1;
      }
      if (k < 0) {
        const ShTabArgs at = SH_TAB_ARGS();
        unsigned long long *cursor = at.cursor; ChainRec *chain = at.chain; uint32_t *todo = at.todo;
        if (lane == 0) {
          unsigned long long ci = to_chain ? atomicAdd(cursor + 2, 1ull) : (unsigned long long)CHAIN_CAP;
          if (ci < (unsigned long long)CHAIN_CAP) { ChainRec cr; cr.ri = ri; cr.pad = 0u; cr.S = S_tie; chain[ci] = cr; }
          else todo[atomicAdd(cursor + 1, 1ull)] = ri;    // no table for this pair / a full chain list: the general step takes the record
        }
        n_todo += 1;
        continue;
      }
      next = __builtin_amdgcn_readfirstlane(next);
      fb += f; steps += 1;
      if constexpr (BATCH == 2) {
        write_lane(b_next, next, (int)(ri - r0)); write_lane(b_stat, 1, (int)(ri - r0));
      } else if (lane == 0) { const ShTabArgs ao = SH_TAB_ARGS(); ao.scratch[ri] = shard_advance(wk, step, next, ao.last != 0); }
    }
    if constexpr (BATCH == 2) {                          // the grab's sampled records, one per lane
      const ShTabArgs ao = SH_TAB_ARGS();
      if (r0 + (uint32_t)lane < r1 && b_stat != 0)
        ao.scratch[r0 + (uint32_t)lane] = b_stat == 1 ? shard_advance(bw, step, b_next, ao.last != 0) : shard_dead(bw);
    }
  }
  if (lane == 0) {
    DevCounters *ctr = SH_TAB_ARGS().ctr;
    srch += mem.res_bytes;
    if (steps) atomicAdd(&ctr->steps, (unsigned long long)steps);
    if (dead) atomicAdd(&ctr->dead_ends, (unsigned long long)dead);
    if (fb) { atomicAdd(&ctr->fallbacks, (unsigned long long)fb); atomicAdd(&ctr->strat[SRW_STRAT_CHAIN], (unsigned long long)fb); }
    if (fast) atomicAdd(&ctr->ent_reads, (unsigned long long)fast);
    if (srch) atomicAdd(&ctr->trials, srch);
    if (n_tab) atomicAdd(&ctr->strat[SRW_STRAT_EDGE_TABLE], (unsigned long long)n_tab);
    if (n_mask) atomicAdd(&ctr->strat[SRW_STRAT_EDGE_MASK], (unsigned long long)n_mask);
    if (n_first) atomicAdd(&ctr->strat[SRW_STRAT_SCAN], (unsigned long long)n_first);
    if (n_todo) atomicAdd(&ctr->strat[SRW_STAT_HANDED_OVER], (unsigned long long)n_todo);
  }
}

// p != 1, q == 1 on a shard: one record per LANE (k_walk_q1's step, one super-step at a time; round 2 ran this case one wave
// per record through k_sh_step).  Per record: the row of curr (local), the Philox draw, and — second-order steps — the pair's
// return-edge record from the shard's hash (edge_tables.hip:build_shard_rev_hash) + q1_pick over the local compact records and
// exact prefix sums.  An irregular row or a draw within rounding distance of a CDF boundary puts the record on the todo list
// (k_sh_step redoes exactly those); k_sh_scatter buckets the scratch records.
#ifndef SRW_SHQ1_WAVES
#define SRW_SHQ1_WAVES 1          // (minimum waves per SIMD asked of the compiler: 1 = whatever the kernel needs — 86 VGPRs, 5 waves)
#endif
template <bool NT>
__global__ __launch_bounds__(TPB, SRW_SHQ1_WAVES) void k_sh_step_q1(GraphView g, ShardIO io, int32_t first_walk, int32_t step, int32_t last, RngSpec rng, float p,
                                                    SWalker *__restrict__ scratch, unsigned long long *cursor, uint32_t *__restrict__ todo,
                                                    ChainRec *__restrict__ chain, DevCounters *ctr, uint32_t max_ret, uint32_t *__restrict__ many) {
  __shared__ uint32_t pre[SHARD_MAX_WORLD + 1];
  const uint32_t n_in = shard_in_prefix(io, pre);
  unsigned long long steps = 0, dead = 0, reads = 0, n_todo = 0;
  uint32_t lo, hi;
  shard_slice(n_in, TPB, lo, hi);
  for (uint32_t base = lo; base < hi; base += TPB) {
    const uint32_t ri = base + threadIdx.x;
    if (ri >= hi) continue;
    const SWalker wk = shard_in_record(io, pre, ri);
    const Row *rp = row_of(g, wk.curr);
    Row r; r.off = 0; r.deg = 0; r.flags = 0;
    if (rp) r = *rp;
    if (r.deg == 0) { if (step > 1) ++dead; scratch[ri] = shard_dead(wk); continue; }
    bool handed = (r.flags & ROW_IRREGULAR) != 0;
    CfoEnt e; int32_t k = -1;
    if (!handed) {
      const uint32_t iter = (uint32_t)(first_walk + wk.lw % io.batch);
      const uint32_t m = walk_bits24(rng.seed, iter, (uint32_t)rng_source(g, wk.src), (uint32_t)step);
      const CfoEnt *crow = g.cfo + r.off;
      if (step == 1) { unsigned rd; e = cfo_pick<NT>(crow, r.deg, m, rd, k); reads += rd; }
      else {
        uint32_t rv = REV_NONE, pos0 = 0u;
        float w0 = 0.0f;
        if (pair_lookup_lane(g.rh, g.rh_buckets, (uint32_t)((int64_t)wk.prev - g.vmin), (uint32_t)((int64_t)wk.curr - g.vmin), rv, pos0))
          w0 = g.ent[r.off + pos0].w;
        else rv = REV_NONE;
        // many parallel return edges (hub <-> hub multi-edges, a hub's self-loops): every prefix value walks the whole run in ONE
        // lane, and a super-step ends with its slowest lane (RMAT-24: 450 ms per iteration, 140 without them) — one wave per
        // such record instead (k_sh_step_q1w)
        if (rv != REV_NONE && (rv >> 24) > max_ret) { many[atomicAdd(cursor + 3, 1ull)] = ri; continue; }
        double S_tie = 0.0;
        const int why = q1_pick<NT>(g, r, crow, rv, (int32_t)pos0, w0, wk.prev, m, p, e, k, reads, &S_tie);
        if (why == 2) {       // a tie: the exact chain, its quotients computed by the whole GPU (k_chain_d)
          const unsigned long long ci = atomicAdd(cursor + 2, 1ull);
          if (ci < (unsigned long long)CHAIN_CAP) { ChainRec cr; cr.ri = ri; cr.pad = 0u; cr.S = S_tie; chain[ci] = cr; continue; }
        }
        handed = why != 0;
      }
    }
    if (handed) { todo[atomicAdd(cursor + 1, 1ull)] = ri; ++n_todo; continue; }
    scratch[ri] = shard_advance(wk, step, e.id, last != 0);
    ++steps;
  }
  flush_counters(ctr, steps, dead, 0, 0, reads, 0);
  const unsigned long long tot = wave_sum_u64(steps);
  if (lane_id() == 0 && tot) atomicAdd(&ctr->strat[SRW_STRAT_Q1_LANE], tot);
  const unsigned long long nt = wave_sum_u64(n_todo);
  if (lane_id() == 0 && nt) atomicAdd(&ctr->strat[SRW_STAT_HANDED_OVER], nt);
}

__device__ inline SWalker shard_record_uniform(const ShardIO &io, const uint32_t *pre, uint32_t ri) {
  SWalker wk = shard_in_record(io, pre, ri);
  wk.lw = __builtin_amdgcn_readfirstlane(wk.lw); wk.src = __builtin_amdgcn_readfirstlane(wk.src);
  wk.prev = __builtin_amdgcn_readfirstlane(wk.prev); wk.curr = __builtin_amdgcn_readfirstlane(wk.curr);
  return wk;
}
// The records of k_sh_step_q1's "many return edges" list, one wave each (wave_pick_returns): picked -> scratch, a tie -> the chain
// list, a row without usable prefix sums -> the general step's todo list.
__global__ __launch_bounds__(TPB) void k_sh_step_q1w(GraphView g, ShardIO io, int32_t first_walk, int32_t step, int32_t last, RngSpec rng, float p,
                                                     SWalker *__restrict__ scratch, unsigned long long *cursor, const uint32_t *__restrict__ many,
                                                     uint32_t *__restrict__ todo, ChainRec *__restrict__ chain, DevCounters *ctr) {
  __shared__ uint32_t pre[SHARD_MAX_WORLD + 1];
  shard_in_prefix(io, pre);
  const int lane = lane_id();
  const uint32_t n = (uint32_t)cursor[3];
  unsigned long long steps = 0, n_todo = 0;
  for (uint32_t ti = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6); ti < n; ti += gridDim.x * (TPB / 64)) {
    const uint32_t ri = many[ti];
    const SWalker wk = shard_record_uniform(io, pre, ri);
    const Row r = uniform_row(*row_of(g, wk.curr));        // (listed: the row exists and is regular)
    const uint32_t xprev = (uint32_t)((int64_t)wk.prev - g.vmin);
    uint32_t rv = 0u;
    int32_t nr = 0; int64_t so = r.off;
    if (pair_lookup_wave(g.rh, g.rh_buckets, xprev, (uint32_t)((int64_t)wk.curr - g.vmin), rv)) {
      so = r.off + (int64_t)(rv & 0xFFFFFFu); nr = (int32_t)(rv >> 24);
      if (nr >= 255) {                                     // the count saturated: the run of prev in the sorted row
        nr = 0;
        for (int64_t c = so;; c += 64) {
          const unsigned long long m = __ballot(c + lane < r.off + r.deg && g.sids[c + lane] == xprev);
          nr += __popcll(m);
          if (m != ~0ull) break;
        }
      }
    }
    Bias b = make_bias(g, p, 1.0f, wk.prev, true);
    const uint32_t iter = (uint32_t)(first_walk + wk.lw % io.batch);
    const float u = draw_uniform(rng, iter, (uint32_t)__builtin_amdgcn_readfirstlane(rng_source(g, wk.src)), (uint32_t)step);
    unsigned f = 0; double S_tie = 0.0;
    const int32_t k = wave_pick_returns<false>(g, r, b, so, nr, u, f, &S_tie);
    if (k == CHAIN_NEEDED) {
      unsigned long long ci = CHAIN_CAP;
      if (lane == 0) ci = atomicAdd(cursor + 2, 1ull);
      ci = (unsigned long long)__builtin_amdgcn_readfirstlane((int)(ci < (unsigned long long)CHAIN_CAP ? ci : CHAIN_CAP));
      if (ci < (unsigned long long)CHAIN_CAP) { if (lane == 0) { ChainRec cr; cr.ri = ri; cr.pad = 0u; cr.S = S_tie; chain[ci] = cr; } continue; }
    }
    if (k < 0) { if (lane == 0) { todo[atomicAdd(cursor + 1, 1ull)] = ri; ++n_todo; } continue; }
    if (lane == 0) { scratch[ri] = shard_advance(wk, step, g.ent[r.off + k].id, last != 0); ++steps; }
  }
  if (lane == 0 && steps) { atomicAdd(&ctr->steps, steps); atomicAdd(&ctr->strat[SRW_STRAT_PREFIX], steps); }
  if (lane == 0 && n_todo) atomicAdd(&ctr->strat[SRW_STAT_HANDED_OVER], n_todo);
}

// ---- the exact chain for the table steps whose draw sits on a CDF boundary ---------------------------------------------
// RandomSample.sample's running sum (RandomSample.scala:18-22) is sequential by nature, but only its ADDITIONS are: the
// quotients fl(w'_k / S) — the entry loads, the membership probes, the divides: what costs — are independent.  So:
//   k_chain_setup  one thread: row length, scratch offset and first work unit of every listed record (records whose
//                  quotients do not fit the scratch array go to the general step)
//   k_chain_d      the whole GPU: one wave per work unit of 256 candidates computes their quotients into the scratch array
//   k_chain_seq    one wave per record: the chain over the stored quotients, 1024 per round (chain_round_fast: one integer
//                  sum per round while no rounding tie / binade crossing / answer is in it), next round prefetched
// ~3 000 ties per iteration at config 3's size, each up to a million candidates long: one wave alone took 10-40 ms for one.
// pass_cur (whole-graph walks: ALL ties of a launch are listed at once — several GB of quotients at config 3): the records are taken
// in several passes of the chain kernels, each one as many as fit the scratch array; *pass_cur = the first record not taken yet.
__global__ void k_chain_setup(GraphView g, ShardIO io, const ChainRec *__restrict__ list, unsigned long long *cursor /* [1] todo_n, [2] chain_n */,
                              ChainMeta *__restrict__ meta, uint32_t *__restrict__ totals /* [0] work units, [1] records */, long long d_cap,
                              uint32_t *__restrict__ todo, unsigned long long *pass_cur) {
  __shared__ uint32_t pre[SHARD_MAX_WORLD + 1];
  __shared__ int32_t degs[CHAIN_CAP];
  shard_in_prefix(io, pre);
  const unsigned long long n_all = cursor[2];
  const uint32_t n = (uint32_t)(n_all < (unsigned long long)CHAIN_CAP ? n_all : (unsigned long long)CHAIN_CAP);
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {       // the rows' lengths, all lanes (one dependent pair of loads each)
    const SWalker wk = shard_in_record(io, pre, list[i].ri);
    degs[i] = g.rows[(int64_t)wk.curr - g.vmin].deg;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const uint32_t start = pass_cur ? (uint32_t)(*pass_cur < (unsigned long long)n ? *pass_cur : (unsigned long long)n) : 0u;
  uint32_t next_start = start;
  bool stopped = false;
  long long off = 0; uint32_t units = 0;
  for (uint32_t i = 0; i < n; ++i) {
    ChainMeta m; m.d_off = off; m.deg = 0; m.u_off = units;          // deg 0: not in this pass
    if (i >= start && !stopped) {
      const int32_t deg = degs[i];
      if (off + (long long)deg <= d_cap) {
        m.deg = deg; off += (long long)((deg + 255) & ~255); units += (uint32_t)((deg + 255) >> 8); next_start = i + 1;
      } else if (!pass_cur) todo[atomicAdd(cursor + 1, 1ull)] = list[i].ri;     // scratch full: the general step
      else if (off == 0) next_start = i + 1;                                   // longer than the whole scratch array: stays unresolved
      else stopped = true;                                                     // the next pass starts here
    }
    meta[i] = m;
  }
  totals[0] = units; totals[1] = n;
  if (pass_cur) *pass_cur = next_start;
}
__global__ __launch_bounds__(TPB) void k_chain_d(GraphView g, ShardIO io, float p, float q, const ChainRec *__restrict__ list,
                                                 const ChainMeta *__restrict__ meta, const uint32_t *__restrict__ totals, double *__restrict__ D,
                                                 ChainUnits cu) {
  __shared__ uint32_t pre[SHARD_MAX_WORLD + 1];
  shard_in_prefix(io, pre);
  const int lane = lane_id();
  const uint32_t n_units = totals[0], n = totals[1];
  const uint32_t gw = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6), nw = gridDim.x * (TPB / 64);
  uint32_t i = 0;
  for (uint32_t u = gw; u < n_units; u += nw) {
    ChainMeta m = meta[i];                                 // the record unit u belongs to (u grows: i only moves forward)
    while (u >= m.u_off + (uint32_t)((m.deg + 255) >> 8) && i + 1 < n) m = meta[++i];
    const int32_t base4 = (int32_t)(u - m.u_off) * 256;
    const SWalker wk = shard_record_uniform(io, pre, list[i].ri);
    const Row r = uniform_row(g.rows[(int64_t)wk.curr - g.vmin]);
    const bool need = q != 1.0f;                           // q == 1 (k_sh_step_q1's ties): only the return edges are biased
    Row mr; mr.off = 0; mr.deg = 0; mr.flags = 0;
    if (need) mr = uniform_row(g.mrows[(int64_t)wk.prev - g.vmin]);
    Bias b;
    b.p = p; b.q = q; b.prev = wk.prev; b.second_order = true; b.need_member = need; b.vmin = g.vmin;
    b.prev_sids = need ? g.msids + mr.off : nullptr; b.prev_deg = mr.deg; b.prev_hub = mr.flags >> ROW_HUB_SHIFT;
    Member cm; cm.mode = need ? 1 : 0; cm.bm = nullptr; cm.seg_base = 0;
    cm.hub = (b.prev_hub && g.hub_bm) ? g.hub_bm + (int64_t)(b.prev_hub - 1) * g.hub_words : nullptr;
    cm.ehash = g.ehash; cm.ehash_mask = g.ehash_mask;
    if (!cm.hub && !g.ehash && g.bf_off && mr.deg >= BF_MIN_DEG) {
      const uint32_t bo = g.bf_off[(int64_t)wk.prev - g.vmin];
      if (bo != BF_NONE) { cm.bf = g.bf_bits + bo; cm.bf_nw = bf_words(mr.deg); }
    }
    double d4[4];
    chain_quotients4(g.ent + r.off, r.deg, base4, b, list[i].S, &cm, d4);
    double *out = D + m.d_off + base4;
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) out[uu * 64 + lane] = d4[uu];          // (padding up to the unit's 256 slots holds 0.0)
    const double us = wave_sum_f64((d4[0] + d4[1]) + (d4[2] + d4[3]));    // approximate: only places the unit in a binade (k_chain_scan)
    if (lane == 0) cu.usum[u] = us;
  }
}
// One wave per record: the approximate accumulator at every unit's start and end (a plain scan of the units' sums) names the
// binade the unit is expected to run in (-1: the two ends differ).  A guess only: k_chain_seq checks it against the exact accumulator.
__global__ __launch_bounds__(TPB) void k_chain_scan(const ChainMeta *__restrict__ meta, const uint32_t *__restrict__ totals, ChainUnits cu) {
  const int lane = lane_id();
  const uint32_t i = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (i >= totals[1]) return;
  const ChainMeta m = meta[i];
  if (m.deg == 0) return;
  const int32_t nu = (m.deg + 255) >> 8;
  double carry = 0.0;
  for (int32_t base = 0; base < nu; base += 64) {
    const int32_t j = base + lane;
    const double v = j < nu ? cu.usum[m.u_off + j] : 0.0;
    double incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const double t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    const double a0 = carry + (incl - v), a1 = carry + incl;
    const unsigned long long b0 = (unsigned long long)__double_as_longlong(a0), b1 = (unsigned long long)__double_as_longlong(a1);
    const int e0 = (int)((b0 >> 52) & 0x7FFull), e1 = (int)((b1 >> 52) & 0x7FFull);
    if (j < nu) cu.ue[m.u_off + j] = (e0 == e1 && e0 != 0 && e0 != 0x7FF && !(b0 >> 63)) ? e0 : -1;
    carry += readlane_f64(incl, 63);
  }
}
// The whole GPU, one wave per unit: the unit's 256 quotients as ONE integer increment of the accumulator in the guessed binade
// (chain_round_fast's argument: without a rounding tie the maps N -> N + c commute); CHAIN_UNIT_SLOW when an element sits on a
// tie or would leave the binade by itself, or the unit has no guess.
constexpr unsigned long long CHAIN_UNIT_SLOW = ~0ull;
__global__ __launch_bounds__(TPB) void k_chain_u(const uint32_t *__restrict__ totals, const double *__restrict__ D, ChainUnits cu) {
  const int lane = lane_id();
  const uint32_t n_units = totals[0];
  const uint32_t gw = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6), nw = gridDim.x * (TPB / 64);
  for (uint32_t u = gw; u < n_units; u += nw) {
    const int e = cu.ue[u];
    double d4[4];
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) d4[uu] = D[(long long)u * 256 + uu * 64 + lane];
    unsigned long long loc = 0ull; bool odd = e < 0;
    if (e >= 0) {
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        unsigned long long c0 = 0ull, c1 = 0ull;
        chain_elem_map(d4[uu], e - 1023, c0, c1);
        odd |= (c0 != c1) || (c0 >> 53);
        loc += c0;
      }
    }
    const bool slow = __any(odd);
    const unsigned long long tot = wave_sum_u64(loc);
    if (lane == 0) cu.utot[u] = (slow || (tot >> 53)) ? CHAIN_UNIT_SLOW : tot;
  }
}
// One wave per record, 64 units (16 384 quotients) per iteration: lane l holds unit j + l's integer increment; a wave scan gives the
// accumulator after each unit, exactly, while the guessed binade is the accumulator's and the sum stays inside it.  The first unit
// that is slow (a tie inside, no / wrong guess), would leave the binade or reaches p is evaluated element by element
// (chain_group64: the reference's additions) — the first ~20 units of a row (the accumulator climbs through the small binades),
// one per binade crossing afterwards, and the answer's unit.  A 10^6-candidate row: ~64 iterations + ~25 slow units, where
// one integer sum per 1024 quotients took ~1000 dependent rounds (2.7 ms per super-step on RMAT-24's hubs).
__global__ __launch_bounds__(TPB) void k_chain_seq(GraphView g, ShardIO io, int32_t first_walk, int32_t step, int32_t last, RngSpec rng,
                                                   const ChainRec *__restrict__ list, const ChainMeta *__restrict__ meta,
                                                   const uint32_t *__restrict__ totals, const double *__restrict__ D, ChainUnits cu,
                                                   SWalker *__restrict__ scratch, DevCounters *ctr, int strat) {
  __shared__ uint32_t pre[SHARD_MAX_WORLD + 1];
  shard_in_prefix(io, pre);
  const int lane = lane_id();
  const uint32_t i = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (i >= totals[1]) return;
  const ChainMeta m = meta[i];
  if (m.deg == 0) return;                                 // handed to the general step by k_chain_setup
  const uint32_t ri = list[i].ri;
  const SWalker wk = shard_record_uniform(io, pre, ri);
  const Row r = uniform_row(g.rows[(int64_t)wk.curr - g.vmin]);
  const uint32_t iter = (uint32_t)(first_walk + wk.lw % io.batch);
  if (list[i].pad) step = (int32_t)list[i].pad;           // whole-graph walks: every tie has its own step (TieSink)
  const double p = (double)draw_uniform(rng, iter, (uint32_t)__builtin_amdgcn_readfirstlane(rng_source(g, wk.src)), (uint32_t)step);
  const double *d = D + m.d_off;
  const int32_t nu = (r.deg + 255) >> 8;
  double acc = 0.0;
  int32_t k_hit = -1, j = 0;
  [[maybe_unused]] unsigned n_slow = 0;
  while (j < nu && k_hit < 0) {
    const unsigned long long ab = (unsigned long long)__double_as_longlong(acc);
    const int ea = (int)((ab >> 52) & 0x7FFull);
    int f = 0;                                            // units absorbed by this iteration
    if (!(ea == 0 || ea == 0x7FF || (ab >> 63))) {
      const unsigned long long N0 = (ab & ((1ull << 52) - 1ull)) | (1ull << 52);
      const bool valid = j + lane < nu;
      const unsigned long long tot = valid ? cu.utot[m.u_off + j + lane] : 0ull;
      const int eg = valid ? cu.ue[m.u_off + j + lane] : ea;
      const bool slow = valid && (tot == CHAIN_UNIT_SLOW || eg != ea);
      unsigned long long incl = slow ? 0ull : tot;        // (lanes behind the first stop are not used)
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const unsigned long long t = shfl_up_u64(incl, off); if (lane >= off) incl += t; }
      const unsigned long long N = N0 + incl;
      const double a = __longlong_as_double((long long)(((unsigned long long)ea << 52) | (N & ((1ull << 52) - 1ull))));
      const unsigned long long stop = __ballot(slow || (valid && (N >= (1ull << 53) || !(a < p))));
      const int n_valid = min(64, nu - j);
      f = stop ? __ffsll((long long)stop) - 1 : n_valid;
      if (f > 0) acc = readlane_f64(a, f - 1);
      j += f;
      if (!stop) continue;
    }
    // unit j, element by element
    ++n_slow;
#pragma unroll 1
    for (int u = 0; u < 4; ++u) {
      const int32_t b0 = j * 256 + u * 64;
      if (b0 >= r.deg) break;                             // wave-uniform
      const int cnt = min(64, r.deg - b0);
      const double dv = lane < cnt ? d[b0 + lane] : 0.0;
      const int fh = chain_group64(acc, dv, cnt, p);
      if (fh >= 0) { k_hit = b0 + fh; break; }
    }
    ++j;
  }
  if (k_hit < 0) k_hit = 0;                               // edges.head (:24)
  if (lane == 0) {
    const int32_t next = g.ent[r.off + k_hit].id;
    SWalker nw = shard_advance(wk, step, next, last != 0);
    nw.pad0 = k_hit;
    scratch[ri] = nw;
    if (strat >= 0) {                                     // (whole-graph walks: k_walk_general counts the step it takes from here)
      atomicAdd(&ctr->steps, 1ull); atomicAdd(&ctr->fallbacks, 1ull);
      atomicAdd(&ctr->strat[SRW_STRAT_CHAIN], 1ull); atomicAdd(&ctr->strat[strat], 1ull);
    }
#ifdef SRW_PHASE_TIMING
    atomicAdd(&ctr->dbg[20], (unsigned long long)n_slow); atomicAdd(&ctr->dbg[21], (unsigned long long)nu);
#endif
  }
}

// Buckets a super-step's sampled records (scratch, input order) into the destination chunks in ONE pass, like the second
// half of k_sh_step_cfo: per tile of TPB * SH_R records the waves count survivors per destination and returns per home rank
// in LDS, 2 * world threads move the block's counts onto the device-wide chunk cursors, the lanes store straight into the
// chunks; the last block writes the chunk headers and clears the cursors.  Replaces k_sh_offsets + k_sh_bucket (and the
// per-block count matrix) for the table steps, whose records are not sampled by fixed slices.
__global__ __launch_bounds__(TPB) void k_sh_scatter(GraphView g, ShardIO io, int32_t step, const SWalker *__restrict__ recs,
                                                    uint32_t *__restrict__ cursors, ShardDst dst, uint32_t *__restrict__ overflow, DevCounters *ctr) {
  __shared__ uint32_t cnt[2 * SHARD_MAX_WORLD], gbase[2 * SHARD_MAX_WORLD], pre[SHARD_MAX_WORLD + 1];
  __shared__ uint32_t is_last;
  const int lane = lane_id();
  if (threadIdx.x < 2 * SHARD_MAX_WORLD) cnt[threadIdx.x] = 0u;
  const uint32_t n_in = shard_in_prefix(io, pre);          // contains the __syncthreads() cnt needs
  uint32_t lo, hi;
  shard_slice(n_in, TPB * SH_R, lo, hi);
  for (uint32_t base = lo; base < hi; base += TPB * SH_R) {
    SWalker w[SH_R];
    int32_t o[SH_R], hm[SH_R];
    uint32_t wpos[SH_R], rpos[SH_R];
#pragma unroll
    for (int r = 0; r < SH_R; ++r) {
      const uint32_t ri = base + (uint32_t)r * TPB + threadIdx.x;
      o[r] = -1; hm[r] = -1; wpos[r] = 0; rpos[r] = 0;
      w[r].lw = 0; w[r].src = 0; w[r].prev = 0; w[r].curr = 0; w[r].v = 0; w[r].kind = SK_RET; w[r].pad0 = w[r].pad1 = 0;
      if (ri < hi) {
        w[r] = recs[ri];
        if (w[r].kind == SK_WALKER_RET) o[r] = owner_of_tab(w[r].curr, io.world, g.owner_tab, g.vmin, g.n_slots);
        hm[r] = owner_of_tab(w[r].src, io.world, g.owner_tab, g.vmin, g.n_slots);
        if (hm[r] == io.rank) { hm[r] = -1; shard_return_home(io, w[r], step); }
      }
    }
#pragma unroll
    for (int r = 0; r < SH_R; ++r) {
      unsigned long long todo = __ballot(o[r] >= 0);
      while (todo) {
        const int d = __builtin_amdgcn_readlane(o[r], __ffsll((long long)todo) - 1);
        const unsigned long long m = __ballot(o[r] == d);
        uint32_t b0 = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) b0 = atomicAdd(&cnt[d], (uint32_t)__popcll(m));
        b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, leader);
        if (o[r] == d) wpos[r] = b0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
      }
      todo = __ballot(hm[r] >= 0);
      while (todo) {
        const int d = __builtin_amdgcn_readlane(hm[r], __ffsll((long long)todo) - 1);
        const unsigned long long m = __ballot(hm[r] == d);
        uint32_t b0 = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) b0 = atomicAdd(&cnt[SHARD_MAX_WORLD + d], (uint32_t)__popcll(m));
        b0 = (uint32_t)__builtin_amdgcn_readlane((int)b0, leader);
        if (hm[r] == d) rpos[r] = b0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        todo &= ~m;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * io.world) {
      const int d = (int)threadIdx.x < io.world ? (int)threadIdx.x : (int)threadIdx.x - io.world;
      const int idx = (int)threadIdx.x < io.world ? d : SHARD_MAX_WORLD + d;
      const uint32_t c = cnt[idx];
      cnt[idx] = 0u;
      uint32_t gb = 0;
      if (c) {
        gb = atomicAdd(&cursors[idx], c);
        if ((uint64_t)gb + c > (uint64_t)((int)threadIdx.x < io.world ? io.cap_w : io.cap_r)) atomicOr(overflow, 1u);
      }
      gbase[idx] = gb;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SH_R; ++r) {
      if (o[r] >= 0) {
        const uint32_t pos = gbase[o[r]] + wpos[r];
        if (pos < (uint32_t)io.cap_w) reinterpret_cast<WWalker *>(dst.p[o[r]] + 16)[pos] = shard_wire_of(w[r]);
      }
      if (hm[r] >= 0) {
        const uint32_t pos = gbase[SHARD_MAX_WORLD + hm[r]] + rpos[r];
        if (pos < (uint32_t)io.cap_r) reinterpret_cast<WRet *>(dst.p[hm[r]] + 16 + (int64_t)io.cap_w * SW_BYTES)[pos] = shard_ret_of(w[r]);
      }
    }
    __syncthreads();                                      // gbase is rewritten by the next tile
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) is_last = atomicAdd(&cursors[SH_CUR_DONE], 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (is_last) {
    __threadfence();
    if ((int)threadIdx.x < 2 * io.world) {
      const bool rets = (int)threadIdx.x >= io.world;
      const int d = rets ? (int)threadIdx.x - io.world : (int)threadIdx.x;
      const uint32_t total = atomicExch(&cursors[rets ? SHARD_MAX_WORLD + d : d], 0u);
      const uint32_t cap = (uint32_t)(rets ? io.cap_r : io.cap_w);
      reinterpret_cast<uint32_t *>(dst.p[d])[rets ? 1 : 0] = total < cap ? total : cap;
      atomicMax(&ctr->why[rets ? 1 : 0], (unsigned long long)total);      // the fullest chunk of the batch (run_shard_finish, SRW_TIMING)
    }
    if (threadIdx.x == 0) cursors[SH_CUR_DONE] = 0u;
  }
}

// ---- unit hooks ------------------------------------------------------------------------------------------
__global__ void k_hook_pick(const Ent *row, int32_t deg, Bias b, float r, float *out_w, int64_t *index) {
  const int lane = lane_id();
  if (out_w)
    for (int32_t k = lane; k < deg; k += 64) out_w[k] = biased_weight(b, row[k].id, row[k].w);
  if (index) {
    unsigned f = 0;
    int32_t k = wave_pick(row, deg, b, r, f);
    if (lane == 0) *index = k;
  }
}
__global__ void k_hook_rng(uint32_t seed, const uint32_t *iter, const uint32_t *src, const uint32_t *step, int64_t n,
                           float *out) {
  RngSpec rng; rng.mode = 1; rng.const_r = 0.f; rng.seed = seed;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = draw_uniform(rng, iter[i], src[i], step[i]);
}

void read_counters(srw_handle *h, srw_walk_stats *stats) {
  DevCounters c;
  SRW_HIP(hipMemcpyAsync(&c, h->counters.p, sizeof(c), hipMemcpyDeviceToHost, h->stream));
  SRW_HIP(hipStreamSynchronize(h->stream));
#ifdef SRW_PHASE_TIMING
  {
    static const char *nm[10] = {"total", "prefix", "binned:ret-edges", "binned:P1", "binned:P2", "binned:W", "binned:search+resolve",
                                 "scan:fill", "scan:pass1", "scan:pass2"};
    fprintf(stderr, "[phase] wave-ms:");
    for (int i = 0; i < 10; ++i) fprintf(stderr, " %s %.0f", nm[i], (double)c.dbg[i] * 1024.0 / 100e3);
    fprintf(stderr, "\n[phase] W calls %llu elems %llu windows %llu | P1 calls %llu elems %llu | binned steps %llu\n", c.dbg[10], c.dbg[11],
            c.dbg[12], c.dbg[13], c.dbg[14], c.dbg[15]);
    {
      static const char *sn[12] = {"edge_table", "p1", "p2", "w", "p3", "scan", "prefix", "chain", "edge_mask", "-", "-", "-"};
      fprintf(stderr, "[phase] per-strategy wave-ms (whole step) / steps / us per step:");
      for (int i = 0; i < 12; ++i)
        if (c.strat[i]) fprintf(stderr, " %s %.0f / %llu / %.2f", sn[i], (double)c.dbg[24 + i] * 1024.0 / 100e3, c.strat[i],
                                (double)c.dbg[24 + i] * 1024.0 / 100.0 / (double)c.strat[i]);
      fprintf(stderr, "\n");
    }
    if (c.strat[SRW_STRAT_EDGE_TABLE])                   // the lean table kernel's own statistics (valid when it served the table steps)
      fprintf(stderr, "[lean] wave-ms: mask steps %.0f (%llu steps), table steps %.0f (%llu steps; %.0f with deg(prev) > 1024) | chunks %llu: "
              "no specials %llu, N(prev) staged %llu, hub bitmap %llu, edge hash %llu; candidates per chunk %.0f, rounds per chunk %.2f\n",
              (double)c.dbg[1] * 1024.0 / 100e3, c.strat[SRW_STRAT_EDGE_MASK], (double)c.dbg[2] * 1024.0 / 100e3, c.strat[SRW_STRAT_EDGE_TABLE],
              (double)c.dbg[3] * 1024.0 / 100e3, c.dbg[15], c.dbg[10], c.dbg[13], c.dbg[12], c.dbg[14],
              (double)c.dbg[11] / (double)std::max<unsigned long long>(c.dbg[15], 1), (double)c.dbg[16] / (double)std::max<unsigned long long>(c.dbg[15], 1));
    fprintf(stderr, "[phase] W detail wave-ms: wait-B %.0f clear+insert %.0f wait-A %.0f probe %.0f\n", (double)c.dbg[16] * 1024.0 / 100e3,
            (double)c.dbg[17] * 1024.0 / 100e3, (double)c.dbg[18] * 1024.0 / 100e3, (double)c.dbg[19] * 1024.0 / 100e3);
  }
#endif
  if (getenv("SRW_DEBUG_HANDOVER"))
    fprintf(stderr, "[handover] walkers %llu | q1 reasons: irregular row %llu, non-positive sum %llu, boundary draw %llu\n",
            c.strat[SRW_STAT_HANDED_OVER], c.why[0], c.why[1], c.why[2]);
  if (!stats) return;
  stats->n_steps = (int64_t)c.steps; stats->dead_ends = (int64_t)c.dead_ends;
  stats->sum_deg_curr = (int64_t)c.sum_deg_curr; stats->sum_deg_prev = (int64_t)c.sum_deg_prev;
  stats->ent_reads = (int64_t)c.ent_reads; stats->fallbacks = (int64_t)c.fallbacks; stats->trials = (int64_t)c.trials;
  for (int i = 0; i < 12; ++i) stats->strategy_steps[i] = (int64_t)c.strat[i];
  stats->edge_tables = (h->g.has_eb && h->g.use_eb) ? h->g.eb_tables : 0;
  stats->edge_table_bytes = (h->g.has_eb && h->g.use_eb) ? h->g.eb_bytes : 0;
}

void check_params(const srw_walk_params &P) {
  if (P.walk_length < 0) throw Error(SRW_ERR_INVALID, "walk_length must be >= 0");
  // --numWalks 0: the reference's `0 until numWalks` loop runs zero times and still writes an empty path/ + _SUCCESS
  if (P.num_walks < 0) throw Error(SRW_ERR_INVALID, "num_walks must be >= 0");
  if (P.rng_mode != SRW_RNG_CONST && P.rng_mode != SRW_RNG_PHILOX) throw Error(SRW_ERR_INVALID, "bad rng_mode");
  if (P.sampler != SRW_SAMPLER_REFERENCE && P.sampler != SRW_SAMPLER_ALIAS) throw Error(SRW_ERR_INVALID, "bad sampler");
  if (P.sampler == SRW_SAMPLER_ALIAS && P.rng_mode != SRW_RNG_PHILOX)
    throw Error(SRW_ERR_INVALID, "Mode A draws several uniforms per step: it needs SRW_RNG_PHILOX");
}

}  // namespace

// Chain scratch of a handle: record list + meta + totals in one buffer, the quotients of up to d_cap candidates in another,
// the per-unit summaries in a third; behind them the whole-graph walk's tie records (TieSink), their output and a cursor of their own.
namespace {
struct ChainBufs {
  ChainRec *list; ChainMeta *meta; uint32_t *totals; double *D; ChainUnits cu; long long d_cap;
  uint32_t *tie_hdr; WWalker *tie_recs; SWalker *tie_out; unsigned long long *tie_cur; uint32_t *tie_skip;
};
ChainBufs chain_bufs(srw_handle *h) {
  static const long long d_cap = (long long)(getenv("SRW_CHAIN_SCRATCH_MB") ? atof(getenv("SRW_CHAIN_SCRATCH_MB")) : 512.0) * (1 << 20) / 8;
  const size_t n_units = (size_t)(d_cap / 256) + CHAIN_CAP;
  const size_t core = (size_t)CHAIN_CAP * (sizeof(ChainRec) + sizeof(ChainMeta)) + 64 + n_units * 24;
  const size_t tie = 64 + (size_t)CHAIN_CAP * (sizeof(WWalker) + sizeof(SWalker) + 4) + 64;
  h->chain_buf.ensure(core + tie);
  h->chain_d.ensure((size_t)d_cap);
  ChainBufs b;
  char *base = h->chain_buf.p;
  b.list = reinterpret_cast<ChainRec *>(base);
  b.meta = reinterpret_cast<ChainMeta *>(base + (size_t)CHAIN_CAP * sizeof(ChainRec));
  b.totals = reinterpret_cast<uint32_t *>(base + (size_t)CHAIN_CAP * (sizeof(ChainRec) + sizeof(ChainMeta)));
  char *ub = base + (size_t)CHAIN_CAP * (sizeof(ChainRec) + sizeof(ChainMeta)) + 64;
  b.cu.usum = reinterpret_cast<double *>(ub);
  b.cu.utot = reinterpret_cast<unsigned long long *>(ub + n_units * 8);
  b.cu.ue = reinterpret_cast<int32_t *>(ub + n_units * 16);
  b.D = h->chain_d.p; b.d_cap = d_cap;
  char *tb = base + ((core + 63) & ~(size_t)63);
  b.tie_cur = reinterpret_cast<unsigned long long *>(tb);                    // [0..3]: the cursor array of k_chain_setup ([1] skipped, [2] listed)
  b.tie_hdr = reinterpret_cast<uint32_t *>(tb + 32);                         // 16-byte chunk header, the records right behind it
  b.tie_recs = reinterpret_cast<WWalker *>(tb + 48);
  b.tie_out = reinterpret_cast<SWalker *>(tb + 48 + (size_t)CHAIN_CAP * sizeof(WWalker));
  b.tie_skip = reinterpret_cast<uint32_t *>(tb + 48 + (size_t)CHAIN_CAP * (sizeof(WWalker) + sizeof(SWalker)));
  return b;
}
// draws on a CDF boundary (listed by the step kernel at cursor[2]): quotients by the whole GPU, their units summarised, then one
// short sequential pass per record; what does not fit goes onto the todo list `skipped` (count at cursor[1]) — the general step's
void enqueue_chain(srw_handle *h, const ChainBufs &cb, const GraphView &gv, const ShardIO &io, const srw_walk_params &P, int32_t step, int32_t last,
                   const RngSpec &rng, SWalker *scratch, int strat, unsigned long long *cursor, uint32_t *skipped, unsigned long long *pass_cur = nullptr) {
  hipStream_t st = h->stream;
  hipLaunchKernelGGL(k_chain_setup, dim3(1), dim3(256), 0, st, gv, io, cb.list, cursor, cb.meta, cb.totals, cb.d_cap, skipped, pass_cur);
  hipLaunchKernelGGL(k_chain_d, dim3(h->n_cus * 4), dim3(TPB), 0, st, gv, io, P.p, P.q, cb.list, cb.meta, cb.totals, cb.D, cb.cu);
  hipLaunchKernelGGL(k_chain_scan, dim3(CHAIN_CAP / (TPB / 64)), dim3(TPB), 0, st, cb.meta, cb.totals, cb.cu);
  hipLaunchKernelGGL(k_chain_u, dim3(h->n_cus * 4), dim3(TPB), 0, st, cb.totals, (const double *)cb.D, cb.cu);
  hipLaunchKernelGGL(k_chain_seq, dim3(CHAIN_CAP / (TPB / 64)), dim3(TPB), 0, st, gv, io, P.first_walk, step, last, rng, cb.list, cb.meta, cb.totals,
                     (const double *)cb.D, cb.cu, scratch, h->counters.p, strat);
}
}  // namespace

namespace {
struct LaunchInfo { int kind; int record_bytes; };

// Leaves no asynchronous work behind (copies into caller / pinned buffers, kernels writing staging buffers) when a
// pipelined entry point exits — normally or through an exception (I/O error in the writer, HIP error).
struct StreamDrain {
  srw_handle *h;
  explicit StreamDrain(srw_handle *hh) : h(hh) {}
  ~StreamDrain() {
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
  }
};

// Enqueue the walk kernel(s) of num_walks iterations starting at P.first_walk into d_paths / d_lens.
// Compacted ids (graph_build.hip:compact_ids): rank -> input id over the written part of every path, one wave per path.
__global__ void k_paths_to_ids(int32_t *__restrict__ paths, const int32_t *__restrict__ lens, int64_t n_walkers, int64_t stride,
                               const int32_t *__restrict__ orig_id) {
  const int lane = lane_id();
  for (int64_t w = blockIdx.x * (int64_t)(TPB / 64) + (threadIdx.x >> 6); w < n_walkers; w += (int64_t)gridDim.x * (TPB / 64)) {
    const int32_t len = lens[w];
    int32_t *row = paths + w * stride;
    for (int32_t k = lane; k < len; k += 64) row[k] = orig_id[row[k]];
  }
}

LaunchInfo launch_walk(srw_handle *h, const srw_walk_params &P, int32_t num_walks, int32_t first_walk, int32_t *d_paths,
                       int32_t *d_lens) {
  Graph &g = h->g;
  { const char *e = getenv("SRW_DEBUG_CHAIN_DEG"); g.dbg_chain_deg = e && *e ? atoi(e) : 0; }      // tests: every table step on a long row is a "tie"
  hipStream_t st = h->stream;
  const int64_t n_walkers = (int64_t)num_walks * g.n_vertices;
  const bool alias = P.sampler == SRW_SAMPLER_ALIAS;
  bool first_order_compact = false;
  const bool first_order = !alias && (P.p == 1.0f && P.q == 1.0f) && !(P.flags & SRW_WALK_FORCE_GENERAL);
  RngSpec rng; rng.mode = P.rng_mode; rng.const_r = P.const_r; rng.seed = P.seed;
  GraphView gv = g.view();
  if (alias) {
    int64_t blocks = (n_walkers + TPB - 1) / TPB;
    const size_t al_bytes = (size_t)g.n_entries * sizeof(AEnt);
    // cached loads by default: the accepted record's link is a second load of the same sector and should hit L2
    // (measured at weighted RMAT-24, p=.25 q=4: 5.4 G steps/s cached vs 5.3 G nontemporal; p=4 q=.5: 18.9 vs 15.6)
    (void)al_bytes;
    const bool nt = (P.flags & SRW_WALK_NT_LOADS) != 0;
    if (nt)
      hipLaunchKernelGGL(k_walk_alias<true>, dim3((unsigned)blocks), dim3(TPB), 0, st, gv, g.verts.p, g.n_vertices, n_walkers,
                         P.walk_length, first_walk, P.seed, P.p, P.q, d_paths, d_lens, h->counters.p);
    else
      hipLaunchKernelGGL(k_walk_alias<false>, dim3((unsigned)blocks), dim3(TPB), 0, st, gv, g.verts.p, g.n_vertices, n_walkers,
                         P.walk_length, first_walk, P.seed, P.p, P.q, d_paths, d_lens, h->counters.p);
  } else if (first_order) {
    int64_t blocks = (n_walkers + TPB - 1) / TPB;
    // Load policy for the linked records: once the table is far larger than L2 + Infinity Cache (32 + 256 MiB) a
    // record is used once per fetch, and an L1-bypassing load avoids pulling its whole 128-B line (measured on
    // MI355X: RMAT-26 19.6 -> 27.1 G steps/s); cache-resident tables keep the default policy (RMAT-20 35 vs 29).
    const size_t fo_bytes = (size_t)g.n_entries * sizeof(FoEnt);
    const bool nt = (P.flags & SRW_WALK_NT_LOADS) ? true
                    : (P.flags & SRW_WALK_CACHED_LOADS) ? false : fo_bytes > ((size_t)2 << 30);
    const int occ = (P.flags >> 8) & 0xF;   // experiment switch: requested min waves/SIMD (0 = compiler's choice)
    // 16-byte compact records: Philox draws only (p = m * 2^-24), and only if the whole graph qualified at build time
    const bool compact = g.has_cfo && P.rng_mode == SRW_RNG_PHILOX && !(P.flags & SRW_WALK_NO_COMPACT);
    first_order_compact = compact;
#define SRW_LAUNCH_FO(NTV, MW, CP)                                                                                  \
  hipLaunchKernelGGL((k_walk_first_order<NTV, MW, CP>), dim3((unsigned)blocks), dim3(TPB), 0, st, gv, g.verts.p,     \
                     g.n_vertices, n_walkers, P.walk_length, first_walk, rng, d_paths, d_lens, h->counters.p)
    if (compact) {
      const bool ntc = (P.flags & SRW_WALK_NT_LOADS) ? true : (P.flags & SRW_WALK_CACHED_LOADS) ? false
                       : (size_t)g.n_entries * sizeof(CfoEnt) > ((size_t)2 << 30);
      if (ntc) SRW_LAUNCH_FO(true, 1, true); else SRW_LAUNCH_FO(false, 1, true);
    } else if (nt) { if (occ == 8) SRW_LAUNCH_FO(true, 8, false); else if (occ == 7) SRW_LAUNCH_FO(true, 7, false); else SRW_LAUNCH_FO(true, 1, false); }
    else    { if (occ == 8) SRW_LAUNCH_FO(false, 8, false); else if (occ == 7) SRW_LAUNCH_FO(false, 7, false); else SRW_LAUNCH_FO(false, 1, false); }
#undef SRW_LAUNCH_FO
  } else {
    // persistent waves taking walkers from a cursor: enough blocks to fill every CU at the kernel's occupancy
    int64_t blocks = std::min<int64_t>((n_walkers * 64 + TPB - 1) / TPB, (int64_t)h->n_cus * 8);
    h->walk_cursor.ensure(2);
    SRW_HIP(hipMemsetAsync(h->walk_cursor.p, 0, 2 * sizeof(unsigned long long), st));
    const int32_t tune = (int32_t)(((P.flags >> 12) & 15) | ((P.flags & SRW_WALK_NO_BINNED) ? 16 : 0));
    // every (prev -> curr) pair has a table: the lean kernel walks, the general one only redoes what it hands over
    const bool lean = gv.eb_off && g.eb_complete && P.q != 1.0f && tune == 0 && !getenv("SRW_NO_LEAN_KERNEL");
    // p != 1, q == 1: one walker per lane over the first-order guide table + exact prefix sums + return-edge positions
    const bool q1 = P.q == 1.0f && P.p != 1.0f && P.rng_mode == SRW_RNG_PHILOX && g.has_cfo && g.has_pq && g.has_rev &&
                    g.pq_bad_rows == 0 && tune == 0 && !(P.flags & SRW_WALK_NO_PREFIX) && !getenv("SRW_NO_Q1_KERNEL");
    const int32_t *todo = nullptr;
    // Steps whose draw sits on a CDF boundary (the per-lane / lean kernels cannot decide them): recorded by those kernels, resolved by
    // the chain kernels (the whole GPU + one short pass per record, in passes that fit the quotient scratch) and taken from there by
    // k_walk_general when it redoes the walker — one wave alone took up to ~50 ms for the chain over a 10^6-candidate row
    const int32_t *todo_tie = nullptr; const ChainRec *tie_list = nullptr; const SWalker *tie_out = nullptr;
    TieSink tie; tie.hdr = nullptr; tie.recs = nullptr; tie.list = nullptr; tie.cur = nullptr; tie.todo_tie = nullptr;
    ChainBufs cb;
    const bool ties = lean && P.rng_mode == SRW_RNG_PHILOX && !getenv("SRW_NO_TIE_KERNELS");   // (the per-lane q == 1 kernel: measured, no gain — the records cost its registers what the redo saves)
    if (q1 || lean) h->walk_todo.ensure(2 * (size_t)n_walkers);          // handed-over walkers | their tie records
    if (ties) {
      cb = chain_bufs(h);
      SRW_HIP(hipMemsetAsync(cb.tie_cur, 0, 48, st));                                   // cursor array + chunk header
      SRW_HIP(hipMemsetAsync(cb.tie_out, 0xFF, (size_t)CHAIN_CAP * sizeof(SWalker), st));   // kind = -1: not resolved
      tie.hdr = cb.tie_hdr; tie.recs = cb.tie_recs; tie.list = cb.list; tie.cur = cb.tie_cur + 2;
      tie.todo_tie = h->walk_todo.p + n_walkers;
    }
    auto resolve_ties = [&]() {
      if (!ties) return;
      ShardIO io; io.recv = reinterpret_cast<const char *>(cb.tie_hdr); io.chunk_bytes = 0; io.cap_w = CHAIN_CAP; io.cap_r = 0;
      io.world = 1; io.rank = 0; io.batch = 0x7FFFFFFF; io.pt = nullptr; io.lens = nullptr; io.n_rows = 0;      // lw = the iteration's offset
      srw_walk_params Pc = P; Pc.first_walk = first_walk;
      static const int n_pass = getenv("SRW_TIE_PASSES") ? std::max(1, atoi(getenv("SRW_TIE_PASSES"))) : 8;
      for (int pass = 0; pass < n_pass; ++pass)          // (a pass with nothing left is five empty launches)
        enqueue_chain(h, cb, gv, io, Pc, 0, 0, rng, cb.tie_out, -1, cb.tie_cur, cb.tie_skip, cb.tie_cur + 3);
      todo_tie = tie.todo_tie; tie_list = cb.list; tie_out = cb.tie_out;
    };
    if (q1) {
      const int64_t qb = (n_walkers + TPB - 1) / TPB;
      const uint32_t q1_max_ret = getenv("SRW_Q1_MAX_RET") ? (uint32_t)atoi(getenv("SRW_Q1_MAX_RET")) : 16u;   // more parallel return edges: the whole wave takes the step
      if ((size_t)g.n_entries * sizeof(CfoEnt) > ((size_t)2 << 30))
        hipLaunchKernelGGL(k_walk_q1<true>, dim3((unsigned)qb), dim3(TPB), 0, st, gv, g.verts.p, g.n_vertices, n_walkers, P.walk_length,
                           first_walk, rng, P.p, d_paths, d_lens, h->counters.p, h->walk_todo.p, h->walk_cursor.p + 1, q1_max_ret);
      else
        hipLaunchKernelGGL(k_walk_q1<false>, dim3((unsigned)qb), dim3(TPB), 0, st, gv, g.verts.p, g.n_vertices, n_walkers, P.walk_length,
                           first_walk, rng, P.p, d_paths, d_lens, h->counters.p, h->walk_todo.p, h->walk_cursor.p + 1, q1_max_ret);
      todo = h->walk_todo.p;
    }
    if (lean) {
      int64_t lb = std::min<int64_t>((n_walkers * 64 + TPB - 1) / TPB, (int64_t)h->n_cus * 16);
      TabArgs ta;
      ta.g = gv; ta.verts = g.verts.p; ta.n_verts = g.n_vertices; ta.n_walkers = n_walkers; ta.L = P.walk_length; ta.first_walk = first_walk;
      ta.rng = rng; ta.p = P.p; ta.q = P.q; ta.paths = d_paths; ta.lens = d_lens; ta.ctr = h->counters.p; ta.cursor = h->walk_cursor.p;
      ta.todo = h->walk_todo.p; ta.todo_n = h->walk_cursor.p + 1; ta.tie = tie;
      // SRW_TABLE_GROUPS=1: one walker per 16 lanes (walk_groups.hip) — measured and not kept as the default, profiles/r06_group_kernel.md
      const bool groups = getenv("SRW_TABLE_GROUPS") && atoi(getenv("SRW_TABLE_GROUPS")) != 0;
      // One walker per LANE (walk_lanes.hip, mode 2: table steps per lane, the rest served by the wave) where the standing tables have
      // chunks of 64 candidates — config 3: 585 against 602 ms per iteration, directed RMAT-23 ef 27 (p = 4, q = .5): 132 against 206 ms;
      // one walker per WAVE (k_walk_tables) where a graph that fills the GPU left only chunks of >= 256: every table step would be served
      // (config 5's stand-in: 4.13 against 3.34 s) — profiles/r06_lane_kernel.md.  SRW_TABLE_LANES=<mode> forces the lane kernel (bit 0: whole
      // rows per lane, bit 1: table steps per lane), -1 the wave kernel.
      const int lanes = getenv("SRW_TABLE_LANES") ? atoi(getenv("SRW_TABLE_LANES")) : (gv.ebp.min_sh <= 6 ? 2 : -1);
      if (getenv("SRW_TABLE_ROUNDS") && atoi(getenv("SRW_TABLE_ROUNDS")) != 0) {        // the table walk in rounds of two kernels (walk_rounds.hip)
        launch_walk_tables_rounds(h, ta, gv.bf_off != nullptr, getenv("SRW_LANE_CSH") ? atoi(getenv("SRW_LANE_CSH")) : 6, st);
      } else if (lanes >= 0) {
        launch_walk_tables_lanes(ta, gv.bf_off != nullptr, lanes, getenv("SRW_LANE_CSH") ? atoi(getenv("SRW_LANE_CSH")) : 6, h->n_cus, st);
      } else if (groups) {
        launch_walk_tables_groups(ta, gv.bf_off != nullptr, h->n_cus, st);
      } else if (gv.bf_off) {
        hipLaunchKernelGGL((k_walk_tables<true>), dim3((unsigned)lb), dim3(TPB), 0, st, ta);
      } else {
        hipLaunchKernelGGL((k_walk_tables<false>), dim3((unsigned)lb), dim3(TPB), 0, st, ta);
      }
      SRW_HIP(hipMemsetAsync(h->walk_cursor.p, 0, sizeof(unsigned long long), st));
      todo = h->walk_todo.p;
      resolve_ties();
    }
    hipLaunchKernelGGL(k_walk_general, dim3((unsigned)blocks), dim3(TPB), 0, st, gv, g.verts.p, g.n_vertices, n_walkers,
                       P.walk_length, first_walk, rng, P.p, P.q, d_paths, d_lens, h->counters.p, h->walk_cursor.p, tune, todo,
                       h->walk_cursor.p + 1, todo_tie, tie_list, tie_out);
  }
  SRW_HIP(hipGetLastError());
  LaunchInfo li;
  li.kind = alias ? 3 : first_order ? 1 : 2;
  li.record_bytes = first_order_compact ? 16 : (first_order || alias) ? 32 : 0;
  // compacted ids: the kernels walked over ranks; the paths leave with the ids of the input
  if (g.compact && n_walkers > 0) {
    const int64_t nb = std::min<int64_t>((n_walkers + TPB / 64 - 1) / (TPB / 64), (int64_t)h->n_cus * 32);
    hipLaunchKernelGGL(k_paths_to_ids, dim3((unsigned)nb), dim3(TPB), 0, st, d_paths, d_lens, n_walkers, (int64_t)P.walk_length + 2,
                       (const int32_t *)g.orig_id.p);
    SRW_HIP(hipGetLastError());
  }
  return li;
}

void prepare_tables(srw_handle *h, const srw_walk_params &P) {
  // SRW_TIMING: where the cold start of a call goes, phase by phase (each phase synchronises the stream when timing is on)
  const bool timing = getenv("SRW_TIMING") != nullptr;
  auto t_phase = std::chrono::steady_clock::now();
  auto phase = [&](const char *name) {
    if (!timing) return;
    (void)hipStreamSynchronize(h->stream);
    const auto t = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(t - t_phase).count();
    if (ms >= 20.0) fprintf(stderr, "[timing] prepare_tables: %s %.0f ms\n", name, ms);
    t_phase = t;
  };
  const bool alias = P.sampler == SRW_SAMPLER_ALIAS;
  const bool first_order = !alias && (P.p == 1.0f && P.q == 1.0f) && !(P.flags & SRW_WALK_FORCE_GENERAL);
  if (first_order) build_first_order_tables(h, P.rng_mode != SRW_RNG_PHILOX || (P.flags & SRW_WALK_NO_COMPACT));
  else build_membership(h);                           // sorted rows: general and alias kernels only
  // p != 1, q == 1 (k_walk_q1): the compact first-order records (guide + linked rows) and the return-edge positions
  if (!alias && !first_order && P.q == 1.0f && P.p != 1.0f && P.rng_mode == SRW_RNG_PHILOX && h->cfg.world == 1 &&
      !(P.flags & (SRW_WALK_NO_PREFIX | SRW_WALK_NO_COMPACT)) && !getenv("SRW_NO_Q1_KERNEL")) {
    build_first_order_tables(h, false);
    build_rev_table(h);
  }
  phase("membership / first-order tables / return edges");
  if (alias) build_alias_tables(h);
  phase("alias tables");
  const bool general = !alias && !first_order;
  // optional accelerators, most valuable first (each one skips itself when HBM is short):
  // exact base prefix sums for the search samplers ...
  if (general && !(P.p == 1.0f && P.q == 1.0f) && !(P.flags & SRW_WALK_NO_PREFIX)) build_pq_tables(h, P.p, P.q);
  else if (general) h->g.has_pq = false;
  // ... the edge hash set answers "x in N(prev)?" in one probe: Mode A's rejection test, and the general kernel's
  // candidate-by-candidate membership (small rows, the located chunk of the binned search)
  bool want_ehash = P.q != 1.0f && !(P.flags & SRW_WALK_NO_EDGE_HASH) && (alias || (general && h->cfg.world == 1));
  // neighbor-set bitmaps of the hub rows: the general kernel's "x in N(prev)" for steps that come from a hub
  const bool want_hub = general && P.q != 1.0f && h->cfg.world == 1 && !(P.flags & SRW_WALK_NO_HUB_BITMAPS);
  const bool want_eb = general && P.q != 1.0f && h->cfg.world == 1 && h->g.has_pq && !(P.flags & SRW_WALK_NO_EDGE_TABLES) &&
                       !(P.flags & SRW_WALK_NO_BINNED);
  const int eb_mode = (P.flags & SRW_WALK_EDGE_TABLES_ALL) ? 1 : 0;
  phase("prefix sums of the (p, q) base weights");
  if (want_eb) build_unit_ids(h);                     // unit-weight graphs: 4-byte ids for the table steps (before the tables are sized)
  phase("unit-weight ids");
  const char *env_hub = getenv("SRW_HUB_BUDGET_GB"), *env_cap = getenv("SRW_EB_CHUNKS");
  // The edge hash (8 B x 2-3 per entry) against table resolution: when a COMPLETE 64-chunk set of per-edge tables fits only
  // without the hash, the hash goes — the located chunks' probes of a long non-hub N(prev) fall back to the sorted row, and
  // every step still gains from chunks half as long (config 5's stand-in: 34 GB of hash; 32 chunks + hash 1.67e8 steps/s,
  // 32 chunks without 1.56e8, 64 chunks without 2.0e8 — s68, s69).
  // ---- what HBM is spent on, in this order -------------------------------------------------------------------------
  // A table step is bound by the memory requests it issues (profiles/r04_request_attribution.md), and what removes requests is table
  // resolution: (1) a COMPLETE set (a cut set sends its uncovered steps to the on-the-fly samplers of the monolithic kernel) with as many
  // chunks per table as fit (256 / 128 / 64 / 32), (2) the smallest chunk (64 / 128 / 256 candidates), (3) chunk masks for the rows up
  // to 16 384 / 4 096 candidates whose N(prev) is too long for the LDS staging, (4) finer tables (up to 4 096 / 1 024 / 512 chunks) for
  // the unmasked pairs with a long N(prev).  The edge hash (8 B x 2-3 per entry: one probe per candidate of a located chunk that has
  // neither mask nor staged N(prev) nor hub bitmap) is kept only when dropping it would not buy a finer set — without it the long rows
  // get their neighbor-set filters (4 B per entry) in front of the sorted rows (config 5's stand-in: 34 GB of hash; s68, s69, r04 s109).
  // The hub bitmaps take what is left (at least 16 GB are set aside for them).
  bool drop_ehash = false;
  size_t hub_cap = want_eb ? (size_t)16 << 30 : (size_t)64 << 30;
  size_t table_cap_used = 0;
  int eb_cap = EB_BINS;
  if (env_hub && *env_hub) hub_cap = (size_t)(atof(env_hub) * (double)((size_t)1 << 30));
  if (env_cap && *env_cap) eb_cap = atoi(env_cap);
  struct TabPlan { int cap = 0, min_sh = 8, cm = 0, fine = 0, ratio = 0; size_t need = 0; bool complete = false; };
  auto finer = [](const TabPlan &a, const TabPlan &b) {      // a strictly finer than b
    if (a.complete != b.complete) return a.complete;
    if (a.cap != b.cap) return a.cap > b.cap;
    if (a.min_sh != b.min_sh) return a.min_sh < b.min_sh;
    return std::min(a.cm, 16384) > std::min(b.cm, 16384);          // (long masks, the finer tables of the unmasked pairs and the mask ratio are refinements: not worth the hash — config 3: ratio 16
                                 //  without the hash 771 ms per iteration, ratio 4 with it 711 ms, r04 s112)
  };
  // what the walk itself allocates after the tables: one call's paths and lengths (+ hand-over lists, chain scratch, the build's HBM-scratch bins)
  size_t reserve = (size_t)P.num_walks * (size_t)h->g.n_vertices * ((size_t)P.walk_length + 3) * 4 + ((size_t)8 << 30);
  if (reserve > ((size_t)64 << 30)) reserve = (size_t)64 << 30;      // (srw_walk_to_host / _and_save stream one iteration at a time)
  if (const char *r = getenv("SRW_EB_RESERVE_GB"); r && *r) reserve = (size_t)(atof(r) * (double)((size_t)1 << 30));
  h->g.eb_reserve = reserve;
  const int64_t job_walks = std::max<int64_t>(h->planned_walks > 0 ? h->planned_walks : 10, P.num_walks);   // srw_plan_walks
  std::map<std::tuple<int, int, int, int, int>, size_t> size_cache;
  auto set_size = [&](int cap, int sh, int cm, int fine, int ratio = 0) {   // bytes of the complete set under this geometry (one pass over the entries; cached)
    Graph &g = h->g;
    const auto key = std::make_tuple(cap, sh, cm, fine, ratio);
    auto it = size_cache.find(key);
    if (it != size_cache.end()) return it->second;
    g.eb_min_sh_sel = sh; g.eb_cm_sel = cm; g.eb_fine_cap_sel = fine; g.eb_cm_ratio_sel = ratio;
    const size_t n = edge_tables_full_bytes(h, eb_mode, cap);
    if (getenv("SRW_TIMING")) fprintf(stderr, "[timing] table plan: %d chunks of >= %d, masks <= %d (ratio %d), fine %d: %.1f GB\n", cap, 1 << sh, cm, ratio, fine, (double)n / 1e9);
    size_cache[key] = n;
    return n;
  };
  auto plan_tables = [&](size_t free_b) {                    // the finest geometry whose complete set fits into free_b next to the reserve and the bitmaps' minimum
    TabPlan t;
    const size_t ceiling = (size_t)230 << 30;                 // (eb_off counts 64-byte units in 32 bits: 256 GiB)
    auto fits = [&](size_t n, size_t margin) { return n > 0 && n < ceiling && free_b > n + reserve + margin; };
    const bool fixed_cap = env_cap && *env_cap;
    if (fixed_cap) { t.cap = eb_cap; t.need = set_size(t.cap, 8, 0, 0); t.complete = fits(t.need, (size_t)8 << 30); }
    else
      for (int c : {256, 128, 64, 32}) {
        const size_t n = set_size(c, 8, 0, 0);
        if (fits(n, (size_t)(c > 64 ? 40 : 8) << 30)) { t.cap = c; t.need = n; t.complete = true; break; }
      }
    if (!t.complete) { if (!t.cap) { t.cap = EB_BINS; t.need = set_size(t.cap, 8, 0, 0); } return t; }
    if (eb_mode) return t;
    if (!getenv("SRW_EB_MIN_SH"))
      for (int sh : {6, 7}) {
        const size_t n = set_size(t.cap, sh, 0, 0);
        if (fits(n, (size_t)40 << 30)) { t.min_sh = sh; t.need = n; break; }
      }
    // chunk masks: every row up to 16 384 candidates (4 096 if memory is short).  LONG masks (round 6: rows beyond 16 384 candidates for the
    // pairs with a long N(prev) — the hub -> hub pairs whose located chunks probe prev's bitmap once per candidate) are built when asked for
    // (SRW_EB_CM_MAX > 16 384) but never planned: at config 3 they need +100 GB up to 32 768 candidates (no change in the iteration: 588 ms),
    // +150 GB up to 131 072, +184 GB for every row — the probes come from the longest rows (profiles/r06_long_masks.md)
    if (!getenv("SRW_EB_CM_MAX"))
      for (int cm : {16384, 4096}) {
        const size_t n = set_size(t.cap, t.min_sh, cm, 0);
        if (fits(n, (size_t)24 << 30)) { t.cm = cm; t.need = n; break; }
      }
    // Finer tables for the unmasked pairs with a long N(prev).  Up to 512 chunks they fill the wave's LDS bins like every other table and
    // cost nothing to build (config 3: +40 ms, +15 GB, 643 -> 602 ms per iteration); beyond that the bins live in an HBM scratch
    // (+1.6 - 8 s by box and 40 GB more for 602 -> 573 ms): only for a job long enough to pay for them — srw_plan_walks,
    // profiles/r05_table_build.md.
    if (!getenv("SRW_EB_FINE_CAP"))
      for (int fc : {4096, 1024, 512}) {
        if (fc <= t.cap) break;
        if (fc > BIN_CAP && job_walks < 64) continue;
        const size_t n = set_size(t.cap, t.min_sh, t.cm, fc);
        if (fits(n, (size_t)24 << 30)) { t.fine = fc; t.need = n; break; }
      }
    if (t.cm && !getenv("SRW_EB_CM_RATIO"))
      for (int ratio : {16, 4}) {
        const size_t n = set_size(t.cap, t.min_sh, t.cm, t.fine, ratio);
        if (fits(n, (size_t)24 << 30)) { t.ratio = ratio; t.need = n; break; }
      }
    return t;
  };
  TabPlan plan;
  bool standing = false;
  if (want_eb) {
    Graph &g = h->g;
    uint32_t pb, qb; memcpy(&pb, &P.p, 4); memcpy(&qb, &P.q, 4);
    standing = g.has_eb && !g.eb_sharded && g.eb_pbits == pb && g.eb_qbits == qb && g.eb_mode == eb_mode && (!want_hub || g.has_hub);
    if (standing) {
      drop_ehash = g.eb_no_ehash;                                  // standing tables: as they were built, with the bitmaps they were built with
      eb_cap = g.eb_cap;
      if (want_hub) hub_cap = g.hub_budget_cap;
    } else {
      uint64_t slots = 1024;
      while (slots < (uint64_t)g.n_entries + (uint64_t)g.n_entries / 2) slots <<= 1;      // build_edge_hash's sizing
      const size_t eh_bytes = (size_t)slots * 8;
      size_t free_b = 0, total_b = 0;
      SRW_HIP(hipMemGetInfo(&free_b, &total_b));
      free_b += g.hub_bm.n * sizeof(uint32_t) + g.eb_bins.n * sizeof(double) + g.em_bits.n * sizeof(uint32_t) + g.eb_off.n * sizeof(uint32_t);
      if (g.has_ehash) free_b += g.ehash.n * sizeof(uint64_t);     // free_b: with neither hash nor tables nor bitmaps nor filters
      if (g.has_bf) free_b += (g.bf_off.n + g.bf_bits.n) * sizeof(uint32_t);
      const size_t filters = (size_t)g.n_entries * 4 + (size_t)g.n_slots * 4;
      const TabPlan with = (want_ehash && free_b > eh_bytes) ? plan_tables(free_b - eh_bytes) : TabPlan();
      const TabPlan without = plan_tables(free_b > filters ? free_b - filters : 0);
      // equal plans: with chunk masks the few probes left are one request each in the hash (config 3: 711 ms with it, 771 ms without, r04 s112);
      // without masks every located chunk probes all of its candidates, and the filters answer most of those from L2 (config 5's stand-in, the
      // same 128 x 256 plan: 4 560 ms without the hash, 5 555 ms with it, r04 u2)
      drop_ehash = want_ehash && (finer(without, with) || (!finer(with, without) && without.complete && without.cm == 0 && !eb_mode));
      plan = (want_ehash && !drop_ehash) ? with : without;
      if (getenv("SRW_TIMING"))
        fprintf(stderr, "[timing] table plan: %.1f GB free; with the edge hash (%.1f GB): %s %d chunks of >= %d, masks <= %d (ratio %d), fine %d (%.1f GB); without: %s %d chunks of >= %d, "
                "masks <= %d (ratio %d), fine %d (%.1f GB) -> %s\n", (double)free_b / 1e9, (double)eh_bytes / 1e9, with.complete ? "complete" : "cut", with.cap, 1 << with.min_sh, with.cm, with.ratio, with.fine,
                (double)with.need / 1e9, without.complete ? "complete" : "cut", without.cap, 1 << without.min_sh, without.cm, without.ratio, without.fine, (double)without.need / 1e9,
                drop_ehash ? "the hash goes" : "the hash stays");
      eb_cap = plan.cap ? plan.cap : eb_cap;
      g.eb_min_sh_sel = plan.min_sh; g.eb_cm_sel = plan.cm; g.eb_fine_cap_sel = plan.fine; g.eb_cm_ratio_sel = plan.ratio;
      if (plan.need > ((size_t)160 << 30)) table_cap_used = (size_t)230 << 30;
      const size_t used = (drop_ehash || !want_ehash) ? filters : eh_bytes;
      const size_t keep = plan.need + reserve + ((size_t)8 << 30) + used;
      if (!(env_hub && *env_hub) && want_hub && plan.need > 0 && free_b > keep + ((size_t)16 << 30))
        hub_cap = std::min<size_t>(free_b - keep, (size_t)96 << 30);
    }
  }
  if (want_eb && want_ehash && getenv("SRW_EB_DROP_EHASH")) drop_ehash = true;      // tests: the traded configuration on any graph
  if (drop_ehash) {
    want_ehash = false;
    if (h->g.has_ehash) { h->g.ehash.release(); h->g.has_ehash = false; }
  }
  phase("table plan (sizing passes)");
  if (want_ehash) build_edge_hash(h);
  h->g.use_ehash = want_ehash;
  phase("edge hash");
  // no edge hash for the table steps (traded above, or not wanted): the long rows' neighbor-set filters answer most of the
  // located chunks' probes from L2 (config 5's stand-in: 2.0e8 -> 2.73e8 steps/s, 3 GB).  With the hash they are not worth
  // their registers (config 3: -1 ... -4 %): k_walk_tables<false>.
  if (want_eb && !want_ehash && !getenv("SRW_NO_ROW_FILTERS")) build_row_filters(h);
  phase("row filters");
  if (want_hub) build_hub_bitmaps(h, ((P.flags >> 15) & 1) ? 1 : 1024, hub_cap);
  h->g.use_hub = want_hub;
  phase("hub bitmaps");
  // ... and, last (they take what HBM is left), the per-edge bias tables: the most expensive (prev, curr) pairs get
  // their N(prev) ∩ N(curr) corrections precomputed once per (p, q) instead of once per visit
  if (want_eb) {
    h->g.eb_budget_gb = table_cap_used ? (table_cap_used >> 30) : drop_ehash ? 200 : 160;
    // The sizing above works from hipMemGetInfo; if an allocation of the build fails all the same (fragmentation), the
    // walk goes on with a coarser set, or with none (the on-the-fly samplers) — an optional accelerator never fails a walk.
    for (int attempt = 0; attempt < 2; ++attempt) {
      try { build_edge_tables(h, P.p, P.q, eb_mode, attempt == 0 ? eb_cap : 32); break; }
      catch (const Error &e) {
        // a mapping call of the progressively mapped table buffer refused for another reason than memory (vm_buf.h): once more, the
        // buffer as one hipMalloc — an optional accelerator never fails a walk
        const bool remap = e.code == SRW_ERR_HIP && vm_buf_broken().load() && attempt == 0 && std::string(e.what()).find("mapping the table buffer") != std::string::npos;
        if (e.code != SRW_ERR_NOMEM && !remap) throw;
        (void)hipGetLastError();
        (void)hipStreamSynchronize(h->stream);          // (segments of the build may be running over the chunks that did get mapped)
        if (remap) {
          h->g.eb_bins.release(); h->g.em_bits.release(); h->g.has_eb = false;
          if (getenv("SRW_TIMING")) fprintf(stderr, "[timing] per-edge tables: %s — once more with one allocation\n", e.what());
          attempt = -1;                                   // (the loop's ++ makes it attempt 0 again, now without the mapped range)
          continue;
        }
        Graph &g = h->g;
        g.eb_bins.release(); g.em_bits.release(); g.has_eb = false; g.eb_complete = false; g.eb_tables = 0; g.eb_bytes = 0;
        h->g.eb_budget_gb = 160; g.eb_min_sh_sel = 8; g.eb_cm_sel = 0; g.eb_fine_cap_sel = 0; g.eb_cm_ratio_sel = 0;
        if (getenv("SRW_TIMING")) fprintf(stderr, "[timing] per-edge tables: %s — %s\n", e.what(), attempt == 0 && eb_cap > 32 ? "retrying with 32 chunks" : "walking without them");
        if (eb_cap <= 32) break;
      }
    }
    h->g.eb_no_ehash = drop_ehash;
  }
  h->g.use_eb = want_eb;
  phase("per-edge tables");
}
double timed_prepare_tables(srw_handle *h, const srw_walk_params &P) {   // builders synchronise the stream themselves
  const auto t0 = std::chrono::steady_clock::now();
  prepare_tables(h, P);
  SRW_HIP(hipStreamSynchronize(h->stream));
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace

void run_walk(srw_handle *h, const srw_walk_params &P, srw_walk_stats *stats) {
  Graph &g = h->g;
  if (!g.loaded) throw Error(SRW_ERR_INVALID, "no graph loaded");
  if (h->cfg.world != 1) throw Error(SRW_ERR_INVALID, "srw_walk needs a whole-graph handle (world == 1); use srw_shard_*");
  check_params(P);
  hipStream_t st = h->stream;
  const int64_t n_walkers = (int64_t)P.num_walks * g.n_vertices;
  if (n_walkers >= ((int64_t)1 << 31)) throw Error(SRW_ERR_INVALID, "more than 2^31 walkers in one call: lower num_walks");
  const int32_t stride = P.walk_length + 2;
  if (n_walkers == 0) {   // empty graph or num_walks == 0: nothing to walk
    h->res.n_walkers = 0; h->res.stride = stride; h->res.valid = true;
    if (stats) { memset(stats, 0, sizeof(*stats)); }
    return;
  }
  const double setup_ms = timed_prepare_tables(h, P);
  h->res.valid = false;
  h->res.paths.ensure((size_t)n_walkers * stride);
  h->res.lens.ensure((size_t)n_walkers);
  h->res.n_walkers = n_walkers; h->res.stride = stride;
  h->counters.ensure(1);
  SRW_HIP(hipMemsetAsync(h->counters.p, 0, sizeof(DevCounters), st));
  SRW_HIP(hipEventRecord(h->ev0, st));
  LaunchInfo li = launch_walk(h, P, P.num_walks, P.first_walk, h->res.paths.p, h->res.lens.p);
  SRW_HIP(hipEventRecord(h->ev1, st));
  srw_walk_stats local;
  srw_walk_stats *s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  read_counters(h, s);
  float ms = 0.f;
  SRW_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  s->kernel_ms = ms; s->setup_ms = setup_ms; s->n_walkers = n_walkers; s->kernel_kind = li.kind; s->record_bytes = li.record_bytes;
  h->res.valid = true;
}

// numWalks iterations streamed to the host: kernel of iteration i on the compute stream, D2H of iteration i-1 on the
// copy stream, two staging buffers; events order "kernel done -> copy" and "copy done -> buffer reuse".
void run_walk_to_host(srw_handle *h, const srw_walk_params &P, int32_t *paths, int32_t *lens, srw_walk_stats *stats) {
  Graph &g = h->g;
  if (!g.loaded) throw Error(SRW_ERR_INVALID, "no graph loaded");
  if (h->cfg.world != 1) throw Error(SRW_ERR_INVALID, "srw_walk_to_host needs a whole-graph handle (world == 1)");
  check_params(P);
  hipStream_t st = h->stream;
  const int64_t nv = g.n_vertices;
  if (nv >= ((int64_t)1 << 31)) throw Error(SRW_ERR_INVALID, "too many vertices");
  const int32_t stride = P.walk_length + 2;
  if (nv == 0 || P.num_walks == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return; }
  const double setup_ms = timed_prepare_tables(h, P);
  if (!h->copy_stream) SRW_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    h->stage_paths[i].ensure((size_t)nv * stride);
    h->stage_lens[i].ensure((size_t)nv);
    if (!h->stage_done[i]) SRW_HIP(hipEventCreateWithFlags(&h->stage_done[i], hipEventDisableTiming));
    if (!h->kernel_done[i]) SRW_HIP(hipEventCreateWithFlags(&h->kernel_done[i], hipEventDisableTiming));
  }
  h->res.valid = false;
  h->counters.ensure(1);
  StreamDrain drain(h);
  SRW_HIP(hipMemsetAsync(h->counters.p, 0, sizeof(DevCounters), st));
  SRW_HIP(hipEventRecord(h->ev0, st));
  LaunchInfo li{0, 0};
  for (int32_t it = 0; it < P.num_walks; ++it) {
    const int b = it & 1;
    if (it >= 2) SRW_HIP(hipStreamWaitEvent(st, h->stage_done[b], 0));           // buffer b was copied out
    li = launch_walk(h, P, 1, P.first_walk + it, h->stage_paths[b].p, h->stage_lens[b].p);
    SRW_HIP(hipEventRecord(h->kernel_done[b], st));
    SRW_HIP(hipStreamWaitEvent(h->copy_stream, h->kernel_done[b], 0));
    SRW_HIP(hipMemcpyAsync(paths + (size_t)it * nv * stride, h->stage_paths[b].p, (size_t)nv * stride * 4,
                           hipMemcpyDeviceToHost, h->copy_stream));
    SRW_HIP(hipMemcpyAsync(lens + (size_t)it * nv, h->stage_lens[b].p, (size_t)nv * 4, hipMemcpyDeviceToHost, h->copy_stream));
    SRW_HIP(hipEventRecord(h->stage_done[b], h->copy_stream));
  }
  SRW_HIP(hipEventRecord(h->ev1, st));
  SRW_HIP(hipStreamSynchronize(h->copy_stream));
  srw_walk_stats local;
  srw_walk_stats *s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  read_counters(h, s);
  float ms = 0.f;
  SRW_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  s->kernel_ms = ms; s->setup_ms = setup_ms; s->n_walkers = (int64_t)P.num_walks * nv; s->kernel_kind = li.kind; s->record_bytes = li.record_bytes;
}

// randomWalk + save fused and streamed (Main.doRandomWalk, M/Main.scala:53-62): the paths never exist as a whole on
// the host.  Per walk iteration: kernel on the compute stream -> D2H into a pinned ring slot on the copy stream ->
// the host formats and appends the PREVIOUS iteration's slice to <output>/path/part-* while the GPU works on this one.
void run_walk_and_save(srw_handle *h, const srw_walk_params &P, const char *output_dir, int n_parts, bool write_crc,
                       srw_walk_stats *stats, int64_t *dead_per_iter) {
  Graph &g = h->g;
  if (!g.loaded) throw Error(SRW_ERR_INVALID, "no graph loaded");
  if (h->cfg.world != 1) throw Error(SRW_ERR_INVALID, "srw_walk_and_save needs a whole-graph handle (world == 1)");
  check_params(P);
  hipStream_t st = h->stream;
  const int64_t nv = g.n_vertices;
  const int32_t stride = P.walk_length + 2;
  PathWriter writer(output_dir, n_parts, (int64_t)P.num_walks * nv, write_crc);   // fails first if <output>/path exists
  if (nv == 0 || P.num_walks == 0) { writer.close(); if (stats) memset(stats, 0, sizeof(*stats)); return; }   // empty part-00000 + _SUCCESS
  const double setup_ms = timed_prepare_tables(h, P);
  if (!h->copy_stream) SRW_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  const size_t need = (size_t)nv * stride * 4;
  bool device_format = (P.flags & SRW_WALK_DEVICE_FORMAT) != 0;
  const size_t cap = format_capacity(nv, stride, g.id_lo, g.id_hi);
  if (device_format) {        // two text slots in HBM: fall back to the host formatter when they do not fit
    size_t free_b = 0, total_b = 0;
    SRW_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t have = h->fmt_text[0].n + h->fmt_text[1].n;
    const size_t slots = P.num_walks > 1 ? 2 : 1;
    if (slots * cap > have && free_b < slots * cap - have + need * slots + ((size_t)4 << 30)) device_format = false;
  }
  const int n_slots = P.num_walks > 1 ? 2 : 1;             // (a single iteration never touches the second staging / text slot: tens of GB at the headline's size)
  for (int i = 0; i < 2; ++i) {
    if (i < n_slots) { h->stage_paths[i].ensure((size_t)nv * stride); h->stage_lens[i].ensure((size_t)nv); }
    if (!h->stage_done[i]) SRW_HIP(hipEventCreateWithFlags(&h->stage_done[i], hipEventDisableTiming));
    if (!h->kernel_done[i]) SRW_HIP(hipEventCreateWithFlags(&h->kernel_done[i], hipEventDisableTiming));
    // pinned ring: the ids only travel to the host when the host formats them; the lengths always do (dead-end counts)
    if (!device_format && h->pin_cap < need) {
      if (h->pin_paths[i]) (void)hipHostFree(h->pin_paths[i]);
      h->pin_paths[i] = nullptr;
      SRW_HIP(hipHostMalloc((void **)&h->pin_paths[i], need ? need : 4, hipHostMallocDefault));
    }
    if (h->pin_lens_cap < (size_t)nv) {
      if (h->pin_lens[i]) (void)hipHostFree(h->pin_lens[i]);
      h->pin_lens[i] = nullptr;
      SRW_HIP(hipHostMalloc((void **)&h->pin_lens[i], (size_t)nv * 4 + 4, hipHostMallocDefault));
    }
  }
  if (!device_format) h->pin_cap = std::max(h->pin_cap, need);
  h->pin_lens_cap = std::max(h->pin_lens_cap, (size_t)nv);
  h->res.valid = false;
  h->counters.ensure(1);
  StreamDrain drain(h);
  SRW_HIP(hipMemsetAsync(h->counters.p, 0, sizeof(DevCounters), st));
  SRW_HIP(hipEventRecord(h->ev0, st));
  LaunchInfo li{0, 0};
  if (device_format) {
    // Device-side formatter (path_format.hip): the GPU turns iteration `it` into text while the host copies out and
    // writes the text of iteration `it - 1`; the host never touches the ids.
    for (int i = 0; i < n_slots; ++i) { h->fmt_text[i].ensure(cap); h->fmt_len[i].ensure((size_t)nv + 1); h->fmt_off[i].ensure((size_t)nv + 1); }
    const int32_t N = P.num_walks;
    auto launch = [&](int32_t it) {
      const int b = it & 1;
      li = launch_walk(h, P, 1, P.first_walk + it, h->stage_paths[b].p, h->stage_lens[b].p);
      format_paths_device(h, h->stage_paths[b].p, h->stage_lens[b].p, nv, stride, h->fmt_len[b].p, h->fmt_off[b].p,
                          h->fmt_text[b].p);
      SRW_HIP(hipEventRecord(h->kernel_done[b], st));
    };
    // The text leaves the device in slices of whole lines through the ring of pinned buffers (path_format.hip:drain_text): the next
    // slices are copied while the writer's threads put the earlier ones into their part files.
    const size_t slice_cap = text_slice_cap(stride, cap);
    ensure_pinned_text(h, slice_cap, (size_t)nv + 1);
    launch(0);
    if (N > 1) launch(1);
    for (int32_t k = 0; k < N; ++k) {
      const int b = k & 1;
      const auto tk0 = std::chrono::steady_clock::now();
      SRW_HIP(hipEventSynchronize(h->kernel_done[b]));                      // walk + format of iteration k done
      if (getenv("SRW_TIMING")) fprintf(stderr, "[timing] walk_and_save: waited %.0f ms for walk + format of iteration %d\n",
                                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tk0).count(), k);
      unsigned long long *off = h->pin_off[b];
      SRW_HIP(hipMemcpyAsync(off, h->fmt_off[b].p, ((size_t)nv + 1) * 8, hipMemcpyDeviceToHost, h->copy_stream));
      SRW_HIP(hipMemcpyAsync(h->pin_lens[b], h->stage_lens[b].p, (size_t)nv * 4, hipMemcpyDeviceToHost, h->copy_stream));
      SRW_HIP(hipStreamSynchronize(h->copy_stream));
      drain_text(h, writer, h->fmt_text[b].p, off, nv, slice_cap, [&] { if (k + 2 < N) launch(k + 2); });   // every byte of device slot b is out: reuse it
      if (dead_per_iter) {
        int64_t dead = 0;
        for (int64_t i = 0; i < nv; ++i) dead += (h->pin_lens[b][i] >= 2 && h->pin_lens[b][i] < stride);
        dead_per_iter[k] = dead;
      }
    }
    SRW_HIP(hipEventRecord(h->ev1, st));
    writer.close();
    srw_walk_stats local;
    srw_walk_stats *s = stats ? stats : &local;
    memset(s, 0, sizeof(*s));
    read_counters(h, s);
    float ms = 0.f;
    SRW_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    s->kernel_ms = ms; s->setup_ms = setup_ms; s->n_walkers = (int64_t)P.num_walks * nv; s->kernel_kind = li.kind; s->record_bytes = li.record_bytes;
    return;
  }
  auto consume = [&](int32_t it) {      // host side of iteration `it`: wait for its slice, format + append
    const int b = it & 1;
    SRW_HIP(hipEventSynchronize(h->stage_done[b]));
    if (dead_per_iter) {
      int64_t dead = 0;
      for (int64_t i = 0; i < nv; ++i) dead += (h->pin_lens[b][i] >= 2 && h->pin_lens[b][i] < stride);
      dead_per_iter[it] = dead;
    }
    writer.append(h->pin_paths[b], h->pin_lens[b], nv, stride);
  };
  for (int32_t it = 0; it < P.num_walks; ++it) {
    const int b = it & 1;
    // device staging slot b is free once its previous copy finished; the pinned slot b once the host consumed it
    if (it >= 2) SRW_HIP(hipStreamWaitEvent(st, h->stage_done[b], 0));
    li = launch_walk(h, P, 1, P.first_walk + it, h->stage_paths[b].p, h->stage_lens[b].p);
    SRW_HIP(hipEventRecord(h->kernel_done[b], st));
    if (it >= 2) consume(it - 2);                                   // frees pinned slot b before it is overwritten
    SRW_HIP(hipStreamWaitEvent(h->copy_stream, h->kernel_done[b], 0));
    SRW_HIP(hipMemcpyAsync(h->pin_paths[b], h->stage_paths[b].p, need, hipMemcpyDeviceToHost, h->copy_stream));
    SRW_HIP(hipMemcpyAsync(h->pin_lens[b], h->stage_lens[b].p, (size_t)nv * 4, hipMemcpyDeviceToHost, h->copy_stream));
    SRW_HIP(hipEventRecord(h->stage_done[b], h->copy_stream));
  }
  SRW_HIP(hipEventRecord(h->ev1, st));
  for (int32_t it = std::max(0, P.num_walks - 2); it < P.num_walks; ++it) consume(it);
  writer.close();
  srw_walk_stats local;
  srw_walk_stats *s = stats ? stats : &local;
  memset(s, 0, sizeof(*s));
  read_counters(h, s);
  float ms = 0.f;
  SRW_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  s->kernel_ms = ms; s->setup_ms = setup_ms; s->n_walkers = (int64_t)P.num_walks * nv; s->kernel_kind = li.kind; s->record_bytes = li.record_bytes;
}

// ---- vertex-sharded walk: host side of one rank (see the kernels above) -------------------------------------------
void shard_layout(const srw_handle *h, int32_t batch, double slack, srw_shard_layout *out) {
  const int64_t world = h->cfg.world;
  if (batch < 1) throw Error(SRW_ERR_INVALID, "batch must be >= 1");
  if (!(slack >= 1.0)) slack = 1.25;
  // walkers alive at any time <= batch * nVertices, spread over world^2 (sender, receiver) pairs; owner = id mod world
  // (or the recorded partition) mixes hubs and leaves, so the pairs are even up to sampling noise
  const double per_pair = (double)batch * (double)h->g.n_vertices / (double)(world * world);
  const int64_t cap = (int64_t)(per_pair * slack) + 4096;
  if (cap * world >= ((int64_t)1 << 31)) throw Error(SRW_ERR_INVALID, "shard chunks too large (world * capacity must stay below 2^31 records): lower the batch");
  out->cap_walkers = cap; out->cap_rets = cap;
  out->chunk_bytes = 16 + cap * SW_BYTES + cap * PR_BYTES;
}

namespace {
ShardIO make_io(const srw_handle *h, int32_t batch, const srw_shard_layout &lay, const void *d_recv, int32_t *d_lens) {
  ShardIO io;
  io.recv = (const char *)d_recv; io.chunk_bytes = lay.chunk_bytes; io.cap_w = (int32_t)lay.cap_walkers; io.cap_r = (int32_t)lay.cap_rets;
  io.world = h->cfg.world; io.rank = h->cfg.rank; io.batch = batch;
  io.pt = h->shard_pt.p; io.lens = d_lens; io.n_rows = h->g.n_local_vertices * batch;
  return io;
}
void check_shard(const srw_handle *h, int32_t batch, const srw_shard_layout &lay) {
  if (!h->g.loaded) throw Error(SRW_ERR_INVALID, "no graph loaded");
  if (h->cfg.world > SHARD_MAX_WORLD) throw Error(SRW_ERR_INVALID, "world larger than 64 shards");
  if (batch < 1 || lay.cap_walkers < 1 || lay.cap_rets < 1 || lay.chunk_bytes != 16 + lay.cap_walkers * SW_BYTES + lay.cap_rets * PR_BYTES)
    throw Error(SRW_ERR_INVALID, "bad shard layout");
  if ((int64_t)batch * h->g.n_local_vertices >= ((int64_t)1 << 31)) throw Error(SRW_ERR_INVALID, "batch * local vertices must stay below 2^31");
}
}  // namespace

// Seeds this rank's batch * n_local walkers into its receive buffer, path slot 0 and lens; clears the counters.
// p = q = 1, Philox draws and linked compact records on every shard (srw_shard_rows_commit): the fused kernel
static bool shard_fo_linked(const srw_handle *h, const srw_walk_params &P) {
  return h->g.cfo_linked && P.p == 1.0f && P.q == 1.0f && !(P.flags & (SRW_WALK_FORCE_GENERAL | SRW_WALK_NO_COMPACT)) &&
         P.rng_mode == SRW_RNG_PHILOX;
}

void run_shard_begin(srw_handle *h, const srw_walk_params &P, int32_t batch, const srw_shard_layout &lay, void *d_recv,
                     int32_t *d_paths, int32_t *d_lens, int64_t stride) {
  check_shard(h, batch, lay);
  Graph &g = h->g;
  hipStream_t st = h->stream;
  const int64_t n = g.n_local_vertices * batch;
  h->counters.ensure(1);
  h->shard_flag.ensure(1);
  SRW_HIP(hipMemsetAsync(h->counters.p, 0, sizeof(DevCounters), st));
  SRW_HIP(hipMemsetAsync(h->shard_flag.p, 0, 4, st));
  h->shard_pt.ensure((size_t)std::max<int64_t>(n, 1) * (size_t)stride);                // slot-major staging of this batch's paths (k_sh_apply)
  const ShardIO io = make_io(h, batch, lay, d_recv, d_lens);
  const int blocks = (int)std::min<int64_t>(std::max<int64_t>((n + TPB - 1) / TPB, 1), 8192);
  const bool linked = shard_fo_linked(h, P);
  h->shard_cur.ensure((size_t)SH_CUR_DONE + 1);
  SRW_HIP(hipMemsetAsync(h->shard_cur.p, 0, ((size_t)SH_CUR_DONE + 1) * 4, st));
  hipLaunchKernelGGL(k_sh_seed, dim3(blocks), dim3(TPB), 0, st, g.verts.p, g.n_local_vertices, io, (char *)d_recv, h->shard_pt.p, d_lens, stride,
                     linked ? (const Row *)g.rows.p : (const Row *)nullptr, g.vmin, h->shard_flag.p);
  SRW_HIP(hipGetLastError());
}


// One super-step, enqueued on the handle's stream without any host synchronisation: returns of the previous
// super-step applied, every incoming walker sampled once, walkers and path returns bucketed into dst[0 .. world).
void run_shard_superstep(srw_handle *h, const srw_walk_params &P, int32_t batch, int32_t step, const srw_shard_layout &lay,
                         const void *d_recv, void *const *dst, int32_t *d_paths, int32_t *d_lens, int64_t stride) {
  check_shard(h, batch, lay);
  check_params(P);
  if (step < 1 || step > P.walk_length + 1) throw Error(SRW_ERR_INVALID, "step out of range");
  Graph &g = h->g;
  hipStream_t st = h->stream;
  const int32_t world = h->cfg.world;
  const bool first_order = P.p == 1.0f && P.q == 1.0f && !(P.flags & SRW_WALK_FORCE_GENERAL);
  const bool linked = shard_fo_linked(h, P);
  if (world > 1 && P.q != 1.0f && !g.mrows.p)
    throw Error(SRW_ERR_INVALID, "this shard was loaded with SRW_CFG_NO_MEMBERSHIP: it can only run walks with q == 1 "
                                 "(q != 1 needs the neighbor sets of vertices the shard does not own)");
  if (first_order) { build_first_order_tables(h, true); g.use_eb = false; }
  else {
    build_membership(h);
    if (!(P.p == 1.0f && P.q == 1.0f) && !(P.flags & SRW_WALK_NO_PREFIX)) build_pq_tables(h, P.p, P.q);
    else g.has_pq = false;
    // q != 1: the per-edge tables of the pairs into this shard's rows (built at the first super-step of a (p, q))
    if (P.sampler == SRW_SAMPLER_REFERENCE) prepare_shard_tables(h, P); else g.use_eb = false;
  }
  const bool tables = !first_order && g.has_eb && g.use_eb && g.eb_sharded && P.q != 1.0f;
  // p != 1, q == 1: one record per lane when every row holds the prefix-sum certificate and the compact records exist
  bool q1 = false;
  if (!first_order && P.q == 1.0f && P.p != 1.0f && P.rng_mode == SRW_RNG_PHILOX && P.sampler == SRW_SAMPLER_REFERENCE && g.has_pq &&
      g.pq_bad_rows == 0 && ((P.flags >> 12) & 15) == 0 && !(P.flags & (SRW_WALK_NO_PREFIX | SRW_WALK_NO_COMPACT | SRW_WALK_NO_BINNED)) &&
      !getenv("SRW_NO_Q1_KERNEL") && g.n_entries > 0 && build_local_cfo(h)) {
    build_shard_rev_hash(h);
    q1 = true;
  }
  { const char *e = getenv("SRW_DEBUG_CHAIN_DEG"); g.dbg_chain_deg = e && *e ? atoi(e) : 0; }
  const ShardIO io = make_io(h, batch, lay, d_recv, d_lens);
  ShardDst sd;
  for (int d = 0; d < SHARD_MAX_WORLD; ++d) sd.p[d] = d < world ? (char *)dst[d] : nullptr;
  const int n_blocks = h->n_cus * 4;
  h->shard_scratch.ensure((size_t)world * (size_t)lay.cap_walkers * sizeof(SWalker));
  SWalker *scratch = reinterpret_cast<SWalker *>(h->shard_scratch.p);
  h->shard_blk.ensure((size_t)n_blocks * 2 * world);
  h->shard_flag.ensure(1);
  RngSpec rng; rng.mode = P.rng_mode; rng.const_r = P.const_r; rng.seed = P.seed;
  const int32_t last = step == P.walk_length + 1 ? 1 : 0;
  // SRW_SHARD_PROFILE=1 (debug): per-kernel hipEvent times, synchronising after each kernel, printed at the last step
  static const bool prof = getenv("SRW_SHARD_PROFILE") != nullptr;
  double (&acc)[4] = h->shard_prof_acc, (&mx)[4] = h->shard_prof_mx;      // per handle: one host thread per device calls this (cluster.cpp)
  auto timed = [&](int slot, auto &&launch) {
    if (!prof) { launch(); return; }
    SRW_HIP(hipEventRecord(h->ev0, st)); launch(); SRW_HIP(hipEventRecord(h->ev1, st)); SRW_HIP(hipEventSynchronize(h->ev1));
    float ms = 0.f; SRW_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1)); acc[slot] += ms; mx[slot] = std::max(mx[slot], (double)ms);
    if (slot == 1 && getenv("SRW_SHARD_PROFILE_STEPS")) fprintf(stderr, "[shard step] rank %d step %d: %.2f ms\n", h->cfg.rank, step, ms);
  };
  const int64_t n_rows = g.n_local_vertices * batch;
  if (step > 1) timed(0, [&] { hipLaunchKernelGGL(k_sh_apply, dim3(n_blocks), dim3(TPB), 0, st, io, h->shard_pt.p, d_lens, n_rows, step - 1); });
  if (linked) {      // sampling + bucketing in one pass; no scratch, no per-block counts
    h->shard_cur.ensure((size_t)SH_CUR_DONE + 1);
    timed(1, [&] {
      if ((size_t)g.n_entries * sizeof(CfoEnt) > ((size_t)2 << 30))
        hipLaunchKernelGGL(k_sh_step_cfo<true>, dim3(n_blocks), dim3(TPB), 0, st, g.view(), io, P.first_walk, step, last, rng,
                           h->shard_cur.p, sd, h->shard_flag.p, h->counters.p);
      else
        hipLaunchKernelGGL(k_sh_step_cfo<false>, dim3(n_blocks), dim3(TPB), 0, st, g.view(), io, P.first_walk, step, last, rng,
                           h->shard_cur.p, sd, h->shard_flag.p, h->counters.p);
    });
    SRW_HIP(hipGetLastError());
    if (prof && last)
      fprintf(stderr, "[shard profile] rank %d: apply %.1f ms, fused step %.1f ms (cumulative)\n", h->cfg.rank, acc[0], acc[1]);
    return;
  }
  if (q1) {          // per-lane step -> ties through the chain kernels, the rest it hands over through the general step -> one fused bucketing pass
    h->walk_cursor.ensure(4);                       // [1] todo records, [2] chain records, [3] records with many return edges
    const size_t n_rec = (size_t)world * (size_t)lay.cap_walkers;
    h->walk_todo.ensure(2 * n_rec);                 // todo list | many-returns list
    uint32_t *many_list = (uint32_t *)h->walk_todo.p + n_rec;
    h->shard_cur.ensure((size_t)SH_CUR_DONE + 1);
    SRW_HIP(hipMemsetAsync(h->walk_cursor.p, 0, 4 * sizeof(unsigned long long), st));
    const ChainBufs cb = chain_bufs(h);
    ChainRec *chain_list = cb.list;
    const GraphView gv = g.view();
    // a latency-bound kernel of fixed slices: exactly as many blocks as are resident at once
    int (&q1_occ)[2] = h->q1_occ;                     // per handle (one host thread per device)
    const bool ntq = (size_t)g.n_entries * sizeof(CfoEnt) > ((size_t)2 << 30);
    if (!q1_occ[ntq]) {
      int nb = 0;
      if (ntq) SRW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sh_step_q1<true>, TPB, 0));
      else SRW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_sh_step_q1<false>, TPB, 0));
      q1_occ[ntq] = std::max(1, nb);
      if (const char *e = getenv("SRW_SH_Q1_BLOCKS"); e && *e) q1_occ[ntq] = std::max(1, atoi(e));
    }
    const int qb = h->n_cus * q1_occ[ntq];
    const uint32_t q1_max_ret = getenv("SRW_Q1_MAX_RET") ? (uint32_t)atoi(getenv("SRW_Q1_MAX_RET")) : 16u;
    timed(1, [&] {
      if (ntq)
        hipLaunchKernelGGL(k_sh_step_q1<true>, dim3(qb), dim3(TPB), 0, st, gv, io, P.first_walk, step, last, rng, P.p, scratch, h->walk_cursor.p,
                           (uint32_t *)h->walk_todo.p, chain_list, h->counters.p, q1_max_ret, many_list);
      else
        hipLaunchKernelGGL(k_sh_step_q1<false>, dim3(qb), dim3(TPB), 0, st, gv, io, P.first_walk, step, last, rng, P.p, scratch, h->walk_cursor.p,
                           (uint32_t *)h->walk_todo.p, chain_list, h->counters.p, q1_max_ret, many_list);
      hipLaunchKernelGGL(k_sh_step_q1w, dim3(h->n_cus * 8), dim3(TPB), 0, st, gv, io, P.first_walk, step, last, rng, P.p, scratch, h->walk_cursor.p,
                         (const uint32_t *)many_list, (uint32_t *)h->walk_todo.p, chain_list, h->counters.p);
    });
    timed(2, [&] { enqueue_chain(h, cb, gv, io, P, step, last, rng, scratch, (int)SRW_STRAT_Q1_LANE, h->walk_cursor.p, (uint32_t *)h->walk_todo.p); });
    timed(2, [&] {
      hipLaunchKernelGGL(k_sh_step, dim3(n_blocks), dim3(TPB), 0, st, gv, io, P.first_walk, step, last, rng, P.p, P.q, scratch, h->shard_blk.p,
                         h->counters.p, (const uint32_t *)h->walk_todo.p, (const unsigned long long *)(h->walk_cursor.p + 1));
    });
    timed(3, [&] { hipLaunchKernelGGL(k_sh_scatter, dim3(n_blocks), dim3(TPB), 0, st, gv, io, step, scratch, h->shard_cur.p, sd, h->shard_flag.p, h->counters.p); });
    SRW_HIP(hipGetLastError());
    if (prof && last)
      fprintf(stderr, "[shard profile] rank %d: apply %.1f ms, per-lane q = 1 step %.1f ms, general step (handed over) %.1f ms, scatter %.1f ms (cumulative)\n", h->cfg.rank,
              acc[0], acc[1], acc[2], acc[3]);
    return;
  }
  if (tables) {      // lean table step (persistent waves) -> the records without a table through the general step -> one fused bucketing pass
    h->walk_cursor.ensure(4);                       // [0] record cursor, [1] todo records, [2] chain records
    h->walk_todo.ensure((size_t)world * (size_t)lay.cap_walkers);
    h->shard_cur.ensure((size_t)SH_CUR_DONE + 1);
    SRW_HIP(hipMemsetAsync(h->walk_cursor.p, 0, 4 * sizeof(unsigned long long), st));
    const ChainBufs cb = chain_bufs(h);
    ChainRec *chain_list = cb.list;
    const GraphView gv = g.view();
    const char *sge = getenv("SRW_SH_GRAB");
    const int grab_n = std::min(64, sge && *sge ? std::max(1, atoi(sge)) : SH_GRAB);   // (<= 64: one record per lane in the BATCH prologue)
    const char *sbe = getenv("SRW_SH_BATCH");             // (read per super-step: tools/shard_tables_bench.py alternates the variants on one set of tables)
    const int sh_batch = sbe && *sbe ? atoi(sbe) : 2;
    const char *sbl = getenv("SRW_SH_BLOCKS");            // (per super-step, like the two below: tests alternate the variants on one handle)
    const int tb_mult = sbl && *sbl ? std::max(1, atoi(sbl)) : 8;
    const int tb = h->n_cus * tb_mult;
    timed(1, [&] {
      ShTabArgs ta;
      ta.g = gv; ta.io = io; ta.first_walk = P.first_walk; ta.step = step; ta.last = last; ta.rng = rng; ta.p = P.p; ta.q = P.q; ta.scratch = scratch;
      ta.cursor = h->walk_cursor.p; ta.todo = (uint32_t *)h->walk_todo.p; ta.ctr = h->counters.p; ta.grab_n = grab_n; ta.chain = chain_list;
      if (sh_batch == 2) {
        if (gv.bf_off) hipLaunchKernelGGL((k_sh_step_tab<true, 2>), dim3(tb), dim3(TPB), 0, st, ta);
        else hipLaunchKernelGGL((k_sh_step_tab<false, 2>), dim3(tb), dim3(TPB), 0, st, ta);
      } else if (sh_batch == 1) {
        if (gv.bf_off) hipLaunchKernelGGL((k_sh_step_tab<true, 1>), dim3(tb), dim3(TPB), 0, st, ta);
        else hipLaunchKernelGGL((k_sh_step_tab<false, 1>), dim3(tb), dim3(TPB), 0, st, ta);
      } else {
        if (gv.bf_off) hipLaunchKernelGGL((k_sh_step_tab<true, 0>), dim3(tb), dim3(TPB), 0, st, ta);
        else hipLaunchKernelGGL((k_sh_step_tab<false, 0>), dim3(tb), dim3(TPB), 0, st, ta);
      }
    });
    // draws on a CDF boundary of a table step
    timed(2, [&] { enqueue_chain(h, cb, gv, io, P, step, last, rng, scratch, (int)SRW_STRAT_EDGE_TABLE, h->walk_cursor.p, (uint32_t *)h->walk_todo.p); });
    timed(2, [&] {      // (the few records without a table, or whose tie is not a table step's)
      hipLaunchKernelGGL(k_sh_step, dim3(n_blocks), dim3(TPB), 0, st, gv, io, P.first_walk, step, last, rng, P.p, P.q, scratch, h->shard_blk.p,
                         h->counters.p, (const uint32_t *)h->walk_todo.p, (const unsigned long long *)(h->walk_cursor.p + 1));
    });
    timed(3, [&] { hipLaunchKernelGGL(k_sh_scatter, dim3(n_blocks), dim3(TPB), 0, st, gv, io, step, scratch, h->shard_cur.p, sd, h->shard_flag.p, h->counters.p); });
    SRW_HIP(hipGetLastError());
    if (prof && last)
      fprintf(stderr, "[shard profile] rank %d: apply %.1f ms, table step %.1f ms (longest super-step %.1f ms), chain + general step (ties, todo) %.1f ms, scatter %.1f ms (cumulative)\n", h->cfg.rank,
              acc[0], acc[1], mx[1], acc[2], acc[3]);
    return;
  }
  timed(1, [&] {
    if (first_order) {
      // records larger than the caches are read once per fetch: L1-bypassing loads (as k_walk_first_order)
      if ((size_t)g.n_entries * sizeof(FoEnt) > ((size_t)2 << 30))
        hipLaunchKernelGGL(k_sh_step_fo<true>, dim3(n_blocks), dim3(TPB), 0, st, g.view(), io, P.first_walk, step, last, rng,
                           scratch, h->shard_blk.p, h->counters.p);
      else
        hipLaunchKernelGGL(k_sh_step_fo<false>, dim3(n_blocks), dim3(TPB), 0, st, g.view(), io, P.first_walk, step, last, rng,
                           scratch, h->shard_blk.p, h->counters.p);
    } else
      hipLaunchKernelGGL(k_sh_step, dim3(n_blocks), dim3(TPB), 0, st, g.view(), io, P.first_walk, step, last, rng, P.p, P.q,
                         scratch, h->shard_blk.p, h->counters.p, (const uint32_t *)nullptr, (const unsigned long long *)nullptr);
  });
  timed(2, [&] { hipLaunchKernelGGL(k_sh_offsets, dim3(1), dim3(1024), 0, st, h->shard_blk.p, n_blocks, io, sd, h->shard_flag.p); });
  timed(3, [&] {
    hipLaunchKernelGGL(k_sh_bucket, dim3(n_blocks), dim3(TPB), 0, st, g.view(), io, first_order ? TPB : TPB / 64, step,
                       scratch, h->shard_blk.p, sd);
  });
  SRW_HIP(hipGetLastError());
  if (prof && last) {
    fprintf(stderr, "[shard profile] rank %d: apply %.1f ms, step %.1f ms, offsets %.1f ms, bucket %.1f ms (cumulative)\n", h->cfg.rank, acc[0],
            acc[1], acc[2], acc[3]);
  }
}

// After the exchange that follows the last super-step: its path returns.
void run_shard_flush(srw_handle *h, const srw_walk_params &P, int32_t batch, const srw_shard_layout &lay, const void *d_recv,
                     int32_t *d_paths, int32_t *d_lens, int64_t stride) {
  check_shard(h, batch, lay);
  const ShardIO io = make_io(h, batch, lay, d_recv, d_lens);
  const int64_t n_rows = h->g.n_local_vertices * batch;
  hipLaunchKernelGGL(k_sh_apply, dim3(h->n_cus * 4), dim3(TPB), 0, h->stream, io, h->shard_pt.p, d_lens, n_rows, P.walk_length + 1);
  if (n_rows > 0) {      // the staging becomes the caller's [row][L + 2] matrix (-1 beyond each row's length)
    const int64_t tb = std::min<int64_t>((n_rows + 63) / 64, (int64_t)h->n_cus * 16);
    hipLaunchKernelGGL(k_sh_transpose, dim3((unsigned)tb), dim3(TPB), 0, h->stream, (const int32_t *)h->shard_pt.p, (const int32_t *)d_lens, n_rows, stride, d_paths);
  }
  SRW_HIP(hipGetLastError());
  // compacted ids: the home rank's paths are complete now (one flush per begin); they leave with the ids of the input
  const int64_t n = h->g.n_local_vertices * batch;
  if (h->g.compact && n > 0) {
    const int64_t nb = std::min<int64_t>((n + TPB / 64 - 1) / (TPB / 64), (int64_t)h->n_cus * 32);
    hipLaunchKernelGGL(k_paths_to_ids, dim3((unsigned)nb), dim3(TPB), 0, h->stream, d_paths, d_lens, n, stride, (const int32_t *)h->g.orig_id.p);
    SRW_HIP(hipGetLastError());
  }
}

// Synchronises the handle's stream; counters accumulated since run_shard_begin and the overflow flag.
void run_shard_finish(srw_handle *h, srw_walk_stats *stats, int32_t *overflow) {
  srw_walk_stats local; srw_walk_stats *s = stats ? stats : &local; memset(s, 0, sizeof(*s));
  uint32_t flag = 0;
  h->shard_flag.ensure(1);
  SRW_HIP(hipMemcpyAsync(&flag, h->shard_flag.p, 4, hipMemcpyDeviceToHost, h->stream));
  read_counters(h, s);                                  // synchronises
  if (getenv("SRW_TIMING")) {
    unsigned long long fill[2] = {0, 0};
    SRW_HIP(hipMemcpy(fill, h->counters.p->why, 16, hipMemcpyDeviceToHost));
    if (fill[0] || fill[1]) fprintf(stderr, "[shard %d/%d] fullest chunk of the batch: %llu walkers, %llu returns%s\n", h->cfg.rank, h->cfg.world, fill[0], fill[1], flag ? " (OVERFLOW)" : "");
  }
  if (overflow) *overflow = (int32_t)flag;
}

void hook_sample(srw_handle *h, const float *w, int64_t n, float r, int64_t *index) {
  hook_second_order(h, 1.0f, 1.0f, 0, nullptr, 0, nullptr, w, n, r, nullptr, index);
}

void hook_second_order(srw_handle *h, float p, float q, int32_t prev_id, const int32_t *prev_ids, int64_t n_prev,
                       const int32_t *curr_ids, const float *curr_w, int64_t n, float r, float *out_w,
                       int64_t *index) {
  if (n <= 0) { if (index) *index = -1; return; }
  if (n > 0x7FFFFFFF || n_prev > 0x7FFFFFFF) throw Error(SRW_ERR_INVALID, "list too long");
  hipStream_t st = h->stream;
  std::vector<Ent> row((size_t)n);
  int32_t vmin = prev_id;
  for (int64_t k = 0; k < n; ++k) { row[k].id = curr_ids ? curr_ids[k] : (int32_t)k; row[k].w = curr_w[k]; vmin = std::min(vmin, row[k].id); }
  for (int64_t k = 0; k < n_prev; ++k) vmin = std::min(vmin, prev_ids[k]);
  std::vector<uint32_t> sp((size_t)n_prev);
  for (int64_t k = 0; k < n_prev; ++k) sp[k] = (uint32_t)((int64_t)prev_ids[k] - vmin);
  std::sort(sp.begin(), sp.end());
  DevBuf<Ent> d_row; DevBuf<uint32_t> d_sp; DevBuf<float> d_w; DevBuf<int64_t> d_idx;
  d_row.alloc((size_t)n); d_sp.alloc((size_t)n_prev); d_w.alloc((size_t)n); d_idx.alloc(1);
  SRW_HIP(hipMemcpyAsync(d_row.p, row.data(), (size_t)n * sizeof(Ent), hipMemcpyHostToDevice, st));
  if (n_prev) SRW_HIP(hipMemcpyAsync(d_sp.p, sp.data(), (size_t)n_prev * 4, hipMemcpyHostToDevice, st));
  Bias b; b.p = p; b.q = q; b.prev = prev_id; b.second_order = (prev_ids != nullptr) || (p != 1.0f) || (q != 1.0f);
  b.need_member = b.second_order && q != 1.0f; b.prev_sids = d_sp.p; b.prev_deg = (int32_t)n_prev; b.vmin = vmin;
  hipLaunchKernelGGL(k_hook_pick, dim3(1), dim3(64), 0, st, d_row.p, (int32_t)n, b, r, out_w ? d_w.p : nullptr,
                     index ? d_idx.p : nullptr);
  SRW_HIP(hipGetLastError());
  if (out_w) SRW_HIP(hipMemcpyAsync(out_w, d_w.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (index) SRW_HIP(hipMemcpyAsync(index, d_idx.p, 8, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
}

void hook_rng(srw_handle *h, uint32_t seed, const uint32_t *iter, const uint32_t *src, const uint32_t *step,
              int64_t n, float *out) {
  if (n <= 0) return;
  hipStream_t st = h->stream;
  DevBuf<uint32_t> a, b, c; DevBuf<float> o;
  a.alloc((size_t)n); b.alloc((size_t)n); c.alloc((size_t)n); o.alloc((size_t)n);
  SRW_HIP(hipMemcpyAsync(a.p, iter, (size_t)n * 4, hipMemcpyHostToDevice, st));
  SRW_HIP(hipMemcpyAsync(b.p, src, (size_t)n * 4, hipMemcpyHostToDevice, st));
  SRW_HIP(hipMemcpyAsync(c.p, step, (size_t)n * 4, hipMemcpyHostToDevice, st));
  int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_hook_rng, dim3(blocks), dim3(256), 0, st, seed, a.p, b.p, c.p, n, o.p);
  SRW_HIP(hipGetLastError());
  SRW_HIP(hipMemcpyAsync(out, o.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
}

}  // namespace srw
