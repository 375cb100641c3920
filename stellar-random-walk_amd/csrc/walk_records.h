// walk_records.h — records and argument blocks shared by the walk kernels' translation units (walk_kernels.hip, walk_groups.hip).
#pragma once
#include "engine.h"
#include "sampling.h"

namespace srw {

// Records shared by the whole-graph kernels and the vertex-sharded ones (described with the k_sh_* kernels in walk_kernels.hip).
struct alignas(16) WWalker { int32_t lw, src, prev, curr; };      // on the wire: 16 bytes
struct alignas(16) SWalker { int32_t lw, src, prev, curr, v, kind, pad0, pad1; };   // pad0: position of the chosen candidate (chain kernels)
enum : int32_t { SK_WALKER_RET = 1, SK_RET = 2, SK_DEAD = 3 };   // walker + return; return only (last step); death notice only
constexpr int CHAIN_CAP = 4096;     // draws on a CDF boundary per super-step / per launch that the chain kernels take (more: the general step)
struct alignas(16) ChainRec { uint32_t ri, pad; double S; };      // record index, (whole-graph walks: the step), the reference's sum of the biased row
// Whole-graph walks: where the table kernels leave a table step whose draw sits on a CDF boundary — one wire record per tie (lw = the
// iteration's offset) behind a chunk header, so that the chain kernels of the sharded walk read them like a super-step's input;
// k_walk_general, which redoes the handed-over walkers, takes the resolved step from the chain kernels' output.
struct TieSink {
  uint32_t *hdr;                  // [0] records written (the chunk header the chain kernels read)
  WWalker *recs;                  // [CHAIN_CAP]
  ChainRec *list;                 // [CHAIN_CAP]
  unsigned long long *cur;        // [2] ties met
  int32_t *todo_tie;              // per todo entry: its record, or -1
};

constexpr int TPB = 256;

// The table kernels take their arguments as ONE struct and read them again from the kernarg segment where a walker / a step
// needs them (device_common.h:fresh_args) instead of holding their ~130 dwords in SGPRs next to the walker's state.
struct TabArgs {
  GraphView g;                     // (first: fresh_graph() reads the same bytes)
  const int32_t *verts; int64_t n_verts, n_walkers; int32_t L, first_walk; RngSpec rng; float p, q;
  int32_t *paths, *lens; DevCounters *ctr; unsigned long long *cursor; int32_t *todo; unsigned long long *todo_n; TieSink tie;
};

// walk_groups.hip: the table walk with one walker per 16 lanes (four per wave)
void launch_walk_tables_groups(const TabArgs &ta, bool row_filters, int n_cus, hipStream_t st);
// walk_lanes.hip: ... with one walker per lane; mode bit 0: mask rows per lane, bit 1: table steps per lane (0: every step served by the wave)
void launch_walk_tables_lanes(const TabArgs &ta, bool row_filters, int mode, int max_csh, int n_cus, hipStream_t st);

// walk_rounds.hip: ... in rounds of two kernels (lanes advance, the wave serves one parked step per walker)
void launch_walk_tables_rounds(srw_handle *h, const TabArgs &ta, bool row_filters, int max_csh, hipStream_t st);

}  // namespace srw
