// engine.h — host-side state behind srw_handle and the internal interfaces between translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <stdexcept>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>
#include <algorithm>

#include "../../include/stellar_rw.h"
#include "device_common.h"

namespace srw {

struct Error : std::runtime_error {
  int32_t code;
  Error(int32_t c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define SRW_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      throw ::srw::Error(SRW_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
  } while (0)

// RAII device buffer.
template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
  }
  void alloc(size_t count) {
    release();
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e != hipSuccess)
      throw Error(e == hipErrorOutOfMemory ? SRW_ERR_NOMEM : SRW_ERR_HIP,
                  std::string("hipMalloc(") + std::to_string(count * sizeof(T)) + " B): " + hipGetErrorString(e));
    n = count;
  }
  void ensure(size_t count) { if (count > n) alloc(count); }
};

}  // namespace srw
#include "vm_buf.h"
namespace srw {

struct Graph {
  bool loaded = false;
  int32_t vmin = 0, vmax = -1;
  int64_t n_slots = 0;
  int64_t n_lines = 0;
  bool symmetric = false;         // built from an undirected edge list
  bool has_member = false;        // sids / sperm built (lazily, build_membership)
  int64_t n_entries_global = 0;   // "edges: N" of the whole graph
  int64_t n_entries = 0;          // entries stored on this handle (== global when world == 1)
  int64_t n_vertices = 0;         // present vertices of the whole graph
  int64_t n_local_vertices = 0;   // present vertices owned by this handle
  DevBuf<Row> rows;               // [n_slots]
  DevBuf<Ent> ent;                // [n_entries]
  DevBuf<uint32_t> sids;          // [n_entries] (id - vmin), sorted inside each row
  DevBuf<Row> mrows;              // sharded mode only: row table of the whole graph for the membership test
  DevBuf<uint32_t> msids;         // sharded mode only: sorted ids of the whole graph
  DevBuf<int32_t> owner_tab;      // sharded + SRW_CFG_OWNER_FROM_PARTITIONS: partition id per slot
  DevBuf<uint32_t> sperm;         // [n_entries] input-order position (inside the row) of each sorted entry
  DevBuf<float> sw;               // [n_entries] weight of each sorted entry (= ent[off + sperm].w)
  DevBuf<FoEnt> fo;               // [n_entries], built lazily
  bool has_fo = false;
  DevBuf<CfoEnt> cfo;             // [n_entries] compact lattice records (optional)
  bool has_cfo = false;
  bool cfo_rejected = false;      // the compact table was tried and some entry needed an escape
  DevBuf<Row> rows_all;           // vertex-sharded walk: every vertex's row descriptor as its OWNER stores it (srw_shard_rows_*)
  bool cfo_linked = false;        // cfo holds records whose links point into the owners' tables (k_sh_step_cfo only)
  DevBuf<AEnt> al;                // [n_entries] Mode A alias records, built lazily
  DevBuf<double> pq;              // [n_entries] per-(p,q) exact base prefix sums (general kernel fast path)
  DevBuf<uint8_t> pq_ok;          // [n_slots]
  bool has_pq = false; uint32_t pq_pbits = 0, pq_qbits = 0;
  DevBuf<uint32_t> hub_bm;        // [n_hubs][hub_words] neighbor-set bitmaps of the highest-degree rows over the id slots
  int64_t hub_words = 0; int64_t n_hubs = 0; int32_t hub_min_deg = 0; size_t hub_budget_cap = 0; bool has_hub = false, use_hub = false;
  DevBuf<uint64_t> ehash;         // Mode A: edge hash set (optional, built lazily when q != 1)
  uint64_t ehash_mask = 0; bool has_ehash = false; bool use_ehash = false;   // built / used by the current call
  DevBuf<uint32_t> eb_off;        // [n_entries] per-edge bias tables (edge_tables.hip): offset of entry e's table, 64-B units
  VmBuf<double> eb_bins;          // the tables (their pages are mapped while the build fills them: vm_buf.h)
  DevBuf<uint32_t> em_bits;       // membership masks of the pairs whose curr row has 33 .. eb_mask_max candidates
  int32_t eb_mask_max = 0, eb_f32 = 0, eb_cap = 64;
  DevBuf<RevEnt> rev;             // [n_entries] the return edge(s) of every entry (k_walk_q1), built lazily
  bool has_rev = false;
  int64_t pq_bad_rows = 0;        // rows the per-call certificate turned away (build_pq_tables)
  bool eb_complete = false;       // the HBM budget did not bind: every pair into a certified row has a table
  bool eb_no_ehash = false;       // the standing tables were built with the edge hash traded for their resolution (prepare_tables)
  size_t eb_reserve = (size_t)24 << 30;   // what build_edge_tables leaves free for the walk's own buffers (prepare_tables: this call's paths + 8 GB)
  size_t eb_budget_gb = 160;      // build_edge_tables' ceiling for this call (SRW_EB_BUDGET_GB overrides)
  bool has_eb = false, use_eb = false; uint32_t eb_pbits = 0, eb_qbits = 0; int32_t eb_min_sh = 8, eb_mode = 0;
  int64_t eb_tables = 0, eb_bytes = 0; double eb_build_ms = 0.0;
  DevBuf<double> rsum;            // [n_slots] Mode A: exact row weight sums
  bool has_al = false;
  DevBuf<int32_t> verts;          // owned present vertices, ascending
  DevBuf<int32_t> vrank;          // global rank (among all present vertices) of each entry of verts
  DevBuf<PairSlot> ph;            // sharded per-edge tables: pair (prev, curr) -> table / mask word (device_common.h)
  uint32_t ph_buckets = 0;
  bool eb_sharded = false;        // the standing per-edge tables are keyed by the pair hash (built by build_shard_edge_tables)
  DevBuf<PairSlot> rh;            // sharded q == 1 walks: return edges of the pairs into this shard's rows (build_shard_rev_hash)
  uint32_t rh_buckets = 0; bool has_rh = false;
  EbPolicy ebp = {8, 64, 0, 1024, 1024, 6, 0, 1, 0, 0};       // geometry of the standing per-edge tables (GraphView::ebp)
  int32_t eb_cm_sel = 0;          // what the next table build uses (prepare_tables / prepare_shard_tables choose them): chunk masks up to this row length,
  int32_t eb_cm_ratio_sel = 0;    // ... chunk masks also where deg(curr) <= ratio x deg(prev) (0: only where N(prev) is too long for the LDS staging),
  int32_t eb_fine_cap_sel = 0;    // ... chunks per table at most for the pairs with a long N(prev) and no mask (0: no finer tables)
  int32_t eb_min_sh_sel = 8;      // log2 of the smallest table chunk the next table build uses (prepare_tables / prepare_shard_tables choose it)
  int32_t dbg_chain_deg = 0;      // SRW_DEBUG_CHAIN_DEG (read by run_shard_superstep)
  bool has_cfo_local = false;     // sharded: compact first-order records over the LOCAL rows (guide + ids; their links are not used)
  DevBuf<int32_t> ids32;          // unit-weight graphs: the neighbor ids alone, input order (GraphView::ids32); unit_w: -1 not checked yet
  bool has_ids32 = false; int unit_w = -1;
  double pq_unit = 0.0;           // unit-weight graphs: fl(1 / q) of the call — the prefix sums are (k + 1) * pq_unit, no pq array (GraphView::pq_unit)
  DevBuf<uint32_t> bf_off, bf_bits; // neighbor-set filters of the rows beyond 1024 neighbors (GraphView::bf_off), built with the per-edge tables
  bool has_bf = false;
  // Compacted ids (sparse id spaces, SRW_CFG_COMPACT_IDS): slots are ranks among the sorted distinct input ids
  bool compact = false;
  DevBuf<int32_t> orig_id;        // [n_slots] rank -> input id (ascending)
  std::vector<int32_t> h_orig_id; // host mirror (boundary look-ups: srw_graph_neighbors, partitions)
  int32_t id_lo = 0, id_hi = -1;  // smallest / largest id a path can print (text capacity of the formatter)
  std::vector<int32_t> part_of;   // VCut: last pId recorded per dst slot, -1 none (host side; empty if unused)
  GraphView view() const { return GraphView{rows.p, ent.p, sids.p, sperm.p, has_fo ? fo.p : nullptr, (has_cfo || cfo_linked || has_cfo_local) ? cfo.p : nullptr, has_al ? al.p : nullptr, has_al ? rsum.p : nullptr,
                     mrows.p ? mrows.p : rows.p, msids.p ? msids.p : sids.p, (has_pq && pq_unit == 0.0) ? pq.p : nullptr, has_pq ? pq_ok.p : nullptr, (has_ehash && use_ehash) ? ehash.p : nullptr, ehash_mask,
                     symmetric ? 1 : 0, owner_tab.p, vmin, n_slots, sw.p,
                     (has_hub && use_hub) ? hub_bm.p : nullptr, hub_words,
                     (has_eb && use_eb && !eb_sharded) ? eb_off.p : nullptr, eb_bins.p, eb_min_sh, em_bits.p, eb_mask_max, eb_f32,
                     has_rev ? rev.p : nullptr, eb_cap, compact ? orig_id.p : nullptr,
                     (has_bf && use_eb && !(has_ehash && use_ehash)) ? bf_off.p : nullptr, bf_bits.p,
                     (has_eb && use_eb && eb_sharded) ? ph.p : nullptr, ph_buckets, has_rh ? rh.p : nullptr, rh_buckets, ebp, has_ids32 ? ids32.p : nullptr, (has_pq && pq_unit != 0.0) ? pq_unit : 0.0, dbg_chain_deg}; }
  // id <-> slot at the boundary (api.cpp): -1 if the id cannot be a vertex of this graph
  int64_t slot_of_id(int32_t v) const {
    if (!compact) { const int64_t s = (int64_t)v - vmin; return (s < 0 || s >= n_slots) ? -1 : s; }
    auto it = std::lower_bound(h_orig_id.begin(), h_orig_id.end(), v);
    return (it == h_orig_id.end() || *it != v) ? -1 : (int64_t)(it - h_orig_id.begin());
  }
  int32_t id_of_slot(int64_t s) const { return compact ? h_orig_id[(size_t)s] : (int32_t)(s + vmin); }
};

struct WalkResult {
  DevBuf<int32_t> paths, lens;
  int64_t n_walkers = 0;
  int32_t stride = 0;
  bool valid = false;
};

struct DevCounters {  // device-side accumulators, one 64-bit word each
  unsigned long long steps, dead_ends, sum_deg_curr, sum_deg_prev, ent_reads, fallbacks, owned_entries, trials;
  unsigned long long strat[12];   // general kernel, steps per sampler: SRW_STRAT_* (include/stellar_rw.h)
  unsigned long long why[4];      // hand-overs of k_walk_q1 by cause: irregular row, non-positive sum, boundary draw (SRW_DEBUG_HANDOVER)
#ifdef SRW_PHASE_TIMING
  unsigned long long dbg[40];   // wave-time per phase of the general kernel, 100 MHz ticks >> 10 (tools/phase_timing.py)
#endif
};

}  // namespace srw

struct srw_handle {
  srw_config cfg{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string last_error;
  srw::Graph g;
  srw::WalkResult res;
  srw::DevBuf<srw::DevCounters> counters;
  srw::DevBuf<unsigned long long> walk_cursor;   // [0] next walker of the persistent kernels, [1] walkers handed over by k_walk_tables
  srw::DevBuf<int32_t> walk_todo;                // their indices
  srw::DevBuf<char> round_state;                 // the table walk in rounds (walk_rounds.hip): 80 B of parked state per walker ...
  srw::DevBuf<int32_t> round_list;               // ... the two work lists (walkers a lane advances | walkers whose next step the wave serves) ...
  srw::DevBuf<unsigned long long> round_ctr;     // ... their lengths and the kernels' cursors
  int n_cus = 256;
  int64_t planned_walks = 0;                     // srw_plan_walks: the job's numWalks (0: unknown -> the reference's default 10)
  double shard_prof_acc[4] = {0, 0, 0, 0}, shard_prof_mx[4] = {0, 0, 0, 0};   // SRW_SHARD_PROFILE: per-kernel times of the super-steps (run_shard_superstep)
  int q1_occ[2] = {0, 0};           // resident blocks per CU of k_sh_step_q1<false / true> (queried once per handle)
  int dev_share = 1;                             // handles of one cluster on this device (virtual shards of a single-GPU box): optional structures take 1 / dev_share of what is free
  srw::DevBuf<char> shard_scratch;               // sampled 32-byte records before bucketing (persistent)
  srw::DevBuf<uint32_t> shard_blk;               // [blocks][2 * world] per-block survivor / return counts, then write cursors
  srw::DevBuf<uint32_t> shard_flag;              // chunk overflow flag of the sharded walk
  srw::DevBuf<uint32_t> shard_cur;               // fused first-order step: device-wide chunk cursors + finished-block counter
  srw::DevBuf<int32_t> shard_pt;                 // home rank's paths of the current batch, slot-major [L + 2][rows] (k_sh_apply -> k_sh_transpose)
  srw::DevBuf<char> chain_buf;                   // sharded table steps whose draw sits on a CDF boundary: record list, meta, totals ...
  srw::DevBuf<double> chain_d;                   // ... and the quotients w'_k / S of their rows (k_chain_*)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // A second walker population on this handle (srw_shard_select): its own super-step context — scratch, cursors, counters, path
  // staging, chain buffers — and its own stream, so that one population's kernels run while the other's chunks are on the wire
  // (the serial shuffle / count rhythm of RandomWalk.scala:91-162 has no such overlap).  The parked context lives here; selecting
  // swaps it with the members above.
  struct ShardCtx {
    srw::DevBuf<srw::DevCounters> counters; srw::DevBuf<unsigned long long> walk_cursor; srw::DevBuf<int32_t> walk_todo;
    srw::DevBuf<char> shard_scratch; srw::DevBuf<uint32_t> shard_blk, shard_flag, shard_cur; srw::DevBuf<int32_t> shard_pt;
    srw::DevBuf<char> chain_buf; srw::DevBuf<double> chain_d;
    hipStream_t stream = nullptr; bool own_stream = false; hipEvent_t ev0 = nullptr, ev1 = nullptr; bool init = false;
  } shard_parked;
  int shard_population = 0;                      // which population's context is in the members above (0 / 1)
  // srw_walk_to_host: second stream + two staging buffers for compute/copy overlap
  hipStream_t copy_stream = nullptr;
  srw::DevBuf<int32_t> stage_paths[2], stage_lens[2];
  hipEvent_t stage_done[2] = {nullptr, nullptr}, kernel_done[2] = {nullptr, nullptr};
  int32_t *pin_paths[2] = {nullptr, nullptr}, *pin_lens[2] = {nullptr, nullptr};  // pinned ring of srw_walk_and_save
  size_t pin_cap = 0, pin_lens_cap = 0;
  // device-side formatter (SRW_WALK_DEVICE_FORMAT): per staging slot the text + line offsets, one pinned text buffer
  srw::DevBuf<char> fmt_text[2]; srw::DevBuf<unsigned long long> fmt_len[2], fmt_off[2]; srw::DevBuf<char> fmt_temp;
  static constexpr int PIN_RING = 6;             // pinned text slices in flight between the device and the writer's threads
  char *pin_text[PIN_RING] = {nullptr}; unsigned long long *pin_off[2] = {nullptr, nullptr};
  size_t pin_text_cap[PIN_RING] = {0}, pin_off_cap = 0;
  hipEvent_t pin_copied[PIN_RING] = {nullptr};
};

namespace srw {

// ---- api.cpp ----
void set_create_error(const std::string &m);     // message behind srw_last_error(NULL)

// ---- edgelist.cpp (host) ----
// Uninitialised, non-copyable array: the tokenizer's threads fill it in place (a std::vector would zero 268 MB first
// at config 2, on one thread).
template <class T>
struct RawVec {
  T *p = nullptr; size_t n = 0;
  RawVec() = default;
  RawVec(const RawVec &) = delete;
  RawVec &operator=(const RawVec &) = delete;
  RawVec(RawVec &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  RawVec &operator=(RawVec &&o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
  ~RawVec() { free(p); }
  void resize_uninit(size_t k) {
    free(p); p = nullptr; n = 0;
    if (k) { p = (T *)malloc(k * sizeof(T)); if (!p) throw std::bad_alloc(); n = k; }
  }
  T *data() { return p; }
  const T *data() const { return p; }
  size_t size() const { return n; }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
};
struct ParsedLines {
  RawVec<int32_t> src, dst, pid;
  RawVec<float> w;
};
void parse_edgelist_file(const char *path, bool weighted, bool partitioned, ParsedLines &out);

// ---- writer.cpp (host) ----
void write_path_files(const int32_t *paths, const int32_t *lens, int64_t n_walkers, int64_t stride,
                      const char *output_dir, int n_parts, bool write_crc);
// Incremental form: consecutive slices of the canonical walker order are appended as they arrive from the GPU.
class PathWriter {
 public:
  PathWriter(const char *output_dir, int n_parts, int64_t total_walkers, bool write_crc);  // throws SRW_ERR_EXISTS
  ~PathWriter();
  void append(const int32_t *paths, const int32_t *lens, int64_t n, int64_t stride);
  // already formatted lines (device formatter): off[0..n] are byte offsets, text[0] is the byte at offset `base`
  // token >= 0 (and no .crc files asked): the bytes are handed to a pool of writer threads — part files are written in parallel — and
  // `text` must stay untouched until wait_token(token) returns (token: the caller's name for its buffer, < 16).  token < 0: written on return.
  void append_text(const char *text, const unsigned long long *off, int64_t n, unsigned long long base = 0, int token = -1);
  void wait_token(int token);      // token < 0: everything handed over so far
  bool token_idle(int token);      // nothing handed over with this token is still being written
  void close();  // finishes the parts, writes _SUCCESS
 private:
  struct Impl;
  Impl *p_;
};

// ---- edgelist_device.hip ----
// Device-side tokenizer for two-column integer edge lists (+ a short decimal weight column); false = not that shape (or any doubt): use the host tokenizer.
bool load_edgelist_device(srw_handle *h, const char *path, bool directed, bool weighted);

// ---- path_format.hip ----
size_t format_capacity(int64_t n, int64_t stride, int32_t vmin, int32_t vmax);
void ensure_pinned_text(srw_handle *h, size_t slice_cap, size_t n_off);
size_t text_slice_cap(int64_t stride, size_t text_bytes);
// text [off[0], off[n]) of d_text (whole lines, offsets on the host) -> the writer, through the ring of pinned slices; all_copied():
// every byte has left d_text (it may be overwritten)
void drain_text(srw_handle *h, PathWriter &writer, const char *d_text, const unsigned long long *off, int64_t n, size_t slice_cap,
                const std::function<void()> &all_copied);
bool write_result_device(srw_handle *h, const char *output_dir, int n_parts, bool write_crc);
void format_paths_device(srw_handle *h, const int32_t *d_paths, const int32_t *d_lens, int64_t n, int64_t stride,
                         unsigned long long *d_len_bytes, unsigned long long *d_off, char *d_text);

// ---- graph_build.hip ----
void check_id_range(int32_t vmin, int32_t vmax);   // throws SRW_ERR_NOMEM when the dense slot tables cannot hold [vmin, vmax]
// Lines already on the device (d_src/d_dst/d_w; d_w may be null = 1.0f).  Builds rows/ent/sids/verts.
// Compacted ids: decision + the rewrite of the lines (d_src / d_dst become ranks, [vmin, vmax] becomes [0, n_ids - 1]);
// the IdMap travels into the Graph through build_graph_from_device_lines.
struct IdMap {
  bool compact = false;
  DevBuf<int32_t> orig_id;
  std::vector<int32_t> h_orig_id;
  int32_t id_lo = 0, id_hi = -1;
};
bool ids_are_sparse(const srw_handle *h, int64_t n_ids, int32_t vmin, int32_t vmax);
void compact_ids(srw_handle *h, int32_t *d_src, int32_t *d_dst, int64_t n_lines, int32_t &vmin, int32_t &vmax, IdMap &m);
void build_graph_from_device_lines(srw_handle *h, const int32_t *d_src, const int32_t *d_dst, const float *d_w,
                                   int64_t n_lines, bool directed, int32_t vmin, int32_t vmax,
                                   const int32_t *host_owner_tab = nullptr, IdMap *idmap = nullptr);
// Vertex-sharded handles: the lines block by block (a slice of device arrays, a generated block, an uploaded block)
using LineFetch = std::function<void(int64_t i0, int64_t n, const int32_t *&src, const int32_t *&dst, const float *&w)>;
void build_graph_blocked(srw_handle *h, const LineFetch &fetch, int64_t n_lines, bool directed, int32_t vmin, int32_t vmax,
                         const int32_t *host_owner_tab = nullptr, IdMap *idmap = nullptr);
void generate_rmat_block(srw_handle *h, int32_t scale, int64_t first_edge, int64_t n_edges, uint32_t seed, bool weighted,
                         int32_t *d_src, int32_t *d_dst, float *d_w);
void build_graph_from_host_rows(srw_handle *h, const int32_t *vids, const int64_t *offs, int64_t n_rows,
                                const int32_t *ids, const float *w);
void generate_rmat_lines(srw_handle *h, int32_t scale, int64_t n_edges, uint32_t seed, bool weighted,
                         DevBuf<int32_t> &d_src, DevBuf<int32_t> &d_dst, DevBuf<float> &d_w);
void build_membership(srw_handle *h);
bool graph_has_unit_weights(srw_handle *h);     // graph_build.hip: every w == 1.0f (checked once per graph)
void build_unit_ids(srw_handle *h);               // graph_build.hip: ids32 of a unit-weight graph (a no-op otherwise or when HBM is short)
void build_row_filters(srw_handle *h);            // word-blocked Bloom filters of the long rows (GraphView::bf_off)
void build_first_order_tables(srw_handle *h, bool want_exact);
void build_pq_tables(srw_handle *h, float p, float q);

// ---- alias_tables.hip ----
void build_alias_tables(srw_handle *h);
void build_edge_hash(srw_handle *h);
void build_hub_bitmaps(srw_handle *h, int32_t min_deg, size_t budget_cap);   // graph_build.hip

// ---- edge_tables.hip ----
// Per-edge bias tables for the general kernel under (p, q): mode 0 = automatic (most expensive pairs first, within the
// HBM budget), 1 = every certified pair (tests: tiny chunks, no cost threshold).  Needs build_pq_tables first.
void build_edge_tables(srw_handle *h, float p, float q, int mode, int bins_cap);
size_t edge_tables_full_bytes(srw_handle *h, int mode, int bins_cap);   // HBM of a complete set of per-edge tables (0: none possible)
// Vertex-sharded handles (and whole-graph handles walked through srw_shard_*): the tables of the pairs (prev -> curr) whose
// curr this handle owns, keyed by the pair hash; a complete set at the finest resolution that fits, or none.
void prepare_shard_tables(srw_handle *h, const srw_walk_params &P);
void build_shard_rev_hash(srw_handle *h);        // return edges of the pairs into this shard's rows, keyed by (prev, curr)
bool build_local_cfo(srw_handle *h);             // sampler_tables.hip: compact records over the local rows (false: some entry needs an escape)
void build_rev_table(srw_handle *h);             // return-edge positions (k_walk_q1: p != 1, q == 1)

// ---- walk_kernels.hip ----
void run_walk(srw_handle *h, const srw_walk_params &P, srw_walk_stats *stats);
void run_walk_to_host(srw_handle *h, const srw_walk_params &P, int32_t *paths, int32_t *lens, srw_walk_stats *stats);
void run_walk_and_save(srw_handle *h, const srw_walk_params &P, const char *output_dir, int n_parts, bool write_crc,
                       srw_walk_stats *stats, int64_t *dead_per_iter);
void shard_layout(const srw_handle *h, int32_t batch, double slack, srw_shard_layout *out);
void run_shard_begin(srw_handle *h, const srw_walk_params &P, int32_t batch, const srw_shard_layout &lay, void *d_recv,
                     int32_t *d_paths, int32_t *d_lens, int64_t stride);
void shard_rows_export(srw_handle *h, void *d_rows, int64_t n_slots);
void shard_rows_merge(srw_handle *h, void *d_rows, const void *d_other, int64_t n_slots);
bool shard_rows_commit(srw_handle *h, const void *d_rows_all, int64_t n_slots);
void shard_rows_release(srw_handle *h);
void run_shard_superstep(srw_handle *h, const srw_walk_params &P, int32_t batch, int32_t step, const srw_shard_layout &lay,
                         const void *d_recv, void *const *dst, int32_t *d_paths, int32_t *d_lens, int64_t stride);
void run_shard_flush(srw_handle *h, const srw_walk_params &P, int32_t batch, const srw_shard_layout &lay, const void *d_recv,
                     int32_t *d_paths, int32_t *d_lens, int64_t stride);
void run_shard_finish(srw_handle *h, srw_walk_stats *stats, int32_t *overflow);
// embedding.hip: skip-gram + hierarchical softmax over host paths
void w2v_fit(srw_handle *h, const int32_t *paths, const int32_t *lens, int64_t n, int64_t stride, const srw_w2v_params &P,
             std::vector<int32_t> &vocab_ids, std::vector<float> &vectors);
void w2v_fit_device(srw_handle *h, const int32_t *d_paths, const int32_t *d_lens, int64_t n, int64_t stride, const srw_w2v_params &P,
                    std::vector<int32_t> &vocab_ids, std::vector<float> &vectors);
void w2v_huffman(const int64_t *counts, int64_t n_vocab, int32_t *code_len, uint8_t *codes, int32_t *points);   // host only (test hook)
// writer.cpp: <output>/vec part files + <output>/bin
void write_vectors(const int32_t *vocab_ids, const float *vectors, int64_t n_vocab, int32_t dim, const char *output_dir, int n_parts);
void write_word2vec_parquet(const std::string &path, const std::function<void(std::string &, int64_t)> &put_name, const float *vectors,
                            int64_t n, int32_t dim);   // parquet_model.cpp: <output>/bin/data as Word2VecModel.save writes it
void write_vectors_words(const char *const *words, const float *vectors, int64_t n_vocab, int32_t dim, const char *output_dir, int n_parts);
std::string java_float_to_string(float x);
// probe.hip: measurement hooks of bench.py's roofline object
void probe_request_rate(srw_handle *h, int64_t table_bytes, double *reads_per_s, double *table_gib);
void result_scan_sums(srw_handle *h, int64_t *out3);
void hook_sample(srw_handle *h, const float *w, int64_t n, float r, int64_t *index);
void hook_second_order(srw_handle *h, float p, float q, int32_t prev_id, const int32_t *prev_ids, int64_t n_prev,
                       const int32_t *curr_ids, const float *curr_w, int64_t n, float r, float *out_w,
                       int64_t *index);
void hook_rng(srw_handle *h, uint32_t seed, const uint32_t *iter, const uint32_t *src, const uint32_t *step,
              int64_t n, float *out);

}  // namespace srw
