// device_common.h — HBM data layout + device primitives shared by the gfx950 kernels.
//
// Layout:
//   rows[slot]  16 B  {off:int64, deg:int32, flags:u32}   slot = vertex id - vmin, dense
//   ent[e]       8 B  {id:int32, w:f32}                   adjacency in input-line order (CSR payload)
//   sids[e]      4 B  u32 (id - vmin) sorted inside each row   (membership test of computeSecondOrderWeights)
//   sperm[e]     4 B  u32 input-order position inside the row of sorted entry e (reverse membership marking)
//   fo[e]       32 B  {cdf:f64, id:int32, guide:int32, noff:int64, ndeg:int32, nflags:u32}
//                     first-order exact CDF + guide table + the neighbor's row descriptor (p = q = 1)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace srw {

struct alignas(16) Row {
  int64_t off;
  int32_t deg;
  uint32_t flags;
};
enum : uint32_t { ROW_PRESENT = 1u, ROW_IRREGULAR = 2u, ROW_ALIAS_IRREGULAR = 4u,
                  ROW_PQ_OK = 8u,
                  ROW_PQ_F32 = 16u };  // (with ROW_PQ_OK) every sum the samplers form over this row is a multiple of 2^G below 2^(24+G):
                                       // exactly representable in binary32 -> its per-edge tables are stored as floats
// bits 5 .. 7 (with ROW_PQ_OK): code c > 0: a chunk of up to 2^(c + 5) candidates of this row weighs less than 65 536 units of 2^G under the
// call's (p, q) — the chunk prefixes of such tables are stored as 16-bit deltas (edge_tables.hip); bits 8 .. 15: G + 128, the row's unit exponent
// (every variant w, fl(w/p), fl(w/q) of every candidate is a multiple of 2^G); both set by k_pq_*
constexpr int ROW_U16_SHIFT = 5;
constexpr int ROW_G_SHIFT = 8;
constexpr uint32_t ROW_PQ_BITS = ROW_PQ_OK | ROW_PQ_F32 | (7u << ROW_U16_SHIFT) | (0xFFu << ROW_G_SHIFT);
constexpr int ROW_HUB_SHIFT = 16;  // Row::flags >> 16 = 1 + ordinal of the row's neighbor-set bitmap (0: none)

struct alignas(8) Ent {
  int32_t id;
  float w;
};

struct alignas(32) FoEnt {
  double cdf;    // acc after adding this entry, computed exactly as RandomSample.sample does (:18-20)
  int32_t id;
  int32_t guide; // first k with cdf_k >= ceil(j * 2^24 / deg) * 2^-24, for bucket j = this position
  // row descriptor of the neighbor `id` ("linked" record): the walker that picks this entry already holds
  // the next vertex's row, so a step costs ONE random 32-byte access instead of a row load + a record load
  int64_t noff;
  int32_t ndeg;
  uint32_t nflags;
};

// Compact first-order record for LATTICE draws (p = m * 2^-24, i.e. the Philox stream): cdf >= p  <=>
// floor(cdf * 2^24) >= m, and since m <= 2^24 - 1, min(floor(cdf * 2^24), 2^24 - 1) decides exactly like the f64
// compare: 24 bits.  One 16-byte load.
//   cg   : bits 0..23  c = min(floor(cdf * 2^24), 2^24 - 1)     bits 24..31  low 8 bits of the guide delta
//   id   : neighbor id
//   link : bits 0..35 noff | bits 36..39 high 4 bits of the guide delta | bits 40..62 ndeg (< 2^23 - 1) |
//          bit 63 next row needs the generic path
// guide delta gd = j - guide[j] as a signed 12-bit number (-2047 .. 2047; weighted hub rows need both signs and
// hundreds); CFO_GD_SAT marks an entry whose delta does not fit: the pick then searches the row's (monotone) c values.
struct alignas(16) CfoEnt {
  uint32_t cg;
  int32_t id;
  uint64_t link;
};
constexpr uint32_t CFO_NDEG_MAX = (1u << 23) - 2u;
constexpr int32_t CFO_GD_SAT = -2048;
constexpr uint64_t CFO_NOFF_MASK = (1ull << 36) - 1ull;
__host__ __device__ inline int32_t cfo_delta(uint32_t cg, uint64_t link) {
  const uint32_t d = (cg >> 24) | ((uint32_t)((link >> 36) & 0xFull) << 8);
  return (int32_t)(d << 20) >> 20;                     // sign-extend 12 bits
}

// Mode A record: alias slot of entry k of a row + the neighbor's id / weight / row descriptor (linked).
struct alignas(32) AEnt {
  float prob;    // P(keep slot k | slot k drawn) = kept_k / T, exact-integer construction (alias_tables.hip)
  int32_t alias; // position (inside the row) used otherwise
  int32_t id;
  float wrev;    // total weight of the edge(s) id -> (this row's vertex) when exactly representable in f32 (it is the
                 // return-edge weight W_prev of the NEXT step's outlier folding); NaN = look it up at walk time
  int64_t noff;
  int32_t ndeg;
  uint32_t nflags;
};


struct alignas(16) RevEnt { uint32_t cl, pos0; float w0; uint32_t pad; };

struct PairSlot;
// geometry policy of the per-edge tables (sampling.h:eb_pair_geometry): shared by the planner, the builder and the walk
struct EbPolicy { int32_t min_sh, cap, cm_max, cm_min_du, fine_min_du, fine_sh, fine_cap, f32, u16, cm_ratio; };
struct GraphView {
  const Row *rows;
  const Ent *ent;
  const uint32_t *sids;
  const uint32_t *sperm;
  const FoEnt *fo;
  const CfoEnt *cfo;    // null unless the compact table is valid for the whole graph
  const AEnt *al;
  const double *rsum;   // Mode A: exact weight sum of each alias-regular row
  const Row *mrows;     // membership structure: rows/sids of the WHOLE graph (== rows/sids when world == 1;
  const uint32_t *msids;  //   replicated on every shard so that N(prev) is available wherever curr lives)
  const double *pq;     // per-call exact prefix sums of the base weights fl(w / q) inside each row (null if not built)
  const uint8_t *pq_ok; // per slot: row certified for the prefix-sum sampler under the call's (p, q)
  const uint64_t *ehash;  // Mode A: open-addressing set of directed edges, key = (row slot << 32) | (id - vmin); null if absent
  uint64_t ehash_mask;
  int32_t symmetric;    // 1: undirected load (x in N(y) <=> y in N(x)): membership may probe the shorter row
  const int32_t *owner_tab;  // sharded + SRW_CFG_OWNER_FROM_PARTITIONS: partition id per slot (-1 unknown), else null
  int32_t vmin;
  int64_t n_slots;
  const float *sw;      // weight of each sorted entry (same indexing as sids / sperm)
  const uint32_t *hub_bm;   // neighbor-set bitmaps of the hub rows (ordinal in Row::flags >> ROW_HUB_SHIFT, 0 = none)
  int64_t hub_words;        // 32-bit words per bitmap = ceil(n_slots / 32)
  // per-edge bias tables of the current call's (p, q) (edge_tables.hip): entry e = (prev -> curr) owns
  // eb_bins[eb_off[e] * 8 ..): exact chunk prefixes of the corrections of N(prev) ∩ N(curr) over curr's positions
  const uint32_t *eb_off;   // [n_entries] offset in 64-byte units, EB_NONE = no table for this edge; null if not built
  const double *eb_bins;
  int32_t eb_min_sh;        // smallest chunk shift of a table (8: chunks of >= 256 candidates; tests use 2)
  // rows of curr with at most eb_mask_max candidates: eb_off[e] is the pair's membership mask itself (<= 32 candidates)
  // or the offset, in 16-byte units, of its mask words in em_bits (bit k = candidate k of N(curr) is in N(prev))
  const uint32_t *em_bits;
  int32_t eb_mask_max;
  int32_t eb_f32;           // 1: tables of ROW_PQ_F32 rows hold floats (half the bytes)
  // rev[e], e = (u -> v): cl = (count << 24) | index in v's SORTED row of the first entry that leads back to u (the return
  // edges of the step u -> v -> ?; sperm / sw give their input-order positions and weights), REV_NONE if there is none
  // (directed graphs); pos0 / w0 = input-order position and weight of that first return edge, so that the common case of
  // ONE return edge costs one 16-byte request instead of three; null if not built
  const RevEnt *rev;
  int32_t eb_cap;           // chunks per table at most (32 … 256: finer tables on hub rows when HBM allows, a complete coarser set when it does not; edge_tables.hip)
  // Compacted id space (graph_build.hip:compact_ids): every id inside the engine is the RANK of the vertex among the sorted
  // distinct ids of the input (vmin = 0), orig_id[rank] is the id the files carry.  Null: ids are used as they are.
  const int32_t *orig_id;
  // Neighbor-set filters of the long rows (deg > 1024: too long for the LDS staging of binned_resolve): a word-blocked Bloom
  // filter per row, 16-32 bits per neighbor, every element's 3 bits inside ONE 32-bit word — "x in N(prev)?" is one 4-byte read
  // that the ~500 candidates of a located chunk keep in L2, and only a positive (members, ~1 % false positives) goes on to the
  // exact test (edge hash / sorted row).  bf_off[slot]: first word of the row's filter in bf_bits, BF_NONE = none.  Null if not built.
  const uint32_t *bf_off;
  const uint32_t *bf_bits;
  // Vertex-sharded handles: the per-edge tables of the pairs (prev -> curr) whose CURR this shard owns are found through a
  // hash of the pair (PairSlot below) instead of eb_off[entry of prev's row] — prev's row lives on another shard, and a
  // walker arrives with (prev, curr) only.  Null on whole-graph handles walked through srw_walk.
  const PairSlot *ph;
  uint32_t ph_buckets;
  // Vertex-sharded handles, p != 1 with q == 1 (k_sh_step_q1): the return edges of the pair (prev -> curr), curr owned — what
  // rev[e] holds on a whole-graph handle (val = count << 24 | index in curr's sorted row, pad = input-order position of the first one)
  const PairSlot *rh;
  uint32_t rh_buckets;
  EbPolicy ebp;            // geometry of the standing per-edge tables (chunk sizes, chunk masks, finer tables of the pairs with a long N(prev))
  // Unit-weight graphs (every w == 1.0f: unweighted inputs, config 5): the ids alone, in input order — the table steps then read 4 bytes per
  // candidate instead of the 8-byte (id, w) entry (graph_build.hip:build_unit_ids; null otherwise)
  const int32_t *ids32;
  // ... and no prefix-sum array: the exact prefix of the base weights fl(1 / q) of a unit-weight row is (k + 1) * pq_unit (0: read pq)
  double pq_unit;
  int32_t dbg_chain_deg;   // tests (SRW_DEBUG_CHAIN_DEG): sharded steps on rows at least this long are treated as draws on a CDF boundary (0: off)
};
// The exact prefix sums PQ[k] of a row's base weights fl(w / q): the per-call array (8 B per entry) — or, on a unit-weight graph, the
// closed form (k + 1) * fl(1 / q): k + 1 < 2^24 and a 24-bit constant, the product is exact and equals the sum of k + 1 copies.
struct PqRow {
  const double *p; double unit;
  __device__ PqRow(const GraphView &g, int64_t off) : p(g.pq ? g.pq + off : nullptr), unit(g.pq_unit) {}
  __device__ inline double operator[](int64_t k) const { return p ? p[k] : (double)(k + 1) * unit; }
};
// A kernel whose arguments are ONE struct A (the GraphView first in it) can read them again from the kernarg segment where it needs
// them: ~130 dwords of arguments held in SGPRs for the whole kernel leave the compiler no SGPRs for the walker's state (it goes
// through spill VGPRs: a v_readlane per use — profiles/r04_valu_issue.md).  fresh_args<A>() hands out a copy whose fields are
// loaded (s_load, the scalar cache) at the point of the call, and only those the code that follows uses: the asm makes the pointer
// opaque, so the loads are neither hoisted out of the walk loop nor kept live across it.
template <typename A>
__device__ inline A fresh_args() {
#if defined(__HIP_DEVICE_COMPILE__)
  uintptr_t ka = (uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  return *(const __attribute__((address_space(4))) A *)ka;
#else
  return A();                      // (the host pass of hipcc only parses device functions)
#endif
}
__device__ inline GraphView fresh_graph() { return fresh_args<GraphView>(); }
__device__ inline bool pq_ready(const GraphView &g) { return g.pq != nullptr || g.pq_unit != 0.0; }
constexpr uint32_t BF_NONE = 0xFFFFFFFFu;
constexpr int32_t BF_MIN_DEG = 1025;
// words of the filter of a row of `deg` neighbors (a power of two, 16-32 bits per neighbor)
__host__ __device__ inline uint32_t bf_words(int32_t deg) {
  const uint32_t half = (uint32_t)(deg - 1) >> 1;       // the smallest power of two above half (closed form: the walk asks per table step)
  return half ? 1u << (32 - __builtin_clz(half)) : 1u;
}
// word and 3-bit mask of id slot x in a filter of nw words
__host__ __device__ inline void bf_hash(uint32_t x, uint32_t nw, uint32_t &word, uint32_t &mask) {
  uint32_t h = x * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13; h *= 0xC2B2AE3Du; h ^= h >> 16;
  word = h & (nw - 1u);
  uint32_t k = (x ^ 0x7F4A7C15u) * 0x2545F491u; k ^= k >> 16; k *= 0x9E3779B1u; k ^= k >> 15;
  mask = (1u << (k & 31u)) | (1u << ((k >> 5) & 31u)) | (1u << ((k >> 10) & 31u));
}
// The id the walk's Philox stream is keyed with (ctr = (iteration, SOURCE ID, step, 0)) is the input's, not the rank.
__device__ inline int32_t rng_source(const GraphView &g, int32_t src) { return g.orig_id ? g.orig_id[src - g.vmin] : src; }
constexpr uint32_t REV_NONE = 0xFFFFFFFFu;
constexpr int REV_MAX_RETURNS = 4;          // return edges of one step the per-lane kernel keeps in registers
constexpr uint32_t EB_NONE = 0xFFFFFFFFu;

// Philox4x32-10 (Random123).  Same constants as oracle/srw_oracle.c:orc_philox4x32_10.
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 24 random bits of the walk stream: ctr = (walk iteration, source id, step index, 0), key = (seed, 0).
__host__ __device__ inline uint32_t walk_bits24(uint32_t seed, uint32_t iter, uint32_t src, uint32_t step) {
  uint32_t o[4];
  philox4x32_10(iter, src, step, 0u, seed, 0u, o);
  return o[0] >> 8;
}

struct RngSpec {
  int32_t mode;   // SRW_RNG_*
  float const_r;
  uint32_t seed;
};

__device__ inline float draw_uniform(const RngSpec &rng, uint32_t iter, uint32_t src, uint32_t step) {
  if (rng.mode == 0) return rng.const_r;
  return (float)walk_bits24(rng.seed, iter, src, step) * (1.0f / 16777216.0f);
}

// One walker per wave: the draws of 64 consecutive steps at once — lane l computes step base + l (the same keyed Philox value the
// per-step call gives), the step reads its lane.  10 Philox rounds are ~130 scalar instructions per step when every lane computes
// the same one; here they are ~130 vector instructions per 64 steps (the table kernels are bound by instruction issue,
// profiles/r04_valu_issue.md).
struct WaveDraws {
  float u = 0.0f;
  // step = first, first + 1, ... in order (the walk loop): a new batch every 64 steps
  __device__ inline float at(const RngSpec &rng, uint32_t iter, uint32_t src, int32_t step, int32_t first) {
    const int32_t i = (step - first) & 63;            // (wave-uniform)
    if (i == 0) u = draw_uniform(rng, iter, src, (uint32_t)(step + (int32_t)(threadIdx.x & 63u)));
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(u), i));
  }
};

// Which shard owns vertex v.  The reference partitions by HashPartitioner = nonNegativeMod(id.hashCode, n) with
// Int.hashCode = identity (RandomWalk.scala:16) — and inherits the skew of the ids: an RMAT graph without a vertex
// permutation puts 44 % of its edge endpoints on ids whose three low bits are 000, so `v mod 8` hands one shard 44 % of
// the walkers and another 1.4 % (measured: chunk overflow up to 16x the mean).  The partition is invisible in the output
// (paths are identical for any partitioning, tests), so the ids are mixed first; a VCut input's own partition ids
// (owner_of_tab) are honoured as given.
__host__ __device__ inline int32_t owner_of(int32_t v, int32_t world) {
  uint32_t h = (uint32_t)v * 0x9E3779B1u;
  h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
  return (int32_t)(h % (uint32_t)world);
}
// Edge-existence test through the hash set (linear probing, EMPTY = all ones).
// 32-bit mixing only (a 64-bit multiply costs ~8 VALU instructions on CDNA): murmur3's finalizer over the two halves of
// the key; tables above 2^32 slots take their high index bits from a second, independent mix.
__device__ inline uint64_t edge_hash(uint64_t k, uint64_t mask) {
  const uint32_t row = (uint32_t)(k >> 32), id = (uint32_t)k;
  uint32_t a = (row * 0x9E3779B1u) ^ id;
  a ^= a >> 16; a *= 0x85EBCA6Bu; a ^= a >> 13; a *= 0xC2B2AE35u; a ^= a >> 16;
  uint64_t h = a;
  if (mask >> 32) {
    uint32_t b = id * 0x27D4EB2Fu + row;
    b ^= b >> 15; b *= 0x2C1B3C6Du; b ^= b >> 12; b *= 0x297A2D39u; b ^= b >> 15;
    h |= (uint64_t)b << 32;
  }
  return h & mask;
}
__device__ inline bool edge_exists(const uint64_t *tab, uint64_t mask, uint32_t row_slot, uint32_t id_slot) {
  const uint64_t key = ((uint64_t)row_slot << 32) | id_slot;
  uint64_t s = edge_hash(key, mask);
  while (true) {
    const uint64_t v = tab[s];
    if (v == key) return true;
    if (v == 0xFFFFFFFFFFFFFFFFull) return false;
    s = (s + 1) & mask;
  }
}

// ---- pair hash of a shard's per-edge tables -----------------------------------------------------------------------
// One 16-byte slot per pair (u -> x), x owned by the shard: key = u slot << 32 | x slot, val = what eb_off[e] holds on a
// whole-graph handle (table offset in 64-byte units, mask offset in 16-byte units, or the inline mask of a row of <= 32
// candidates).  Buckets of four slots = one 64-byte line; linear probing over slots from the bucket's first one, so a
// key sits in the run of occupied slots that starts there.  A step's probe (8 slots = two lines read by eight lanes of
// the wave, issued together with the row of curr) almost always ends in the first line (load <= 0.55).
struct alignas(16) PairSlot { unsigned long long key; uint32_t val, pad; };
constexpr unsigned long long PAIR_EMPTY = 0xFFFFFFFFFFFFFFFFull;
__host__ __device__ inline uint32_t pair_bucket(uint32_t u, uint32_t x, uint32_t n_buckets) {
  uint32_t a = (u * 0x9E3779B1u) ^ (x * 0x85EBCA77u + 0x165667B1u);
  a ^= a >> 16; a *= 0x85EBCA6Bu; a ^= a >> 13; a *= 0xC2B2AE35u; a ^= a >> 16;
  return (uint32_t)(((uint64_t)a * (uint64_t)n_buckets) >> 32);
}
__device__ inline void pair_insert(PairSlot *tab, uint32_t n_buckets, uint32_t u, uint32_t x, uint32_t val, uint32_t pad = 0u) {
  const unsigned long long key = ((unsigned long long)u << 32) | x;
  const uint64_t cap = (uint64_t)n_buckets * 4;
  uint64_t s = (uint64_t)pair_bucket(u, x, n_buckets) * 4;
  while (true) {
    const unsigned long long old = atomicCAS(&tab[s].key, PAIR_EMPTY, key);
    if (old == PAIR_EMPTY || old == key) { tab[s].val = val; tab[s].pad = pad; return; }
    if (++s == cap) s = 0;
  }
}
// per-lane lookup (every lane its own pair): one 16-byte slot per probe
__device__ inline bool pair_lookup_lane(const PairSlot *tab, uint32_t n_buckets, uint32_t u, uint32_t x, uint32_t &val_out, uint32_t &pad_out) {
  const unsigned long long key = ((unsigned long long)u << 32) | x;
  const uint64_t cap = (uint64_t)n_buckets * 4;
  uint64_t s = (uint64_t)pair_bucket(u, x, n_buckets) * 4;
  while (true) {
    const uint4 q = *reinterpret_cast<const uint4 *>(tab + s);
    const unsigned long long k = ((unsigned long long)q.y << 32) | q.x;
    if (k == key) { val_out = q.z; pad_out = q.w; return true; }
    if (k == PAIR_EMPTY) return false;
    if (++s == cap) s = 0;
  }
}
// wave-uniform lookup: all 64 lanes call with the same (u, x); true = found (val_out wave-uniform)
__device__ inline bool pair_lookup_wave(const PairSlot *tab, uint32_t n_buckets, uint32_t u, uint32_t x, uint32_t &val_out) {
  const unsigned long long key = ((unsigned long long)u << 32) | x;
  const uint64_t cap = (uint64_t)n_buckets * 4;
  uint64_t s0 = (uint64_t)pair_bucket(u, x, n_buckets) * 4;
  const int lane = (int)(threadIdx.x & 63u);
  while (true) {
    unsigned long long k = 0ull; uint32_t v = 0u;
    if (lane < 8) {
      uint64_t s = s0 + (uint64_t)lane; if (s >= cap) s -= cap;
      const uint4 q = *reinterpret_cast<const uint4 *>(tab + s);
      k = ((unsigned long long)q.y << 32) | q.x; v = q.z;
    }
    const unsigned long long hit = __ballot(lane < 8 && k == key), emp = __ballot(lane < 8 && k == PAIR_EMPTY);
    const int fh = hit ? __ffsll((long long)hit) - 1 : 64, fe = emp ? __ffsll((long long)emp) - 1 : 64;
    if (fh < fe) { val_out = (uint32_t)__builtin_amdgcn_readlane((int)v, fh); return true; }
    if (fe < 64) return false;
    s0 += 8; if (s0 >= cap) s0 -= cap;
  }
}

// Owner with an optional partition table (VCut routing: the partition recorded for the vertex, modulo world).
__host__ __device__ inline int32_t owner_of_tab(int32_t v, int32_t world, const int32_t *tab, int32_t vmin, int64_t n_slots) {
  if (tab) {
    int64_t s = (int64_t)v - vmin;
    if (s >= 0 && s < n_slots && tab[s] >= 0) return tab[s] % world;
  }
  return owner_of(v, world);
}

}  // namespace srw
