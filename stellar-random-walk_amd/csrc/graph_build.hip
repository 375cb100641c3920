// graph_build.hip — COO lines -> CSR in HBM, entirely on the device (gfx950).
//
// Replaces the adjacency assembly of the reference: flatMap + reduceByKey(_ ++ _) + partitionBy
// (M/algorithm/UniformRandomWalk.scala:26-42, M/algorithm/VCutRandomWalk.scala:19-54) and the GraphMap
// fill (M/algorithm/GraphMap.scala:23-64).  The neighbor list of v is the concatenation, in input-line
// order, of every line's contribution to v, so the entry stream [(src_i -> dst_i), (dst_i -> src_i)]_i is
// STABLY sorted by owning vertex (rocPRIM LSD radix sort is stable).  HBM-bound integer work: no MFMA.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <unordered_set>

#include "engine.h"

namespace srw {
namespace {

constexpr uint32_t KEY_SENTINEL = 0xFFFFFFFFu;
constexpr int TPB = 256;

inline int grid_for(int64_t n, int cap = 256 * 8 * 4) {
  int64_t b = (n + TPB - 1) / TPB;
  if (b < 1) b = 1;
  return (int)std::min<int64_t>(b, cap);
}

__device__ inline uint64_t pack_ent(int32_t id, float w) {
  return ((uint64_t)__float_as_uint(w) << 32) | (uint32_t)id;
}

// One line -> one (directed) or two (undirected) adjacency entries, in line order.
__global__ void k_expand(const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                         const float *__restrict__ w, int64_t n_lines, int directed, int32_t vmin, int32_t rank,
                         int32_t world, uint32_t *__restrict__ keys, uint64_t *__restrict__ vals,
                         uint32_t *__restrict__ present, unsigned long long *owned, uint32_t *__restrict__ gkeys,
                         const int32_t *__restrict__ otab, int64_t n_slots) {
  unsigned long long cnt = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_lines; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t s = src[i], d = dst[i];
    float ww = w ? w[i] : 1.0f;
    uint32_t ks = (uint32_t)((int64_t)s - vmin), kd = (uint32_t)((int64_t)d - vmin);
    present[ks] = 1u;
    present[kd] = 1u;
    bool os = world == 1 || owner_of_tab(s, world, otab, vmin, n_slots) == rank;
    if (directed) {
      keys[i] = os ? ks : KEY_SENTINEL;
      vals[i] = pack_ent(d, ww);
      if (gkeys) gkeys[i] = ks;
      cnt += os;
    } else {
      bool od = world == 1 || owner_of_tab(d, world, otab, vmin, n_slots) == rank;
      keys[2 * i] = os ? ks : KEY_SENTINEL;
      vals[2 * i] = pack_ent(d, ww);
      keys[2 * i + 1] = od ? kd : KEY_SENTINEL;
      vals[2 * i + 1] = pack_ent(s, ww);
      if (gkeys) { gkeys[2 * i] = ks; gkeys[2 * i + 1] = kd; }
      cnt += (unsigned)os + (unsigned)od;
    }
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(owned, cnt);
}

__global__ void k_rows_init(Row *rows, const uint32_t *present, int64_t n_slots) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x) {
    Row r; r.off = 0; r.deg = 0; r.flags = present[v] ? ROW_PRESENT : 0u;
    rows[v] = r;
  }
}
__global__ void k_rows_start(const uint32_t *__restrict__ keys, int64_t n, Row *rows) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    if (e == 0 || keys[e] != keys[e - 1]) rows[keys[e]].off = e;
}
__global__ void k_rows_deg(const uint32_t *__restrict__ keys, int64_t n, Row *rows) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    if (e == n - 1 || keys[e + 1] != keys[e]) rows[keys[e]].deg = (int32_t)(e + 1 - rows[keys[e]].off);
}

__global__ void k_low32(const uint64_t *__restrict__ in, int64_t n, uint32_t *__restrict__ out) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = (uint32_t)in[e];
}
// sorted ids + the weight of each sorted entry (so that a match in the sorted row needs no second, dependent read)
__global__ void k_member_finish(const uint64_t *__restrict__ mk2, const uint32_t *__restrict__ sperm,
                                const Row *__restrict__ rows, const Ent *__restrict__ ent, int64_t n,
                                uint32_t *__restrict__ sids, float *__restrict__ sw) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t k = mk2[e];
    sids[e] = (uint32_t)k;
    sw[e] = ent[rows[k >> 32].off + sperm[e]].w;
  }
}
__global__ void k_owned_flags(const uint32_t *__restrict__ present, int64_t n_slots, int32_t vmin, int32_t rank,
                              int32_t world, const int32_t *__restrict__ otab, uint32_t *__restrict__ out) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x)
    out[v] = (present[v] && owner_of_tab((int32_t)(v + vmin), world, otab, vmin, n_slots) == rank) ? 1u : 0u;
}
__global__ void k_scatter_verts(const uint32_t *__restrict__ flags, const uint32_t *__restrict__ local_pos,
                                const uint32_t *__restrict__ global_pos, int64_t n_slots, int32_t vmin,
                                int32_t *__restrict__ verts, int32_t *__restrict__ vrank) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x)
    if (flags[v]) {
      verts[local_pos[v]] = (int32_t)(v + vmin);
      vrank[local_pos[v]] = (int32_t)global_pos[v];
    }
}

// RMAT edge i = f(seed, i): 4 levels per Philox call; quadrant thresholds floor({.57,.76,.95} * 2^32).
// Same arithmetic as oracle/srw_oracle.c:orc_rmat_edges / orc_rmat_weight (build-defined synthetic input).
__global__ void k_rmat(int32_t scale, uint32_t seed, int64_t first, int64_t n_edges, int weighted, int32_t *__restrict__ src,
                       int32_t *__restrict__ dst, float *__restrict__ w) {
  const uint32_t T1 = 2448131358u, T2 = 3264175144u, T3 = 4080218930u;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_edges; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t e = (uint64_t)(first + i);                     // edge index of the stream: any rank can generate any slice
    uint32_t s = 0, d = 0, o[4] = {0, 0, 0, 0};
    for (int l = 0; l < scale; ++l) {
      if ((l & 3) == 0) philox4x32_10((uint32_t)e, (uint32_t)(e >> 32), (uint32_t)(l >> 2), 0x524D4154u, seed, 1u, o);
      uint32_t x = o[l & 3];
      uint32_t rb = (x >= T2), cb = ((x >= T1 && x < T2) || (x >= T3)) ? 1u : 0u;
      s = (s << 1) | rb;
      d = (d << 1) | cb;
    }
    src[i] = (int32_t)s;
    dst[i] = (int32_t)d;
    if (weighted) {
      uint32_t a = s < d ? s : d, b = s < d ? d : s;
      uint32_t x = b * 0x85EBCA77u;
      uint32_t hsh = seed ^ (a * 0x9E3779B1u) ^ ((x << 13) | (x >> 19));
      hsh ^= hsh >> 16; hsh *= 0x85EBCA6Bu; hsh ^= hsh >> 13; hsh *= 0xC2B2AE35u; hsh ^= hsh >> 16;
      w[i] = (float)(1u + (hsh & 15u));
    }
  }
}

__global__ void k_gmember_keys(const uint32_t *__restrict__ gkeys, const uint64_t *__restrict__ vals, int64_t n, int32_t vmin,
                               uint64_t *__restrict__ out) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = ((uint64_t)gkeys[e] << 32) | (uint32_t)((int64_t)(int32_t)(uint32_t)vals[e] - vmin);
}
__global__ void k_mrows_init(Row *rows, int64_t n_slots) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x) {
    Row r; r.off = 0; r.deg = 0; r.flags = 0; rows[v] = r;
  }
}
__global__ void k_mrows_start(const uint64_t *__restrict__ k64, int64_t n, Row *rows) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    if (e == 0 || (k64[e] >> 32) != (k64[e - 1] >> 32)) rows[k64[e] >> 32].off = e;
}
__global__ void k_mrows_deg(const uint64_t *__restrict__ k64, int64_t n, Row *rows) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    if (e == n - 1 || (k64[e + 1] >> 32) != (k64[e] >> 32)) rows[k64[e] >> 32].deg = (int32_t)(e + 1 - rows[k64[e] >> 32].off);
}

int bits_for(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b)) ++b;
  return b;
}

// ---- neighbor-set filters of the long rows (GraphView::bf_off) ----
__global__ void k_bf_sizes(const Row *__restrict__ rows, int64_t n_slots, unsigned long long *__restrict__ sizes) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x)
    sizes[v] = rows[v].deg >= BF_MIN_DEG ? (unsigned long long)bf_words(rows[v].deg) : 0ull;
}
__global__ void k_bf_offsets(const Row *__restrict__ rows, int64_t n_slots, const unsigned long long *__restrict__ pre, uint32_t *__restrict__ off) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x)
    off[v] = rows[v].deg >= BF_MIN_DEG ? (uint32_t)pre[v] : BF_NONE;
}
// msids != null: rows / ids come from the replicated membership structure of a sharded handle (slot ids, sorted)
__global__ void k_bf_fill(const Row *__restrict__ rows, const Ent *__restrict__ ent, const uint32_t *__restrict__ msids, int64_t n_slots, int32_t vmin,
                          const uint32_t *__restrict__ off, uint32_t *__restrict__ bits, unsigned long long *next_slot) {
  const int lane = threadIdx.x & 63;
  while (true) {
    unsigned long long grab = 0;
    if (lane == 0) grab = atomicAdd(next_slot, 16ull);
    grab = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab);
    if ((int64_t)grab >= n_slots) break;
    for (int64_t v = (int64_t)grab; v < (int64_t)grab + 16 && v < n_slots; ++v) {
      const Row r = rows[v];
      if (r.deg < BF_MIN_DEG) continue;
      const uint32_t nw = bf_words(r.deg);
      uint32_t *f = bits + off[v];
      for (int32_t k = lane; k < r.deg; k += 64) {
        uint32_t word, mask;
        bf_hash(msids ? msids[r.off + k] : (uint32_t)((int64_t)ent[r.off + k].id - vmin), nw, word, mask);
        atomicOr(&f[word], mask);
      }
    }
  }
}

// ---- compacted ids (sparse id spaces) ----
// key = id with the sign bit flipped: unsigned order == int32 order
__global__ void k_ids_concat(const int32_t *__restrict__ src, const int32_t *__restrict__ dst, int64_t n, uint32_t *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    out[i] = (uint32_t)src[i] ^ 0x80000000u; out[n + i] = (uint32_t)dst[i] ^ 0x80000000u;
  }
}
__global__ void k_ids_unflip(const uint32_t *__restrict__ in, int64_t n, int32_t *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (int32_t)(in[i] ^ 0x80000000u);
}
// ids[i] <- rank of ids[i] in tab (ascending, distinct, contains every id)
__global__ void k_ids_rank(int32_t *__restrict__ ids, int64_t n, const int32_t *__restrict__ tab, int64_t m) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t v = ids[i];
    int64_t lo = 0, hi = m - 1;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (tab[mid] < v) lo = mid + 1; else hi = mid; }
    ids[i] = (int32_t)lo;
  }
}

// keys/vals: n_total unsorted entries; n_owned of them carry a real key.  present: [n_slots] flags.
void finish_build(srw_handle *h, DevBuf<uint32_t> &keys, DevBuf<uint64_t> &vals, int64_t n_total, int64_t n_owned,
                  DevBuf<uint32_t> &present, int32_t vmin, int32_t vmax, bool sharded, bool has_sentinels = true) {
  hipStream_t st = h->stream;
  Graph &g = h->g;
  g.vmin = vmin; g.vmax = vmax;
  g.n_slots = (int64_t)vmax - (int64_t)vmin + 1;
  g.n_entries = n_owned;
  g.has_fo = false;
  g.fo.release();
  g.has_cfo = false;
  g.cfo_rejected = false;
  g.cfo.release();
  g.cfo_linked = false;
  g.rows_all.release();
  g.has_al = false;
  g.al.release();
  g.has_pq = false;
  g.pq.release();
  g.has_ehash = false;
  g.has_hub = false; g.hub_bm.release(); g.n_hubs = 0;
  g.ehash.release();

  // 1. stable sort of the entry stream by owning vertex
  DevBuf<uint32_t> keys2; DevBuf<uint64_t> vals2; DevBuf<char> temp;
  keys2.alloc((size_t)n_total); vals2.alloc((size_t)n_total);
  int key_bits = (sharded && has_sentinels) ? 32 : bits_for((uint64_t)std::max<int64_t>(g.n_slots - 1, 1));
  if (n_total > 0) {
    size_t tb = 0;
    SRW_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys2.p, vals.p, vals2.p, (size_t)n_total, 0u,
                                      (unsigned)key_bits, st));
    temp.alloc(tb);
    SRW_HIP(rocprim::radix_sort_pairs((void *)temp.p, tb, keys.p, keys2.p, vals.p, vals2.p, (size_t)n_total, 0u,
                                      (unsigned)key_bits, st));
  }
  SRW_HIP(hipStreamSynchronize(st));
  keys.release(); vals.release(); temp.release();

  // 2. payload: the sorted values ARE the {id, w} records
  if (n_owned == n_total) {
    g.ent.release();
    g.ent.p = reinterpret_cast<Ent *>(vals2.p); g.ent.n = vals2.n;
    vals2.p = nullptr; vals2.n = 0;
  } else {
    g.ent.alloc((size_t)n_owned);
    if (n_owned) SRW_HIP(hipMemcpyAsync(g.ent.p, vals2.p, (size_t)n_owned * sizeof(Ent), hipMemcpyDeviceToDevice, st));
    SRW_HIP(hipStreamSynchronize(st));
    vals2.release();
  }

  // 3. row table
  g.rows.alloc((size_t)g.n_slots);
  hipLaunchKernelGGL(k_rows_init, dim3(grid_for(g.n_slots)), dim3(TPB), 0, st, g.rows.p, present.p, g.n_slots);
  if (n_owned) {
    hipLaunchKernelGGL(k_rows_start, dim3(grid_for(n_owned)), dim3(TPB), 0, st, keys2.p, n_owned, g.rows.p);
    hipLaunchKernelGGL(k_rows_deg, dim3(grid_for(n_owned)), dim3(TPB), 0, st, keys2.p, n_owned, g.rows.p);
  }

  // 4. per-row sorted ids: built lazily (build_membership) — first-order walks never need them
  g.has_member = false;
  g.sids.release(); g.sperm.release(); g.sw.release();
  keys2.release(); temp.release();

  // 5. vertex list (walker seeds, ascending id) + global ranks
  DevBuf<uint32_t> gpos, lflags, lpos;
  gpos.alloc((size_t)g.n_slots);
  size_t tb = 0;
  SRW_HIP(rocprim::exclusive_scan(nullptr, tb, present.p, gpos.p, 0u, (size_t)g.n_slots, rocprim::plus<uint32_t>(), st));
  temp.alloc(tb);
  SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, present.p, gpos.p, 0u, (size_t)g.n_slots, rocprim::plus<uint32_t>(), st));
  uint32_t last_pos = 0, last_flag = 0;
  SRW_HIP(hipMemcpyAsync(&last_pos, gpos.p + (g.n_slots - 1), 4, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipMemcpyAsync(&last_flag, present.p + (g.n_slots - 1), 4, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  g.n_vertices = (int64_t)last_pos + last_flag;
  const uint32_t *flags_p = present.p; const uint32_t *lpos_p = gpos.p;
  g.n_local_vertices = g.n_vertices;
  if (sharded) {
    lflags.alloc((size_t)g.n_slots); lpos.alloc((size_t)g.n_slots);
    hipLaunchKernelGGL(k_owned_flags, dim3(grid_for(g.n_slots)), dim3(TPB), 0, st, present.p, g.n_slots, vmin,
                       h->cfg.rank, h->cfg.world, (const int32_t *)g.owner_tab.p, lflags.p);
    SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, lflags.p, lpos.p, 0u, (size_t)g.n_slots, rocprim::plus<uint32_t>(), st));
    SRW_HIP(hipMemcpyAsync(&last_pos, lpos.p + (g.n_slots - 1), 4, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipMemcpyAsync(&last_flag, lflags.p + (g.n_slots - 1), 4, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
    g.n_local_vertices = (int64_t)last_pos + last_flag;
    flags_p = lflags.p; lpos_p = lpos.p;
  }
  g.verts.alloc((size_t)g.n_local_vertices);
  g.vrank.alloc((size_t)g.n_local_vertices);
  hipLaunchKernelGGL(k_scatter_verts, dim3(grid_for(g.n_slots)), dim3(TPB), 0, st, flags_p, lpos_p, gpos.p, g.n_slots,
                     vmin, g.verts.p, g.vrank.p);
  SRW_HIP(hipStreamSynchronize(st));
  SRW_HIP(hipGetLastError());
  g.loaded = true;
}

}  // namespace

// Every per-vertex structure is dense over slot = id - vmin: the row table, the presence scan, the
// optional owner table, hub bitmaps.  A sparse id space (two ids 2^31 apart) would need tens of GB for nothing: whole-graph
// handles compact it (compact_ids below); a sharded handle refuses it up front with a clear message instead of failing
// somewhere inside an allocation (documented deviation).
void check_id_range(int32_t vmin, int32_t vmax) {
  const int64_t n_slots = (int64_t)vmax - (int64_t)vmin + 1;
  if (n_slots <= 0) throw Error(SRW_ERR_INVALID, "empty vertex id range");
  size_t free_b = 0, total_b = 0;
  const double per_slot = 48.0;                 // rows 16 B + presence/scan/sort temporaries + host mirrors
  const double need = (double)n_slots * per_slot;
  const bool have = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
  if (n_slots >= (int64_t)0xFFFFFFFEll || (have && need > 0.5 * (double)free_b)) {
    char msg[320];
    snprintf(msg, sizeof msg,
             "vertex ids span [%d, %d] = %lld slots: the dense per-vertex tables would need %.1f GB (%.1f GB of HBM free). "
             "This engine indexes vertices by id - min(id); renumber the ids compactly (the reference's HashMap-keyed GraphMap "
             "has no such limit)", vmin, vmax, (long long)n_slots, need / 1e9, (double)free_b / 1e9);
    throw Error(SRW_ERR_NOMEM, msg);
  }
}

// The reference keys vertices in a HashMap (GraphMap.scala:13-15) and takes any int32 ids.  When the id space is sparse
// (the dense tables would be mostly holes, or would not fit at all) the ids are COMPACTED at load: slot = rank of the id
// among the sorted distinct ids of the input.  The rank is monotone, so every order-based structure (sorted rows,
// membership searches, the ascending source order of the output) is the same as over the ids themselves; the two places
// where the id VALUE matters are translated: the Philox key of a walker (rng_source) and the ids of the finished paths
// (k_paths_to_ids at the end of launch_walk / run_shard_flush).  Sharded handles compact the same way: every shard sees the
// whole edge list, so every shard computes the same ranks, and owner(v) is taken over the rank.
bool ids_are_sparse(const srw_handle *h, int64_t n_ids, int32_t vmin, int32_t vmax) {
  if (h->cfg.flags & SRW_CFG_COMPACT_IDS) return true;
  const int64_t n_slots = (int64_t)vmax - (int64_t)vmin + 1;
  if (n_slots > 16 * n_ids + ((int64_t)1 << 22)) return true;      // > 256 B of row descriptors per id that occurs
  // Shards must ALL take the same decision (ownership is taken over the rank when the ids are compacted, over the id when
  // they are not): on a sharded handle it is a pure function of the input — never of this device's free memory, which
  // differs between GPUs and between shards loaded one after the other on one GPU.  Dense tables that do not
  // fit then fail the load (check_id_range, with its message) instead of silently changing the id space on one shard.
  if (h->cfg.world > 1) return false;
  try { check_id_range(vmin, vmax); } catch (const Error &) { return true; }
  return false;
}

void compact_ids(srw_handle *h, int32_t *d_src, int32_t *d_dst, int64_t n_lines, int32_t &vmin, int32_t &vmax, IdMap &m) {
  hipStream_t st = h->stream;
  DevBuf<uint32_t> a, b, uniq; DevBuf<char> temp; DevBuf<unsigned long long> cnt;
  const size_t n2 = (size_t)n_lines * 2;
  a.alloc(n2); b.alloc(n2); cnt.alloc(1);
  hipLaunchKernelGGL(k_ids_concat, dim3(grid_for(n_lines)), dim3(TPB), 0, st, d_src, d_dst, n_lines, a.p);
  size_t tb = 0;
  SRW_HIP(rocprim::radix_sort_keys(nullptr, tb, a.p, b.p, n2, 0u, 32u, st));
  temp.alloc(tb);
  SRW_HIP(rocprim::radix_sort_keys((void *)temp.p, tb, a.p, b.p, n2, 0u, 32u, st));
  SRW_HIP(rocprim::unique(nullptr, tb, b.p, a.p, cnt.p, n2, rocprim::equal_to<uint32_t>(), st));
  temp.alloc(tb);
  SRW_HIP(rocprim::unique((void *)temp.p, tb, b.p, a.p, cnt.p, n2, rocprim::equal_to<uint32_t>(), st));
  unsigned long long n_u = 0;
  SRW_HIP(hipMemcpyAsync(&n_u, cnt.p, 8, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  b.release(); temp.release();
  if (n_u > 0x7FFFFFFFull) throw Error(SRW_ERR_INVALID, "more than 2^31 - 1 distinct vertex ids");   // ranks are int32
  m.orig_id.alloc((size_t)n_u);
  hipLaunchKernelGGL(k_ids_unflip, dim3(grid_for((int64_t)n_u)), dim3(TPB), 0, st, a.p, (int64_t)n_u, m.orig_id.p);
  hipLaunchKernelGGL(k_ids_rank, dim3(grid_for(n_lines)), dim3(TPB), 0, st, d_src, n_lines, (const int32_t *)m.orig_id.p, (int64_t)n_u);
  hipLaunchKernelGGL(k_ids_rank, dim3(grid_for(n_lines)), dim3(TPB), 0, st, d_dst, n_lines, (const int32_t *)m.orig_id.p, (int64_t)n_u);
  SRW_HIP(hipGetLastError());
  m.h_orig_id.resize((size_t)n_u);
  SRW_HIP(hipMemcpyAsync(m.h_orig_id.data(), m.orig_id.p, (size_t)n_u * 4, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  m.compact = true; m.id_lo = vmin; m.id_hi = vmax;
  vmin = 0; vmax = (int32_t)(n_u - 1);
}

// Optional accelerator of the table steps (binned_resolve): skipped when HBM is short or the offsets would not fit 32 bits.
namespace {
__global__ void k_unit_check(const Ent *__restrict__ ent, int64_t n, unsigned int *flag) {
  bool bad = false;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) bad |= ent[e].w != 1.0f;
  if (__any(bad) && (threadIdx.x & 63u) == 0u) atomicOr(flag, 1u);
}
__global__ void k_unit_ids(const Ent *__restrict__ ent, int64_t n, int32_t *__restrict__ ids) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) ids[e] = ent[e].id;
}
}  // namespace
// Unit-weight graphs: the table steps of the biased walk read ids32[e] (4 B) instead of ent[e] (8 B) — they are bound by memory requests
// (config 5's stand-in: 0.9 of the request ceiling), and a located chunk's entries are a third of them.
bool graph_has_unit_weights(srw_handle *h) {
  Graph &g = h->g;
  if (g.n_entries <= 0) return false;
  hipStream_t st = h->stream;
  if (g.unit_w < 0) {
    DevBuf<unsigned int> flag; flag.alloc(1);
    SRW_HIP(hipMemsetAsync(flag.p, 0, 4, st));
    hipLaunchKernelGGL(k_unit_check, dim3(h->n_cus * 8), dim3(TPB), 0, st, g.ent.p, g.n_entries, flag.p);
    unsigned int f = 0;
    SRW_HIP(hipMemcpyAsync(&f, flag.p, 4, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
    g.unit_w = f ? 0 : 1;
  }
  return g.unit_w == 1;
}
void build_unit_ids(srw_handle *h) {
  Graph &g = h->g;
  if (g.has_ids32 || g.unit_w == 0 || g.n_entries <= 0 || getenv("SRW_NO_UNIT_IDS")) return;
  hipStream_t st = h->stream;
  if (g.unit_w < 0) graph_has_unit_weights(h);
  if (false) {
    DevBuf<unsigned int> flag; flag.alloc(1);
    SRW_HIP(hipMemsetAsync(flag.p, 0, 4, st));
    hipLaunchKernelGGL(k_unit_check, dim3(h->n_cus * 8), dim3(TPB), 0, st, g.ent.p, g.n_entries, flag.p);
    unsigned int f = 0;
    SRW_HIP(hipMemcpyAsync(&f, flag.p, 4, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
    g.unit_w = f ? 0 : 1;
  }
  if (!g.unit_w) return;
  size_t free_b = 0, total_b = 0;
  SRW_HIP(hipMemGetInfo(&free_b, &total_b));
  if (free_b < (size_t)g.n_entries * 4 + ((size_t)32 << 30)) return;          // optional: not when HBM is short
  g.ids32.alloc((size_t)g.n_entries);
  hipLaunchKernelGGL(k_unit_ids, dim3(h->n_cus * 8), dim3(TPB), 0, st, g.ent.p, g.n_entries, g.ids32.p);
  SRW_HIP(hipGetLastError());
  g.has_ids32 = true;
}

void build_row_filters(srw_handle *h) {
  Graph &g = h->g;
  if (g.has_bf || g.n_entries_global <= 0) return;
  if (h->cfg.world != 1 && !g.mrows.p) return;          // a shard without the membership structure never asks "x in N(prev)?"
  hipStream_t st = h->stream;
  const Row *frows = g.mrows.p ? g.mrows.p : g.rows.p;   // N(prev) of ANY vertex: the whole graph's rows
  const uint32_t *fsids = g.mrows.p ? g.msids.p : nullptr;
  DevBuf<unsigned long long> sizes, next_slot; DevBuf<char> temp;
  sizes.alloc((size_t)g.n_slots + 1); next_slot.alloc(1);
  SRW_HIP(hipMemsetAsync(sizes.p + g.n_slots, 0, 8, st));
  hipLaunchKernelGGL(k_bf_sizes, dim3(grid_for(g.n_slots)), dim3(TPB), 0, st, frows, g.n_slots, sizes.p);
  size_t tb = 0;
  SRW_HIP(rocprim::exclusive_scan(nullptr, tb, sizes.p, sizes.p, 0ull, (size_t)g.n_slots + 1, rocprim::plus<unsigned long long>(), st));
  temp.alloc(tb);
  SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, sizes.p, sizes.p, 0ull, (size_t)g.n_slots + 1, rocprim::plus<unsigned long long>(), st));
  unsigned long long total = 0;
  SRW_HIP(hipMemcpyAsync(&total, sizes.p + g.n_slots, 8, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  if (total == 0 || total >= 0xFFFFFFF0ull) return;
  size_t free_b = 0, total_b = 0;
  SRW_HIP(hipMemGetInfo(&free_b, &total_b));
  if (free_b < (size_t)total * 4 + (size_t)g.n_slots * 4 + ((size_t)(h->cfg.world > 1 ? 8 : 32) << 30)) return;
  g.bf_off.alloc((size_t)g.n_slots); g.bf_bits.alloc((size_t)total);
  hipLaunchKernelGGL(k_bf_offsets, dim3(grid_for(g.n_slots)), dim3(TPB), 0, st, frows, g.n_slots, sizes.p, g.bf_off.p);
  SRW_HIP(hipMemsetAsync(g.bf_bits.p, 0, (size_t)total * 4, st));
  SRW_HIP(hipMemsetAsync(next_slot.p, 0, 8, st));
  hipLaunchKernelGGL(k_bf_fill, dim3(256 * 8), dim3(TPB), 0, st, frows, g.ent.p, fsids, g.n_slots, g.vmin, g.bf_off.p, g.bf_bits.p, next_slot.p);
  SRW_HIP(hipGetLastError());
  SRW_HIP(hipStreamSynchronize(st));
  g.has_bf = true;
  if (getenv("SRW_TIMING")) fprintf(stderr, "[timing] row filters: %.2f GB for the rows beyond %d neighbors\n", (double)total * 4 / 1e9, BF_MIN_DEG - 1);
}

static void install_id_map(Graph &g, IdMap *m) {
  g.id_lo = g.vmin; g.id_hi = g.vmax;
  if (!m || !m->compact) return;
  g.compact = true; g.orig_id = std::move(m->orig_id); g.h_orig_id = std::move(m->h_orig_id);
  g.id_lo = m->id_lo; g.id_hi = m->id_hi;
}

namespace {
// org.apache.spark.HashPartitioner on Int keys (Utils.nonNegativeMod(key.hashCode, numPartitions), RandomWalk.scala:16) as a partition table
__global__ void k_owner_mod(int32_t *__restrict__ tab, int64_t n_slots, int32_t vmin, int32_t world) {
  for (int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; s < n_slots; s += (int64_t)gridDim.x * blockDim.x) {
    const int64_t id = s + (int64_t)vmin;
    const int64_t r = id % (int64_t)world;
    tab[s] = (int32_t)(r < 0 ? r + world : r);
  }
}
}  // namespace
void build_graph_from_device_lines(srw_handle *h, const int32_t *d_src, const int32_t *d_dst, const float *d_w,
                                   int64_t n_lines, bool directed, int32_t vmin, int32_t vmax,
                                   const int32_t *host_owner_tab, IdMap *idmap) {
  if (n_lines <= 0) throw Error(SRW_ERR_INVALID, "empty edge list");
  if (h->cfg.world > 1 && !getenv("SRW_BUILD_WHOLE")) {      // a shard keeps only what it owns: blocks of the line arrays
    build_graph_blocked(h, [&](int64_t i0, int64_t, const int32_t *&s, const int32_t *&d, const float *&w) {
      s = d_src + i0; d = d_dst + i0; w = d_w ? d_w + i0 : nullptr;
    }, n_lines, directed, vmin, vmax, host_owner_tab, idmap);
    return;
  }
  int64_t n_slots = (int64_t)vmax - (int64_t)vmin + 1;
  check_id_range(vmin, vmax);
  hipStream_t st = h->stream;
  Graph &g = h->g;
  g = Graph();
  g.n_lines = n_lines;
  g.symmetric = !directed;
  int64_t n_total = directed ? n_lines : 2 * n_lines;
  g.n_entries_global = n_total;
  bool sharded = h->cfg.world > 1;
  if (sharded && host_owner_tab && (h->cfg.flags & SRW_CFG_OWNER_FROM_PARTITIONS)) {
    g.owner_tab.alloc((size_t)n_slots);
    SRW_HIP(hipMemcpyAsync(g.owner_tab.p, host_owner_tab, (size_t)n_slots * 4, hipMemcpyHostToDevice, st));
  } else if (sharded && (h->cfg.flags & SRW_CFG_OWNER_HASH_PARTITIONER) && !idmap) {
    g.owner_tab.alloc((size_t)n_slots);         // HashPartitioner's map as a partition table: owner = nonNegativeMod(id, world)
    hipLaunchKernelGGL(k_owner_mod, dim3(grid_for(n_slots)), dim3(TPB), 0, st, g.owner_tab.p, n_slots, vmin, h->cfg.world);
  }

  // SRW_CFG_NO_MEMBERSHIP: a shard that will only run q == 1 walks skips the replicated membership structure
  const bool want_membership = sharded && !(h->cfg.flags & SRW_CFG_NO_MEMBERSHIP);
  DevBuf<uint32_t> keys, present, gkeys; DevBuf<uint64_t> vals;
  keys.alloc((size_t)n_total); vals.alloc((size_t)n_total); present.alloc((size_t)n_slots);
  if (want_membership) gkeys.alloc((size_t)n_total);
  SRW_HIP(hipMemsetAsync(present.p, 0, (size_t)n_slots * 4, st));
  h->counters.ensure(1);
  SRW_HIP(hipMemsetAsync(h->counters.p, 0, sizeof(DevCounters), st));
  hipLaunchKernelGGL(k_expand, dim3(grid_for(n_lines)), dim3(TPB), 0, st, d_src, d_dst, d_w, n_lines, directed ? 1 : 0,
                     vmin, h->cfg.rank, h->cfg.world, keys.p, vals.p, present.p, &h->counters.p->owned_entries,
                     want_membership ? gkeys.p : (uint32_t *)nullptr, (const int32_t *)g.owner_tab.p, n_slots);
  unsigned long long owned = 0;
  SRW_HIP(hipMemcpyAsync(&owned, &h->counters.p->owned_entries, 8, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  DevBuf<Row> mrows; DevBuf<uint32_t> msids;
  if (want_membership) {
    // replicated membership structure of the WHOLE graph (N(prev) must be testable on the shard that owns curr):
    // sort every entry by (row, id - vmin); keep the row boundaries and the sorted ids (4 B/entry)
    DevBuf<uint64_t> mk, mk2; DevBuf<char> temp;
    mk.alloc((size_t)n_total); mk2.alloc((size_t)n_total);
    hipLaunchKernelGGL(k_gmember_keys, dim3(grid_for(n_total)), dim3(TPB), 0, st, gkeys.p, vals.p, n_total, vmin, mk.p);
    int id_bits = bits_for((uint64_t)std::max<int64_t>(n_slots - 1, 1));
    size_t tb = 0;
    SRW_HIP(rocprim::radix_sort_keys(nullptr, tb, mk.p, mk2.p, (size_t)n_total, 0u, (unsigned)(32 + id_bits), st));
    temp.alloc(tb);
    SRW_HIP(rocprim::radix_sort_keys((void *)temp.p, tb, mk.p, mk2.p, (size_t)n_total, 0u, (unsigned)(32 + id_bits), st));
    mrows.alloc((size_t)n_slots); msids.alloc((size_t)n_total);
    hipLaunchKernelGGL(k_mrows_init, dim3(grid_for(n_slots)), dim3(TPB), 0, st, mrows.p, n_slots);
    hipLaunchKernelGGL(k_mrows_start, dim3(grid_for(n_total)), dim3(TPB), 0, st, mk2.p, n_total, mrows.p);
    hipLaunchKernelGGL(k_mrows_deg, dim3(grid_for(n_total)), dim3(TPB), 0, st, mk2.p, n_total, mrows.p);
    hipLaunchKernelGGL(k_low32, dim3(grid_for(n_total)), dim3(TPB), 0, st, mk2.p, n_total, msids.p);
    SRW_HIP(hipStreamSynchronize(st));
    gkeys.release();
  }
  finish_build(h, keys, vals, n_total, (int64_t)owned, present, vmin, vmax, sharded);
  if (want_membership) { g.mrows = std::move(mrows); g.msids = std::move(msids); }
  install_id_map(g, idmap);
}

namespace {
struct OwnedKey { __host__ __device__ bool operator()(uint32_t k) const { return k != KEY_SENTINEL; } };
}  // namespace

// The same construction for a VERTEX-SHARDED handle, block by block: the reference's partitionBy hands a partition only its
// rows (UniformRandomWalk.scala:41-42); round 2's shards each expanded, keyed and sorted the WHOLE entry stream and cut
// afterwards (24 B per adjacency entry of the whole graph per shard: ~190 GB at RMAT-27 with the membership structure).
// Here the lines come in blocks (`fetch`: a slice of arrays already on the device, a block generated on the spot, or a block
// uploaded from the host); pass 1 counts the owned entries and marks the present ids, pass 2 expands every block again and
// appends its OWNED (key, entry) pairs — stable, so a row keeps its input-line order — to arrays of exactly the owned size,
// which are then sorted.  Peak: 24 B per OWNED entry + two blocks; the replicated membership structure, when wanted, is
// still the whole graph's (8 + 8 B per entry while it is sorted, 4 B after).
void build_graph_blocked(srw_handle *h, const LineFetch &fetch, int64_t n_lines, bool directed, int32_t vmin, int32_t vmax,
                         const int32_t *host_owner_tab, IdMap *idmap) {
  if (n_lines <= 0) throw Error(SRW_ERR_INVALID, "empty edge list");
  const int64_t n_slots = (int64_t)vmax - (int64_t)vmin + 1;
  check_id_range(vmin, vmax);
  hipStream_t st = h->stream;
  Graph &g = h->g;
  g = Graph();
  g.n_lines = n_lines;
  g.symmetric = !directed;
  const int per = directed ? 1 : 2;
  const int64_t n_total = n_lines * per;
  g.n_entries_global = n_total;
  if (host_owner_tab && (h->cfg.flags & SRW_CFG_OWNER_FROM_PARTITIONS)) {
    g.owner_tab.alloc((size_t)n_slots);
    SRW_HIP(hipMemcpyAsync(g.owner_tab.p, host_owner_tab, (size_t)n_slots * 4, hipMemcpyHostToDevice, st));
  } else if ((h->cfg.flags & SRW_CFG_OWNER_HASH_PARTITIONER) && !idmap) {
    g.owner_tab.alloc((size_t)n_slots);         // HashPartitioner's map as a partition table: owner = nonNegativeMod(id, world)
    hipLaunchKernelGGL(k_owner_mod, dim3(grid_for(n_slots)), dim3(TPB), 0, st, g.owner_tab.p, n_slots, vmin, h->cfg.world);
  }
  const bool want_membership = !(h->cfg.flags & SRW_CFG_NO_MEMBERSHIP);
  int64_t BL = (int64_t)32 << 20;
  if (const char *e = getenv("SRW_BUILD_BLOCK_LINES"); e && *e) BL = std::max<int64_t>(1, atoll(e));
  BL = std::min(BL, n_lines);
  DevBuf<uint32_t> bkeys, bgk, present; DevBuf<uint64_t> bvals, mk; DevBuf<unsigned long long> cnt; DevBuf<char> temp;
  bkeys.alloc((size_t)(BL * per)); bvals.alloc((size_t)(BL * per)); present.alloc((size_t)n_slots); cnt.alloc(2);
  if (want_membership) { bgk.alloc((size_t)(BL * per)); mk.alloc((size_t)n_total); }
  SRW_HIP(hipMemsetAsync(present.p, 0, (size_t)n_slots * 4, st));
  SRW_HIP(hipMemsetAsync(cnt.p, 0, 16, st));
  auto expand = [&](int64_t i0, int64_t n, unsigned long long *owned_ctr, bool with_gkeys) {
    const int32_t *s = nullptr, *d = nullptr; const float *w = nullptr;
    fetch(i0, n, s, d, w);
    hipLaunchKernelGGL(k_expand, dim3(grid_for(n)), dim3(TPB), 0, st, s, d, w, n, directed ? 1 : 0, vmin, h->cfg.rank, h->cfg.world,
                       bkeys.p, bvals.p, present.p, owned_ctr, with_gkeys ? bgk.p : (uint32_t *)nullptr, (const int32_t *)g.owner_tab.p, n_slots);
  };
  // pass 1: owned entries, present ids, membership keys
  for (int64_t i0 = 0; i0 < n_lines; i0 += BL) {
    const int64_t n = std::min(BL, n_lines - i0);
    expand(i0, n, cnt.p, want_membership);
    if (want_membership)
      hipLaunchKernelGGL(k_gmember_keys, dim3(grid_for(n * per)), dim3(TPB), 0, st, bgk.p, bvals.p, n * per, vmin, mk.p + i0 * per);
    SRW_HIP(hipStreamSynchronize(st));            // the fetch of the next block may reuse its staging
  }
  unsigned long long owned = 0;
  SRW_HIP(hipMemcpyAsync(&owned, cnt.p, 8, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  SRW_HIP(hipGetLastError());
  bgk.release();
  // pass 2: the owned (key, entry) pairs, in line order
  DevBuf<uint32_t> keys; DevBuf<uint64_t> vals;
  keys.alloc((size_t)std::max<unsigned long long>(owned, 1)); vals.alloc((size_t)std::max<unsigned long long>(owned, 1));
  {
    auto flags = rocprim::make_transform_iterator(bkeys.p, OwnedKey());
    size_t tb = 0, tb2 = 0;
    SRW_HIP(rocprim::select(nullptr, tb, bkeys.p, flags, keys.p, cnt.p + 1, (size_t)(BL * per), st));
    SRW_HIP(rocprim::select(nullptr, tb2, bvals.p, flags, vals.p, cnt.p + 1, (size_t)(BL * per), st));
    temp.alloc(std::max(tb, tb2));
    tb = tb2 = std::max(tb, tb2);
    int64_t off = 0;
    for (int64_t i0 = 0; i0 < n_lines; i0 += BL) {
      const int64_t n = std::min(BL, n_lines - i0);
      expand(i0, n, cnt.p + 1, false);
      size_t t1 = tb;
      SRW_HIP(rocprim::select((void *)temp.p, t1, bkeys.p, flags, keys.p + off, cnt.p + 1, (size_t)(n * per), st));
      t1 = tb;
      SRW_HIP(rocprim::select((void *)temp.p, t1, bvals.p, flags, vals.p + off, cnt.p + 1, (size_t)(n * per), st));
      unsigned long long got = 0;
      SRW_HIP(hipMemcpyAsync(&got, cnt.p + 1, 8, hipMemcpyDeviceToHost, st));
      SRW_HIP(hipStreamSynchronize(st));
      off += (int64_t)got;
    }
    if ((unsigned long long)off != owned) throw Error(SRW_ERR_INVALID, "sharded build: the two passes over the lines disagree");
  }
  bkeys.release(); bvals.release(); temp.release();
  DevBuf<Row> mrows; DevBuf<uint32_t> msids;
  if (want_membership) {
    DevBuf<uint64_t> mk2;
    mk2.alloc((size_t)n_total);
    const int id_bits = bits_for((uint64_t)std::max<int64_t>(n_slots - 1, 1));
    size_t tb = 0;
    SRW_HIP(rocprim::radix_sort_keys(nullptr, tb, mk.p, mk2.p, (size_t)n_total, 0u, (unsigned)(32 + id_bits), st));
    temp.alloc(tb);
    SRW_HIP(rocprim::radix_sort_keys((void *)temp.p, tb, mk.p, mk2.p, (size_t)n_total, 0u, (unsigned)(32 + id_bits), st));
    SRW_HIP(hipStreamSynchronize(st));
    mk.release(); temp.release();
    mrows.alloc((size_t)n_slots); msids.alloc((size_t)n_total);
    hipLaunchKernelGGL(k_mrows_init, dim3(grid_for(n_slots)), dim3(TPB), 0, st, mrows.p, n_slots);
    hipLaunchKernelGGL(k_mrows_start, dim3(grid_for(n_total)), dim3(TPB), 0, st, mk2.p, n_total, mrows.p);
    hipLaunchKernelGGL(k_mrows_deg, dim3(grid_for(n_total)), dim3(TPB), 0, st, mk2.p, n_total, mrows.p);
    hipLaunchKernelGGL(k_low32, dim3(grid_for(n_total)), dim3(TPB), 0, st, mk2.p, n_total, msids.p);
    SRW_HIP(hipStreamSynchronize(st));
  }
  finish_build(h, keys, vals, (int64_t)owned, (int64_t)owned, present, vmin, vmax, true, /*has_sentinels=*/false);
  if (want_membership) { g.mrows = std::move(mrows); g.msids = std::move(msids); }
  install_id_map(g, idmap);
}

void build_graph_from_host_rows(srw_handle *h, const int32_t *vids, const int64_t *offs, int64_t n_rows,
                                const int32_t *ids, const float *w) {
  if (n_rows <= 0) throw Error(SRW_ERR_INVALID, "no vertices");
  if (h->cfg.world > 1) throw Error(SRW_ERR_INVALID, "srw_load_adjacency needs world == 1");
  // GraphMap.addVertex: first occurrence of a vertex wins (GraphMap.scala:24-25,37).
  int32_t vmin = vids[0], vmax = vids[0];
  int64_t n_ent_in = offs[n_rows];
  for (int64_t i = 0; i < n_rows; ++i) { vmin = std::min(vmin, vids[i]); vmax = std::max(vmax, vids[i]); }
  for (int64_t e = 0; e < n_ent_in; ++e) { vmin = std::min(vmin, ids[e]); vmax = std::max(vmax, ids[e]); }
  IdMap idmap;
  if (ids_are_sparse(h, n_rows + n_ent_in, vmin, vmax)) {      // this surface hands over host arrays: compact on the host
    std::vector<int32_t> &u = idmap.h_orig_id;
    u.assign(vids, vids + n_rows); u.insert(u.end(), ids, ids + n_ent_in);
    std::sort(u.begin(), u.end()); u.erase(std::unique(u.begin(), u.end()), u.end());
    idmap.compact = true; idmap.id_lo = vmin; idmap.id_hi = vmax;
    vmin = 0; vmax = (int32_t)(u.size() - 1);
  }
  auto slot = [&](int32_t v) -> uint32_t {
    if (!idmap.compact) return (uint32_t)((int64_t)v - vmin);
    return (uint32_t)(std::lower_bound(idmap.h_orig_id.begin(), idmap.h_orig_id.end(), v) - idmap.h_orig_id.begin());
  };
  int64_t n_slots = (int64_t)vmax - (int64_t)vmin + 1;
  check_id_range(vmin, vmax);
  std::vector<uint32_t> hkeys; std::vector<uint64_t> hvals; std::vector<uint32_t> hpresent((size_t)n_slots, 0u);
  hkeys.reserve((size_t)n_ent_in); hvals.reserve((size_t)n_ent_in);
  for (int64_t i = 0; i < n_rows; ++i) {
    uint32_t k = slot(vids[i]);
    if (hpresent[k]) continue;
    hpresent[k] = 1u;
    for (int64_t e = offs[i]; e < offs[i + 1]; ++e) {
      float ww = w ? w[e] : 1.0f; uint32_t wb; memcpy(&wb, &ww, 4);
      hkeys.push_back(k);
      hvals.push_back(((uint64_t)wb << 32) | (uint32_t)(idmap.compact ? (int32_t)slot(ids[e]) : ids[e]));
    }
  }
  hipStream_t st = h->stream;
  Graph &g = h->g;
  g = Graph();
  int64_t n_total = (int64_t)hkeys.size();
  g.n_lines = n_rows; g.n_entries_global = n_total;
  DevBuf<uint32_t> keys, present; DevBuf<uint64_t> vals;
  keys.alloc((size_t)n_total); vals.alloc((size_t)n_total); present.alloc((size_t)n_slots);
  if (n_total) {
    SRW_HIP(hipMemcpyAsync(keys.p, hkeys.data(), (size_t)n_total * 4, hipMemcpyHostToDevice, st));
    SRW_HIP(hipMemcpyAsync(vals.p, hvals.data(), (size_t)n_total * 8, hipMemcpyHostToDevice, st));
  }
  SRW_HIP(hipMemcpyAsync(present.p, hpresent.data(), (size_t)n_slots * 4, hipMemcpyHostToDevice, st));
  SRW_HIP(hipStreamSynchronize(st));
  finish_build(h, keys, vals, n_total, n_total, present, vmin, vmax, false);
  if (idmap.compact) {
    idmap.orig_id.alloc(idmap.h_orig_id.size());
    SRW_HIP(hipMemcpy(idmap.orig_id.p, idmap.h_orig_id.data(), idmap.h_orig_id.size() * 4, hipMemcpyHostToDevice));
  }
  install_id_map(g, &idmap);
}

namespace {
// (row slot << 32 | id - vmin) and the input-order position of every entry, from the row table
__global__ void k_member_keys_rows(const Row *__restrict__ rows, const Ent *__restrict__ ent, int64_t n_slots, int32_t vmin,
                                   uint64_t *__restrict__ mk, uint32_t *__restrict__ li, unsigned long long *next_slot) {
  const int lane = threadIdx.x & 63;
  while (true) {
    unsigned long long grab = 0;
    if (lane == 0) grab = atomicAdd(next_slot, 4ull);
    grab = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab);
    if ((int64_t)grab >= n_slots) break;
    for (int64_t v = (int64_t)grab; v < (int64_t)grab + 4 && v < n_slots; ++v) {
      const Row r = rows[v];
      for (int32_t k = lane; k < r.deg; k += 64) {
        mk[r.off + k] = ((uint64_t)(uint32_t)v << 32) | (uint32_t)((int64_t)ent[r.off + k].id - vmin);
        li[r.off + k] = (uint32_t)k;
      }
    }
  }
}
}  // namespace

// Membership structure of computeSecondOrderWeights' `exists` (RandomSample.scala:37): per-row sorted ids + the
// input-order position of each sorted entry.  Needed by the general and alias kernels only, so it is built on first use.
void build_membership(srw_handle *h) {
  Graph &g = h->g;
  if (g.has_member) return;
  hipStream_t st = h->stream;
  const int64_t n = g.n_entries;
  g.sids.alloc((size_t)std::max<int64_t>(n, 1));
  g.sperm.alloc((size_t)std::max<int64_t>(n, 1));
  g.sw.alloc((size_t)std::max<int64_t>(n, 1));
  if (n > 0) {
    DevBuf<uint64_t> mk, mk2; DevBuf<uint32_t> li; DevBuf<char> temp; DevBuf<unsigned long long> next_slot;
    mk.alloc((size_t)n); mk2.alloc((size_t)n); li.alloc((size_t)n); next_slot.alloc(1);
    SRW_HIP(hipMemsetAsync(next_slot.p, 0, 8, st));
    hipLaunchKernelGGL(k_member_keys_rows, dim3(256 * 8), dim3(TPB), 0, st, g.rows.p, g.ent.p, g.n_slots, g.vmin, mk.p, li.p,
                       next_slot.p);
    int id_bits = bits_for((uint64_t)std::max<int64_t>(g.n_slots - 1, 1));
    size_t tb = 0;
    SRW_HIP(rocprim::radix_sort_pairs(nullptr, tb, mk.p, mk2.p, li.p, g.sperm.p, (size_t)n, 0u, (unsigned)(32 + id_bits), st));
    temp.alloc(tb);
    SRW_HIP(rocprim::radix_sort_pairs((void *)temp.p, tb, mk.p, mk2.p, li.p, g.sperm.p, (size_t)n, 0u,
                                      (unsigned)(32 + id_bits), st));
    hipLaunchKernelGGL(k_member_finish, dim3(grid_for(n)), dim3(TPB), 0, st, mk2.p, g.sperm.p, g.rows.p, g.ent.p, n, g.sids.p,
                       g.sw.p);
    SRW_HIP(hipStreamSynchronize(st));
    SRW_HIP(hipGetLastError());
  }
  g.has_member = true;
}

namespace {
__global__ void k_hub_degrees(const Row *__restrict__ rows, int64_t n_slots, uint32_t *__restrict__ deg) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x)
    deg[v] = (uint32_t)rows[v].deg;
}
// ordinals for the rows at or above the threshold (first come first served up to max_hubs), kept in Row::flags
__global__ void k_hub_assign(Row *__restrict__ rows, int64_t n_slots, int32_t thr, unsigned long long max_hubs,
                             unsigned long long *counter, uint32_t *__restrict__ hub_slot) {
  for (int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; v < n_slots; v += (int64_t)gridDim.x * blockDim.x) {
    uint32_t f = rows[v].flags & ((1u << ROW_HUB_SHIFT) - 1u);
    if (rows[v].deg >= thr && rows[v].deg > 0) {
      const unsigned long long o = atomicAdd(counter, 1ull);
      if (o < max_hubs) { f |= (uint32_t)(o + 1) << ROW_HUB_SHIFT; hub_slot[o] = (uint32_t)v; }
    }
    rows[v].flags = f;
  }
}
__global__ void k_hub_fill(const Row *__restrict__ rows, const Ent *__restrict__ ent, const uint32_t *__restrict__ msids, const uint32_t *__restrict__ hub_slot,
                           int64_t n_hubs, int32_t vmin, int64_t words, uint32_t *__restrict__ bm,
                           unsigned long long *next_hub) {
  const int lane = threadIdx.x & 63;
  while (true) {
    unsigned long long grab = 0;
    if (lane == 0) grab = atomicAdd(next_hub, 1ull);
    grab = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(grab >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)grab);
    if ((int64_t)grab >= n_hubs) break;
    const Row r = rows[hub_slot[grab]];
    uint32_t *mine = bm + (int64_t)grab * words;
    for (int32_t k = lane; k < r.deg; k += 64) {
      const uint32_t x = msids ? msids[r.off + k] : (uint32_t)((int64_t)ent[r.off + k].id - vmin);
      atomicOr(&mine[x >> 5], 1u << (x & 31));
    }
  }
}
}  // namespace

// Neighbor-set bitmaps over the id slots for the highest-degree rows, as many as the memory budget allows: a step
// whose PREVIOUS vertex is such a hub tests "x in N(prev)" with one L2-resident bit read per candidate instead of
// intersecting two sorted rows (sampling.h, strategy P3).  Steps land on a vertex in proportion to its degree, so a
// few thousand bitmaps cover most second-order steps of a power-law graph.
void build_hub_bitmaps(srw_handle *h, int32_t min_deg, size_t budget_cap) {
  Graph &g = h->g;
  if (g.has_hub && g.hub_min_deg == min_deg && g.hub_budget_cap == budget_cap) return;
  hipStream_t st = h->stream;
  g.has_hub = false; g.n_hubs = 0;
  // A sharded handle asks "x in N(prev)?" for a prev it does not own: the hubs are those of the WHOLE graph, their ordinals
  // live in the membership rows (Row::flags of mrows, otherwise unused) and their bitmaps are filled from the sorted ids.
  Row *hrows = g.mrows.p ? g.mrows.p : g.rows.p;
  const uint32_t *hsids = g.mrows.p ? g.msids.p : nullptr;
  if (h->cfg.world != 1 && !g.mrows.p) { g.has_hub = true; g.hub_min_deg = min_deg; g.hub_budget_cap = budget_cap; return; }
  const int64_t words = (g.n_slots + 31) / 32;
  size_t free_b = 0, total_b = 0;
  SRW_HIP(hipMemGetInfo(&free_b, &total_b));
  free_b += g.hub_bm.n * sizeof(uint32_t);                       // the previous set is released below
  const size_t budget = std::min<size_t>(budget_cap, free_b > ((size_t)24 << 30) ? (free_b - ((size_t)24 << 30)) / 2 : 0);
  int64_t max_hubs = (int64_t)(budget / ((size_t)words * 4));
  max_hubs = std::min<int64_t>(max_hubs, ((int64_t)1 << (32 - ROW_HUB_SHIFT)) - 2);
  // threshold = degree of the max_hubs-th largest row (device sort of the degrees), but never below min_deg
  int32_t thr = min_deg;
  if (max_hubs > 0 && g.n_slots > max_hubs) {
    DevBuf<uint32_t> d_in, d_out; DevBuf<char> temp;
    d_in.alloc((size_t)g.n_slots); d_out.alloc((size_t)g.n_slots);
    hipLaunchKernelGGL(k_hub_degrees, dim3(grid_for(g.n_slots)), dim3(TPB), 0, st, hrows, g.n_slots, d_in.p);
    size_t tb = 0;
    SRW_HIP(rocprim::radix_sort_keys_desc(nullptr, tb, d_in.p, d_out.p, (size_t)g.n_slots, 0u, 32u, st));
    temp.alloc(tb);
    SRW_HIP(rocprim::radix_sort_keys_desc((void *)temp.p, tb, d_in.p, d_out.p, (size_t)g.n_slots, 0u, 32u, st));
    uint32_t kth = 0;
    SRW_HIP(hipMemcpyAsync(&kth, d_out.p + (max_hubs - 1), 4, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
    thr = std::max<int32_t>(min_deg, (int32_t)std::min<uint32_t>(kth, 0x7FFFFFFFu));
  }
  DevBuf<unsigned long long> counter; counter.alloc(1);
  DevBuf<uint32_t> hub_slot; hub_slot.alloc((size_t)std::max<int64_t>(max_hubs, 1));
  SRW_HIP(hipMemsetAsync(counter.p, 0, 8, st));
  // max_hubs == 0 (no memory): the kernel still clears stale ordinals
  hipLaunchKernelGGL(k_hub_assign, dim3(grid_for(g.n_slots)), dim3(TPB), 0, st, hrows, g.n_slots, thr,
                     (unsigned long long)std::max<int64_t>(max_hubs, 0), counter.p, hub_slot.p);
  unsigned long long n = 0;
  SRW_HIP(hipMemcpyAsync(&n, counter.p, 8, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  g.n_hubs = (int64_t)std::min<unsigned long long>(n, (unsigned long long)std::max<int64_t>(max_hubs, 0));
  g.hub_words = words; g.hub_min_deg = min_deg; g.hub_budget_cap = budget_cap;
  g.hub_bm.release();
  if (g.n_hubs > 0) {
    g.hub_bm.alloc((size_t)g.n_hubs * words);
    SRW_HIP(hipMemsetAsync(g.hub_bm.p, 0, (size_t)g.n_hubs * words * 4, st));
    SRW_HIP(hipMemsetAsync(counter.p, 0, 8, st));
    hipLaunchKernelGGL(k_hub_fill, dim3(256 * 8), dim3(TPB), 0, st, hrows, g.ent.p, hsids, hub_slot.p, g.n_hubs, g.vmin, words,
                       g.hub_bm.p, counter.p);
    SRW_HIP(hipGetLastError());
    SRW_HIP(hipStreamSynchronize(st));
  }
  g.has_hub = true;
}

void generate_rmat_lines(srw_handle *h, int32_t scale, int64_t n_edges, uint32_t seed, bool weighted,
                         DevBuf<int32_t> &d_src, DevBuf<int32_t> &d_dst, DevBuf<float> &d_w) {
  if (scale < 1 || scale > 30) throw Error(SRW_ERR_INVALID, "rmat scale must be in [1, 30]");
  if (n_edges <= 0) throw Error(SRW_ERR_INVALID, "rmat n_edges must be > 0");
  d_src.alloc((size_t)n_edges); d_dst.alloc((size_t)n_edges);
  if (weighted) d_w.alloc((size_t)n_edges);
  hipLaunchKernelGGL(k_rmat, dim3(grid_for(n_edges)), dim3(TPB), 0, h->stream, scale, seed, (int64_t)0, n_edges, weighted ? 1 : 0,
                     d_src.p, d_dst.p, weighted ? d_w.p : nullptr);
  SRW_HIP(hipGetLastError());
}

void generate_rmat_block(srw_handle *h, int32_t scale, int64_t first_edge, int64_t n_edges, uint32_t seed, bool weighted,
                         int32_t *d_src, int32_t *d_dst, float *d_w) {
  hipLaunchKernelGGL(k_rmat, dim3(grid_for(n_edges)), dim3(TPB), 0, h->stream, scale, seed, first_edge, n_edges, weighted ? 1 : 0,
                     d_src, d_dst, weighted ? d_w : nullptr);
  SRW_HIP(hipGetLastError());
}

}  // namespace srw
