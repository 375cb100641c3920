// api.cpp — the C ABI (include/stellar_rw.h).  Everything here is glue: argument checks, host<->device
// staging, and translation of C++ exceptions into status codes + srw_last_error().
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "engine.h"

using namespace srw;

namespace {
std::mutex g_err_mu;
std::string g_create_error;

template <typename F>
int32_t guarded(srw_handle *h, F &&f) {
  try {
    if (h) SRW_HIP(hipSetDevice(h->cfg.device));
    f();
    return SRW_OK;
  } catch (const Error &e) {
    if (h) h->last_error = e.what();
    else { std::lock_guard<std::mutex> l(g_err_mu); g_create_error = e.what(); }
    return e.code;
  } catch (const std::bad_alloc &) {
    if (h) h->last_error = "host allocation failed";
    return SRW_ERR_NOMEM;
  } catch (const std::exception &e) {
    if (h) h->last_error = e.what();
    return SRW_ERR_INVALID;
  }
}

void need(bool ok, const char *msg) { if (!ok) throw Error(SRW_ERR_INVALID, msg); }
}  // namespace

// what srw_last_error(NULL) reports: failures of the entry points that have no handle yet (srw_create, srw_cluster_create)
void srw::set_create_error(const std::string &m) { std::lock_guard<std::mutex> l(g_err_mu); g_create_error = m; }

namespace {

void load_lines(srw_handle *h, const int32_t *src, const int32_t *dst, const float *w, const int32_t *pid,
                int64_t n, bool directed) {
  if (n == 0) {   // an empty edge list is a valid (empty) graph: zero vertices, zero paths, an empty part-00000
    h->g = Graph();
    h->g.loaded = true; h->g.symmetric = !directed;
    h->g.vmin = 0; h->g.vmax = -1; h->g.n_slots = 0;
    h->res.valid = false;
    return;
  }
  int32_t vmin = src[0], vmax = src[0];
  for (int64_t i = 0; i < n; ++i) {
    vmin = std::min(vmin, std::min(src[i], dst[i]));
    vmax = std::max(vmax, std::max(src[i], dst[i]));
  }
  const bool sparse = ids_are_sparse(h, 2 * n, vmin, vmax);
  if (!sparse) check_id_range(vmin, vmax);       // before any allocation that is proportional to the id range
  hipStream_t st = h->stream;
  if (h->cfg.world > 1 && !sparse && !getenv("SRW_BUILD_WHOLE")) {
    // a vertex-sharded handle uploads the lines block by block and keeps only the entries it owns (graph_build.hip)
    std::vector<int32_t> part_of;
    if (pid) {
      part_of.assign((size_t)((int64_t)vmax - vmin + 1), -1);
      for (int64_t i = 0; i < n; ++i) {                      // VCut: last put wins (GraphMap.scala:28-32)
        part_of[(size_t)((int64_t)dst[i] - vmin)] = pid[i];
        if (!directed) part_of[(size_t)((int64_t)src[i] - vmin)] = pid[i];
      }
    }
    DevBuf<int32_t> bs, bd; DevBuf<float> bw;
    build_graph_blocked(h, [&](int64_t i0, int64_t cnt, const int32_t *&s, const int32_t *&d, const float *&ww) {
      bs.ensure((size_t)cnt); bd.ensure((size_t)cnt);
      SRW_HIP(hipMemcpyAsync(bs.p, src + i0, (size_t)cnt * 4, hipMemcpyHostToDevice, st));
      SRW_HIP(hipMemcpyAsync(bd.p, dst + i0, (size_t)cnt * 4, hipMemcpyHostToDevice, st));
      if (w) { bw.ensure((size_t)cnt); SRW_HIP(hipMemcpyAsync(bw.p, w + i0, (size_t)cnt * 4, hipMemcpyHostToDevice, st)); }
      s = bs.p; d = bd.p; ww = w ? bw.p : nullptr;
    }, n, directed, vmin, vmax, part_of.empty() ? nullptr : part_of.data(), nullptr);
    h->g.part_of = std::move(part_of);
    return;
  }
  DevBuf<int32_t> d_src, d_dst; DevBuf<float> d_w;
  d_src.alloc((size_t)n); d_dst.alloc((size_t)n);
  SRW_HIP(hipMemcpyAsync(d_src.p, src, (size_t)n * 4, hipMemcpyHostToDevice, st));
  SRW_HIP(hipMemcpyAsync(d_dst.p, dst, (size_t)n * 4, hipMemcpyHostToDevice, st));
  if (w) { d_w.alloc((size_t)n); SRW_HIP(hipMemcpyAsync(d_w.p, w, (size_t)n * 4, hipMemcpyHostToDevice, st)); }
  SRW_HIP(hipStreamSynchronize(st));
  // VCut: vertexPartitionMap.put(dst, pId) for every adjacency entry, last put wins (GraphMap.scala:28-32).
  IdMap idmap;
  if (sparse) compact_ids(h, d_src.p, d_dst.p, n, vmin, vmax, idmap);     // [vmin, vmax] is the rank range from here on
  std::vector<int32_t> part_of;
  if (pid) {
    auto slot = [&](int32_t v) -> size_t {
      if (!idmap.compact) return (size_t)((int64_t)v - vmin);
      return (size_t)(std::lower_bound(idmap.h_orig_id.begin(), idmap.h_orig_id.end(), v) - idmap.h_orig_id.begin());
    };
    part_of.assign((size_t)((int64_t)vmax - vmin + 1), -1);
    for (int64_t i = 0; i < n; ++i) {
      part_of[slot(dst[i])] = pid[i];
      if (!directed) part_of[slot(src[i])] = pid[i];
    }
  }
  build_graph_from_device_lines(h, d_src.p, d_dst.p, w ? d_w.p : nullptr, n, directed, vmin, vmax,
                                part_of.empty() ? nullptr : part_of.data(), &idmap);
  h->g.part_of = std::move(part_of);
}
}  // namespace

extern "C" {

int32_t srw_create(const srw_config *cfg, srw_handle **out) {
  if (!out) return SRW_ERR_INVALID;
  *out = nullptr;
  srw_handle *h = nullptr;
  int32_t rc = guarded(nullptr, [&] {
    srw_config c{};
    if (cfg) c = *cfg;
    if (c.world <= 0) { c.world = 1; c.rank = 0; }
    need(c.rank >= 0 && c.rank < c.world, "rank must be in [0, world)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
      throw Error(SRW_ERR_HIP, std::string("no usable HIP device (this library has no CPU fallback): ") +
                                   (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    need(c.device >= 0 && c.device < ndev, "device ordinal out of range");
    SRW_HIP(hipSetDevice(c.device));
    h = new srw_handle();
    h->cfg = c;
    SRW_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    SRW_HIP(hipEventCreate(&h->ev0));
    SRW_HIP(hipEventCreate(&h->ev1));
    h->counters.alloc(1);
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c.device) == hipSuccess && ncu > 0) h->n_cus = ncu;
  });
  if (rc != SRW_OK) { delete h; return rc; }
  *out = h;
  return SRW_OK;
}

void srw_destroy(srw_handle *h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (int i = 0; i < 2; ++i) {
    if (h->stage_done[i]) (void)hipEventDestroy(h->stage_done[i]);
    if (h->kernel_done[i]) (void)hipEventDestroy(h->kernel_done[i]);
  }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  for (int i = 0; i < 2; ++i) { if (h->pin_paths[i]) (void)hipHostFree(h->pin_paths[i]); if (h->pin_lens[i]) (void)hipHostFree(h->pin_lens[i]); }
  for (int i = 0; i < srw_handle::PIN_RING; ++i) { if (h->pin_text[i]) (void)hipHostFree(h->pin_text[i]); if (h->pin_copied[i]) (void)hipEventDestroy(h->pin_copied[i]); }
  for (int i = 0; i < 2; ++i) if (h->pin_off[i]) (void)hipHostFree(h->pin_off[i]);
  if (h->shard_parked.init) {
    if (h->shard_parked.stream) (void)hipStreamSynchronize(h->shard_parked.stream);
    if (h->shard_parked.ev0) (void)hipEventDestroy(h->shard_parked.ev0);
    if (h->shard_parked.ev1) (void)hipEventDestroy(h->shard_parked.ev1);
    if (h->shard_parked.own_stream && h->shard_parked.stream) (void)hipStreamDestroy(h->shard_parked.stream);
  }
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char *srw_last_error(const srw_handle *h) {
  if (h) return h->last_error.c_str();
  std::lock_guard<std::mutex> l(g_err_mu);
  return g_create_error.c_str();
}

// srw_shard_select(h, 1) swaps the handle's stream, counters and cursors with the second population's: every entry point outside the
// super-step family would run on that population's stream with its cursors.  They refuse instead.
static void need_population0(const srw_handle *h, const char *what) {
  if (h->shard_population != 0)
    throw srw::Error(SRW_ERR_INVALID, std::string(what) + ": population 1 is selected on this handle (srw_shard_select(h, 0) first)");
}

int32_t srw_set_stream(srw_handle *h, void *hip_stream) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    if (h->own_stream && h->stream) { SRW_HIP(hipStreamSynchronize(h->stream)); SRW_HIP(hipStreamDestroy(h->stream)); }
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
  });
}

int32_t srw_plan_walks(srw_handle *h, int64_t num_walks) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    if (num_walks < 0) throw srw::Error(SRW_ERR_INVALID, "srw_plan_walks: num_walks < 0");
    h->planned_walks = num_walks;
  });
}

int32_t srw_load_edgelist(srw_handle *h, const char *path, int32_t directed, int32_t weighted, int32_t partitioned,
                          int32_t rdd_partitions) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need_population0(h, "srw_load_edgelist");
    need(path != nullptr, "path is null");
    const auto t0 = std::chrono::steady_clock::now();
    // two integer columns and nothing unusual: tokenized on the device (edgelist_device.hip); anything else: below
    if (!partitioned && !getenv("SRW_HOST_TOKENIZER") && load_edgelist_device(h, path, directed != 0, weighted != 0)) {
      if (getenv("SRW_TIMING"))
        fprintf(stderr, "[timing] loadGraph: device tokenizer + device CSR build %.1f ms\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      return;
    }
    ParsedLines L;
    parse_edgelist_file(path, weighted != 0, partitioned != 0, L);
    const auto t1 = std::chrono::steady_clock::now();
    if (partitioned) {
      // a missing / unparsable pId is Random.nextInt(rddPartitions) in the reference (VCutRandomWalk.scala:24-25,
      // unseeded); partition ids never change walk results, so a deterministic hash stands in.
      int32_t np = rdd_partitions > 0 ? rdd_partitions : 1;
      for (size_t i = 0; i < L.pid.size(); ++i)
        if (L.pid[i] < 0) L.pid[i] = (int32_t)(((uint32_t)L.src[i] * 0x9E3779B1u ^ (uint32_t)L.dst[i] * 0x85EBCA77u) % (uint32_t)np);
    }
    load_lines(h, L.src.data(), L.dst.data(), L.w.data(), partitioned ? L.pid.data() : nullptr, (int64_t)L.src.size(),
               directed != 0);
    if (getenv("SRW_TIMING"))
      fprintf(stderr, "[timing] loadGraph: tokenizer %.1f ms (%zu lines), upload + device CSR build %.1f ms\n",
              std::chrono::duration<double, std::milli>(t1 - t0).count(), L.src.size(),
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  });
}

int32_t srw_load_coo(srw_handle *h, const int32_t *src, const int32_t *dst, const float *w, const int32_t *pid,
                     int64_t n_lines, int32_t directed) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need_population0(h, "srw_load_coo");
    need(n_lines == 0 || (src && dst), "src/dst are null");
    load_lines(h, src, dst, w, pid, n_lines, directed != 0);
  });
}

int32_t srw_load_adjacency(srw_handle *h, const int32_t *vids, const int64_t *offs, int64_t n_rows, const int32_t *ids,
                           const float *w, const int32_t *pids) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need_population0(h, "srw_load_adjacency");
    need(vids && offs && (ids || offs[n_rows] == 0), "null argument");
    build_graph_from_host_rows(h, vids, offs, n_rows, ids, w);
    h->g.part_of.clear();
    if (pids) {
      h->g.part_of.assign((size_t)h->g.n_slots, -1);
      std::vector<char> seen((size_t)h->g.n_slots, 0);
      for (int64_t i = 0; i < n_rows; ++i) {
        size_t s = (size_t)h->g.slot_of_id(vids[i]);
        if (seen[s]) continue;   // a re-added vertex is ignored entirely (GraphMap.scala:37)
        seen[s] = 1;
        for (int64_t e = offs[i]; e < offs[i + 1]; ++e) h->g.part_of[(size_t)h->g.slot_of_id(ids[e])] = pids[e];
      }
    }
  });
}

int32_t srw_generate_rmat(srw_handle *h, int32_t scale, int64_t n_edges, uint32_t seed, int32_t weighted, int32_t directed) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need_population0(h, "srw_generate_rmat");
    DevBuf<int32_t> d_src, d_dst; DevBuf<float> d_w;
    if (h->cfg.world > 1 && !getenv("SRW_BUILD_WHOLE")) {
      // a shard generates the stream block by block (edge i is a pure function of (seed, i)) and keeps what it owns
      if (scale < 1 || scale > 30) throw Error(SRW_ERR_INVALID, "rmat scale must be in [1, 30]");
      if (n_edges <= 0) throw Error(SRW_ERR_INVALID, "rmat n_edges must be > 0");
      build_graph_blocked(h, [&](int64_t i0, int64_t cnt, const int32_t *&s, const int32_t *&d, const float *&ww) {
        d_src.ensure((size_t)cnt); d_dst.ensure((size_t)cnt);
        if (weighted) d_w.ensure((size_t)cnt);
        generate_rmat_block(h, scale, i0, cnt, seed, weighted != 0, d_src.p, d_dst.p, weighted ? d_w.p : nullptr);
        s = d_src.p; d = d_dst.p; ww = weighted ? d_w.p : nullptr;
      }, n_edges, directed != 0, 0, (int32_t)(((int64_t)1 << scale) - 1));
      h->g.part_of.clear();
      return;
    }
    generate_rmat_lines(h, scale, n_edges, seed, weighted != 0, d_src, d_dst, d_w);
    build_graph_from_device_lines(h, d_src.p, d_dst.p, weighted ? d_w.p : nullptr, n_edges, directed != 0, 0,
                                  (int32_t)(((int64_t)1 << scale) - 1));
    h->g.part_of.clear();
  });
}

int32_t srw_graph_stats(const srw_handle *h, int64_t *n_vertices, int64_t *n_entries) {
  if (!h || !h->g.loaded) return SRW_ERR_INVALID;
  if (n_vertices) *n_vertices = h->g.n_vertices;
  if (n_entries) *n_entries = h->g.n_entries_global;
  return SRW_OK;
}

int32_t srw_graph_vertices(const srw_handle *ch, int32_t *out) {
  srw_handle *h = const_cast<srw_handle *>(ch);
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need(h->g.loaded && (out || h->g.n_local_vertices == 0), "no graph / null out");
    if (h->g.n_local_vertices > 0)
      SRW_HIP(hipMemcpy(out, h->g.verts.p, (size_t)h->g.n_local_vertices * 4, hipMemcpyDeviceToHost));
    if (h->g.compact) for (int64_t i = 0; i < h->g.n_local_vertices; ++i) out[i] = h->g.id_of_slot(out[i]);
  });
}

int32_t srw_graph_neighbors(const srw_handle *ch, int32_t v, int32_t *ids, float *w, int64_t cap, int64_t *n) {
  srw_handle *h = const_cast<srw_handle *>(ch);
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need(h->g.loaded && n, "no graph / null n");
    const Graph &g = h->g;
    const int64_t s = g.slot_of_id(v);
    if (s < 0) { *n = -1; return; }
    Row r;
    SRW_HIP(hipMemcpy(&r, g.rows.p + s, sizeof(Row), hipMemcpyDeviceToHost));
    if (!(r.flags & ROW_PRESENT)) { *n = -1; return; }       // case None => null
    *n = r.deg;
    int64_t m = std::min<int64_t>(cap, r.deg);
    if (m > 0 && (ids || w)) {
      std::vector<Ent> tmp((size_t)m);
      SRW_HIP(hipMemcpy(tmp.data(), g.ent.p + r.off, (size_t)m * sizeof(Ent), hipMemcpyDeviceToHost));
      for (int64_t k = 0; k < m; ++k) { if (ids) ids[k] = g.compact ? g.id_of_slot(tmp[k].id) : tmp[k].id; if (w) w[k] = tmp[k].w; }
    }
  });
}

int32_t srw_graph_partition(const srw_handle *h, int32_t v, int32_t *pid, int32_t *known) {
  if (!h || !h->g.loaded || !known) return SRW_ERR_INVALID;
  *known = 0;
  const int64_t s = h->g.slot_of_id(v);
  if (h->g.part_of.empty() || s < 0 || h->g.part_of[(size_t)s] < 0) return SRW_OK;
  if (pid) *pid = h->g.part_of[(size_t)s];
  *known = 1;
  return SRW_OK;
}

int32_t srw_alias_row(srw_handle *h, int32_t v, float *prob, int32_t *alias, int64_t cap, int64_t *n, int32_t *regular) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need(h->g.loaded && n, "no graph / null n");
    build_membership(h);
    build_alias_tables(h);
    const Graph &g = h->g;
    const int64_t s = g.slot_of_id(v);
    if (s < 0) { *n = -1; return; }
    Row r;
    SRW_HIP(hipMemcpy(&r, g.rows.p + s, sizeof(Row), hipMemcpyDeviceToHost));
    if (!(r.flags & ROW_PRESENT)) { *n = -1; return; }
    *n = r.deg;
    if (regular) *regular = (r.flags & ROW_ALIAS_IRREGULAR) ? 0 : 1;
    int64_t m = std::min<int64_t>(cap, r.deg);
    if (m > 0) {
      std::vector<AEnt> tmp((size_t)m);
      SRW_HIP(hipMemcpy(tmp.data(), g.al.p + r.off, (size_t)m * sizeof(AEnt), hipMemcpyDeviceToHost));
      for (int64_t k = 0; k < m; ++k) { if (prob) prob[k] = tmp[k].prob; if (alias) alias[k] = tmp[k].alias; }
    }
  });
}

int32_t srw_walk(srw_handle *h, const srw_walk_params *params, srw_walk_stats *stats) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { need_population0(h, "srw_walk"); need(params != nullptr, "params is null"); run_walk(h, *params, stats); });
}

int32_t srw_walk_to_host(srw_handle *h, const srw_walk_params *params, int32_t *paths, int32_t *lens, srw_walk_stats *stats) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { need_population0(h, "srw_walk_to_host"); need(params && paths && lens, "null argument"); run_walk_to_host(h, *params, paths, lens, stats); });
}

int32_t srw_walk_and_save(srw_handle *h, const srw_walk_params *params, const char *output_dir, int32_t n_parts,
                          int32_t write_crc, srw_walk_stats *stats, int64_t *dead_ends_per_iteration) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need_population0(h, "srw_walk_and_save");
    need(params && output_dir, "null argument");
    run_walk_and_save(h, *params, output_dir, n_parts, write_crc != 0, stats, dead_ends_per_iteration);
  });
}

int32_t srw_host_alloc(size_t bytes, void **out) {
  if (!out) return SRW_ERR_INVALID;
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? SRW_ERR_NOMEM : SRW_ERR_HIP; }
  return SRW_OK;
}
void srw_host_free(void *p) { if (p) (void)hipHostFree(p); }

int32_t srw_fetch_paths(const srw_handle *ch, int32_t *paths, int32_t *lens) {
  srw_handle *h = const_cast<srw_handle *>(ch);
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need(h->res.valid, "no walk result");
    if (h->res.n_walkers == 0) return;
    if (paths) SRW_HIP(hipMemcpy(paths, h->res.paths.p, (size_t)h->res.n_walkers * h->res.stride * 4, hipMemcpyDeviceToHost));
    if (lens) SRW_HIP(hipMemcpy(lens, h->res.lens.p, (size_t)h->res.n_walkers * 4, hipMemcpyDeviceToHost));
  });
}

int32_t srw_device_paths(const srw_handle *h, void **d_paths, void **d_lens, int64_t *n_walkers, int32_t *stride) {
  if (!h || !h->res.valid) return SRW_ERR_INVALID;
  if (d_paths) *d_paths = h->res.paths.p;
  if (d_lens) *d_lens = h->res.lens.p;
  if (n_walkers) *n_walkers = h->res.n_walkers;
  if (stride) *stride = h->res.stride;
  return SRW_OK;
}

int32_t srw_write_paths(const srw_handle *ch, const char *output_dir, int32_t n_parts, int32_t write_crc) {
  srw_handle *h = const_cast<srw_handle *>(ch);
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need(h->res.valid && output_dir, "no walk result / null output_dir");
    // the result is in HBM: format it there (path_format.hip) unless told otherwise or memory is short
    if (!getenv("SRW_HOST_FORMATTER") && write_result_device(h, output_dir, n_parts, write_crc != 0)) return;
    std::vector<int32_t> paths((size_t)h->res.n_walkers * h->res.stride + 1), lens((size_t)h->res.n_walkers + 1);
    if (h->res.n_walkers > 0) {
      SRW_HIP(hipMemcpy(paths.data(), h->res.paths.p, (size_t)h->res.n_walkers * h->res.stride * 4, hipMemcpyDeviceToHost));
      SRW_HIP(hipMemcpy(lens.data(), h->res.lens.p, (size_t)h->res.n_walkers * 4, hipMemcpyDeviceToHost));
    }
    write_path_files(paths.data(), lens.data(), h->res.n_walkers, h->res.stride, output_dir, n_parts, write_crc != 0);
  });
}

int32_t srw_shard_capacity(const srw_handle *h, int64_t *n_local_vertices, int64_t *n_global_vertices) {
  if (!h || !h->g.loaded) return SRW_ERR_INVALID;
  if (n_local_vertices) *n_local_vertices = h->g.n_local_vertices;
  if (n_global_vertices) *n_global_vertices = h->g.n_vertices;
  return SRW_OK;
}

int32_t srw_shard_vertex_ranks(const srw_handle *h, int32_t *out) {
  if (!h || !h->g.loaded || !out) return SRW_ERR_INVALID;
  return guarded(const_cast<srw_handle *>(h), [&] {
    if (h->g.n_local_vertices > 0)
      SRW_HIP(hipMemcpy(out, h->g.vrank.p, (size_t)h->g.n_local_vertices * 4, hipMemcpyDeviceToHost));
  });
}

int32_t srw_shard_layout_for(const srw_handle *h, int32_t batch, double slack, srw_shard_layout *out) {
  if (!h || !out) return SRW_ERR_INVALID;
  return guarded(const_cast<srw_handle *>(h), [&] {
    need(h->g.loaded, "no graph loaded");
    shard_layout(h, batch, slack, out);
  });
}

int32_t srw_shard_select(srw_handle *h, int32_t population) {
  if (!h || population < 0 || population > 1) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    if (population == h->shard_population) return;
    auto &k = h->shard_parked;
    if (!k.init) {                                     // first use of the second population: its own stream and events
      k.init = true;                                   // (a flag, not "stream == null": a caller's stream may be the null stream — torch's default)
      SRW_HIP(hipSetDevice(h->cfg.device));
      SRW_HIP(hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking)); k.own_stream = true;
      SRW_HIP(hipEventCreate(&k.ev0)); SRW_HIP(hipEventCreate(&k.ev1));
    }
    std::swap(h->counters, k.counters); std::swap(h->walk_cursor, k.walk_cursor); std::swap(h->walk_todo, k.walk_todo);
    std::swap(h->shard_scratch, k.shard_scratch); std::swap(h->shard_blk, k.shard_blk); std::swap(h->shard_flag, k.shard_flag);
    std::swap(h->shard_cur, k.shard_cur); std::swap(h->shard_pt, k.shard_pt); std::swap(h->chain_buf, k.chain_buf); std::swap(h->chain_d, k.chain_d);
    std::swap(h->stream, k.stream); std::swap(h->own_stream, k.own_stream); std::swap(h->ev0, k.ev0); std::swap(h->ev1, k.ev1);
    h->shard_population = population;
  });
}

int32_t srw_shard_begin(srw_handle *h, const srw_walk_params *params, int32_t batch, const srw_shard_layout *layout,
                        void *d_recv, void *d_paths, void *d_lens) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need(params && layout && d_recv && d_paths && d_lens, "null argument");
    run_shard_begin(h, *params, batch, *layout, d_recv, (int32_t *)d_paths, (int32_t *)d_lens, (int64_t)params->walk_length + 2);
  });
}

int32_t srw_shard_superstep(srw_handle *h, const srw_walk_params *params, int32_t batch, int32_t step,
                            const srw_shard_layout *layout, const void *d_recv, void *const *dst_chunks, void *d_paths,
                            void *d_lens) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need(params && layout && d_recv && dst_chunks && d_paths && d_lens, "null argument");
    run_shard_superstep(h, *params, batch, step, *layout, d_recv, dst_chunks, (int32_t *)d_paths, (int32_t *)d_lens,
                        (int64_t)params->walk_length + 2);
  });
}

int32_t srw_shard_flush(srw_handle *h, const srw_walk_params *params, int32_t batch, const srw_shard_layout *layout,
                        const void *d_recv, void *d_paths, void *d_lens) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need(params && layout && d_recv && d_paths && d_lens, "null argument");
    run_shard_flush(h, *params, batch, *layout, d_recv, (int32_t *)d_paths, (int32_t *)d_lens, (int64_t)params->walk_length + 2);
  });
}

int32_t srw_shard_finish(srw_handle *h, srw_walk_stats *stats, int32_t *overflow) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { run_shard_finish(h, stats, overflow); });
}

int32_t srw_device_alloc(srw_handle *h, int64_t bytes, void **d_ptr) {
  if (!h || !d_ptr || bytes < 0) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    void *p = nullptr;
    const hipError_t e = hipMalloc(&p, (size_t)std::max<int64_t>(bytes, 1));
    if (e != hipSuccess) { (void)hipGetLastError(); throw Error(SRW_ERR_NOMEM, std::string("hipMalloc of ") + std::to_string(bytes) + " bytes: " + hipGetErrorString(e)); }
    *d_ptr = p;
  });
}
int32_t srw_device_free(srw_handle *h, void *d_ptr) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { if (d_ptr) SRW_HIP(hipFree(d_ptr)); });
}

int32_t srw_shard_rows_count(const srw_handle *h, int64_t *n_slots) {
  if (!h || !h->g.loaded || !n_slots) return SRW_ERR_INVALID;
  *n_slots = h->g.n_slots;
  return SRW_OK;
}
int32_t srw_shard_rows_export(srw_handle *h, void *d_rows, int64_t n_slots) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { need(d_rows, "null argument"); shard_rows_export(h, d_rows, n_slots); });
}
int32_t srw_shard_rows_merge(srw_handle *h, void *d_rows, const void *d_other_rows, int64_t n_slots) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { need(d_rows && d_other_rows, "null argument"); shard_rows_merge(h, d_rows, d_other_rows, n_slots); });
}
int32_t srw_shard_rows_commit(srw_handle *h, const void *d_rows_all, int64_t n_slots, int32_t *linked) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need(d_rows_all && linked, "null argument");
    *linked = shard_rows_commit(h, d_rows_all, n_slots) ? 1 : 0;
  });
}
int32_t srw_shard_rows_release(srw_handle *h) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { shard_rows_release(h); });
}

int32_t srw_w2v_fit(srw_handle *h, const int32_t *paths, const int32_t *lens, int64_t n, int64_t stride, const srw_w2v_params *params,
                    int32_t **vocab_ids, float **vectors, int64_t *n_vocab) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need_population0(h, "srw_w2v_fit");
    need(params && vocab_ids && vectors && n_vocab && (n == 0 || (paths && lens)) && stride >= 1 && n >= 0, "null or bad argument");
    std::vector<int32_t> ids; std::vector<float> vec;
    w2v_fit(h, paths, lens, n, stride, *params, ids, vec);
    *n_vocab = (int64_t)ids.size();
    *vocab_ids = (int32_t *)malloc(std::max<size_t>(ids.size() * 4, 1));
    *vectors = (float *)malloc(std::max<size_t>(vec.size() * 4, 1));
    if (!*vocab_ids || !*vectors) throw Error(SRW_ERR_NOMEM, "host allocation failed");
    memcpy(*vocab_ids, ids.data(), ids.size() * 4); memcpy(*vectors, vec.data(), vec.size() * 4);
  });
}

int32_t srw_w2v_fit_device(srw_handle *h, const void *d_paths, const void *d_lens, int64_t n, int64_t stride, const srw_w2v_params *params,
                           int32_t **vocab_ids, float **vectors, int64_t *n_vocab) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] {
    need_population0(h, "srw_w2v_fit_device");
    need(params && vocab_ids && vectors && n_vocab && stride >= 1 && n >= 0, "null or bad argument");
    const int32_t *dp = (const int32_t *)d_paths, *dl = (const int32_t *)d_lens;
    if (!dp && !dl) {                       // the handle's own last walk (srw_device_paths)
      if (!h->res.valid) throw Error(SRW_ERR_INVALID, "srw_w2v_fit_device: no walk result on this handle");
      dp = h->res.paths.p; dl = h->res.lens.p; n = h->res.n_walkers; stride = h->res.stride;
    }
    need(dp && dl, "null device pointer");
    std::vector<int32_t> ids; std::vector<float> vec;
    w2v_fit_device(h, dp, dl, n, stride, *params, ids, vec);
    *n_vocab = (int64_t)ids.size();
    *vocab_ids = (int32_t *)malloc(std::max<size_t>(ids.size() * 4, 1));
    *vectors = (float *)malloc(std::max<size_t>(vec.size() * 4, 1));
    if (!*vocab_ids || !*vectors) throw Error(SRW_ERR_NOMEM, "host allocation failed");
    memcpy(*vocab_ids, ids.data(), ids.size() * 4); memcpy(*vectors, vec.data(), vec.size() * 4);
  });
}

int32_t srw_w2v_huffman(const int64_t *counts, int64_t n_vocab, int32_t *code_len, uint8_t *codes, int32_t *points) {
  if (!counts || !code_len || !codes || !points || n_vocab < 0) return SRW_ERR_INVALID;
  try { w2v_huffman(counts, n_vocab, code_len, codes, points); return SRW_OK; }
  catch (const Error &e) { set_create_error(e.what()); return e.code; }
  catch (const std::exception &e) { set_create_error(e.what()); return SRW_ERR_INVALID; }
}

int32_t srw_w2v_save(const int32_t *vocab_ids, const float *vectors, int64_t n_vocab, int32_t dim, const char *output_dir, int32_t n_parts) {
  if (!output_dir || n_vocab < 0 || dim < 1 || (n_vocab > 0 && (!vocab_ids || !vectors))) return SRW_ERR_INVALID;
  try { write_vectors(vocab_ids, vectors, n_vocab, dim, output_dir, n_parts); return SRW_OK; }
  catch (const Error &e) { set_create_error(e.what()); return e.code; }
  catch (const std::exception &e) { set_create_error(e.what()); return SRW_ERR_INVALID; }
}

int32_t srw_w2v_save_words(const char *const *words, const float *vectors, int64_t n_vocab, int32_t dim, const char *output_dir, int32_t n_parts) {
  if (!output_dir || n_vocab < 0 || dim < 1 || (n_vocab > 0 && (!words || !vectors))) return SRW_ERR_INVALID;
  try { write_vectors_words(words, vectors, n_vocab, dim, output_dir, n_parts); return SRW_OK; }
  catch (const Error &e) { set_create_error(e.what()); return e.code; }
  catch (const std::exception &e) { set_create_error(e.what()); return SRW_ERR_INVALID; }
}

int32_t srw_probe_request_rate(srw_handle *h, int64_t table_bytes, double *reads_per_s, double *table_gib) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { need_population0(h, "srw_probe_request_rate"); need(reads_per_s, "null argument"); probe_request_rate(h, table_bytes, reads_per_s, table_gib); });
}

int32_t srw_result_scan_sums(srw_handle *h, int64_t *sums) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { need(sums, "null argument"); result_scan_sums(h, sums); });
}

int32_t srw_sample(srw_handle *h, const float *w, int64_t n, float r, int64_t *index) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { need(w && index, "null argument"); hook_sample(h, w, n, r, index); });
}

int32_t srw_second_order_weights(srw_handle *h, float p, float q, int32_t prev_id, const int32_t *prev_ids, int64_t n_prev,
                                 const int32_t *curr_ids, const float *curr_w, int64_t n, float *out_w) {
  if (!h) return SRW_ERR_INVALID;
  static const int32_t dummy = 0;
  return guarded(h, [&] {
    need(curr_ids && curr_w && out_w, "null argument");
    hook_second_order(h, p, q, prev_id, prev_ids ? prev_ids : &dummy, prev_ids ? n_prev : 0, curr_ids, curr_w, n, 0.f,
                      out_w, nullptr);
  });
}

int32_t srw_second_order_sample(srw_handle *h, float p, float q, int32_t prev_id, const int32_t *prev_ids, int64_t n_prev,
                                const int32_t *curr_ids, const float *curr_w, int64_t n, float r, int64_t *index) {
  if (!h) return SRW_ERR_INVALID;
  static const int32_t dummy = 0;
  return guarded(h, [&] {
    need(curr_ids && curr_w && index, "null argument");
    hook_second_order(h, p, q, prev_id, prev_ids ? prev_ids : &dummy, prev_ids ? n_prev : 0, curr_ids, curr_w, n, r,
                      nullptr, index);
  });
}

int32_t srw_rng_uniform(srw_handle *h, uint32_t seed, const uint32_t *iter, const uint32_t *src, const uint32_t *step,
                        int64_t n, float *out) {
  if (!h) return SRW_ERR_INVALID;
  return guarded(h, [&] { need(iter && src && step && out, "null argument"); hook_rng(h, seed, iter, src, step, n, out); });
}

int32_t srw_parse_edgelist(const char *path, int32_t weighted, int32_t partitioned, int32_t **src, int32_t **dst, float **w,
                           int32_t **pid, int64_t *n_lines, char *err, size_t errlen) {
  if (!path || !n_lines) return SRW_ERR_INVALID;
  try {
    ParsedLines L;
    parse_edgelist_file(path, weighted != 0, partitioned != 0, L);
    size_t n = L.src.size();
    auto dup = [n](const void *p, size_t el) { void *q = malloc(std::max<size_t>(n * el, 1)); memcpy(q, p, n * el); return q; };
    if (src) *src = (int32_t *)dup(L.src.data(), 4);
    if (dst) *dst = (int32_t *)dup(L.dst.data(), 4);
    if (w) *w = (float *)dup(L.w.data(), 4);
    if (pid) *pid = (int32_t *)dup(L.pid.data(), 4);
    *n_lines = (int64_t)n;
    return SRW_OK;
  } catch (const Error &e) {
    if (err && errlen) snprintf(err, errlen, "%s", e.what());
    return e.code;
  } catch (const std::exception &e) {
    if (err && errlen) snprintf(err, errlen, "%s", e.what());
    return SRW_ERR_INVALID;
  }
}

void srw_free(void *p) { free(p); }

int32_t srw_save_paths(const int32_t *paths, const int32_t *lens, int64_t n_walkers, int64_t stride, const char *output_dir,
                       int32_t n_parts, int32_t write_crc) {
  if (!paths || !lens || !output_dir || n_walkers < 0) return SRW_ERR_INVALID;
  try {
    write_path_files(paths, lens, n_walkers, stride, output_dir, n_parts, write_crc != 0);
    return SRW_OK;
  } catch (const Error &e) {
    std::lock_guard<std::mutex> l(g_err_mu);
    g_create_error = e.what();
    return e.code;
  } catch (const std::exception &) {
    return SRW_ERR_IO;
  }
}

#include "version.h"      // build/version.h: SRW_GIT_REV (Makefile)
const char *srw_version(void) { return "stellar_rw gfx950 r6 " SRW_GIT_REV; }

}  // extern "C"
