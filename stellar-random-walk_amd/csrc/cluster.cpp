// cluster.cpp — the vertex-sharded walk inside ONE process over several devices (include/stellar_rw.h: srw_cluster_*).
//
// Replaces the Spark super-step loop with its shuffle (M/algorithm/RandomWalk.scala:91-162, transferWalkersToTheirPartitions
// :186-192) for callers that are a single process: the stellar-rw CLI (--gpus N) and the JNI host.  One sharded srw_handle
// per device (owner(v) = mix32(v) mod world, RandomWalk.scala:16, or the VCut partition ids); peer access enabled;
// every super-step each shard's bucket kernel stores chunk (me -> d) straight into device d's receive buffer over xGMI
// (fully connected: every pair has its own link), and the shards' streams are ordered by one event per shard per
// super-step — no collective library and no host synchronisation inside a walk iteration.  Receive buffers are double
// buffered by super-step parity: a peer may run at most one super-step ahead of the slowest shard.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "engine.h"

using namespace srw;

struct srw_cluster {
  std::vector<srw_handle *> sh;
  std::vector<int32_t> dev;
  std::string last_error;
  std::vector<DevBuf<char>> recv[2];
  std::vector<DevBuf<int32_t>> paths, lens;
  std::vector<hipEvent_t> ev;
  std::vector<std::vector<int32_t>> vrank;        // host copy: global rank of each local vertex
  struct Batch { int32_t it0, n; };
  std::vector<Batch> batches;                      // of the last walk
  int32_t walk_length = 0, num_walks = 0;
  bool valid = false;
  int rows_linked = -1;                            // -1: not tried for this graph; 0: some shard could not; 1: every shard linked
  int32_t world() const { return (int32_t)sh.size(); }
};

namespace {
template <typename F>
int32_t cguard(srw_cluster *c, F &&f) {
  try { f(); return SRW_OK; }
  catch (const Error &e) { if (c) c->last_error = e.what(); return e.code; }
  catch (const std::bad_alloc &) { if (c) c->last_error = "host allocation failed"; return SRW_ERR_NOMEM; }
  catch (const std::exception &e) { if (c) c->last_error = e.what(); return SRW_ERR_INVALID; }
}
void ck(srw_cluster *c, int r, int32_t rc) {
  if (rc != SRW_OK) throw Error(rc, std::string("shard ") + std::to_string(r) + ": " + srw_last_error(c->sh[(size_t)r]));
}
void each(srw_cluster *c, const std::function<int32_t(int, srw_handle *)> &f) {
  for (int r = 0; r < c->world(); ++r) ck(c, r, f(r, c->sh[(size_t)r]));
}
}  // namespace

namespace {
// Row descriptors across the shards (srw_shard_rows_*): every shard exports its table, takes the element-wise maximum
// with the others' (peer reads), and derives its linked first-order records.  All or nothing.
void link_rows(srw_cluster *c) {
  const int32_t world = c->world();
  c->rows_linked = 0;
  if (getenv("SRW_SHARD_NO_LINKS")) return;
  int64_t n_slots = 0;
  ck(c, 0, srw_shard_rows_count(c->sh[0], &n_slots));
  if (n_slots <= 0) return;
  std::vector<DevBuf<char>> own((size_t)world), acc((size_t)world);
  for (int r = 0; r < world; ++r) {
    SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
    own[(size_t)r].alloc((size_t)n_slots * 16);
    ck(c, r, srw_shard_rows_export(c->sh[(size_t)r], own[(size_t)r].p, n_slots));
  }
  bool all = true;
  try {
  for (int r = 0; r < world; ++r) {
    SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
    acc[(size_t)r].alloc((size_t)n_slots * 16);
    SRW_HIP(hipMemcpyAsync(acc[(size_t)r].p, own[(size_t)r].p, (size_t)n_slots * 16, hipMemcpyDeviceToDevice, c->sh[(size_t)r]->stream));   // (stream order with the merges)
    for (int o = 0; o < world; ++o)
      if (o != r) ck(c, r, srw_shard_rows_merge(c->sh[(size_t)r], acc[(size_t)r].p, own[(size_t)o].p, n_slots));
    int32_t linked = 0;
    ck(c, r, srw_shard_rows_commit(c->sh[(size_t)r], acc[(size_t)r].p, n_slots, &linked));
    acc[(size_t)r].release();
    all = all && linked != 0;
  }
  } catch (...) {
    // a shard that already committed must not keep linked records while the others do not: linked and unlinked walkers
    // interpret srw_walker.prev / kind differently (all or nothing)
    for (int r = 0; r < world; ++r) (void)srw_shard_rows_release(c->sh[(size_t)r]);
    c->rows_linked = 0;
    throw;
  }
  for (int r = 0; r < world; ++r) { SRW_HIP(hipSetDevice(c->dev[(size_t)r])); own[(size_t)r].release(); }
  if (!all) { for (int r = 0; r < world; ++r) ck(c, r, srw_shard_rows_release(c->sh[(size_t)r])); return; }
  c->rows_linked = 1;
}
}  // namespace

extern "C" {

int32_t srw_cluster_create(const int32_t *devices, int32_t n_devices, int32_t flags, srw_cluster **out) {
  if (!devices || n_devices < 1 || n_devices > 64 || !out) return SRW_ERR_INVALID;
  auto *c = new srw_cluster();
  const int32_t rc = cguard(c, [&] {
    for (int r = 0; r < n_devices; ++r) {
      srw_config cfg; cfg.device = devices[r]; cfg.rank = r; cfg.world = n_devices; cfg.flags = flags;
      srw_handle *h = nullptr;
      const int32_t rc1 = srw_create(&cfg, &h);
      if (rc1 != SRW_OK) throw Error(rc1, std::string("srw_create on device ") + std::to_string(devices[r]) + ": " + srw_last_error(nullptr));
      c->sh.push_back(h); c->dev.push_back(devices[r]);
    }
    for (int a = 0; a < n_devices; ++a)
      for (int b = 0; b < n_devices; ++b) {
        if (devices[a] == devices[b]) continue;
        SRW_HIP(hipSetDevice(devices[a]));
        int can = 0;
        SRW_HIP(hipDeviceCanAccessPeer(&can, devices[a], devices[b]));
        if (!can) throw Error(SRW_ERR_HIP, "devices " + std::to_string(devices[a]) + " and " + std::to_string(devices[b]) + " have no peer access");
        const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) SRW_HIP(e);
        (void)hipGetLastError();
      }
    for (int a = 0; a < n_devices; ++a) {
      int same = 0;
      for (int b = 0; b < n_devices; ++b) same += devices[a] == devices[b];
      c->sh[(size_t)a]->dev_share = same;
    }
    c->recv[0].resize((size_t)n_devices); c->recv[1].resize((size_t)n_devices);
    c->paths.resize((size_t)n_devices); c->lens.resize((size_t)n_devices); c->vrank.resize((size_t)n_devices);
    c->ev.assign((size_t)n_devices, nullptr);
    for (int r = 0; r < n_devices; ++r) {
      SRW_HIP(hipSetDevice(devices[r]));
      SRW_HIP(hipEventCreateWithFlags(&c->ev[(size_t)r], hipEventDisableTiming));
    }
  });
  if (rc != SRW_OK) {      // no cluster to ask: the message goes where srw_last_error(NULL) finds it
    if (c->last_error.find("srw_create on device") != 0) set_create_error("srw_cluster_create: " + c->last_error);
    else set_create_error(c->last_error);
    srw_cluster_destroy(c);
    return rc;
  }
  *out = c;
  return SRW_OK;
}

void srw_cluster_destroy(srw_cluster *c) {
  if (!c) return;
  for (size_t r = 0; r < c->sh.size(); ++r) {
    (void)hipSetDevice(c->dev[r]);
    if (c->sh[r] && c->sh[r]->stream) (void)hipStreamSynchronize(c->sh[r]->stream);
    if (r < c->ev.size() && c->ev[r]) (void)hipEventDestroy(c->ev[r]);
    for (int b = 0; b < 2; ++b) if (r < c->recv[b].size()) c->recv[b][r].release();
    if (r < c->paths.size()) { c->paths[r].release(); c->lens[r].release(); }
  }
  for (srw_handle *h : c->sh) srw_destroy(h);
  delete c;
}

const char *srw_cluster_last_error(const srw_cluster *c) { return c ? c->last_error.c_str() : "null cluster"; }
srw_handle *srw_cluster_shard(srw_cluster *c, int32_t rank) { return (c && rank >= 0 && rank < c->world()) ? c->sh[(size_t)rank] : nullptr; }

int32_t srw_cluster_load_edgelist(srw_cluster *c, const char *path, int32_t directed, int32_t weighted, int32_t partitioned,
                                  int32_t rdd_partitions) {
  if (!c) return SRW_ERR_INVALID;
  return cguard(c, [&] {
    c->valid = false; c->rows_linked = -1;
    each(c, [&](int, srw_handle *h) { return srw_load_edgelist(h, path, directed, weighted, partitioned, rdd_partitions); });
    for (auto &v : c->vrank) v.clear();
  });
}
int32_t srw_cluster_load_coo(srw_cluster *c, const int32_t *src, const int32_t *dst, const float *w, const int32_t *pid,
                             int64_t n_lines, int32_t directed) {
  if (!c) return SRW_ERR_INVALID;
  return cguard(c, [&] {
    c->valid = false; c->rows_linked = -1;
    each(c, [&](int, srw_handle *h) { return srw_load_coo(h, src, dst, w, pid, n_lines, directed); });
    for (auto &v : c->vrank) v.clear();
  });
}
int32_t srw_cluster_generate_rmat(srw_cluster *c, int32_t scale, int64_t n_edges, uint32_t seed, int32_t weighted, int32_t directed) {
  if (!c) return SRW_ERR_INVALID;
  return cguard(c, [&] {
    c->valid = false; c->rows_linked = -1;
    each(c, [&](int, srw_handle *h) { return srw_generate_rmat(h, scale, n_edges, seed, weighted, directed); });
    for (auto &v : c->vrank) v.clear();
  });
}
int32_t srw_cluster_graph_stats(const srw_cluster *c, int64_t *n_vertices, int64_t *n_entries) {
  if (!c || c->sh.empty()) return SRW_ERR_INVALID;
  return srw_graph_stats(c->sh[0], n_vertices, n_entries);   // every shard reports the whole graph's counts
}

int32_t srw_cluster_walk(srw_cluster *c, const srw_walk_params *params, int32_t batch, srw_walk_stats *stats) {
  if (!c || !params) return SRW_ERR_INVALID;
  return cguard(c, [&] {
    const srw_walk_params P0 = *params;
    const int32_t world = c->world();
    if (P0.num_walks < 0 || P0.walk_length < 0) throw Error(SRW_ERR_INVALID, "bad walk parameters");
    if (P0.sampler != SRW_SAMPLER_REFERENCE) throw Error(SRW_ERR_INVALID, "the vertex-sharded walk runs Mode R");
    const int64_t stride = (int64_t)P0.walk_length + 2;
    std::vector<int64_t> n_local((size_t)world, 0);
    int64_t n_global = 0;
    for (int r = 0; r < world; ++r) ck(c, r, srw_shard_capacity(c->sh[(size_t)r], &n_local[(size_t)r], &n_global));
    c->valid = false; c->batches.clear();
    c->walk_length = P0.walk_length; c->num_walks = P0.num_walks;
    srw_walk_stats tot; memset(&tot, 0, sizeof tot);
    tot.kernel_kind = (P0.p == 1.0f && P0.q == 1.0f && !(P0.flags & SRW_WALK_FORCE_GENERAL)) ? 1 : 2;
    if (n_global == 0 || P0.num_walks == 0) { if (stats) *stats = tot; c->valid = true; return; }
    if (tot.kernel_kind == 1 && P0.rng_mode == SRW_RNG_PHILOX && c->rows_linked < 0) link_rows(c);
    if (batch <= 0) {   // as many iterations per population as keep a shard's chunk buffers under ~2 GiB
      const int64_t per_iter = std::max<int64_t>(1, n_global / world * 30);      // 24 B of chunk space per resident walker x slack
      batch = (int32_t)std::max<int64_t>(1, std::min<int64_t>(P0.num_walks, ((int64_t)2 << 30) / per_iter));
    }
    batch = std::min(batch, P0.num_walks);
    for (int r = 0; r < world; ++r) {
      SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
      c->paths[(size_t)r].ensure((size_t)std::max<int64_t>(1, (int64_t)P0.num_walks * n_local[(size_t)r] * stride));
      c->lens[(size_t)r].ensure((size_t)std::max<int64_t>(1, (int64_t)P0.num_walks * n_local[(size_t)r]));
      if (c->vrank[(size_t)r].size() != (size_t)n_local[(size_t)r]) {
        c->vrank[(size_t)r].assign((size_t)n_local[(size_t)r], 0);
        if (n_local[(size_t)r]) ck(c, r, srw_shard_vertex_ranks(c->sh[(size_t)r], c->vrank[(size_t)r].data()));
      }
    }
    const auto t0 = std::chrono::steady_clock::now();
    double slack = 1.25;
    for (int32_t it0 = 0; it0 < P0.num_walks;) {
      const int32_t B = std::min(batch, P0.num_walks - it0);
      srw_walk_params P = P0; P.first_walk = P0.first_walk + it0; P.num_walks = B;
      srw_shard_layout lay;
      ck(c, 0, srw_shard_layout_for(c->sh[0], B, slack, &lay));
      const size_t buf_bytes = (size_t)world * (size_t)lay.chunk_bytes;
      for (int r = 0; r < world; ++r) {
        SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
        for (int b = 0; b < 2; ++b) c->recv[b][(size_t)r].ensure(buf_bytes);
      }
      auto pth = [&](int r) { return c->paths[(size_t)r].p + (int64_t)it0 * n_local[(size_t)r] * stride; };
      auto len = [&](int r) { return c->lens[(size_t)r].p + (int64_t)it0 * n_local[(size_t)r]; };
      for (int r = 0; r < world; ++r) ck(c, r, srw_shard_begin(c->sh[(size_t)r], &P, B, &lay, c->recv[0][(size_t)r].p, pth(r), len(r)));
      std::vector<void *> dst((size_t)world);
      for (int32_t step = 1; step <= P.walk_length + 1; ++step) {
        const int cur = (step - 1) & 1, nxt = cur ^ 1;
        for (int r = 0; r < world; ++r) {
          for (int d = 0; d < world; ++d) dst[(size_t)d] = c->recv[nxt][(size_t)d].p + (size_t)r * (size_t)lay.chunk_bytes;
          ck(c, r, srw_shard_superstep(c->sh[(size_t)r], &P, B, step, &lay, c->recv[cur][(size_t)r].p, dst.data(), pth(r), len(r)));
          SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
          SRW_HIP(hipEventRecord(c->ev[(size_t)r], c->sh[(size_t)r]->stream));
        }
        if (world > 1)
          for (int r = 0; r < world; ++r) {
            SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
            for (int o = 0; o < world; ++o)
              if (o != r) SRW_HIP(hipStreamWaitEvent(c->sh[(size_t)r]->stream, c->ev[(size_t)o], 0));
          }
      }
      const int fin = (P.walk_length + 1) & 1;
      for (int r = 0; r < world; ++r) ck(c, r, srw_shard_flush(c->sh[(size_t)r], &P, B, &lay, c->recv[fin][(size_t)r].p, pth(r), len(r)));
      bool overflow = false;
      srw_walk_stats bt; memset(&bt, 0, sizeof bt);
      for (int r = 0; r < world; ++r) {
        srw_walk_stats s; int32_t of = 0;
        ck(c, r, srw_shard_finish(c->sh[(size_t)r], &s, &of));
        overflow |= of != 0;
        bt.n_steps += s.n_steps; bt.dead_ends += s.dead_ends; bt.sum_deg_curr += s.sum_deg_curr; bt.sum_deg_prev += s.sum_deg_prev;
        bt.ent_reads += s.ent_reads; bt.fallbacks += s.fallbacks; bt.trials += s.trials;
        for (int i = 0; i < 12; ++i) bt.strategy_steps[i] += s.strategy_steps[i];
        bt.edge_tables += s.edge_tables; bt.edge_table_bytes += s.edge_table_bytes;      // per shard: summed = the whole graph's set
      }
      if (overflow) {                 // a chunk was too small for this graph's skew: same batch again with more room
        if (getenv("SRW_TIMING")) fprintf(stderr, "[cluster] chunk overflow at slack %.2f (batch %d, iteration %d): retrying\n", slack, B, it0);
        slack *= 2.0;
        if (slack > 64.0) throw Error(SRW_ERR_NOMEM, "vertex-sharded walk: chunk overflow persists at 64x slack");
        continue;
      }
      tot.n_steps += bt.n_steps; tot.dead_ends += bt.dead_ends; tot.sum_deg_curr += bt.sum_deg_curr; tot.sum_deg_prev += bt.sum_deg_prev;
      tot.ent_reads += bt.ent_reads; tot.fallbacks += bt.fallbacks; tot.trials += bt.trials;
      for (int i = 0; i < 12; ++i) tot.strategy_steps[i] += bt.strategy_steps[i];
      tot.edge_tables = bt.edge_tables; tot.edge_table_bytes = bt.edge_table_bytes;
      c->batches.push_back({it0, B});
      it0 += B;
    }
    tot.kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    tot.n_walkers = (int64_t)P0.num_walks * n_global;
    tot.record_bytes = 16;
    if (stats) *stats = tot;
    c->valid = true;
  });
}

int32_t srw_cluster_fetch_paths(srw_cluster *c, int32_t *paths, int32_t *lens) {
  if (!c) return SRW_ERR_INVALID;
  return cguard(c, [&] {
    if (!c->valid) throw Error(SRW_ERR_INVALID, "no walk result");
    const int32_t world = c->world();
    const int64_t stride = (int64_t)c->walk_length + 2;
    int64_t n_global = 0, nl = 0;
    ck(c, 0, srw_shard_capacity(c->sh[0], &nl, &n_global));
    std::vector<int32_t> hp, hl;
    for (int r = 0; r < world; ++r) {
      int64_t n_local = 0, ng = 0;
      ck(c, r, srw_shard_capacity(c->sh[(size_t)r], &n_local, &ng));
      const int64_t rows = (int64_t)c->num_walks * n_local;
      if (rows == 0) continue;
      SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
      hl.resize((size_t)rows);
      SRW_HIP(hipMemcpy(hl.data(), c->lens[(size_t)r].p, (size_t)rows * 4, hipMemcpyDeviceToHost));
      if (paths) { hp.resize((size_t)(rows * stride)); SRW_HIP(hipMemcpy(hp.data(), c->paths[(size_t)r].p, (size_t)(rows * stride) * 4, hipMemcpyDeviceToHost)); }
      const std::vector<int32_t> &vr = c->vrank[(size_t)r];
      for (const auto &b : c->batches) {
        const int64_t base = (int64_t)b.it0 * n_local;
        for (int64_t lw = 0; lw < (int64_t)b.n * n_local; ++lw) {
          const int64_t it = b.it0 + lw % b.n, lv = lw / b.n;
          const int64_t canon = it * n_global + vr[(size_t)lv];
          if (lens) lens[canon] = hl[(size_t)(base + lw)];
          if (paths) memcpy(paths + canon * stride, hp.data() + (base + lw) * stride, (size_t)stride * 4);
        }
      }
    }
  });
}

int32_t srw_cluster_walk_and_save(srw_cluster *c, const srw_walk_params *params, const char *output_dir, int32_t n_parts,
                                  int32_t write_crc, srw_walk_stats *stats) {
  if (!c || !params || !output_dir) return SRW_ERR_INVALID;
  return cguard(c, [&] {
    int64_t nv = 0, ne = 0;
    ck(c, 0, srw_graph_stats(c->sh[0], &nv, &ne));
    const int64_t n_walkers = (int64_t)params->num_walks * nv, stride = (int64_t)params->walk_length + 2;
    PathWriter writer(output_dir, n_parts, n_walkers, write_crc != 0);       // fails first if <output>/path exists
    int32_t rc = srw_cluster_walk(c, params, 0, stats);
    if (rc != SRW_OK) throw Error(rc, c->last_error);
    std::vector<int32_t> paths((size_t)std::max<int64_t>(1, n_walkers * stride)), lens((size_t)std::max<int64_t>(1, n_walkers));
    rc = srw_cluster_fetch_paths(c, paths.data(), lens.data());
    if (rc != SRW_OK) throw Error(rc, c->last_error);
    if (n_walkers > 0) writer.append(paths.data(), lens.data(), n_walkers, stride);
    writer.close();
  });
}

}  // extern "C"
