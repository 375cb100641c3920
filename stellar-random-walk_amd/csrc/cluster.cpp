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
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"

using namespace srw;

struct srw_cluster {
  std::vector<srw_handle *> sh;
  std::vector<int32_t> dev;
  std::string last_error;
  std::vector<DevBuf<char>> recv[2][2];            // [population][super-step parity][shard]
  std::vector<DevBuf<int32_t>> paths, lens;
  std::vector<hipEvent_t> ev[2][2];                // [population][parity][shard]
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
// One host thread per DEVICE: a load (CSR build: seconds per shard) runs on all devices at once instead of shard after shard
// (round 2: one thread drove everything — loads 8x the single-shard time).  Handles that share a device (virtual shards: tests,
// one-GPU measurements) are loaded one after the other by that device's thread: their builds size themselves from hipMemGetInfo
// (row filters, edge hash, hub bitmaps, the blocked build's present[] and sort buffers), and concurrent builds on one device
// would race each other's decisions and add up their peaks.  SRW_CLUSTER_SERIAL=1: shard after shard (debugging).
void each(srw_cluster *c, const std::function<int32_t(int, srw_handle *)> &f) {
  const int world = c->world();
  if (world == 1 || getenv("SRW_CLUSTER_SERIAL")) {
    for (int r = 0; r < world; ++r) ck(c, r, f(r, c->sh[(size_t)r]));
    return;
  }
  std::vector<int32_t> rc((size_t)world, SRW_OK);
  std::vector<std::vector<int>> by_dev;                 // shards grouped by device, in rank order
  for (int r = 0; r < world; ++r) {
    size_t gidx = 0;
    for (; gidx < by_dev.size(); ++gidx) if (c->dev[(size_t)by_dev[gidx][0]] == c->dev[(size_t)r]) break;
    if (gidx == by_dev.size()) by_dev.emplace_back();
    by_dev[gidx].push_back(r);
  }
  std::vector<std::thread> th;
  for (const auto &grp : by_dev)
    th.emplace_back([&, grp] {
      (void)hipSetDevice(c->dev[(size_t)grp[0]]);
      for (int r : grp) rc[(size_t)r] = f(r, c->sh[(size_t)r]);
    });
  for (auto &t : th) t.join();
  for (int r = 0; r < world; ++r) ck(c, r, rc[(size_t)r]);
}
// sense-reversing spin barrier for the per-device threads of a walk (a super-step is tens of microseconds of host work)
struct SpinBarrier {
  explicit SpinBarrier(int n) : n_(n) {}
  void wait() {
    const int gen = gen_.load(std::memory_order_acquire);
    if (count_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) { count_.store(0, std::memory_order_relaxed); gen_.store(gen + 1, std::memory_order_release); }
    else while (gen_.load(std::memory_order_acquire) == gen) std::this_thread::yield();
  }
  int n_; std::atomic<int> count_{0}, gen_{0};
};
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
    for (int pp = 0; pp < 2; ++pp) for (int b = 0; b < 2; ++b) c->recv[pp][b].resize((size_t)n_devices);
    c->paths.resize((size_t)n_devices); c->lens.resize((size_t)n_devices); c->vrank.resize((size_t)n_devices);
    for (int pp = 0; pp < 2; ++pp) for (int b = 0; b < 2; ++b) c->ev[pp][b].assign((size_t)n_devices, nullptr);
    for (int r = 0; r < n_devices; ++r) {
      SRW_HIP(hipSetDevice(devices[r]));
      for (int pp = 0; pp < 2; ++pp) for (int b = 0; b < 2; ++b) SRW_HIP(hipEventCreateWithFlags(&c->ev[pp][b][(size_t)r], hipEventDisableTiming));
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
    for (int pp = 0; pp < 2; ++pp) for (int b = 0; b < 2; ++b) if (r < c->ev[pp][b].size() && c->ev[pp][b][r]) (void)hipEventDestroy(c->ev[pp][b][r]);
    for (int pp = 0; pp < 2; ++pp) for (int b = 0; b < 2; ++b) if (r < c->recv[pp][b].size()) c->recv[pp][b][r].release();
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
    if (c->world() == 1) {
      ck(c, 0, srw_load_edgelist(c->sh[0], path, directed, weighted, partitioned, rdd_partitions));
    } else {
      // The file is tokenized ONCE (host tokenizer, all cores) and every shard builds its rows from the shared lines — each shard
      // tokenizing its own copy held world x 12-16 bytes per line on the host at the same time.
      int32_t *src = nullptr, *dst = nullptr, *pid = nullptr; float *w = nullptr; int64_t n = 0;
      char err[512]; err[0] = 0;
      const int32_t rc = srw_parse_edgelist(path, weighted, partitioned, &src, &dst, &w, partitioned ? &pid : nullptr, &n, err, sizeof(err));
      if (rc != SRW_OK) throw Error(rc, err[0] ? err : "cannot parse the edge list");
      struct Free { int32_t *a, *b, *c; float *d; ~Free() { srw_free(a); srw_free(b); srw_free(c); srw_free(d); } } fr{src, dst, pid, w};
      if (partitioned && pid) {
        // a missing / unparsable pId: as srw_load_edgelist (the reference draws Random.nextInt(rddPartitions), VCutRandomWalk.scala:24-25)
        const int32_t np = rdd_partitions > 0 ? rdd_partitions : 1;
        for (int64_t i = 0; i < n; ++i)
          if (pid[i] < 0) pid[i] = (int32_t)(((uint32_t)src[i] * 0x9E3779B1u ^ (uint32_t)dst[i] * 0x85EBCA77u) % (uint32_t)np);
      }
      each(c, [&](int, srw_handle *h) { return srw_load_coo(h, src, dst, w, partitioned ? pid : nullptr, n, directed); });
    }
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

}  // extern "C"

namespace {
// One or two walker populations (each: P.num_walks = B walk iterations starting at P.first_walk) on every shard: begin, walk_length + 1
// super-steps, flush, finish.  pth / len: per shard, device buffers of B * n_local rows.  One host thread per device; per super-step
// every thread enqueues its shard's kernels — population after population, each on its own stream (srw_shard_select) —, records
// their events, meets the others at ONE barrier (so that every event of this super-step IS recorded) and makes each population's
// stream wait for the other shards' events of that population; events and receive buffers alternate by super-step parity.  With two
// populations a device runs B's kernels while A's chunks are still arriving from its peers (and, on one device, B fills the tails of
// A's kernels): the shuffle / count rhythm of RandomWalk.scala:91-162 without its serialisation.  No host synchronisation with the
// devices inside.  Pop::overflow: a chunk of that population was too small.
struct Pop { srw_walk_params P; int32_t B; std::vector<int32_t *> pth, len; srw_walk_stats bt; bool overflow; };
void run_populations(srw_cluster *c, std::vector<Pop> &pops, double slack) {
  const int32_t world = c->world();
  const int np = (int)pops.size();
  const int32_t L = pops[0].P.walk_length;
  std::vector<srw_shard_layout> lay((size_t)np);
  for (int q = 0; q < np; ++q) {
    ck(c, 0, srw_shard_layout_for(c->sh[0], pops[(size_t)q].B, slack, &lay[(size_t)q]));
    const size_t buf_bytes = (size_t)world * (size_t)lay[(size_t)q].chunk_bytes;
    for (int r = 0; r < world; ++r) {
      SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
      for (int b = 0; b < 2; ++b) c->recv[q][b][(size_t)r].ensure(buf_bytes);
    }
  }
  SpinBarrier bar(world);
  std::atomic<bool> failed{false};
  std::vector<std::string> err((size_t)world);
  std::vector<int32_t> err_code((size_t)world, SRW_OK);
  std::vector<std::vector<srw_walk_stats>> st((size_t)np, std::vector<srw_walk_stats>((size_t)world));
  std::vector<std::vector<int32_t>> of((size_t)np, std::vector<int32_t>((size_t)world, 0));
  auto body = [&](int r) {
    srw_handle *h = c->sh[(size_t)r];
    auto guard = [&](auto &&f) {                      // a failing shard keeps meeting the others at the barriers
      if (failed.load(std::memory_order_acquire)) return;
      try { f(); }
      catch (const Error &e) { err[(size_t)r] = e.what(); err_code[(size_t)r] = e.code; failed.store(true, std::memory_order_release); }
      catch (const std::exception &e) { err[(size_t)r] = e.what(); err_code[(size_t)r] = SRW_ERR_INVALID; failed.store(true, std::memory_order_release); }
    };
    auto sel = [&](int q) { if (np > 1) ck(c, r, srw_shard_select(h, q)); };
    guard([&] {
      SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
      for (int q = 0; q < np; ++q) {
        sel(q);
        ck(c, r, srw_shard_begin(h, &pops[(size_t)q].P, pops[(size_t)q].B, &lay[(size_t)q], c->recv[q][0][(size_t)r].p, pops[(size_t)q].pth[(size_t)r], pops[(size_t)q].len[(size_t)r]));
      }
    });
    std::vector<void *> dst((size_t)world);
    for (int32_t step = 1; step <= L + 1; ++step) {
      const int cur = (step - 1) & 1, nxt = cur ^ 1;
      guard([&] {
        for (int q = 0; q < np; ++q) {
          sel(q);
          for (int d = 0; d < world; ++d) dst[(size_t)d] = c->recv[q][nxt][(size_t)d].p + (size_t)r * (size_t)lay[(size_t)q].chunk_bytes;
          ck(c, r, srw_shard_superstep(h, &pops[(size_t)q].P, pops[(size_t)q].B, step, &lay[(size_t)q], c->recv[q][cur][(size_t)r].p, dst.data(),
                                       pops[(size_t)q].pth[(size_t)r], pops[(size_t)q].len[(size_t)r]));
          SRW_HIP(hipEventRecord(c->ev[q][cur][(size_t)r], h->stream));
        }
      });
      if (world > 1) {
        bar.wait();
        guard([&] {
          for (int q = 0; q < np; ++q) {
            sel(q);
            for (int o = 0; o < world; ++o)
              if (o != r) SRW_HIP(hipStreamWaitEvent(h->stream, c->ev[q][cur][(size_t)o], 0));
          }
        });
      }
    }
    const int fin = (L + 1) & 1;
    guard([&] {
      for (int q = 0; q < np; ++q) {
        sel(q);
        ck(c, r, srw_shard_flush(h, &pops[(size_t)q].P, pops[(size_t)q].B, &lay[(size_t)q], c->recv[q][fin][(size_t)r].p, pops[(size_t)q].pth[(size_t)r], pops[(size_t)q].len[(size_t)r]));
      }
      for (int q = 0; q < np; ++q) { sel(q); ck(c, r, srw_shard_finish(h, &st[(size_t)q][(size_t)r], &of[(size_t)q][(size_t)r])); }
    });
    if (np > 1) (void)srw_shard_select(h, 0);        // whatever happened: population 0's context is the handle's own again
  };
  if (world == 1) body(0);
  else {
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r) th.emplace_back(body, r);
    for (auto &t : th) t.join();
  }
  for (int r = 0; r < world; ++r)
    if (err_code[(size_t)r] != SRW_OK) throw Error(err_code[(size_t)r], err[(size_t)r]);
  for (int q = 0; q < np; ++q) {
    Pop &pp = pops[(size_t)q];
    pp.overflow = false;
    memset(&pp.bt, 0, sizeof pp.bt);
    for (int r = 0; r < world; ++r) {
      const srw_walk_stats &x = st[(size_t)q][(size_t)r];
      pp.overflow |= of[(size_t)q][(size_t)r] != 0;
      pp.bt.n_steps += x.n_steps; pp.bt.dead_ends += x.dead_ends; pp.bt.sum_deg_curr += x.sum_deg_curr; pp.bt.sum_deg_prev += x.sum_deg_prev;
      pp.bt.ent_reads += x.ent_reads; pp.bt.fallbacks += x.fallbacks; pp.bt.trials += x.trials;
      for (int i = 0; i < 12; ++i) pp.bt.strategy_steps[i] += x.strategy_steps[i];
      pp.bt.edge_tables += x.edge_tables; pp.bt.edge_table_bytes += x.edge_table_bytes;      // per shard: summed = the whole graph's set
    }
  }
}
bool run_batch(srw_cluster *c, const srw_walk_params &P, int32_t B, double slack, const std::vector<int32_t *> &pth,
               const std::vector<int32_t *> &len, srw_walk_stats &bt) {
  std::vector<Pop> pops(1);
  pops[0].P = P; pops[0].B = B; pops[0].pth = pth; pops[0].len = len;
  run_populations(c, pops, slack);
  bt = pops[0].bt;
  return pops[0].overflow;
}

struct WalkPlan { std::vector<int64_t> n_local; int64_t n_global = 0, stride = 0; int32_t batch = 1; int kind = 1; };
WalkPlan plan_walk(srw_cluster *c, const srw_walk_params &P0, int32_t batch) {
  const int32_t world = c->world();
  if (P0.num_walks < 0 || P0.walk_length < 0) throw Error(SRW_ERR_INVALID, "bad walk parameters");
  if (P0.sampler != SRW_SAMPLER_REFERENCE) throw Error(SRW_ERR_INVALID, "the vertex-sharded walk runs Mode R");
  WalkPlan w;
  w.stride = (int64_t)P0.walk_length + 2;
  w.n_local.assign((size_t)world, 0);
  for (int r = 0; r < world; ++r) ck(c, r, srw_shard_capacity(c->sh[(size_t)r], &w.n_local[(size_t)r], &w.n_global));
  w.kind = (P0.p == 1.0f && P0.q == 1.0f && !(P0.flags & SRW_WALK_FORCE_GENERAL)) ? 1 : 2;
  if (w.n_global == 0 || P0.num_walks == 0) return w;
  if (w.kind == 1 && P0.rng_mode == SRW_RNG_PHILOX && c->rows_linked < 0) link_rows(c);
  if (batch <= 0) {   // as many iterations per population as keep a shard's chunk buffers under ~2 GiB
    const int64_t per_iter = std::max<int64_t>(1, w.n_global / world * 30);      // 24 B of chunk space per resident walker x slack
    batch = (int32_t)std::max<int64_t>(1, std::min<int64_t>(P0.num_walks, ((int64_t)2 << 30) / per_iter));
  }
  w.batch = std::min(batch, std::max(P0.num_walks, 1));
  for (int r = 0; r < world; ++r) {
    if (c->vrank[(size_t)r].size() != (size_t)w.n_local[(size_t)r]) {
      c->vrank[(size_t)r].assign((size_t)w.n_local[(size_t)r], 0);
      if (w.n_local[(size_t)r]) ck(c, r, srw_shard_vertex_ranks(c->sh[(size_t)r], c->vrank[(size_t)r].data()));
    }
  }
  return w;
}
void add_stats(srw_walk_stats &tot, const srw_walk_stats &bt) {
  tot.n_steps += bt.n_steps; tot.dead_ends += bt.dead_ends; tot.sum_deg_curr += bt.sum_deg_curr; tot.sum_deg_prev += bt.sum_deg_prev;
  tot.ent_reads += bt.ent_reads; tot.fallbacks += bt.fallbacks; tot.trials += bt.trials;
  for (int i = 0; i < 12; ++i) tot.strategy_steps[i] += bt.strategy_steps[i];
  tot.edge_tables = bt.edge_tables; tot.edge_table_bytes = bt.edge_table_bytes;
}
}  // namespace

extern "C" {

int32_t srw_cluster_walk(srw_cluster *c, const srw_walk_params *params, int32_t batch, srw_walk_stats *stats) {
  if (!c || !params) return SRW_ERR_INVALID;
  return cguard(c, [&] {
    const srw_walk_params P0 = *params;
    const int32_t world = c->world();
    c->valid = false; c->batches.clear();
    const WalkPlan w = plan_walk(c, P0, batch);
    c->walk_length = P0.walk_length; c->num_walks = P0.num_walks;
    srw_walk_stats tot; memset(&tot, 0, sizeof tot);
    tot.kernel_kind = w.kind;
    if (w.n_global == 0 || P0.num_walks == 0) { if (stats) *stats = tot; c->valid = true; return; }
    for (int r = 0; r < world; ++r) {
      SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
      c->paths[(size_t)r].ensure((size_t)std::max<int64_t>(1, (int64_t)P0.num_walks * w.n_local[(size_t)r] * w.stride));
      c->lens[(size_t)r].ensure((size_t)std::max<int64_t>(1, (int64_t)P0.num_walks * w.n_local[(size_t)r]));
    }
    const auto t0 = std::chrono::steady_clock::now();
    double slack = 1.25;
    std::vector<int32_t *> pth((size_t)world), len((size_t)world);
    // A batch of B >= 2 iterations runs as TWO populations of B / 2 and B - B / 2 on two streams per shard (run_populations):
    // identical paths (a population is defined by its iterations), SRW_CLUSTER_POPULATIONS=1 for the single-population form.
    // (world 1 has no exchange to hide: two half populations only double the launches there — 0.61 against 0.68 of the replicated
    //  kernel at RMAT-24, r04 s114 — so one population unless SRW_CLUSTER_POPULATIONS=2 asks for two)
    const int pops_env = getenv("SRW_CLUSTER_POPULATIONS") ? atoi(getenv("SRW_CLUSTER_POPULATIONS")) : 0;
    const bool two = pops_env == 2 || (pops_env != 1 && world > 1);
    for (int32_t it0 = 0; it0 < P0.num_walks;) {
      const int32_t B = std::min(w.batch, P0.num_walks - it0);
      std::vector<Pop> pops((two && B >= 2) ? 2 : 1);
      int32_t off = 0;
      for (size_t q = 0; q < pops.size(); ++q) {
        const int32_t Bq = pops.size() == 1 ? B : (q == 0 ? B / 2 : B - B / 2);
        pops[q].P = P0; pops[q].P.first_walk = P0.first_walk + it0 + off; pops[q].P.num_walks = Bq; pops[q].B = Bq;
        pops[q].pth.resize((size_t)world); pops[q].len.resize((size_t)world);
        for (int r = 0; r < world; ++r) {
          pops[q].pth[(size_t)r] = c->paths[(size_t)r].p + (int64_t)(it0 + off) * w.n_local[(size_t)r] * w.stride;
          pops[q].len[(size_t)r] = c->lens[(size_t)r].p + (int64_t)(it0 + off) * w.n_local[(size_t)r];
        }
        off += Bq;
      }
      run_populations(c, pops, slack);
      bool overflow = false;
      for (const Pop &pp : pops) overflow |= pp.overflow;
      if (overflow) {      // a chunk was too small for this graph's skew: same batch again with more room
        if (getenv("SRW_TIMING")) fprintf(stderr, "[cluster] chunk overflow at slack %.2f (batch %d, iteration %d): retrying\n", slack, B, it0);
        slack *= 2.0;
        if (slack > 64.0 * world) throw Error(SRW_ERR_NOMEM, "vertex-sharded walk: chunk overflow persists at 64 x world slack");
        continue;
      }
      off = 0;
      for (const Pop &pp : pops) { add_stats(tot, pp.bt); c->batches.push_back({it0 + off, pp.B}); off += pp.B; }
      it0 += B;
    }
    tot.kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    tot.n_walkers = (int64_t)P0.num_walks * w.n_global;
    tot.record_bytes = 16;
    if (stats) *stats = tot;
    c->valid = true;
  });
}

int32_t srw_cluster_fetch_paths(srw_cluster *c, int32_t *paths, int32_t *lens) {
  if (!c) return SRW_ERR_INVALID;
  return cguard(c, [&] {
    if (!c->valid) throw Error(SRW_ERR_INVALID, "no walk result (srw_cluster_walk_and_save streams its paths to the files and keeps none; call srw_cluster_walk)");
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
  // RandomWalk.save is per partition (RandomWalk.scala:234-241); here the job is streamed batch by batch and, inside a batch,
  // slice by slice of the canonical order (walk iteration major, source id ascending): every shard hands over the rows of its
  // vertices whose global rank falls into the slice (a contiguous local range: local vertices are in id order), the host puts
  // them at their canonical position and the writer formats and appends the slice.  Host memory: one slice + its staging
  // (SRW_CLUSTER_SLICE_ROWS rows, default 2 M = 1.3 GB at walkLength 80) instead of every path of every iteration (config 4,
  // numWalks 10: 207 GB); device memory: one batch of paths per shard instead of all numWalks.
  return cguard(c, [&] {
    const srw_walk_params P0 = *params;
    const int32_t world = c->world();
    int64_t nv = 0, ne = 0;
    ck(c, 0, srw_graph_stats(c->sh[0], &nv, &ne));
    PathWriter writer(output_dir, n_parts, (int64_t)P0.num_walks * nv, write_crc != 0);       // fails first if <output>/path exists
    c->valid = false; c->batches.clear();
    const WalkPlan w = plan_walk(c, P0, 0);
    srw_walk_stats tot; memset(&tot, 0, sizeof tot);
    tot.kernel_kind = w.kind;
    if (w.n_global == 0 || P0.num_walks == 0) { writer.close(); if (stats) *stats = tot; return; }
    const int64_t stride = w.stride;
    for (int r = 0; r < world; ++r) {
      SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
      c->paths[(size_t)r].ensure((size_t)std::max<int64_t>(1, (int64_t)w.batch * w.n_local[(size_t)r] * stride));
      c->lens[(size_t)r].ensure((size_t)std::max<int64_t>(1, (int64_t)w.batch * w.n_local[(size_t)r]));
    }
    int64_t slice_rows = 2 << 20;
    if (const char *e = getenv("SRW_CLUSTER_SLICE_ROWS"); e && *e) slice_rows = std::max<int64_t>(1, atoll(e));
    slice_rows = std::min<int64_t>(slice_rows, w.n_global);
    std::vector<int32_t> sl_paths((size_t)(slice_rows * stride)), sl_lens((size_t)slice_rows);
    std::vector<std::vector<int32_t>> st_paths((size_t)world), st_lens((size_t)world);       // per-shard staging of a slice's rows
    double walk_ms = 0.0, slack = 1.25;
    std::vector<int32_t *> pth((size_t)world), len((size_t)world);
    for (int r = 0; r < world; ++r) { pth[(size_t)r] = c->paths[(size_t)r].p; len[(size_t)r] = c->lens[(size_t)r].p; }
    for (int32_t it0 = 0; it0 < P0.num_walks;) {
      const int32_t B = std::min(w.batch, P0.num_walks - it0);
      srw_walk_params P = P0; P.first_walk = P0.first_walk + it0; P.num_walks = B;
      srw_walk_stats bt;
      const auto t0 = std::chrono::steady_clock::now();
      const bool overflow = run_batch(c, P, B, slack, pth, len, bt);
      walk_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (overflow) {
        slack *= 2.0;
        if (slack > 64.0 * world) throw Error(SRW_ERR_NOMEM, "vertex-sharded walk: chunk overflow persists at 64 x world slack");
        continue;
      }
      add_stats(tot, bt);
      // the batch's rows leave in canonical order: iteration by iteration, slice by slice of the global ranks
      for (int32_t i = 0; i < B; ++i) {
        for (int64_t g0 = 0; g0 < w.n_global; g0 += slice_rows) {
          const int64_t g1 = std::min<int64_t>(g0 + slice_rows, w.n_global);
          std::vector<std::thread> th;
          std::vector<int32_t> rc((size_t)world, SRW_OK);
          std::vector<std::string> em((size_t)world);
          for (int r = 0; r < world; ++r)
            th.emplace_back([&, r] {
              try {
                const std::vector<int32_t> &vr = c->vrank[(size_t)r];
                const int64_t l0 = std::lower_bound(vr.begin(), vr.end(), (int32_t)g0) - vr.begin();
                const int64_t l1 = std::lower_bound(vr.begin(), vr.end(), (int32_t)std::min<int64_t>(g1, 0x7FFFFFFF)) - vr.begin();
                const int64_t n = l1 - l0;
                if (n <= 0) return;
                SRW_HIP(hipSetDevice(c->dev[(size_t)r]));
                std::vector<int32_t> &sp = st_paths[(size_t)r], &sn = st_lens[(size_t)r];
                if (sp.size() < (size_t)(n * stride)) sp.resize((size_t)(n * stride));
                if (sn.size() < (size_t)n) sn.resize((size_t)n);
                // row lw = local vertex * B + iteration in the batch: the slice's rows of iteration i are B rows apart
                SRW_HIP(hipMemcpy2D(sp.data(), (size_t)stride * 4, c->paths[(size_t)r].p + ((int64_t)l0 * B + i) * stride, (size_t)B * stride * 4,
                                    (size_t)stride * 4, (size_t)n, hipMemcpyDeviceToHost));
                SRW_HIP(hipMemcpy2D(sn.data(), 4, c->lens[(size_t)r].p + (int64_t)l0 * B + i, (size_t)B * 4, 4, (size_t)n, hipMemcpyDeviceToHost));
                for (int64_t k = 0; k < n; ++k) {
                  const int64_t at = (int64_t)vr[(size_t)(l0 + k)] - g0;
                  memcpy(sl_paths.data() + at * stride, sp.data() + k * stride, (size_t)stride * 4);
                  sl_lens[(size_t)at] = sn[(size_t)k];
                }
              } catch (const Error &e) { rc[(size_t)r] = e.code; em[(size_t)r] = e.what(); }
            });
          for (auto &t : th) t.join();
          for (int r = 0; r < world; ++r) if (rc[(size_t)r] != SRW_OK) throw Error(rc[(size_t)r], em[(size_t)r]);
          writer.append(sl_paths.data(), sl_lens.data(), g1 - g0, stride);
        }
      }
      it0 += B;
    }
    writer.close();
    tot.kernel_ms = walk_ms;
    tot.n_walkers = (int64_t)P0.num_walks * w.n_global;
    tot.record_bytes = 16;
    if (stats) *stats = tot;
  });
}

}  // extern "C"
