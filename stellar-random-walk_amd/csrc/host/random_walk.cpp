// random_walk.cpp — see random_walk.h.
#include "random_walk.h"

#include <chrono>
#include <cstdlib>
#include <iostream>

namespace randomwalk {
namespace algorithm {
namespace {
// SRW_TIMING=1: phase wall times on stderr (stdout stays the reference's lines)
struct Phase {
  const char *name; std::chrono::steady_clock::time_point t0;
  explicit Phase(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
  ~Phase() {
    if (!getenv("SRW_TIMING")) return;
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "[timing] " << name << ": " << ms << " ms\n";
  }
};
void check(srw_handle *h, int32_t rc, const char *what) {
  if (rc != SRW_OK) throw std::runtime_error(std::string(what) + ": " + srw_last_error(h));
}
}  // namespace

int64_t GraphMap::getNumVertices() const { int64_t v = 0, e = 0; srw_graph_stats(h_, &v, &e); return v; }
int64_t GraphMap::getNumEdges() const { int64_t v = 0, e = 0; srw_graph_stats(h_, &v, &e); return e; }
bool GraphMap::getNeighbors(int32_t vid, std::vector<std::pair<int32_t, float>> &out) const {
  int64_t n = 0;
  check(h_, srw_graph_neighbors(h_, vid, nullptr, nullptr, 0, &n), "getNeighbors");
  out.clear();
  if (n < 0) return false;
  std::vector<int32_t> ids((size_t)n); std::vector<float> w((size_t)n);
  if (n) check(h_, srw_graph_neighbors(h_, vid, ids.data(), w.data(), n, &n), "getNeighbors");
  for (int64_t k = 0; k < n; ++k) out.emplace_back(ids[(size_t)k], w[(size_t)k]);
  return true;
}
bool GraphMap::getPartition(int32_t vid, int32_t &pid) const {
  int32_t known = 0;
  srw_graph_partition(h_, vid, &pid, &known);
  return known != 0;
}

static void split(const std::vector<std::pair<int32_t, float>> &e, std::vector<int32_t> &ids, std::vector<float> &w) {
  ids.resize(e.size()); w.resize(e.size());
  for (size_t k = 0; k < e.size(); ++k) { ids[k] = e[k].first; w[k] = e[k].second; }
}

std::pair<int32_t, float> RandomSample::sample(const std::vector<std::pair<int32_t, float>> &edges) const {
  std::vector<int32_t> ids; std::vector<float> w; split(edges, ids, w);
  int64_t k = 0;
  check(h_, srw_sample(h_, w.data(), (int64_t)w.size(), nextFloat_(), &k), "sample");
  return edges.at((size_t)k);
}
std::vector<std::pair<int32_t, float>> RandomSample::computeSecondOrderWeights(
    float p, float q, int32_t prevId, const std::vector<std::pair<int32_t, float>> &prevNeighbors,
    const std::vector<std::pair<int32_t, float>> &currNeighbors) const {
  std::vector<int32_t> pi, ci; std::vector<float> pw, cw; split(prevNeighbors, pi, pw); split(currNeighbors, ci, cw);
  std::vector<float> out(cw.size());
  static const int32_t none = 0;
  check(h_, srw_second_order_weights(h_, p, q, prevId, pi.empty() ? &none : pi.data(), (int64_t)pi.size(), ci.data(),
                                     cw.data(), (int64_t)cw.size(), out.data()), "computeSecondOrderWeights");
  std::vector<std::pair<int32_t, float>> r;
  for (size_t k = 0; k < ci.size(); ++k) r.emplace_back(ci[k], out[k]);
  return r;
}
std::pair<int32_t, float> RandomSample::secondOrderSample(float p, float q, int32_t prevId,
                                                           const std::vector<std::pair<int32_t, float>> &prevNeighbors,
                                                           const std::vector<std::pair<int32_t, float>> &currNeighbors) const {
  std::vector<int32_t> pi, ci; std::vector<float> pw, cw; split(prevNeighbors, pi, pw); split(currNeighbors, ci, cw);
  int64_t k = 0;
  static const int32_t none = 0;
  check(h_, srw_second_order_sample(h_, p, q, prevId, pi.empty() ? &none : pi.data(), (int64_t)pi.size(), ci.data(),
                                    cw.data(), (int64_t)cw.size(), nextFloat_(), &k), "secondOrderSample");
  return {ci.at((size_t)k), cw.at((size_t)k)};   // sample returns the biased tuple's id; weight of the original edge kept for reference
}

RandomWalk::RandomWalk(const Params &config, std::ostream *log) : config_(config), log_(log) {
  srw_config c{};
  c.device = config.device; c.rank = 0; c.world = 1;
  Phase ph("create (HIP runtime, context, code objects)");
  int32_t rc = srw_create(&c, &h_);
  if (rc != SRW_OK) throw std::runtime_error(std::string("srw_create: ") + srw_last_error(nullptr));
}
RandomWalk::~RandomWalk() { srw_destroy(h_); }

void RandomWalk::printGraphStats() {
  check(h_, srw_graph_stats(h_, &nVertices, &nEdges), "graph stats");
  if (log_) {
    // UniformRandomWalk.scala:69-79 / VCutRandomWalk.scala:80-90.  One process = one "JVM": a single pair of totals.
    *log_ << "edges: " << nEdges << "\n" << "vertices: " << nVertices << "\n"
          << "E Partitions: " << nEdges << "\n" << "V Partitions: " << nVertices << "\n";
  }
}

void UniformRandomWalk::loadGraph() {
  Phase ph("loadGraph (parse + upload + device CSR build)");
  check(h_, srw_load_edgelist(h_, config_.input.c_str(), config_.directed, config_.weighted, /*partitioned=*/0,
                              config_.rddPartitions), "loadGraph");
  printGraphStats();
}
void VCutRandomWalk::loadGraph() {
  Phase ph("loadGraph (parse + upload + device CSR build)");
  check(h_, srw_load_edgelist(h_, config_.input.c_str(), config_.directed, config_.weighted, config_.partitioned ? 1 : 0,
                              config_.rddPartitions), "loadGraph");
  printGraphStats();
}

void RandomWalk::executeOnDevice() {
  loadGraph();
  Phase ph("randomWalk (kernels; the paths stay in HBM)");
  srw_walk_params P{};
  P.p = (float)config_.p; P.q = (float)config_.q;               // .toFloat, :112
  P.walk_length = config_.walkLength; P.num_walks = config_.numWalks; P.first_walk = 0;
  // (--constR is honoured here as in walkImpl / executeAndSave, and it switches the alias sampler off: a constant draw is a Mode R notion)
  P.rng_mode = config_.hasConstR ? SRW_RNG_CONST : SRW_RNG_PHILOX; P.const_r = config_.hasConstR ? config_.constR : 0.0f; P.seed = (uint32_t)config_.seed;
  P.sampler = (config_.alias && !config_.hasConstR) ? SRW_SAMPLER_ALIAS : SRW_SAMPLER_REFERENCE;
  srw_walk_stats st{};
  check(h_, srw_walk(h_, &P, &st), "randomWalk");
  if (log_) {
    std::vector<int32_t> lens((size_t)config_.numWalks * (size_t)nVertices);
    check(h_, srw_fetch_paths(h_, nullptr, lens.data()), "randomWalk");
    const int32_t stride = config_.walkLength + 2;
    for (int it = 0; it < config_.numWalks; ++it) {
      int64_t dead = 0;
      const int32_t *ln = lens.data() + (size_t)it * nVertices;
      for (int64_t i = 0; i < nVertices; ++i) dead += (ln[i] >= 2 && ln[i] < stride);
      *log_ << "Unfinished Walkers: 0\n";                       // :154
      if (dead) *log_ << "Wrong Transports: 0\n" << "Zero Neighbors: " << dead << "\n";  // :155-160
    }
  }
}
void RandomWalk::saveFromDevice(int partitions, const std::string &output) const {
  Phase ph("save (device formatter + part files)");
  int32_t rc = srw_write_paths(h_, output.c_str(), partitions, config_.crc ? 1 : 0);
  if (rc == SRW_ERR_EXISTS)
    throw std::runtime_error("org.apache.hadoop.mapred.FileAlreadyExistsException: Output directory " + output + "/" +
                             common::Property::pathSuffix + " already exists");
  check(h_, rc, "save");
}

Paths RandomWalk::walkImpl(bool useConst, float constR) {
  Phase ph("randomWalk (kernels + path transfer, overlapped)");
  Paths out;
  out.stride = config_.walkLength + 2;
  out.n = (int64_t)config_.numWalks * nVertices;
  if (srw_host_alloc((size_t)out.n * out.stride * 4, (void **)&out.ids) != SRW_OK ||
      srw_host_alloc((size_t)out.n * 4, (void **)&out.lens) != SRW_OK)
    throw std::runtime_error("cannot allocate pinned host memory for the paths");
  // for (_ <- 0 until config.numWalks) (:82): all iterations in one call; iteration i's kernel overlaps the PCIe
  // transfer of iteration i-1
  srw_walk_params P{};
  P.p = (float)config_.p; P.q = (float)config_.q;               // .toFloat, :112
  P.walk_length = config_.walkLength; P.num_walks = config_.numWalks; P.first_walk = 0;
  P.rng_mode = useConst ? SRW_RNG_CONST : SRW_RNG_PHILOX; P.const_r = constR; P.seed = (uint32_t)config_.seed;
  P.sampler = (config_.alias && !useConst) ? SRW_SAMPLER_ALIAS : SRW_SAMPLER_REFERENCE;
  srw_walk_stats st{};
  check(h_, srw_walk_to_host(h_, &P, out.ids, out.lens, &st), "randomWalk");
  if (log_) {
    for (int it = 0; it < config_.numWalks; ++it) {
      // acc2 ("Zero Neighbors", :117): walkers that hit a dead end inside the second-order loop of this iteration
      int64_t dead = 0;
      const int32_t *ln = out.lens + (size_t)it * nVertices;
      for (int64_t i = 0; i < nVertices; ++i) dead += (ln[i] >= 2 && ln[i] < out.stride);
      *log_ << "Unfinished Walkers: 0\n";                       // :154 (one super-step per iteration on one GPU)
      if (dead) *log_ << "Wrong Transports: 0\n" << "Zero Neighbors: " << dead << "\n";  // :155-160
    }
  }
  return out;
}
// --gpus N: the graph is sharded by source vertex over N devices of this node and the walkers cross shards every
// super-step (srw_cluster_*: what replaces transferWalkersToTheirPartitions, RandomWalk.scala:186-192); same files.
void RandomWalk::executeAndSaveSharded(int partitions, const std::string &output) {
  std::vector<int32_t> devs;
  // SRW_CLUSTER_SAME_DEVICE=1 (tests on a one-GPU box): every shard on --device
  const bool same = getenv("SRW_CLUSTER_SAME_DEVICE") != nullptr;
  for (int i = 0; i < config_.gpus; ++i) devs.push_back(same ? config_.device : config_.device + i);
  srw_cluster *cl = nullptr;
  // q == 1: no shard ever tests "x in N(prev)" — skip the replicated membership structure (memory per shard ~ 1 / N)
  const int32_t cflags = (config_.partitioned ? SRW_CFG_OWNER_FROM_PARTITIONS : 0) | ((float)config_.q == 1.0f ? SRW_CFG_NO_MEMBERSHIP : 0);
  if (srw_cluster_create(devs.data(), (int32_t)devs.size(), cflags, &cl) != SRW_OK)
    throw std::runtime_error(std::string("srw_cluster_create: ") + srw_last_error(nullptr));
  struct Guard { srw_cluster *c; ~Guard() { srw_cluster_destroy(c); } } guard{cl};
  auto ckc = [&](int32_t rc, const char *what) {
    if (rc == SRW_ERR_EXISTS)
      throw std::runtime_error("org.apache.hadoop.mapred.FileAlreadyExistsException: Output directory " + output + "/" +
                               common::Property::pathSuffix + " already exists");
    if (rc != SRW_OK) throw std::runtime_error(std::string(what) + ": " + srw_cluster_last_error(cl));
  };
  {
    Phase ph("loadGraph (parse + upload + device CSR build, one shard per GPU)");
    ckc(srw_cluster_load_edgelist(cl, config_.input.c_str(), config_.directed, config_.weighted, config_.partitioned ? 1 : 0,
                                  config_.rddPartitions), "loadGraph");
    ckc(srw_cluster_graph_stats(cl, &nVertices, &nEdges), "graph stats");
    if (log_)
      *log_ << "edges: " << nEdges << "\n" << "vertices: " << nVertices << "\n" << "E Partitions: " << nEdges << "\n"
            << "V Partitions: " << nVertices << "\n";
  }
  Phase ph("randomWalk + save (vertex-sharded super-steps over xGMI, then format + write)");
  srw_walk_params P{};
  P.p = (float)config_.p; P.q = (float)config_.q;
  P.walk_length = config_.walkLength; P.num_walks = config_.numWalks; P.first_walk = 0;
  P.rng_mode = config_.hasConstR ? SRW_RNG_CONST : SRW_RNG_PHILOX; P.const_r = config_.constR; P.seed = (uint32_t)config_.seed;
  P.sampler = SRW_SAMPLER_REFERENCE;
  srw_walk_stats st{};
  ckc(srw_cluster_walk_and_save(cl, &P, output.c_str(), partitions, config_.crc ? 1 : 0, &st), "randomWalk");
  if (log_)
    for (int it = 0; it < config_.numWalks; ++it) *log_ << "Unfinished Walkers: 0\n";
  if (log_ && st.dead_ends) *log_ << "Wrong Transports: 0\n" << "Zero Neighbors: " << st.dead_ends << "\n";
}

void RandomWalk::executeAndSave(int partitions, const std::string &output) {
  if (config_.gpus > 1) { executeAndSaveSharded(partitions, output); return; }
  loadGraph();
  Phase ph("randomWalk + save (kernel / PCIe / format+write pipelined)");
  srw_walk_params P{};
  P.p = (float)config_.p; P.q = (float)config_.q;
  P.walk_length = config_.walkLength; P.num_walks = config_.numWalks; P.first_walk = 0;
  P.rng_mode = config_.hasConstR ? SRW_RNG_CONST : SRW_RNG_PHILOX; P.const_r = config_.constR; P.seed = (uint32_t)config_.seed;
  P.sampler = (config_.alias && !config_.hasConstR) ? SRW_SAMPLER_ALIAS : SRW_SAMPLER_REFERENCE;
  srw_walk_stats st{};
  std::vector<int64_t> dead((size_t)std::max(config_.numWalks, 1), 0);
  if (config_.deviceFormat) P.flags |= SRW_WALK_DEVICE_FORMAT;
  int32_t rc = srw_walk_and_save(h_, &P, output.c_str(), partitions, config_.crc ? 1 : 0, &st, dead.data());
  if (rc == SRW_ERR_EXISTS)
    throw std::runtime_error("org.apache.hadoop.mapred.FileAlreadyExistsException: Output directory " + output + "/" +
                             common::Property::pathSuffix + " already exists");
  check(h_, rc, "randomWalk");
  if (log_)
    for (int it = 0; it < config_.numWalks; ++it) {
      *log_ << "Unfinished Walkers: 0\n";
      if (dead[(size_t)it]) *log_ << "Wrong Transports: 0\n" << "Zero Neighbors: " << dead[(size_t)it] << "\n";
    }
}

Paths RandomWalk::randomWalk() { return walkImpl(config_.hasConstR, config_.constR); }
Paths RandomWalk::randomWalk(float constR) { return walkImpl(true, constR); }

void RandomWalk::save(const Paths &paths, int partitions, const std::string &output) const {
  Phase ph("save (format + write)");
  int32_t rc = srw_save_paths(paths.ids, paths.lens, paths.n, paths.stride, output.c_str(), partitions,
                              config_.crc ? 1 : 0);
  if (rc == SRW_ERR_EXISTS)
    throw std::runtime_error("org.apache.hadoop.mapred.FileAlreadyExistsException: Output directory " + output + "/" +
                             common::Property::pathSuffix + " already exists");
  if (rc != SRW_OK) throw std::runtime_error("save failed");
}

}  // namespace algorithm
}  // namespace randomwalk
