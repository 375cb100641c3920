// random_walk.h — host-side mirror of the reference's walk driver interface, on top of the C ABI.
//   trait RandomWalk          M/algorithm/RandomWalk.scala:12-242
//   UniformRandomWalk         M/algorithm/UniformRandomWalk.scala
//   VCutRandomWalk            M/algorithm/VCutRandomWalk.scala
//   GraphMap                  M/algorithm/GraphMap.scala  (reads go to the CSR in HBM)
//   RandomSample              M/algorithm/RandomSample.scala (device arithmetic through the unit hooks)
// Same method names and argument meaning; RDDs become plain host vectors, errors become std::runtime_error
// (the JVM exceptions of the reference).
#pragma once
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "params.h"
#include "stellar_rw.h"

namespace randomwalk {
namespace algorithm {

using common::Params;

struct Paths {  // RDD[Array[Int]]: n paths of up to stride ids, in pinned host memory (srw_host_alloc)
  int32_t *ids = nullptr;   // [n][stride], unused tail -1
  int32_t *lens = nullptr;  // [n]
  int64_t n = 0;
  int32_t stride = 0;
  Paths() = default;
  Paths(const Paths &) = delete;
  Paths &operator=(const Paths &) = delete;
  Paths(Paths &&o) noexcept : ids(o.ids), lens(o.lens), n(o.n), stride(o.stride) { o.ids = nullptr; o.lens = nullptr; }
  ~Paths() { srw_host_free(ids); srw_host_free(lens); }
  std::vector<int32_t> path(int64_t i) const { return std::vector<int32_t>(ids + i * stride, ids + i * stride + lens[i]); }
};

// object GraphMap: the per-process adjacency store.  Backed by the handle's CSR in HBM.
class GraphMap {
 public:
  explicit GraphMap(srw_handle *h) : h_(h) {}
  int64_t getNumVertices() const;                                   // GraphMap.scala:87-89
  int64_t getNumEdges() const;                                      // :91-93
  // getNeighbors (:109-120): returns false for `null`; an empty vector for a vertex with no out-edges.
  bool getNeighbors(int32_t vid, std::vector<std::pair<int32_t, float>> &out) const;
  bool getPartition(int32_t vid, int32_t &pid) const;               // :66-68
 private:
  srw_handle *h_;
};

// case class RandomSample(nextFloat): the arithmetic runs on the GPU through the unit hooks.
class RandomSample {
 public:
  RandomSample(srw_handle *h, std::function<float()> nextFloat) : h_(h), nextFloat_(std::move(nextFloat)) {}
  std::pair<int32_t, float> sample(const std::vector<std::pair<int32_t, float>> &edges) const;   // RandomSample.scala:12-25
  std::vector<std::pair<int32_t, float>> computeSecondOrderWeights(                               // :27-44
      float p, float q, int32_t prevId, const std::vector<std::pair<int32_t, float>> &prevNeighbors,
      const std::vector<std::pair<int32_t, float>> &currNeighbors) const;
  std::pair<int32_t, float> secondOrderSample(float p, float q, int32_t prevId,                   // :55-62
                                              const std::vector<std::pair<int32_t, float>> &prevNeighbors,
                                              const std::vector<std::pair<int32_t, float>> &currNeighbors) const;
 private:
  srw_handle *h_;
  std::function<float()> nextFloat_;
};

class RandomWalk {  // trait RandomWalk
 public:
  RandomWalk(const Params &config, std::ostream *log);
  virtual ~RandomWalk();
  RandomWalk(const RandomWalk &) = delete;
  RandomWalk &operator=(const RandomWalk &) = delete;

  int64_t nVertices = 0;  // RandomWalk.scala:23 (Int there; int64 here, see SURVEY §7 hard part 5)
  int64_t nEdges = 0;     // :24

  Paths execute() { loadGraph(); return randomWalk(); }            // :31-33
  virtual void loadGraph() = 0;                                     // :41
  // randomWalk (:75-176).  Default RNG: keyed Philox (config.seed); pass constR to inject
  // nextFloat = () => constR like the reference's tests (T/UniformRandomWalkTest.scala:183-185).
  Paths randomWalk();
  Paths randomWalk(float constR);
  void save(const Paths &paths, int partitions, const std::string &output) const;  // :234-241
  // loadGraph + randomWalk + save fused and streamed (what Main.doRandomWalk does, M/Main.scala:53-62), without
  // materialising the paths on the host
  void executeAndSave(int partitions, const std::string &output);
  void executeAndSaveSharded(int partitions, const std::string &output);   // --gpus N > 1 (srw_cluster_*)
  // loadGraph + randomWalk with the paths LEFT IN HBM (srw_walk) for a consumer on the device — the embedding stage of
  // `--cmd node2vec` (M/Main.scala:113-117 feeds randomWalk's RDD to Word2Vec.fit) — and save() straight from there
  void executeOnDevice();
  void saveFromDevice(int partitions, const std::string &output) const;
  GraphMap graphMap() const { return GraphMap(h_); }
  srw_handle *handle() const { return h_; }

 protected:
  Paths walkImpl(bool useConst, float constR);
  void printGraphStats();
  Params config_;
  srw_handle *h_ = nullptr;
  std::ostream *log_;
};

class UniformRandomWalk : public RandomWalk {
 public:
  using RandomWalk::RandomWalk;
  void loadGraph() override;  // UniformRandomWalk.scala:17-88
};

class VCutRandomWalk : public RandomWalk {
 public:
  using RandomWalk::RandomWalk;
  void loadGraph() override;  // VCutRandomWalk.scala:13-98
};

}  // namespace algorithm
}  // namespace randomwalk
