// params.h — mirror of M/common/Params.scala:7-23, M/common/CommandParser.scala:9-12 (TaskName) and
// M/common/Property.scala:5-7.  Same field names, types and defaults as the reference's case class.
#pragma once
#include <cstdint>
#include <string>

namespace randomwalk {
namespace common {

enum class TaskName { node2vec, randomwalk, embedding };
inline const char *toString(TaskName t) {
  return t == TaskName::node2vec ? "node2vec" : t == TaskName::randomwalk ? "randomwalk" : "embedding";
}

struct Property {  // output sub-directory names
  static constexpr const char *modelSuffix = "bin";
  static constexpr const char *pathSuffix = "path";
  static constexpr const char *vectorSuffix = "vec";
};

struct Params {
  int w2vIter = 10;
  double w2vLr = 0.025;
  int w2vPartitions = 1;
  int w2vDim = 128;
  int w2vWindow = 10;
  int walkLength = 80;
  int numWalks = 10;
  double p = 1.0;
  double q = 1.0;
  bool weighted = true;
  bool directed = false;
  std::string input;   // null in the reference until --input is given
  std::string output;
  int rddPartitions = 200;
  bool singleOutput = true;
  bool partitioned = false;
  TaskName cmd = TaskName::node2vec;
  // ---- build extensions (absent from the reference; all optional) ----
  bool hasInput = false, hasOutput = false, hasCmd = false;
  int64_t seed = 42;        // --seed: Philox key (the reference's RNG is clock-seeded and not reproducible)
  bool hasConstR = false;   // --constR: inject nextFloat = () => constR, as the reference's tests do
  float constR = 0.0f;
  int device = 0;           // --device: HIP ordinal (first of --gpus consecutive ordinals)
  int gpus = 1;             // --gpus: > 1 = graph sharded by source vertex over that many GPUs of this node (srw_cluster_*)
  bool crc = false;         // --crc: also write Hadoop .crc side files
  bool alias = false;       // --sampler alias: Mode A (alias tables + rejection) instead of the reference-exact Mode R
  bool deviceFormat = true;   // --deviceFormat: the GPU formats the path text (SRW_WALK_DEVICE_FORMAT); false = host threads
};

}  // namespace common
}  // namespace randomwalk
