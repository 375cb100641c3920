// command_parser.h — mirror of M/common/CommandParser.scala (scopt OptionParser "2nd Order Random Walk + Word2Vec").
#pragma once
#include <optional>
#include <string>
#include <vector>

#include "params.h"

namespace randomwalk {
namespace common {

struct CommandParser {
  // Option names, CommandParser.scala:14-31
  static constexpr const char *WALK_LENGTH = "walkLength";
  static constexpr const char *NUM_WALKS = "numWalks";
  static constexpr const char *P = "p";
  static constexpr const char *Q = "q";
  static constexpr const char *RDD_PARTITIONS = "rddPartitions";
  static constexpr const char *WEIGHTED = "weighted";
  static constexpr const char *DIRECTED = "directed";
  static constexpr const char *W2V_PARTITIONS = "w2vPartitions";
  static constexpr const char *INPUT = "input";
  static constexpr const char *OUTPUT = "output";
  static constexpr const char *CMD = "cmd";
  static constexpr const char *PARTITIONED = "partitioned";
  static constexpr const char *LEARNING_RATE = "lr";
  static constexpr const char *ITERATION = "iter";
  static constexpr const char *DIMENSION = "dim";
  static constexpr const char *WINDOW = "window";
  static constexpr const char *SINGLE_OUTPUT = "singleOutput";

  // CommandParser.parse (:107-109): nullopt (after printing scopt-style errors + usage to `err`) where the
  // reference returns None; throws std::out_of_range for an unknown --cmd value (TaskName.withName throws
  // NoSuchElementException, :73).
  static std::optional<Params> parse(const std::vector<std::string> &args, std::string *err = nullptr);
  static std::string usage();
};

}  // namespace common
}  // namespace randomwalk
