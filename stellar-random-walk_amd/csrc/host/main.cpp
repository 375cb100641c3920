// main.cpp — `stellar-rw`: native stand-in for `spark-submit --class au.csiro.data61.randomwalk.Main`
// (M/Main.scala:18-27,53-69,109-127) for the --cmd randomwalk path.  Same flags, same stdout lines, same
// <output>/path layout.  --cmd node2vec / embedding (MLlib Word2Vec) are out of scope and rejected.
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <memory>

#include "command_parser.h"
#include "random_walk.h"

using namespace randomwalk;
using common::CommandParser;
using common::Params;
using common::TaskName;

static int getNumOutputPartition(const Params &param) {  // Main.scala:64-69
  return param.singleOutput ? 1 : param.rddPartitions;
}

static void doRandomWalk(const Params &param) {  // Main.scala:53-62
  std::unique_ptr<algorithm::RandomWalk> rw;
  if (param.partitioned) rw.reset(new algorithm::VCutRandomWalk(param, &std::cout));
  else rw.reset(new algorithm::UniformRandomWalk(param, &std::cout));
  // rw.execute() followed by rw.save(paths, n, output), fused: the paths stream GPU -> pinned ring -> part files
  rw->executeAndSave(getNumOutputPartition(param), param.output);
}

int main(int argc, char **argv) {
  const auto t_main = std::chrono::steady_clock::now();
  struct AtExit { std::chrono::steady_clock::time_point t0; ~AtExit() { if (getenv("SRW_TIMING")) std::cerr << "[timing] main() body: " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() << " ms\n"; } } at_exit{t_main};
  std::vector<std::string> args(argv + 1, argv + argc);
  for (auto &a : args)
    if (a == "--help") { std::cout << CommandParser::usage(); return 0; }
  try {
    std::string err;
    auto params = CommandParser::parse(args, &err);
    if (!params) {                       // case None => sys.exit(1), Main.scala:25
      std::cerr << err << CommandParser::usage();
      return 1;
    }
    switch (params->cmd) {               // runJob, Main.scala:112-125
      case TaskName::randomwalk:
        doRandomWalk(*params);
        break;
      case TaskName::node2vec:
      case TaskName::embedding:
        std::cerr << "--cmd " << common::toString(params->cmd)
                  << ": the Word2Vec embedding stage (MLlib) is outside this engine's scope; only --cmd randomwalk is served\n";
        return 2;
    }
    std::cout.flush();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "Exception in thread \"main\" " << e.what() << "\n";
    return 1;
  }
}
