// main.cpp — `stellar-rw`: native stand-in for `spark-submit --class au.csiro.data61.randomwalk.Main`
// (M/Main.scala:18-27,53-69,109-127) for the --cmd randomwalk path.  Same flags, same stdout lines, same
// <output>/path layout.  --cmd node2vec / embedding (M/Main.scala:113-124): the Word2Vec stage is the build's GPU skip-gram +
// hierarchical softmax (csrc/embedding.hip; MLlib's Word2Vec is absent from the reference tree: parity unpinned), <output>/vec
// holds "id\tv0\t..." lines as Main.saveModelAndFeatures writes them, <output>/bin the model directory (metadata + vectors as text).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>

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

// configureWord2Vec + fit + saveModelAndFeatures (Main.scala:36-44,77-97): setMinCount(0), --lr, --iter, --dim, --window;
// --w2vPartitions only shapes MLlib's own parallelism (here: one logical partition, one wave per sentence)
static void fitAndSave(srw_handle *h, const int32_t *ids, const int32_t *lens, int64_t n, int64_t stride, const Params &param) {
  srw_w2v_params wp;
  wp.dim = param.w2vDim; wp.window = param.w2vWindow; wp.iterations = param.w2vIter; wp.learning_rate = (float)param.w2vLr;
  wp.seed = (uint32_t)param.seed; wp.threads = 0;
  int32_t *vocab = nullptr; float *vec = nullptr; int64_t nv = 0;
  if (srw_w2v_fit(h, ids, lens, n, stride, &wp, &vocab, &vec, &nv) != SRW_OK) throw std::runtime_error(std::string("word2vec: ") + srw_last_error(h));
  const int32_t rc = srw_w2v_save(vocab, vec, nv, wp.dim, param.output.c_str(), getNumOutputPartition(param));
  srw_free(vocab); srw_free(vec);
  if (rc != SRW_OK) throw std::runtime_error(std::string("org.apache.hadoop.mapred.FileAlreadyExistsException: ") + srw_last_error(nullptr));
}

static void doNode2vec(const Params &param) {   // Main.scala:113-117: doRandomWalk, then Word2Vec on the paths
  std::unique_ptr<algorithm::RandomWalk> rw;
  if (param.partitioned) rw.reset(new algorithm::VCutRandomWalk(param, &std::cout));
  else rw.reset(new algorithm::UniformRandomWalk(param, &std::cout));
  algorithm::Paths paths = rw->execute();
  rw->save(paths, getNumOutputPartition(param), param.output);
  fitAndSave(rw->handle(), paths.ids, paths.lens, paths.n, paths.stride, param);
}

static void doEmbedding(const Params &param) {  // Main.scala:119-124: textFile(input).map(_.split("\\s+")) -> Word2Vec
  std::ifstream in(param.input);
  if (!in) throw std::runtime_error("Input path does not exist: " + param.input);
  std::vector<std::vector<int32_t>> rows;
  std::string line;
  size_t stride = 1;
  while (std::getline(in, line)) {
    std::vector<int32_t> r;
    std::istringstream ls(line);
    std::string tok;
    while (ls >> tok) {                        // (the reference keeps any token as a word; this build takes vertex ids)
      char *end = nullptr;
      const long v = strtol(tok.c_str(), &end, 10);
      if (*end != 0 || tok.empty()) throw std::runtime_error("embedding: token '" + tok + "' is not a vertex id");
      r.push_back((int32_t)v);
    }
    stride = std::max(stride, r.size());
    rows.push_back(std::move(r));
  }
  std::vector<int32_t> ids(rows.size() * stride, -1), lens(rows.size());
  for (size_t i = 0; i < rows.size(); ++i) { lens[i] = (int32_t)rows[i].size(); std::copy(rows[i].begin(), rows[i].end(), ids.begin() + i * stride); }
  srw_config cfg; memset(&cfg, 0, sizeof(cfg)); cfg.device = 0; cfg.rank = 0; cfg.world = 1;
  srw_handle *h = nullptr;
  if (srw_create(&cfg, &h) != SRW_OK) throw std::runtime_error(std::string("srw_create: ") + srw_last_error(nullptr));
  try { fitAndSave(h, ids.data(), lens.data(), (int64_t)rows.size(), (int64_t)stride, param); }
  catch (...) { srw_destroy(h); throw; }
  srw_destroy(h);
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
        doNode2vec(*params);
        break;
      case TaskName::embedding:
        doEmbedding(*params);
        break;
    }
    std::cout.flush();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "Exception in thread \"main\" " << e.what() << "\n";
    return 1;
  }
}
