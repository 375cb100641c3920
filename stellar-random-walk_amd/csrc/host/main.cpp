// main.cpp — `stellar-rw`: native stand-in for `spark-submit --class au.csiro.data61.randomwalk.Main`
// (M/Main.scala:18-27,53-69,109-127) for the --cmd randomwalk path.  Same flags, same stdout lines, same
// <output>/path layout.  --cmd node2vec / embedding (M/Main.scala:113-124): the Word2Vec stage is the build's GPU skip-gram +
// hierarchical softmax (csrc/embedding.hip; MLlib's Word2Vec is absent from the reference tree: parity unpinned), <output>/vec
// holds "id\tv0\t..." lines as Main.saveModelAndFeatures writes them, <output>/bin the model directory (metadata JSON + Parquet data, as Word2VecModel.save).
// `--cmd embedding --input X`: X is a file or a directory of part files (context.textFile), its tokens are words (any string).
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

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

// configureWord2Vec (Main.scala:77-97): setMinCount(0), --lr, --iter, --dim, --window; --w2vPartitions only shapes MLlib's own
// parallelism (here: one logical partition, one wave per sentence, Hogwild across the waves as word2vec.c across its threads — so the
// vectors of two runs differ in their last bits even with the same --seed; SRW_W2V_DETERMINISTIC=1 trains in ONE wave, sentence after
// sentence: reproducible, and what the tests compare with the oracle's restatement)
static srw_w2v_params w2vParams(const Params &param) {
  srw_w2v_params wp;
  wp.dim = param.w2vDim; wp.window = param.w2vWindow; wp.iterations = param.w2vIter; wp.learning_rate = (float)param.w2vLr;
  wp.seed = (uint32_t)param.seed;
  const char *det = getenv("SRW_W2V_DETERMINISTIC");
  wp.threads = (det && *det == '1') ? 1 : 0;
  return wp;
}
static void throwSave(int32_t rc) {
  if (rc != SRW_OK) throw std::runtime_error(std::string("org.apache.hadoop.mapred.FileAlreadyExistsException: ") + srw_last_error(nullptr));
}

static void doNode2vec(const Params &param) {   // Main.scala:113-117: doRandomWalk, then Word2Vec on the paths
  std::unique_ptr<algorithm::RandomWalk> rw;
  if (param.partitioned) rw.reset(new algorithm::VCutRandomWalk(param, &std::cout));
  else rw.reset(new algorithm::UniformRandomWalk(param, &std::cout));
  // the reference feeds randomWalk's RDD to Word2Vec.fit without collecting it: the paths stay in HBM between the two stages
  // (srw_walk -> srw_write_paths for <output>/path -> srw_w2v_fit_device), they cross PCIe once, as text
  rw->executeOnDevice();
  rw->saveFromDevice(getNumOutputPartition(param), param.output);
  const srw_w2v_params wp = w2vParams(param);
  int32_t *vocab = nullptr; float *vec = nullptr; int64_t nv = 0;
  const auto t0 = std::chrono::steady_clock::now();
  if (srw_w2v_fit_device(rw->handle(), nullptr, nullptr, 0, 1, &wp, &vocab, &vec, &nv) != SRW_OK)
    throw std::runtime_error(std::string("word2vec: ") + srw_last_error(rw->handle()));
  const auto t1 = std::chrono::steady_clock::now();
  const int32_t rc = srw_w2v_save(vocab, vec, nv, wp.dim, param.output.c_str(), getNumOutputPartition(param));
  if (getenv("SRW_TIMING"))
    std::cerr << "[timing] Word2Vec fit (vocabulary + " << wp.iterations << " iterations, paths in HBM): " << std::chrono::duration<double, std::milli>(t1 - t0).count()
              << " ms; model + vectors saved: " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count() << " ms\n";
  srw_free(vocab); srw_free(vec);
  throwSave(rc);
}

// String.split("\\s+") as Scala applies it to a line (Main.scala:121): runs of \s (space, \t, \n, \v, \f, \r) separate; a line that
// STARTS with whitespace yields a leading empty token, trailing empty tokens are dropped, and an empty line is ONE empty token
static void javaSplit(const std::string &line, std::vector<std::string> &out) {
  out.clear();
  auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; };
  size_t i = 0;
  const size_t n = line.size();
  if (n == 0) { out.emplace_back(); return; }
  if (ws(line[0])) { out.emplace_back(); while (i < n && ws(line[i])) ++i; }
  while (i < n) {
    size_t j = i;
    while (j < n && !ws(line[j])) ++j;
    out.emplace_back(line, i, j - i);
    i = j;
    while (i < n && ws(line[i])) ++i;
  }
  if (out.size() == 1 && out[0].empty() && n > 0) out.clear();      // a line of whitespace only: "" then nothing -> Java drops the trailing empties: []
}

// the files behind context.textFile(path): the file itself, or — a directory, e.g. the randomwalk stage's own <output>/path — its
// part files in name order (Hadoop skips names that start with '_' or '.': _SUCCESS, .part-00000.crc)
static std::vector<std::string> inputFiles(const std::string &path) {
  struct stat sb;
  if (stat(path.c_str(), &sb) != 0) throw std::runtime_error("org.apache.hadoop.mapred.InvalidInputException: Input path does not exist: " + path);
  std::vector<std::string> files;
  if (!S_ISDIR(sb.st_mode)) { files.push_back(path); return files; }
  DIR *d = opendir(path.c_str());
  if (!d) throw std::runtime_error("Input path does not exist: " + path);
  while (dirent *e = readdir(d)) {
    const std::string name = e->d_name;
    if (name.empty() || name[0] == '_' || name[0] == '.') continue;
    struct stat fb;
    if (stat((path + "/" + name).c_str(), &fb) == 0 && S_ISREG(fb.st_mode)) files.push_back(path + "/" + name);
  }
  closedir(d);
  std::sort(files.begin(), files.end());
  return files;
}

// Fast path of doEmbedding for what the randomwalk stage writes — lines of canonical int32 tokens: the files are mapped, cut into pieces
// at line ends and parsed by all host threads straight into ids (the general path below keeps every token as a std::string: 849 M of
// them for config 2's <output>/path).  Anything else — a token that is not a canonical int32 (as std::to_string prints it), a line that
// starts with white space (split's empty first token), an empty line (one empty token) — returns false and the general path decides.
static bool parseNumericFast(const std::vector<std::string> &files, std::vector<int32_t> &ids, std::vector<int32_t> &lens, size_t &stride) {
  struct Map { const char *p = nullptr; size_t n = 0; };
  std::vector<Map> maps(files.size());
  struct Unmap { std::vector<Map> &m; ~Unmap() { for (auto &x : m) if (x.p && x.n) munmap((void *)x.p, x.n); } } unmap{maps};
  struct Piece { size_t file; size_t b, e; std::vector<int32_t> toks, lens; bool ok = true; };
  std::vector<Piece> pieces;
  for (size_t f = 0; f < files.size(); ++f) {
    const int fd = open(files[f].c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return false; }
    if (sb.st_size > 0) {
      void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m == MAP_FAILED) { close(fd); return false; }
      maps[f].p = (const char *)m; maps[f].n = (size_t)sb.st_size;
    }
    close(fd);
    const size_t target = (size_t)32 << 20;
    size_t b = 0;
    while (b < maps[f].n) {
      size_t e = std::min(maps[f].n, b + target);
      while (e < maps[f].n && maps[f].p[e - 1] != '\n') ++e;           // (to the end of the line the cut falls into)
      Piece pc; pc.file = f; pc.b = b; pc.e = e;
      pieces.push_back(std::move(pc));
      b = e;
    }
  }
  auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f' || c == '\r'; };
  std::atomic<size_t> next{0};
  std::atomic<bool> bad{false};
  auto work = [&] {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= pieces.size() || bad.load(std::memory_order_relaxed)) return;
      Piece &pc = pieces[i];
      const char *p = maps[pc.file].p + pc.b, *end = maps[pc.file].p + pc.e;
      while (p < end) {
        const char *le = (const char *)memchr(p, '\n', (size_t)(end - p));
        if (!le) le = end;
        if (p == le || ws(*p)) { pc.ok = false; bad = true; return; }      // an empty line / a leading blank: an empty token
        int32_t n_in_row = 0;
        const char *q = p;
        while (q < le) {
          bool neg = false;
          if (*q == '-') { neg = true; ++q; }
          const char *d0 = q;
          long long v = 0;
          while (q < le && *q >= '0' && *q <= '9' && q - d0 < 11) v = v * 10 + (*q++ - '0');
          const long nd = (long)(q - d0);
          if (nd == 0 || (q < le && !ws(*q)) || (nd > 1 && *d0 == '0') || (neg && v == 0) || (neg ? -v < (long long)INT32_MIN : v > (long long)INT32_MAX)) { pc.ok = false; bad = true; return; }
          if (n_in_row == 1000) { pc.lens.push_back(n_in_row); n_in_row = 0; }       // MLlib's maxSentenceLength
          pc.toks.push_back((int32_t)(neg ? -v : v)); ++n_in_row;
          while (q < le && ws(*q)) ++q;
        }
        pc.lens.push_back(n_in_row);
        p = le < end ? le + 1 : end;
      }
    }
  };
  {
    const unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), (unsigned)std::max<size_t>(pieces.size(), 1)));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(work);
    for (auto &x : th) x.join();
  }
  if (bad) return false;
  size_t n_rows = 0;
  stride = 1;
  for (const auto &pc : pieces) { n_rows += pc.lens.size(); for (int32_t l : pc.lens) stride = std::max(stride, (size_t)l); }
  ids.assign(n_rows * stride, -1); lens.resize(n_rows);
  std::vector<size_t> row0(pieces.size() + 1, 0);
  for (size_t i = 0; i < pieces.size(); ++i) row0[i + 1] = row0[i] + pieces[i].lens.size();
  std::atomic<size_t> nx{0};
  auto fill = [&] {
    for (;;) {
      const size_t i = nx.fetch_add(1);
      if (i >= pieces.size()) return;
      const Piece &pc = pieces[i];
      size_t t = 0;
      for (size_t r = 0; r < pc.lens.size(); ++r) {
        lens[row0[i] + r] = pc.lens[r];
        memcpy(ids.data() + (row0[i] + r) * stride, pc.toks.data() + t, (size_t)pc.lens[r] * 4);
        t += (size_t)pc.lens[r];
      }
    }
  };
  {
    const unsigned nt = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), (unsigned)std::max<size_t>(pieces.size(), 1)));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(fill);
    for (auto &x : th) x.join();
  }
  return n_rows > 0;
}

static void fitAndSaveIds(const Params &param, const std::vector<int32_t> &ids, const std::vector<int32_t> &lens, size_t stride) {
  srw_config cfg; memset(&cfg, 0, sizeof(cfg)); cfg.device = param.device; cfg.rank = 0; cfg.world = 1;
  srw_handle *h = nullptr;
  if (srw_create(&cfg, &h) != SRW_OK) throw std::runtime_error(std::string("srw_create: ") + srw_last_error(nullptr));
  int32_t *vocab = nullptr; float *vec = nullptr; int64_t nv = 0;
  const srw_w2v_params wp = w2vParams(param);
  if (srw_w2v_fit(h, ids.data(), lens.data(), (int64_t)lens.size(), (int64_t)stride, &wp, &vocab, &vec, &nv) != SRW_OK) {
    const std::string msg = std::string("word2vec: ") + srw_last_error(h);
    srw_destroy(h);
    throw std::runtime_error(msg);
  }
  srw_destroy(h);
  const int32_t rc = srw_w2v_save(vocab, vec, nv, wp.dim, param.output.c_str(), getNumOutputPartition(param));
  srw_free(vocab); srw_free(vec);
  throwSave(rc);
}

static void doEmbedding(const Params &param) {  // Main.scala:119-124: textFile(input).map(_.split("\\s+")) -> Word2Vec
  const std::vector<std::string> files = inputFiles(param.input);
  {
    std::vector<int32_t> fids, flens; size_t fstride = 1;
    const auto t0 = std::chrono::steady_clock::now();
    if (!getenv("SRW_EMBEDDING_GENERAL_PARSER") && parseNumericFast(files, fids, flens, fstride)) {      // (the switch: tests compare the two parsers)
      if (getenv("SRW_TIMING")) std::cerr << "[timing] embedding input: " << flens.size() << " sentences of canonical ids parsed by the host threads in "
                                          << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() << " ms\n";
      fitAndSaveIds(param, fids, flens, fstride);
      return;
    }
  }
  std::vector<std::vector<std::string>> rows;
  std::vector<std::string> toks;
  for (const std::string &fn : files) {
    std::ifstream in(fn);
    if (!in) throw std::runtime_error("Input path does not exist: " + fn);
    std::string line;
    // MLlib's Word2Vec cuts a sentence after maxSentenceLength = 1000 words (its default; Main.configureWord2Vec does not change it): a
    // longer line is several sentences, and no window crosses the cut
    while (std::getline(in, line)) {
      javaSplit(line, toks);
      if (toks.size() <= 1000) { rows.push_back(toks); continue; }
      for (size_t i = 0; i < toks.size(); i += 1000)
        rows.emplace_back(toks.begin() + (long)i, toks.begin() + (long)std::min(toks.size(), i + 1000));
    }
  }
  size_t stride = 1, n_tok = 0;
  for (const auto &r : rows) { stride = std::max(stride, r.size()); n_tok += r.size(); }
  if (n_tok == 0) throw std::runtime_error("java.lang.IllegalArgumentException: requirement failed: The vocabulary size should be > 0 (no words under " + param.input + ")");
  // Words are strings in the reference.  A text of canonical int32 tokens (what the randomwalk stage writes) keeps the ids as they are,
  // so that `--cmd embedding` on <output>/path trains exactly what `--cmd node2vec` trains; any other text gets a dictionary (the
  // distinct tokens in byte order) and its vectors are saved under the words themselves.
  bool numeric = true;
  for (const auto &r : rows) {
    for (const auto &t : r) {
      errno = 0;
      char *end = nullptr;
      const long long v = t.empty() ? 0 : strtoll(t.c_str(), &end, 10);
      if (t.empty() || *end != 0 || errno == ERANGE || v > INT32_MAX || v < INT32_MIN || std::to_string(v) != t) { numeric = false; break; }
    }
    if (!numeric) break;
  }
  std::vector<std::string> dict;
  if (!numeric) {
    for (const auto &r : rows) dict.insert(dict.end(), r.begin(), r.end());
    std::sort(dict.begin(), dict.end());
    dict.erase(std::unique(dict.begin(), dict.end()), dict.end());
  }
  std::vector<int32_t> ids(rows.size() * stride, -1), lens(rows.size());
  for (size_t i = 0; i < rows.size(); ++i) {
    lens[i] = (int32_t)rows[i].size();
    for (size_t k = 0; k < rows[i].size(); ++k)
      ids[i * stride + k] = numeric ? (int32_t)strtoll(rows[i][k].c_str(), nullptr, 10)
                                    : (int32_t)(std::lower_bound(dict.begin(), dict.end(), rows[i][k]) - dict.begin());
  }
  srw_config cfg; memset(&cfg, 0, sizeof(cfg)); cfg.device = param.device; cfg.rank = 0; cfg.world = 1;
  srw_handle *h = nullptr;
  if (srw_create(&cfg, &h) != SRW_OK) throw std::runtime_error(std::string("srw_create: ") + srw_last_error(nullptr));
  int32_t *vocab = nullptr; float *vec = nullptr; int64_t nv = 0;
  const srw_w2v_params wp = w2vParams(param);
  if (srw_w2v_fit(h, ids.data(), lens.data(), (int64_t)rows.size(), (int64_t)stride, &wp, &vocab, &vec, &nv) != SRW_OK) {
    const std::string msg = std::string("word2vec: ") + srw_last_error(h);
    srw_destroy(h);
    throw std::runtime_error(msg);
  }
  srw_destroy(h);
  int32_t rc;
  if (numeric) rc = srw_w2v_save(vocab, vec, nv, wp.dim, param.output.c_str(), getNumOutputPartition(param));
  else {
    std::vector<const char *> words((size_t)nv);
    for (int64_t r = 0; r < nv; ++r) words[(size_t)r] = dict[(size_t)vocab[r]].c_str();
    rc = srw_w2v_save_words(words.data(), vec, nv, wp.dim, param.output.c_str(), getNumOutputPartition(param));
  }
  srw_free(vocab); srw_free(vec);
  throwSave(rc);
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
