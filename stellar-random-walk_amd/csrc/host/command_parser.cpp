// command_parser.cpp — scopt-compatible option parsing for the flag table of M/common/CommandParser.scala:34-90.
#include "command_parser.h"

#include <algorithm>
#include <cstdlib>
#include <sstream>
#include <stdexcept>

namespace randomwalk {
namespace common {
namespace {

bool readInt(const std::string &s, int &out) {  // scopt Read.intRead = _.toInt
  if (s.empty()) return false;
  size_t i = (s[0] == '-' || s[0] == '+') ? 1 : 0;
  if (i == s.size()) return false;
  long long v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (s[i] - '0');
    if (v > 2147483648LL) return false;
  }
  if (s[0] == '-') v = -v;
  if (v > 2147483647LL || v < -2147483648LL) return false;
  out = (int)v;
  return true;
}
bool readLong(const std::string &s, int64_t &out) {
  char *e = nullptr;
  long long v = strtoll(s.c_str(), &e, 10);
  if (s.empty() || *e) return false;
  out = v;
  return true;
}
bool readDouble(const std::string &s, double &out) {  // _.toDouble
  if (s.empty()) return false;
  char *e = nullptr;
  out = strtod(s.c_str(), &e);
  if (*e == 'd' || *e == 'D' || *e == 'f' || *e == 'F') ++e;
  return *e == 0;
}
bool readBool(const std::string &s0, bool &out) {  // scopt Read.booleanRead
  std::string s = s0;
  std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return (char)tolower(c); });
  if (s == "true" || s == "yes" || s == "1") { out = true; return true; }
  if (s == "false" || s == "no" || s == "0") { out = false; return true; }
  return false;
}

}  // namespace

std::string CommandParser::usage() {
  Params d;
  std::ostringstream o;
  o << "Main\nUsage: 2nd Order Random Walk + Word2Vec [options]\n\n"
    << "  --walkLength <value>     walkLength: " << d.walkLength << "\n"
    << "  --numWalks <value>       numWalks: " << d.numWalks << "\n"
    << "  --p <value>              return parameter p: " << d.p << "\n"
    << "  --q <value>              in-out parameter q: " << d.q << "\n"
    << "  --rddPartitions <value>  Number of RDD partitions in running Random Walk and Word2vec: " << d.rddPartitions << "\n"
    << "  --weighted <value>       weighted: true\n"
    << "  --directed <value>       directed: false\n"
    << "  --singleOutput <value>   generate single output file: true\n"
    << "  --w2vPartitions <value>  Number of partitions in word2vec: " << d.w2vPartitions << "\n"
    << "  --input <value>          Input edge file path: empty\n"
    << "  --output <value>         Output path: empty\n"
    << "  --cmd <value>            command: node2vec\n"
    << "  --partitioned <value>    Whether the graph is partitioned: false\n"
    << "  --lr <value>             Learning rate in word2vec: " << d.w2vLr << "\n"
    << "  --iter <value>           Number of iterations in word2vec: " << d.w2vIter << "\n"
    << "  --dim <value>            Number of dimensions in word2vec: " << d.w2vDim << "\n"
    << "  --window <value>         Window size in word2vec: " << d.w2vWindow << "\n"
    << "MI355X build extensions:\n"
    << "  --seed <value>           Philox seed of the walk RNG: 42\n"
    << "  --constR <value>         inject a constant nextFloat (test hook of the reference)\n"
    << "  --device <value>         HIP device ordinal: 0\n"
    << "  --gpus <value>           GPUs of this node to shard the graph over (by source vertex; 1 = whole graph on one GPU): 1\n"
    << "  --crc <value>            also write Hadoop .crc side files: false\n"
    << "  --sampler <value>        reference (bit-identical CDF inversion) | alias (alias tables + rejection): reference\n"
    << "  --deviceFormat <value>   format the path text on the GPU (false: on host threads): true\n"
    << "Environment: SRW_W2V_DETERMINISTIC=1 trains word2vec in one wave, sentence after sentence (reproducible vectors; the default "
       "trains one wave per sentence, Hogwild, and two runs differ in their last bits even with the same --seed)\n";
  return o.str();
}

std::optional<Params> CommandParser::parse(const std::vector<std::string> &args, std::string *err) {
  Params c;
  std::ostringstream e;
  bool ok = true;
  auto fail = [&](const std::string &m) { e << "Error: " << m << "\n"; ok = false; };
  for (size_t i = 0; i < args.size(); ++i) {
    std::string a = args[i];
    if (a.rfind("--", 0) != 0) { fail("Unknown argument '" + a + "'"); continue; }
    std::string name = a.substr(2), val;
    bool inline_val = false;
    size_t eq = name.find_first_of("=:");
    if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); inline_val = true; }
    static const char *known[] = {WALK_LENGTH, NUM_WALKS, P, Q, RDD_PARTITIONS, WEIGHTED, DIRECTED, SINGLE_OUTPUT,
                                  W2V_PARTITIONS, INPUT, OUTPUT, CMD, PARTITIONED, LEARNING_RATE, ITERATION, DIMENSION,
                                  WINDOW, "seed", "constR", "device", "gpus", "crc", "sampler", "deviceFormat"};
    if (std::find_if(std::begin(known), std::end(known), [&](const char *k) { return name == k; }) == std::end(known)) {
      fail("Unknown option " + a);
      continue;
    }
    if (!inline_val) {
      if (i + 1 >= args.size()) { fail("Missing value after " + a); continue; }
      val = args[++i];
    }
    auto asInt = [&](int &dst) { if (!readInt(val, dst)) fail("Option --" + name + " failed when given '" + val + "'. For input string: \"" + val + "\""); };
    auto asDouble = [&](double &dst) { if (!readDouble(val, dst)) fail("Option --" + name + " failed when given '" + val + "'. For input string: \"" + val + "\""); };
    auto asBool = [&](bool &dst) { if (!readBool(val, dst)) fail("Option --" + name + " failed when given '" + val + "'. '" + val + "' is not a boolean."); };
    if (name == WALK_LENGTH) asInt(c.walkLength);
    else if (name == NUM_WALKS) asInt(c.numWalks);
    else if (name == P) asDouble(c.p);
    else if (name == Q) asDouble(c.q);
    else if (name == RDD_PARTITIONS) asInt(c.rddPartitions);
    else if (name == WEIGHTED) asBool(c.weighted);
    else if (name == DIRECTED) asBool(c.directed);
    else if (name == SINGLE_OUTPUT) asBool(c.singleOutput);
    else if (name == W2V_PARTITIONS) asInt(c.w2vPartitions);
    else if (name == INPUT) { c.input = val; c.hasInput = true; }
    else if (name == OUTPUT) { c.output = val; c.hasOutput = true; }
    else if (name == CMD) {
      if (val == "node2vec") c.cmd = TaskName::node2vec;
      else if (val == "randomwalk") c.cmd = TaskName::randomwalk;
      else if (val == "embedding") c.cmd = TaskName::embedding;
      else throw std::out_of_range("No value found for '" + val + "'");  // Enumeration.withName
      c.hasCmd = true;
    }
    else if (name == PARTITIONED) asBool(c.partitioned);
    else if (name == LEARNING_RATE) asDouble(c.w2vLr);
    else if (name == ITERATION) asInt(c.w2vIter);
    else if (name == DIMENSION) asInt(c.w2vDim);
    else if (name == WINDOW) asInt(c.w2vWindow);
    else if (name == "seed") { if (!readLong(val, c.seed)) fail("Option --seed failed when given '" + val + "'"); }
    else if (name == "constR") { double d; if (readDouble(val, d)) { c.constR = (float)d; c.hasConstR = true; } else fail("Option --constR failed when given '" + val + "'"); }
    else if (name == "device") asInt(c.device);
    else if (name == "gpus") { asInt(c.gpus); if (c.gpus < 1 || c.gpus > 64) fail("Option --gpus must be in 1 .. 64"); }
    else if (name == "crc") asBool(c.crc);
    else if (name == "deviceFormat") asBool(c.deviceFormat);
    else if (name == "sampler") {
      if (val == "alias") c.alias = true; else if (val == "reference") c.alias = false;
      else fail("Option --sampler failed when given '" + val + "' (reference | alias)");
    }
  }
  if (!c.hasInput) fail("Missing option --input");    // .required(), CommandParser.scala:64-67
  if (!c.hasOutput) fail("Missing option --output");  // :68-71
  if (!c.hasCmd) fail("Missing option --cmd");        // :72-75
  if (!ok) {
    e << "Try --help for more information.\n";
    if (err) *err = e.str();
    return std::nullopt;
  }
  return c;
}

}  // namespace common
}  // namespace randomwalk
