// edgelist.cpp — host-side edge-list tokenizer with the reference's exact token rules.
//
// Replaces the per-line lambda of UniformRandomWalk.loadGraph (M/algorithm/UniformRandomWalk.scala:26-34)
// and VCutRandomWalk.loadGraph (M/algorithm/VCutRandomWalk.scala:21-34):
//   * records are Hadoop TextInputFormat lines: terminated by \n, \r\n or a lone \r; a final unterminated
//     line counts; an empty line is a record (and makes the reference throw);
//   * String.split("\\s+") — \s = [ \t\n\x0B\f\r]; a separator at index 0 yields a leading "" token,
//     trailing empty tokens are dropped;
//   * ids by Integer.parseInt (optional sign, decimal digits, int32 range) — failure kills the job;
//   * weight = Float.parseFloat of the LAST column iff weighted && parts.length > 2 (> 3 when partitioned),
//     Try(...).getOrElse(1.0f); pId = parts(2).toInt iff partitioned && parts.length > 2.
// The file is split into byte ranges parsed by one std::thread each and concatenated in file order.
#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "engine.h"

namespace srw {
namespace {

inline bool java_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == 0x0B || c == '\f' || c == '\r'; }

// Integer.parseInt reads its digits with Character.digit(s.charAt(i), 10): any decimal digit of the Basic Multilingual
// Plane counts ("١٢" and "１２" are 12), one UTF-16 char at a time — so a digit beyond the BMP (a surrogate pair) does not.
// The line arrives as UTF-8 (Hadoop Text); a malformed or overlong sequence decodes to U+FFFD there: not a digit.
// Zero code points of the BMP's Nd blocks (Unicode 6.2 = Java 8, plus U+0DE6 and U+A9F0 of later JDKs).
constexpr uint16_t ND_ZERO[] = {
    0x0660, 0x06F0, 0x07C0, 0x0966, 0x09E6, 0x0A66, 0x0AE6, 0x0B66, 0x0BE6, 0x0C66, 0x0CE6, 0x0D66, 0x0DE6, 0x0E50, 0x0ED0, 0x0F20,
    0x1040, 0x1090, 0x17E0, 0x1810, 0x1946, 0x19D0, 0x1A80, 0x1A90, 0x1B50, 0x1BB0, 0x1C40, 0x1C50, 0xA620, 0xA8D0, 0xA900, 0xA9D0,
    0xA9F0, 0xAA50, 0xABF0, 0xFF10};
inline int java_digit(const char *s, size_t n, size_t &i) {   // digit value or -1; i moves past the character
  const unsigned char c = (unsigned char)s[i];
  if (c < 0x80) { ++i; return (c >= '0' && c <= '9') ? (int)(c - '0') : -1; }
  uint32_t cp;
  if ((c & 0xE0) == 0xC0 && i + 1 < n && ((unsigned char)s[i + 1] & 0xC0) == 0x80) {
    cp = ((uint32_t)(c & 0x1F) << 6) | ((unsigned char)s[i + 1] & 0x3F);
    if (cp < 0x80) return -1;
    i += 2;
  } else if ((c & 0xF0) == 0xE0 && i + 2 < n && ((unsigned char)s[i + 1] & 0xC0) == 0x80 && ((unsigned char)s[i + 2] & 0xC0) == 0x80) {
    cp = ((uint32_t)(c & 0x0F) << 12) | (((uint32_t)(unsigned char)s[i + 1] & 0x3F) << 6) | ((unsigned char)s[i + 2] & 0x3F);
    if (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF)) return -1;
    i += 3;
  } else return -1;
  for (uint16_t z : ND_ZERO) if (cp >= z && cp <= (uint32_t)z + 9) return (int)(cp - z);
  return -1;
}

bool parse_int_token(const char *s, size_t n, int32_t &out) {
  if (n == 0) return false;
  size_t i = 0;
  bool neg = false;
  if (s[0] == '-' || s[0] == '+') {
    neg = s[0] == '-';
    i = 1;
    if (n == 1) return false;
  }
  int64_t v = 0;
  while (i < n) {
    const int d = java_digit(s, n, i);
    if (d < 0) return false;
    v = v * 10 + d;
    if (v > 2147483648LL) return false;
  }
  if (neg) v = -v;
  if (v > 2147483647LL) return false;
  out = (int32_t)v;
  return true;
}

inline bool is_hex_digit(char c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'); }

// java.lang.Float.parseFloat grammar (FloatingDecimal.readJavaFormatString); value by glibc strtof, which is
// correctly rounded to binary32 like the JDK's.
bool parse_float_token(const char *s, size_t n, float &out) {
  char buf[160];
  if (n == 0 || n >= sizeof(buf) - 1) return false;
  size_t i = 0;
  bool neg = false;
  if (s[i] == '+' || s[i] == '-') { neg = s[i] == '-'; ++i; }
  if (i >= n) return false;
  if (n - i == 3 && !memcmp(s + i, "NaN", 3)) { out = NAN; return true; }
  if (n - i == 8 && !memcmp(s + i, "Infinity", 8)) { out = neg ? -INFINITY : INFINITY; return true; }
  size_t end = n;
  const bool hex = n - i >= 2 && s[i] == '0' && (s[i + 1] == 'x' || s[i + 1] == 'X');
  const char last = s[end - 1];
  if (last == 'f' || last == 'F' || last == 'd' || last == 'D') {
    if (!hex) --end;
    else {
      bool seen_p = false;
      for (size_t t = i; t < end; ++t) seen_p |= (s[t] == 'p' || s[t] == 'P');
      if (seen_p) --end;
    }
  }
  size_t j = i;
  int digits = 0;
  if (hex) {
    j += 2;
    while (j < end && is_hex_digit(s[j])) { ++j; ++digits; }
    if (j < end && s[j] == '.') { ++j; while (j < end && is_hex_digit(s[j])) { ++j; ++digits; } }
    if (!digits || j >= end || (s[j] != 'p' && s[j] != 'P')) return false;
    ++j;
    if (j < end && (s[j] == '+' || s[j] == '-')) ++j;
    int ed = 0;
    while (j < end && s[j] >= '0' && s[j] <= '9') { ++j; ++ed; }
    if (!ed || j != end) return false;
  } else {
    while (j < end && s[j] >= '0' && s[j] <= '9') { ++j; ++digits; }
    if (j < end && s[j] == '.') { ++j; while (j < end && s[j] >= '0' && s[j] <= '9') { ++j; ++digits; } }
    if (!digits) return false;
    if (j < end && (s[j] == 'e' || s[j] == 'E')) {
      ++j;
      if (j < end && (s[j] == '+' || s[j] == '-')) ++j;
      int ed = 0;
      while (j < end && s[j] >= '0' && s[j] <= '9') { ++j; ++ed; }
      if (!ed) return false;
    }
    if (j != end) return false;
  }
  memcpy(buf, s, end);
  buf[end] = 0;
  char *ep = nullptr;
  float v = strtof(buf, &ep);
  if (ep != buf + end) return false;
  out = v;
  return true;
}

struct Chunk {
  std::vector<int32_t> src, dst, pid;
  std::vector<float> w;
  int64_t first_line = 0, n_lines = 0;
  int64_t bad_line = -1;  // 1-based, relative to chunk start
  std::string bad_msg;
};

// Parses the records that START in [b, e).  `data` spans the whole file.
void parse_range(const char *data, size_t size, size_t b, size_t e, bool weighted, bool partitioned, Chunk &out) {
  size_t pos = b;
  // a range that begins right after the \r of a \r\n pair must not treat the \n as an empty record
  if (pos > 0 && pos < size && data[pos] == '\n' && data[pos - 1] == '\r') ++pos;
  const int wcols = partitioned ? 3 : 2;
  while (pos < e && pos < size) {
    size_t lb = pos;
    while (pos < size && data[pos] != '\n' && data[pos] != '\r') ++pos;
    size_t le = pos;
    if (pos < size) pos += (data[pos] == '\r' && pos + 1 < size && data[pos + 1] == '\n') ? 2 : 1;
    ++out.n_lines;
    // tokenise
    const char *t0 = nullptr, *t1 = nullptr, *t2 = nullptr, *tl = nullptr;
    size_t l0 = 0, l1 = 0, l2 = 0, ll = 0;
    int np = 0;
    size_t i = lb;
    if (le == lb) { np = 1; t0 = data + lb; l0 = 0; }           // "".split -> [""]
    else {
      if (java_ws((unsigned char)data[lb])) { np = 1; t0 = data + lb; l0 = 0; }  // leading "" token
      while (i < le) {
        while (i < le && java_ws((unsigned char)data[i])) ++i;
        if (i >= le) break;
        size_t tb = i;
        while (i < le && !java_ws((unsigned char)data[i])) ++i;
        const char *tp = data + tb; size_t tn = i - tb;
        if (np == 0) { t0 = tp; l0 = tn; } else if (np == 1) { t1 = tp; l1 = tn; } else if (np == 2) { t2 = tp; l2 = tn; }
        tl = tp; ll = tn;
        ++np;
      }
      if (np == 1 && l0 == 0) np = 0;                             // all-whitespace line -> []
    }
    if (np < 2) { out.bad_line = out.n_lines; out.bad_msg = "fewer than two columns (ArrayIndexOutOfBounds / NumberFormatException in the reference)"; return; }
    float w = 1.0f;
    if (weighted && np > wcols) { float f; if (parse_float_token(tl, ll, f)) w = f; }
    int32_t pid = -1;
    if (partitioned && np > 2) { int32_t pv; if (parse_int_token(t2, l2, pv)) pid = pv; }
    int32_t s, d;
    if (!parse_int_token(t0, l0, s) || !parse_int_token(t1, l1, d)) {
      out.bad_line = out.n_lines; out.bad_msg = "NumberFormatException for vertex id"; return;
    }
    out.src.push_back(s); out.dst.push_back(d); out.w.push_back(w); out.pid.push_back(pid);
  }
}

}  // namespace

namespace {
// the lines of one file's (decompressed) bytes
void parse_buffer(const char *data, size_t size, bool weighted, bool partitioned, ParsedLines &out) {
  out = ParsedLines();
  if (size == 0) return;
  unsigned hw = std::thread::hardware_concurrency();
  size_t nthreads = std::max<size_t>(1, std::min<size_t>(hw ? hw : 1, size / (1 << 20) + 1));
  // split points: each range starts at a record start
  std::vector<size_t> cut(nthreads + 1, size);
  cut[0] = 0;
  // Hadoop's LineRecordReader skips a UTF-8 byte order mark at the start of a file (skipUtfByteOrderMark, MAPREDUCE-5777,
  // in the Hadoop 2.6+ that Spark 2.2 ships with): the first line of such a file parses in the reference
  if (size >= 3 && (unsigned char)data[0] == 0xEF && (unsigned char)data[1] == 0xBB && (unsigned char)data[2] == 0xBF) cut[0] = 3;
  for (size_t t = 1; t < nthreads; ++t) {
    size_t p = size / nthreads * t;
    while (p < size && data[p - 1] != '\n' && data[p - 1] != '\r') ++p;
    if (p < size && data[p] == '\n' && data[p - 1] == '\r') ++p;
    cut[t] = std::max(p, cut[t - 1]);      // (cut[0] may be 3: never before the byte order mark)
  }
  std::vector<Chunk> chunks(nthreads);
  std::vector<std::thread> th;
  for (size_t t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] { parse_range(data, size, cut[t], cut[t + 1], weighted, partitioned, chunks[t]); });
  for (auto &x : th) x.join();
  int64_t lines_before = 0;
  size_t total = 0;
  for (auto &c : chunks) {
    if (c.bad_line >= 0)
      throw Error(SRW_ERR_PARSE, "line " + std::to_string(lines_before + c.bad_line) + ": " + c.bad_msg);
    lines_before += c.n_lines;
    total += c.src.size();
  }
  // every thread copies its chunk to its final place (no zero-fill, no serial concatenation)
  out.src.resize_uninit(total); out.dst.resize_uninit(total); out.w.resize_uninit(total); out.pid.resize_uninit(total);
  std::vector<size_t> at(nthreads + 1, 0);
  for (size_t t = 0; t < nthreads; ++t) at[t + 1] = at[t] + chunks[t].src.size();
  th.clear();
  for (size_t t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      const Chunk &c = chunks[t];
      const size_t k = c.src.size();
      if (!k) return;
      memcpy(out.src.data() + at[t], c.src.data(), k * 4);
      memcpy(out.dst.data() + at[t], c.dst.data(), k * 4);
      memcpy(out.w.data() + at[t], c.w.data(), k * 4);
      memcpy(out.pid.data() + at[t], c.pid.data(), k * 4);
    });
  for (auto &x : th) x.join();
}

// One input file.  `name.gz` is decompressed first, as sc.textFile does (Hadoop picks the codec by the file extension;
// gzip members may be concatenated) — zlib's gzread; the other codecs (.bz2, .snappy, .lz4, .deflate) are not read.
void parse_one_file(const char *path, bool weighted, bool partitioned, ParsedLines &out) {
  const size_t len = strlen(path);
  if (len > 3 && strcmp(path + len - 3, ".gz") == 0) {
    {   // zlib would pass a file without the gzip magic through unchanged; Hadoop's GzipCodec fails on it ("not in gzip format")
      FILE *f = fopen(path, "rb");
      if (!f) throw Error(SRW_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
      unsigned char m[2] = {0, 0};
      const size_t got = fread(m, 1, 2, f);
      fclose(f);
      if (got == 0) { out = ParsedLines(); return; }                    // an empty file: no lines
      if (got < 2 || m[0] != 0x1F || m[1] != 0x8B) throw Error(SRW_ERR_IO, std::string(path) + ": not in gzip format");
    }
    gzFile gz = gzopen(path, "rb");
    if (!gz) throw Error(SRW_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
    gzbuffer(gz, 1u << 20);
    std::vector<char> buf;
    size_t used = 0;
    while (true) {
      if (buf.size() - used < ((size_t)4 << 20)) buf.resize(std::max<size_t>(buf.size() * 2, (size_t)16 << 20));
      const int got = gzread(gz, buf.data() + used, (unsigned)std::min<size_t>(buf.size() - used, (size_t)1 << 30));
      if (got < 0) { int en = 0; const std::string m = gzerror(gz, &en); gzclose(gz); throw Error(SRW_ERR_IO, std::string(path) + ": " + m); }
      if (got == 0) break;
      used += (size_t)got;
    }
    // a stream that ends in the middle of a member is only reported at close ("unexpected end of file"): Hadoop's
    // DecompressorStream throws EOFException there and the job fails
    if (gzclose(gz) != Z_OK) throw Error(SRW_ERR_IO, std::string(path) + ": unexpected end of the gzip stream");
    parse_buffer(buf.data(), used, weighted, partitioned, out);
    return;
  }
  int fd = open(path, O_RDONLY);
  if (fd < 0) throw Error(SRW_ERR_IO, std::string("cannot open ") + path + ": " + strerror(errno));
  struct stat sb;
  if (fstat(fd, &sb) != 0) { close(fd); throw Error(SRW_ERR_IO, std::string("cannot stat ") + path); }
  const size_t size = (size_t)sb.st_size;
  if (size == 0) { close(fd); out = ParsedLines(); return; }
  const char *data = (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (data == MAP_FAILED) throw Error(SRW_ERR_IO, std::string("cannot mmap ") + path);
  struct Unmap { const char *p; size_t n; ~Unmap() { munmap((void *)p, n); } } unmap{data, size};
  parse_buffer(data, size, weighted, partitioned, out);
}
}  // namespace

// `--input` may name a directory, as with sc.textFile (FileInputFormat): every file in it whose name does not start with
// '_' or '.' (hiddenFileFilter: _SUCCESS, .crc files) is read, here in byte order of the names (what HDFS lists; the
// adjacency order — and with it the walk — follows the line order, so it is fixed like this), each with its own byte
// order mark rule and its own last unterminated line.  A subdirectory is "Not a file", as in the old mapred API.
void parse_edgelist_file(const char *path, bool weighted, bool partitioned, ParsedLines &out) {
  struct stat sb;
  if (stat(path, &sb) != 0 || !S_ISDIR(sb.st_mode)) { parse_one_file(path, weighted, partitioned, out); return; }
  std::vector<std::string> names;
  DIR *dir = opendir(path);
  if (!dir) throw Error(SRW_ERR_IO, std::string("cannot open directory ") + path + ": " + strerror(errno));
  while (struct dirent *de = readdir(dir)) {
    const std::string n = de->d_name;
    if (n.empty() || n[0] == '_' || n[0] == '.') continue;
    names.push_back(n);
  }
  closedir(dir);
  std::sort(names.begin(), names.end());
  std::vector<ParsedLines> parts(names.size());
  size_t total = 0;
  for (size_t i = 0; i < names.size(); ++i) {
    const std::string f = std::string(path) + "/" + names[i];
    struct stat fs;
    if (stat(f.c_str(), &fs) != 0 || !S_ISREG(fs.st_mode)) throw Error(SRW_ERR_IO, "Not a file: " + f);
    try { parse_one_file(f.c_str(), weighted, partitioned, parts[i]); }
    catch (const Error &e) { throw Error(e.code, names[i] + ": " + e.what()); }
    total += parts[i].src.size();
  }
  out = ParsedLines();
  out.src.resize_uninit(total); out.dst.resize_uninit(total); out.w.resize_uninit(total); out.pid.resize_uninit(total);
  size_t at = 0;
  for (auto &pt : parts) {
    const size_t k = pt.src.size();
    if (!k) continue;
    memcpy(out.src.data() + at, pt.src.data(), k * 4); memcpy(out.dst.data() + at, pt.dst.data(), k * 4);
    memcpy(out.w.data() + at, pt.w.data(), k * 4); memcpy(out.pid.data() + at, pt.pid.data(), k * 4);
    at += k;
  }
}

}  // namespace srw
