// writer.cpp — path files.  Replaces RandomWalk.save (M/algorithm/RandomWalk.scala:234-241):
//   path.mkString("\t")  -> repartition(n) -> saveAsTextFile(s"$output/path")  (M/common/Property.scala:6)
// i.e. <output>/path/part-00000 .. part-(n-1) (5-digit), one '\n'-terminated line of TAB-joined decimal ids
// per path, an empty _SUCCESS marker, and (optionally) the Hadoop ChecksumFileSystem ".name.crc" side files.
// The job fails if <output>/path already exists.  The reference's line order is unspecified (repartition
// after union); this writer emits the canonical order (walk iteration major, source id ascending) in
// contiguous slices per part.  Formatting is done by one std::thread per slice of walkers.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <memory>
#include <thread>

#include "engine.h"

namespace srw {
namespace {

inline char *put_int(char *p, int32_t v) {
  char tmp[12];
  uint32_t u = v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
  int n = 0;
  do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
  if (v < 0) *p++ = '-';
  while (n) *p++ = tmp[--n];
  return p;
}

uint32_t crc32_ieee(const unsigned char *d, size_t n) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; ++i) c = table[(c ^ d[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// Hadoop ChecksumFileSystem side file: "crc\0", int32 BE bytesPerSum (512), then one BE CRC32 per 512-byte chunk.
void write_crc_file(const std::string &dir, const std::string &name, const std::string &content) {
  std::string fn = dir + "/." + name + ".crc";
  FILE *f = fopen(fn.c_str(), "wb");
  if (!f) throw Error(SRW_ERR_IO, "cannot write " + fn);
  const unsigned char hdr[8] = {'c', 'r', 'c', 0, 0, 0, 2, 0};
  fwrite(hdr, 1, 8, f);
  for (size_t off = 0; off < content.size(); off += 512) {
    size_t n = std::min<size_t>(512, content.size() - off);
    uint32_t c = crc32_ieee((const unsigned char *)content.data() + off, n);
    unsigned char be[4] = {(unsigned char)(c >> 24), (unsigned char)(c >> 16), (unsigned char)(c >> 8), (unsigned char)c};
    fwrite(be, 1, 4, f);
  }
  fclose(f);
}

struct Piece {
  std::unique_ptr<char[]> buf;  // uninitialised storage: no zero-fill pass over gigabytes
  size_t len = 0;
};

void format_range(const int32_t *paths, const int32_t *lens, int64_t stride, int64_t b, int64_t e, Piece &out) {
  size_t cap = 1;
  for (int64_t w = b; w < e; ++w) cap += (size_t)lens[w] * 12 + 1;
  out.buf.reset(new char[cap]);
  char *p = out.buf.get();
  for (int64_t w = b; w < e; ++w) {
    const int32_t *row = paths + w * stride;
    for (int32_t t = 0; t < lens[w]; ++t) {
      if (t) *p++ = '\t';
      p = put_int(p, row[t]);
    }
    *p++ = '\n';
  }
  out.len = (size_t)(p - out.buf.get());
}

void pwrite_all(int fd, const char *p, size_t n, off_t off, const std::string &fn) {
  while (n) {
    ssize_t k = pwrite(fd, p, n, off);
    if (k <= 0) throw Error(SRW_ERR_IO, "short write " + fn);
    p += k; n -= (size_t)k; off += k;
  }
}

}  // namespace

void write_path_files(const int32_t *paths, const int32_t *lens, int64_t n_walkers, int64_t stride,
                      const char *output_dir, int n_parts, bool write_crc) {
  if (n_parts < 1) n_parts = 1;
  std::string out(output_dir);
  mkdir(out.c_str(), 0777);  // the output root may exist
  std::string dir = out + "/path";
  if (mkdir(dir.c_str(), 0777) != 0) {
    if (errno == EEXIST) throw Error(SRW_ERR_EXISTS, "Output directory " + dir + " already exists");
    throw Error(SRW_ERR_IO, "cannot create " + dir + ": " + strerror(errno));
  }
  const int64_t per = (n_walkers + n_parts - 1) / n_parts;
  unsigned hw = std::thread::hardware_concurrency();
  if (!hw) hw = 1;
  for (int part = 0; part < n_parts; ++part) {
    int64_t b = std::min<int64_t>((int64_t)part * per, n_walkers), e = std::min<int64_t>(b + per, n_walkers);
    int64_t n = e - b;
    int nt = (int)std::max<int64_t>(1, std::min<int64_t>(hw, n / 4096 + 1));
    std::vector<Piece> pieces((size_t)nt);
    {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) {
        int64_t tb = b + n * t / nt, te = b + n * (t + 1) / nt;
        th.emplace_back([&, t, tb, te] { format_range(paths, lens, stride, tb, te, pieces[(size_t)t]); });
      }
      for (auto &x : th) x.join();
    }
    char name[32];
    snprintf(name, sizeof(name), "part-%05d", part);
    std::string fn = dir + "/" + name;
    int fd = open(fn.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) throw Error(SRW_ERR_IO, "cannot write " + fn);
    // every piece knows its byte offset once all are formatted: the threads write their pieces concurrently
    std::vector<off_t> offs((size_t)nt + 1, 0);
    for (int t = 0; t < nt; ++t) offs[(size_t)t + 1] = offs[(size_t)t] + (off_t)pieces[(size_t)t].len;
    std::vector<std::string> errs((size_t)nt);
    {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
          try { pwrite_all(fd, pieces[(size_t)t].buf.get(), pieces[(size_t)t].len, offs[(size_t)t], fn); }
          catch (const Error &e) { errs[(size_t)t] = e.what(); }
        });
      for (auto &x : th) x.join();
    }
    close(fd);
    for (auto &e : errs) if (!e.empty()) throw Error(SRW_ERR_IO, e);
    if (write_crc) {
      std::string all;
      all.reserve((size_t)offs[(size_t)nt]);
      for (auto &pc : pieces) all.append(pc.buf.get(), pc.len);
      write_crc_file(dir, name, all);
    }
  }
  std::string ok = dir + "/_SUCCESS";
  FILE *f = fopen(ok.c_str(), "wb");
  if (!f) throw Error(SRW_ERR_IO, "cannot write " + ok);
  fclose(f);
  if (write_crc) write_crc_file(dir, "_SUCCESS", std::string());
}

}  // namespace srw
