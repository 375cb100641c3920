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

#include <charconv>
#include <chrono>
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <string>
#include <cstdlib>
#include <cstdio>
#include <functional>
#include <cstring>
#include <memory>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

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

// ---- incremental writer: slices of the canonical walker order are appended as they arrive -------------------------
struct PathWriter::Impl {
  std::string dir;
  int n_parts = 1;
  int64_t total = 0, per = 0, next = 0;     // walkers: total, per part, already written
  bool write_crc = false;
  // the open part: jobs of the pool keep it alive (closed when the writer has moved on AND its last queued byte is written)
  struct PartFd {
    int fd = -1; std::mutex one_writer; off_t reserved = 0, final_size = -1;
    ~PartFd() { if (fd >= 0) { if (reserved > 0 && final_size >= 0 && final_size < reserved) (void)!ftruncate(fd, final_size); ::close(fd); } }   // (gives back what the reservation overshot)
  };   // (one inode: 12.4 GB/s with one writer, less with more)
  std::shared_ptr<PartFd> part;
  int fd = -1; int cur_part = -1; off_t file_off = 0;
  // ---- the pool behind append_text(..., token >= 0): part files are written in PARALLEL (one inode takes ~12 GB/s of buffered
  // writes on the GPU box whatever the number of threads, 8 files 74 GB/s: tools/microbench_filewrite.cpp; --singleOutput false
  // writes rddPartitions = 200 of them, Params.scala:20-21) while the caller copies the next slices out of the device
  struct Job { std::shared_ptr<PartFd> f; const char *src; size_t bytes; off_t off; int token; std::string fn; };
  static constexpr int MAX_TOKENS = 16;
  std::vector<std::thread> pool;
  std::deque<Job> jobs;
  std::mutex mu;
  std::condition_variable cv_job, cv_done;
  int pending[MAX_TOKENS] = {0}; int pending_all = 0;
  bool stop = false;
  std::string pool_error;
  double pool_job_ms = 0; size_t pool_bytes = 0; long long pool_jobs = 0;     // SRW_TIMING
  void start_pool() {
    if (!pool.empty()) return;
    const int nt = (int)std::max(2u, std::min(16u, hw / 2));
    for (int t = 0; t < nt; ++t)
      pool.emplace_back([this] {
        for (;;) {
          Job j;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_job.wait(lk, [&] { return stop || !jobs.empty(); });
            if (jobs.empty()) return;
            j = std::move(jobs.front()); jobs.pop_front();
          }
          std::string err;
          const auto tj0 = std::chrono::steady_clock::now();
          try { std::lock_guard<std::mutex> one(j.f->one_writer); pwrite_all(j.f->fd, j.src, j.bytes, j.off, j.fn); } catch (const Error &er) { err = er.what(); }
          j.f.reset();
          const double tj = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj0).count();
          {
            std::lock_guard<std::mutex> lk(mu);
            if (!err.empty() && pool_error.empty()) pool_error = err;
            --pending[j.token]; --pending_all;
            pool_job_ms += tj; pool_bytes += j.bytes; ++pool_jobs;
          }
          cv_done.notify_all();
        }
      });
  }
  void stop_pool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_job.notify_all();
    for (auto &t : pool) t.join();
    pool.clear();
  }
  void enqueue(const char *src, size_t bytes, off_t off, int token, const std::string &fn) {
    start_pool();
    { std::lock_guard<std::mutex> lk(mu); jobs.push_back(Job{part, src, bytes, off, token, fn}); ++pending[token]; ++pending_all; }
    cv_job.notify_one();
  }
  void wait_token(int token) {            // every byte handed over with this token is written (or failed: throws)
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return (token < 0 ? pending_all : pending[token]) == 0; });
    if (!pool_error.empty()) throw Error(SRW_ERR_IO, pool_error);
  }
  std::string crc_tail;                      // bytes of the current part not yet covered by a full 512-byte chunk
  std::vector<uint32_t> crcs;
  unsigned hw = 1;
  std::string part_name(int p) const { char b[32]; snprintf(b, sizeof(b), "part-%05d", p); return b; }
  void open_part(int p) {
    close_part();
    std::string fn = dir + "/" + part_name(p);
    fd = ::open(fn.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) throw Error(SRW_ERR_IO, "cannot write " + fn);
    part = std::make_shared<PartFd>(); part->fd = fd;
    cur_part = p; file_off = 0; crc_tail.clear(); crcs.clear();
  }
  void crc_feed(const char *p, size_t n) {
    if (!write_crc) return;
    size_t i = 0;
    if (!crc_tail.empty()) {
      size_t take = std::min(n, 512 - crc_tail.size());
      crc_tail.append(p, take); i = take;
      if (crc_tail.size() == 512) { crcs.push_back(crc32_ieee((const unsigned char *)crc_tail.data(), 512)); crc_tail.clear(); }
    }
    for (; i + 512 <= n; i += 512) crcs.push_back(crc32_ieee((const unsigned char *)p + i, 512));
    if (i < n) crc_tail.append(p + i, n - i);
  }
  void close_part() {
    if (fd < 0) return;
    if (part) part->final_size = file_off;
    part.reset(); fd = -1;                     // (closed now, or by the pool after the part's last queued write)
    if (write_crc) {
      if (!crc_tail.empty()) crcs.push_back(crc32_ieee((const unsigned char *)crc_tail.data(), crc_tail.size()));
      std::string fn = dir + "/." + part_name(cur_part) + ".crc";
      FILE *f = fopen(fn.c_str(), "wb");
      if (!f) throw Error(SRW_ERR_IO, "cannot write " + fn);
      const unsigned char hdr[8] = {'c', 'r', 'c', 0, 0, 0, 2, 0};
      fwrite(hdr, 1, 8, f);
      for (uint32_t c : crcs) { unsigned char be[4] = {(unsigned char)(c >> 24), (unsigned char)(c >> 16), (unsigned char)(c >> 8), (unsigned char)c}; fwrite(be, 1, 4, f); }
      fclose(f);
    }
  }
  // append walkers [b, e) of `paths` (local indices) to the currently open part
  double t_format = 0, t_write = 0;   // SRW_TIMING: where the writer's wall time goes
  void append_range(const int32_t *paths, const int32_t *lens, int64_t stride, int64_t b, int64_t e) {
    const int64_t n = e - b;
    if (n <= 0) return;
    auto t0 = std::chrono::steady_clock::now();
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(hw, n / 4096 + 1));
    std::vector<Piece> pieces((size_t)nt);
    {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) {
        int64_t tb = b + n * t / nt, te = b + n * (t + 1) / nt;
        th.emplace_back([&, t, tb, te] { format_range(paths, lens, stride, tb, te, pieces[(size_t)t]); });
      }
      for (auto &x : th) x.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    std::vector<off_t> offs((size_t)nt + 1, file_off);
    for (int t = 0; t < nt; ++t) offs[(size_t)t + 1] = offs[(size_t)t] + (off_t)pieces[(size_t)t].len;
    std::vector<std::string> errs((size_t)nt);
    const std::string fn = dir + "/" + part_name(cur_part);
    {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
          try { pwrite_all(fd, pieces[(size_t)t].buf.get(), pieces[(size_t)t].len, offs[(size_t)t], fn); }
          catch (const Error &er) { errs[(size_t)t] = er.what(); }
        });
      for (auto &x : th) x.join();
    }
    for (auto &er : errs) if (!er.empty()) throw Error(SRW_ERR_IO, er);
    file_off = offs[(size_t)nt];
    for (auto &pc : pieces) crc_feed(pc.buf.get(), pc.len);
    auto t2 = std::chrono::steady_clock::now();
    t_format += std::chrono::duration<double, std::milli>(t1 - t0).count();
    t_write += std::chrono::duration<double, std::milli>(t2 - t1).count();
  }
};

PathWriter::PathWriter(const char *output_dir, int n_parts, int64_t total_walkers, bool write_crc) : p_(new Impl()) {
  p_->n_parts = n_parts < 1 ? 1 : n_parts;
  p_->total = total_walkers; p_->write_crc = write_crc;
  p_->per = (total_walkers + p_->n_parts - 1) / p_->n_parts;
  p_->hw = std::max(1u, std::thread::hardware_concurrency());
  std::string out(output_dir);
  mkdir(out.c_str(), 0777);  // the output root may exist
  p_->dir = out + "/path";
  if (mkdir(p_->dir.c_str(), 0777) != 0) {
    const int err = errno;                       // before anything that may call malloc/free and clobber it
    std::string d = p_->dir;
    delete p_; p_ = nullptr;
    errno = err;
    if (err == EEXIST) throw Error(SRW_ERR_EXISTS, "Output directory " + d + " already exists");
    throw Error(SRW_ERR_IO, "cannot create " + d + ": " + strerror(errno));
  }
}
PathWriter::~PathWriter() { if (p_) { p_->stop_pool(); p_->part.reset(); delete p_; } }

void PathWriter::append(const int32_t *paths, const int32_t *lens, int64_t n, int64_t stride) {
  int64_t done = 0;
  while (done < n) {
    const int64_t g = p_->next;                                   // global index of the next walker
    const int part = p_->per > 0 ? (int)std::min<int64_t>(g / p_->per, p_->n_parts - 1) : 0;
    if (part != p_->cur_part) p_->open_part(part);
    const int64_t part_end = std::min<int64_t>((int64_t)(part + 1) * p_->per, p_->total);
    const int64_t take = std::min<int64_t>(n - done, std::max<int64_t>(part_end - g, 1));
    p_->append_range(paths, lens, stride, done, done + take);
    done += take; p_->next += take;
  }
}

// Lines formatted on the device: split at the part boundaries (by walker count, as append does) and write the bytes.
void PathWriter::append_text(const char *text, const unsigned long long *off, int64_t n, unsigned long long base, int token) {
  if (token >= Impl::MAX_TOKENS) throw Error(SRW_ERR_INVALID, "append_text: token out of range");
  const bool async = token >= 0 && !p_->write_crc;          // (the .crc side files are computed in file order: synchronous)
  int64_t done = 0;
  while (done < n) {
    const int64_t g = p_->next;
    const int part = p_->per > 0 ? (int)std::min<int64_t>(g / p_->per, p_->n_parts - 1) : 0;
    if (part != p_->cur_part) p_->open_part(part);
    const int64_t part_end = std::min<int64_t>((int64_t)(part + 1) * p_->per, p_->total);
    const int64_t take = std::min<int64_t>(n - done, std::max<int64_t>(part_end - g, 1));
    const char *src = text + (off[done] - base);
    const size_t bytes = (size_t)(off[done + take] - off[done]);
    auto t0 = std::chrono::steady_clock::now();
    // measured on the GPU box (3.3 GB into the page cache): reserving the extent first and writing 64 MB per thread
    // (one or two threads per slice) takes ~295 ms; 2 MB per thread ~345 ms, 1 MB per thread ~585 ms (contention)
    if (!async) (void)posix_fallocate(p_->fd, p_->file_off, (off_t)bytes);      // (async: the reservation would queue behind the part's writers on the inode)
    if (async) {                                                  // one job per piece of a part
      // the part's extent is reserved ONCE, before its first writer exists (a reservation per piece would queue behind the writers on
      // the inode): what this piece's bytes per walker make of the walkers the part still gets, beyond the end of file, returned on close
      if (p_->part && p_->part->reserved == 0 && take > 0) {
        const off_t est = (off_t)((double)bytes / (double)take * (double)std::max<int64_t>(part_end - g, take) * 1.02) + 4096;
        if (fallocate(p_->fd, FALLOC_FL_KEEP_SIZE, p_->file_off, est) == 0) p_->part->reserved = p_->file_off + est; else p_->part->reserved = -1;
      }
      p_->enqueue(src, bytes, p_->file_off, token, p_->dir + "/" + p_->part_name(p_->cur_part));
      p_->file_off += (off_t)bytes;
      p_->t_write += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      done += take; p_->next += take;
      continue;
    }
    const int nt = (int)std::max<size_t>(1, std::min<size_t>(p_->hw, bytes / ((size_t)64 << 20) + 1));
    std::vector<std::string> errs((size_t)nt);
    const std::string fn = p_->dir + "/" + p_->part_name(p_->cur_part);
    {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) {
        const size_t b = bytes * (size_t)t / (size_t)nt, e = bytes * (size_t)(t + 1) / (size_t)nt;
        th.emplace_back([&, t, b, e] {
          try { pwrite_all(p_->fd, src + b, e - b, p_->file_off + (off_t)b, fn); }
          catch (const Error &er) { errs[(size_t)t] = er.what(); }
        });
      }
      for (auto &x : th) x.join();
    }
    for (auto &er : errs) if (!er.empty()) throw Error(SRW_ERR_IO, er);
    p_->file_off += (off_t)bytes;
    p_->crc_feed(src, bytes);
    p_->t_write += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    done += take; p_->next += take;
  }
}

void PathWriter::wait_token(int token) { p_->wait_token(token); }
bool PathWriter::token_idle(int token) { std::lock_guard<std::mutex> lk(p_->mu); return p_->pending[token] == 0; }

void PathWriter::close() {
  p_->wait_token(-1);                                          // everything queued is on its way to the disk; a failed write throws here
  p_->stop_pool();
  if (p_->cur_part < 0 && p_->n_parts > 0) p_->open_part(0);   // zero paths: still an (empty) part-00000
  if (getenv("SRW_TIMING"))
    fprintf(stderr, "[timing] writer: format %.1f ms, pwrite %.1f ms; pool: %lld jobs, %.1f GB, %.0f thread-ms in pwrite (%.1f GB/s per thread)\n", p_->t_format,
            p_->t_write, p_->pool_jobs, (double)p_->pool_bytes / 1e9, p_->pool_job_ms, p_->pool_job_ms > 0 ? (double)p_->pool_bytes / 1e6 / p_->pool_job_ms : 0.0);
  // parts that received no walker still exist as empty files, as with repartition(n)
  for (int part = p_->cur_part + 1; part < p_->n_parts; ++part) p_->open_part(part);
  p_->close_part();
  std::string ok = p_->dir + "/_SUCCESS";
  FILE *f = fopen(ok.c_str(), "wb");
  if (!f) throw Error(SRW_ERR_IO, "cannot write " + ok);
  fclose(f);
  if (p_->write_crc) write_crc_file(p_->dir, "_SUCCESS", std::string());
}

void write_path_files(const int32_t *paths, const int32_t *lens, int64_t n_walkers, int64_t stride,
                      const char *output_dir, int n_parts, bool write_crc) {
  PathWriter w(output_dir, n_parts, n_walkers, write_crc);
  w.append(paths, lens, n_walkers, stride);
  w.close();
}

}  // namespace srw

// ---- the embedding stage's output (Main.saveModelAndFeatures, M/Main.scala:77-97) -----------------------------------------------
namespace srw {
// java.lang.Float.toString: the shortest decimal that round-trips to the same float (the JDK's algorithm prints one digit more in
// rare cases: not reproduced), as d.ddd for 1e-3 <= |x| < 1e7 and as d.dddE[-]n otherwise; "NaN", "Infinity", "-Infinity", "0.0", "-0.0".
// (the digits: std::to_chars' shortest round-trip form — the same digits as the smallest %.{k}e that reads back as x, which is what this
//  function searched for with snprintf + strtof until round 5, at ~1 us per float: 134 M floats of a config-2 model took minutes)
void java_float_append(std::string &out, float x) {
  if (x != x) { out += "NaN"; return; }
  if (x == 0.0f) { out += std::signbit(x) ? "-0.0" : "0.0"; return; }
  if (std::isinf(x)) { out += x > 0 ? "Infinity" : "-Infinity"; return; }
  char buf[48];
  const auto res = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);     // [-]d[.ddd]e[+-]XX
  const char *p = buf, *end = res.ptr;
  if (*p == '-') { out.push_back('-'); ++p; }
  char digits[16]; int nd = 0;
  while (p < end && *p != 'e') { if (*p != '.') digits[nd++] = *p; ++p; }
  ++p;                                                                                           // 'e'
  int exp10 = 0; { bool eneg = false; if (*p == '-') { eneg = true; ++p; } else if (*p == '+') ++p; while (p < end) exp10 = exp10 * 10 + (*p++ - '0'); if (eneg) exp10 = -exp10; }
  while (nd > 1 && digits[nd - 1] == '0') --nd;
  const float ax = std::fabs(x);
  if (ax >= 1e-3f && ax < 1e7f) {
    if (exp10 >= 0) {
      for (int i = 0; i <= exp10; ++i) out.push_back(i < nd ? digits[i] : '0');
      out.push_back('.');
      if (nd > exp10 + 1) out.append(digits + exp10 + 1, (size_t)(nd - exp10 - 1)); else out.push_back('0');
    } else {
      out += "0.";
      out.append((size_t)(-exp10 - 1), '0');
      out.append(digits, (size_t)nd);
    }
  } else {
    out.push_back(digits[0]); out.push_back('.');
    if (nd > 1) out.append(digits + 1, (size_t)(nd - 1)); else out.push_back('0');
    out.push_back('E');
    out += std::to_string(exp10);
  }
}
std::string java_float_to_string(float x) { std::string s; java_float_append(s, x); return s; }

// saveModelAndFeatures (M/Main.scala:36-44): model.save(<out>/bin) FIRST (Spark fails there if the directory exists), then the
// "<word>\t<v0>\t..." lines under <out>/vec.  Nothing is overwritten: either directory existing is FileAlreadyExists before a byte
// is written; part files go through a 1 MiB buffer (vocab x dim x ~12 B of text never sits in memory as one string); a failure
// midway removes what this call created, so that a rerun does not trip over a half-written <out>/vec.
void write_vectors_named(const std::function<void(std::string &, int64_t)> &put_name, const float *vectors, int64_t n_vocab, int32_t dim,
                         const char *output_dir, int n_parts) {
  if (n_parts < 1) n_parts = 1;
  const std::string out(output_dir);
  const std::string vdir = out + "/vec", mdir = out + "/bin";
  struct stat sb;
  if (stat(mdir.c_str(), &sb) == 0) throw Error(SRW_ERR_EXISTS, "Output directory " + mdir + " already exists");
  if (stat(vdir.c_str(), &sb) == 0) throw Error(SRW_ERR_EXISTS, "Output directory " + vdir + " already exists");
  if (mkdir(out.c_str(), 0777) != 0 && errno != EEXIST) throw Error(SRW_ERR_IO, "cannot create " + out + ": " + strerror(errno));
  std::vector<std::string> made_files, made_dirs;
  auto mk = [&](const std::string &d) {
    if (mkdir(d.c_str(), 0777) != 0) throw Error(SRW_ERR_IO, "cannot create " + d + ": " + strerror(errno));
    made_dirs.push_back(d);
  };
  struct Sink {                                   // buffered part-file writer
    FILE *f = nullptr; std::string fn, buf;
    void open(const std::string &name) {
      fn = name; f = fopen(fn.c_str(), "wb");
      if (!f) throw Error(SRW_ERR_IO, "cannot open " + fn + ": " + strerror(errno));
      buf.clear(); buf.reserve((size_t)1 << 20);
    }
    void flush() { if (!buf.empty() && fwrite(buf.data(), 1, buf.size(), f) != buf.size()) throw Error(SRW_ERR_IO, "write error on " + fn); buf.clear(); }
    void close() { flush(); FILE *g = f; f = nullptr; if (fclose(g) != 0) throw Error(SRW_ERR_IO, "write error on " + fn); }
    ~Sink() { if (f) fclose(f); }
  };
  // the rows of a part, formatted by all host threads in batches (a batch's pieces are written in row order; at most a batch of text is
  // in memory), then through the part's buffer
  const int nt = (int)std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
  auto rows = [&](Sink &sk, int64_t r0, int64_t r1) {
    const int64_t per_thread = 1024;
    std::vector<std::string> piece((size_t)nt);
    for (int64_t b0 = r0; b0 < r1; b0 += per_thread * nt) {
      const int64_t b1 = std::min<int64_t>(r1, b0 + per_thread * nt);
      const int used = (int)((b1 - b0 + per_thread - 1) / per_thread);
      auto work = [&](int t) {
        std::string &o = piece[(size_t)t]; o.clear();
        for (int64_t r = b0 + t * per_thread; r < std::min<int64_t>(b1, b0 + (t + 1) * per_thread); ++r) {
          put_name(o, r);
          for (int32_t j = 0; j < dim; ++j) { o.push_back('\t'); java_float_append(o, vectors[r * dim + j]); }
          o.push_back('\n');
        }
      };
      if (used == 1) work(0);
      else {
        std::vector<std::thread> th;
        for (int t = 0; t < used; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
      }
      for (int t = 0; t < used; ++t) {
        sk.flush();
        if (fwrite(piece[(size_t)t].data(), 1, piece[(size_t)t].size(), sk.f) != piece[(size_t)t].size()) throw Error(SRW_ERR_IO, "write error on " + sk.fn);
      }
    }
  };
  auto file = [&](const std::string &fn, const std::string &text) {
    Sink sk; sk.open(fn); made_files.push_back(fn); sk.buf = text; sk.close();
  };
  try {
    // the model directory as Word2VecModel.save lays it out: metadata/ (one JSON line) + data/ (Parquet: parquet_model.cpp)
    mk(mdir); mk(mdir + "/metadata"); mk(mdir + "/data");
    file(mdir + "/metadata/part-00000", "{\"class\":\"org.apache.spark.mllib.feature.Word2VecModel\",\"version\":\"1.0\",\"vectorSize\":" + std::to_string(dim) +
                                            ",\"numWords\":" + std::to_string(n_vocab) + "}\n");
    file(mdir + "/metadata/_SUCCESS", "");
    made_files.push_back(mdir + "/data/part-00000.parquet");
    write_word2vec_parquet(made_files.back(), put_name, vectors, n_vocab, dim);       // (word: string, vector: array<float>), as Spark stores the model
    file(mdir + "/data/_SUCCESS", "");
    // repartition(numPartitions): contiguous blocks of the vocabulary (the reference's order of lines is unspecified)
    mk(vdir);
    const int64_t per = (n_vocab + n_parts - 1) / n_parts;
    for (int p = 0; p < n_parts; ++p) {
      char name[32]; snprintf(name, sizeof(name), "/part-%05d", p);
      Sink sk; sk.open(vdir + name); made_files.push_back(sk.fn);
      rows(sk, (int64_t)p * per, std::min<int64_t>(n_vocab, (int64_t)(p + 1) * per));
      sk.close();
    }
    file(vdir + "/_SUCCESS", "");
  } catch (...) {
    for (auto it = made_files.rbegin(); it != made_files.rend(); ++it) unlink(it->c_str());
    for (auto it = made_dirs.rbegin(); it != made_dirs.rend(); ++it) rmdir(it->c_str());
    throw;
  }
}
void write_vectors(const int32_t *vocab_ids, const float *vectors, int64_t n_vocab, int32_t dim, const char *output_dir, int n_parts) {
  write_vectors_named([&](std::string &b, int64_t r) { b += std::to_string(vocab_ids[r]); }, vectors, n_vocab, dim, output_dir, n_parts);
}
void write_vectors_words(const char *const *words, const float *vectors, int64_t n_vocab, int32_t dim, const char *output_dir, int n_parts) {
  write_vectors_named([&](std::string &b, int64_t r) { b += words[r]; }, vectors, n_vocab, dim, output_dir, n_parts);
}
}  // namespace srw
