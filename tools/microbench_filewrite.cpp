// Scratch: how fast can one process put text into ONE file of the page cache?  (the end-to-end path of srw_walk_and_save is bound by it:
// 22 GB of path text per iteration of the headline graph).  g++ -O2 -pthread tools/microbench_filewrite.cpp -o /tmp/fw && /tmp/fw /tmp/fw.dat 8
//   pwrite  N threads, each a contiguous share of every 64 MiB slice (what writer.cpp:append_text does, with N = 2)
//   mmap    the file ftruncate()d to its size and mapped MAP_SHARED; N threads memcpy their share of every slice
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
#include <algorithm>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const char *fn = argc > 1 ? argv[1] : "/tmp/fw.dat";
  const size_t gb = argc > 2 ? (size_t)atoi(argv[2]) : 4, total = gb << 30, slice = (size_t)64 << 20;
  char *src = (char *)malloc(slice);
  for (size_t i = 0; i < slice; ++i) src[i] = (char)('0' + i % 10);
  for (int mode = 0; mode < (getenv("FW_ALL") ? 3 : 1); ++mode)                       // 0 pwrite, 1 mmap, 2 mmap + MAP_POPULATE per slice (madvise)
    for (int nt : {1, 2, 4, 8, 16, 32}) {
      unlink(fn);
      int fd = open(fn, O_RDWR | O_CREAT | O_TRUNC, 0644);
      if (fd < 0) { perror("open"); return 1; }
      const double t0 = now();
      char *map = nullptr;
      if (mode) {
        if (ftruncate(fd, (off_t)total) != 0) { perror("ftruncate"); return 1; }
        map = (char *)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) { perror("mmap"); return 1; }
      } else (void)posix_fallocate(fd, 0, (off_t)total);
      for (size_t off = 0; off < total; off += slice) {
        if (mode == 2) (void)madvise(map + off, slice, MADV_POPULATE_WRITE);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) {
          const size_t b = slice * (size_t)t / (size_t)nt, e = slice * (size_t)(t + 1) / (size_t)nt;
          th.emplace_back([=] {
            if (mode) memcpy(map + off + b, src + b, e - b);
            else { size_t o = b; while (o < e) { ssize_t k = pwrite(fd, src + o, e - o, (off_t)(off + o)); if (k <= 0) { perror("pwrite"); exit(1); } o += (size_t)k; } }
          });
        }
        for (auto &x : th) x.join();
      }
      if (map) munmap(map, total);
      close(fd);
      const double dt = now() - t0;
      printf("%-8s %2d threads: %.2f s = %.1f GB/s\n", mode == 0 ? "pwrite" : mode == 1 ? "mmap" : "mmap+pop", nt, dt, (double)total / dt / 1e9);
      fflush(stdout);
    }
  // several FILES at once, one writer thread each (part files written concurrently)
  for (int nf : {1, 2, 4, 8, 16}) {
    std::vector<std::thread> th;
    const double t0 = now();
    for (int f = 0; f < nf; ++f)
      th.emplace_back([=] {
        char name[512]; snprintf(name, sizeof(name), "%s.%d", fn, f);
        unlink(name);
        int fd = open(name, O_RDWR | O_CREAT | O_TRUNC, 0644);
        const size_t share = total / (size_t)nf;
        (void)posix_fallocate(fd, 0, (off_t)share);
        for (size_t off = 0; off < share; off += slice) {
          size_t o = 0; const size_t e = std::min(slice, share - off);
          while (o < e) { ssize_t k = pwrite(fd, src + o, e - o, (off_t)(off + o)); if (k <= 0) { perror("pwrite"); exit(1); } o += (size_t)k; }
        }
        close(fd);
      });
    for (auto &x : th) x.join();
    const double dt = now() - t0;
    printf("%2d files, one pwrite thread each: %.2f s = %.1f GB/s\n", nf, dt, (double)total / dt / 1e9);
    for (int f = 0; f < nf; ++f) { char name[512]; snprintf(name, sizeof(name), "%s.%d", fn, f); unlink(name); }
  }
  unlink(fn);
  return 0;
}
