#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  const char *fn = argv[1]; const size_t total = (size_t)atoi(argv[2]) << 30, slice = (size_t)64 << 20;
  char *src; if (posix_memalign((void **)&src, 4096, slice)) return 1;
  for (size_t i = 0; i < slice; ++i) src[i] = (char)('0' + i % 10);
  for (int nt : {1, 4, 16}) {
    unlink(fn);
    int fd = open(fn, O_RDWR | O_CREAT | O_TRUNC | O_DIRECT, 0644);
    if (fd < 0) { perror("open O_DIRECT"); return 1; }
    if (posix_fallocate(fd, 0, (off_t)total)) perror("fallocate");
    const double t0 = now();
    for (size_t off = 0; off < total; off += slice) {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t) {
        const size_t b = slice / nt * t, e = slice / nt * (t + 1);
        th.emplace_back([=] { size_t o = b; while (o < e) { ssize_t k = pwrite(fd, src + o, e - o, (off_t)(off + o)); if (k <= 0) { perror("pwrite"); exit(1); } o += (size_t)k; } });
      }
      for (auto &x : th) x.join();
    }
    close(fd);
    const double dt = now() - t0;
    printf("O_DIRECT pwrite %2d threads: %.2f s = %.1f GB/s\n", nt, dt, (double)total / dt / 1e9); fflush(stdout);
  }
  unlink(fn);
}
