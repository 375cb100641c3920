// Scratch: a big two-column edge-list text file, fast (threads format their shares, one pwrite each): RMAT-ish skew is not needed to
// time the tokenizer — uniform ids below 2^scale.   g++ -O2 -pthread tools/gen_edgelist.cpp -o /tmp/gen && /tmp/gen /tmp/e.txt 26 1073741824
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>
static inline uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
int main(int argc, char **argv) {
  const char *fn = argv[1]; const int scale = atoi(argv[2]); const int64_t n = atoll(argv[3]);
  const int T = 64;
  std::vector<std::string> parts(T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; ++t)
    th.emplace_back([&, t] {
      std::string &s = parts[t]; s.reserve((size_t)(n / T + 1) * 18);
      char buf[64];
      for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
        const uint64_t h = mix((uint64_t)i);
        const int k = snprintf(buf, sizeof(buf), "%u %u\n", (unsigned)(h & ((1ull << scale) - 1)), (unsigned)((h >> 32) & ((1ull << scale) - 1)));
        s.append(buf, (size_t)k);
      }
    });
  for (auto &x : th) x.join();
  int fd = open(fn, O_WRONLY | O_CREAT | O_TRUNC, 0644);
  std::vector<off_t> off(T + 1, 0);
  for (int t = 0; t < T; ++t) off[t + 1] = off[t] + (off_t)parts[t].size();
  th.clear();
  for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t o = 0; while (o < parts[t].size()) { ssize_t k = pwrite(fd, parts[t].data() + o, parts[t].size() - o, off[t] + (off_t)o); if (k <= 0) exit(1); o += (size_t)k; } });
  for (auto &x : th) x.join();
  close(fd);
  printf("%s: %lld lines, %.2f GB\n", fn, (long long)n, (double)off[T] / 1e9);
  return 0;
}
