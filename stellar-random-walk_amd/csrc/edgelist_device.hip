// edgelist_device.hip — device-side tokenizer for the common edge-list shape:
// every line is "<int><blanks><int>[blanks]" or, with `weighted`, "<int><blanks><int><blanks><simple decimal>[blanks]", over the
// characters 0-9 + - . space TAB; lines end in '\n'.  A simple decimal is [sign] digits [. digits] with at most 7
// significant digits and at most 10 fraction digits: then M / 10^k (M < 10^7 < 2^24, 10^k exact) evaluated in f64 and
// rounded to f32 IS the correctly rounded binary32 value Float.parseFloat returns — a decimal this short is never
// within 2^-48 (relative) of a binary32 rounding boundary unless it sits on it, so the f64 quotient cannot round
// across one (argument in DESIGN.md §4.1; checked against strtof by the tests).
// Same acceptance rules as the host tokenizer (edgelist.cpp) and UniformRandomWalk.loadGraph
// (M/algorithm/UniformRandomWalk.scala:23-43) on that shape: Java split("\\s+") (leading blank => empty first token
// => NumberFormatException), Integer.parseInt (optional single sign, >= 1 digit, int32 range).  ANYTHING else — a
// third column (weights / partition ids), CR, other characters, an empty line, an overflow — makes the function
// return false and the caller runs the host tokenizer, which parses it or raises the reference's error.
//   k_nl_count : newline count per 4 KB block + character check      -> exclusive scan (rocPRIM)
//   k_nl_pos   : byte position of every newline
//   k_parse    : one thread per line -> src[], dst[], min / max id
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstring>
#include <rocprim/rocprim.hpp>

#include "engine.h"
#include "wave_primitives.h"

namespace srw {
namespace {
constexpr int TTPB = 256;
constexpr int BYTES_PER_THREAD = 16;
constexpr int64_t BLOCK_BYTES = (int64_t)TTPB * BYTES_PER_THREAD;

__constant__ double kPow10[11] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10};   // all exact in f64

__device__ inline bool tok_char_ok(unsigned char c) {
  return (c >= '0' && c <= '9') || c == ' ' || c == '\t' || c == '\n' || c == '-' || c == '+' || c == '.';
}

__global__ __launch_bounds__(TTPB) void k_nl_count(const unsigned char *__restrict__ text, int64_t size,
                                                   uint32_t *__restrict__ blk, uint32_t *err) {
  __shared__ uint32_t cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const int64_t b0 = blockIdx.x * BLOCK_BYTES + (int64_t)threadIdx.x * BYTES_PER_THREAD;
  uint32_t c = 0; bool bad = false;
  for (int i = 0; i < BYTES_PER_THREAD; ++i) {
    const int64_t p = b0 + i;
    if (p < size) { const unsigned char ch = text[p]; c += ch == '\n'; bad |= !tok_char_ok(ch); }
  }
  c = (uint32_t)wave_sum_u64(c);
  if (lane_id() == 0 && c) atomicAdd(&cnt, c);
  if (__any(bad) && lane_id() == 0) atomicOr(err, 1u);
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = cnt;
}

__global__ __launch_bounds__(TTPB) void k_nl_pos(const unsigned char *__restrict__ text, int64_t size,
                                                 const uint32_t *__restrict__ blkoff, int64_t *__restrict__ nlpos) {
  __shared__ uint32_t wsum[TTPB / 64];
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const int64_t b0 = blockIdx.x * BLOCK_BYTES + (int64_t)threadIdx.x * BYTES_PER_THREAD;
  uint32_t c = 0;
  for (int i = 0; i < BYTES_PER_THREAD; ++i) { const int64_t p = b0 + i; if (p < size) c += text[p] == '\n'; }
  uint32_t incl = c;
  for (int o = 1; o < 64; o <<= 1) { uint32_t x = __shfl_up(incl, o); if (lane >= o) incl += x; }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  uint32_t base = blkoff[blockIdx.x];
  for (int w = 0; w < wv; ++w) base += wsum[w];
  uint32_t r = base + incl - c;
  for (int i = 0; i < BYTES_PER_THREAD; ++i) { const int64_t p = b0 + i; if (p < size && text[p] == '\n') nlpos[r++] = p; }
}

// (__noinline__ on purpose: with both token parsers inlined into k_parse, hipcc 7.2 -O3 produced a kernel that stored 0 for
// the first column — same source, tests/test_gpu_parity.py::test_device_tokenizer_* catch it; as calls they are correct)
__device__ __noinline__ bool parse_int_token(const unsigned char *t, int64_t &p, int64_t end, int32_t &out) {
  bool neg = false;
  if (p < end && (t[p] == '-' || t[p] == '+')) { neg = t[p] == '-'; ++p; }
  if (p >= end || t[p] < '0' || t[p] > '9') return false;          // "", "-", "+", "-x"
  int64_t v = 0;
  while (p < end && t[p] >= '0' && t[p] <= '9') {
    v = v * 10 + (t[p] - '0');
    if (v > 2147483648ll) return false;                              // beyond int32 either way
    ++p;
  }
  if (neg) v = -v;
  if (v > 2147483647ll) return false;
  if (p < end && t[p] != ' ' && t[p] != '\t') return false;       // "12-3", "1+2": one token, not an int
  out = (int32_t)v;
  return true;
}

// [sign] digits [. digits], <= 7 significant digits, <= 10 fraction digits, delimited by a blank or the line end
__device__ __noinline__ bool parse_simple_decimal(const unsigned char *t, int64_t &p, int64_t end, float &out) {
  bool neg = false;
  if (p < end && (t[p] == '-' || t[p] == '+')) { neg = t[p] == '-'; ++p; }
  int64_t M = 0; int sig = 0, frac = 0, ndig = 0;
  bool seen_dot = false;
  while (p < end) {
    const unsigned char c = t[p];
    if (c >= '0' && c <= '9') {
      ++ndig;
      if (sig > 0 || c != '0') ++sig;
      if (sig > 7) return false;
      M = M * 10 + (c - '0');
      if (seen_dot && ++frac > 10) return false;
    } else if (c == '.' && !seen_dot) seen_dot = true;
    else break;
    ++p;
  }
  if (ndig == 0) return false;                                         // ".", "-", "": unparsable -> host (1.0f there)
  if (p < end && t[p] != ' ' && t[p] != '\t') return false;
  const float v = (float)((double)M / kPow10[frac]);
  out = neg ? -v : v;
  return true;
}

__global__ void k_parse(const unsigned char *__restrict__ text, int64_t size, const int64_t *__restrict__ nlpos,
                        int64_t n_nl, int64_t n_lines, int32_t *__restrict__ src, int32_t *__restrict__ dst,
                        float *__restrict__ wout, int32_t *minmax, uint32_t *err) {
  int32_t lo = 2147483647, hi = -2147483647 - 1;
  bool bad = false;
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < n_lines; j += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = j == 0 ? 0 : nlpos[j - 1] + 1;
    const int64_t end = j < n_nl ? nlpos[j] : size;
    int32_t a = 0, b = 0;
    bool ok = parse_int_token(text, p, end, a);                      // a leading blank fails here, as in Java
    if (ok) {
      const int64_t q = p;
      while (p < end && (text[p] == ' ' || text[p] == '\t')) ++p;
      ok = p > q && parse_int_token(text, p, end, b);
    }
    float w = 1.0f;                                                   // two columns: weight 1.0f (:29-32)
    if (ok) {
      while (p < end && (text[p] == ' ' || text[p] == '\t')) ++p;    // trailing blanks are dropped by split
      if (p < end && wout) {                                          // weighted: the third (= last) column
        ok = parse_simple_decimal(text, p, end, w);
        while (ok && p < end && (text[p] == ' ' || text[p] == '\t')) ++p;
      }
      ok = ok && p == end;                                            // anything further -> host tokenizer
    }
    if (!ok) { bad = true; continue; }
    src[j] = a; dst[j] = b;
    if (wout) wout[j] = w;
    lo = min(lo, min(a, b)); hi = max(hi, max(a, b));
  }
  lo = wave_min_i32(lo); hi = wave_max_i32(hi);
  if (lane_id() == 0) { atomicMin(&minmax[0], lo); atomicMax(&minmax[1], hi); }
  if (__any(bad) && lane_id() == 0) atomicOr(err, 1u);
}
}  // namespace

bool load_edgelist_device(srw_handle *h, const char *path, bool directed, bool weighted) {
  { const size_t len = strlen(path); if (len > 3 && strcmp(path + len - 3, ".gz") == 0) return false; }      // compressed: the host tokenizer inflates it
  int fd = open(path, O_RDONLY);
  if (fd < 0) return false;                                           // the host path reports the error
  struct stat sb;
  if (fstat(fd, &sb) != 0 || sb.st_size <= 0) { close(fd); return false; }
  const int64_t size = (int64_t)sb.st_size;
  {   // the text, the newline positions (8 B per line) and the parsed columns (8-12 B per line) are in HBM together: ~2.2x the file at
      // 17 bytes per line — the 1 B-edge graph's 19 GB file fits many times over (rounds 1-4 stopped at 8 GB and sent it to the host
      // tokenizer); a file that does not fit next to a reserve goes to the host tokenizer, which streams it.  SRW_DEVICE_TOKENIZER_MAX_MB: tests.
    size_t free_b = 0, total_b = 0;
    if (const hipError_t me = hipMemGetInfo(&free_b, &total_b); me != hipSuccess) { close(fd); SRW_HIP(me); }
    uint64_t cap = free_b > ((size_t)16 << 30) ? (uint64_t)((free_b - ((size_t)16 << 30)) / 3) : 0;
    if (const char *e = getenv("SRW_DEVICE_TOKENIZER_MAX_MB"); e && *e) cap = (uint64_t)atoll(e) << 20;
    if ((uint64_t)size > cap) { close(fd); return false; }
  }
  const unsigned char *data = (const unsigned char *)mmap(nullptr, (size_t)size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (data == MAP_FAILED) return false;
  if (size >= 3 && data[0] == 0xEF && data[1] == 0xBB && data[2] == 0xBF) { munmap((void *)data, (size_t)size); return false; }   // byte order mark: the host tokenizer skips it (as Hadoop does)
  {   // cheap shape check on the first line: exactly two tokens, else do not even upload
    int64_t e = 0; int tokens = 0; bool in_tok = false;
    while (e < size && data[e] != '\n' && e < 4096) { const bool ws = data[e] == ' ' || data[e] == '\t'; if (!ws && !in_tok) ++tokens; in_tok = !ws; ++e; }
    if (tokens != 2 && !(weighted && tokens == 3)) { munmap((void *)data, (size_t)size); return false; }
  }
  hipStream_t st = h->stream;
  struct Unmap { const unsigned char *p; size_t n; ~Unmap() { if (p) munmap((void *)p, n); } } unmap{data, (size_t)size};
  DevBuf<unsigned char> d_text;
  try { d_text.alloc((size_t)size); }
  catch (const Error &) { return false; }        // no room for the text in HBM: the host tokenizer streams it instead
  hipError_t ce = hipMemcpyAsync(d_text.p, data, (size_t)size, hipMemcpyHostToDevice, st);
  if (ce == hipSuccess) ce = hipStreamSynchronize(st);
  munmap((void *)data, (size_t)size); unmap.p = nullptr;
  SRW_HIP(ce);
  const int64_t n_blocks = (size + BLOCK_BYTES - 1) / BLOCK_BYTES;
  DevBuf<uint32_t> blk, blkoff, flags; DevBuf<char> temp; DevBuf<int32_t> minmax;
  blk.alloc((size_t)n_blocks + 1); blkoff.alloc((size_t)n_blocks + 1); flags.alloc(1); minmax.alloc(2);
  SRW_HIP(hipMemsetAsync(flags.p, 0, 4, st));
  SRW_HIP(hipMemsetAsync(blk.p + n_blocks, 0, 4, st));
  hipLaunchKernelGGL(k_nl_count, dim3((unsigned)n_blocks), dim3(TTPB), 0, st, d_text.p, size, blk.p, flags.p);
  if (size >= ((int64_t)1 << 32) || getenv("SRW_TOKENIZER_COUNT64")) {      // (the variable: tests take this branch on a small file)
    // a file this long can hold 2^32 newlines or more: the 32-bit scan below would wrap, the 2^31-line guard would pass on the wrapped
    // count and k_nl_pos would write past nlpos.  The total in 64 bits first; too many lines -> the host tokenizer (which streams).
    DevBuf<unsigned long long> tot; tot.alloc(1);
    auto wide = rocprim::make_transform_iterator(blk.p, [] __device__(uint32_t c) { return (unsigned long long)c; });
    size_t rb = 0;
    SRW_HIP(rocprim::reduce(nullptr, rb, wide, tot.p, 0ull, (size_t)n_blocks, rocprim::plus<unsigned long long>(), st));
    DevBuf<char> rtemp; rtemp.alloc(rb);
    SRW_HIP(rocprim::reduce((void *)rtemp.p, rb, wide, tot.p, 0ull, (size_t)n_blocks, rocprim::plus<unsigned long long>(), st));
    unsigned long long n_nl64 = 0;
    SRW_HIP(hipMemcpyAsync(&n_nl64, tot.p, 8, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
    if (n_nl64 >= (1ull << 31) - 1ull) return false;
  }
  size_t tb = 0;
  SRW_HIP(rocprim::exclusive_scan(nullptr, tb, blk.p, blkoff.p, 0u, (size_t)n_blocks + 1, rocprim::plus<uint32_t>(), st));
  temp.alloc(tb);
  SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, blk.p, blkoff.p, 0u, (size_t)n_blocks + 1, rocprim::plus<uint32_t>(), st));
  uint32_t n_nl32 = 0, bad = 0; unsigned char last = 0;
  SRW_HIP(hipMemcpyAsync(&n_nl32, blkoff.p + n_blocks, 4, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipMemcpyAsync(&bad, flags.p, 4, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipMemcpyAsync(&last, d_text.p + (size - 1), 1, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  if (bad) return false;
  const int64_t n_nl = n_nl32, n_lines = n_nl + (last != '\n' ? 1 : 0);
  if (n_lines <= 0 || n_lines >= ((int64_t)1 << 31)) return false;
  DevBuf<int64_t> nlpos; nlpos.alloc((size_t)std::max<int64_t>(n_nl, 1));
  hipLaunchKernelGGL(k_nl_pos, dim3((unsigned)n_blocks), dim3(TTPB), 0, st, d_text.p, size, blkoff.p, nlpos.p);
  DevBuf<int32_t> d_src, d_dst; DevBuf<float> d_w;
  d_src.alloc((size_t)n_lines); d_dst.alloc((size_t)n_lines);
  if (weighted) d_w.alloc((size_t)n_lines);
  const int32_t init[2] = {2147483647, -2147483647 - 1};
  SRW_HIP(hipMemcpyAsync(minmax.p, init, 8, hipMemcpyHostToDevice, st));
  const int gp = (int)std::min<int64_t>((n_lines + TTPB - 1) / TTPB, (int64_t)h->n_cus * 16);
  hipLaunchKernelGGL(k_parse, dim3(gp), dim3(TTPB), 0, st, d_text.p, size, nlpos.p, n_nl, n_lines, d_src.p, d_dst.p,
                     weighted ? d_w.p : nullptr, minmax.p, flags.p);
  int32_t mm[2];
  SRW_HIP(hipMemcpyAsync(mm, minmax.p, 8, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipMemcpyAsync(&bad, flags.p, 4, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  SRW_HIP(hipGetLastError());
  if (bad) return false;
  d_text.release(); nlpos.release();
  IdMap idmap;                                   // a sparse id space is compacted to ranks (graph_build.hip:compact_ids)
  if (ids_are_sparse(h, 2 * n_lines, mm[0], mm[1])) compact_ids(h, d_src.p, d_dst.p, n_lines, mm[0], mm[1], idmap);
  build_graph_from_device_lines(h, d_src.p, d_dst.p, weighted ? d_w.p : nullptr, n_lines, directed, mm[0], mm[1], nullptr, &idmap);
  h->g.part_of.clear();
  return true;
}

}  // namespace srw
