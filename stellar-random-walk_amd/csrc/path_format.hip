// path_format.hip — device-side text formatter of the path output.
// Same bytes as writer.cpp / RandomWalk.save (M/algorithm/RandomWalk.scala:234-241): decimal ids joined by ONE TAB, no
// trailing tab, '\n' per line, canonical walker order.  Two kernels, one wave per walker, one lane per path slot:
//   k_fmt_len   : bytes of every line                      -> exclusive scan (rocPRIM) -> byte offset of every line
//   k_fmt_write : every lane writes its own number (+ separator) at  line offset + wave-prefix of the lengths
// The host then only copies the text out and pwrite()s it (run_walk_and_save with SRW_WALK_DEVICE_FORMAT).
#include <chrono>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "engine.h"
#include "wave_primitives.h"

namespace srw {
namespace {
constexpr int FTPB = 256;

__device__ inline int dec_len(int32_t v) {
  uint32_t u = v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
  int n = 1;
  while (u >= 10u) { u /= 10u; ++n; }
  return n + (v < 0 ? 1 : 0);
}

__global__ void k_fmt_len(const int32_t *__restrict__ paths, const int32_t *__restrict__ lens, int64_t n, int64_t stride,
                          unsigned long long *__restrict__ out) {
  const int lane = lane_id();
  for (int64_t w = (blockIdx.x * (int64_t)FTPB + threadIdx.x) >> 6; w < n; w += (int64_t)gridDim.x * (FTPB / 64)) {
    const int32_t len = lens[w];
    unsigned long long b = 0;
    for (int32_t t = lane; t < len; t += 64) b += (unsigned long long)(dec_len(paths[w * stride + t]) + 1);   // + TAB or '\n'
    b = wave_sum_u64(b);
    if (lane == 0) out[w] = b ? b : 1ull;                       // an empty path would still be an empty line
  }
}

__global__ void k_fmt_write(const int32_t *__restrict__ paths, const int32_t *__restrict__ lens, int64_t n, int64_t stride,
                            const unsigned long long *__restrict__ off, char *__restrict__ text) {
  const int lane = lane_id();
  for (int64_t w = (blockIdx.x * (int64_t)FTPB + threadIdx.x) >> 6; w < n; w += (int64_t)gridDim.x * (FTPB / 64)) {
    const int32_t len = lens[w];
    char *line = text + off[w];
    if (len <= 0) { if (lane == 0) line[0] = '\n'; continue; }
    unsigned carry = 0;
    for (int32_t base = 0; base < len; base += 64) {
      const int32_t t = base + lane;
      const bool ok = t < len;
      const int32_t v = ok ? paths[w * stride + t] : 0;
      const unsigned mine = ok ? (unsigned)dec_len(v) + 1u : 0u;
      unsigned incl = mine;
      for (int o = 1; o < 64; o <<= 1) { unsigned x = __shfl_up(incl, o); if (lane >= o) incl += x; }
      if (ok) {
        char *p = line + carry + (incl - mine);
        uint32_t u = v < 0 ? (uint32_t)(-(int64_t)v) : (uint32_t)v;
        const int nd = (int)mine - 1;                            // characters of the number, sign included
        if (v < 0) p[0] = '-';
        for (int i = nd - 1; i >= (v < 0 ? 1 : 0); --i) { p[i] = (char)('0' + u % 10u); u /= 10u; }
        p[nd] = (t == len - 1) ? '\n' : '\t';
      }
      carry += (unsigned)__shfl((int)incl, 63);
    }
  }
}
}  // namespace

// Formats walkers [0, n) of (d_paths, d_lens) into d_text; d_off[0..n] receives the byte offset of every line and the
// total.  Returns the text capacity needed (callers size d_text with format_capacity first).  Stream-ordered.
size_t format_capacity(int64_t n, int64_t stride, int32_t vmin, int32_t vmax) {
  // every number takes at most (digits of the largest magnitude + sign) characters plus one separator
  auto digits = [](int64_t v) { int d = 1; if (v < 0) v = -v; while (v >= 10) { v /= 10; ++d; } return d; };
  const int per = std::max(digits(vmin) + (vmin < 0 ? 1 : 0), digits(vmax) + (vmax < 0 ? 1 : 0)) + 1;
  return (size_t)n * (size_t)stride * (size_t)per + 16;
}

void format_paths_device(srw_handle *h, const int32_t *d_paths, const int32_t *d_lens, int64_t n, int64_t stride,
                         unsigned long long *d_len_bytes, unsigned long long *d_off, char *d_text) {
  hipStream_t st = h->stream;
  if (n <= 0) return;
  const int blocks = (int)std::min<int64_t>((n * 64 + FTPB - 1) / FTPB, (int64_t)h->n_cus * 16);
  hipLaunchKernelGGL(k_fmt_len, dim3(blocks), dim3(FTPB), 0, st, d_paths, d_lens, n, stride, d_len_bytes);
  // exclusive scan over n + 1 items (the extra zero item yields the total in d_off[n])
  SRW_HIP(hipMemsetAsync(d_len_bytes + n, 0, 8, st));
  size_t tb = 0;
  SRW_HIP(rocprim::exclusive_scan(nullptr, tb, d_len_bytes, d_off, 0ull, (size_t)n + 1, rocprim::plus<unsigned long long>(), st));
  h->fmt_temp.ensure(tb);
  SRW_HIP(rocprim::exclusive_scan((void *)h->fmt_temp.p, tb, d_len_bytes, d_off, 0ull, (size_t)n + 1,
                                  rocprim::plus<unsigned long long>(), st));
  hipLaunchKernelGGL(k_fmt_write, dim3(blocks), dim3(FTPB), 0, st, d_paths, d_lens, n, stride, d_off, d_text);
  SRW_HIP(hipGetLastError());
}

// Pinned staging of the device formatter's output: a ring of text slices (+ the line offsets of one chunk).
void ensure_pinned_text(srw_handle *h, size_t slice_cap, size_t n_off) {
  for (int i = 0; i < srw_handle::PIN_RING; ++i) {
    if (h->pin_text_cap[i] < slice_cap) {
      if (h->pin_text[i]) (void)hipHostFree(h->pin_text[i]);
      h->pin_text[i] = nullptr;
      SRW_HIP(hipHostMalloc((void **)&h->pin_text[i], slice_cap, hipHostMallocDefault));
      h->pin_text_cap[i] = slice_cap;
    }
    if (!h->pin_copied[i]) SRW_HIP(hipEventCreateWithFlags(&h->pin_copied[i], hipEventDisableTiming));
  }
  if (h->pin_off_cap < n_off) {
    for (int i = 0; i < 2; ++i) {
      if (h->pin_off[i]) (void)hipHostFree(h->pin_off[i]);
      h->pin_off[i] = nullptr;
      SRW_HIP(hipHostMalloc((void **)&h->pin_off[i], n_off * 8, hipHostMallocDefault));
    }
    h->pin_off_cap = n_off;
  }
}
// Pinned memory costs ~0.2 ms/MB to allocate and ~0.1 ms/MB to free on this stack, so the text leaves the device in slices of <= 64 MB
// of whole lines — smaller ones for a small output (text_bytes: an upper bound of what one drain moves), so that a karate-sized job
// does not pin 384 MB.  SRW_TEXT_SLICE_KB (tests): the slice size, so that a small graph takes the ring around several times.
size_t text_slice_cap(int64_t stride, size_t text_bytes) {
  size_t cap = std::min<size_t>((size_t)64 << 20, std::max<size_t>((size_t)1 << 20, text_bytes / (2 * srw_handle::PIN_RING)));
  if (const char *e = getenv("SRW_TEXT_SLICE_KB"); e && atoi(e) > 0) cap = (size_t)atoi(e) << 10;
  return std::max<size_t>(cap, (size_t)stride * 12 + 64);
}

// The text of [0, n) walkers leaves d_text through the ring: up to PIN_RING slices are between the copy stream and the writer's
// threads at any time — slice j + 1 .. j + 5 are copied while slice j is written, and slices that fall into different part files are
// written in parallel (writer.cpp).  A ring slot is reused when the writer has released it (wait_token).
void drain_text(srw_handle *h, PathWriter &writer, const char *d_text, const unsigned long long *off, int64_t n, size_t slice_cap,
                const std::function<void()> &all_copied) {
  constexpr int K = srw_handle::PIN_RING;
  auto slice_end = [&](int64_t w0) {                       // largest w1 > w0 with off[w1] - off[w0] <= slice_cap
    int64_t lo = w0 + 1, hi = n;
    while (lo < hi) { const int64_t mid = lo + (hi - lo + 1) / 2; if (off[mid] - off[w0] <= slice_cap) lo = mid; else hi = mid - 1; }
    return lo;
  };
  struct Slice { int64_t w0, w1; };
  Slice ring[K];
  long long issued = 0, handed = 0;
  int64_t w_next = 0;
  const bool timing = getenv("SRW_TIMING") != nullptr;
  double t_wait_writer = 0, t_wait_copy = 0, t_hand = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  // a slot goes free -> being copied into -> copied -> with the writer's threads -> free.  The host never blocks on a slot while a copied
  // slice waits to be handed over (else the writes of the ring's slices would run one after the other)
  for (;;) {
    while (w_next < n && issued - handed < K && writer.token_idle((int)(issued % K))) {
      const int buf = (int)(issued % K);
      const int64_t w1 = slice_end(w_next);
      SRW_HIP(hipMemcpyAsync(h->pin_text[buf], d_text + off[w_next], (size_t)(off[w1] - off[w_next]), hipMemcpyDeviceToHost, h->copy_stream));
      SRW_HIP(hipEventRecord(h->pin_copied[buf], h->copy_stream));
      ring[buf] = Slice{w_next, w1};
      w_next = w1; ++issued;
    }
    if (handed < issued) {
      const int buf = (int)(handed % K);
      const auto t0 = now();
      SRW_HIP(hipEventSynchronize(h->pin_copied[buf]));       // slice `handed` is in pin_text[buf]
      const auto t1 = now();
      ++handed;
      if (handed == issued && w_next >= n && all_copied) all_copied();
      writer.append_text(h->pin_text[buf], off + ring[buf].w0, ring[buf].w1 - ring[buf].w0, off[ring[buf].w0], buf);
      t_wait_copy += ms(t0, t1); t_hand += ms(t1, now());
      continue;
    }
    if (w_next >= n) break;
    const auto t0 = now();
    writer.wait_token((int)(issued % K));                     // every slot is with the writer: wait for the oldest
    t_wait_writer += ms(t0, now());
  }
  if (timing)
    fprintf(stderr, "[timing] text drain: %lld slices, %.1f GB; host waited %.0f ms for copies, %.0f ms for the writer's threads, %.0f ms handing over\n",
            issued, n > 0 ? (double)(off[n] - off[0]) / 1e9 : 0.0, t_wait_copy, t_wait_writer, t_hand);
  if (n <= 0 && all_copied) all_copied();
}

// srw_write_paths on a device-resident result: formatted on the GPU chunk by chunk, copied out in slices of whole lines
// through the two pinned buffers (slice j + 1 in flight while slice j is written).  false = not enough HBM for a chunk's
// text: the caller formats on the host.
bool write_result_device(srw_handle *h, const char *output_dir, int n_parts, bool write_crc) {
  const int64_t n = h->res.n_walkers, stride = h->res.stride;
  Graph &g = h->g;
  const size_t per_walker = format_capacity(1, stride, g.id_lo, g.id_hi);
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)(((size_t)1 << 30) / per_walker)));
  const size_t cap = (size_t)chunk * per_walker + 16;
  size_t free_b = 0, total_b = 0;
  SRW_HIP(hipMemGetInfo(&free_b, &total_b));
  if (h->fmt_text[0].n < cap && free_b < cap + (size_t)chunk * 16 + ((size_t)2 << 30)) return false;
  PathWriter writer(output_dir, n_parts, n, write_crc);
  if (n == 0) { writer.close(); return true; }
  hipStream_t st = h->stream;
  if (!h->copy_stream) SRW_HIP(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  h->fmt_text[0].ensure(cap); h->fmt_len[0].ensure((size_t)chunk + 1); h->fmt_off[0].ensure((size_t)chunk + 1);
  const size_t slice_cap = text_slice_cap(stride, (size_t)n * per_walker);
  ensure_pinned_text(h, slice_cap, (size_t)chunk + 1);
  for (int64_t c0 = 0; c0 < n; c0 += chunk) {
    const int64_t m = std::min<int64_t>(chunk, n - c0);
    format_paths_device(h, h->res.paths.p + c0 * stride, h->res.lens.p + c0, m, stride, h->fmt_len[0].p, h->fmt_off[0].p,
                        h->fmt_text[0].p);
    SRW_HIP(hipMemcpyAsync(h->pin_off[0], h->fmt_off[0].p, ((size_t)m + 1) * 8, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
    drain_text(h, writer, h->fmt_text[0].p, h->pin_off[0], m, slice_cap, nullptr);
  }
  writer.close();
  return true;
}

}  // namespace srw
