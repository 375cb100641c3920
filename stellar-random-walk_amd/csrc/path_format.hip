// path_format.hip — device-side text formatter of the path output.
// Same bytes as writer.cpp / RandomWalk.save (M/algorithm/RandomWalk.scala:234-241): decimal ids joined by ONE TAB, no
// trailing tab, '\n' per line, canonical walker order.  Two kernels, one wave per walker, one lane per path slot:
//   k_fmt_len   : bytes of every line                      -> exclusive scan (rocPRIM) -> byte offset of every line
//   k_fmt_write : every lane writes its own number (+ separator) at  line offset + wave-prefix of the lengths
// The host then only copies the text out and pwrite()s it (run_walk_and_save with SRW_WALK_DEVICE_FORMAT).
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

// Pinned staging of the device formatter's output: two text slices (+ the line offsets of one chunk).
void ensure_pinned_text(srw_handle *h, size_t slice_cap, size_t n_off) {
  for (int i = 0; i < 2; ++i)
    if (h->pin_text_cap[i] < slice_cap) {
      if (h->pin_text[i]) (void)hipHostFree(h->pin_text[i]);
      h->pin_text[i] = nullptr;
      SRW_HIP(hipHostMalloc((void **)&h->pin_text[i], slice_cap, hipHostMallocDefault));
      h->pin_text_cap[i] = slice_cap;
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
  const size_t slice_cap = std::max<size_t>((size_t)64 << 20, (size_t)stride * 12 + 64);
  ensure_pinned_text(h, slice_cap, (size_t)chunk + 1);
  for (int64_t c0 = 0; c0 < n; c0 += chunk) {
    const int64_t m = std::min<int64_t>(chunk, n - c0);
    format_paths_device(h, h->res.paths.p + c0 * stride, h->res.lens.p + c0, m, stride, h->fmt_len[0].p, h->fmt_off[0].p,
                        h->fmt_text[0].p);
    SRW_HIP(hipMemcpyAsync(h->pin_off[0], h->fmt_off[0].p, ((size_t)m + 1) * 8, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
    const unsigned long long *off = h->pin_off[0];
    auto slice_end = [&](int64_t w0) {
      int64_t lo = w0 + 1, hi = m;
      while (lo < hi) { const int64_t mid = lo + (hi - lo + 1) / 2; if (off[mid] - off[w0] <= slice_cap) lo = mid; else hi = mid - 1; }
      return lo;
    };
    auto copy_slice = [&](int64_t w0, int64_t w1, int buf) {
      SRW_HIP(hipMemcpyAsync(h->pin_text[buf], h->fmt_text[0].p + off[w0], (size_t)(off[w1] - off[w0]), hipMemcpyDeviceToHost,
                             h->copy_stream));
    };
    int64_t w0 = 0, w1 = slice_end(0);
    int buf = 0;
    copy_slice(w0, w1, buf);
    while (w0 < m) {
      SRW_HIP(hipStreamSynchronize(h->copy_stream));
      const int64_t n0 = w1, n1 = n0 < m ? slice_end(n0) : n0;
      if (n0 < m) copy_slice(n0, n1, buf ^ 1);
      writer.append_text(h->pin_text[buf], off + w0, w1 - w0, off[w0]);
      w0 = n0; w1 = n1; buf ^= 1;
    }
  }
  writer.close();
  return true;
}

}  // namespace srw
