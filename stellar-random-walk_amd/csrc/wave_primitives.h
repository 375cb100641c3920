// wave_primitives.h — 64-lane wavefront building blocks for the exact (reference-bit-identical) sampler.
//
// The reference's RandomSample.sample (M/algorithm/RandomSample.scala:12-25) is two LEFT-TO-RIGHT f64
// accumulations.  A parallel scan rounds differently in the last ulps and flips the chosen neighbor whenever
// the 24-bit uniform lands on a CDF boundary (SURVEY §8c "rounding known-answers"), so:
//   * the sum  S = foldLeft(0.0)(_ + w)  is taken in parallel ONLY when a certificate proves that every
//     partial sum in ANY order is exactly representable (then order cannot matter);
//   * the running CDF  acc += w / S  is evaluated as a wave-uniform sequential chain: the 64 quotients of a
//     chunk are computed in parallel (the expensive part: loads, bias, membership, f64 divide), then added in
//     lane order with v_readlane + v_add_f64 — the same additions, in the same order, as the reference.
#pragma once
#include "device_common.h"

namespace srw {

__device__ inline int lane_id() { return (int)(threadIdx.x & 63u); }

__device__ inline double readlane_f64(double v, int lane) {
  uint64_t b = (uint64_t)__double_as_longlong(v);
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// A value the whole wave holds (one walker per wave: rows, degrees, table geometry): back into SGPRs, so that the arithmetic that
// follows is scalar — the table kernels are bound by VALU issue (profiles/r04_valu_issue.md), a wave-uniform integer op on the
// vector unit costs a 64-lane instruction slot.
__device__ inline int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline int64_t uni(int64_t v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
template <typename T> __device__ inline const T *uni(const T *p) { return reinterpret_cast<const T *>((uintptr_t)uni((int64_t)(uintptr_t)p)); }

// ... and the other way: once the scalar unit is the busier one (448 scalar against 302 vector instructions per table step,
// profiles/r04_valu_issue.md) a self-contained piece of wave-uniform integer arithmetic is cheaper on the vector unit: on_vector() hides
// the value's uniformity from the compiler, what is computed from it runs as vector instructions, uni() brings the results back.
__device__ inline int32_t on_vector(int32_t v) { int32_t r; asm("v_mov_b32 %0, %1" : "=v"(r) : "s"(v)); return r; }

// v's lane j (wave-uniform) = val (wave-uniform).  (v_writelane takes one SGPR operand on gfx9: the lane index would go through m0,
// which inline asm may not clobber — a compare and a select are two vector instructions.)
__device__ inline void write_lane(int32_t &v, int32_t val, int j) { v = (int)(threadIdx.x & 63u) == j ? val : v; }

// (the results are the wave's: handed back through SGPRs — uni — so that the compiler sees the branches they decide as uniform)
__device__ inline double wave_sum_f64(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return __longlong_as_double((long long)uni((int64_t)__double_as_longlong(v)));
}
// min / max over the wave by DPP (row shifts + row broadcasts, like sampling.h:wave_incl_scan_f64: lane 63 ends up with the whole
// wave's) instead of six ds_bpermute butterflies; order does not matter for min / max, the value is the same
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_src_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false); }   // lanes without a source keep v
__device__ inline int wave_min_i32(int v) {
  v = min(v, dpp_src_i32<0x111, 0xF>(v)); v = min(v, dpp_src_i32<0x112, 0xF>(v)); v = min(v, dpp_src_i32<0x114, 0xF>(v));
  v = min(v, dpp_src_i32<0x118, 0xF>(v)); v = min(v, dpp_src_i32<0x142, 0xA>(v)); v = min(v, dpp_src_i32<0x143, 0xC>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ inline int wave_max_i32(int v) {
  v = max(v, dpp_src_i32<0x111, 0xF>(v)); v = max(v, dpp_src_i32<0x112, 0xF>(v)); v = max(v, dpp_src_i32<0x114, 0xF>(v));
  v = max(v, dpp_src_i32<0x118, 0xF>(v)); v = max(v, dpp_src_i32<0x142, 0xA>(v)); v = max(v, dpp_src_i32<0x143, 0xC>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ inline uint32_t wave_min_u32(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
  return uni(v);
}
__device__ inline uint32_t wave_max_u32(uint32_t v) {
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
  return uni(v);
}
__device__ inline unsigned long long wave_sum_u64(unsigned long long v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return (unsigned long long)uni((int64_t)v);
}

// Exactness certificate for a sum of f32 values widened to f64.
// Every nonzero finite f32 x with unbiased exponent e is an integer multiple of 2^(max(e,-126)-23).  If all
// addends are multiples of u = 2^ue_min and n * 2^(e_max+1) <= 2^53 * u, then every partial sum, in any
// order, is a multiple of u of magnitude < 2^53 * u — exactly representable — so no addition ever rounds
// and the result equals the reference's left-to-right sum bit for bit.
struct SumCert {
  int emin;  // min over nonzero addends of max(e, -126)
  int emax;  // max over nonzero addends of max(e, -126)
  bool bad;  // NaN / Inf seen
  __device__ SumCert() : emin(1 << 20), emax(-(1 << 20)), bad(false) {}
  __device__ inline void add(float x) {              // (branch-free: one candidate per lane, a branch here is an exec-mask round trip)
    const uint32_t b = __float_as_uint(x);
    const int ex = (int)((b >> 23) & 0xFFu);
    const bool inf = ex == 255, skip = inf || (b & 0x7FFFFFFFu) == 0u;
    bad |= inf;
    const int e = ex ? ex - 127 : -126;
    emin = skip ? emin : min(emin, e);
    emax = skip ? emax : max(emax, e);
  }
};

__device__ inline int ceil_log2_i64(int64_t n) {     // smallest b with 2^b >= n (closed form: the mask step of the table kernels asks per step)
  return n <= 1 ? 0 : 64 - __builtin_clzll((unsigned long long)(n - 1));
}

// true when the sum of `n` addends described by the (wave-reduced) certificate is order-independent
__device__ inline bool sum_is_exact(int emin, int emax, bool bad, int64_t n) {
  if (bad) return false;
  if (emax < emin) return true;  // all zeros
  return ceil_log2_i64(n) + emax - emin <= 29;  // log2(n) + e_max + 1 <= 53 + (e_min - 23)
}

}  // namespace srw
