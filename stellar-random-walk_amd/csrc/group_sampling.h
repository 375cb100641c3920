// group_sampling.h — the certified samplers of sampling.h for SUB-WAVE GROUPS: one walker per 16 lanes (one DPP row), four
// walkers per wave.  Same arithmetic as wave_pick_first / wave_pick_masked / binned_resolve<ABS> (RandomSample.scala:12-44 under the
// row certificates: exact sums in any order, divide-free certain-miss / certain-hit compares), therefore the same picks bit for bit.
//
// Why: with one walker per WAVE the table step is bound by instruction issue (profiles/r04_valu_issue.md: ~350 vector + ~420 scalar
// instructions per walk step, both units > 80 % busy) — a 64-lane instruction per value that ONE walker needs.  A block of 64 chunk
// prefixes or 64 candidates is four per lane for a group of 16; the scans are the four row_shr steps a DPP row has natively, the
// walker's state lives in (replicated) vector registers, and nothing of a step is scalar work any more.
//
// Conventions: every function is called by whole groups (all 16 lanes of a group take the same branches; different groups of a wave
// may diverge); values named "group-uniform" are equal in the 16 lanes of a group.  A block of 64 items is laid out four CONSECUTIVE
// items per lane (lane gl holds items 4 gl .. 4 gl + 3): 16-byte loads, one row scan over the lanes' totals.
#pragma once
#include "sampling.h"

namespace srw {
namespace g16 {

__device__ inline int gl_id() { return (int)(threadIdx.x & 15u); }        // lane inside the group
__device__ inline int g_first_lane() { return (int)(threadIdx.x & 48u); } // the group's first lane inside the wave

// value of lane j (0 .. 15, group-uniform) of my group
__device__ inline int32_t grp_get(int32_t v, int j) { return __builtin_amdgcn_ds_bpermute((g_first_lane() | j) << 2, v); }
__device__ inline uint32_t grp_get(uint32_t v, int j) { return (uint32_t)grp_get((int32_t)v, j); }
__device__ inline float grp_get(float v, int j) { return __uint_as_float(grp_get(__float_as_uint(v), j)); }
__device__ inline double grp_get(double v, int j) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  const uint32_t lo = grp_get((uint32_t)b, j), hi = grp_get((uint32_t)(b >> 32), j);
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// first lane (0 .. 15) of my group whose predicate holds, -1 if none.  (Lanes of groups that are not executing contribute nothing.)
__device__ inline int grp_first(bool pred) {
  const unsigned long long m = __ballot(pred);
  const uint32_t mine = (uint32_t)(m >> (threadIdx.x & 48u)) & 0xFFFFu;
  return mine ? (int)__builtin_ctz(mine) : -1;
}
__device__ inline bool grp_any(bool pred) {
  const unsigned long long m = __ballot(pred);
  return ((uint32_t)(m >> (threadIdx.x & 48u)) & 0xFFFFu) != 0u;
}

// inclusive scan over the 16 lanes of a group: the four row_shr steps of sampling.h:wave_incl_scan_f64 (lanes without a source add 0.0)
__device__ inline double grp_incl_scan_f64(double v) {
  v = dpp_add_f64<0x111, 0xF>(v);
  v = dpp_add_f64<0x112, 0xF>(v);
  v = dpp_add_f64<0x114, 0xF>(v);
  v = dpp_add_f64<0x118, 0xF>(v);
  return v;
}
// lane gl gets lane gl - 1's value, lane 0 of the group gets `fill`
__device__ inline double grp_shr1_f64(double v, double fill) {
  const uint64_t b = (uint64_t)__double_as_longlong(v), f = (uint64_t)__double_as_longlong(fill);
  const int lo = __builtin_amdgcn_update_dpp((int)(uint32_t)f, (int)(uint32_t)b, 0x111, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(uint32_t)(f >> 32), (int)(uint32_t)(b >> 32), 0x111, 0xF, 0xF, false);
  return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}
// the group's total in every lane (row rotations by 8, 4, 2, 1; the order differs per lane: for sums that are exact in any order)
template <int CTRL>
__device__ inline double dpp_ror_add_f64(double v) {
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, false);
  return v + __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}
__device__ inline double grp_total_f64(double v) {
  v = dpp_ror_add_f64<0x128>(v); v = dpp_ror_add_f64<0x124>(v); v = dpp_ror_add_f64<0x122>(v); v = dpp_ror_add_f64<0x121>(v);
  return v;
}
template <int CTRL>
__device__ inline int dpp_ror_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
__device__ inline int grp_min_i32(int v) {
  v = min(v, dpp_ror_i32<0x128>(v)); v = min(v, dpp_ror_i32<0x124>(v)); v = min(v, dpp_ror_i32<0x122>(v)); v = min(v, dpp_ror_i32<0x121>(v));
  return v;
}
__device__ inline int grp_max_i32(int v) {
  v = max(v, dpp_ror_i32<0x128>(v)); v = max(v, dpp_ror_i32<0x124>(v)); v = max(v, dpp_ror_i32<0x122>(v)); v = max(v, dpp_ror_i32<0x121>(v));
  return v;
}

// Four consecutive candidates k .. k + 3 of a row (input order): ids + weights (1.0f on unit-weight graphs, GraphView::ids32).
// Candidates beyond `last` (the last valid position) come back as (id 0, w 0) with valid = false.
struct Cand4 { int32_t id[4]; float w[4]; bool valid[4]; };
__device__ inline void load_cand4(const GraphView &g, int64_t roff, int32_t k, int32_t last, Cand4 &c) {
#pragma unroll
  for (int t = 0; t < 4; ++t) { c.id[t] = 0; c.w[t] = 0.0f; c.valid[t] = k + t <= last; }
  if (g.ids32) {
    const int32_t *p = g.ids32 + roff + k;
    if (k + 3 <= last) {
      const U32x4 v = *reinterpret_cast<const U32x4 *>(p);
      c.id[0] = (int32_t)v.a; c.id[1] = (int32_t)v.b; c.id[2] = (int32_t)v.c; c.id[3] = (int32_t)v.d;
      c.w[0] = c.w[1] = c.w[2] = c.w[3] = 1.0f;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) if (c.valid[t]) { c.id[t] = p[t]; c.w[t] = 1.0f; }
    }
  } else {
    const Ent *p = g.ent + roff + k;
    if (k + 3 <= last) {
      const U32x4 a = *reinterpret_cast<const U32x4 *>(p), b = *reinterpret_cast<const U32x4 *>(p + 2);
      c.id[0] = (int32_t)a.a; c.w[0] = __uint_as_float(a.b); c.id[1] = (int32_t)a.c; c.w[1] = __uint_as_float(a.d);
      c.id[2] = (int32_t)b.a; c.w[2] = __uint_as_float(b.b); c.id[3] = (int32_t)b.c; c.w[3] = __uint_as_float(b.d);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) if (c.valid[t]) { const Ent e = p[t]; c.id[t] = e.id; c.w[t] = e.w; }
    }
  }
}

// The certified compares of binned_resolve (sampling.h): num = exact numerator A'_k, pS = fl(p S).
__device__ inline bool not_miss(int32_t k, double num, double pS) { return !(num * (1.0 + (double)(k + 8) * 0x1p-51) < pS); }
__device__ inline bool sure_hit(int32_t k, double num, double pS) { return num * (1.0 - (double)(k + 8) * 0x1p-51) >= pS; }

// One block of 64 candidate weights (four per lane, w[t] = 0 for the invalid ones) appended to the exact numerator `carry`:
// the first candidate that is not a certain miss.  Returns the group's lane that holds it (-1: none; carry then includes the block);
// found_t = its slot in that lane, found_hit = it is a certain hit.  All group-uniform.
__device__ inline int scan_block(const float (&w)[4], const bool (&valid)[4], int32_t k_first /* position of the lane's first candidate */,
                                 double &carry, double pS, int &found_t, bool &found_hit) {
  const double c0 = (double)w[0], c1 = c0 + (double)w[1], c2 = c1 + (double)w[2], c3 = c2 + (double)w[3];
  const double incl = grp_incl_scan_f64(c3);
  const double base = carry + grp_shr1_f64(incl, 0.0);            // pure additions of exact values: no cancellation
  int ft = 4; bool fh = false;
  {
    const double n3 = base + c3, n2 = base + c2, n1 = base + c1, n0 = base + c0;
    if (valid[3] && not_miss(k_first + 3, n3, pS)) { ft = 3; fh = sure_hit(k_first + 3, n3, pS); }
    if (valid[2] && not_miss(k_first + 2, n2, pS)) { ft = 2; fh = sure_hit(k_first + 2, n2, pS); }
    if (valid[1] && not_miss(k_first + 1, n1, pS)) { ft = 1; fh = sure_hit(k_first + 1, n1, pS); }
    if (valid[0] && not_miss(k_first + 0, n0, pS)) { ft = 0; fh = sure_hit(k_first + 0, n0, pS); }
  }
  const int f = grp_first(ft < 4);
  if (f >= 0) {
    const int32_t packed = grp_get((int32_t)(ft | (fh ? 8 : 0)), f);
    found_t = packed & 3; found_hit = (packed & 8) != 0;
  } else {
    carry += grp_get(incl, 15);
  }
  return f;
}

// ---- rows evaluated whole: the first step of a walk (RandomWalk.scala:51-66: RandomSample.sample on the raw row) and the second-order
// steps into rows of fewer than 256 candidates with a precomputed membership mask (edge_tables.hip) — wave_pick_first / wave_pick_masked.
// Pass 1: S = the exact parallel sum under the certificate; pass 2: the blocks of 64 from the start of the row (block 0 stays in
// registers between the passes).  Returns the pick's position (id_out), or CHAIN_NEEDED (no certificate / a draw within rounding
// distance of a CDF boundary: the caller hands the walker over).
__device__ inline int32_t grp_pick_row(const GraphView &g, const Row &rc, bool second, int32_t prev, const BiasDiv &bdiv, uint32_t eo,
                                       float r, int32_t &id_out) {
  const int gl = gl_id();
  const int32_t deg = rc.deg;
  const int64_t roff = rc.off;
  const int ni = (deg + 63) >> 6;
  const uint32_t *words = (second && deg > 32) ? g.em_bits + (size_t)eo * 4 : nullptr;
  // weights of block i (four per lane), biased on a second-order step: the divisor by class (return edge: p, member of N(prev): 1, else q)
  auto block = [&](int i, float (&w)[4], int32_t (&id)[4], bool (&valid)[4]) {
    Cand4 c;
    load_cand4(g, roff, i * 64 + 4 * gl, deg - 1, c);
    uint32_t mw = 0u;
    if (second) mw = words ? words[2 * i + (gl >> 3)] : ((i == 0 && gl < 8) ? eo : 0u);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      valid[t] = c.valid[t]; id[t] = c.id[t];
      float x = c.w[t];
      if (second) x = bdiv(x, c.id[t] == prev, ((mw >> ((4 * gl + t) & 31)) & 1u) != 0u);
      w[t] = c.valid[t] ? x : 0.0f;
    }
  };
  float w0[4]; int32_t id0[4]; bool v0[4];
  double S;
  bool have0 = false;
  if (!second && g.ids32) {
    S = (double)deg;                                   // unit weights: nothing to read, nothing to certify (deg < 2^29)
  } else {
    double part = 0.0;
    SumCert cert;
    bool neg = false;
    for (int i = 0; i < ni; ++i) {
      float w[4]; int32_t id[4]; bool valid[4];
      block(i, w, id, valid);
#pragma unroll
      for (int t = 0; t < 4; ++t) { part += (double)w[t]; cert.add(w[t]); neg |= !(w[t] >= 0.0f); }   // (an invalid slot adds +0.0: skipped by the certificate)
      if (i == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { w0[t] = w[t]; id0[t] = id[t]; v0[t] = valid[t]; }
      }
    }
    have0 = true;
    const int emin = grp_min_i32(cert.emin), emax = grp_max_i32(cert.emax);
    const bool bad = grp_any(cert.bad || neg);
    S = grp_total_f64(part);                           // (used only under the certificate below: exact in any order)
    if (bad || !sum_is_exact(emin, emax, false, deg) || !(S > 0.0)) return CHAIN_NEEDED;     // (S = 0: the reference divides by zero -> chain)
  }
  const double pS = (double)r * S;
  double carry = 0.0;
  int32_t res = -3;                                    // -3: still looking
  for (int i = 0; i < ni && res == -3; ++i) {
    float w[4]; int32_t id[4]; bool valid[4];
    if (i == 0 && have0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { w[t] = w0[t]; id[t] = id0[t]; valid[t] = v0[t]; }
    } else {
      block(i, w, id, valid);
      if (i == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t) id0[t] = id[t];
      }
    }
    int ft = 0; bool fh = false;
    const int f = scan_block(w, valid, i * 64 + 4 * gl, carry, pS, ft, fh);
    if (f >= 0) {
      const int32_t idm = ft == 0 ? id[0] : ft == 1 ? id[1] : ft == 2 ? id[2] : id[3];
      id_out = grp_get(idm, f);
      res = fh ? i * 64 + 4 * f + ft : CHAIN_NEEDED;   // within rounding distance of a boundary: exact chain
    }
  }
  if (res == -3) { id_out = grp_get(id0[0], 0); res = 0; }      // edges.head (:24)
  return res;
}

// ---- second-order step through the pair's table (edge_tables.hip): tree search over the chunk prefixes, then the located chunk candidate
// by candidate — binned_resolve<ABS = true, BF, CHAIN = false> for a group.  `stage`: 1 024 words of LDS shared by the wave's four groups
// (a short N(prev) is staged and searched there, one group at a time).  Returns the pick's position (id_out), -1 (non-positive sum: the
// general kernel takes the walker) or CHAIN_NEEDED (S_out = the row's sum).
template <bool BF>
__device__ inline int32_t grp_pick_table(const GraphView &g, const Row &rc, int32_t prev, const Row &rprev, float p_, float q_, uint32_t eo,
                                         float r, uint32_t *stage, int32_t &id_out, double &S_out, unsigned long long &res_bytes) {
  const int gl = gl_id();
  const int32_t deg = rc.deg;
  const uint32_t rflags = rc.flags;
  const int64_t roff = rc.off;
  const int32_t m = rprev.deg;
  const uint32_t *B = g.sids + rprev.off;            // sorted (id - vmin) of N(prev)
  const uint32_t xprev = (uint32_t)((int64_t)prev - g.vmin);
  const uint32_t prev_hub = rprev.flags >> ROW_HUB_SHIFT;
  const uint32_t *hubbits = (prev_hub && g.hub_bm) ? g.hub_bm + (int64_t)(prev_hub - 1) * g.hub_words : nullptr;
  const PairGeom pg = eb_pair_geometry(deg, m, g.ebp);
  const int csh = pg.csh;
  const int32_t n_bins = pg.n_bins;
  const bool f32t = g.ebp.f32 && (rflags & ROW_PQ_F32);
  const bool u16t = eb_pair_u16(rflags, csh, g.ebp);
  const double unit = u16t ? eb_row_unit(rflags) : 0.0;
  const EbLayout lay = eb_layout(f32t, n_bins, pg.cmask, deg, u16t);
  const char *table = reinterpret_cast<const char *>(g.eb_bins) + (size_t)eo * 64;
  const unsigned long long *cmask = pg.cmask ? reinterpret_cast<const unsigned long long *>(table + (size_t)lay.cm_off * 64) : nullptr;
  auto chunk_end = [&](int32_t j) { const int64_t e = (((int64_t)j + 1) << csh) - 1; return (int32_t)(e < deg ? e : deg - 1); };
  const double p = (double)r;
  double S = 0.0, pS = 0.0;
  int32_t res = -3;                                  // -3: still looking
  // ---- the tree: one block of <= 64 values per level, top level first; S is the last element of the top level
  const int nlev = lay.n2 ? 3 : lay.n1 ? 2 : 1;
  int32_t blk = 0, jc = 0;
  double prev_val = 0.0, b_prev = 0.0, b_this = 0.0;
  for (int it = 0; it < nlev && res == -3; ++it) {
    const int L = nlev - 1 - it;
    const uint32_t off = L == 2 ? lay.l2_off : L == 1 ? lay.l1_off : lay.l0_off;
    const int32_t cnt = L == 2 ? lay.n2 : L == 1 ? lay.n1 : n_bins;
    const int32_t i0 = blk * 64 + 4 * gl;
    const char *lvl = table + (size_t)off * 64;
    bool in_r[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) in_r[t] = i0 + t < cnt;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (L == 0 && u16t) {                              // the chunks' own masses, 16 bits each: prefix = what precedes the block + a scan
      uint32_t lo = 0u, hi = 0u;
      if (in_r[0]) { const uint2 q = *reinterpret_cast<const uint2 *>(lvl + (size_t)i0 * 2); lo = q.x; hi = q.y; }
      const double d0 = in_r[0] ? (double)(lo & 0xFFFFu) * unit : 0.0, d1 = in_r[1] ? (double)(lo >> 16) * unit : 0.0;
      const double d2 = in_r[2] ? (double)(hi & 0xFFFFu) * unit : 0.0, d3 = in_r[3] ? (double)(hi >> 16) * unit : 0.0;
      const double c0 = d0, c1 = c0 + d1, c2 = c1 + d2, c3 = c2 + d3;
      const double base = prev_val + grp_shr1_f64(grp_incl_scan_f64(c3), 0.0);   // (exact: every partial sum is a multiple of 2^G under the row certificate)
      v[0] = base + c0; v[1] = base + c1; v[2] = base + c2; v[3] = base + c3;
    } else if (f32t) {
      if (in_r[0]) {
        const float4 q = *reinterpret_cast<const float4 *>(lvl + (size_t)i0 * 4);
        v[0] = (double)q.x; v[1] = (double)q.y; v[2] = (double)q.z; v[3] = (double)q.w;
      }
    } else {
      if (in_r[0]) {
        const double2 q0 = *reinterpret_cast<const double2 *>(lvl + (size_t)i0 * 8), q1 = *reinterpret_cast<const double2 *>(lvl + (size_t)i0 * 8 + 16);
        v[0] = q0.x; v[1] = q0.y; v[2] = q1.x; v[3] = q1.y;
      }
    }
    if (it == 0) {                                     // (the top level has at most 64 elements)
      const int e = cnt - 1, t = e & 3;
      const double sv = t == 0 ? v[0] : t == 1 ? v[1] : t == 2 ? v[2] : v[3];
      S = grp_get(sv, e >> 2);
      if (!(S > 0.0)) { res = -1; break; }
      if (g.dbg_chain_deg && deg >= g.dbg_chain_deg) { res = CHAIN_NEEDED; break; }   // tests: the chain kernels on every long row
      pS = p * S;
    }
    int ft = 4;
    double v_this = 0.0, v_before = 0.0;
    const double v3_left = grp_shr1_f64(v[3], prev_val);   // the element just before this lane's first one
#pragma unroll
    for (int t = 3; t >= 0; --t) {
      const int64_t je = (((int64_t)(i0 + t) + 1) << (6 * L)) - 1;      // the chunk this element is the prefix of
      const int32_t j = (int32_t)(je < n_bins ? je : n_bins - 1);
      if (in_r[t] && not_miss(chunk_end(j), v[t], pS)) { ft = t; v_this = v[t]; v_before = t ? v[t - 1] : v3_left; }
    }
    const int f = grp_first(ft < 4);
    if (f < 0) { id_out = load_ent(g, g.ent + roff, 0).id; res = 0; break; }      // even the last candidate is a certain miss -> edges.head
    const int fe = 4 * f + grp_get((int32_t)ft, f);
    const double vt = grp_get(v_this, f), vb = grp_get(v_before, f);
    if (L == 0) { jc = blk * 64 + fe; b_this = vt; b_prev = vb; }
    else { blk = blk * 64 + fe; prev_val = vb; }
  }
  if (res != -3) { S_out = S; return res; }
  // ---- the located chunk, 64 candidates per round
  const int32_t k0 = (int32_t)((int64_t)jc << csh), k1 = chunk_end(jc);
  // q > 1 and p <= q: every correction (w - w/q for a member, w/p - w/q for a return edge) is >= 0; q < 1 and p >= q: every one is
  // <= 0.  Then "the chunk's corrections sum to exactly 0" means "no special in the chunk" (two reads of the base prefix sums).
  const bool one_sign = (q_ > 1.0f && p_ <= q_) || (q_ < 1.0f && p_ >= q_);
  bool no_specials = false;
  if (one_sign && !cmask) {
    const PqRow PQ(g, roff);
    no_specials = (b_this - b_prev) - (PQ[k1] - (k0 ? PQ[k0 - 1] : 0.0)) == 0.0;
  }
  const BiasDiv bdiv(p_, q_);
  // membership of the candidates in N(prev), by what the pair has: its chunk masks, a short N(prev) staged in LDS, prev's hub bitmap,
  // the row filter + the exact test, the edge hash, the sorted row
  const bool probe = !no_specials && !cmask;
  const bool staged = probe && !hubbits && m > 0 && m <= 1024;
  const uint32_t *bf = nullptr; uint32_t bf_nw = 0;
  if (BF && probe && !staged && !hubbits && g.bf_off && m >= BF_MIN_DEG) {
    const uint32_t bo = g.bf_off[xprev];
    if (bo != BF_NONE) { bf = g.bf_bits + bo; bf_nw = bf_words(m); }
  }
  double carry = b_prev;                             // A'_{k0-1}
  for (int32_t base = k0; base <= k1 && res == -3; base += 64) {
    res_bytes += 8ull * (unsigned long long)((k1 - base + 1) < 64 ? (k1 - base + 1) : 64);
    Cand4 c;
    load_cand4(g, roff, base + 4 * gl, k1, c);
    unsigned long long mwc = 0ull;
    if (cmask) mwc = cmask[base >> 6];
    uint32_t xs[4]; bool want[4], in[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      xs[t] = (uint32_t)((int64_t)c.id[t] - g.vmin); in[t] = false;
      want[t] = !no_specials && c.valid[t] && c.id[t] != prev;
    }
    if (cmask) {
#pragma unroll
      for (int t = 0; t < 4; ++t) in[t] = want[t] && ((mwc >> (4 * gl + t)) & 1ull);
    }
    // a short N(prev): staged in the wave's LDS region (sorted, padded to a power of two) and searched there — one group at a time
    {
      const unsigned long long sm = __ballot(staged);
      if (sm) {
        for (int gq = 0; gq < 4; ++gq) {
          if (!((sm >> (16 * gq)) & 1ull)) continue;                            // (wave-uniform)
          if ((int)(threadIdx.x & 48u) == 16 * gq) {
            const int levels = m > 1 ? 32 - __builtin_clz((unsigned)(m - 1)) : 0;   // ceil(log2 m): the padded length is a power of two
            const int P2 = 1 << levels;
            for (int32_t t0 = 4 * gl; t0 < P2; t0 += 64) {
              if (t0 + 3 < m) {
                const U32x4 q = *reinterpret_cast<const U32x4 *>(B + t0);
                stage[t0] = q.a; stage[t0 + 1] = q.b; stage[t0 + 2] = q.c; stage[t0 + 3] = q.d;
              } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) if (t0 + t < P2) stage[t0 + t] = t0 + t < m ? B[t0 + t] : 0xFFFFFFFFu;
              }
            }
            __builtin_amdgcn_wave_barrier();
            uint32_t pos[4] = {0u, 0u, 0u, 0u};
            for (int st = levels > 0 ? (1 << (levels - 1)) : 0; st >= 1; st >>= 1) {
              uint32_t pr[4];
#pragma unroll
              for (int t = 0; t < 4; ++t) pr[t] = stage[pos[t] + st - 1];
#pragma unroll
              for (int t = 0; t < 4; ++t) if (pr[t] < xs[t]) pos[t] += st;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) in[t] = want[t] && stage[pos[t]] == xs[t];
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    if (probe && !staged) {
      if (hubbits) {
        uint32_t wd[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wd[t] = want[t] ? hubbits[xs[t] >> 5] : 0u;
#pragma unroll
        for (int t = 0; t < 4; ++t) in[t] = (wd[t] >> (xs[t] & 31)) & 1u;
      } else if (BF && bf) {                          // long N(prev), no bitmap: the row's filter first, the exact test on a positive
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          in[t] = false;
          if (want[t]) {
            uint32_t word, mask;
            bf_hash(xs[t], bf_nw, word, mask);
            if ((bf[word] & mask) == mask)
              in[t] = g.ehash ? edge_exists(g.ehash, g.ehash_mask, xprev, xs[t]) : sorted_contains(B, m, xs[t]);
          }
        }
      } else if (g.ehash) {
        edge_exists_n<4>(g.ehash, g.ehash_mask, xprev, xs, want, in);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) in[t] = want[t] && sorted_contains(B, m, xs[t]);
      }
    }
    // the candidates' variants w' (base weight fl(w / q) + the correction the tables hold, exact under the row certificate)
    float w[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = c.valid[t] ? bdiv(c.w[t], !no_specials && c.id[t] == prev, in[t]) : 0.0f;
    int ft = 0; bool fh = false;
    const int f = scan_block(w, c.valid, base + 4 * gl, carry, pS, ft, fh);
    if (f >= 0) {
      const int32_t idm = ft == 0 ? c.id[0] : ft == 1 ? c.id[1] : ft == 2 ? c.id[2] : c.id[3];
      id_out = grp_get(idm, f);
      res = fh ? base + 4 * f + ft : CHAIN_NEEDED;
    }
  }
  if (res == -3) res = CHAIN_NEEDED;                 // (no candidate of the chunk decided: the exact chain)
  S_out = S;
  return res;
}

}  // namespace g16
}  // namespace srw
