// lane_sampling.h — the certified samplers of sampling.h for ONE LANE: a walker's step taken sequentially by its lane (walk_lanes.hip,
// walk_rounds.hip).  Same arithmetic as wave_pick_first / wave_pick_masked / binned_resolve<ABS> under the row certificate ROW_PQ_OK (every
// sum of variants exact in any order; the divide-free certain-miss / certain-hit compares), therefore the same picks bit for bit.
#pragma once
#include "group_sampling.h"

namespace srw {
namespace lane {

#ifndef SRW_LANE_NB
#define SRW_LANE_NB 2                 // 16-byte loads in flight per array and trip of a lane's candidate loops (x 4 candidates)
#endif
constexpr int LNB = SRW_LANE_NB;
constexpr int32_t LANE_ROW_MAX = 255;                 // whole rows a lane walks itself (two passes): every row with a membership mask

// A lane's loops are chains of dependent round trips unless the loads of a trip are issued together: candidates are taken NB x 4 at a
// time (NB 16-byte loads per array in flight), whole small levels at once.
// A run of candidates [k0, k1] of a row appended, in order, to the exact numerator `acc`: the first candidate that is not a certain
// miss (sampling.h:binned_resolve's compares).  members: bit ((k - kw) & 63) of the word of position k says "candidate k is in N(prev)";
// mw: the words of positions kw, kw + 64, ... (kw a multiple of 64 at or before k0; the run covers at most four words); hub: prev's
// neighbor-set bitmap to probe instead (or null).  Returns its position and sets id_out / hit, or -1 (acc then holds the whole run).
template <int NB>
__device__ inline int32_t lane_scan(const GraphView &g, int64_t roff, int32_t k0, int32_t k1, int32_t kw, const unsigned long long (&mw)[4],
                                    bool members, const uint32_t *hub, bool returns, bool biased, int32_t prev, const BiasDiv &bdiv,
                                    double &acc, double pS, int32_t &id_out, bool &hit) {
  // num < lo  =>  fl(num (1 + t_k)) < pS for every k <= k1: a certain miss without the two products
  const double lo = pS * (1.0 - ((double)(k1 + 8) * 0x1p-50 + 0x1p-50));
  int32_t res = -1;
  for (int32_t kb = k0; kb <= k1 && res < 0; kb += 4 * NB) {
    g16::Cand4 c[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) g16::load_cand4(g, roff, kb + 4 * b, k1, c[b]);
    uint32_t hb[NB][4];
    if (hub) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t) { hb[b][t] = 0u; if (c[b].valid[t] && c[b].id[t] != prev) hb[b][t] = hub[(uint32_t)((int64_t)c[b].id[t] - g.vmin) >> 5]; }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int32_t kq = kb + 4 * b;
      if (res < 0 && kq <= k1) {
        const int wsel = (kq - kw) >> 6;
        const unsigned long long word = wsel == 0 ? mw[0] : wsel == 1 ? mw[1] : wsel == 2 ? mw[2] : mw[3];
        float w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool is_prev = returns && c[b].id[t] == prev;
          bool in = false;
          if (hub) in = !is_prev && ((hb[b][t] >> ((uint32_t)((int64_t)c[b].id[t] - g.vmin) & 31u)) & 1u);
          else if (members) in = !is_prev && ((word >> ((kq + t - kw) & 63)) & 1ull);
          w[t] = !c[b].valid[t] ? 0.0f : biased ? bdiv(c[b].w[t], is_prev, in) : c[b].w[t];
        }
        // (the four together first: a block that stays below `lo` is four certain misses; one exit, every index a constant — a `return`
        //  per candidate is merged by the compiler into one with a variable index, and the candidates go through scratch)
        const double a0 = acc + (double)w[0], a1 = a0 + (double)w[1], a2 = a1 + (double)w[2], a3 = a2 + (double)w[3];
        if (!(a3 < lo)) {
          if (c[b].valid[3] && g16::not_miss(kq + 3, a3, pS)) { res = kq + 3; id_out = c[b].id[3]; hit = g16::sure_hit(kq + 3, a3, pS); }
          if (c[b].valid[2] && g16::not_miss(kq + 2, a2, pS)) { res = kq + 2; id_out = c[b].id[2]; hit = g16::sure_hit(kq + 2, a2, pS); }
          if (c[b].valid[1] && g16::not_miss(kq + 1, a1, pS)) { res = kq + 1; id_out = c[b].id[1]; hit = g16::sure_hit(kq + 1, a1, pS); }
          if (c[b].valid[0] && g16::not_miss(kq + 0, a0, pS)) { res = kq + 0; id_out = c[b].id[0]; hit = g16::sure_hit(kq + 0, a0, pS); }
        }
        acc = a3;
      }
    }
  }
  return res;
}

// A whole row of at most 255 candidates that holds the certificate (ROW_PQ_OK: every sum of variants is exact in any order): the first
// step of a walk (raw weights, RandomWalk.scala:51-66) or a second-order step whose pair has a membership mask — wave_pick_first /
// wave_pick_masked, sequentially: S in one pass, the scan in a second.  k >= 0, or CHAIN_NEEDED.
__device__ inline int32_t lane_pick_row(const GraphView &g, const Row &r, bool second, int32_t prev, const BiasDiv &bdiv, uint32_t eo, float u, int32_t &id_out) {
  unsigned long long mw[4] = {eo, 0ull, 0ull, 0ull};
  if (second && r.deg > 32) {
    const uint2 *wp = reinterpret_cast<const uint2 *>(g.em_bits + (size_t)eo * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) if (i * 64 < r.deg) { const uint2 q = wp[i]; mw[i] = ((unsigned long long)q.y << 32) | q.x; }
  }
  double S = 0.0;
  if (!second && g.ids32) S = (double)r.deg;         // unit weights: nothing to read
  else {
    for (int32_t kb = 0; kb < r.deg; kb += 4 * LNB) {
      g16::Cand4 c[LNB];
#pragma unroll
      for (int b = 0; b < LNB; ++b) g16::load_cand4(g, r.off, kb + 4 * b, r.deg - 1, c[b]);
#pragma unroll
      for (int b = 0; b < LNB; ++b) {
        const int32_t kq = kb + 4 * b;
        const int wsel = kq >> 6;
        const unsigned long long word = wsel == 0 ? mw[0] : wsel == 1 ? mw[1] : wsel == 2 ? mw[2] : mw[3];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool is_prev = second && c[b].id[t] == prev;
          const float x = second ? bdiv(c[b].w[t], is_prev, !is_prev && ((word >> ((kq + t) & 63)) & 1ull)) : c[b].w[t];
          S += c[b].valid[t] ? (double)x : 0.0;
        }
      }
    }
  }
  if (!(S > 0.0)) return CHAIN_NEEDED;                // (S = 0: the reference divides by zero -> chain)
  const double pS = (double)u * S;
  double acc = 0.0;
  bool hit = false;
  const int32_t k = lane_scan<LNB>(g, r.off, 0, r.deg - 1, 0, mw, second, nullptr, second, second, prev, bdiv, acc, pS, id_out, hit);
  if (k < 0) { id_out = load_ent(g, g.ent + r.off, 0).id; return 0; }      // edges.head (:24)
  return hit ? k : CHAIN_NEEDED;
}

// A second-order step through the pair's table: binned_resolve<ABS>, sequentially — per level the block's absolute prefixes (a short
// block: all of it in one round trip; else a binary search: the compare is monotone, prefixes and tolerances both grow) or a run
// over the 16-bit deltas, 32 at a time, then the located chunk.
// k >= 0; -1 (non-positive sum); CHAIN_NEEDED (S_out = the row's sum); LANE_SERVE: the wave takes the step.
constexpr int32_t LANE_SERVE = -4;
__device__ inline int32_t lane_pick_table(const GraphView &g, const Row &rc, int32_t prev, const Row &rprev, float p_, float q_, uint32_t eo,
                                          float u, int max_csh, int32_t &id_out, double &S_out, uint32_t &res_bytes) {
  const int32_t deg = rc.deg, m = rprev.deg;
  const uint32_t rflags = rc.flags;
  const PairGeom pg = eb_pair_geometry(deg, m, g.ebp);
  const int csh = pg.csh;
  if (csh > max_csh) return LANE_SERVE;              // longer located chunks: the wave's rounds
  const uint32_t prev_hub = rprev.flags >> ROW_HUB_SHIFT;
  const uint32_t *hubbits = (prev_hub && g.hub_bm) ? g.hub_bm + (int64_t)(prev_hub - 1) * g.hub_words : nullptr;
  const bool one_sign = (q_ > 1.0f && p_ <= q_) || (q_ < 1.0f && p_ >= q_);
  if (!pg.cmask && !hubbits && !one_sign) return LANE_SERVE;
  const int32_t n_bins = pg.n_bins;
  const bool f32t = g.ebp.f32 && (rflags & ROW_PQ_F32);
  const bool u16t = eb_pair_u16(rflags, csh, g.ebp);
  const double unit = u16t ? eb_row_unit(rflags) : 0.0;
  const EbLayout lay = eb_layout(f32t, n_bins, pg.cmask, deg, u16t);
  const char *table = reinterpret_cast<const char *>(g.eb_bins) + (size_t)eo * 64;
  auto chunk_end = [&](int32_t j) { const int64_t e = (((int64_t)j + 1) << csh) - 1; return (int32_t)(e < deg ? e : deg - 1); };
  const int nlev = lay.n2 ? 3 : lay.n1 ? 2 : 1;
  double S = 0.0, pS = 0.0, prev_val = 0.0, b_prev = 0.0, b_this = 0.0;
  int32_t blk = 0, jc = 0;
  for (int it = 0; it < nlev; ++it) {
    const int L = nlev - 1 - it;
    const uint32_t off = L == 2 ? lay.l2_off : L == 1 ? lay.l1_off : lay.l0_off;
    const int32_t cnt = L == 2 ? lay.n2 : L == 1 ? lay.n1 : n_bins;
    const char *lvl = table + (size_t)off * 64;
    const int32_t i_first = blk * 64;
    const int32_t n_here = cnt - i_first < 64 ? cnt - i_first : 64;
    if (L == 0 && u16t) {
      const unsigned short *d = reinterpret_cast<const unsigned short *>(lvl) + i_first;
      if (it == 0) {                                 // the only level: S = the sum of all the deltas (cnt <= 64; the level is padded to 64 bytes)
        uint32_t tot = 0u;
        for (int32_t i = 0; i < n_here; i += 32) {
          U32x4 q[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) q[b] = *reinterpret_cast<const U32x4 *>(d + i + 8 * b);
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const uint32_t x[4] = {q[b].a, q[b].b, q[b].c, q[b].d};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const int32_t e = i + 8 * b + 2 * t;
              tot += (e < n_here ? x[t] & 0xFFFFu : 0u) + (e + 1 < n_here ? x[t] >> 16 : 0u);
            }
          }
        }
        S = (double)tot * unit;
        if (!(S > 0.0)) { S_out = S; return -1; }
        if (g.dbg_chain_deg && deg >= g.dbg_chain_deg) { S_out = S; return CHAIN_NEEDED; }
        pS = (double)u * S;
      }
      // (every partial sum is an exact multiple of 2^G under the row certificate: integer units, one conversion per test; a pair of
      //  deltas that stays below the threshold in units is two certain misses)
      const double lo = pS * (1.0 - ((double)(deg + 8) * 0x1p-50 + 0x1p-50));
      const double thr_d = (lo - prev_val) / unit - 2.0;
      const uint32_t thr = thr_d > 0.0 ? (thr_d < 4294967040.0 ? (uint32_t)thr_d : 0xFFFFFF00u) : 0u;
      uint32_t au = 0u;
      int32_t found = -1;
      for (int32_t i = 0; i < n_here && found < 0; i += 32) {
        U32x4 q[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) q[b] = *reinterpret_cast<const U32x4 *>(d + i + 8 * b);      // (64 bytes of a level padded to 64 bytes)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint32_t x[4] = {q[b].a, q[b].b, q[b].c, q[b].d};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int32_t e = i + 8 * b + 2 * t;
            if (found >= 0 || e >= n_here) break;
            const uint32_t d0 = x[t] & 0xFFFFu, d1 = e + 1 < n_here ? x[t] >> 16 : 0u;
            if (au + d0 + d1 < thr) { au += d0 + d1; continue; }
            const uint32_t a0 = au + d0, a1 = a0 + d1;
            const double v0 = prev_val + (double)a0 * unit, v1 = prev_val + (double)a1 * unit, vb = prev_val + (double)au * unit;
            if (g16::not_miss(chunk_end(i_first + e), v0, pS)) { found = e; b_this = v0; b_prev = vb; }
            else if (e + 1 < n_here && g16::not_miss(chunk_end(i_first + e + 1), v1, pS)) { found = e + 1; b_this = v1; b_prev = v0; }
            au = a1;
          }
        }
      }
      if (found < 0) { id_out = load_ent(g, g.ent + rc.off, 0).id; S_out = S; return 0; }    // even the last candidate is a certain miss -> edges.head
      jc = i_first + found;
    } else {
      auto value = [&](int32_t i) { return f32t ? (double)reinterpret_cast<const float *>(lvl)[i_first + i] : reinterpret_cast<const double *>(lvl)[i_first + i]; };
      auto elem_miss = [&](int32_t i, double v) {
        const int64_t je = (((int64_t)(i_first + i) + 1) << (6 * L)) - 1;      // the chunk this element is the prefix of
        const int32_t j = (int32_t)(je < n_bins ? je : n_bins - 1);
        return !g16::not_miss(chunk_end(j), v, pS);
      };
      int32_t lo_i = 0, hi_i = n_here;               // first element that is not a certain miss
      double v_lo = prev_val, v_hi = 0.0;            // value just before lo_i, value at hi_i
      if (n_here <= 8) {                             // a short block (the upper levels of tables of up to 512 chunks): all of it at once
        double v[8];
        if (f32t) {
          const float4 q0 = *reinterpret_cast<const float4 *>(lvl + (size_t)i_first * 4), q1 = *reinterpret_cast<const float4 *>(lvl + (size_t)i_first * 4 + 16);
          v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;      // (a level is padded to 64 bytes)
        } else {
          const double *dp = reinterpret_cast<const double *>(lvl) + i_first;
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = dp[i];
        }
        if (it == 0) {
          const int e = cnt - 1;
          S = e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : e == 3 ? v[3] : e == 4 ? v[4] : e == 5 ? v[5] : e == 6 ? v[6] : v[7];
          if (!(S > 0.0)) { S_out = S; return -1; }
          if (g.dbg_chain_deg && deg >= g.dbg_chain_deg) { S_out = S; return CHAIN_NEEDED; }
          pS = (double)u * S;
        }
        lo_i = n_here;
#pragma unroll
        for (int i = 7; i >= 0; --i)
          if (i < n_here && !elem_miss(i, v[i])) { lo_i = i; v_hi = v[i]; v_lo = i ? v[i - 1] : prev_val; }
      } else {
        if (it == 0) {                               // (the top level has at most 64 elements)
          S = value(cnt - 1);
          if (!(S > 0.0)) { S_out = S; return -1; }
          if (g.dbg_chain_deg && deg >= g.dbg_chain_deg) { S_out = S; return CHAIN_NEEDED; }
          pS = (double)u * S;
        }
        while (lo_i < hi_i) {
          const int32_t mid = (lo_i + hi_i) >> 1;
          const double v = value(mid);
          if (!elem_miss(mid, v)) { hi_i = mid; v_hi = v; } else { lo_i = mid + 1; v_lo = v; }
        }
      }
      if (lo_i >= n_here) { id_out = load_ent(g, g.ent + rc.off, 0).id; S_out = S; return 0; }
      if (L == 0) { jc = i_first + lo_i; b_this = v_hi; b_prev = v_lo; }
      else { blk = i_first + lo_i; prev_val = v_lo; }
    }
  }
  S_out = S;
  // ---- the located chunk (at most 256 candidates: four mask words)
  const int32_t k0 = (int32_t)((int64_t)jc << csh), k1 = chunk_end(jc);
  const BiasDiv bdiv(p_, q_);
  unsigned long long mw[4] = {0ull, 0ull, 0ull, 0ull};
  bool members = false, returns = true;
  const uint32_t *hub = nullptr;
  if (pg.cmask) {
    const unsigned long long *cm = reinterpret_cast<const unsigned long long *>(table + (size_t)lay.cm_off * 64) + (k0 >> 6);
#pragma unroll
    for (int i = 0; i < 4; ++i) if (k0 + 64 * i <= k1) mw[i] = cm[i];
    members = true;
  } else {
    bool no_specials = false;
    if (one_sign) {                                  // "the chunk's corrections sum to exactly 0" means "no special in the chunk"
      const PqRow PQ(g, rc.off);
      no_specials = (b_this - b_prev) - (PQ[k1] - (k0 ? PQ[k0 - 1] : 0.0)) == 0.0;
    }
    if (no_specials) returns = false;
    else if (hubbits) hub = hubbits;
    else return LANE_SERVE;                          // a short N(prev) (LDS staging), the row filters, the edge hash: the wave
  }
  double acc = b_prev;                               // A'_{k0-1}
  bool hit = false;
  const int32_t k = hub ? lane_scan<2>(g, rc.off, k0, k1, k0, mw, members, hub, returns, true, prev, bdiv, acc, pS, id_out, hit)
                        : lane_scan<LNB>(g, rc.off, k0, k1, k0, mw, members, hub, returns, true, prev, bdiv, acc, pS, id_out, hit);
  {                                                  // (counted as the wave kernel counts it: the rounds of 64 up to the one that decided)
    const int32_t rounds = (((k >= 0 ? k : k1) - k0) >> 6) + 1, all = k1 - k0 + 1;
    res_bytes += 8u * (uint32_t)(all < 64 * rounds ? all : 64 * rounds);
  }
  if (k < 0 || !hit) return CHAIN_NEEDED;
  return k;
}

}  // namespace lane
}  // namespace srw
