// embedding.hip — the stage behind `--cmd node2vec` / `--cmd embedding`: skip-gram with hierarchical softmax
// over the walk paths, on the GPU (gfx950).
//
// Reference: M/Main.scala:36-44 (configureWord2Vec: learning rate, iterations, partitions, minCount 0, vector size, window),
// :77-97 / :113-124 (fit on the paths as lists of id strings, then saveModelAndFeatures).  The model itself is
// org.apache.spark.mllib.feature.Word2Vec (Spark 2.2, pom.xml:125-135) — a DEPENDENCY THAT IS NOT IN /root/reference: what follows
// restates its published algorithm (skip-gram + hierarchical softmax, the word2vec.c scheme MLlib ports: vocabulary sorted by count,
// Huffman codes of at most 40 bits, a 1 000-entry sigmoid table over [-6, 6), a window shrunk by a random b per position, the linear
// decay of the learning rate down to 1e-4 of its start, syn0 uniform in (-0.5, 0.5) / dim, syn1 zero).  PARITY UNPINNED: MLlib seeds
// itself from the clock, trains its partitions Hogwild-free but in an order Spark decides and sums their rows, so no output of the
// reference can be reproduced bit for bit even by the reference; the build defines a seed (hash-based draws, below), trains one
// logical partition, and is checked against its own CPU restatement (oracle/srw_oracle.c:orc_w2v_fit) — within a float tolerance in
// the sequential mode (threads == 1), statistically in the Hogwild mode.
//
// Device side: sentences (the paths as vocabulary indices) in HBM, syn0 / syn1 as [vocab][dim] floats, one WAVE per sentence, the
// positions of a sentence in order (as MLlib walks them), lanes across the vector: a (position, context word) pair reads syn0[context]
// once, then per Huffman node of the centre word one dot product (wave reduction), one table look-up, two axpys — syn1[node] is
// updated in place (Hogwild across waves, as word2vec.c across threads), syn0[context] after the pair's nodes.  The vectors of a
// graph's embedding are small next to the walk's tables (vocab x dim x 8 bytes) and cache-resident for the upper Huffman nodes.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "engine.h"
#include "wave_primitives.h"

namespace srw {
namespace {
constexpr int TPB = 256;
constexpr int EXP_TABLE_SIZE = 1000;
constexpr float MAX_EXP = 6.0f;
constexpr int MAX_CODE_LENGTH = 40;

// the build's seeded draws (shared with oracle/srw_oracle.c:w2v_hash): a 32-bit mix of (seed, a, b, c)
__host__ __device__ inline uint32_t w2v_hash(uint32_t seed, uint32_t a, uint32_t b, uint32_t c) {
  uint32_t h = seed ^ 0x9E3779B9u;
  h ^= a + 0x7F4A7C15u + (h << 6) + (h >> 2); h *= 0x85EBCA6Bu; h ^= h >> 13;
  h ^= b + 0x165667B1u + (h << 6) + (h >> 2); h *= 0xC2B2AE35u; h ^= h >> 16;
  h ^= c + 0x27D4EB2Fu + (h << 6) + (h >> 2); h *= 0x9E3779B1u; h ^= h >> 15;
  return h;
}

struct W2vDev {
  const int64_t *sent_off; const int32_t *sent; int64_t n_sent;
  const int64_t *words_before;          // words of the sentences before this one (the learning-rate schedule)
  const int32_t *code_off; const int32_t *points; const uint8_t *codes;
  float *syn0, *syn1; int32_t dim, window; uint32_t seed; int32_t iter, n_iter; int64_t total_words; float lr;
};

__device__ inline float wave_sum_f32(float v) {
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <int ND>
__global__ __launch_bounds__(TPB) void k_w2v_train(W2vDev d, int64_t n_waves) {
  __shared__ float exp_table[EXP_TABLE_SIZE];
  for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += blockDim.x) {
    const float e = (float)exp(((double)i / EXP_TABLE_SIZE * 2.0 - 1.0) * (double)MAX_EXP);
    exp_table[i] = e / (e + 1.0f);
  }
  __syncthreads();
  const int lane = lane_id();
  const int64_t wave = blockIdx.x * (int64_t)(TPB / 64) + (threadIdx.x >> 6);
  const int D = d.dim;
  for (int64_t s = wave; s < d.n_sent; s += n_waves) {
    const int64_t o0 = d.sent_off[s], o1 = d.sent_off[s + 1];
    const int32_t len = (int32_t)(o1 - o0);
    // learning rate of this sentence: linear decay over all iterations' words, floor at 1e-4 of the start
    float alpha;
    {
      const double done = (double)d.iter * (double)d.total_words + (double)d.words_before[s];
      double a = (double)d.lr * (1.0 - done / ((double)d.n_iter * (double)d.total_words + 1.0));
      if (a < (double)d.lr * 0.0001) a = (double)d.lr * 0.0001;
      alpha = (float)a;
    }
    for (int32_t pos = 0; pos < len; ++pos) {
      const int32_t word = d.sent[o0 + pos];
      const int32_t b = (int32_t)(w2v_hash(d.seed, (uint32_t)d.iter, (uint32_t)s, (uint32_t)pos) % (uint32_t)d.window);
      const int32_t c0 = d.code_off[word], c1 = d.code_off[word + 1];
      for (int32_t a = b; a < d.window * 2 + 1 - b; ++a) {
        if (a == d.window) continue;
        const int32_t c = pos - d.window + a;
        if (c < 0 || c >= len) continue;
        const int32_t last = d.sent[o0 + c];
        float *r0 = d.syn0 + (int64_t)last * D;
        float v0[ND], neu[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) { const int j = lane + 64 * i; v0[i] = j < D ? r0[j] : 0.0f; neu[i] = 0.0f; }
        for (int32_t t = c0; t < c1; ++t) {
          float *r1 = d.syn1 + (int64_t)d.points[t] * D;
          float v1[ND];
          float part = 0.0f;
#pragma unroll
          for (int i = 0; i < ND; ++i) { const int j = lane + 64 * i; v1[i] = j < D ? r1[j] : 0.0f; part += v0[i] * v1[i]; }
          const float f = wave_sum_f32(part);
          if (f > -MAX_EXP && f < MAX_EXP) {
            const int ind = (int)((f + MAX_EXP) * ((float)EXP_TABLE_SIZE / MAX_EXP / 2.0f));
            const float g = (1.0f - (float)d.codes[t] - exp_table[ind]) * alpha;
#pragma unroll
            for (int i = 0; i < ND; ++i) {
              const int j = lane + 64 * i;
              neu[i] += g * v1[i];
              if (j < D) r1[j] = v1[i] + g * v0[i];
            }
          }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) { const int j = lane + 64 * i; if (j < D) r0[j] = v0[i] + neu[i]; }
      }
    }
  }
}

// Sum over the wave by DPP (row shifts, then the two row broadcasts: lane 63 ends up with the whole wave's) instead of six
// ds_bpermute butterflies — the dot product of a node update is the head of its dependent chain.
template <int CTRL, int ROW_MASK>
__device__ inline float dpp_add_f32(float v) {      // lanes without a source (or outside the row mask) add 0
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true));
}
__device__ inline float wave_sum_f32_dpp(float v) {
  v = dpp_add_f32<0x111, 0xF>(v); v = dpp_add_f32<0x112, 0xF>(v); v = dpp_add_f32<0x114, 0xF>(v); v = dpp_add_f32<0x118, 0xF>(v);
  v = dpp_add_f32<0x142, 0xA>(v); v = dpp_add_f32<0x143, 0xC>(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Four sums over the wave at the price of one and a half: two quad exchanges leave lane l with the quad's partial sum of value (l & 3),
// two row shifts by multiples of four add the quads of a row (lanes 12..15 hold the row's), two butterflies add the rows.
template <int CTRL>
__device__ inline float dpp_mov_f32(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ inline float wave_sum4_f32(float a, float b, float c, float d, int lane) {
  const bool odd = (lane & 1) != 0, hi = (lane & 2) != 0;
  const float ab = (odd ? b : a) + dpp_mov_f32<0xB1>(odd ? a : b);        // quad_perm [1,0,3,2]: even lanes a-pairs, odd lanes b-pairs
  const float cd = (odd ? d : c) + dpp_mov_f32<0xB1>(odd ? c : d);
  float x = (hi ? cd : ab) + dpp_mov_f32<0x4E>(hi ? ab : cd);             // quad_perm [2,3,0,1]: lane & 3 -> a, b, c, d over the quad
  x += dpp_mov_f32<0x114>(x);                                             // row_shr:4, row_shr:8 (lanes without a source add 0)
  x += dpp_mov_f32<0x118>(x);
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}

// The training kernel (round 5).  MLlib's loop (and word2vec.c's) visits, for ONE position, every context word of the window and for
// each of them every Huffman node of the CENTRE word: the same <= 40 rows of syn1, read and written once per context — ~11 times at
// window 10.  Here a position loads the centre word's rows ONCE into registers (R rows of ND floats per lane; the nodes of a longer
// code go through memory as before), runs its contexts over them — same operations in the same order, so the sequential mode computes
// what it did — and writes them back once; the next context's syn0 row is fetched while the current one's node chain runs (a context
// word that repeats takes the row just written).  Everything that is the wave's (sentence, position, code, node index, f, g) stays in
// scalar registers.  RMAT-20, one walk per vertex, dim 128, window 10: see profiles/r05_embedding.md.
template <int ND, int R>
__global__ __launch_bounds__(TPB) void k_w2v_train_rows(W2vDev d, int64_t n_waves) {
  __shared__ float exp_table[EXP_TABLE_SIZE];
  for (int i = threadIdx.x; i < EXP_TABLE_SIZE; i += blockDim.x) {
    const float e = (float)exp(((double)i / EXP_TABLE_SIZE * 2.0 - 1.0) * (double)MAX_EXP);
    exp_table[i] = e / (e + 1.0f);
  }
  __syncthreads();
  const int lane = lane_id();
  const int64_t wave = blockIdx.x * (int64_t)(TPB / 64) + (int64_t)uni((int32_t)(threadIdx.x >> 6));
  const int D = d.dim, W = d.window;
  const bool vecwin = 2 * W + 1 <= 64;              // the window's tokens in one register (lane a = context slot a)
  constexpr int G = 4;
  static_assert(R % G == 0, "register rows come in groups");
  for (int64_t s = wave; s < d.n_sent; s += n_waves) {
    const int64_t o0 = uni(d.sent_off[s]), o1 = uni(d.sent_off[s + 1]);
    const int32_t len = (int32_t)(o1 - o0);
    float alpha;
    {
      const double done = (double)d.iter * (double)d.total_words + (double)uni(d.words_before[s]);
      double a = (double)d.lr * (1.0 - done / ((double)d.n_iter * (double)d.total_words + 1.0));
      if (a < (double)d.lr * 0.0001) a = (double)d.lr * 0.0001;
      alpha = (float)a;
    }
    for (int32_t pos = 0; pos < len; ++pos) {
      const int32_t word = uni(d.sent[o0 + pos]);
      const int32_t b = (int32_t)(w2v_hash(d.seed, (uint32_t)d.iter, (uint32_t)s, (uint32_t)pos) % (uint32_t)W);
      const int32_t c0 = uni(d.code_off[word]), n = uni(d.code_off[word + 1]) - c0;
      const int32_t nr = n < R ? n : R;
      int32_t pts = 0, cds = 0;                     // lane t: node t of the centre word's path (n <= 40 < 64)
      if (lane < n) { pts = d.points[c0 + lane]; cds = d.codes[c0 + lane]; }
      const unsigned long long code_bits = __ballot(cds != 0);
      int32_t wtok = -1;
      if (vecwin) { const int32_t c = pos - W + lane; if (lane <= 2 * W && c >= 0 && c < len) wtok = d.sent[o0 + c]; }
      auto tok = [&](int32_t a) { return vecwin ? __builtin_amdgcn_readlane(wtok, a) : uni(d.sent[o0 + pos - W + a]); };
      const int32_t a_end = 2 * W + 1 - b;
      auto next_ctx = [&](int32_t a) {              // the next context slot after a that is a word of the sentence
        for (++a; a < a_end; ++a) { const int32_t c = pos - W + a; if (a != W && c >= 0 && c < len) break; }
        return a;
      };
      float v1[R][ND];
#pragma unroll
      for (int t = 0; t < R; ++t) {
#pragma unroll
        for (int i = 0; i < ND; ++i) v1[t][i] = 0.0f;
        if (t < nr) {
          const float *r1 = d.syn1 + (int64_t)__builtin_amdgcn_readlane(pts, t) * D;
#pragma unroll
          for (int i = 0; i < ND; ++i) { const int j = lane + 64 * i; if (j < D) v1[t][i] = r1[j]; }
        }
      }
      int32_t a = next_ctx(b - 1), last = -1;
      float v0[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i) v0[i] = 0.0f;
      if (a < a_end) {
        last = tok(a);
        const float *r0 = d.syn0 + (int64_t)last * D;
#pragma unroll
        for (int i = 0; i < ND; ++i) { const int j = lane + 64 * i; if (j < D) v0[i] = r0[j]; }
      }
      while (a < a_end) {
        const int32_t an = next_ctx(a);
        int32_t lastn = -1;
        float v0n[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) v0n[i] = 0.0f;
        if (an < a_end) {
          lastn = tok(an);
          if (lastn != last) {
            const float *rn = d.syn0 + (int64_t)lastn * D;
#pragma unroll
            for (int i = 0; i < ND; ++i) { const int j = lane + 64 * i; if (j < D) v0n[i] = rn[j]; }
          }
        }
        float neu[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) neu[i] = 0.0f;
        // the nodes of one context are independent of each other (v0 changes after the context, every node has its own row): four at
        // a time without a branch between them, so that their reductions and table look-ups overlap; a slot past the code (its row is
        // zero and never written back) and an |f| >= 6 get g = 0
#pragma unroll
        for (int t0 = 0; t0 < R; t0 += G) {
          if (t0 < nr) {
            float part[G], g[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
              part[u] = 0.0f;
#pragma unroll
              for (int i = 0; i < ND; ++i) part[u] += v0[i] * v1[t0 + u][i];
            }
            // the four dot products in ONE reduction (wave_sum4_f32: lanes 12..15 of every row end up with node t0 + (lane & 3)'s), then
            // f -> g once, in those lanes, for all four
            const float f = wave_sum4_f32(part[0], part[1], part[2], part[3], lane);
            const int u_l = lane & 3;
            const bool ok = f > -MAX_EXP && f < MAX_EXP && t0 + u_l < nr;
            const int ind = ok ? (int)((f + MAX_EXP) * ((float)EXP_TABLE_SIZE / MAX_EXP / 2.0f)) : 0;
            const float code = (float)(((uint32_t)(code_bits >> t0) >> u_l) & 1u);
            const float gv = ok ? (1.0f - code - exp_table[ind]) * alpha : 0.0f;
#pragma unroll
            for (int u = 0; u < G; ++u) g[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gv), 12 + u));
#pragma unroll
            for (int u = 0; u < G; ++u) {
#pragma unroll
              for (int i = 0; i < ND; ++i) { neu[i] += g[u] * v1[t0 + u][i]; v1[t0 + u][i] = v1[t0 + u][i] + g[u] * v0[i]; }
            }
          }
        }
        for (int32_t t = R; t < n; ++t) {           // a code longer than the register rows (rare words): through memory
          float *r1 = d.syn1 + (int64_t)__builtin_amdgcn_readlane(pts, t) * D;
          float w1[ND];
          float part = 0.0f;
#pragma unroll
          for (int i = 0; i < ND; ++i) { const int j = lane + 64 * i; w1[i] = j < D ? r1[j] : 0.0f; part += v0[i] * w1[i]; }
          const float f = wave_sum_f32_dpp(part);
          if (f > -MAX_EXP && f < MAX_EXP) {
            const int ind = (int)((f + MAX_EXP) * ((float)EXP_TABLE_SIZE / MAX_EXP / 2.0f));
            const float g = (1.0f - (float)((code_bits >> t) & 1ull) - exp_table[ind]) * alpha;
#pragma unroll
            for (int i = 0; i < ND; ++i) { const int j = lane + 64 * i; neu[i] += g * w1[i]; if (j < D) r1[j] = w1[i] + g * v0[i]; }
          }
        }
        {
          float *r0 = d.syn0 + (int64_t)last * D;
#pragma unroll
          for (int i = 0; i < ND; ++i) { const int j = lane + 64 * i; v0[i] = v0[i] + neu[i]; if (j < D) r0[j] = v0[i]; }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) v0[i] = (an < a_end && lastn == last) ? v0[i] : v0n[i];
        last = lastn; a = an;
      }
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if (t < nr) {
          float *r1 = d.syn1 + (int64_t)__builtin_amdgcn_readlane(pts, t) * D;
#pragma unroll
          for (int i = 0; i < ND; ++i) { const int j = lane + 64 * i; if (j < D) r1[j] = v1[t][i]; }
        }
      }
    }
  }
}

// word2vec.c's CreateBinaryTree over counts sorted in DESCENDING order: codes and inner-node paths of every word
void huffman(const std::vector<int64_t> &cn, std::vector<int32_t> &code_off, std::vector<int32_t> &points, std::vector<uint8_t> &codes) {
  const int64_t V = (int64_t)cn.size();
  code_off.assign((size_t)V + 1, 0);
  points.clear(); codes.clear();
  if (V == 0) return;
  if (V == 1) { code_off[1] = 0; return; }
  std::vector<int64_t> count((size_t)V * 2 + 1, (int64_t)1e15);
  std::vector<int32_t> parent((size_t)V * 2 + 1, 0);
  std::vector<uint8_t> binary((size_t)V * 2 + 1, 0);
  for (int64_t a = 0; a < V; ++a) count[(size_t)a] = cn[(size_t)a];
  int64_t pos1 = V - 1, pos2 = V, min1, min2;
  for (int64_t a = 0; a < V - 1; ++a) {
    if (pos1 >= 0 && count[(size_t)pos1] < count[(size_t)pos2]) { min1 = pos1; --pos1; } else { min1 = pos2; ++pos2; }
    if (pos1 >= 0 && count[(size_t)pos1] < count[(size_t)pos2]) { min2 = pos1; --pos1; } else { min2 = pos2; ++pos2; }
    count[(size_t)(V + a)] = count[(size_t)min1] + count[(size_t)min2];
    parent[(size_t)min1] = (int32_t)(V + a); parent[(size_t)min2] = (int32_t)(V + a);
    binary[(size_t)min2] = 1;
  }
  uint8_t code[MAX_CODE_LENGTH + 1]; int32_t point[MAX_CODE_LENGTH + 1];
  for (int64_t a = 0; a < V; ++a) {
    int64_t b = a; int i = 0;
    while (true) {
      if (i >= MAX_CODE_LENGTH) throw Error(SRW_ERR_INVALID, "word2vec: a Huffman code exceeds 40 bits");
      code[i] = binary[(size_t)b]; point[i] = (int32_t)b; ++i;
      b = parent[(size_t)b];
      if (b == V * 2 - 2) break;
    }
    code_off[(size_t)a + 1] = code_off[(size_t)a] + i;
    points.push_back((int32_t)(V - 2));                       // the root
    for (int k = 0; k < i; ++k) {
      codes.push_back(code[i - k - 1]);
      if (k + 1 < i) points.push_back(point[i - k - 1] - (int32_t)V);
    }
  }
}
// ---- vocabulary and sentences on the device ----------------------------------------------------------------------------------
// The reference hands the walk's RDD straight to Word2Vec.fit (M/Main.scala:113-117); here the paths of srw_walk are in HBM already
// (srw_device_paths) and stay there: lengths -> sentence offsets (scan), tokens flattened, sorted (radix) and run-length encoded into
// (id, count), ordered by descending count (stable: ties by ascending id — minCount 0, Main.scala:41), every token replaced by its
// vocabulary index by a search in the sorted ids.  Only the counts travel to the host (the Huffman tree is a sequential O(V) pass).
__global__ void k_w2v_len64(const int32_t *__restrict__ lens, int64_t n, int64_t stride, long long *__restrict__ out, uint32_t *bad) {
  const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (w > n) return;
  int32_t l = w < n ? lens[w] : 0;
  if (l < 0 || l > stride) { atomicOr(bad, 1u); l = 0; }
  out[w] = l;
}
__global__ __launch_bounds__(TPB) void k_w2v_flatten(const int32_t *__restrict__ paths, const long long *__restrict__ off, int64_t n, int64_t stride,
                                                     int32_t *__restrict__ flat) {
  const int lane = lane_id();
  const int64_t w = blockIdx.x * (int64_t)(TPB / 64) + (threadIdx.x >> 6);
  if (w >= n) return;
  const int64_t o0 = off[w], len = off[w + 1] - o0;
  for (int64_t k = lane; k < len; k += 64) flat[o0 + k] = paths[w * stride + k];
}
__global__ void k_w2v_widen(const unsigned int *__restrict__ in, int64_t n, unsigned long long *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void k_w2v_rank(const int32_t *__restrict__ order, const int32_t *__restrict__ uniq, int64_t V, int32_t *__restrict__ rank_of,
                           int32_t *__restrict__ vocab) {
  const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (r >= V) return;
  const int32_t u = order[r];
  rank_of[u] = (int32_t)r; vocab[r] = uniq[u];
}
__global__ void k_w2v_remap(int32_t *__restrict__ flat, int64_t total, const int32_t *__restrict__ uniq, int64_t V, const int32_t *__restrict__ rank_of) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int32_t x = flat[i];
  int64_t lo = 0, hi = V - 1;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (uniq[mid] < x) lo = mid + 1; else hi = mid; }
  flat[i] = rank_of[lo];
}
__global__ void k_w2v_init(float *__restrict__ syn0, int64_t V, int32_t dim, uint32_t seed) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= V * dim) return;
  const uint32_t r = (uint32_t)(i / dim), j = (uint32_t)(i % dim);
  syn0[i] = ((float)(w2v_hash(seed, 0xA11CEu, r, j) >> 8) * (1.0f / 16777216.0f) - 0.5f) / (float)dim;
}
}  // namespace

// paths [n][stride] (ids, lens) in HBM -> vocabulary (ids by descending count, ties by ascending id) + vectors [vocab][dim]
void w2v_fit_device(srw_handle *h, const int32_t *d_paths, const int32_t *d_lens, int64_t n, int64_t stride, const srw_w2v_params &P,
                    std::vector<int32_t> &vocab_ids, std::vector<float> &vectors) {
  if (P.dim < 1 || P.dim > 1024 || P.window < 1 || P.iterations < 0 || !(P.learning_rate > 0.0f))
    throw Error(SRW_ERR_INVALID, "word2vec: dim in 1..1024, window >= 1, iterations >= 0, learning rate > 0");
  hipStream_t st = h->stream;
  vocab_ids.clear(); vectors.clear();
  // sentence offsets
  DevBuf<long long> d_off; d_off.alloc((size_t)n + 1);
  DevBuf<uint32_t> d_bad; d_bad.alloc(1);
  SRW_HIP(hipMemsetAsync(d_bad.p, 0, 4, st));
  hipLaunchKernelGGL(k_w2v_len64, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, st, d_lens, n, stride, d_off.p, d_bad.p);
  SRW_HIP(hipGetLastError());
  DevBuf<char> temp; size_t tb = 0;
  SRW_HIP(rocprim::exclusive_scan(nullptr, tb, d_off.p, d_off.p, 0ll, (size_t)n + 1, rocprim::plus<long long>(), st));
  temp.alloc(tb);
  SRW_HIP(rocprim::exclusive_scan((void *)temp.p, tb, d_off.p, d_off.p, 0ll, (size_t)n + 1, rocprim::plus<long long>(), st));
  long long total = 0; uint32_t bad = 0;
  SRW_HIP(hipMemcpyAsync(&total, d_off.p + n, 8, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipMemcpyAsync(&bad, d_bad.p, 4, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipStreamSynchronize(st));
  if (bad) throw Error(SRW_ERR_INVALID, "word2vec: a path length outside [0, stride]");
  if (total == 0) return;
  // tokens, then the vocabulary (id, count) — chunk by chunk (the reference's defaults numWalks = 10, walkLength = 80 exceed
  // 2^32 tokens from ~5.3 M vertices on, and rocprim's run_length_encode takes a 32-bit size): every chunk of at most 2^30 tokens is
  // sorted and run-length encoded on its own, its (id, count) runs are merged into the standing set (sort by id + reduce_by_key), the
  // counts are 64-bit.  SRW_W2V_VOCAB_CHUNK (tokens): tests merge many small chunks.
  DevBuf<int32_t> d_sent, d_sorted, d_uniq_c; DevBuf<unsigned int> d_cnt_c, d_runs;
  d_sent.alloc((size_t)total);
  hipLaunchKernelGGL(k_w2v_flatten, dim3((unsigned)((n + TPB / 64 - 1) / (TPB / 64))), dim3(TPB), 0, st, d_paths, d_off.p, n, stride, d_sent.p);
  SRW_HIP(hipGetLastError());
  size_t CH = (size_t)1 << 30;
  if (const char *e = getenv("SRW_W2V_VOCAB_CHUNK"); e && atoll(e) > 0) CH = (size_t)atoll(e);
  const size_t ch_max = std::min<size_t>(CH, (size_t)total);
  d_sorted.alloc(ch_max); d_uniq_c.alloc(ch_max); d_cnt_c.alloc(ch_max); d_runs.alloc(1);
  DevBuf<int32_t> d_uniq; DevBuf<unsigned long long> d_cnt;       // the standing set: ids ascending, their counts
  size_t R = 0;
  for (size_t c0 = 0; c0 < (size_t)total; c0 += CH) {
    const size_t nc = std::min(CH, (size_t)total - c0);
    SRW_HIP(rocprim::radix_sort_keys(nullptr, tb, d_sent.p + c0, d_sorted.p, nc, 0, 32, st));
    temp.alloc(tb);
    SRW_HIP(rocprim::radix_sort_keys((void *)temp.p, tb, d_sent.p + c0, d_sorted.p, nc, 0, 32, st));
    SRW_HIP(rocprim::run_length_encode(nullptr, tb, d_sorted.p, (unsigned int)nc, d_uniq_c.p, d_cnt_c.p, d_runs.p, st));
    temp.alloc(tb);
    SRW_HIP(rocprim::run_length_encode((void *)temp.p, tb, d_sorted.p, (unsigned int)nc, d_uniq_c.p, d_cnt_c.p, d_runs.p, st));
    unsigned int runs = 0;
    SRW_HIP(hipMemcpyAsync(&runs, d_runs.p, 4, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
    // standing set + this chunk's runs -> sorted by id -> equal ids summed
    const size_t m = R + runs;
    DevBuf<int32_t> k_in, k_sorted, k_out; DevBuf<unsigned long long> v_in, v_sorted, v_out; DevBuf<unsigned long long> d_m;
    k_in.alloc(m); v_in.alloc(m);
    if (R) {
      SRW_HIP(hipMemcpyAsync(k_in.p, d_uniq.p, R * 4, hipMemcpyDeviceToDevice, st));
      SRW_HIP(hipMemcpyAsync(v_in.p, d_cnt.p, R * 8, hipMemcpyDeviceToDevice, st));
    }
    SRW_HIP(hipMemcpyAsync(k_in.p + R, d_uniq_c.p, (size_t)runs * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_w2v_widen, dim3((unsigned)((runs + 255) / 256)), dim3(256), 0, st, d_cnt_c.p, (int64_t)runs, v_in.p + R);
    SRW_HIP(hipGetLastError());
    if (R == 0) { d_uniq = std::move(k_in); d_cnt = std::move(v_in); R = m; continue; }      // (the first chunk's runs are sorted and distinct already)
    k_sorted.alloc(m); v_sorted.alloc(m); k_out.alloc(m); v_out.alloc(m); d_m.alloc(1);
    SRW_HIP(rocprim::radix_sort_pairs(nullptr, tb, k_in.p, k_sorted.p, v_in.p, v_sorted.p, m, 0, 32, st));
    temp.alloc(tb);
    SRW_HIP(rocprim::radix_sort_pairs((void *)temp.p, tb, k_in.p, k_sorted.p, v_in.p, v_sorted.p, m, 0, 32, st));
    SRW_HIP(rocprim::reduce_by_key(nullptr, tb, k_sorted.p, v_sorted.p, m, k_out.p, v_out.p, d_m.p, rocprim::plus<unsigned long long>(), rocprim::equal_to<int32_t>(), st));
    temp.alloc(tb);
    SRW_HIP(rocprim::reduce_by_key((void *)temp.p, tb, k_sorted.p, v_sorted.p, m, k_out.p, v_out.p, d_m.p, rocprim::plus<unsigned long long>(), rocprim::equal_to<int32_t>(), st));
    unsigned long long merged = 0;
    SRW_HIP(hipMemcpyAsync(&merged, d_m.p, 8, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
    d_uniq = std::move(k_out); d_cnt = std::move(v_out); R = (size_t)merged;
  }
  d_sorted.release(); d_uniq_c.release(); d_cnt_c.release();
  const int64_t V = (int64_t)R;
  // vocabulary order: count descending, ties by ascending id (a stable sort of the id-ordered runs)
  DevBuf<unsigned long long> d_cnt_s; DevBuf<int32_t> d_iota, d_order, d_rank, d_vocab;
  d_cnt_s.alloc((size_t)V); d_iota.alloc((size_t)V); d_order.alloc((size_t)V); d_rank.alloc((size_t)V); d_vocab.alloc((size_t)V);
  {
    std::vector<int32_t> iota((size_t)V);
    std::iota(iota.begin(), iota.end(), 0);
    SRW_HIP(hipMemcpyAsync(d_iota.p, iota.data(), (size_t)V * 4, hipMemcpyHostToDevice, st));
    SRW_HIP(rocprim::radix_sort_pairs_desc(nullptr, tb, d_cnt.p, d_cnt_s.p, d_iota.p, d_order.p, (size_t)V, 0, 64, st));
    temp.alloc(tb);
    SRW_HIP(rocprim::radix_sort_pairs_desc((void *)temp.p, tb, d_cnt.p, d_cnt_s.p, d_iota.p, d_order.p, (size_t)V, 0, 64, st));
    SRW_HIP(hipStreamSynchronize(st));      // (iota is a host vector)
  }
  hipLaunchKernelGGL(k_w2v_rank, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, d_order.p, d_uniq.p, V, d_rank.p, d_vocab.p);
  SRW_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_w2v_remap, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d_sent.p, (int64_t)total, d_uniq.p, V, d_rank.p);
  SRW_HIP(hipGetLastError());
  vocab_ids.resize((size_t)V);
  std::vector<unsigned long long> cnt_h((size_t)V);
  SRW_HIP(hipMemcpyAsync(vocab_ids.data(), d_vocab.p, (size_t)V * 4, hipMemcpyDeviceToHost, st));
  SRW_HIP(hipMemcpyAsync(cnt_h.data(), d_cnt_s.p, (size_t)V * 8, hipMemcpyDeviceToHost, st));
  DevBuf<float> d_syn0, d_syn1;
  d_syn0.alloc((size_t)V * P.dim); d_syn1.alloc((size_t)V * P.dim);
  hipLaunchKernelGGL(k_w2v_init, dim3((unsigned)(((int64_t)V * P.dim + 255) / 256)), dim3(256), 0, st, d_syn0.p, V, P.dim, P.seed);
  SRW_HIP(hipGetLastError());
  SRW_HIP(hipMemsetAsync(d_syn1.p, 0, (size_t)V * P.dim * 4, st));
  SRW_HIP(hipStreamSynchronize(st));
  d_uniq.release(); d_cnt.release(); d_iota.release(); d_order.release(); d_rank.release(); d_vocab.release(); d_cnt_s.release();
  vectors.resize((size_t)V * (size_t)P.dim);
  if (V >= 2 && P.iterations > 0) {
    std::vector<int64_t> cn((size_t)V);
    for (int64_t r = 0; r < V; ++r) cn[(size_t)r] = (int64_t)cnt_h[(size_t)r];
    std::vector<int32_t> code_off, points; std::vector<uint8_t> codes;
    huffman(cn, code_off, points, codes);
    DevBuf<int32_t> d_coff, d_points; DevBuf<uint8_t> d_codes;
    d_coff.alloc((size_t)V + 1); d_points.alloc(std::max<size_t>(points.size(), 1)); d_codes.alloc(std::max<size_t>(codes.size(), 1));
    SRW_HIP(hipMemcpyAsync(d_coff.p, code_off.data(), ((size_t)V + 1) * 4, hipMemcpyHostToDevice, st));
    SRW_HIP(hipMemcpyAsync(d_points.p, points.data(), points.size() * 4, hipMemcpyHostToDevice, st));
    SRW_HIP(hipMemcpyAsync(d_codes.p, codes.data(), codes.size(), hipMemcpyHostToDevice, st));
    W2vDev d;
    d.sent_off = reinterpret_cast<const int64_t *>(d_off.p); d.sent = d_sent.p; d.n_sent = n; d.words_before = reinterpret_cast<const int64_t *>(d_off.p);
    d.code_off = d_coff.p; d.points = d_points.p; d.codes = d_codes.p;
    d.syn0 = d_syn0.p; d.syn1 = d_syn1.p; d.dim = P.dim; d.window = P.window; d.seed = P.seed; d.n_iter = P.iterations; d.total_words = total; d.lr = P.learning_rate;
    const int nd = (P.dim + 63) / 64;
    static const bool rows_in_memory = getenv("SRW_W2V_ROWS_IN_MEMORY") && atoi(getenv("SRW_W2V_ROWS_IN_MEMORY")) != 0;   // (the round-4 kernel, for A / B)
    void (*kern)(W2vDev, int64_t);
    if (rows_in_memory) kern = nd <= 1 ? k_w2v_train<1> : nd <= 2 ? k_w2v_train<2> : nd <= 4 ? k_w2v_train<4> : nd <= 8 ? k_w2v_train<8> : k_w2v_train<16>;
    else kern = nd <= 1 ? k_w2v_train_rows<1, 32> : nd <= 2 ? k_w2v_train_rows<2, 24> : nd <= 4 ? k_w2v_train_rows<4, 24>
              : nd <= 8 ? k_w2v_train_rows<8, 8> : k_w2v_train_rows<16, 4>;
    // threads == 1: ONE wave walks the sentences in order (the sequential form the oracle restates); else one wave per sentence at a
    // time, Hogwild, as many waves as the GPU holds at once (sentences are of one length: a second, partial round of waves would idle
    // a third of the chip)
    int per_cu = 0;
    SRW_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), TPB, 0));
    const int64_t resident = (int64_t)h->n_cus * std::max(per_cu, 1) * (TPB / 64);
    const int64_t n_waves = P.threads == 1 ? 1 : std::min<int64_t>(std::max<int64_t>(n, 1), resident);
    const int blocks = (int)((n_waves + TPB / 64 - 1) / (TPB / 64));
    for (int32_t k = 0; k < P.iterations; ++k) {
      d.iter = k;
      const dim3 grid(P.threads == 1 ? 1 : blocks), block(P.threads == 1 ? 64 : TPB);
      hipLaunchKernelGGL(kern, grid, block, 0, st, d, n_waves);
      SRW_HIP(hipGetLastError());
    }
    SRW_HIP(hipMemcpyAsync(vectors.data(), d_syn0.p, vectors.size() * 4, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
  } else {
    SRW_HIP(hipMemcpyAsync(vectors.data(), d_syn0.p, vectors.size() * 4, hipMemcpyDeviceToHost, st));
    SRW_HIP(hipStreamSynchronize(st));
  }
}

// test hook: the codes (code_len[V], codes[V][40], points[V][40]: inner-node rows of syn1, root first) of a vocabulary by its counts
void w2v_huffman(const int64_t *counts, int64_t n_vocab, int32_t *code_len, uint8_t *codes_out, int32_t *points_out) {
  std::vector<int64_t> cn(counts, counts + n_vocab);
  for (int64_t a = 1; a < n_vocab; ++a) if (cn[(size_t)a] > cn[(size_t)a - 1]) throw Error(SRW_ERR_INVALID, "word2vec: counts must be in descending order");
  std::vector<int32_t> code_off, points; std::vector<uint8_t> codes;
  huffman(cn, code_off, points, codes);
  for (int64_t a = 0; a < n_vocab; ++a) {
    const int32_t c0 = code_off[(size_t)a], len = code_off[(size_t)a + 1] - c0;
    code_len[a] = len;
    for (int32_t k = 0; k < MAX_CODE_LENGTH; ++k) {
      codes_out[a * MAX_CODE_LENGTH + k] = k < len ? codes[(size_t)(c0 + k)] : 0;
      points_out[a * MAX_CODE_LENGTH + k] = k < len ? points[(size_t)(c0 + k)] : -1;
    }
  }
}

// the same from HOST paths (a parsed paths file, `--cmd embedding`): one upload, then the device form
void w2v_fit(srw_handle *h, const int32_t *paths, const int32_t *lens, int64_t n, int64_t stride, const srw_w2v_params &P,
             std::vector<int32_t> &vocab_ids, std::vector<float> &vectors) {
  for (int64_t w = 0; w < n; ++w)
    if (lens[w] < 0 || lens[w] > stride) throw Error(SRW_ERR_INVALID, "word2vec: a path length outside [0, stride]");
  DevBuf<int32_t> d_paths, d_lens;
  d_paths.alloc((size_t)std::max<int64_t>(n * stride, 1)); d_lens.alloc((size_t)std::max<int64_t>(n, 1));
  SRW_HIP(hipMemcpyAsync(d_paths.p, paths, (size_t)(n * stride) * 4, hipMemcpyHostToDevice, h->stream));
  SRW_HIP(hipMemcpyAsync(d_lens.p, lens, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
  SRW_HIP(hipStreamSynchronize(h->stream));
  w2v_fit_device(h, d_paths.p, d_lens.p, n, stride, P, vocab_ids, vectors);
}
}  // namespace srw
