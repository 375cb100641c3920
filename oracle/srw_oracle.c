/*
 * srw_oracle.c — CPU restatement (plain C) of the reference's random-walk hot path.
 * TEST INFRASTRUCTURE ONLY — see srw_oracle.h.  Parity pinned against the reference's own
 * known-answer tests (tests/test_oracle_reference_vectors.py).
 *
 * Compile with -O2 -ffp-contract=off -fno-fast-math: the f64 accumulation order below IS the spec.
 */
#define _GNU_SOURCE
#include "srw_oracle.h"

#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

/* ============================================================================================ */
/* RNG                                                                                          */
/* ============================================================================================ */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float orc_walk_uniform(uint32_t seed, uint32_t iter, uint32_t src, uint32_t step) {
  uint32_t ctr[4] = {iter, src, step, 0u}, key[2] = {seed, 0u}, out[4];
  orc_philox4x32_10(ctr, key, out);
  return (float)(out[0] >> 8) * (1.0f / 16777216.0f); /* exact: 24-bit integer * 2^-24 */
}

void orc_java_random_floats(int64_t seed, int n, float *out) {
  uint64_t s = ((uint64_t)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
  for (int i = 0; i < n; ++i) {
    s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
    out[i] = (float)(uint32_t)(s >> 24) / (float)(1 << 24);
  }
}

/* ============================================================================================ */
/* Sampler — M/algorithm/RandomSample.scala                                                     */
/* ============================================================================================ */

/* RandomSample.sample, RandomSample.scala:12-25.
 *   :14  sum = foldLeft(0.0)(w1 + w2)      left-to-right, Double accumulator, Float widened
 *   :16  p = nextFloat()                   (the caller supplies r)
 *   :18-22 acc += w / sum ; first acc >= p  (Float/Double -> f64 divide, f64 add, f64 compare)
 *   :24  edges.head                         fallback */
int64_t orc_sample_index(const float *w, int64_t n, float r) {
  if (n <= 0) return -1;
  double sum = 0.0;
  for (int64_t k = 0; k < n; ++k) sum = sum + (double)w[k];
  double p = (double)r;
  double acc = 0.0;
  for (int64_t k = 0; k < n; ++k) {
    acc += (double)w[k] / sum;
    if (acc >= p) return k;
  }
  return 0;
}

/* RandomSample.computeSecondOrderWeights, RandomSample.scala:27-44 — linear `exists`, f32 divides. */
void orc_second_order_weights(float p, float q, int32_t prev_id, const int32_t *prev_ids, int64_t n_prev,
                              const int32_t *curr_ids, const float *curr_w, int64_t n, float *out_w) {
  for (int64_t k = 0; k < n; ++k) {
    int32_t dst = curr_ids[k];
    float w = curr_w[k];
    float unnorm = w / q;                 /* :33 */
    if (dst == prev_id) {
      unnorm = w / p;                     /* :35 */
    } else {
      for (int64_t j = 0; j < n_prev; ++j) /* :37 prevNeighbors.exists(_._1 == dstId) */
        if (prev_ids[j] == dst) { unnorm = w; break; }
    }
    out_w[k] = unnorm;
  }
}

int64_t orc_second_order_sample_index(float p, float q, int32_t prev_id, const int32_t *prev_ids,
                                      int64_t n_prev, const int32_t *curr_ids, const float *curr_w,
                                      int64_t n, float r) {
  if (n <= 0) return -1;
  float *tmp = (float *)malloc(sizeof(float) * (size_t)n);
  orc_second_order_weights(p, q, prev_id, prev_ids, n_prev, curr_ids, curr_w, n, tmp);
  int64_t k = orc_sample_index(tmp, n, r);
  free(tmp);
  return k;
}

/* ============================================================================================ */
/* GraphMap — M/algorithm/GraphMap.scala                                                        */
/* ============================================================================================ */

typedef struct { int32_t key; int32_t val; uint8_t used; } imap_slot;
typedef struct { imap_slot *s; int64_t cap, n; } imap;

static void imap_init(imap *m) { m->cap = 64; m->n = 0; m->s = (imap_slot *)calloc((size_t)m->cap, sizeof(imap_slot)); }
static void imap_free(imap *m) { free(m->s); m->s = NULL; m->cap = m->n = 0; }
static uint64_t imap_hash(int32_t k) { uint64_t x = (uint32_t)k; x *= 0x9E3779B97F4A7C15ULL; return x >> 20; }
static imap_slot *imap_find(const imap *m, int32_t k) {
  uint64_t i = imap_hash(k) & (uint64_t)(m->cap - 1);
  while (m->s[i].used) { if (m->s[i].key == k) return &m->s[i]; i = (i + 1) & (uint64_t)(m->cap - 1); }
  return NULL;
}
static void imap_put(imap *m, int32_t k, int32_t v);
static void imap_grow(imap *m) {
  imap old = *m;
  m->cap = old.cap * 2; m->n = 0; m->s = (imap_slot *)calloc((size_t)m->cap, sizeof(imap_slot));
  for (int64_t i = 0; i < old.cap; ++i) if (old.s[i].used) imap_put(m, old.s[i].key, old.s[i].val);
  free(old.s);
}
static void imap_put(imap *m, int32_t k, int32_t v) {
  if ((m->n + 1) * 2 > m->cap) imap_grow(m);
  uint64_t i = imap_hash(k) & (uint64_t)(m->cap - 1);
  while (m->s[i].used) { if (m->s[i].key == k) { m->s[i].val = v; return; } i = (i + 1) & (uint64_t)(m->cap - 1); }
  m->s[i].used = 1; m->s[i].key = k; m->s[i].val = v; m->n++;
}

struct orc_graphmap {
  imap src_vertex_map;        /* srcVertexMap: id -> row index, -1 = no out-edges (GraphMap.scala:13) */
  imap vertex_partition_map;  /* vertexPartitionMap (:21) */
  int32_t *offsets, *lengths; /* :14-15 */
  int64_t rows_cap;
  int32_t *edge_ids; float *edge_w; /* edges (:16) */
  int64_t edges_cap;
  int32_t index_counter, offset_counter; /* :17-18 */
};

orc_graphmap *orc_graphmap_new(void) {
  orc_graphmap *g = (orc_graphmap *)calloc(1, sizeof(*g));
  imap_init(&g->src_vertex_map); imap_init(&g->vertex_partition_map);
  return g;
}
void orc_graphmap_free(orc_graphmap *g) {
  if (!g) return;
  imap_free(&g->src_vertex_map); imap_free(&g->vertex_partition_map);
  free(g->offsets); free(g->lengths); free(g->edge_ids); free(g->edge_w); free(g);
}
void orc_graphmap_reset(orc_graphmap *g) { /* GraphMap.reset :98-107 */
  imap_free(&g->src_vertex_map); imap_free(&g->vertex_partition_map);
  imap_init(&g->src_vertex_map); imap_init(&g->vertex_partition_map);
  g->index_counter = 0; g->offset_counter = 0;
}
static void gm_update_indices(orc_graphmap *g, int32_t v, int64_t out_degree) { /* updateIndices :58-64 */
  if (g->index_counter + 1 > g->rows_cap) {
    g->rows_cap = g->rows_cap ? g->rows_cap * 2 : 64;
    g->offsets = (int32_t *)realloc(g->offsets, sizeof(int32_t) * (size_t)g->rows_cap);
    g->lengths = (int32_t *)realloc(g->lengths, sizeof(int32_t) * (size_t)g->rows_cap);
  }
  imap_put(&g->src_vertex_map, v, g->index_counter);
  g->offsets[g->index_counter] = g->offset_counter;
  g->lengths[g->index_counter] = (int32_t)out_degree;
  g->index_counter++;
}
static void gm_push_edge(orc_graphmap *g, int32_t dst, float w) {
  if (g->offset_counter + 1 > g->edges_cap) {
    g->edges_cap = g->edges_cap ? g->edges_cap * 2 : 256;
    g->edge_ids = (int32_t *)realloc(g->edge_ids, sizeof(int32_t) * (size_t)g->edges_cap);
    g->edge_w = (float *)realloc(g->edge_w, sizeof(float) * (size_t)g->edges_cap);
  }
  g->edge_ids[g->offset_counter] = dst; g->edge_w[g->offset_counter] = w; g->offset_counter++;
}
void orc_graphmap_add_vertex_p(orc_graphmap *g, int32_t v, const int32_t *ids, const int32_t *pids,
                               const float *w, int64_t n) {
  if (imap_find(&g->src_vertex_map, v)) return;             /* case Some(value) => value (:37,:54) */
  if (n > 0) {
    gm_update_indices(g, v, n);
    for (int64_t k = 0; k < n; ++k) {
      gm_push_edge(g, ids[k], w[k]);
      if (pids) imap_put(&g->vertex_partition_map, ids[k], pids[k]); /* :31 */
    }
  } else {
    imap_put(&g->src_vertex_map, v, -1);                     /* addVertex(vId) :83-85 */
  }
}
void orc_graphmap_add_vertex(orc_graphmap *g, int32_t v, const int32_t *ids, const float *w, int64_t n) {
  orc_graphmap_add_vertex_p(g, v, ids, NULL, w, n);
}
int64_t orc_graphmap_num_vertices(const orc_graphmap *g) { return g->src_vertex_map.n; }
int64_t orc_graphmap_num_edges(const orc_graphmap *g) { return g->offset_counter; }
int64_t orc_graphmap_get_neighbors(const orc_graphmap *g, int32_t v, int32_t *ids, float *w, int64_t cap) {
  imap_slot *s = imap_find(&g->src_vertex_map, v);
  if (!s) return -1;            /* case None => null */
  if (s->val == -1) return 0;   /* Array.empty */
  int64_t off = g->offsets[s->val], len = g->lengths[s->val];
  for (int64_t k = 0; k < len && k < cap; ++k) {
    if (ids) ids[k] = g->edge_ids[off + k];
    if (w) w[k] = g->edge_w[off + k];
  }
  return len;
}
int orc_graphmap_get_partition(const orc_graphmap *g, int32_t v, int32_t *pid) {
  imap_slot *s = imap_find(&g->vertex_partition_map, v);
  if (!s) return 0;
  *pid = s->val; return 1;
}

/* ============================================================================================ */
/* Edge-list parsing — UniformRandomWalk.scala:23-43, VCutRandomWalk.scala:19-41                */
/* ============================================================================================ */

static int java_ws(unsigned char c) { return c == ' ' || c == '\t' || c == '\n' || c == 0x0B || c == '\f' || c == '\r'; }

/* Character.digit(ch, 10) of the UTF-16 char encoded as UTF-8 at s[*i ..): Integer.parseInt takes any decimal digit of the
 * Basic Multilingual Plane, char by char (a surrogate pair is never a digit); malformed UTF-8 is U+FFFD after Hadoop's
 * Text.toString, i.e. not a digit.  Zero code points of the BMP's Nd blocks (Unicode 6.2 = Java 8, + U+0DE6, U+A9F0). */
static const uint16_t ND_ZERO[] = {
    0x0660, 0x06F0, 0x07C0, 0x0966, 0x09E6, 0x0A66, 0x0AE6, 0x0B66, 0x0BE6, 0x0C66, 0x0CE6, 0x0D66, 0x0DE6, 0x0E50, 0x0ED0, 0x0F20,
    0x1040, 0x1090, 0x17E0, 0x1810, 0x1946, 0x19D0, 0x1A80, 0x1A90, 0x1B50, 0x1BB0, 0x1C40, 0x1C50, 0xA620, 0xA8D0, 0xA900, 0xA9D0,
    0xA9F0, 0xAA50, 0xABF0, 0xFF10};
static int java_digit(const char *s, size_t n, size_t *i) {
  unsigned char c = (unsigned char)s[*i];
  if (c < 0x80) { ++*i; return (c >= '0' && c <= '9') ? c - '0' : -1; }
  uint32_t cp;
  if ((c & 0xE0) == 0xC0 && *i + 1 < n && ((unsigned char)s[*i + 1] & 0xC0) == 0x80) {
    cp = ((uint32_t)(c & 0x1F) << 6) | ((unsigned char)s[*i + 1] & 0x3F);
    if (cp < 0x80) return -1;
    *i += 2;
  } else if ((c & 0xF0) == 0xE0 && *i + 2 < n && ((unsigned char)s[*i + 1] & 0xC0) == 0x80 && ((unsigned char)s[*i + 2] & 0xC0) == 0x80) {
    cp = ((uint32_t)(c & 0x0F) << 12) | (((uint32_t)(unsigned char)s[*i + 1] & 0x3F) << 6) | ((unsigned char)s[*i + 2] & 0x3F);
    if (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF)) return -1;
    *i += 3;
  } else return -1;
  for (size_t k = 0; k < sizeof(ND_ZERO) / sizeof(ND_ZERO[0]); ++k)
    if (cp >= ND_ZERO[k] && cp <= (uint32_t)ND_ZERO[k] + 9) return (int)(cp - ND_ZERO[k]);
  return -1;
}

/* java.lang.Integer.parseInt on [s, s+n): 1 ok, 0 NumberFormatException */
static int java_parse_int(const char *s, size_t n, int32_t *out) {
  if (n == 0) return 0;
  size_t i = 0; int neg = 0;
  if (s[0] == '-' || s[0] == '+') { neg = (s[0] == '-'); i = 1; if (n == 1) return 0; }
  int64_t v = 0;
  while (i < n) {
    int d = java_digit(s, n, &i);
    if (d < 0) return 0;
    v = v * 10 + d;
    if (v > 2147483648LL) return 0;
  }
  if (neg) v = -v;
  if (v > 2147483647LL || v < -2147483648LL) return 0;
  *out = (int32_t)v; return 1;
}

/* java.lang.Float.parseFloat on [s, s+n) (already whitespace-free): 1 ok, 0 NumberFormatException.
 * Grammar: [sign] (NaN | Infinity | HexFloat | Decimal) with optional f/F/d/D suffix; conversion is
 * correctly rounded to binary32 (glibc strtof). */
static int java_parse_float(const char *s, size_t n, float *out) {
  char buf[128];
  if (n == 0 || n >= sizeof(buf) - 1) return 0;
  size_t i = 0;
  int neg = 0;
  if (s[i] == '+' || s[i] == '-') { neg = (s[i] == '-'); ++i; }
  if (i >= n) return 0;
  if (n - i == 3 && memcmp(s + i, "NaN", 3) == 0) { *out = NAN; return 1; }
  if (n - i == 8 && memcmp(s + i, "Infinity", 8) == 0) { *out = neg ? -INFINITY : INFINITY; return 1; }
  size_t end = n;
  int hex = (n - i >= 2 && s[i] == '0' && (s[i + 1] == 'x' || s[i + 1] == 'X'));
  if (end > i && (s[end - 1] == 'f' || s[end - 1] == 'F' || s[end - 1] == 'd' || s[end - 1] == 'D')) {
    if (!hex) end--;            /* in a hex literal a trailing d/f before 'p' is a digit; after the */
    else {                      /* exponent it is a suffix */
      size_t pp = i; int seen_p = 0;
      for (; pp < end; ++pp) if (s[pp] == 'p' || s[pp] == 'P') { seen_p = 1; break; }
      if (seen_p) end--;
    }
  }
  size_t j = i;
  if (hex) {
    j += 2; int digits = 0;
    while (j < end && ((s[j] >= '0' && s[j] <= '9') || (s[j] >= 'a' && s[j] <= 'f') || (s[j] >= 'A' && s[j] <= 'F'))) { ++j; ++digits; }
    if (j < end && s[j] == '.') { ++j; while (j < end && ((s[j] >= '0' && s[j] <= '9') || (s[j] >= 'a' && s[j] <= 'f') || (s[j] >= 'A' && s[j] <= 'F'))) { ++j; ++digits; } }
    if (!digits) return 0;
    if (j >= end || (s[j] != 'p' && s[j] != 'P')) return 0; /* binary exponent is mandatory in Java */
    ++j; if (j < end && (s[j] == '+' || s[j] == '-')) ++j;
    int ed = 0; while (j < end && s[j] >= '0' && s[j] <= '9') { ++j; ++ed; }
    if (!ed || j != end) return 0;
  } else {
    int digits = 0;
    while (j < end && s[j] >= '0' && s[j] <= '9') { ++j; ++digits; }
    if (j < end && s[j] == '.') { ++j; while (j < end && s[j] >= '0' && s[j] <= '9') { ++j; ++digits; } }
    if (!digits) return 0;
    if (j < end && (s[j] == 'e' || s[j] == 'E')) {
      ++j; if (j < end && (s[j] == '+' || s[j] == '-')) ++j;
      int ed = 0; while (j < end && s[j] >= '0' && s[j] <= '9') { ++j; ++ed; }
      if (!ed) return 0;
    }
    if (j != end) return 0;
  }
  memcpy(buf, s, end); buf[end] = 0;
  char *ep = NULL;
  float v = strtof(buf, &ep);
  if (ep != buf + end) return 0;
  *out = v; return 1;
}

typedef struct { int32_t *src, *dst, *pid; float *w; int64_t n, cap; } line_vec;
static void lv_push(line_vec *v, int32_t s, int32_t d, float w, int32_t pid) {
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2 : 1024;
    v->src = (int32_t *)realloc(v->src, sizeof(int32_t) * (size_t)v->cap);
    v->dst = (int32_t *)realloc(v->dst, sizeof(int32_t) * (size_t)v->cap);
    v->pid = (int32_t *)realloc(v->pid, sizeof(int32_t) * (size_t)v->cap);
    v->w = (float *)realloc(v->w, sizeof(float) * (size_t)v->cap);
  }
  v->src[v->n] = s; v->dst[v->n] = d; v->w[v->n] = w; v->pid[v->n] = pid; v->n++;
}

/* One text line -> (src, dst, weight, pId).  Returns 0 and fills err where the reference throws. */
static int parse_line(const char *s, size_t n, int weighted, int partitioned, int64_t lineno,
                      int32_t *src, int32_t *dst, float *w, int32_t *pid, char *err, size_t errlen) {
  /* Java String.split("\\s+"): a separator at index 0 yields a leading "" token; trailing empty
   * tokens are dropped. */
  enum { MAXP = 64 };
  const char *tok[MAXP]; size_t len[MAXP]; int np = 0;
  size_t i = 0;
  if (n == 0) { tok[0] = s; len[0] = 0; np = 1; }           /* "".split -> [""] */
  else {
    if (java_ws((unsigned char)s[0])) { tok[np] = s; len[np] = 0; np++; }
    while (i < n) {
      while (i < n && java_ws((unsigned char)s[i])) ++i;
      if (i >= n) break;
      size_t b = i;
      while (i < n && !java_ws((unsigned char)s[i])) ++i;
      if (np < MAXP) { tok[np] = s + b; len[np] = i - b; np++; }
      else { tok[MAXP - 1] = s + b; len[MAXP - 1] = i - b; } /* keep "last" meaningful */
    }
    if (np == 1 && len[0] == 0) np = 0;                      /* all-whitespace line -> [] */
  }
  if (np < 2) { snprintf(err, errlen, "line %lld: fewer than two columns", (long long)lineno); return 0; }
  /* weight: UniformRandomWalk.scala:29-32 / VCutRandomWalk.scala:29-32 */
  *w = 1.0f;
  int wcols = partitioned ? 3 : 2;
  if (weighted && np > wcols) { float f; if (java_parse_float(tok[np - 1], len[np - 1], &f)) *w = f; }
  /* pId: VCutRandomWalk.scala:23-26 (a missing/unparsable pId is Random.nextInt there; -1 here) */
  *pid = -1;
  if (partitioned && np > 2) { int32_t pv; if (java_parse_int(tok[2], len[2], &pv)) *pid = pv; }
  /* ids: UniformRandomWalk.scala:34 */
  if (!java_parse_int(tok[0], len[0], src) || !java_parse_int(tok[1], len[1], dst)) {
    snprintf(err, errlen, "line %lld: NumberFormatException for vertex id", (long long)lineno); return 0;
  }
  return 1;
}

struct orc_graph {
  int64_t n_lines; int32_t *l_src, *l_dst, *l_pid; float *l_w;
  int32_t vmin, vmax; int64_t n_slots;
  int32_t *uid;                   /* sparse id space: the sorted distinct ids, slot = rank (NULL: slot = id - vmin) */
  uint8_t *present; int64_t *off; /* n_slots + 1 */
  int32_t *ids; float *w; int32_t *sorted_ids;
  int64_t n_entries, n_vertices;
  float *a_prob; int32_t *a_alias; uint8_t *a_regular; /* Mode A tables, built lazily */
};

static int cmp_i32(const void *a, const void *b) { int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }
/* index of id v in the per-vertex arrays; -1 if v cannot be a vertex (presence is checked by the callers) */
static inline int64_t raw_slot(const orc_graph *g, int32_t v) {
  if (g->n_slots == 0 || v < g->vmin || v > g->vmax) return -1;
  if (!g->uid) return (int64_t)v - g->vmin;
  int64_t lo = 0, hi = g->n_slots - 1;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (g->uid[mid] < v) lo = mid + 1; else hi = mid; }
  return g->uid[lo] == v ? lo : -1;
}
static inline int32_t slot_id(const orc_graph *g, int64_t s) { return g->uid ? g->uid[s] : (int32_t)(s + g->vmin); }

/* sorted_ids: every row's ids in ascending order (the walk's sorted membership test when faithful == 0).  Rows are independent:
 * the threads take blocks of slots from a shared cursor (full-size parity checks sort 1.8e9 entries). */
typedef struct { orc_graph *g; int64_t next; pthread_mutex_t mu; } sort_job;
static void *sort_worker(void *arg) {
  sort_job *j = (sort_job *)arg;
  for (;;) {
    pthread_mutex_lock(&j->mu);
    int64_t v0 = j->next; j->next += 4096;
    pthread_mutex_unlock(&j->mu);
    if (v0 >= j->g->n_slots) break;
    int64_t v1 = v0 + 4096 < j->g->n_slots ? v0 + 4096 : j->g->n_slots;
    for (int64_t v = v0; v < v1; ++v)
      qsort(j->g->sorted_ids + j->g->off[v], (size_t)(j->g->off[v + 1] - j->g->off[v]), sizeof(int32_t), cmp_i32);
  }
  return NULL;
}
static void sort_rows(orc_graph *g) {
  long nc = sysconf(_SC_NPROCESSORS_ONLN);
  int nt = g->n_entries < ((int64_t)1 << 22) ? 1 : (int)(nc < 1 ? 1 : nc > 64 ? 64 : nc);
  sort_job j; j.g = g; j.next = 0; pthread_mutex_init(&j.mu, NULL);
  if (nt == 1) { sort_worker(&j); pthread_mutex_destroy(&j.mu); return; }
  pthread_t th[64]; int started = 0;
  for (int t = 0; t < nt; ++t) if (pthread_create(&th[started], NULL, sort_worker, &j) == 0) ++started;
  if (!started) sort_worker(&j);
  for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
  pthread_mutex_destroy(&j.mu);
}

/* Adjacency of v = concatenation, in line order, of every line's contribution to v
 * (UniformRandomWalk.scala:35-41: flatMap then reduceByKey(_ ++ _); canonical order). */
static orc_graph *graph_build(line_vec *lv, int directed) {
  orc_graph *g = (orc_graph *)calloc(1, sizeof(*g));
  g->n_lines = lv->n; g->l_src = lv->src; g->l_dst = lv->dst; g->l_pid = lv->pid; g->l_w = lv->w;
  if (lv->n == 0) { g->vmin = 0; g->vmax = -1; g->n_slots = 0; g->off = (int64_t *)calloc(1, sizeof(int64_t)); return g; }
  int32_t vmin = lv->src[0], vmax = lv->src[0];
  for (int64_t i = 0; i < lv->n; ++i) {
    if (lv->src[i] < vmin) vmin = lv->src[i]; if (lv->src[i] > vmax) vmax = lv->src[i];
    if (lv->dst[i] < vmin) vmin = lv->dst[i]; if (lv->dst[i] > vmax) vmax = lv->dst[i];
  }
  g->vmin = vmin; g->vmax = vmax; g->n_slots = (int64_t)vmax - (int64_t)vmin + 1;
  /* The reference keys vertices in a HashMap (GraphMap.scala:13-15): any int32 ids.  This index is an array; when the id
   * space is sparse the array is over the RANK of the id among the sorted distinct ids instead of id - vmin. */
  if (g->n_slots > 64 * lv->n + ((int64_t)1 << 20)) {
    g->uid = (int32_t *)malloc(sizeof(int32_t) * (size_t)lv->n * 2);
    for (int64_t i = 0; i < lv->n; ++i) { g->uid[2 * i] = lv->src[i]; g->uid[2 * i + 1] = lv->dst[i]; }
    qsort(g->uid, (size_t)lv->n * 2, sizeof(int32_t), cmp_i32);
    int64_t m = 0;
    for (int64_t i = 0; i < 2 * lv->n; ++i) if (m == 0 || g->uid[m - 1] != g->uid[i]) g->uid[m++] = g->uid[i];
    g->n_slots = m;
  }
  g->present = (uint8_t *)calloc((size_t)g->n_slots, 1);
  g->off = (int64_t *)calloc((size_t)g->n_slots + 1, sizeof(int64_t));
  for (int64_t i = 0; i < lv->n; ++i) {
    int64_t s = raw_slot(g, lv->src[i]), d = raw_slot(g, lv->dst[i]);
    g->present[s] = 1; g->present[d] = 1;
    g->off[s + 1]++;
    if (!directed) g->off[d + 1]++;
  }
  for (int64_t v = 0; v < g->n_slots; ++v) { g->off[v + 1] += g->off[v]; g->n_vertices += g->present[v]; }
  g->n_entries = g->off[g->n_slots];
  g->ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g->n_entries ? g->n_entries : 1));
  g->w = (float *)malloc(sizeof(float) * (size_t)(g->n_entries ? g->n_entries : 1));
  int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)g->n_slots);
  memcpy(cur, g->off, sizeof(int64_t) * (size_t)g->n_slots);
  for (int64_t i = 0; i < lv->n; ++i) {
    int64_t s = raw_slot(g, lv->src[i]), d = raw_slot(g, lv->dst[i]);
    g->ids[cur[s]] = lv->dst[i]; g->w[cur[s]] = lv->w[i]; cur[s]++;          /* (src, [(dst, w)]) */
    if (!directed) { g->ids[cur[d]] = lv->src[i]; g->w[cur[d]] = lv->w[i]; cur[d]++; } /* (dst, [(src, w)]) */
  }
  free(cur);
  g->sorted_ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g->n_entries ? g->n_entries : 1));
  memcpy(g->sorted_ids, g->ids, sizeof(int32_t) * (size_t)g->n_entries);
  sort_rows(g);
  return g;
}

orc_graph *orc_graph_from_coo(const int32_t *src, const int32_t *dst, const float *w, int64_t n_lines, int directed) {
  line_vec lv; memset(&lv, 0, sizeof(lv));
  size_t n = (size_t)(n_lines > 0 ? n_lines : 1);
  lv.src = (int32_t *)malloc(sizeof(int32_t) * n); lv.dst = (int32_t *)malloc(sizeof(int32_t) * n);
  lv.pid = (int32_t *)malloc(sizeof(int32_t) * n); lv.w = (float *)malloc(sizeof(float) * n);
  lv.n = n_lines; lv.cap = (int64_t)n;
  memcpy(lv.src, src, sizeof(int32_t) * (size_t)n_lines); memcpy(lv.dst, dst, sizeof(int32_t) * (size_t)n_lines);
  for (int64_t i = 0; i < n_lines; ++i) { lv.pid[i] = -1; lv.w[i] = w ? w[i] : 1.0f; }
  return graph_build(&lv, directed);
}

orc_graph *orc_graph_load_edgelist(const char *path, int directed, int weighted, int partitioned,
                                   char *err, size_t errlen) {
  char ebuf[256]; if (!err) { err = ebuf; errlen = sizeof(ebuf); }
  err[0] = 0;
  FILE *f = fopen(path, "rb");
  if (!f) { snprintf(err, errlen, "cannot open %s: %s", path, strerror(errno)); return NULL; }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  char *buf = (char *)malloc((size_t)sz + 1);
  if (sz > 0 && fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); snprintf(err, errlen, "read error"); return NULL; }
  fclose(f);
  line_vec lv; memset(&lv, 0, sizeof(lv));
  /* Hadoop LineRecordReader: a line ends at \n, \r\n or a lone \r; a final unterminated line counts. */
  size_t pos = 0; int64_t lineno = 0;
  /* LineRecordReader.skipUtfByteOrderMark (Hadoop 2.6+, MAPREDUCE-5777): a UTF-8 BOM at the start of the file is not part of line 1 */
  if (sz >= 3 && (unsigned char)buf[0] == 0xEF && (unsigned char)buf[1] == 0xBB && (unsigned char)buf[2] == 0xBF) pos = 3;
  while (pos < (size_t)sz) {
    size_t b = pos;
    while (pos < (size_t)sz && buf[pos] != '\n' && buf[pos] != '\r') ++pos;
    size_t e = pos;
    if (pos < (size_t)sz) { if (buf[pos] == '\r' && pos + 1 < (size_t)sz && buf[pos + 1] == '\n') pos += 2; else pos += 1; }
    ++lineno;
    int32_t s, d, pid; float w;
    if (!parse_line(buf + b, e - b, weighted, partitioned, lineno, &s, &d, &w, &pid, err, errlen)) {
      free(buf); free(lv.src); free(lv.dst); free(lv.pid); free(lv.w); return NULL;
    }
    lv_push(&lv, s, d, w, pid);
  }
  free(buf);
  return graph_build(&lv, directed);
}

void orc_graph_free(orc_graph *g) {
  if (!g) return;
  free(g->l_src); free(g->l_dst); free(g->l_pid); free(g->l_w);
  free(g->uid); free(g->present); free(g->off); free(g->ids); free(g->w); free(g->sorted_ids);
  free(g->a_prob); free(g->a_alias); free(g->a_regular); free(g);
}
int64_t orc_graph_num_vertices(const orc_graph *g) { return g->n_vertices; }
int64_t orc_graph_num_entries(const orc_graph *g) { return g->n_entries; }
int64_t orc_graph_num_lines(const orc_graph *g) { return g->n_lines; }
void orc_graph_vertices(const orc_graph *g, int32_t *out) {
  int64_t k = 0;
  for (int64_t v = 0; v < g->n_slots; ++v) if (g->present[v]) out[k++] = slot_id(g, v);
}
static inline int64_t slot_of(const orc_graph *g, int32_t v) {
  int64_t s = raw_slot(g, v);
  return (s >= 0 && g->present[s]) ? s : -1;
}
int64_t orc_graph_degree(const orc_graph *g, int32_t v) {
  int64_t s = slot_of(g, v); if (s < 0) return -1; return g->off[s + 1] - g->off[s];
}
int64_t orc_graph_neighbors(const orc_graph *g, int32_t v, int32_t *ids, float *w, int64_t cap) {
  int64_t s = slot_of(g, v); if (s < 0) return -1;
  int64_t n = g->off[s + 1] - g->off[s];
  for (int64_t k = 0; k < n && k < cap; ++k) { if (ids) ids[k] = g->ids[g->off[s] + k]; if (w) w[k] = g->w[g->off[s] + k]; }
  return n;
}
void orc_graph_lines(const orc_graph *g, int32_t *src, int32_t *dst, float *w, int32_t *pid) {
  for (int64_t i = 0; i < g->n_lines; ++i) {
    if (src) src[i] = g->l_src[i]; if (dst) dst[i] = g->l_dst[i];
    if (w) w[i] = g->l_w[i]; if (pid) pid[i] = g->l_pid[i];
  }
}

/* ============================================================================================ */
/* Mode A — exact-integer alias tables + rejection (build-defined)              */
/* ============================================================================================ */
typedef unsigned __int128 u128;

/* Alias-regular: every weight finite and >= 0, some weight > 0, and ceil_log2(n) + e_max - e_min <= 29 (the
 * exactness certificate of the Mode R sum).  Then W_k = w_k * 2^(23 - e_min) are exact integers,
 * T = sum W_k < 2^53, m_k = W_k * n < 2^53, and the table is defined by integer prefix sums:
 *   lights (m_k < T) in index order, deficits d_i = T - m, D_i = prefix sums; heavies (m_k >= T), excesses
 *   e_j = m - T, E_j = prefix sums;
 *   light i : keeps m, alias = first heavy j with E_j > D_{i-1};
 *   heavy j : first light i with D_i > E_j; if D_{i-1} < E_j it keeps T - (D_i - E_j), alias = heavy j+1;
 *             otherwise (no such light, or the previous light ended exactly at E_j) it keeps T. */
int orc_alias_row(const float *w, int64_t n, float *prob, int32_t *alias) {
  if (n <= 0) return 0;
  int emin = 1 << 20, emax = -(1 << 20);
  for (int64_t k = 0; k < n; ++k) {
    uint32_t b; memcpy(&b, &w[k], 4);
    int ex = (int)((b >> 23) & 0xFF);
    if (ex == 255) return 0;                 /* NaN / Inf */
    if ((b & 0x7FFFFFFFu) == 0) continue;
    if (b >> 31) return 0;                   /* negative */
    int e = ex ? ex - 127 : -126;
    if (e < emin) emin = e; if (e > emax) emax = e;
  }
  if (emax < emin) return 0;                 /* all zero */
  int cl = 0; while (((int64_t)1 << cl) < n) ++cl;
  if (cl + emax - emin > 29) return 0;
  uint64_t *W = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
  uint64_t T = 0;
  for (int64_t k = 0; k < n; ++k) { W[k] = (uint64_t)ldexp((double)w[k], 23 - emin); T += W[k]; }
  int64_t a = 0, b = 0;
  int64_t *L = (int64_t *)malloc(sizeof(int64_t) * (size_t)n), *H = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
  u128 *D = (u128 *)malloc(sizeof(u128) * (size_t)(n + 1)), *E = (u128 *)malloc(sizeof(u128) * (size_t)(n + 1));
  D[0] = 0; E[0] = 0;
  for (int64_t k = 0; k < n; ++k) {
    uint64_t m = W[k] * (uint64_t)n;
    if (m >= T) { H[b] = k; E[b + 1] = E[b] + (u128)(m - T); ++b; }
    else { L[a] = k; D[a + 1] = D[a] + (u128)(T - m); ++a; }
  }
  int64_t j = 0;                                   /* lights: two-pointer over E */
  for (int64_t i = 0; i < a; ++i) {
    while (j < b && E[j + 1] <= D[i]) ++j;        /* first heavy with E_j > D_{i-1} */
    int64_t k = L[i];
    alias[k] = (int32_t)H[j];
    prob[k] = (float)((double)(W[k] * (uint64_t)n) / (double)T);
  }
  int64_t i = 0;                                   /* heavies: two-pointer over D */
  for (j = 0; j < b; ++j) {
    while (i < a && D[i + 1] <= E[j + 1]) ++i;    /* first light with D_i > E_j */
    int64_t k = H[j];
    if (i < a && D[i] < E[j + 1]) {                /* that light started inside this heavy's excess interval */
      uint64_t x = (uint64_t)(D[i + 1] - E[j + 1]);
      alias[k] = (int32_t)H[j + 1];
      prob[k] = (float)((double)(T - x) / (double)T);
    } else { alias[k] = (int32_t)k; prob[k] = 1.0f; }
  }
  free(W); free(L); free(H); free(D); free(E);
  return 1;
}

static void graph_build_alias(orc_graph *g) {
  if (g->a_prob) return;
  g->a_prob = (float *)malloc(sizeof(float) * (size_t)(g->n_entries ? g->n_entries : 1));
  g->a_alias = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g->n_entries ? g->n_entries : 1));
  g->a_regular = (uint8_t *)calloc((size_t)(g->n_slots ? g->n_slots : 1), 1);
  for (int64_t v = 0; v < g->n_slots; ++v) {
    int64_t n = g->off[v + 1] - g->off[v];
    if (n > 0) g->a_regular[v] = (uint8_t)orc_alias_row(g->w + g->off[v], n, g->a_prob + g->off[v], g->a_alias + g->off[v]);
  }
}

int orc_graph_alias_row(const orc_graph *g, int32_t v, float *prob, int32_t *alias) {
  int64_t s = raw_slot(g, v);
  if (s < 0 || !g->present[s]) return -1;
  graph_build_alias((orc_graph *)g);
  int64_t n = g->off[s + 1] - g->off[s];
  for (int64_t k = 0; k < n; ++k) { prob[k] = g->a_prob[g->off[s] + k]; alias[k] = g->a_alias[g->off[s] + k]; }
  return g->a_regular[s];
}

/* ============================================================================================ */
/* Walk — RandomWalk.scala:51-66 (first step), :95-139 (second-order loop)                      */
/* ============================================================================================ */

static inline float draw(const orc_walk_params *P, uint32_t iter, int32_t src, uint32_t step) {
  return P->rng_mode == ORC_RNG_CONST ? P->const_r : orc_walk_uniform(P->seed, iter, (uint32_t)src, step);
}

static int sorted_contains(const int32_t *a, int64_t n, int32_t x) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { int64_t mid = lo + ((hi - lo) >> 1); if (a[mid] < x) lo = mid + 1; else hi = mid; }
  return lo < n && a[lo] == x;
}

/* secondOrderSample on CSR rows.  faithful: literally computeSecondOrderWeights + sample on a
 * scratch array (linear exists).  fast: same arithmetic, membership by binary search. */
static int64_t second_order_pick(const orc_graph *g, const orc_walk_params *P, int32_t prev, int64_t ps,
                                 int64_t cs, float r, float *scratch) {
  const int32_t *cid = g->ids + g->off[cs]; const float *cw = g->w + g->off[cs];
  int64_t n = g->off[cs + 1] - g->off[cs];
  int64_t np = g->off[ps + 1] - g->off[ps];
  if (P->faithful) {
    orc_second_order_weights(P->p, P->q, prev, g->ids + g->off[ps], np, cid, cw, n, scratch);
  } else {
    const int32_t *sp = g->sorted_ids + g->off[ps];
    for (int64_t k = 0; k < n; ++k) {
      float w = cw[k], u = w / P->q;
      if (cid[k] == prev) u = w / P->p; else if (sorted_contains(sp, np, cid[k])) u = w;
      scratch[k] = u;
    }
  }
  return orc_sample_index(scratch, n, r);
}

/* Mode A step: trial t draws (x0..x3) = Philox(ctr = (iter, src, step, t), key = (seed, 0xA11A5));
 * slot j = ((x0:x1) * n) >> 64, coin u2 = (x2 >> 8) * 2^-24 picks j or alias[j]; a second-order step accepts
 * the candidate iff u3 * Q < bias with u3 = (x3 >> 8) * 2^-24, Q = max(1, 1/q),
 * bias = 1/p (dst == prev), 1 (dst in N(prev)), 1/q (else); when 1/p > Q the excess (1/p - Q) * w of the return
 * edge(s) is sampled through an appendix branch (second Philox call, key word 0xA11A6).  An irregular row uses the
 * reference's CDF inversion with u2 of trial 0. */
static int sorted_contains(const int32_t *a, int64_t n, int32_t x);
static int64_t second_order_pick(const orc_graph *g, const orc_walk_params *P, int32_t prev, int64_t ps,
                                 int64_t cs, float r, float *scratch);
static int64_t alias_pick(const orc_graph *g, const orc_walk_params *P, int second_order, int32_t prev, int64_t ps,
                          int64_t cs, uint32_t iter, int32_t src, uint32_t step, float *scratch) {
  int64_t n = g->off[cs + 1] - g->off[cs];
  const int32_t *cid = g->ids + g->off[cs];
  uint32_t key[2] = {P->seed, 0xA11A5u}, o[4];
  if (!g->a_regular[cs]) {
    uint32_t ctr[4] = {iter, (uint32_t)src, step, 0u};
    orc_philox4x32_10(ctr, key, o);
    float u = (float)(o[2] >> 8) * (1.0f / 16777216.0f);
    if (!second_order) return orc_sample_index(g->w + g->off[cs], n, u);
    orc_walk_params Q = *P; Q.faithful = 0;
    return second_order_pick(g, &Q, prev, ps, cs, u, scratch);
  }
  const float inv_p = 1.0f / P->p, inv_q = 1.0f / P->q;
  const float Q = inv_q > 1.0f ? inv_q : 1.0f;           /* envelope WITHOUT the return edge */
  const int biased = second_order && !(P->p == 1.0f && P->q == 1.0f);
  const float *cw = g->w + g->off[cs];
  /* outlier folding (KnightKing): when 1/p > Q the return edge(s) get an "appendix" of area (1/p - Q) * Wprev on top
   * of the envelope area Q * S; a trial first chooses appendix vs envelope by area. */
  int fold = 0; double a = 0.0, tot = 0.0, Wprev = 0.0;
  if (biased && inv_p > Q) {
    for (int64_t k2 = 0; k2 < n; ++k2) if (cid[k2] == prev) Wprev += (double)cw[k2];
    if (Wprev > 0.0) {
      double S = 0.0;
      for (int64_t k2 = 0; k2 < n; ++k2) S += (double)cw[k2];   /* exact: the row is alias-regular */
      a = ((double)inv_p - (double)Q) * Wprev; tot = (double)Q * S + a; fold = 1;
    }
  }
  int64_t k = 0;
  for (uint32_t t = 0; t < 65536u; ++t) {
    uint32_t ctr[4] = {iter, (uint32_t)src, step, t};
    orc_philox4x32_10(ctr, key, o);
    if (fold) {
      uint32_t key2[2] = {P->seed, 0xA11A6u}, y[4];
      orc_philox4x32_10(ctr, key2, y);
      float u5 = (float)(y[0] >> 8) * (1.0f / 16777216.0f);
      if ((double)u5 * tot < a) {                          /* appendix: return to prev, occurrence ~ w, input order */
        float u6 = (float)(y[1] >> 8) * (1.0f / 16777216.0f);
        double target = (double)u6 * Wprev, cum = 0.0; int64_t last = 0;
        for (int64_t k2 = 0; k2 < n; ++k2) if (cid[k2] == prev) { cum += (double)cw[k2]; last = k2; if (cum >= target) return k2; }
        return last;
      }
    }
    uint64_t r64 = ((uint64_t)o[0] << 32) | o[1];
    int64_t j = (int64_t)(((u128)r64 * (u128)(uint64_t)n) >> 64);
    float u2 = (float)(o[2] >> 8) * (1.0f / 16777216.0f);
    k = (u2 < g->a_prob[g->off[cs] + j]) ? j : g->a_alias[g->off[cs] + j];
    if (!biased) return k;
    float bias = inv_q;
    if (cid[k] == prev) bias = inv_p;
    else if (sorted_contains(g->sorted_ids + g->off[ps], g->off[ps + 1] - g->off[ps], cid[k])) bias = 1.0f;
    float u3 = (float)(o[3] >> 8) * (1.0f / 16777216.0f);
    if (u3 * Q < bias) return k;
  }
  return k;
}

static int32_t walk_one(const orc_graph *g, const orc_walk_params *P, int32_t src, uint32_t iter,
                        int32_t *path, float **scratch, int64_t *scratch_cap) {
  int32_t L2 = P->walk_length + 2;
  int32_t len = 0;
  path[len++] = src;
  int64_t cs = slot_of(g, src);
  if (cs < 0) return len;
  int64_t n = g->off[cs + 1] - g->off[cs];
  if (n == 0) return len;                                       /* dead end, RandomWalk.scala:59-62 */
  int64_t k = P->sampler == 1 ? alias_pick(g, P, 0, src, cs, cs, iter, src, 1, *scratch)
                              : orc_sample_index(g->w + g->off[cs], n, draw(P, iter, src, 1)); /* :57 */
  int32_t prev = src; int64_t ps = cs;
  int32_t curr = g->ids[g->off[cs] + k];
  path[len++] = curr;
  while (len != L2) {                                           /* :103 */
    cs = slot_of(g, curr);
    n = g->off[cs + 1] - g->off[cs];
    if (n == 0) break;                                          /* :115-120 */
    if (n > *scratch_cap) { *scratch_cap = n * 2; *scratch = (float *)realloc(*scratch, sizeof(float) * (size_t)*scratch_cap); }
    k = P->sampler == 1 ? alias_pick(g, P, 1, prev, ps, cs, iter, src, (uint32_t)len, *scratch)
                        : second_order_pick(g, P, prev, ps, cs, draw(P, iter, src, (uint32_t)len), *scratch); /* :112-113 */
    prev = curr; ps = cs;
    curr = g->ids[g->off[cs] + k];
    path[len++] = curr;                                         /* :114 */
  }
  return len;
}

int32_t orc_seq_walk(const orc_graph *g, int32_t src, int32_t iter, const orc_walk_params *P, int32_t *out_path) {
  /* T/UniformRandomWalkTest.scala:293-321 (doSecondOrderRandomWalk): first-order step, then
   * walkLength second-order steps with getNeighbors(prev) re-read each time. */
  orc_walk_params Q = *P; Q.faithful = 1;
  float *scratch = NULL; int64_t cap = 0;
  int32_t len = 0;
  out_path[len++] = src;
  int64_t n = orc_graph_degree(g, src);
  if (n <= 0) return len;
  {
    int64_t s = slot_of(g, src);
    int64_t k = orc_sample_index(g->w + g->off[s], n, draw(&Q, (uint32_t)iter, src, 1));
    out_path[len++] = g->ids[g->off[s] + k];
  }
  for (int32_t t = 0; t < P->walk_length; ++t) {
    int32_t curr = out_path[len - 1], prev = out_path[len - 2];
    int64_t cs = slot_of(g, curr), ps = slot_of(g, prev);
    int64_t cn = g->off[cs + 1] - g->off[cs];
    if (cn <= 0) break;
    if (cn > cap) { cap = cn * 2; scratch = (float *)realloc(scratch, sizeof(float) * (size_t)cap); }
    int64_t k = second_order_pick(g, &Q, prev, ps, cs, draw(&Q, (uint32_t)iter, src, (uint32_t)len), scratch);
    out_path[len++] = g->ids[g->off[cs] + k];
  }
  free(scratch);
  return len;
}

typedef struct {
  const orc_graph *g; const orc_walk_params *P; const int32_t *starts; int64_t n_starts;
  int32_t *paths, *lens; int tid, nthreads; int64_t steps;
} walk_job;

static void *walk_thread(void *arg) {
  walk_job *J = (walk_job *)arg;
  const orc_walk_params *P = J->P;
  int32_t L2 = P->walk_length + 2;
  float *scratch = NULL; int64_t cap = 0; int64_t steps = 0;
  int64_t total = (int64_t)P->num_walks * J->n_starts;
  for (int64_t wi = J->tid; wi < total; wi += J->nthreads) {
    int64_t it = wi / J->n_starts, vi = wi % J->n_starts;
    int32_t *path = J->paths + wi * L2;
    for (int32_t t = 0; t < L2; ++t) path[t] = -1;
    int32_t len = walk_one(J->g, P, J->starts[vi], (uint32_t)(P->first_walk + it), path, &scratch, &cap);
    J->lens[wi] = len; steps += len - 1;
  }
  free(scratch);
  J->steps = steps;
  return NULL;
}

int64_t orc_walk(const orc_graph *g, const orc_walk_params *P, const int32_t *sources, int64_t n_sources,
                 int32_t *paths, int32_t *lens) {
  int32_t *all = NULL;
  if (P->sampler == 1) graph_build_alias((orc_graph *)g);
  if (!sources) {
    n_sources = g->n_vertices;
    all = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_sources ? n_sources : 1));
    orc_graph_vertices(g, all); sources = all;
  }
  int nt = P->threads > 0 ? P->threads : 1;
  if (nt > 256) nt = 256;
  walk_job jobs[256]; pthread_t th[256];
  for (int t = 0; t < nt; ++t) {
    jobs[t] = (walk_job){g, P, sources, n_sources, paths, lens, t, nt, 0};
    if (nt > 1) pthread_create(&th[t], NULL, walk_thread, &jobs[t]); else walk_thread(&jobs[t]);
  }
  int64_t steps = 0;
  for (int t = 0; t < nt; ++t) { if (nt > 1) pthread_join(th[t], NULL); steps += jobs[t].steps; }
  free(all);
  return steps;
}

/* ============================================================================================ */
/* Writer — RandomWalk.scala:234-241 (path.mkString("\t"), saveAsTextFile(output/path))         */
/* ============================================================================================ */

int orc_write_paths(const int32_t *paths, const int32_t *lens, int64_t n_walkers, int64_t stride,
                    const char *output_dir, int n_parts) {
  char dir[4096], fn[4200];
  mkdir(output_dir, 0777);
  snprintf(dir, sizeof(dir), "%s/path", output_dir);
  if (mkdir(dir, 0777) != 0) return -1;   /* FileAlreadyExistsException in the reference */
  if (n_parts < 1) n_parts = 1;
  int64_t per = (n_walkers + n_parts - 1) / n_parts;
  for (int part = 0; part < n_parts; ++part) {
    snprintf(fn, sizeof(fn), "%s/part-%05d", dir, part);
    FILE *f = fopen(fn, "wb"); if (!f) return -2;
    int64_t b = part * per, e = b + per; if (e > n_walkers) e = n_walkers;
    for (int64_t wI = b; wI < e; ++wI) {
      for (int32_t t = 0; t < lens[wI]; ++t) fprintf(f, t ? "\t%d" : "%d", paths[wI * stride + t]);
      fputc('\n', f);
    }
    fclose(f);
  }
  snprintf(fn, sizeof(fn), "%s/_SUCCESS", dir);
  FILE *f = fopen(fn, "wb"); if (!f) return -2; fclose(f);
  return 0;
}

/* ============================================================================================ */
/* Synthetic RMAT (a,b,c,d) = (.57,.19,.19,.05) — build-defined, shared with the HIP generator  */
/* ============================================================================================ */

void orc_rmat_edges(int scale, uint32_t seed, int64_t first, int64_t count, int32_t *src, int32_t *dst) {
  const uint32_t T1 = 2448131358u, T2 = 3264175144u, T3 = 4080218930u; /* floor(.57,.76,.95 * 2^32) */
  for (int64_t i = 0; i < count; ++i) {
    uint64_t e = (uint64_t)(first + i);
    uint32_t s = 0, d = 0, out[4] = {0, 0, 0, 0};
    for (int l = 0; l < scale; ++l) {
      if ((l & 3) == 0) {
        uint32_t ctr[4] = {(uint32_t)e, (uint32_t)(e >> 32), (uint32_t)(l >> 2), 0x524D4154u};
        uint32_t key[2] = {seed, 1u};
        orc_philox4x32_10(ctr, key, out);
      }
      uint32_t x = out[l & 3];
      uint32_t rb = (x >= T2), cb = (x >= T1 && x < T2) || (x >= T3);
      s = (s << 1) | rb; d = (d << 1) | cb;
    }
    src[i] = (int32_t)s; dst[i] = (int32_t)d;
  }
}

float orc_rmat_weight(int32_t u, int32_t v, uint32_t seed) {
  uint32_t a = (uint32_t)(u < v ? u : v), b = (uint32_t)(u < v ? v : u);
  uint32_t x = b * 0x85EBCA77u;
  uint32_t h = seed ^ (a * 0x9E3779B1u) ^ ((x << 13) | (x >> 19));
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return (float)(1u + (h & 15u));
}

/* ============================================================================================ */
/* The embedding stage (`--cmd node2vec` / `--cmd embedding`) — PARITY UNPINNED                  */
/* ============================================================================================ */
/* M/Main.scala:36-44,113-124 hand the paths to org.apache.spark.mllib.feature.Word2Vec (Spark 2.2, pom.xml:125-135), which is NOT
 * in /root/reference and seeds itself from the clock.  What is restated here is its published algorithm — skip-gram with
 * hierarchical softmax as in word2vec.c, which MLlib ports: vocabulary by descending count, Huffman codes (<= 40 bits), the
 * 1 000-entry sigmoid table over [-6, 6), a window shrunk by a random b per position, the learning rate decaying linearly to 1e-4 of
 * its start, syn0 uniform in (-0.5, 0.5) / dim, syn1 zero — with the build's own seeded draws (w2v_hash) and ONE logical partition,
 * sentence after sentence.  It checks the GPU trainer's sequential mode (threads == 1) within a float tolerance. */
static uint32_t w2v_hash(uint32_t seed, uint32_t a, uint32_t b, uint32_t c) {
  uint32_t h = seed ^ 0x9E3779B9u;
  h ^= a + 0x7F4A7C15u + (h << 6) + (h >> 2); h *= 0x85EBCA6Bu; h ^= h >> 13;
  h ^= b + 0x165667B1u + (h << 6) + (h >> 2); h *= 0xC2B2AE35u; h ^= h >> 16;
  h ^= c + 0x27D4EB2Fu + (h << 6) + (h >> 2); h *= 0x9E3779B1u; h ^= h >> 15;
  return h;
}
typedef struct { int32_t id; int64_t cn; } w2v_word;
static int w2v_cmp_id(const void *a, const void *b) { int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }
static int w2v_cmp_word(const void *a, const void *b) {       /* count descending, id ascending */
  const w2v_word *x = (const w2v_word *)a, *y = (const w2v_word *)b;
  if (x->cn != y->cn) return x->cn > y->cn ? -1 : 1;
  return (x->id > y->id) - (x->id < y->id);
}
/* --- the three pieces the known-answer tests pin (tests/test_w2v_known_answers.py), used by orc_w2v_fit below ------------------ */
/* word2vec.c CreateBinaryTree: counts[V] in descending order -> codelen[V], code[V][40] (root first), point[V][41] (rows of syn1 on
 * the path: point[.][0] = V - 2 = the root, then the inner nodes below it; entries beyond codelen are unused) */
static void w2v_build_tree(const int64_t *cn, int64_t V, int32_t *codelen, uint8_t *code, int32_t *point) {
  int64_t *count = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * V + 1));
  int32_t *parent = (int32_t *)calloc((size_t)(2 * V + 1), sizeof(int32_t));
  uint8_t *binary = (uint8_t *)calloc((size_t)(2 * V + 1), 1);
  for (int64_t a = 0; a < V; ++a) count[a] = cn[a];
  for (int64_t a = V; a < 2 * V + 1; ++a) count[a] = (int64_t)1e15;
  int64_t pos1 = V - 1, pos2 = V;
  for (int64_t a = 0; a < V - 1; ++a) {
    int64_t min1, min2;
    if (pos1 >= 0) { if (count[pos1] < count[pos2]) { min1 = pos1; pos1--; } else { min1 = pos2; pos2++; } } else { min1 = pos2; pos2++; }
    if (pos1 >= 0) { if (count[pos1] < count[pos2]) { min2 = pos1; pos1--; } else { min2 = pos2; pos2++; } } else { min2 = pos2; pos2++; }
    count[V + a] = count[min1] + count[min2];
    parent[min1] = (int32_t)(V + a); parent[min2] = (int32_t)(V + a);
    binary[min2] = 1;
  }
  for (int64_t a = 0; a < V; ++a) {
    uint8_t c[41]; int32_t p[41]; int i = 0; int64_t b = a;
    for (;;) { c[i] = binary[b]; p[i] = (int32_t)b; i++; b = parent[b]; if (b == 2 * V - 2 || i >= 40) break; }
    codelen[a] = i;
    point[a * 41] = (int32_t)(V - 2);
    for (int k = 0; k < i; ++k) { code[a * 40 + i - k - 1] = c[k]; point[a * 41 + i - k] = p[k] - (int32_t)V; }
  }
  free(count); free(parent); free(binary);
}
/* word2vec.c's expTable: sigmoid at 1 000 points of [-6, 6) */
static void w2v_exp_table(float *t) {
  for (int i = 0; i < 1000; ++i) { float e = (float)exp(((double)i / 1000 * 2.0 - 1.0) * 6.0); t[i] = e / (e + 1.0f); }
}
/* one (centre word, context word) pair of skip-gram + hierarchical softmax: r0 = syn0[context]; for every node d of the centre word's
 * path (rows[d] = its row of syn1, code[d] its bit): f = r0 . r1; if |f| < 6: g = (1 - code - expTable[f]) * alpha; neu += g * r1 (the
 * OLD r1); r1 += g * r0.  Then r0 += neu.  neu: dim floats of scratch. */
static void w2v_pair(int32_t dim, float *r0, float *const *rows, const uint8_t *code, int32_t n_nodes, float alpha, const float *exp_table, float *neu) {
  for (int32_t j = 0; j < dim; ++j) neu[j] = 0.0f;
  for (int32_t d = 0; d < n_nodes; ++d) {
    float *r1 = rows[d];
    float f = 0.0f;
    for (int32_t j = 0; j < dim; ++j) f += r0[j] * r1[j];
    if (f > -6.0f && f < 6.0f) {
      const int ind = (int)((f + 6.0f) * (1000.0f / 6.0f / 2.0f));
      const float g = (1.0f - (float)code[d] - exp_table[ind]) * alpha;
      for (int32_t j = 0; j < dim; ++j) neu[j] += g * r1[j];
      for (int32_t j = 0; j < dim; ++j) r1[j] += g * r0[j];
    }
  }
  for (int32_t j = 0; j < dim; ++j) r0[j] += neu[j];
}
/* the same three, exported for the known-answer tests */
void orc_w2v_huffman(const int64_t *counts, int64_t V, int32_t *codelen, uint8_t *codes40, int32_t *points40) {
  if (V < 2) { for (int64_t a = 0; a < V; ++a) codelen[a] = 0; return; }
  int32_t *point = (int32_t *)malloc(sizeof(int32_t) * (size_t)V * 41);
  w2v_build_tree(counts, V, codelen, codes40, point);
  for (int64_t a = 0; a < V; ++a)
    for (int k = 0; k < 40; ++k) points40[a * 40 + k] = k < codelen[a] ? point[a * 41 + k] : -1;
  free(point);
}
void orc_w2v_exp_table(float *out1000) { w2v_exp_table(out1000); }
void orc_w2v_pair_update(int32_t dim, float *syn0_row, float *syn1_rows /* [n_nodes][dim] */, const uint8_t *code, int32_t n_nodes, float alpha) {
  float exp_table[1000];
  w2v_exp_table(exp_table);
  float **rows = (float **)malloc(sizeof(float *) * (size_t)(n_nodes ? n_nodes : 1));
  for (int32_t d = 0; d < n_nodes; ++d) rows[d] = syn1_rows + (size_t)d * dim;
  float *neu = (float *)malloc(sizeof(float) * (size_t)dim);
  w2v_pair(dim, syn0_row, rows, code, n_nodes, alpha, exp_table, neu);
  free(neu); free(rows);
}

/* paths [n][stride] / lens -> *n_vocab, vocab_ids (caller frees), vectors [n_vocab][dim] (caller frees). 0 on success. */
int orc_w2v_fit(const int32_t *paths, const int32_t *lens, int64_t n, int64_t stride, int32_t dim, int32_t window, int32_t iterations,
                float lr, uint32_t seed, int32_t **vocab_ids_out, float **vectors_out, int64_t *n_vocab_out) {
  int64_t total = 0;
  for (int64_t w = 0; w < n; ++w) total += lens[w];
  int32_t *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total ? total : 1));
  { int64_t k = 0; for (int64_t w = 0; w < n; ++w) for (int32_t j = 0; j < lens[w]; ++j) all[k++] = paths[w * stride + j]; }
  int32_t *sorted = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total ? total : 1));
  memcpy(sorted, all, sizeof(int32_t) * (size_t)total);
  qsort(sorted, (size_t)total, sizeof(int32_t), w2v_cmp_id);
  int64_t V = 0;
  w2v_word *voc = (w2v_word *)malloc(sizeof(w2v_word) * (size_t)(total ? total : 1));
  for (int64_t i = 0; i < total; ++i) {
    if (V && voc[V - 1].id == sorted[i]) voc[V - 1].cn++; else { voc[V].id = sorted[i]; voc[V].cn = 1; ++V; }
  }
  free(sorted);
  qsort(voc, (size_t)V, sizeof(w2v_word), w2v_cmp_word);
  /* id -> vocabulary index (binary search over a copy sorted by id) */
  w2v_word *byid = (w2v_word *)malloc(sizeof(w2v_word) * (size_t)(V ? V : 1));
  for (int64_t r = 0; r < V; ++r) { byid[r].id = voc[r].id; byid[r].cn = r; }
  qsort(byid, (size_t)V, sizeof(w2v_word), w2v_cmp_id);    /* (id is the first field) */
  int32_t *sent = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total ? total : 1));
  for (int64_t i = 0; i < total; ++i) {
    int64_t lo = 0, hi = V - 1;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (byid[mid].id < all[i]) lo = mid + 1; else hi = mid; }
    sent[i] = (int32_t)byid[lo].cn;
  }
  free(byid); free(all);
  float *syn0 = (float *)malloc(sizeof(float) * (size_t)(V ? V : 1) * (size_t)dim);
  float *syn1 = (float *)calloc((size_t)(V ? V : 1) * (size_t)dim, sizeof(float));
  for (int64_t r = 0; r < V; ++r)
    for (int32_t j = 0; j < dim; ++j)
      syn0[r * dim + j] = ((float)(w2v_hash(seed, 0xA11CEu, (uint32_t)r, (uint32_t)j) >> 8) * (1.0f / 16777216.0f) - 0.5f) / (float)dim;
  int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)(V ? V : 1));
  for (int64_t r = 0; r < V; ++r) ids[r] = voc[r].id;
  if (V >= 2 && total > 0 && iterations > 0) {
    int64_t *cn = (int64_t *)malloc(sizeof(int64_t) * (size_t)V);
    for (int64_t a = 0; a < V; ++a) cn[a] = voc[a].cn;
    int32_t *codelen = (int32_t *)malloc(sizeof(int32_t) * (size_t)V);
    uint8_t *code = (uint8_t *)malloc((size_t)V * 40);
    int32_t *point = (int32_t *)malloc(sizeof(int32_t) * (size_t)V * 41);
    w2v_build_tree(cn, V, codelen, code, point);
    free(cn);
    float exp_table[1000];
    w2v_exp_table(exp_table);
    float *neu = (float *)malloc(sizeof(float) * (size_t)dim);
    float *rows[40];
    for (int32_t k = 0; k < iterations; ++k) {
      int64_t off = 0;
      for (int64_t s = 0; s < n; ++s) {
        const int32_t len = lens[s];
        double a_ = (double)lr * (1.0 - ((double)k * (double)total + (double)off) / ((double)iterations * (double)total + 1.0));
        if (a_ < (double)lr * 0.0001) a_ = (double)lr * 0.0001;
        const float alpha = (float)a_;
        for (int32_t pos = 0; pos < len; ++pos) {
          const int32_t word = sent[off + pos];
          const int32_t b = (int32_t)(w2v_hash(seed, (uint32_t)k, (uint32_t)s, (uint32_t)pos) % (uint32_t)window);
          for (int32_t d = 0; d < codelen[word]; ++d) rows[d] = syn1 + (int64_t)point[word * 41 + d] * dim;
          for (int32_t a = b; a < window * 2 + 1 - b; ++a) {
            if (a == window) continue;
            const int32_t c = pos - window + a;
            if (c < 0 || c >= len) continue;
            w2v_pair(dim, syn0 + (int64_t)sent[off + c] * dim, rows, code + (size_t)word * 40, codelen[word], alpha, exp_table, neu);
          }
        }
        off += len;
      }
    }
    free(neu); free(codelen); free(code); free(point);
  }
  free(sent); free(syn1); free(voc);
  *vocab_ids_out = ids; *vectors_out = syn0; *n_vocab_out = V;
  return 0;
}
void orc_free(void *p) { free(p); }
