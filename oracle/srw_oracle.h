/*
 * srw_oracle.h — CPU restatement of data61/stellar-random-walk's `--cmd randomwalk` path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The shipped library (libstellar_rw.so) never links,
 * loads or calls anything in oracle/.
 *
 * Parity status: PINNED against the reference's own known-answer tests
 *   (RandomSampleTest.scala:9-94, GraphMapTest.scala:7-33, UniformRandomWalkTest.scala:33-86,181-321;
 *    fixtures karate.txt / testgraph.txt) — see tests/test_oracle_reference_vectors.py.
 * The reference itself (Scala/Spark) cannot be built or run here (no JVM), so there is no oracle/_ref.
 *
 * Paths in comments are relative to /root/reference/randomwalk/src/main/scala/au/csiro/data61/randomwalk/
 * (M/) and .../src/test/scala/au/csiro/data61/randomwalk/algorithm/ (T/).
 */
#ifndef SRW_ORACLE_H
#define SRW_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- RNG ------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Random123).  Not in the reference (it has no seedable RNG, RandomSample.scala:5);
 * the keyed stream below is the build's definition of "fixed seed" (SURVEY §0-2, Appendix A). */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* u = (x0 >> 8) * 2^-24, key = (seed, 0), ctr = (walk iteration, source id, step index, 0). */
float orc_walk_uniform(uint32_t seed, uint32_t iter, uint32_t src, uint32_t step);
/* java.util.Random.nextFloat stream (what scala.util.Random.nextFloat delegates to); for KATs only. */
void orc_java_random_floats(int64_t seed, int n, float *out);

/* ---- sampler: M/algorithm/RandomSample.scala ------------------------------------------------ */
/* RandomSample.sample (:12-25): returns the INDEX of the chosen edge (0 = edges.head fallback);
 * -1 when n == 0 (the reference never calls sample on an empty list). */
int64_t orc_sample_index(const float *w, int64_t n, float r);
/* RandomSample.computeSecondOrderWeights (:27-44), linear `exists` scan exactly as written. */
void orc_second_order_weights(float p, float q, int32_t prev_id, const int32_t *prev_ids, int64_t n_prev,
                              const int32_t *curr_ids, const float *curr_w, int64_t n, float *out_w);
/* RandomSample.secondOrderSample (:55-62). */
int64_t orc_second_order_sample_index(float p, float q, int32_t prev_id, const int32_t *prev_ids,
                                      int64_t n_prev, const int32_t *curr_ids, const float *curr_w,
                                      int64_t n, float r);

/* ---- GraphMap: M/algorithm/GraphMap.scala ---------------------------------------------------- */
typedef struct orc_graphmap orc_graphmap;
orc_graphmap *orc_graphmap_new(void);
void orc_graphmap_free(orc_graphmap *);
void orc_graphmap_reset(orc_graphmap *);                                   /* :98-107 */
/* addVertex(vId, Array[(Int,Float)]) (:41-56); n == 0 -> addVertex(vId) (:83-85). First add wins. */
void orc_graphmap_add_vertex(orc_graphmap *, int32_t v, const int32_t *ids, const float *w, int64_t n);
/* addVertex(vId, Array[(Int,Int,Float)]) (:23-39): also records dst -> pId. */
void orc_graphmap_add_vertex_p(orc_graphmap *, int32_t v, const int32_t *ids, const int32_t *pids,
                               const float *w, int64_t n);
int64_t orc_graphmap_num_vertices(const orc_graphmap *);                   /* :87-89 */
int64_t orc_graphmap_num_edges(const orc_graphmap *);                      /* :91-93 */
/* getNeighbors (:109-120): returns -1 for `null` (unknown vertex), 0 for the empty array, else the
 * length; copies at most cap entries into ids/w when they are non-NULL. */
int64_t orc_graphmap_get_neighbors(const orc_graphmap *, int32_t v, int32_t *ids, float *w, int64_t cap);
/* getPartition (:66-68): 1 and *pid set when known, else 0. */
int orc_graphmap_get_partition(const orc_graphmap *, int32_t v, int32_t *pid);

/* ---- graph load: M/algorithm/UniformRandomWalk.scala:23-43, VCutRandomWalk.scala:19-54 ------- */
typedef struct orc_graph orc_graph;
/* Parse an edge-list text file with the reference's rules (Java split("\\s+"), Integer.parseInt,
 * Float.parseFloat, Try(...).getOrElse(1.0f)).  partitioned != 0 selects the VCut column rules.
 * Returns NULL and fills err on the inputs that make the reference job throw. */
orc_graph *orc_graph_load_edgelist(const char *path, int directed, int weighted, int partitioned,
                                   char *err, size_t errlen);
/* Same construction from already-parsed lines (src,dst,w per line, file order). */
orc_graph *orc_graph_from_coo(const int32_t *src, const int32_t *dst, const float *w, int64_t n_lines,
                              int directed);
void orc_graph_free(orc_graph *);
int64_t orc_graph_num_vertices(const orc_graph *);   /* ids that occur in >= 1 line */
int64_t orc_graph_num_entries(const orc_graph *);    /* directed adjacency entries ("edges: N") */
int64_t orc_graph_num_lines(const orc_graph *);
void orc_graph_vertices(const orc_graph *, int32_t *out); /* ascending id */
int64_t orc_graph_degree(const orc_graph *, int32_t v);   /* -1 = absent */
int64_t orc_graph_neighbors(const orc_graph *, int32_t v, int32_t *ids, float *w, int64_t cap);
/* parsed lines as the loader saw them (for feeding the product's srw_load_coo in tests) */
void orc_graph_lines(const orc_graph *, int32_t *src, int32_t *dst, float *w, int32_t *pid);

/* ---- walk: M/algorithm/RandomWalk.scala:51-66,75-176; T/UniformRandomWalkTest.scala:293-321 --- */
enum { ORC_RNG_CONST = 0, ORC_RNG_PHILOX = 1 };
typedef struct {
  float p, q;          /* already cast .toFloat (RandomWalk.scala:112) */
  int32_t walk_length; /* --walkLength */
  int32_t num_walks;   /* --numWalks */
  int32_t rng_mode;    /* ORC_RNG_* */
  float const_r;       /* nextFloat = () => const_r (T/UniformRandomWalkTest.scala:183-185) */
  uint32_t seed;       /* philox key */
  int32_t first_walk;  /* first walk-iteration index (for the philox counter) */
  int32_t faithful;    /* 1: linear `exists` membership as in the reference; 0: sorted membership */
  int32_t threads;     /* >= 1 */
  int32_t sampler;     /* 0: Mode R (reference CDF inversion); 1: Mode A (alias + rejection, build-defined) */
} orc_walk_params;
/* Whole job: for each walk iteration, one walker per present vertex (ascending id).  paths is
 * [num_walks * nVertices][walk_length + 2] (unused tail = -1), lens per walker.  If `sources` is
 * non-NULL only those n_sources vertices start walkers (bounded CPU-baseline samples).  Returns the
 * number of steps (sum of len-1). */
int64_t orc_walk(const orc_graph *, const orc_walk_params *, const int32_t *sources, int64_t n_sources,
                 int32_t *paths, int32_t *lens);
/* Restatement of the reference's own test oracle doSecondOrderRandomWalk for a single source. */
int32_t orc_seq_walk(const orc_graph *, int32_t src, int32_t iter, const orc_walk_params *, int32_t *out_path);

/* ---- Mode A (build-defined; NOT in the reference, which has no alias tables — SURVEY §0-1) ------------------
 * Exact-integer alias table of one neighbor list (spec in DESIGN.md §4.6): returns 1 and fills prob/alias
 * (alias = position inside the list) when the list is alias-regular, else 0 (Mode A then samples that vertex
 * with the reference's CDF inversion). */
int orc_alias_row(const float *w, int64_t n, float *prob, int32_t *alias);
/* alias table of vertex v of a graph (same construction), -1 if v is absent */
int orc_graph_alias_row(const orc_graph *, int32_t v, float *prob, int32_t *alias);

/* ---- writer: M/algorithm/RandomWalk.scala:234-241, M/common/Property.scala:6 ------------------ */
/* Writes <output>/path/part-%05d + _SUCCESS.  Fails (-1) if <output>/path exists. */
int orc_write_paths(const int32_t *paths, const int32_t *lens, int64_t n_walkers, int64_t stride,
                    const char *output_dir, int n_parts);

/* ---- the embedding stage (`--cmd node2vec`; MLlib Word2Vec is absent from the reference tree: PARITY UNPINNED) ------------------
 * Sequential skip-gram + hierarchical softmax with the build's seeded draws; see srw_oracle.c.  Outputs malloc'ed: orc_free. */
void orc_w2v_huffman(const int64_t *counts, int64_t V, int32_t *codelen, uint8_t *codes40, int32_t *points40);
void orc_w2v_exp_table(float *out1000);
void orc_w2v_pair_update(int32_t dim, float *syn0_row, float *syn1_rows, const uint8_t *code, int32_t n_nodes, float alpha);
int orc_w2v_fit(const int32_t *paths, const int32_t *lens, int64_t n, int64_t stride, int32_t dim, int32_t window, int32_t iterations,
                float lr, uint32_t seed, int32_t **vocab_ids_out, float **vectors_out, int64_t *n_vocab_out);
void orc_free(void *p);

/* ---- synthetic input (build-defined, shared with the HIP generator) ---------- */
void orc_rmat_edges(int scale, uint32_t seed, int64_t first, int64_t count, int32_t *src, int32_t *dst);
float orc_rmat_weight(int32_t u, int32_t v, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif
