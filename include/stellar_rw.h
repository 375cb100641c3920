/*
 * stellar_rw.h — C ABI of libstellar_rw.so, the MI355X-native (gfx950) engine behind the
 * `--cmd randomwalk` path of data61/stellar-random-walk.
 *
 * The reference has no FFI of its own (it is 100 % Scala on Spark); its seams for this path are the
 * Scala-level interfaces cited next to each entry point below.  A JNI shim
 * (Java_au_csiro_data61_randomwalk_algorithm_HipRandomWalk_*) binds exactly these functions — see
 * INTEGRATION.md.  Paths: M/ = randomwalk/src/main/scala/au/csiro/data61/randomwalk/.
 *
 * Conventions: plain C, no exceptions cross the boundary; every function returns an srw status
 * (0 = ok) and srw_last_error() gives the message.  The opaque handle owns all device memory; the
 * caller owns every buffer it passes in.  A handle is single-threaded; distinct handles may be used
 * from distinct threads.  There is NO CPU fallback: every compute entry point runs HIP kernels on the
 * handle's device and fails with SRW_ERR_HIP when no gfx950 device is usable.
 */
#ifndef STELLAR_RW_H
#define STELLAR_RW_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct srw_handle srw_handle;

enum {
  SRW_OK = 0,
  SRW_ERR_INVALID = 1, /* bad argument / bad state */
  SRW_ERR_IO = 2,      /* file cannot be opened / written */
  SRW_ERR_PARSE = 3,   /* an input line on which the reference job throws (NumberFormatException ...) */
  SRW_ERR_HIP = 4,     /* HIP runtime error, or no usable GPU */
  SRW_ERR_EXISTS = 5,  /* <output>/path already exists (FileAlreadyExistsException in the reference) */
  SRW_ERR_NOMEM = 6
};

/* sampler selection */
enum {
  SRW_SAMPLER_REFERENCE = 0, /* Mode R: reference-exact linear-CDF inversion (bit-identical paths) */
  SRW_SAMPLER_ALIAS = 1      /* Mode A: per-vertex alias tables + rejection (statistically identical) */
};
/* RNG selection.  The reference's only determinism hook is the injected nextFloat closure
 * (M/algorithm/RandomSample.scala:5, M/algorithm/RandomWalk.scala:51-52,75-76). */
enum {
  SRW_RNG_CONST = 0,  /* nextFloat = () => const_r, as the reference's tests inject */
  SRW_RNG_PHILOX = 1  /* Philox4x32-10, key (seed,0), ctr (walk iteration, source id, step index, 0),
                         u = (x0 >> 8) * 2^-24 (same 24-bit lattice as java.util.Random.nextFloat) */
};

typedef struct {
  int32_t device; /* HIP device ordinal */
  int32_t rank;   /* vertex sharding: this handle stores the rows of the vertices v with owner(v) == rank,   */
  int32_t world;  /*   owner(v) = mix32(v) mod world — the role of HashPartitioner (RandomWalk.scala:16), with */
                  /*   the ids mixed first so that skewed id spaces (RMAT: hubs at round ids) stay balanced;  */
                  /*   world = 1 keeps the whole graph (single-GPU and replicated modes)                       */
  int32_t flags;  /* SRW_CFG_* */
} srw_config;
enum {
  /* sharded handles (world > 1): a vertex is owned by the partition id recorded for it by a partitioned load
   * (VCutRandomWalk: GraphMap.getPartition(steps.last), M/algorithm/VCutRandomWalk.scala:121-134), modulo world;
   * vertices without a recorded partition fall back to mix32(v) mod world. */
  SRW_CFG_OWNER_FROM_PARTITIONS = 1,
  /* whole-graph handles (world == 1): always compact the vertex ids at load (slot = rank among the sorted distinct ids).
   * Without the flag this happens only when the id space is sparse — the reference's HashMap-keyed GraphMap
   * (M/algorithm/GraphMap.scala:13-15) takes any int32 ids, e.g. "-2147483648 2147483647".  Results are the same
   * either way (the Philox stream stays keyed by the input's ids); the flag exists for tests. */
  SRW_CFG_COMPACT_IDS = 2,
  /* sharded handles (world > 1): do NOT build the replicated membership structure (the sorted neighbor ids of the WHOLE
   * graph on every shard: 4 B per adjacency entry + 16 B per id slot, and the largest sort of the load).  It is what a
   * shard needs to evaluate "x in N(prev)" for a prev it does not own, i.e. only when q != 1
   * (M/algorithm/RandomSample.scala:37); walks with q == 1 (config 4: p = q = 1) never read it.  A walk with q != 1 on
   * such a handle fails with SRW_ERR_INVALID.  With the flag a shard's memory is proportional to 1/world. */
  SRW_CFG_NO_MEMBERSHIP = 4,
  /* sharded handles (world > 1): owner(v) = nonNegativeMod(v, world) — org.apache.spark.HashPartitioner over as many partitions as
   * there are GPUs, the partition map of the reference (RandomWalk.scala:16, UniformRandomWalk.scala:42: partitionBy(new
   * HashPartitioner(rddPartitions)) on Int keys, whose hashCode is the value) — instead of the default mix32(v) mod world.
   * Results are the same under any owner function (tests); the default mixes the ids first because the low bits of an unpermuted
   * RMAT id correlate with the degree (one shard of eight gets 17 % of the walkers).  Ignored when the ids are compacted or a VCut
   * partition table is in force. */
  SRW_CFG_OWNER_HASH_PARTITIONER = 8
};

/* Replaces: SparkContext + GraphMap singleton lifetime (M/Main.scala:21-23, M/algorithm/GraphMap.scala:11). */
int32_t srw_create(const srw_config *cfg, srw_handle **out);
void srw_destroy(srw_handle *h);
/* Message of the last failure on h (h may be NULL for srw_create failures). */
const char *srw_last_error(const srw_handle *h);
/* Launch all kernels on this hipStream_t (default: the handle's own stream). */
int32_t srw_set_stream(srw_handle *h, void *hip_stream);

/* The job's --numWalks (M/common/Params.scala:7-23, default 10): how many walk iterations will run over the sampling
 * tables that the next srw_walk* call builds for its (p, q) — a caller that walks the job in several calls (one iteration
 * per call, batches) announces the total here.  It only steers what is worth BUILDING (the finer per-edge tables of the
 * biased walk cost ~8 s at config 3 and save ~0.1 s per iteration: they are built from 64 planned iterations on); results
 * never depend on it.  0 = unknown: the reference's default (10), or the call's own num_walks if that is larger. */
int32_t srw_plan_walks(srw_handle *h, int64_t num_walks);

/* ---- graph load ---------------------------------------------------------------------------- */
/* Replaces UniformRandomWalk.loadGraph (M/algorithm/UniformRandomWalk.scala:17-88) and, with
 * partitioned != 0, VCutRandomWalk.loadGraph (M/algorithm/VCutRandomWalk.scala:13-98): parses the
 * edge-list text with the reference's token rules, builds the adjacency in HBM as CSR with every
 * neighbor list in input-line order, multi-edges and self-loops kept.  Files of two integer columns (plus, with
 * weighted, a short decimal weight column) are tokenized on the GPU; every other shape and every malformed file
 * goes through the host tokenizer, with identical results and errors (SRW_HOST_TOKENIZER=1 forces the host). */
int32_t srw_load_edgelist(srw_handle *h, const char *path, int32_t directed, int32_t weighted,
                          int32_t partitioned, int32_t rdd_partitions);
/* Same construction from already-parsed lines in file order (host pointers).  w may be NULL (1.0f),
 * pid may be NULL.  Replaces the flatMap/reduceByKey stage, UniformRandomWalk.scala:26-41. */
int32_t srw_load_coo(srw_handle *h, const int32_t *src, const int32_t *dst, const float *w,
                     const int32_t *pid, int64_t n_lines, int32_t directed);
/* Complete adjacency rows, the GraphMap.addVertex surface (M/algorithm/GraphMap.scala:23-56,83-85):
 * row i is vertex vids[i] with neighbors [offs[i], offs[i+1]); first occurrence of a vertex wins. */
int32_t srw_load_adjacency(srw_handle *h, const int32_t *vids, const int64_t *offs, int64_t n_rows,
                           const int32_t *ids, const float *w, const int32_t *pids);
/* Synthetic RMAT (a,b,c,d)=(.57,.19,.19,.05) generated on the device straight into COO -> CSR
 * (BASELINE.md §4).  Edge i is a pure function of (seed, i), so any rank can generate any slice. */
int32_t srw_generate_rmat(srw_handle *h, int32_t scale, int64_t n_edges, uint32_t seed, int32_t weighted,
                          int32_t directed);

/* nVertices / nEdges as the reference prints them ("vertices: N", "edges: N",
 * UniformRandomWalk.scala:69-72); int64 because the reference's Int counters overflow at C4. */
int32_t srw_graph_stats(const srw_handle *h, int64_t *n_vertices, int64_t *n_entries);
/* Present vertex ids, ascending (the walker seeds, UniformRandomWalk.scala:81-87). out[n_vertices]. */
int32_t srw_graph_vertices(const srw_handle *h, int32_t *out);
/* GraphMap.getNeighbors (M/algorithm/GraphMap.scala:109-120): *n = -1 for `null` (unknown vertex),
 * 0 for a vertex without out-edges, else the list length; copies at most cap entries. */
int32_t srw_graph_neighbors(const srw_handle *h, int32_t v, int32_t *ids, float *w, int64_t cap, int64_t *n);
/* GraphMap.getPartition (:66-68): *pid = last partition id recorded for dst v, *known = 0 if none. */
int32_t srw_graph_partition(const srw_handle *h, int32_t v, int32_t *pid, int32_t *known);

/* ---- walk ---------------------------------------------------------------------------------- */
typedef struct {
  float p, q;          /* --p / --q already cast to float (RandomWalk.scala:112) */
  int32_t walk_length; /* --walkLength: a full path has walk_length + 2 ids */
  int32_t num_walks;   /* --numWalks walk iterations in this call */
  int32_t first_walk;  /* index of the first walk iteration (Philox counter word 0) */
  int32_t rng_mode;    /* SRW_RNG_* */
  float const_r;
  uint32_t seed;
  int32_t sampler;     /* SRW_SAMPLER_* */
  int32_t flags;       /* SRW_WALK_* */
} srw_walk_params;
enum {
  SRW_WALK_FORCE_GENERAL = 1, /* use the general second-order kernel even when p == q == 1 (testing) */
  SRW_WALK_NT_LOADS = 2,      /* first-order kernel: force L1-bypassing (nontemporal) record loads */
  SRW_WALK_CACHED_LOADS = 4,  /* first-order kernel: force default-policy loads (default: chosen from the table size) */
  SRW_WALK_NO_COMPACT = 16,   /* first-order kernel: do not use the 16-byte lattice records even when available */
  SRW_WALK_NO_PREFIX = 32,    /* general kernel: always stream N(curr); do not use the exact prefix-sum search */
  SRW_WALK_NO_EDGE_HASH = 64, /* Mode A: membership by binary search in the sorted rows instead of the edge hash set */
  SRW_WALK_NO_BINNED = 128,   /* Mode R, q != 1: never use the binned prefix-sum search (streaming scan instead) */
  SRW_WALK_DEVICE_FORMAT = 131072, /* srw_walk_and_save: the GPU formats the path text (path_format.hip); the host only writes it */
  SRW_WALK_NO_HUB_BITMAPS = 65536, /* Mode R, q != 1: do not build / use the hub rows' neighbor-set bitmaps */
  SRW_WALK_NO_EDGE_TABLES = 262144, /* Mode R, q != 1: do not build / use the per-edge bias tables */
  SRW_WALK_EDGE_TABLES_ALL = 524288 /* test switch: a table for EVERY certified pair, chunks of 4 candidates */
  /* bits 12-14: test switch, force the binned search's membership strategy (1 = P1, 2 = P2, 3 = id-window,
     4 = hub bitmap where there is one); bit 15: test switch, binned search and hub bitmaps on rows of any degree
     (defaults: degree >= 128, hubs of degree >= 1024) */
};

typedef struct {
  int64_t n_walkers;    /* num_walks * nVertices */
  int64_t n_steps;      /* sum over paths of (len - 1): the metric's unit */
  int64_t dead_ends;    /* "Zero Neighbors" accumulator, RandomWalk.scala:117 */
  int64_t sum_deg_curr; /* general kernel: sum of deg(curr) over steps (algorithmic bytes) */
  int64_t sum_deg_prev; /* general kernel: sum of deg(prev) over second-order steps with q != 1 */
  int64_t ent_reads;    /* first-order / alias kernels: table records read; general kernel: steps served by the prefix search */
  int64_t fallbacks;    /* steps that needed the exact sequential fallback */
  int64_t trials;       /* Mode A: alias draws (accepted + rejected) */
  double kernel_ms;     /* hipEvent time of the walk kernels of this call, on the handle's stream */
  int32_t kernel_kind;  /* 1 = first-order guide-table kernel, 2 = general second-order kernel, 3 = alias */
  int32_t record_bytes; /* bytes of one sampling-table record read by this kernel (16 compact / 32 / 0) */
  int64_t strategy_steps[12]; /* general kernel: steps served by each sampler, indexed by SRW_STRAT_* */
  int64_t edge_tables;       /* per-edge bias tables in use by this call (0: none built) */
  int64_t edge_table_bytes;  /* HBM held by them */
  double setup_ms;           /* host wall time this call spent building sampling tables before the first kernel */
} srw_walk_stats;
enum { /* srw_walk_stats.strategy_steps: which sampler of the general (second-order) kernel served a step */
  SRW_STRAT_EDGE_TABLE = 0, /* precomputed per-edge bias table: search + one chunk */
  SRW_STRAT_P1 = 1,         /* binned search, N(prev) looked up in the sorted N(curr) */
  SRW_STRAT_P2 = 2,         /* binned search, candidates probed in the edge hash set / sorted N(prev) */
  SRW_STRAT_W = 3,          /* binned search, sorted-chunk intersection */
  SRW_STRAT_P3 = 4,         /* binned search, hub neighbor-set bitmap */
  SRW_STRAT_SCAN = 5,       /* certified streaming scan (small rows, rows without a certificate, first steps) */
  SRW_STRAT_PREFIX = 6,     /* q == 1: prefix-sum search with the return edges as a short list */
  SRW_STRAT_CHAIN = 7,      /* the reference's sequential f64 chain (irregular rows, draws on a CDF boundary) */
  SRW_STRAT_EDGE_MASK = 8,  /* precomputed per-edge membership mask (rows below 256 candidates): no lookup at all */
  SRW_STRAT_Q1_LANE = 9,    /* p != 1, q == 1: one walker per lane (first-order guide table + exact prefix sums + return edge) */
  SRW_STAT_HANDED_OVER = 10, /* not a sampler: WALKERS the per-edge-table / per-lane kernels handed to the general kernel (redone there) */
  SRW_STAT_TIES_RESOLVED = 11 /* not a sampler: of those, the walkers whose boundary draw the chain kernels resolved ahead of the redo (srw_walk) */
};

/* Replaces RandomWalk.randomWalk (M/algorithm/RandomWalk.scala:75-176) incl. initFirstStep (:51-66):
 * num_walks iterations, one walker per present vertex (ascending id), walker index =
 * iteration * nVertices + rank(vertex).  Paths stay in HBM as int32 [n_walkers][walk_length + 2]
 * (unused tail = -1) plus int32 lens[n_walkers]. */
int32_t srw_walk(srw_handle *h, const srw_walk_params *params, srw_walk_stats *stats);
/* The whole randomWalk loop with the results streamed to HOST buffers: iteration i's kernel runs while iteration
 * i-1's paths travel over PCIe on a second HIP stream (two device staging buffers).  paths is
 * [num_walks * nVertices][walk_length + 2], lens [num_walks * nVertices]; allocate them with srw_host_alloc
 * (pinned) for full PCIe rate — pageable buffers work but serialise.  Replaces RandomWalk.randomWalk returning
 * the union of all iterations' paths (M/algorithm/RandomWalk.scala:168,175). */
int32_t srw_walk_to_host(srw_handle *h, const srw_walk_params *params, int32_t *paths, int32_t *lens,
                         srw_walk_stats *stats);
/* Main.doRandomWalk fused (M/Main.scala:53-62: rw.execute() then rw.save(...)): walks num_walks iterations and
 * writes <output_dir>/path/part-* + _SUCCESS while streaming — iteration i's kernel and PCIe transfer overlap the host's
 * formatting of iteration i-1 (pinned ring of two slices); the paths are never held as a whole in host memory.
 * With SRW_WALK_DEVICE_FORMAT in params->flags the GPU formats the text and the host only copies it out and writes it.
 * dead_ends_per_iteration (optional, [num_walks]) receives the reference's per-iteration "Zero Neighbors" count.
 * Fails with SRW_ERR_EXISTS before any work if <output_dir>/path exists. */
int32_t srw_walk_and_save(srw_handle *h, const srw_walk_params *params, const char *output_dir, int32_t n_parts,
                          int32_t write_crc, srw_walk_stats *stats, int64_t *dead_ends_per_iteration);
/* Pinned (page-locked) host memory for srw_walk_to_host / srw_fetch_paths destinations. */
int32_t srw_host_alloc(size_t bytes, void **out);
void srw_host_free(void *p);
/* Copy the last walk's paths / lens to host buffers (either may be NULL). */
int32_t srw_fetch_paths(const srw_handle *h, int32_t *paths, int32_t *lens);
/* Device pointers of the last walk's result (valid until the next srw_walk / destroy). */
int32_t srw_device_paths(const srw_handle *h, void **d_paths, void **d_lens, int64_t *n_walkers, int32_t *stride);
/* Replaces RandomWalk.save (M/algorithm/RandomWalk.scala:234-241) + Property.pathSuffix: writes
 * <output_dir>/path/part-00000.. (TAB-joined ids, one '\n' per path) and _SUCCESS; canonical line
 * order (walk iteration major, source id ascending).  write_crc != 0 adds Hadoop .crc side files.
 * The text is formatted on the GPU from the device-resident result (SRW_HOST_FORMATTER=1: on host threads). */
int32_t srw_write_paths(const srw_handle *h, const char *output_dir, int32_t n_parts, int32_t write_crc);

/* Mode A table of vertex v (build-defined exact-integer alias construction): prob/alias of each
 * neighbor slot (alias = position inside the row); *regular = 0 when the row is not alias-regular (Mode A
 * then samples it by CDF inversion).  *n as srw_graph_neighbors. */
int32_t srw_alias_row(srw_handle *h, int32_t v, float *prob, int32_t *alias, int64_t cap, int64_t *n, int32_t *regular);

/* ---- vertex-sharded multi-GPU path (one handle per GPU, world > 1) -------------------------- */
/* Replaces the Spark shuffle of the super-step loop: prepareWalkersToTransfer (UniformRandomWalk.scala:103-112,
 * VCutRandomWalk.scala:121-134) + transferWalkersToTheirPartitions (RandomWalk.scala:186-192) + the loop itself
 * (RandomWalk.scala:91-162).  The graph is sharded by source vertex (srw_config.rank / world); a walker standing on v is
 * processed by owner(v); its PATH stays on its home rank = owner(source).  Each super-step every rank consumes one
 * receive buffer of `world` fixed-capacity chunks (one per sender) and fills `world` destination chunks:
 *     chunk = { uint32 n_walkers, n_rets, 0, 0 } | srw_walker[cap_walkers] | srw_path_ret[cap_rets]
 * — 16 + 8 = 24 bytes per walker-step on the wire (the reference ships the path so far and N(prev), RandomWalk.scala:135).
 * The caller moves chunk (me -> d) to rank d's receive buffer slot `me`: one equal-split all-to-all of chunk_bytes per
 * peer (RCCL over xGMI; stellar-random-walk_amd/distributed.py), or — one process, several devices — the kernels store
 * straight into the peers' receive buffers (srw_cluster_*, below).  No host synchronisation per super-step: counts live
 * in the chunk headers; an overflowing chunk drops its surplus and srw_shard_finish reports it (retry with more slack).
 * The keyed RNG makes the paths bit-identical for any world size (tests assert it against the oracle). */
typedef struct {            /* 16 bytes on the wire */
  int32_t lw;               /* home rank's path row: local vertex index * batch + iteration in batch */
  int32_t src;              /* source vertex (Philox key; owner(src) = home rank) */
  int32_t prev, curr;       /* the walker stands on curr, came from prev (linked p = q = 1 walk: prev | curr << 32 = row link of the vertex it stands on) */
} srw_walker;
typedef struct {            /* 8 bytes: one path slot of walker lw, returned to its home rank; the slot is implicit — a return
                               produced by super-step s is slot s */
  int32_t lw;               /* top bit set: death notice (the walker stopped before sampling slot s: its path has s entries) */
  int32_t v;
} srw_path_ret;
typedef struct { int64_t cap_walkers, cap_rets, chunk_bytes; } srw_shard_layout;
/* vertices owned by this handle / present in the whole graph (the walker seeds, UniformRandomWalk.scala:81-87) */
int32_t srw_shard_capacity(const srw_handle *h, int64_t *n_local_vertices, int64_t *n_global_vertices);
/* global rank (position among all present vertices, ascending id) of each local vertex: path row lw of a batch of B
 * iterations is canonical walker (first_walk + lw % B) * nVertices + ranks[lw / B].  out[n_local_vertices] (host). */
int32_t srw_shard_vertex_ranks(const srw_handle *h, int32_t *out);
/* Chunk capacities for `batch` walk iterations sharing their super-steps: slack * batch * nVertices / world^2 + 4096. */
int32_t srw_shard_layout_for(const srw_handle *h, int32_t batch, double slack, srw_shard_layout *out);
/* Two walker populations per handle: population 0 (the default) and 1 each have their own super-step context (scratch, cursors,
 * counters, path staging) and their own stream; srw_shard_begin / _superstep / _flush / _finish act on the selected one.  A driver
 * that interleaves the super-steps of two populations lets one's kernels run while the other's chunks are being exchanged — the
 * overlap the reference's shuffle / count rhythm (RandomWalk.scala:91-162) does not have.  Select 0 again before any other call
 * on the handle: while population 1 is selected, srw_walk*, the loads, srw_w2v_fit* and srw_probe_request_rate fail with
 * SRW_ERR_INVALID (they would run on the second population's stream with its cursors); srw_set_stream sets the SELECTED
 * population's stream. */
int32_t srw_shard_select(srw_handle *h, int32_t population);
/* Seeds this rank's batch * n_local walkers into its own receive buffer d_recv (world * chunk_bytes, device), writes
 * path slot 0 / lens into d_paths [batch * n_local][walk_length + 2] / d_lens (device), clears the counters. */
int32_t srw_shard_begin(srw_handle *h, const srw_walk_params *params, int32_t batch, const srw_shard_layout *layout,
                        void *d_recv, void *d_paths, void *d_lens);
/* Super-step `step` (1 .. walk_length + 1), enqueued on the handle's stream: applies the path returns found in d_recv,
 * samples every walker in d_recv once, writes chunk (me -> d) to dst_chunks[d] (device or peer pointers, d < world). */
int32_t srw_shard_superstep(srw_handle *h, const srw_walk_params *params, int32_t batch, int32_t step,
                            const srw_shard_layout *layout, const void *d_recv, void *const *dst_chunks, void *d_paths,
                            void *d_lens);
/* After the exchange that follows super-step walk_length + 1: applies its path returns. */
int32_t srw_shard_flush(srw_handle *h, const srw_walk_params *params, int32_t batch, const srw_shard_layout *layout,
                        const void *d_recv, void *d_paths, void *d_lens);
/* Synchronises the stream; steps / dead ends since srw_shard_begin; *overflow != 0: a chunk was too small. */
int32_t srw_shard_finish(srw_handle *h, srw_walk_stats *stats, int32_t *overflow);
/* Device memory on the handle's GPU for callers without an allocator of their own (exchange buffers of the sharded
 * walk: one hipMalloc per buffer — RCCL faulted on multi-GB buffers that were sub-blocks of a caching allocator's
 * segment, profiles/r02i).  srw_device_free(h, NULL) is a no-op. */
int32_t srw_device_alloc(srw_handle *h, int64_t bytes, void **d_ptr);
int32_t srw_device_free(srw_handle *h, void *d_ptr);
/* Optional, once per loaded graph, before the first p = q = 1 walk: row descriptors across shards.  The replicated
 * first-order records carry the row descriptor of the neighbor they name, so a step never reads the row table; on a
 * shard that row lives on the neighbor's OWNER.  Each shard exports its row table (16 bytes per dense id slot:
 * int64 offset, int32 degree, uint32 flags; zeros for vertices it does not own), the tables are combined by an
 * element-wise maximum of their int64 words (an all-reduce MAX; inside one process srw_shard_rows_merge over peer
 * pointers), and srw_shard_rows_commit derives 16-byte records whose links point into the owners' tables.  The walkers
 * of a p = q = 1 Philox walk then travel with the link of the vertex they stand on (srw_walker.prev / kind hold it),
 * and sampling + bucketing run as one kernel.  EVERY shard must end up linked (*linked == 1) — if one does not (a
 * record needing an escape), call srw_shard_rows_release on all of them: the unlinked path keeps working. */
int32_t srw_shard_rows_count(const srw_handle *h, int64_t *n_slots);
int32_t srw_shard_rows_export(srw_handle *h, void *d_rows, int64_t n_slots);
int32_t srw_shard_rows_merge(srw_handle *h, void *d_rows, const void *d_other_rows, int64_t n_slots);
int32_t srw_shard_rows_commit(srw_handle *h, const void *d_rows_all, int64_t n_slots, int32_t *linked);
int32_t srw_shard_rows_release(srw_handle *h);

/* ---- the same walk inside ONE process over several devices (CLI --gpus N, the JNI host) ------------------------------ */
/* One sharded handle per device, peer access enabled; the bucket kernels store every chunk directly into the receiving
 * device's buffer over xGMI and the super-steps are ordered by events — no collective library, no host sync per step.
 * devices may repeat an ordinal (several shards on one GPU: how the protocol is tested on a single-GPU box). */
typedef struct srw_cluster srw_cluster;
int32_t srw_cluster_create(const int32_t *devices, int32_t n_devices, int32_t flags, srw_cluster **out);
void srw_cluster_destroy(srw_cluster *c);
const char *srw_cluster_last_error(const srw_cluster *c);
srw_handle *srw_cluster_shard(srw_cluster *c, int32_t rank);   /* the rank's handle (graph queries, tests) */
int32_t srw_cluster_load_edgelist(srw_cluster *c, const char *path, int32_t directed, int32_t weighted, int32_t partitioned,
                                  int32_t rdd_partitions);
int32_t srw_cluster_load_coo(srw_cluster *c, const int32_t *src, const int32_t *dst, const float *w, const int32_t *pid,
                             int64_t n_lines, int32_t directed);
int32_t srw_cluster_generate_rmat(srw_cluster *c, int32_t scale, int64_t n_edges, uint32_t seed, int32_t weighted, int32_t directed);
int32_t srw_cluster_graph_stats(const srw_cluster *c, int64_t *n_vertices, int64_t *n_entries);
/* params->num_walks iterations starting at first_walk, `batch` of them per population (0 = automatic); paths stay on
 * their home devices.  stats: n_steps / dead_ends summed over the shards, kernel_ms = wall time of the super-steps. */
int32_t srw_cluster_walk(srw_cluster *c, const srw_walk_params *params, int32_t batch, srw_walk_stats *stats);
/* The last walk in canonical order (iteration major, source id ascending): paths [num_walks * nVertices][walk_length + 2]. */
int32_t srw_cluster_fetch_paths(srw_cluster *c, int32_t *paths, int32_t *lens);
/* walk + RandomWalk.save (RandomWalk.scala:234-241) over the cluster: <output_dir>/path/part-* + _SUCCESS.  The job is streamed
 * batch by batch (device memory: one batch of paths per shard, host memory: one slice), so NO walk result stays behind:
 * srw_cluster_fetch_paths after this call fails with "no walk result" — use srw_cluster_walk when the paths are wanted in memory. */
int32_t srw_cluster_walk_and_save(srw_cluster *c, const srw_walk_params *params, const char *output_dir, int32_t n_parts,
                                  int32_t write_crc, srw_walk_stats *stats);

/* ---- the embedding stage (`--cmd node2vec` / `--cmd embedding`) ---------------------------------------
 * Replaces Main.configureWord2Vec + Word2Vec.fit + the vector part of saveModelAndFeatures (M/Main.scala:36-44,77-97,113-124):
 * skip-gram with hierarchical softmax over the paths, trained on the GPU (csrc/embedding.hip).  org.apache.spark.mllib.feature.
 * Word2Vec is a dependency that is absent from the reference tree and seeds itself from the clock: PARITY UNPINNED — the build
 * restates the published algorithm with a seed of its own and is checked against its CPU restatement (oracle: orc_w2v_fit). */
typedef struct {
  int32_t dim;            /* --dim     (setVectorSize)    */
  int32_t window;         /* --window  (setWindowSize)    */
  int32_t iterations;     /* --iter    (setNumIterations) */
  float learning_rate;    /* --lr      (setLearningRate)  */
  uint32_t seed;          /* the build's seed (MLlib: Utils.random.nextLong(), unseeded) */
  int32_t threads;        /* 1: one wave, sentences in order (the sequential form); anything else: one wave per sentence, Hogwild */
} srw_w2v_params;
/* paths [n][stride] / lens on the HOST (what srw_fetch_paths returns, or a parsed paths file).  Outputs are malloc'ed (srw_free):
 * vocab_ids[n_vocab] = the ids by descending count (ties: ascending id; minCount 0), vectors[n_vocab * dim]. */
int32_t srw_w2v_fit(srw_handle *h, const int32_t *paths, const int32_t *lens, int64_t n, int64_t stride, const srw_w2v_params *params,
                    int32_t **vocab_ids, float **vectors, int64_t *n_vocab);
/* The same with the paths where srw_walk left them: in HBM (d_paths [n][stride], d_lens [n]: what srw_device_paths returns; both NULL =
 * this handle's last walk result).  This is the reference's hand-over — `randomWalk(...)` feeds Word2Vec.fit without leaving the
 * cluster (M/Main.scala:113-117): tokens are flattened, sorted, run-length encoded and ranked on the device; only the vocabulary's counts
 * (for the Huffman tree) and the trained vectors cross PCIe. */
int32_t srw_w2v_fit_device(srw_handle *h, const void *d_paths, const void *d_lens, int64_t n, int64_t stride, const srw_w2v_params *params,
                           int32_t **vocab_ids, float **vectors, int64_t *n_vocab);
/* Unit-test hook (host only, no GPU): word2vec.c's CreateBinaryTree as the trainer uses it.  counts[n_vocab] in descending order ->
 * code_len[n_vocab], codes[n_vocab][40] (bits, root first), points[n_vocab][40] (rows of syn1 on the path, root = n_vocab - 2 first;
 * -1 beyond the code).  Known answers: tests/test_w2v_known_answers.py. */
int32_t srw_w2v_huffman(const int64_t *counts, int64_t n_vocab, int32_t *code_len, uint8_t *codes, int32_t *points);
/* "<id>\t<v0>\t...\t<v_dim-1>" lines (Main.scala:88-91; floats printed as java.lang.Float.toString prints them) into
 * <output_dir>/vec/part-%05d + _SUCCESS, and the model directory <output_dir>/bin as Word2VecModel.save lays it out: metadata/part-00000
 * (one JSON line) + data/part-00000.parquet — Parquet with Spark's schema for (word: String, vector: Array[Float]), written by
 * csrc/parquet_model.cpp (uncompressed; Spark writes snappy pages, any Parquet reader takes both).  As saveModelAndFeatures: the model first, then the vectors;
 * fails before writing anything if <output_dir>/bin or <output_dir>/vec exists, and removes what it created if it fails midway. */
int32_t srw_w2v_save(const int32_t *vocab_ids, const float *vectors, int64_t n_vocab, int32_t dim, const char *output_dir, int32_t n_parts);
/* The same for a vocabulary of WORDS (`--cmd embedding` on a text whose tokens are not vertex ids: the reference's Word2Vec takes any
 * token, M/Main.scala:119-124): words[r] is the NUL-terminated word of vocabulary row r. */
int32_t srw_w2v_save_words(const char *const *words, const float *vectors, int64_t n_vocab, int32_t dim, const char *output_dir, int32_t n_parts);

/* ---- measurement hooks (bench.py's `roofline` object; not on the walk's path) ------------------------ */
/* The ceiling the walk kernels are held against, measured on this handle's GPU in about a second: dependent, uniformly random
 * 16-byte reads (one chain per lane, 81 hops, the access pattern of a first-order step: one record per step) over a table of
 * `table_bytes` (0: 32 GiB; clipped to the free HBM).  *reads_per_s: reads = L2-miss requests per second; *table_gib (may be
 * NULL): the size actually used.  No reference counterpart: the reference has no device. */
int32_t srw_probe_request_rate(srw_handle *h, int64_t table_bytes, double *reads_per_s, double *table_gib);
/* What the REFERENCE's algorithm reads for the walk this handle just did (srw_walk; paths still resident), counted from the
 * finished paths: sums[0] = sum over the steps of deg(curr) (RandomSample.sample scans N(curr), RandomSample.scala:12-25),
 * sums[1] = sum over the second-order steps of deg(prev) (the `exists` over N(prev), RandomSample.scala:27-44), sums[2] = steps.
 * SURVEY §8(d)'s algorithmic bytes are 20 * sums[2] + 8 * sums[0] (+ 16 * second-order steps + 4 * sums[1] when q != 1). */
int32_t srw_result_scan_sums(srw_handle *h, int64_t *sums);

/* ---- unit hooks (device arithmetic of RandomSample, for parity tests) ----------------------- */
/* RandomSample.sample (M/algorithm/RandomSample.scala:12-25) on the GPU; *index = chosen position. */
int32_t srw_sample(srw_handle *h, const float *w, int64_t n, float r, int64_t *index);
/* RandomSample.computeSecondOrderWeights (:27-44) on the GPU. */
int32_t srw_second_order_weights(srw_handle *h, float p, float q, int32_t prev_id, const int32_t *prev_ids,
                                 int64_t n_prev, const int32_t *curr_ids, const float *curr_w, int64_t n,
                                 float *out_w);
/* RandomSample.secondOrderSample (:55-62) on the GPU. */
int32_t srw_second_order_sample(srw_handle *h, float p, float q, int32_t prev_id, const int32_t *prev_ids,
                                int64_t n_prev, const int32_t *curr_ids, const float *curr_w, int64_t n,
                                float r, int64_t *index);
/* The walk RNG stream evaluated on the GPU: out[i] = u(seed, iter[i], src[i], step[i]). */
int32_t srw_rng_uniform(srw_handle *h, uint32_t seed, const uint32_t *iter, const uint32_t *src,
                        const uint32_t *step, int64_t n, float *out);

/* ---- host-only helpers (no GPU needed) ------------------------------------------------------ */
/* The edge-list tokenizer on its own: parses `path` into caller-visible arrays owned by the library
 * (released by srw_free).  Used by the CLI and by tests of the parse rules. */
int32_t srw_parse_edgelist(const char *path, int32_t weighted, int32_t partitioned, int32_t **src,
                           int32_t **dst, float **w, int32_t **pid, int64_t *n_lines, char *err, size_t errlen);
void srw_free(void *p);
/* RandomWalk.save on host-resident paths (M/algorithm/RandomWalk.scala:234-241): same files as
 * srw_write_paths, for callers that assembled several walk calls themselves. */
int32_t srw_save_paths(const int32_t *paths, const int32_t *lens, int64_t n_walkers, int64_t stride,
                       const char *output_dir, int32_t n_parts, int32_t write_crc);
/* The chunk geometry of the per-edge bias table of a pair (prev -> curr) — what the planner, the table builder and the walk kernels
 * all compute from (deg(curr), deg(prev)) and the 10 policy words {min_shift, max_chunks, mask_max_deg, mask_min_prev_deg,
 * fine_min_prev_deg, fine_shift, fine_max_chunks, f32, u16, mask_ratio} (csrc/sampling.h:eb_pair_geometry).  Chunks of
 * 2^*chunk_shift candidates, *n_chunks of them, *masked = the pair carries chunk masks.  No reference counterpart (the reference
 * recomputes the biased weights of N(curr) at every step, RandomSample.scala:27-44); exported for the CPU test of the closed form. */
int32_t srw_table_geometry(int32_t deg_curr, int32_t deg_prev, const int32_t *policy, int32_t *chunk_shift, int32_t *n_chunks,
                           int32_t *masked);
/* Library / build identification ("stellar_rw gfx950 <git describe>"). */
const char *srw_version(void);

#ifdef __cplusplus
}
#endif
#endif
