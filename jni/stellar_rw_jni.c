/*
 * stellar_rw_jni.c — JNI shim between the Scala host class au.csiro.data61.randomwalk.algorithm.HipRandomWalk
 * (jni/HipRandomWalk.scala) and the C ABI of libstellar_rw.so (include/stellar_rw.h).
 *
 * Replaces, for `--cmd randomwalk`, the JVM-side implementations behind Main.doRandomWalk
 * (randomwalk/src/main/scala/au/csiro/data61/randomwalk/Main.scala:53-62): UniformRandomWalk / VCutRandomWalk
 * .loadGraph, RandomWalk.randomWalk (algorithm/RandomWalk.scala:75-176) and RandomWalk.save (:234-241).
 *
 * Build (needs a JDK; this image has none, so the file is only syntax-checked here against tests/jni_stub/jni.h):
 *   make -C jni JAVA_HOME=/path/to/jdk
 * Every C status other than SRW_OK becomes a Java exception whose class mirrors what the reference job would throw
 * (INTEGRATION.md §5); no exception ever crosses the boundary in the other direction.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "stellar_rw.h"

#define FN(name) Java_au_csiro_data61_randomwalk_algorithm_HipRandomWalk_##name

static srw_handle *H(jlong h) { return (srw_handle *)(intptr_t)h; }

/* status -> exception class of the reference's failure mode */
static void throw_status(JNIEnv *env, int32_t rc, const srw_handle *h) {
  const char *cls = "java/lang/RuntimeException";
  switch (rc) {
    case SRW_ERR_PARSE:   cls = "java/lang/NumberFormatException"; break;                         /* parts(0).toInt */
    case SRW_ERR_EXISTS:  cls = "org/apache/hadoop/mapred/FileAlreadyExistsException"; break;     /* saveAsTextFile */
    case SRW_ERR_IO:      cls = "java/io/IOException"; break;
    case SRW_ERR_INVALID: cls = "java/lang/IllegalArgumentException"; break;
    case SRW_ERR_NOMEM:   cls = "java/lang/OutOfMemoryError"; break;
    default: break;                                                                               /* SRW_ERR_HIP */
  }
  jclass c = (*env)->FindClass(env, cls);
  if (!c) { (*env)->ExceptionClear(env); c = (*env)->FindClass(env, "java/lang/RuntimeException"); }
  if (c) (*env)->ThrowNew(env, c, srw_last_error(h));
}

static void fill_params(srw_walk_params *P, jfloat p, jfloat q, jint walkLength, jint numWalks, jint firstWalk,
                        jboolean useConst, jfloat constR, jint seed, jint flags) {
  memset(P, 0, sizeof *P);
  P->p = p; P->q = q; P->walk_length = walkLength; P->num_walks = numWalks; P->first_walk = firstWalk;
  P->rng_mode = useConst ? SRW_RNG_CONST : SRW_RNG_PHILOX; P->const_r = constR; P->seed = (uint32_t)seed;
  P->sampler = SRW_SAMPLER_REFERENCE; P->flags = flags;
}

JNIEXPORT jlong JNICALL FN(create)(JNIEnv *env, jobject self, jint device) {
  (void)self;
  srw_config c; memset(&c, 0, sizeof c);
  c.device = device; c.rank = 0; c.world = 1;
  srw_handle *h = NULL;
  const int32_t rc = srw_create(&c, &h);
  if (rc != SRW_OK) { throw_status(env, rc, NULL); return 0; }
  return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL FN(destroy)(JNIEnv *env, jobject self, jlong h) {
  (void)env; (void)self;
  srw_destroy(H(h));
}

/* loadGraph(): returns (nVertices, nEdges) as the reference prints them (UniformRandomWalk.scala:69-72) */
JNIEXPORT jlongArray JNICALL FN(loadEdgeList)(JNIEnv *env, jobject self, jlong hh, jstring path, jboolean directed,
                                              jboolean weighted, jboolean partitioned, jint rddPartitions) {
  (void)self;
  const char *p = (*env)->GetStringUTFChars(env, path, NULL);
  if (!p) return NULL;
  const int32_t rc = srw_load_edgelist(H(hh), p, directed, weighted, partitioned, rddPartitions);
  (*env)->ReleaseStringUTFChars(env, path, p);
  if (rc != SRW_OK) { throw_status(env, rc, H(hh)); return NULL; }
  int64_t nv = 0, ne = 0;
  srw_graph_stats(H(hh), &nv, &ne);
  jlong out[2]; out[0] = (jlong)nv; out[1] = (jlong)ne;
  jlongArray a = (*env)->NewLongArray(env, 2);
  if (a) (*env)->SetLongArrayRegion(env, a, 0, 2, out);
  return a;
}

/* one walk iteration kept in HBM (RandomWalk.randomWalk's body for iteration `iteration`); returns the walk-steps */
JNIEXPORT jlong JNICALL FN(walk)(JNIEnv *env, jobject self, jlong hh, jfloat p, jfloat q, jint walkLength,
                                 jint iteration, jfloat constR, jboolean useConst, jint seed) {
  (void)self;
  srw_walk_params P; fill_params(&P, p, q, walkLength, 1, iteration, useConst, constR, seed, 0);
  srw_walk_stats st;
  const int32_t rc = srw_walk(H(hh), &P, &st);
  if (rc != SRW_OK) { throw_status(env, rc, H(hh)); return 0; }
  return (jlong)st.n_steps;
}

/* all numWalks iterations in one call, paths kept in HBM (what `--cmd node2vec` hands to the embedding stage); returns the walk-steps */
JNIEXPORT jlong JNICALL FN(walkAll)(JNIEnv *env, jobject self, jlong hh, jfloat p, jfloat q, jint walkLength,
                                    jint numWalks, jfloat constR, jboolean useConst, jint seed) {
  (void)self;
  srw_walk_params P; fill_params(&P, p, q, walkLength, numWalks, 0, useConst, constR, seed, 0);
  srw_walk_stats st;
  const int32_t rc = srw_walk(H(hh), &P, &st);
  if (rc != SRW_OK) { throw_status(env, rc, H(hh)); return 0; }
  return (jlong)st.n_steps;
}

/* the job's --numWalks for the table planners (srw_plan_walks): callers that walk one iteration per call announce the total */
JNIEXPORT void JNICALL FN(planWalks)(JNIEnv *env, jobject self, jlong hh, jlong numWalks) {
  (void)self;
  const int32_t rc = srw_plan_walks(H(hh), (int64_t)numWalks);
  if (rc != SRW_OK) throw_status(env, rc, H(hh));
}

/* Main.configureWord2Vec + Word2Vec.fit + saveModelAndFeatures (Main.scala:36-44,77-97) on the paths of the last walk WHERE THEY ARE
 * (HBM: srw_w2v_fit_device), then <output>/bin and <output>/vec; returns the vocabulary size.  threads = 1: the deterministic form. */
JNIEXPORT jlong JNICALL FN(w2vFitAndSave)(JNIEnv *env, jobject self, jlong hh, jint dim, jint window, jint iterations, jfloat lr,
                                          jint seed, jint threads, jstring out, jint parts) {
  (void)self;
  srw_w2v_params wp; memset(&wp, 0, sizeof wp);
  wp.dim = dim; wp.window = window; wp.iterations = iterations; wp.learning_rate = lr; wp.seed = (uint32_t)seed; wp.threads = threads;
  int32_t *vocab = NULL; float *vec = NULL; int64_t nv = 0;
  int32_t rc = srw_w2v_fit_device(H(hh), NULL, NULL, 0, 1, &wp, &vocab, &vec, &nv);
  if (rc != SRW_OK) { throw_status(env, rc, H(hh)); return 0; }
  const char *o = (*env)->GetStringUTFChars(env, out, NULL);
  if (!o) { srw_free(vocab); srw_free(vec); return 0; }
  rc = srw_w2v_save(vocab, vec, nv, dim, o, parts);
  (*env)->ReleaseStringUTFChars(env, out, o);
  srw_free(vocab); srw_free(vec);
  if (rc != SRW_OK) { throw_status(env, rc, NULL); return 0; }      /* (SRW_ERR_EXISTS only for an existing model / vector directory; a full disk stays an IOException) */
  return (jlong)nv;
}

/* RandomWalk.save of the last walk: <output>/path/part-* + _SUCCESS + Hadoop .crc side files */
JNIEXPORT void JNICALL FN(writePaths)(JNIEnv *env, jobject self, jlong hh, jstring out, jint parts) {
  (void)self;
  const char *o = (*env)->GetStringUTFChars(env, out, NULL);
  if (!o) return;
  const int32_t rc = srw_write_paths(H(hh), o, parts, /*write_crc=*/1);
  (*env)->ReleaseStringUTFChars(env, out, o);
  if (rc != SRW_OK) throw_status(env, rc, H(hh));
}

/* Main.doRandomWalk fused and streamed (srw_walk_and_save, device-side formatter): returns the per-iteration
 * "Zero Neighbors" counts (RandomWalk.scala:117,157) */
JNIEXPORT jlongArray JNICALL FN(walkAndSave)(JNIEnv *env, jobject self, jlong hh, jfloat p, jfloat q, jint walkLength,
                                             jint numWalks, jfloat constR, jboolean useConst, jint seed, jstring out,
                                             jint parts) {
  (void)self;
  srw_walk_params P; fill_params(&P, p, q, walkLength, numWalks, 0, useConst, constR, seed, SRW_WALK_DEVICE_FORMAT);
  const size_t n = numWalks > 0 ? (size_t)numWalks : 1;
  int64_t *dead = (int64_t *)calloc(n, sizeof(int64_t));
  if (!dead) { throw_status(env, SRW_ERR_NOMEM, NULL); return NULL; }
  const char *o = (*env)->GetStringUTFChars(env, out, NULL);
  if (!o) { free(dead); return NULL; }
  const int32_t rc = srw_walk_and_save(H(hh), &P, o, parts, /*write_crc=*/1, NULL, dead);
  (*env)->ReleaseStringUTFChars(env, out, o);
  jlongArray a = NULL;
  if (rc != SRW_OK) throw_status(env, rc, H(hh));
  else {
    a = (*env)->NewLongArray(env, numWalks > 0 ? numWalks : 0);
    if (a && numWalks > 0) {
      jlong *tmp = (jlong *)malloc(n * sizeof(jlong));
      if (tmp) { for (size_t i = 0; i < n; ++i) tmp[i] = (jlong)dead[i]; (*env)->SetLongArrayRegion(env, a, 0, numWalks, tmp); free(tmp); }
    }
  }
  free(dead);
  return a;
}

/* The same job with the graph sharded by source vertex over `devices` GPUs of this node (srw_cluster_*: replaces
 * transferWalkersToTheirPartitions, RandomWalk.scala:186-192).  Returns (nVertices, nEdges, steps, zeroNeighbors). */
JNIEXPORT jlongArray JNICALL FN(walkAndSaveSharded)(JNIEnv *env, jobject self, jintArray devices, jstring input, jboolean directed,
                                                    jboolean weighted, jboolean partitioned, jint rddPartitions, jfloat p, jfloat q,
                                                    jint walkLength, jint numWalks, jfloat constR, jboolean useConst, jint seed,
                                                    jstring out, jint parts) {
  (void)self;
  jint devs[64];
  jint n = 0;
  {
    /* the JVM's own array length bounds the read; a -1 entry ends the list early (the Scala side may pad) */
    const jsize len = devices ? (*env)->GetArrayLength(env, devices) : 0;
    jint tmp[64];
    const jsize take = len < 64 ? len : 64;
    if (take > 0) (*env)->GetIntArrayRegion(env, devices, 0, take, tmp);
    while (n < take && tmp[n] >= 0) { devs[n] = tmp[n]; ++n; }
    if (n == 0) { throw_status(env, SRW_ERR_INVALID, NULL); return NULL; }
  }
  srw_cluster *c = NULL;
  /* q == 1: no shard ever tests "x in N(prev)" — the replicated neighbor-id structure is skipped (memory per shard ~ 1 / n) */
  int32_t rc = srw_cluster_create((const int32_t *)devs, n, (partitioned ? SRW_CFG_OWNER_FROM_PARTITIONS : 0) |
                                                              (q == 1.0f ? SRW_CFG_NO_MEMBERSHIP : 0), &c);
  if (rc != SRW_OK) { throw_status(env, rc, NULL); return NULL; }
  jlongArray a = NULL;
  const char *in = (*env)->GetStringUTFChars(env, input, NULL);
  const char *o = in ? (*env)->GetStringUTFChars(env, out, NULL) : NULL;
  if (in && o) {
    srw_walk_params P; fill_params(&P, p, q, walkLength, numWalks, 0, useConst, constR, seed, 0);
    srw_walk_stats st; memset(&st, 0, sizeof st);
    int64_t nv = 0, ne = 0;
    rc = srw_cluster_load_edgelist(c, in, directed, weighted, partitioned, rddPartitions);
    if (rc == SRW_OK) rc = srw_cluster_graph_stats(c, &nv, &ne);
    if (rc == SRW_OK) rc = srw_cluster_walk_and_save(c, &P, o, parts, /*write_crc=*/1, &st);
    if (rc != SRW_OK) {
      jclass cls = (*env)->FindClass(env, rc == SRW_ERR_EXISTS ? "org/apache/hadoop/mapred/FileAlreadyExistsException"
                                          : rc == SRW_ERR_PARSE ? "java/lang/NumberFormatException" : "java/lang/RuntimeException");
      if (!cls) { (*env)->ExceptionClear(env); cls = (*env)->FindClass(env, "java/lang/RuntimeException"); }
      if (cls) (*env)->ThrowNew(env, cls, srw_cluster_last_error(c));
    } else {
      jlong res[4]; res[0] = (jlong)nv; res[1] = (jlong)ne; res[2] = (jlong)st.n_steps; res[3] = (jlong)st.dead_ends;
      a = (*env)->NewLongArray(env, 4);
      if (a) (*env)->SetLongArrayRegion(env, a, 0, 4, res);
    }
  }
  if (o) (*env)->ReleaseStringUTFChars(env, out, o);
  if (in) (*env)->ReleaseStringUTFChars(env, input, in);
  srw_cluster_destroy(c);
  return a;
}

/* paths of the last walk as a flat int[] of nWalkers * (walkLength + 2) ids (-1 padded) for callers that feed the
 * embedding stage in-process; lens = path lengths */
JNIEXPORT jintArray JNICALL FN(fetchPaths)(JNIEnv *env, jobject self, jlong hh, jintArray lensOut) {
  (void)self;
  int64_t nw = 0; int32_t stride = 0; void *dp = NULL, *dl = NULL;
  int32_t rc = srw_device_paths(H(hh), &dp, &dl, &nw, &stride);
  if (rc != SRW_OK) { throw_status(env, rc, H(hh)); return NULL; }
  if (nw * (int64_t)stride > 0x7FFFFFF0ll) { throw_status(env, SRW_ERR_NOMEM, H(hh)); return NULL; }
  /* lensOut must hold one length per walker: the copy below writes nw ints into it */
  if (lensOut && (int64_t)(*env)->GetArrayLength(env, lensOut) < nw) { throw_status(env, SRW_ERR_INVALID, H(hh)); return NULL; }
  jintArray a = (*env)->NewIntArray(env, (jsize)(nw * stride));
  if (!a) return NULL;
  /* the device-to-host copies block: they go into C buffers, not into a GetPrimitiveArrayCritical region (which may stall
   * the collector for their whole duration); Set*ArrayRegion then moves the bytes into the Java arrays */
  int32_t *paths = (int32_t *)malloc((size_t)(nw * stride > 0 ? nw * stride : 1) * sizeof(int32_t));
  int32_t *lens = lensOut ? (int32_t *)malloc((size_t)(nw > 0 ? nw : 1) * sizeof(int32_t)) : NULL;
  rc = (paths && (lens || !lensOut)) ? srw_fetch_paths(H(hh), paths, lens) : SRW_ERR_NOMEM;
  if (rc == SRW_OK) {
    (*env)->SetIntArrayRegion(env, a, 0, (jsize)(nw * stride), (const jint *)paths);
    if (lens) (*env)->SetIntArrayRegion(env, lensOut, 0, (jsize)nw, (const jint *)lens);
  }
  free(paths); free(lens);
  if (rc != SRW_OK) { throw_status(env, rc, H(hh)); return NULL; }
  return a;
}

/* GraphMap.getNeighbors (algorithm/GraphMap.scala:109-120): null for an unknown vertex, else the ids in input order */
JNIEXPORT jintArray JNICALL FN(neighbors)(JNIEnv *env, jobject self, jlong hh, jint v) {
  (void)self;
  int64_t n = 0;
  int32_t rc = srw_graph_neighbors(H(hh), v, NULL, NULL, 0, &n);
  if (rc != SRW_OK) { throw_status(env, rc, H(hh)); return NULL; }
  if (n < 0) return NULL;
  jintArray a = (*env)->NewIntArray(env, (jsize)n);
  if (!a || n == 0) return a;
  jint *ids = (jint *)(*env)->GetPrimitiveArrayCritical(env, a, NULL);
  rc = ids ? srw_graph_neighbors(H(hh), v, (int32_t *)ids, NULL, n, &n) : SRW_ERR_NOMEM;
  if (ids) (*env)->ReleasePrimitiveArrayCritical(env, a, ids, 0);
  if (rc != SRW_OK) { throw_status(env, rc, H(hh)); return NULL; }
  return a;
}

JNIEXPORT jstring JNICALL FN(version)(JNIEnv *env, jclass cls) {
  (void)cls;
  return (*env)->NewStringUTF(env, srw_version());
}
