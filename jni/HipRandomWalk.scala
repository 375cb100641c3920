package au.csiro.data61.randomwalk.algorithm

import au.csiro.data61.randomwalk.common.{Params, Property}

/**
  * Host class with the RandomWalk surface (execute / save) of the reference's
  * algorithm/RandomWalk.scala:31-33,234-241; all work happens in libstellar_rw.so (MI355X, gfx950) through the JNI shim
  * jni/stellar_rw_jni.c.  Selected in Main.doRandomWalk (Main.scala:53-62) by
  *
  *   if (sys.env.get("STELLAR_RW_BACKEND").contains("hip")) {
  *     new HipRandomWalk(param).execute(param.output, getNumOutputPartition(param)); return context.emptyRDD
  *   }
  *
  * `seed` keys the Philox stream (walk iteration, source vertex, step index) — paths do not depend on scheduling, GPU
  * count or sharding; `constR` is the reference tests' injected `nextFloat = () => r`.
  */
class HipRandomWalk(config: Params, seed: Int = 42, constR: Option[Float] = None, device: Int = 0)
  extends Serializable {

  System.loadLibrary("stellar_rw_jni") // links libstellar_rw.so

  @native private def create(device: Int): Long
  @native private def destroy(h: Long): Unit
  @native private def loadEdgeList(h: Long, path: String, directed: Boolean, weighted: Boolean,
                                   partitioned: Boolean, rddPartitions: Int): Array[Long]
  @native private def walk(h: Long, p: Float, q: Float, walkLength: Int, iteration: Int,
                           constR: Float, useConst: Boolean, seed: Int): Long
  @native private def walkAll(h: Long, p: Float, q: Float, walkLength: Int, numWalks: Int,
                              constR: Float, useConst: Boolean, seed: Int): Long
  @native private def planWalks(h: Long, numWalks: Long): Unit
  @native private def w2vFitAndSave(h: Long, dim: Int, window: Int, iterations: Int, lr: Float, seed: Int, threads: Int,
                                    output: String, parts: Int): Long
  @native private def writePaths(h: Long, output: String, parts: Int): Unit
  @native private def walkAndSave(h: Long, p: Float, q: Float, walkLength: Int, numWalks: Int,
                                  constR: Float, useConst: Boolean, seed: Int, output: String,
                                  parts: Int): Array[Long]
  @native private def walkAndSaveSharded(devices: Array[Int], input: String, directed: Boolean, weighted: Boolean,
                                         partitioned: Boolean, rddPartitions: Int, p: Float, q: Float,
                                         walkLength: Int, numWalks: Int, constR: Float, useConst: Boolean,
                                         seed: Int, output: String, parts: Int): Array[Long]
  @native private def fetchPaths(h: Long, lens: Array[Int]): Array[Int]
  @native private def neighbors(h: Long, v: Int): Array[Int]

  var nVertices: Long = 0L
  var nEdges: Long = 0L

  /** execute() + save() fused: the streamed pipeline (kernel ‖ device formatter ‖ PCIe ‖ write). */
  def execute(output: String, partitions: Int): Unit = {
    val h = create(device)
    try {
      val Array(v, e) = loadEdgeList(h, config.input, config.directed, config.weighted,
        config.partitioned, config.rddPartitions)
      nVertices = v
      nEdges = e
      println(s"edges: $nEdges")
      println(s"vertices: $nVertices")
      val dead = walkAndSave(h, config.p.toFloat, config.q.toFloat, config.walkLength, config.numWalks,
        constR.getOrElse(0f), constR.isDefined, seed, output, partitions)
      for (i <- 0 until config.numWalks) {
        println("Unfinished Walkers: 0") // RandomWalk.scala:154: every walker finishes inside the launch
        if (dead(i) > 0) println(s"Zero Neighbors: ${dead(i)}")
      }
    } finally destroy(h)
  }

  /**
    * The same job with the graph sharded by source vertex over several GPUs of this node — owner(v) =
    * mix32(v) mod #GPUs, the role of HashPartitioner (RandomWalk.scala:16), or the VCut partition ids when
    * config.partitioned — and the walkers crossing shards every super-step over xGMI: what replaces
    * transferWalkersToTheirPartitions (RandomWalk.scala:186-192).  Same files as execute().
    */
  def executeSharded(output: String, partitions: Int, devices: Seq[Int]): Unit = {
    val Array(v, e, _, dead) = walkAndSaveSharded((devices :+ -1).toArray, config.input, config.directed, config.weighted,
      config.partitioned, config.rddPartitions, config.p.toFloat, config.q.toFloat, config.walkLength,
      config.numWalks, constR.getOrElse(0f), constR.isDefined, seed, output, partitions)
    nVertices = v
    nEdges = e
    println(s"edges: $nEdges")
    println(s"vertices: $nVertices")
    for (_ <- 0 until config.numWalks) println("Unfinished Walkers: 0")
    if (dead > 0) println(s"Zero Neighbors: $dead")
  }

  /**
    * `--cmd node2vec` (Main.scala:113-117): the walk, <output>/path, then Word2Vec on the paths — which stay in HBM between the two
    * stages (the reference hands randomWalk's RDD to Word2Vec.fit without collecting it).  Vectors go to <output>/vec, the model
    * directory to <output>/bin, as Main.saveModelAndFeatures (:36-44).  deterministic = one wave, sentence after sentence.
    */
  def node2vec(output: String, partitions: Int, deterministic: Boolean = false): Long = {
    val h = create(device)
    try {
      val Array(v, e) = loadEdgeList(h, config.input, config.directed, config.weighted,
        config.partitioned, config.rddPartitions)
      nVertices = v
      nEdges = e
      println(s"edges: $nEdges")
      println(s"vertices: $nVertices")
      planWalks(h, config.numWalks)
      walkAll(h, config.p.toFloat, config.q.toFloat, config.walkLength, config.numWalks,
        constR.getOrElse(0f), constR.isDefined, seed)
      for (_ <- 0 until config.numWalks) println("Unfinished Walkers: 0")
      writePaths(h, output, partitions)
      w2vFitAndSave(h, config.w2vDim, config.w2vWindow, config.w2vIter, config.w2vLr.toFloat, seed,
        if (deterministic) 1 else 0, output, partitions)
    } finally destroy(h)
  }

  /** randomWalk() with the paths returned to the JVM (e.g. to feed `--cmd node2vec`), one iteration at a time. */
  def randomWalk(): Iterator[Array[Int]] = {
    val h = create(device)
    val Array(v, e) = loadEdgeList(h, config.input, config.directed, config.weighted,
      config.partitioned, config.rddPartitions)
    nVertices = v
    nEdges = e
    val stride = config.walkLength + 2
    val its = (0 until config.numWalks).iterator.flatMap { i =>
      walk(h, config.p.toFloat, config.q.toFloat, config.walkLength, i, constR.getOrElse(0f), constR.isDefined, seed)
      val lens = new Array[Int](nVertices.toInt)
      val flat = fetchPaths(h, lens)
      (0 until nVertices.toInt).iterator.map(w => java.util.Arrays.copyOfRange(flat, w * stride, w * stride + lens(w)))
    }
    new Iterator[Array[Int]] {
      private var open = true
      def hasNext: Boolean = { val n = its.hasNext; if (!n && open) { destroy(h); open = false }; n }
      def next(): Array[Int] = its.next()
    }
  }
}

object HipRandomWalk {
  @native def version(): String
  val pathSuffix: String = Property.pathSuffix.toString
}
