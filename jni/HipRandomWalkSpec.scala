package au.csiro.data61.randomwalk.algorithm

import java.nio.file.{Files, Paths}
import java.security.MessageDigest

import au.csiro.data61.randomwalk.common.Params
import org.scalatest.FunSuite

import scala.collection.JavaConverters._

/**
  * ScalaTest counterpart of this repository's parity tests, for a machine that has a JDK, ScalaTest and an MI355X
  * (none of which exist where the library is built; see INTEGRATION.md).  It drives HipRandomWalk exactly as
  * Main.doRandomWalk would and checks the part files against known answers that are pinned twice already:
  * tests/test_oracle_reference_vectors.py (CPU oracle, which restates RandomSample / GraphMap / RandomWalk and is itself
  * checked against the vectors of RandomSampleTest, GraphMapTest and UniformRandomWalkTest) and tests/test_gpu_parity.py
  * (the HIP path against that oracle).  The digests are sha256 over the sorted, TAB-joined, newline-terminated path
  * lines, first 16 hex digits — the DERIVED table of tests/test_oracle_reference_vectors.py, kept in step by
  * tests/test_host_cpu.py::test_scala_spec_digests_match_the_python_goldens.
  *
  * Fixture: the reference's own src/test/resources/karate.txt (34 vertices, 78 edge lines).
  */
class HipRandomWalkSpec extends FunSuite {

  private val karate = sys.props.getOrElse("srw.karate", "./src/test/resources/karate.txt")

  private def digestOf(dir: String): (String, Long, Seq[String]) = {
    val parts = Files.list(Paths.get(dir, "path")).iterator().asScala
      .filter(_.getFileName.toString.startsWith("part-")).toSeq.sortBy(_.toString)
    val lines = parts.flatMap(p => Files.readAllLines(p).asScala).sorted
    val steps = lines.map(_.split("\t").length - 1L).sum
    val md = MessageDigest.getInstance("SHA-256")
    lines.foreach(l => md.update((l + "\n").getBytes("UTF-8")))
    (md.digest().take(8).map("%02x".format(_)).mkString, steps, lines)
  }

  private def run(directed: Boolean, walkLength: Int, r: Float, p: Double, q: Double): (String, Long, Seq[String]) = {
    val out = Files.createTempDirectory("srw-spec").resolve("out").toString
    val cfg = Params(input = karate, output = out, directed = directed, weighted = false, walkLength = walkLength,
      numWalks = 1, p = p, q = q, rddPartitions = 8, partitioned = false)
    val rw = new HipRandomWalk(cfg, constR = Some(r))
    rw.execute(out, 1)
    assert(rw.nVertices == 34)
    assert(rw.nEdges == (if (directed) 78 else 156)) // UniformRandomWalkTest "load graph as (un)directed"
    digestOf(out)
  }

  // (directed, walkLength, r, p, q, steps, digest, some path prefixes)
  private val cases = Seq(
    (false, 1, 0.1f, 1.0, 1.0, 68L, "25b2f0fffab1481e", Seq("1\t22\t1", "34\t10\t3")),
    (false, 50, 0.1f, 1.0, 1.0, 1734L, "7780f0ee2a73f2b9", Seq("34\t10\t3\t2\t1\t22\t1\t22")),
    (false, 50, 0.9f, 1.0, 1.0, 1734L, "fe0f7f858ba84ae3", Seq("1\t3\t8\t4\t8\t4", "9\t33\t32\t33\t32")),
    (false, 10, 0.5f, 0.25, 4.0, 374L, "8b314af7cd74348c", Seq("1\t11\t1\t11", "9\t34\t14\t34\t14")),
    (false, 10, 0.3f, 4.0, 0.5, 374L, "b42487837425aa3a", Seq("1\t14\t3\t10\t34\t19\t33\t15\t34\t19\t33\t15")),
    (true, 50, 0.1f, 1.0, 1.0, 32L, "c21670d58f359860", Seq("1\t22", "2\t31\t34", "34")),
    (true, 50, 0.9f, 1.0, 1.0, 54L, "bdef19b134fd9227", Seq("1\t3\t4\t8", "9\t33\t34"))
  )

  for ((directed, len, r, p, q, steps, digest, prefixes) <- cases)
    test(s"karate directed=$directed walkLength=$len r=$r p=$p q=$q") {
      val (d, n, lines) = run(directed, len, r, p, q)
      assert(n == steps)
      for (pre <- prefixes) assert(lines.exists(l => l == pre || l.startsWith(pre + "\t")), pre)
      assert(d == digest)
    }

  test("an existing output directory is refused like saveAsTextFile refuses it") {
    val out = Files.createTempDirectory("srw-spec").resolve("out").toString
    val cfg = Params(input = karate, output = out, directed = false, weighted = false, walkLength = 2, numWalks = 1)
    new HipRandomWalk(cfg, constR = Some(0.5f)).execute(out, 1)
    intercept[org.apache.hadoop.mapred.FileAlreadyExistsException] {
      new HipRandomWalk(cfg, constR = Some(0.5f)).execute(out, 1)
    }
  }

  test("Philox walks do not depend on the number of GPUs") {
    val base = Files.createTempDirectory("srw-spec")
    val cfg = Params(input = karate, output = base.resolve("a").toString, directed = false, weighted = false,
      walkLength = 20, numWalks = 3, p = 0.5, q = 2.0)
    new HipRandomWalk(cfg, seed = 7).execute(base.resolve("a").toString, 1)
    new HipRandomWalk(cfg, seed = 7).executeSharded(base.resolve("b").toString, 1, Seq(0, 0)) // two shards on device 0
    assert(digestOf(base.resolve("a").toString)._1 == digestOf(base.resolve("b").toString)._1)
  }
}
