"""GPU parity tests: the HIP path (through the C ABI) must equal the CPU oracle BIT FOR BIT on the same inputs.

Run on the MI355X box with `pytest -m gpu`.  Every comparison is exact (integer ids / lengths); the only
floating-point outputs (biased weights, RNG uniforms) are compared bitwise as well.
"""
import os

import numpy as np
import pytest

from conftest import KARATE, TESTGRAPH
from helpers import digest, pkg, random_multigraph, rmat_lines

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = pkg().Engine(device=0)
    yield e
    e.close()


def both_walk(eng, g, **kw):
    paths, lens, st = eng.walk(**{k: v for k, v in kw.items() if k != "threads"})
    okw = {k: v for k, v in kw.items() if k != "force_general"}
    rp, rl, rs = g.walk(**okw)
    return (paths, lens, st), (rp, rl, rs)


def assert_same(a, b, what=""):
    (paths, lens, st), (rp, rl, rs) = a, b
    assert np.array_equal(lens, rl), "lens differ " + what
    bad = np.nonzero((paths != rp).any(axis=1))[0]
    assert bad.size == 0, "paths differ %s: first at walker %d\n gpu=%s\n ref=%s" % (
        what, bad[0] if bad.size else -1, paths[bad[0]] if bad.size else None, rp[bad[0]] if bad.size else None)
    assert st["n_steps"] == rs


# ---- device arithmetic of RandomSample (T/RandomSampleTest.scala) ---------------------------------------
@pytest.mark.parametrize("r,expect", [(0.1, 0), (0.4, 1), (0.7, 2)])
def test_sample_kat(eng, r, expect):
    assert eng.sample([1.0, 1.0, 1.0], r) == expect


def test_second_order_kat(eng):
    curr = ([1, 3, 4], [1.0, 1.0, 1.0])
    assert eng.second_order_weights(1.0, 1.0, 1, [2, 4, 5], *curr).tolist() == [1.0, 1.0, 1.0]
    assert eng.second_order_weights(2.0, 2.0, 1, [2, 5], *curr).tolist() == [0.5, 0.5, 0.5]
    assert eng.second_order_weights(2.0, 2.0, 1, [2, 4, 5], *curr).tolist() == [0.5, 0.5, 1.0]
    for r, expect in [(0.1, 1), (0.4, 3), (0.7, 4)]:
        assert curr[0][eng.second_order_sample(1.0, 1.0, 1, [2, 4, 5], *curr, r)] == expect
    for r, expect in [(0.24, 1), (0.26, 3), (0.51, 4), (0.99, 4)]:
        assert curr[0][eng.second_order_sample(2.0, 2.0, 1, [2, 4, 5], *curr, r)] == expect


@pytest.mark.parametrize("deg,r,expect", [(12, 0.5, 6), (14, 0.5, 7), (6, 0.5, 2), (3, 1.5, 0)])
def test_rounding_known_answers(eng, oracle, deg, r, expect):
    assert oracle.sample_index([1.0] * deg, r) == expect
    assert eng.sample([1.0] * deg, r) == expect


def test_sample_random_lists_match_oracle(eng, oracle):
    rng = np.random.default_rng(1)
    for trial in range(60):
        n = int(rng.integers(1, 700))
        kind = trial % 4
        if kind == 0:
            w = np.ones(n, dtype=np.float32)
        elif kind == 1:
            w = rng.integers(1, 17, n).astype(np.float32)
        elif kind == 2:
            w = rng.random(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-3, 4, n).astype(np.float32)
        else:  # breaks the exact-sum certificate -> sequential fallback path
            w = (rng.random(n).astype(np.float32) * np.float32(2.0) ** rng.integers(-60, 60, n).astype(np.float32))
        for r in (0.0, 0.5, 0.25, float(np.float32(rng.random())), 0.99999994):
            assert eng.sample(w, r) == oracle.sample_index(w, r), (trial, n, r)


def test_sample_lattice_ties_all_degrees(eng, oracle):
    # u on the 2^-24 lattice hits dyadic CDF boundaries exactly; the sequential f64 sum decides
    for deg in list(range(1, 130)) + [255, 256, 1000, 4096]:
        w = np.ones(deg, dtype=np.float32)
        for r in (0.5, 0.25, 0.75, 0.125):
            assert eng.sample(w, r) == oracle.sample_index(w, r), (deg, r)


def test_sample_chain_long_rows(eng, oracle):
    """The sequential chain on LONG rows (the wave-parallel exact evaluation of acc = fl(acc + d_i), sampling.h): uniform
    weights (every lattice draw of a power-of-two degree sits on a boundary), power-of-two weights (round-half-even ties in
    every binade), wide dynamic range, zeros — against the oracle's plain loop."""
    rng = np.random.default_rng(7)
    for trial in range(24):
        n = int(rng.integers(3000, 40000)) if trial % 6 else 1 << int(rng.integers(12, 18))
        kind = trial % 6
        if kind == 0:
            w = np.ones(n, dtype=np.float32)
        elif kind == 1:
            w = (rng.integers(1, 5, n) * 0.25).astype(np.float32)
        elif kind == 2:
            w = np.float32(2.0) ** rng.integers(-6, 6, n).astype(np.float32)
        elif kind == 3:
            w = np.exp(rng.random(n) * 20.0 - 10.0).astype(np.float32)
        elif kind == 4:
            w = rng.random(n).astype(np.float32); w[::97] = 0.0
        else:
            w = rng.integers(1, 1000, n).astype(np.float32)
        for r in (0.0, 0.5, 0.25, 0.75, float(np.float32(rng.random())), float(np.float32(rng.random())), 0.99999994):
            assert eng.sample(w, r) == oracle.sample_index(w, r), (trial, n, r)


def test_sample_degenerate_weights(eng, oracle):
    for w in ([0.0, 0.0, 0.0], [1.0, -1.0, 1.0], [float("nan"), 1.0], [float("inf"), 1.0], [-2.0, -3.0],
              [0.0, 5.0, 0.0], [1e38, 1e38, 1e38, 1e38]):
        for r in (0.0, 0.3, 0.9):
            assert eng.sample(w, r) == oracle.sample_index(w, r), (w, r)


def test_f32_division_bitwise(eng, oracle):
    rng = np.random.default_rng(2)
    ids = np.arange(512, dtype=np.int32) + 10
    for p, q in [(0.3, 0.7), (3.0, 7.0), (0.25, 4.0), (1e-3, 1e3), (1.1, 0.9)]:
        w = (rng.random(512).astype(np.float32) * 100).astype(np.float32)
        prev_ids = ids[::3]
        a = eng.second_order_weights(p, q, int(ids[5]), prev_ids, ids, w)
        b = oracle.second_order_weights(p, q, int(ids[5]), prev_ids, ids, w)
        assert a.view(np.uint32).tolist() == b.view(np.uint32).tolist()


def test_power_of_two_bias_is_a_multiplication_with_the_same_bits(eng, oracle):
    """sampling.h:div_exact evaluates weight / q as weight * 2^-k when q = 2^k (and the same for p): the bits must be the
    divide's for every weight — subnormal results, subnormal and huge weights, signed zeros, infinities — at every power of
    two incl. the ends of the range the shortcut takes (2^-126 .. 2^126) and just outside it (2^127, subnormal q)."""
    rng = np.random.default_rng(5)
    special = np.array([0.0, -0.0, 1e-45, 3e-45, 1e-40, 1.1754942e-38, 1.17549435e-38, 1.1754945e-38, 2.3509887e-38, 1.0, 1.0000001,
                        0.99999994, 3.0, 16.0, 1e30, 1.7014117e38, 3.4028235e38, np.inf, 5e-324], dtype=np.float32)
    bits = rng.integers(0, 0x7F800000, size=512 - len(special), dtype=np.int64).astype(np.uint32)     # every exponent, positive
    w = np.concatenate([special, bits.view(np.float32)])
    ids = np.arange(len(w), dtype=np.int32) + 10
    prev_ids = ids[::3]
    pows = [2.0 ** k for k in (-126, -125, -100, -24, -2, -1, 1, 2, 24, 100, 126, 127)] + [float(np.float32(1e-45)), float(np.float32(2.0 ** -127)), 3.0]
    with np.errstate(all="ignore"):
        for q in pows:
            for p in (0.25, 4.0, 2.0 ** -126, 2.0 ** 126, 0.3):
                a = eng.second_order_weights(p, q, int(ids[5]), prev_ids, ids, w)
                b = oracle.second_order_weights(p, q, int(ids[5]), prev_ids, ids, w)
                assert a.view(np.uint32).tolist() == b.view(np.uint32).tolist(), (p, q)


def test_rng_stream(eng, oracle):
    it = np.array([0, 0, 0, 5, 123456], dtype=np.uint32)
    src = np.array([1, 1, 1, 4000000000, 77], dtype=np.uint32)
    st = np.array([1, 2, 3, 81, 9], dtype=np.uint32)
    got = eng.rng_uniform(42, it, src, st)
    assert got[:3].tolist() == [np.float32(0.2174382209777832), np.float32(0.21827632188796997),
                                np.float32(0.9992966651916504)]
    for k in range(5):
        assert float(got[k]) == oracle.walk_uniform(42, int(it[k]), int(src[k]), int(st[k]))


# ---- graph load (T/UniformRandomWalkTest.scala:33-86, T/GraphMapTest.scala) ------------------------------
def test_load_karate(eng, oracle):
    for directed, entries in ((False, 156), (True, 78)):
        eng.load_edgelist(KARATE, directed=directed)
        assert eng.stats() == (34, entries)
        g = oracle.Graph.load(KARATE, directed=directed)
        assert eng.vertices().tolist() == g.vertices().tolist()
        for v in range(0, 36):
            a, b = eng.neighbors(v), g.neighbors(v)
            if b is None:
                assert a is None
            else:
                assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


def test_first_step_testgraph(eng):
    eng.load_edgelist(TESTGRAPH, directed=True)
    paths, lens, st = eng.walk(walk_length=0, rng="const", const_r=0.3)
    got = {int(p[0]): p[:n].tolist() for p, n in zip(paths, lens)}
    assert got == {1: [1, 2], 2: [2]}


def test_graphmap_adjacency_surface(eng):
    # GraphMapTest.scala:7-33 through srw_load_adjacency
    e1, e2, e3, e4 = [(2, 1.0)], [(3, 1.0)], [(3, 1.0)], [(1, 1.0)]
    eng.load_adjacency([(2, e3 + e4), (1, e1 + e2), (3, []), (1, e4)])
    assert eng.stats() == (3, 4)
    assert eng.neighbors(1)[0].tolist() == [2, 3] and eng.neighbors(2)[0].tolist() == [3, 1]
    assert eng.neighbors(3)[0].tolist() == [] and eng.neighbors(99) is None
    eng.load_adjacency([(7, [(8, 3, 1.0), (9, 4, 2.0), (8, 5, 1.0)]), (8, []), (9, [])])
    assert eng.partition(8) == 5 and eng.partition(9) == 4 and eng.partition(7) is None


# ---- walks ------------------------------------------------------------------------------------------------
REF_CASES = [(False, 1, 0.1), (False, 50, 0.1), (False, 50, 0.9), (True, 50, 0.9), (True, 50, 0.1)]


@pytest.mark.parametrize("directed,L,r", REF_CASES)
@pytest.mark.parametrize("force_general", [False, True])
def test_karate_reference_cases(eng, oracle, directed, L, r, force_general):
    # T/UniformRandomWalkTest.scala:181-291 — every path equals the sequential walk
    g = oracle.Graph.load(KARATE, directed=directed)
    eng.load_edgelist(KARATE, directed=directed)
    a, b = both_walk(eng, g, walk_length=L, rng="const", const_r=r, force_general=force_general)
    assert_same(a, b)
    assert a[2]["kernel_kind"] == (2 if force_general else 1)
    for p, n in zip(a[0], a[1]):
        assert p[:n].tolist() == g.seq_walk(int(p[0]), walk_length=L, rng="const", const_r=r).tolist()


DERIVED = [(False, 1, 0.1, 1.0, 1.0, "25b2f0fffab1481e"), (False, 50, 0.1, 1.0, 1.0, "7780f0ee2a73f2b9"),
           (False, 50, 0.9, 1.0, 1.0, "fe0f7f858ba84ae3"), (False, 10, 0.5, 0.25, 4.0, "8b314af7cd74348c"),
           (False, 10, 0.3, 4.0, 0.5, "b42487837425aa3a"), (True, 50, 0.1, 1.0, 1.0, "c21670d58f359860"),
           (True, 50, 0.9, 1.0, 1.0, "bdef19b134fd9227")]


@pytest.mark.parametrize("directed,L,r,p,q,dg", DERIVED)
def test_karate_golden_digests(eng, directed, L, r, p, q, dg):
    eng.load_edgelist(KARATE, directed=directed)
    paths, lens, _ = eng.walk(walk_length=L, p=p, q=q, rng="const", const_r=r)
    assert digest(paths, lens) == dg


PQ = [(1.0, 1.0), (0.25, 4.0), (4.0, 0.5), (2.0, 2.0), (0.5, 1.0)]


@pytest.mark.parametrize("p,q", PQ)
@pytest.mark.parametrize("directed", [False, True])
def test_karate_philox(eng, oracle, p, q, directed):
    g = oracle.Graph.load(KARATE, directed=directed)
    eng.load_edgelist(KARATE, directed=directed)
    assert_same(*both_walk(eng, g, p=p, q=q, walk_length=80, num_walks=3, first_walk=2, seed=123))


@pytest.mark.parametrize("scale,weighted,directed,p,q", [
    (10, False, False, 1.0, 1.0), (12, False, False, 1.0, 1.0), (12, True, False, 1.0, 1.0),
    (12, False, True, 1.0, 1.0), (10, False, False, 0.25, 4.0), (11, True, False, 0.25, 4.0),
    (11, True, True, 4.0, 0.5), (10, True, False, 2.0, 2.0), (11, False, False, 0.5, 1.0)])
def test_rmat_vs_oracle(eng, oracle, scale, weighted, directed, p, q):
    s, d, w = rmat_lines(oracle, scale, edge_factor=8, weighted=weighted)
    g = oracle.Graph.from_coo(s, d, w, directed=directed)
    eng.load_coo(s, d, w, directed=directed)
    assert eng.stats() == (g.num_vertices, g.num_entries)
    assert_same(*both_walk(eng, g, p=p, q=q, walk_length=20, num_walks=2, seed=9, threads=8), what="rmat")


def test_rmat_first_order_equals_general(eng, oracle):
    s, d, w = rmat_lines(oracle, 12, edge_factor=16, weighted=True)
    eng.load_coo(s, d, w)
    a = eng.walk(walk_length=40, seed=5)
    b = eng.walk(walk_length=40, seed=5, force_general=True)
    assert a[2]["kernel_kind"] == 1 and b[2]["kernel_kind"] == 2
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for nt in (True, False):    # both record-load policies of the first-order kernel
        c = eng.walk(walk_length=40, seed=5, nt_loads=nt)
        assert c[2]["kernel_kind"] == 1 and np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


def test_full_size_properties(eng):
    # size-independent properties at a BASELINE-sized graph (RMAT-20, config 2): every path starts at its
    # source, has full length (undirected => no dead ends), every hop is an edge, and the two record-load
    # policies and the iteration-sharded call pattern give identical bytes.
    eng.generate_rmat(20, 16 << 20, seed=42)
    nv, ne = eng.stats()
    assert ne == 2 * (16 << 20)
    verts = eng.vertices()
    p1, l1, st = eng.walk(walk_length=80, num_walks=2, first_walk=0, seed=42, nt_loads=True)
    assert st["n_steps"] == 2 * nv * 81 and (l1 == 82).all()
    assert np.array_equal(p1[:nv, 0], verts) and np.array_equal(p1[nv:, 0], verts)
    p2, l2, _ = eng.walk(walk_length=80, num_walks=1, first_walk=1, seed=42, nt_loads=False)
    assert np.array_equal(p2, p1[nv:])            # iteration 1 alone == second half of the 2-iteration call
    rng = np.random.default_rng(0)
    for wi in rng.integers(0, 2 * nv, 40):         # spot-check that every hop is an adjacency entry
        path = p1[wi]
        for a, b in zip(path[:12], path[1:13]):
            assert int(b) in set(eng.neighbors(int(a))[0].tolist())


def test_device_rmat_generator(eng, oracle):
    scale, n = 11, 8 << 11
    for weighted in (False, True):
        s, d, w = rmat_lines(oracle, scale, edge_factor=8, weighted=weighted)
        g = oracle.Graph.from_coo(s, d, w)
        eng.generate_rmat(scale, n, seed=42, weighted=weighted)
        assert eng.stats() == (g.num_vertices, g.num_entries)
        assert eng.vertices().tolist() == g.vertices().tolist()
        for v in g.vertices()[:50].tolist() + g.vertices()[-20:].tolist():
            a, b = eng.neighbors(v), g.neighbors(v)
            assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
        assert_same(*both_walk(eng, g, walk_length=10, seed=1, threads=8))


@pytest.mark.parametrize("seed", range(6))
def test_random_multigraphs(eng, oracle, seed):
    rng = np.random.default_rng(seed)
    weighted = bool(seed % 2)
    directed = bool((seed // 2) % 2)
    s, d, w = random_multigraph(rng, 60, 300, weighted, id_lo=-20 if seed == 3 else 5)
    g = oracle.Graph.from_coo(s, d, w, directed=directed)
    eng.load_coo(s, d, w, directed=directed)
    assert eng.stats() == (g.num_vertices, g.num_entries)
    for p, q in PQ:
        assert_same(*both_walk(eng, g, p=p, q=q, walk_length=30, num_walks=2, seed=seed), what="pq=%s,%s" % (p, q))
        assert_same(*both_walk(eng, g, p=p, q=q, walk_length=12, rng="const", const_r=0.5), what="const .5")


def test_irregular_weights(eng, oracle):
    # negative / zero-sum / NaN rows: first-order kernel must take the literal sequential scan
    s = np.array([1, 1, 1, 2, 2, 3, 3, 4, 4, 4], dtype=np.int32)
    d = np.array([2, 3, 4, 1, 3, 1, 4, 1, 2, 3], dtype=np.int32)
    w = np.array([1.0, -1.0, 2.0, 0.0, 0.0, np.nan, 1.0, 3.0, 1.0, 1.0], dtype=np.float32)
    g = oracle.Graph.from_coo(s, d, w, directed=True)
    eng.load_coo(s, d, w, directed=True)
    for fg in (False, True):
        a, b = both_walk(eng, g, walk_length=16, num_walks=4, seed=3, force_general=fg)
        assert_same(a, b, what="irregular fg=%s" % fg)
    assert eng.walk(walk_length=16, num_walks=4, seed=3)[2]["fallbacks"] > 0


def test_high_degree_hub(eng, oracle):
    # one hub with 50k neighbors: multi-chunk wave scan + large guide table
    n = 50000
    s = np.concatenate([np.zeros(n, dtype=np.int32), np.arange(1, 200, dtype=np.int32)])
    d = np.concatenate([np.arange(1, n + 1, dtype=np.int32), np.arange(2, 201, dtype=np.int32)])
    rng = np.random.default_rng(0)
    w = rng.integers(1, 9, len(s)).astype(np.float32)
    g = oracle.Graph.from_coo(s, d, w)
    eng.load_coo(s, d, w)
    src = np.arange(0, 64, dtype=np.int32)
    for p, q in [(1.0, 1.0), (0.25, 4.0)]:
        paths, lens, st = eng.walk(p=p, q=q, walk_length=6, seed=11)
        rp, rl, rs = g.walk(p=p, q=q, walk_length=6, seed=11, threads=8)
        assert np.array_equal(paths, rp) and np.array_equal(lens, rl) and st["n_steps"] == rs


def test_writer_matches_oracle(eng, oracle, tmp_path):
    eng.load_edgelist(KARATE, directed=True)
    paths, lens, _ = eng.walk(walk_length=10, num_walks=2, seed=3)
    eng.write_paths(str(tmp_path / "gpu"), n_parts=3)                 # formatted on the device (path_format.hip)
    os.environ["SRW_HOST_FORMATTER"] = "1"
    try:
        eng.write_paths(str(tmp_path / "gpu_host"), n_parts=3)       # ... and by the host formatter
    finally:
        del os.environ["SRW_HOST_FORMATTER"]
    for k in range(3):
        assert (tmp_path / "gpu" / "path" / ("part-%05d" % k)).read_bytes() == \
               (tmp_path / "gpu_host" / "path" / ("part-%05d" % k)).read_bytes()
    assert oracle.write_paths(paths, lens, str(tmp_path / "ref"), 3) == 0
    for name in ("part-00000", "part-00001", "part-00002", "_SUCCESS"):
        assert (tmp_path / "gpu" / "path" / name).read_bytes() == (tmp_path / "ref" / "path" / name).read_bytes()
    with pytest.raises(pkg().SrwError) as ei:
        eng.write_paths(str(tmp_path / "gpu"))
    assert ei.value.code == pkg().ERR_EXISTS


def test_cli_end_to_end(oracle, tmp_path):
    import subprocess
    out = tmp_path / "out"
    r = subprocess.run([pkg().CLI_PATH, "--cmd", "randomwalk", "--numWalks", "1", "--p", "1", "--q", "1",
                        "--walkLength", "10", "--rddPartitions", "10", "--input", KARATE, "--output", str(out),
                        "--partitioned", "false", "--seed", "42"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["edges: 156", "vertices: 34", "E Partitions: 156", "V Partitions: 34",
                                     "Unfinished Walkers: 0"]
    g = oracle.Graph.load(KARATE)
    rp, rl, _ = g.walk(walk_length=10, seed=42)
    want = "".join("\t".join(str(int(x)) for x in p[:n]) + "\n" for p, n in zip(rp, rl))
    assert (out / "path" / "part-00000").read_text() == want
    assert (out / "path" / "_SUCCESS").read_bytes() == b""
    r2 = subprocess.run([pkg().CLI_PATH, "--cmd", "randomwalk", "--input", KARATE, "--output", str(out)],
                        capture_output=True, text=True)
    assert r2.returncode == 1 and "already exists" in r2.stderr


def test_cli_sharded_gpus_flag(oracle, tmp_path):
    """stellar-rw --gpus 3: the graph sharded by source vertex over three shards (all on device 0 here), same files as the
    single-GPU run, for a biased directed walk with dead ends."""
    import subprocess
    env = dict(os.environ, SRW_CLUSTER_SAME_DEVICE="1")
    out = tmp_path / "out3"
    r = subprocess.run([pkg().CLI_PATH, "--cmd", "randomwalk", "--numWalks", "2", "--p", "0.5", "--q", "2", "--walkLength", "12",
                        "--input", KARATE, "--output", str(out), "--directed", "true", "--seed", "7", "--gpus", "3"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines()[:2] == ["edges: 78", "vertices: 34"]
    g = oracle.Graph.load(KARATE, directed=True)
    rp, rl, _ = g.walk(p=0.5, q=2.0, walk_length=12, num_walks=2, seed=7)
    want = "".join("\t".join(str(int(x)) for x in p[:n]) + "\n" for p, n in zip(rp, rl))
    assert (out / "path" / "part-00000").read_text() == want
    r2 = subprocess.run([pkg().CLI_PATH, "--cmd", "randomwalk", "--input", KARATE, "--output", str(out), "--gpus", "3"],
                        capture_output=True, text=True, env=env)
    assert r2.returncode == 1 and "already exists" in r2.stderr


# ---- vertex-sharded path: several shards on one GPU, the in-process cluster (peer stores + events) -----------------
@pytest.mark.parametrize("world,p,q,directed", [(1, 1.0, 1.0, False), (2, 1.0, 1.0, False), (2, 0.5, 1.0, True), (3, 4.0, 1.0, False),
                                                 (2, 0.25, 4.0, False), (3, 4.0, 0.5, True), (8, 1.0, 1.0, True)])
def test_cluster_inprocess_shards(oracle, world, p, q, directed):
    """srw_cluster_*: `world` sharded handles on device 0, chunks stored straight into the receiving shard's buffer,
    super-steps ordered by events, paths kept by the home shard.  Must equal the single-process oracle for every world
    and every batching of the walk iterations."""
    s, d, w = rmat_lines(oracle, 10, edge_factor=8, weighted=True)
    g = oracle.Graph.from_coo(s, d, w, directed=directed)
    with pkg().Cluster([0] * world) as cl:
        cl.load_coo(s, d, w, directed=directed)
        assert cl.stats() == (g.num_vertices, g.num_entries)
        rp, rl, rs = g.walk(p=p, q=q, walk_length=9, num_walks=3, first_walk=4, seed=21, threads=8)
        for batch in (0, 1, 2):
            paths, lens, st = cl.walk(p=p, q=q, walk_length=9, num_walks=3, first_walk=4, seed=21, batch=batch)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp), (world, batch)
            assert st["n_steps"] == rs
        # constant r (the reference tests' injection), L = 0 .. 1
        for L in (0, 1):
            rp, rl, rs = g.walk(p=p, q=q, walk_length=L, rng="const", const_r=0.4, threads=8)
            paths, lens, st = cl.walk(p=p, q=q, walk_length=L, rng="const", const_r=0.4)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp) and st["n_steps"] == rs


@pytest.mark.parametrize("world,p,q", [(3, 1.0, 1.0), (2, 0.25, 4.0), (3, 0.5, 1.0)])
def test_cluster_hash_partitioner_owner(oracle, world, p, q):
    """SRW_CFG_OWNER_HASH_PARTITIONER: owner(v) = nonNegativeMod(v, world), the reference's HashPartitioner map (RandomWalk.scala:16,
    UniformRandomWalk.scala:42), negative ids included — the paths do not depend on the owner function."""
    s, d, w = rmat_lines(oracle, 10, edge_factor=8, weighted=True)
    s = s - 300; d = d - 300                                     # negative ids: nonNegativeMod, not %
    g = oracle.Graph.from_coo(s, d, w, directed=False)
    rp, rl, rs = g.walk(p=p, q=q, walk_length=12, num_walks=2, first_walk=1, seed=5, threads=8)
    with pkg().Cluster([0] * world, hash_partitioner=True) as cl:
        cl.load_coo(s, d, w, directed=False)
        for batch in (1, 2):
            paths, lens, st = cl.walk(p=p, q=q, walk_length=12, num_walks=2, first_walk=1, seed=5, batch=batch)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp) and st["n_steps"] == rs, (world, batch)


def test_cluster_without_replicated_membership(oracle):
    """SRW_CFG_NO_MEMBERSHIP: shards that skip the replicated neighbor-id structure (memory per shard ~ 1 / world) run every
    q == 1 walk — p = q = 1 through the linked records, p != 1 through the general step — and refuse q != 1."""
    P = pkg()
    s, d, w = rmat_lines(oracle, 10, edge_factor=8, weighted=True)
    g = oracle.Graph.from_coo(s, d, w)
    with P.Cluster([0, 0, 0], membership=False) as cl:
        cl.load_coo(s, d, w)
        assert cl.stats() == (g.num_vertices, g.num_entries)
        for p in (1.0, 0.5, 4.0):
            rp, rl, rs = g.walk(p=p, q=1.0, walk_length=12, num_walks=2, seed=8, threads=8)
            paths, lens, st = cl.walk(p=p, q=1.0, walk_length=12, num_walks=2, seed=8)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp) and st["n_steps"] == rs, p
        with pytest.raises(P.SrwError) as ei:
            cl.walk(p=1.0, q=4.0, walk_length=4)
        assert ei.value.code == P.ERR_INVALID and "SRW_CFG_NO_MEMBERSHIP" in str(ei.value)
        rp, rl, rs = g.walk(walk_length=5, seed=1, threads=8)            # the cluster stays usable after the refusal
        paths, lens, st = cl.walk(walk_length=5, seed=1)
        assert np.array_equal(lens, rl) and np.array_equal(paths, rp)


def test_cluster_walk_and_save_and_hub(oracle, tmp_path):
    """A hub every walker runs into (chunk skew: the overflow retry must kick in or the slack must hold) + the
    cluster's RandomWalk.save."""
    n = 20000
    s = np.concatenate([np.zeros(n, np.int32), np.arange(1, n + 1, dtype=np.int32)])
    d = np.concatenate([np.arange(1, n + 1, dtype=np.int32), np.roll(np.arange(1, n + 1, dtype=np.int32), 1)])
    g = oracle.Graph.from_coo(s, d, None, directed=True)                 # leaves point at each other in a ring, 0 -> all
    with pkg().Cluster([0, 0, 0, 0]) as cl:
        cl.load_coo(s, d, None, directed=True)
        rp, rl, rs = g.walk(walk_length=6, num_walks=2, seed=5, threads=8)
        paths, lens, st = cl.walk(walk_length=6, num_walks=2, seed=5)
        assert np.array_equal(lens, rl) and np.array_equal(paths, rp) and st["n_steps"] == rs
        out = tmp_path / "out"
        cl.walk_and_save(str(out), n_parts=3, walk_length=6, num_walks=2, seed=5)
        ref = tmp_path / "ref"
        oracle.write_paths(rp, rl, str(ref), n_parts=3)
        for f in sorted(os.listdir(ref / "path")):
            assert (out / "path" / f).read_bytes() == (ref / "path" / f).read_bytes(), f


def test_sharded_walker_world1_nccl(oracle):
    import torch.distributed as dist
    from importlib import import_module
    pkg()
    sharded = import_module("stellar_random_walk_amd.distributed")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        g = oracle.Graph.load(KARATE)
        drv = sharded.ShardedWalker(device=0)
        drv.load_edgelist(KARATE)
        paths, lens, stats = drv.walk(num_walks=2, p=0.5, walk_length=20, seed=3)
        rp, rl, rs = g.walk(num_walks=2, p=0.5, walk_length=20, seed=3)
        assert np.array_equal(paths, rp) and np.array_equal(lens, rl)
        assert sum(s["n_steps_global"] for s in stats) == rs and len(stats) == 1     # world 1: one batched population
        os.environ["SRW_SHARD_POPULATIONS"] = "2"                                      # two populations on two streams: same paths, in any order of use
        try:
            for _ in range(2):
                p2, l2, s2 = drv.walk(num_walks=2, p=0.5, walk_length=20, seed=3)
                assert np.array_equal(p2, rp) and np.array_equal(l2, rl) and len(s2) == 2
        finally:
            del os.environ["SRW_SHARD_POPULATIONS"]
        p1, l1, s1 = drv.walk(num_walks=2, p=0.5, walk_length=20, seed=3)
        assert np.array_equal(p1, rp) and np.array_equal(l1, rl) and len(s1) == 1
        for q, batch in ((4.0, 1), (0.5, 3)):                                       # q != 1, other batch sizes
            paths, lens, stats = drv.walk(num_walks=3, first_walk=2, batch=batch, p=0.25, q=q, walk_length=15, seed=8)
            rp, rl, rs = g.walk(num_walks=3, first_walk=2, p=0.25, q=q, walk_length=15, seed=8)
            assert np.array_equal(paths, rp) and np.array_equal(lens, rl)
            assert sum(s["n_steps_global"] for s in stats) == rs
        # p = q = 1: the shards exchange their row tables (all-reduce MAX) and walk through the fused linked kernel
        paths, lens, stats = drv.walk(num_walks=3, first_walk=1, batch=2, walk_length=25, seed=11)
        rp, rl, rs = g.walk(num_walks=3, first_walk=1, walk_length=25, seed=11)
        assert drv._linked is True
        assert np.array_equal(paths, rp) and np.array_equal(lens, rl) and sum(s["n_steps_global"] for s in stats) == rs
    finally:
        dist.destroy_process_group()


def test_giant_hub_multi_segment_membership(eng, oracle):
    # deg(hub) = 150 000 > 65 536: the membership bitmap is built in three segments; leaves are chained so that
    # N(prev) members fall into different segments; q != 1 exercises reverse marking AND per-candidate search
    n = 150000
    leaves = np.arange(1, n + 1, dtype=np.int32)
    s = np.concatenate([np.zeros(n, dtype=np.int32), leaves[:-1][::3]])
    d = np.concatenate([leaves, (leaves[:-1] + 1)[::3]])
    rng = np.random.default_rng(5)
    w = rng.integers(1, 5, len(s)).astype(np.float32)
    g = oracle.Graph.from_coo(s, d, w)
    eng.load_coo(s, d, w)
    src = np.concatenate([[0], leaves[::9973]]).astype(np.int32)
    for p, q in [(0.25, 4.0), (4.0, 0.5)]:
        paths, lens, st = eng.walk(p=p, q=q, walk_length=5, seed=17)
        rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=5, seed=17, threads=8)
        verts = eng.vertices()
        idx = np.searchsorted(verts, src)
        assert np.array_equal(paths[idx], rp) and np.array_equal(lens[idx], rl)


@pytest.mark.parametrize("pattern", ["uniform", "dyadic", "integer", "mixed"])
def test_certified_scan_on_lattice_ties(eng, oracle, pattern):
    # one source vertex per degree 1..400 (+ a few large ones); a single first step through the GENERAL kernel
    # (certified parallel scan) with draws that sit exactly on CDF boundaries; must equal the sequential oracle
    rng = np.random.default_rng(7)
    degs = list(range(1, 401)) + [512, 1000, 1024, 4096, 5000]
    src, dst, w = [], [], []
    base = 100000
    for i, dg in enumerate(degs):
        if pattern == "uniform":
            ww = np.ones(dg)
        elif pattern == "dyadic":
            ww = rng.choice([0.25, 0.5, 1.0, 2.0, 4.0], dg)
        elif pattern == "integer":
            ww = rng.integers(1, 17, dg).astype(float)
        else:
            ww = rng.random(dg) * 10.0 ** rng.integers(-2, 3, dg)
        src += [i + 1] * dg
        dst += list(range(base, base + dg))
        w += ww.tolist()
    s, d, w = np.array(src, np.int32), np.array(dst, np.int32), np.array(w, np.float32)
    g = oracle.Graph.from_coo(s, d, w, directed=True)
    eng.load_coo(s, d, w, directed=True)
    for r in (0.5, 0.25, 0.75, 0.125, 0.0, 0.99999994, 0.3333333432674408):
        a = eng.walk(walk_length=0, rng="const", const_r=r, force_general=True)
        b = g.walk(walk_length=0, rng="const", const_r=r)
        assert a[2]["kernel_kind"] == 2
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (pattern, r)
        c = eng.walk(walk_length=0, rng="const", const_r=r)           # first-order guide-table kernel, same draws
        assert np.array_equal(c[0], b[0])


def test_compact_lattice_records(eng, oracle):
    # unweighted graphs qualify for the 16-byte records (Philox draws); results must equal the exact 32-byte path
    # and the oracle; weighted graphs with large guide deltas must fall back to the exact records automatically
    s, d, _ = rmat_lines(oracle, 13, edge_factor=16)
    g = oracle.Graph.from_coo(s, d, None)
    eng.load_coo(s, d, None)
    a = eng.walk(walk_length=30, num_walks=2, seed=77)
    b = eng.walk(walk_length=30, num_walks=2, seed=77, compact=False)
    assert a[2]["record_bytes"] == 16 and b[2]["record_bytes"] == 32
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    rp, rl, _ = g.walk(walk_length=30, num_walks=2, seed=77, threads=8)
    assert np.array_equal(a[0], rp) and np.array_equal(a[1], rl)
    for nt in (True, False):
        c = eng.walk(walk_length=30, num_walks=2, seed=77, nt_loads=nt)
        assert c[2]["record_bytes"] == 16 and np.array_equal(c[0], rp)
    assert eng.walk(walk_length=5, rng="const", const_r=0.5)[2]["record_bytes"] == 32     # const-r: exact records
    eng.load_edgelist(KARATE, directed=True)                                            # dead ends + compact
    a = eng.walk(walk_length=20, num_walks=3, seed=5)
    rp, rl, _ = oracle.Graph.load(KARATE, directed=True).walk(walk_length=20, num_walks=3, seed=5)
    assert a[2]["record_bytes"] == 16 and np.array_equal(a[0], rp) and np.array_equal(a[1], rl)
    # weighted rows: guide deltas of both signs, in the hundreds on hubs -> signed 12-bit deltas, still 16-byte records
    s, d, w = rmat_lines(oracle, 13, edge_factor=16, weighted=True)
    eng.load_coo(s, d, w)
    gw = oracle.Graph.from_coo(s, d, w)
    a = eng.walk(walk_length=12, seed=1)
    rp, rl, _ = gw.walk(walk_length=12, seed=1, threads=8)
    assert a[2]["record_bytes"] == 16 and np.array_equal(a[0], rp) and np.array_equal(a[1], rl)
    b = eng.walk(walk_length=12, seed=1, compact=False)
    assert b[2]["record_bytes"] == 32 and np.array_equal(b[0], rp)
    # one hub whose weights make the guide delta overflow 12 bits (a long run of tiny weights, then heavy ones): the
    # saturated entries are resolved by bisection
    n = 30000
    hs = np.zeros(n, np.int32); hd = np.arange(1, n + 1, dtype=np.int32)
    hw = np.concatenate([np.full(n - 3000, 0.001, np.float32), np.full(3000, 50.0, np.float32)])
    gh = oracle.Graph.from_coo(hs, hd, hw)
    eng.load_coo(hs, hd, hw)
    a = eng.walk(walk_length=6, num_walks=2, seed=3)
    rp, rl, _ = gh.walk(walk_length=6, num_walks=2, seed=3, threads=8)
    assert a[2]["record_bytes"] == 16 and np.array_equal(a[0], rp) and np.array_equal(a[1], rl)
    # table build orders: exact table first (constant-r call), compact derived from it on the next Philox call
    s, d, _ = rmat_lines(oracle, 12, edge_factor=16)
    g = oracle.Graph.from_coo(s, d, None)
    eng.load_coo(s, d, None)
    c = eng.walk(walk_length=9, rng="const", const_r=0.25)
    rc = g.walk(walk_length=9, rng="const", const_r=0.25, threads=8)
    assert c[2]["record_bytes"] == 32 and np.array_equal(c[0], rc[0])
    a = eng.walk(walk_length=30, seed=3)
    rp, rl, _ = g.walk(walk_length=30, seed=3, threads=8)
    assert a[2]["record_bytes"] == 16 and np.array_equal(a[0], rp) and np.array_equal(a[1], rl)
    # ... and compact first, exact table added by a later constant-r call
    eng.load_coo(s, d, None)
    a = eng.walk(walk_length=30, seed=3)
    c = eng.walk(walk_length=9, rng="const", const_r=0.25)
    b = eng.walk(walk_length=30, seed=3)
    assert a[2]["record_bytes"] == 16 and c[2]["record_bytes"] == 32 and b[2]["record_bytes"] == 16
    assert np.array_equal(a[0], rp) and np.array_equal(b[0], rp) and np.array_equal(c[0], rc[0])


@pytest.mark.parametrize("p,q,sampler", [(1.0, 1.0, "reference"), (0.25, 4.0, "reference"), (0.25, 4.0, "alias")])
def test_walk_to_host_overlapped_equals_walk(eng, oracle, p, q, sampler):
    # srw_walk_to_host (per-iteration kernels, D2H on a second stream, two staging buffers) == srw_walk + fetch
    s, d, w = rmat_lines(oracle, 11, edge_factor=8, weighted=True)
    eng.load_coo(s, d, w)
    a = eng.walk(p=p, q=q, walk_length=20, num_walks=5, first_walk=2, seed=4, sampler=sampler)
    for pinned in (True, False):
        b = eng.walk_to_host(pinned=pinned, p=p, q=q, walk_length=20, num_walks=5, first_walk=2, seed=4, sampler=sampler)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert a[2]["n_steps"] == b[2]["n_steps"] and a[2]["kernel_kind"] == b[2]["kernel_kind"]


@pytest.mark.parametrize("n_parts,num_walks", [(1, 4), (3, 5), (7, 1)])
def test_walk_and_save_streamed(eng, oracle, tmp_path, n_parts, num_walks):
    # fused streaming pipeline == walk + oracle writer, byte for byte (incl. part boundaries inside an iteration)
    eng.load_edgelist(KARATE, directed=True)
    g = oracle.Graph.load(KARATE, directed=True)
    st, dead = eng.walk_and_save(str(tmp_path / "gpu"), n_parts=n_parts, write_crc=True, walk_length=15,
                                 num_walks=num_walks, seed=9)
    rp, rl, rs = g.walk(walk_length=15, num_walks=num_walks, seed=9)
    assert st["n_steps"] == rs
    assert oracle.write_paths(rp, rl, str(tmp_path / "ref"), n_parts) == 0
    for k in range(n_parts):
        name = "part-%05d" % k
        a = (tmp_path / "gpu" / "path" / name).read_bytes()
        assert a == (tmp_path / "ref" / "path" / name).read_bytes()
        crc = (tmp_path / "gpu" / "path" / ("." + name + ".crc")).read_bytes()
        import zlib
        assert crc[8:] == b"".join(zlib.crc32(a[o:o + 512]).to_bytes(4, "big") for o in range(0, len(a), 512))
    assert (tmp_path / "gpu" / "path" / "_SUCCESS").read_bytes() == b""
    nv = g.num_vertices
    want_dead = [int(((rl[i * nv:(i + 1) * nv] >= 2) & (rl[i * nv:(i + 1) * nv] < 17)).sum()) for i in range(num_walks)]
    assert dead == want_dead
    with pytest.raises(pkg().SrwError) as ei:
        eng.walk_and_save(str(tmp_path / "gpu"), walk_length=5)
    assert ei.value.code == pkg().ERR_EXISTS
    # device-side formatter (path_format.hip): the same bytes, .crc files and dead-end counts
    st2, dead2 = eng.walk_and_save(str(tmp_path / "gpu_fmt"), n_parts=n_parts, write_crc=True, walk_length=15,
                                   num_walks=num_walks, seed=9, device_format=True)
    assert st2["n_steps"] == rs and dead2 == want_dead
    for k in range(n_parts):
        for name in ("part-%05d" % k, ".part-%05d.crc" % k):
            assert (tmp_path / "gpu_fmt" / "path" / name).read_bytes() == (tmp_path / "gpu" / "path" / name).read_bytes()


@pytest.mark.parametrize("slice_kb,n_parts,crc", [(1, 5, False), (2, 1, False), (1, 37, False), (1, 4, True), (0, 200, False)])
def test_text_leaves_the_device_through_the_ring(eng, oracle, tmp_path, monkeypatch, slice_kb, n_parts, crc):
    """The formatted text goes device -> ring of pinned slices -> the writer's thread pool (part files written in parallel; with .crc
    files synchronously).  Small slices (SRW_TEXT_SLICE_KB, tests only) take a small graph around the ring dozens of times, with part
    boundaries inside slices and slices inside parts: the files are the oracle writer's, byte for byte — both entry points
    (srw_walk_and_save's streamed form and srw_write_paths on a resident result)."""
    if slice_kb:
        monkeypatch.setenv("SRW_TEXT_SLICE_KB", str(slice_kb))
    s, d = oracle.rmat_edges(10, 8 << 10, seed=3)
    g = oracle.Graph.from_coo(s, d, None, directed=False)
    eng.load_coo(s, d, None, directed=False)
    kw = dict(walk_length=30, num_walks=3, seed=6, p=0.5, q=2.0)
    rp, rl, _ = g.walk(threads=8, **kw)
    assert oracle.write_paths(rp, rl, str(tmp_path / "ref"), n_parts) == 0
    eng.walk_and_save(str(tmp_path / "a"), n_parts=n_parts, write_crc=crc, device_format=True, **kw)
    eng.walk(fetch=False, **kw)
    eng.write_paths(str(tmp_path / "b"), n_parts=n_parts, write_crc=crc)
    total = 0
    for k in range(n_parts):
        name = "part-%05d" % k
        want = (tmp_path / "ref" / "path" / name).read_bytes()
        total += len(want)
        for out in ("a", "b"):
            assert (tmp_path / out / "path" / name).read_bytes() == want, (out, name)
            if crc:
                import zlib
                c = (tmp_path / out / "path" / ("." + name + ".crc")).read_bytes()
                assert c[8:] == b"".join(zlib.crc32(want[o:o + 512]).to_bytes(4, "big") for o in range(0, len(want), 512))
    assert total > 200_000 and (tmp_path / "a" / "path" / "_SUCCESS").exists() and (tmp_path / "b" / "path" / "_SUCCESS").exists()
    assert len(list((tmp_path / "a" / "path").glob("part-*"))) == n_parts


def test_device_formatter_negative_ids_and_long_paths(eng, oracle, tmp_path):
    # negative ids (sign character), walkLength + 2 > 64 (two lane rounds per line), several iterations, odd part counts
    rng = np.random.default_rng(12)
    s, d, w = random_multigraph(rng, 300, 2500, True, id_lo=-150)
    g = oracle.Graph.from_coo(s, d, w, directed=True)
    eng.load_coo(s, d, w, directed=True)
    for sampler, parts in (("reference", 3), ("alias", 1)):
        eng.walk_and_save(str(tmp_path / ("h" + sampler)), n_parts=parts, walk_length=100, num_walks=3, seed=4, p=0.5, q=2.0,
                          sampler=sampler)
        eng.walk_and_save(str(tmp_path / ("d" + sampler)), n_parts=parts, walk_length=100, num_walks=3, seed=4, p=0.5, q=2.0,
                          sampler=sampler, device_format=True)
        for k in range(parts):
            a = (tmp_path / ("h" + sampler) / "path" / ("part-%05d" % k)).read_bytes()
            assert a == (tmp_path / ("d" + sampler) / "path" / ("part-%05d" % k)).read_bytes() and (len(a) > 0 or k > 0)
    rp, rl, _ = g.walk(walk_length=100, num_walks=3, seed=4, p=0.5, q=2.0, threads=8)
    assert oracle.write_paths(rp, rl, str(tmp_path / "ref"), 3) == 0
    for k in range(3):
        assert (tmp_path / "ref" / "path" / ("part-%05d" % k)).read_bytes() == \
               (tmp_path / "dreference" / "path" / ("part-%05d" % k)).read_bytes()


@pytest.mark.parametrize("world,p,q", [(2, 1.0, 1.0), (3, 0.5, 2.0)])
def test_sharded_by_user_partitions(oracle, tmp_path, world, p, q):
    # VCut input (src dst pId [w]): shards own the vertices of "their" partition (SRW_CFG_OWNER_FROM_PARTITIONS);
    # the walk result must not depend on the partitioning (what T/VCutRandomWalkTest asserts for the reference)
    rng = np.random.default_rng(4)
    s, d, w = rmat_lines(oracle, 9, edge_factor=8, weighted=True)
    pid = rng.integers(0, 5, len(s)).astype(np.int32)                      # 5 user partitions on `world` GPUs
    f = tmp_path / "vcut.txt"
    f.write_text("".join("%d %d %d %g\n" % (a, b, c, x) for a, b, c, x in zip(s, d, pid, w)))
    g = oracle.Graph.load(str(f), partitioned=True)
    P = pkg()
    with P.Cluster([0] * world, owner_from_partitions=True) as cl:
        cl.load_edgelist(str(f), partitioned=True)
        assert cl.stats() == (g.num_vertices, g.num_entries)
        # ownership follows the recorded partition (last pId seen for the vertex), modulo world
        shards = [cl.shard(r) for r in range(world)]
        owned = [set(e.vertices().tolist()) for e in shards]
        assert sum(len(o) for o in owned) == g.num_vertices and not (owned[0] & owned[1])
        for v in list(owned[0])[:50]:
            assert shards[0].partition(v) % world == 0
        rp, rl, _ = g.walk(p=p, q=q, walk_length=8, first_walk=1, seed=33, threads=8)
        paths, lens, _ = cl.walk(p=p, q=q, walk_length=8, first_walk=1, seed=33)
        assert np.array_equal(paths, rp) and np.array_equal(lens, rl)


def test_table_variants_agree_beyond_cache_size(eng):
    # RMAT-22 (4.3 GB exact table, 2.1 GB compact table: larger than L2 + Infinity Cache, so the L1-bypassing load
    # policy is the default): compact/exact records x both load policies x general kernel give identical bytes
    eng.generate_rmat(22, 16 << 22, seed=42)
    nv, ne = eng.stats()
    base = eng.walk(walk_length=40, seed=3)
    assert base[2]["record_bytes"] == 16 and base[2]["n_steps"] == nv * 41
    for kw in (dict(compact=False), dict(compact=False, nt_loads=False), dict(nt_loads=False)):
        other = eng.walk(walk_length=40, seed=3, **kw)
        assert np.array_equal(base[1], other[1]) and np.array_equal(base[0], other[0]), kw
    st, dead = None, None
    sub = eng.walk(walk_length=3, seed=3, force_general=True)           # exact streaming sampler on the same graph
    assert np.array_equal(sub[0][:, :5], base[0][:, :5])


def test_empty_and_degenerate_inputs(eng, oracle, tmp_path):
    # empty edge list: zero vertices, zero paths, an empty part-00000 + _SUCCESS (what saveAsTextFile leaves behind)
    f = tmp_path / "empty.txt"
    f.write_bytes(b"")
    eng.load_edgelist(str(f))
    assert eng.stats() == (0, 0) and len(eng.vertices()) == 0
    paths, lens, st = eng.walk(walk_length=5, num_walks=3)
    assert len(lens) == 0 and st["n_steps"] == 0
    eng.write_paths(str(tmp_path / "o1"))
    assert (tmp_path / "o1" / "path" / "part-00000").read_bytes() == b"" and (tmp_path / "o1" / "path" / "_SUCCESS").exists()
    st, dead = eng.walk_and_save(str(tmp_path / "o2"), n_parts=2, walk_length=5)
    assert sorted(os.listdir(tmp_path / "o2" / "path")) == ["_SUCCESS", "part-00000", "part-00001"]
    # a single self-loop line, undirected: the vertex gets two entries (src->dst and dst->src) and walks on itself
    f2 = tmp_path / "loop.txt"
    f2.write_text("7 7 2.5\n")
    eng.load_edgelist(str(f2))
    g = oracle.Graph.load(str(f2))
    assert eng.stats() == (1, 2) == (g.num_vertices, g.num_entries)
    for p, q in ((1.0, 1.0), (0.25, 4.0)):
        a = eng.walk(walk_length=6, num_walks=2, p=p, q=q, seed=1)
        b = g.walk(walk_length=6, num_walks=2, p=p, q=q, seed=1)
        assert np.array_equal(a[0], b[0]) and a[0].tolist() == [[7] * 8] * 2


# ---- binned prefix-sum search (Mode R, q != 1): every intersection strategy, every row size ---------------------
BINNED_TUNES = [0, 8 | 1, 8 | 2, 8 | 3, 8 | 4]     # automatic; forced P1 / P2 / id-window / hub bitmap on rows of ANY degree


@pytest.mark.parametrize("case", ["rmat12", "rmat12w", "rmat11wd", "rmat12f", "rmat12x", "multi", "multi_neg"])
def test_binned_search_all_strategies(eng, oracle, case):
    if case.startswith("rmat"):
        sc = int(case[4:6])
        weighted, directed = "w" in case[6:], case.endswith("d")
        s, d, w = rmat_lines(oracle, sc, edge_factor=16, weighted=weighted)
        if case.endswith("f"):      # arbitrary f32 mantissas in a narrow exponent range: the certificate holds
            w = (0.5 + 1.5 * np.random.default_rng(4).random(len(s))).astype(np.float32)
        if case.endswith("x"):      # exponents spread over 2^-20 .. 2^20: hub rows fail the certificate -> scan path
            w = (2.0 ** np.random.default_rng(5).integers(-20, 21, len(s))).astype(np.float32)
    else:
        rng = np.random.default_rng(11)
        weighted, directed = True, False
        s, d, w = random_multigraph(rng, 80, 900, True, id_lo=-30 if case == "multi_neg" else 3)
    g = oracle.Graph.from_coo(s, d, w, directed=directed)
    eng.load_coo(s, d, w, directed=directed)
    for p, q in [(0.25, 4.0), (4.0, 0.5), (2.0, 2.0)]:
        ref = g.walk(p=p, q=q, walk_length=24, num_walks=2, seed=21, threads=8)
        for tune in BINNED_TUNES:      # the on-the-fly strategies: per-edge tables off
            paths, lens, st = eng.walk(p=p, q=q, walk_length=24, num_walks=2, seed=21, binned_tune=tune, edge_tables=False)
            assert np.array_equal(lens, ref[1]) and np.array_equal(paths, ref[0]), (case, p, q, tune)
            if tune and not case.endswith("x"):
                assert st["ent_reads"] > 0, "binned search was not exercised"
                key = {1: "p1", 2: "p2", 3: "w", 4: "p3"}[tune & 7]
                assert st["strategy_steps"][key] > 0 or (key == "p3" and st["strategy_steps"]["w"] + st["strategy_steps"]["p1"] > 0), st
        paths, lens, st = eng.walk(p=p, q=q, walk_length=24, num_walks=2, seed=21)          # default: per-edge tables
        assert np.array_equal(paths, ref[0]) and st["strategy_steps"]["edge_mask"] > 0, st
        paths, lens, st = eng.walk(p=p, q=q, walk_length=24, num_walks=2, seed=21, binned=False)
        assert np.array_equal(paths, ref[0])
        paths, lens, st = eng.walk(p=p, q=q, walk_length=24, num_walks=2, seed=21, hub_bitmaps=False, edge_tables=False)
        assert np.array_equal(paths, ref[0])
        # draws exactly on CDF boundaries (constant r on the 2^-24 lattice)
        for r in (0.5, 0.25):
            refc = g.walk(p=p, q=q, walk_length=8, rng="const", const_r=r, threads=8)
            for tune in BINNED_TUNES:
                pc, lc, _ = eng.walk(p=p, q=q, walk_length=8, rng="const", const_r=r, binned_tune=tune, edge_tables=False)
                assert np.array_equal(pc, refc[0]) and np.array_equal(lc, refc[1]), (case, p, q, tune, r)
            pc, lc, _ = eng.walk(p=p, q=q, walk_length=8, rng="const", const_r=r)
            assert np.array_equal(pc, refc[0]) and np.array_equal(lc, refc[1]), (case, p, q, "tables", r)


def test_binned_search_two_giant_hubs(eng, oracle):
    # hubs 0 and 1 (degree ~100 000 each, > 65 536 so chunks are wider than 64 candidates) share half of their
    # leaves and are adjacent: hub -> hub steps take the id-window intersection with tens of thousands of members
    n = 100000
    leaves0 = np.arange(10, 10 + n, dtype=np.int32)
    leaves1 = np.arange(10 + n // 2, 10 + n // 2 + n, dtype=np.int32)
    s = np.concatenate([np.zeros(n, np.int32), np.ones(n, np.int32), [0, 0]]).astype(np.int32)
    d = np.concatenate([leaves0, leaves1, [1, 1]]).astype(np.int32)        # double edge between the hubs
    rng = np.random.default_rng(3)
    w = rng.integers(1, 9, len(s)).astype(np.float32)
    g = oracle.Graph.from_coo(s, d, w)
    eng.load_coo(s, d, w)
    src = np.concatenate([[0, 1], leaves0[::7919], leaves1[::7877]]).astype(np.int32)
    verts = eng.vertices()
    idx = np.searchsorted(verts, src)
    for p, q in [(0.25, 4.0), (4.0, 0.5)]:
        rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=8, seed=29, threads=8)
        for tune in (0, 3, 1, 4):
            paths, lens, st = eng.walk(p=p, q=q, walk_length=8, seed=29, binned_tune=tune, edge_tables=False)
            assert np.array_equal(paths[idx], rp) and np.array_equal(lens[idx], rl), (p, q, tune)
            assert st["ent_reads"] > 0


# ---- per-edge bias tables (edge_tables.hip): a (prev -> curr) pair's N(prev) ∩ N(curr) corrections precomputed --------
@pytest.mark.parametrize("case", ["rmat12", "rmat12w", "rmat11wd", "rmat12f", "rmat12x", "multi", "multi_neg"])
def test_edge_tables_every_pair(eng, oracle, case):
    """Test switch edge_tables_all: a table for EVERY certified pair, chunks of 4 candidates — every second-order step of
    a certified row goes through the table search; must equal the oracle (and the on-the-fly search) bit for bit."""
    if case.startswith("rmat"):
        sc = int(case[4:6])
        weighted, directed = "w" in case[6:], case.endswith("d")
        s, d, w = rmat_lines(oracle, sc, edge_factor=16, weighted=weighted)
        if case.endswith("f"):
            w = (0.5 + 1.5 * np.random.default_rng(4).random(len(s))).astype(np.float32)
        if case.endswith("x"):      # hub rows fail the certificate: no tables for pairs that lead to them
            w = (2.0 ** np.random.default_rng(5).integers(-20, 21, len(s))).astype(np.float32)
    else:
        rng = np.random.default_rng(11)
        weighted, directed = True, False
        s, d, w = random_multigraph(rng, 80, 900, True, id_lo=-30 if case == "multi_neg" else 3)
    g = oracle.Graph.from_coo(s, d, w, directed=directed)
    eng.load_coo(s, d, w, directed=directed)
    for p, q in [(0.25, 4.0), (4.0, 0.5), (2.0, 2.0)]:
        ref = g.walk(p=p, q=q, walk_length=24, num_walks=2, seed=21, threads=8)
        paths, lens, st = eng.walk(p=p, q=q, walk_length=24, num_walks=2, seed=21, edge_tables_all=True)
        assert np.array_equal(lens, ref[1]) and np.array_equal(paths, ref[0]), (case, p, q)
        assert st["edge_tables"] > 0 and st["strategy_steps"]["edge_table"] > 0, st
        if not case.endswith("x"):   # nearly every second-order step of a certified row is a table step
            assert st["strategy_steps"]["edge_table"] > 0.5 * st["n_steps"], st
        off, _, st0 = eng.walk(p=p, q=q, walk_length=24, num_walks=2, seed=21, edge_tables=False)
        assert st0["edge_tables"] == 0 and st0["strategy_steps"]["edge_table"] == 0 and np.array_equal(off, ref[0])
        for r in (0.5, 0.25, 0.999999):      # draws exactly on CDF boundaries
            refc = g.walk(p=p, q=q, walk_length=8, rng="const", const_r=r, threads=8)
            pc, lc, _ = eng.walk(p=p, q=q, walk_length=8, rng="const", const_r=r, edge_tables_all=True)
            assert np.array_equal(pc, refc[0]) and np.array_equal(lc, refc[1]), (case, p, q, r)


def test_edge_tables_hubs_and_leaves(eng, oracle):
    """Default selection on two 100 000-entry hubs sharing half of their leaves, joined by a double edge (a multi-edge is one pair: its entries share one table): every pair that
    leads INTO a hub (leaf -> hub, hub -> hub) gets chunk prefixes (chunks of 2048 candidates), every pair that leads into
    a leaf (1 .. 2 candidates) an inline membership mask."""
    n = 100000
    leaves0 = np.arange(10, 10 + n, dtype=np.int32)
    leaves1 = np.arange(10 + n // 2, 10 + n // 2 + n, dtype=np.int32)
    s = np.concatenate([np.zeros(n, np.int32), np.ones(n, np.int32), [0, 0]]).astype(np.int32)
    d = np.concatenate([leaves0, leaves1, [1, 1]]).astype(np.int32)
    w = np.random.default_rng(3).integers(1, 9, len(s)).astype(np.float32)
    g = oracle.Graph.from_coo(s, d, w)
    eng.load_coo(s, d, w)
    src = np.concatenate([[0, 1], leaves0[::7919], leaves1[::7877]]).astype(np.int32)
    idx = np.searchsorted(eng.vertices(), src)
    for p, q in [(0.25, 4.0), (4.0, 0.5)]:
        rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=8, seed=29, threads=8)
        paths, lens, st = eng.walk(p=p, q=q, walk_length=8, seed=29)
        assert np.array_equal(paths[idx], rp) and np.array_equal(lens[idx], rl), (p, q)
        assert st["edge_tables"] == 2 * n + 2, st     # every leaf -> hub entry; the double edge 0 -> 1 (and 1 -> 0) is ONE pair: one table, both entries point at it
        assert st["strategy_steps"]["edge_table"] > 0 and st["strategy_steps"]["edge_mask"] > 0, st
        served = sum(st["strategy_steps"][k] for k in ("p1", "p2", "w", "p3"))
        assert served == 0, st                         # nothing is left to the on-the-fly intersections
        again, _, st2 = eng.walk(p=p, q=q, walk_length=8, seed=29)      # tables are reused, not rebuilt
        assert np.array_equal(again, paths) and st2["setup_ms"] < st["setup_ms"] + 50.0
    # a different (p, q) rebuilds them
    rp, rl, _ = g.walk(sources=src, p=0.5, q=2.0, walk_length=8, seed=31, threads=8)
    paths, lens, st = eng.walk(p=0.5, q=2.0, walk_length=8, seed=31)
    assert np.array_equal(paths[idx], rp) and st["edge_tables"] == 2 * n + 2


def test_giant_row_mass_certificate(eng, oracle):
    """A row beyond 524 288 entries (the bound n * 2^(emax+1) would refuse it; the mass-based certificate accepts it):
    hub 0 with 600 000 leaves + hub 1 sharing 50 000 of them; the searches must serve the hub steps."""
    n0, n1 = 600000, 60000
    leaves0 = np.arange(10, 10 + n0, dtype=np.int32)
    leaves1 = np.arange(10 + n0 - 50000, 10 + n0 - 50000 + n1, dtype=np.int32)
    s = np.concatenate([np.zeros(n0, np.int32), np.ones(n1, np.int32), [0]]).astype(np.int32)
    d = np.concatenate([leaves0, leaves1, [1]]).astype(np.int32)
    w = np.random.default_rng(8).integers(1, 17, len(s)).astype(np.float32)
    g = oracle.Graph.from_coo(s, d, w)
    eng.load_coo(s, d, w)
    src = np.concatenate([[0, 1], leaves0[::59999], leaves1[::5999]]).astype(np.int32)
    idx = np.searchsorted(eng.vertices(), src)
    for p, q in [(0.25, 4.0), (4.0, 0.5)]:
        rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=6, seed=77, threads=8)
        for kw in ({}, {"edge_tables": False}, {"edge_tables": False, "hub_bitmaps": False}):
            paths, lens, st = eng.walk(p=p, q=q, walk_length=6, seed=77, **kw)
            assert np.array_equal(paths[idx], rp) and np.array_equal(lens[idx], rl), (p, q, kw)
            served = sum(st["strategy_steps"][k] for k in ("edge_table", "p1", "p2", "w", "p3"))
            assert served > 0.4 * st["n_steps"], st     # every step that lands on a hub is served by a search, not the scan


# ---- p != 1, q == 1: one walker per lane (k_walk_q1) ----------------------------------------------------------------------
@pytest.mark.parametrize("case", ["rmat12", "rmat12w", "rmat11wd", "rmat12f", "rmat12x", "multi", "multi_neg", "karate", "hub", "multi:wave", "rmat12w:wave"])
def test_q1_per_lane_kernel(eng, oracle, monkeypatch, case):
    """Return-edge-only bias: the per-lane kernel (guide table + exact prefix sums + return-edge position) must equal the
    oracle, hand over what it cannot certify (several return edges, irregular rows, lattice ties) and say so.  ":wave": every
    step with more than ONE return edge is taken by the whole wave (wave_pick_returns; the default threshold is 16)."""
    if case.endswith(":wave"):
        case = case[:-5]
        monkeypatch.setenv("SRW_Q1_MAX_RET", "1")
    directed = False
    if case.startswith("rmat"):
        sc = int(case[4:6])
        weighted, directed = "w" in case[6:], case.endswith("d")
        s, d, w = rmat_lines(oracle, sc, edge_factor=16, weighted=weighted)
        if case.endswith("f"):
            w = (0.5 + 1.5 * np.random.default_rng(4).random(len(s))).astype(np.float32)
        if case.endswith("x"):      # exponents spread over 2^-20 .. 2^20: hub rows fail the certificate -> general kernel for all
            w = (2.0 ** np.random.default_rng(5).integers(-20, 21, len(s))).astype(np.float32)
    elif case == "karate":
        eng.load_edgelist(KARATE)
        g = oracle.Graph.load(KARATE)
        s = None
    elif case == "hub":
        n = 30000
        s = np.concatenate([np.zeros(n, np.int32), np.arange(1, n, dtype=np.int32)])
        d = np.concatenate([np.arange(1, n + 1, dtype=np.int32), np.arange(2, n + 1, dtype=np.int32)])
        w = np.random.default_rng(9).integers(1, 5, len(s)).astype(np.float32)
    else:
        s, d, w = random_multigraph(np.random.default_rng(11), 80, 900, True, id_lo=-30 if case == "multi_neg" else 3)
    if s is not None:
        g = oracle.Graph.from_coo(s, d, w, directed=directed)
        eng.load_coo(s, d, w, directed=directed)
    for p in (0.25, 4.0, 0.5):
        ref = g.walk(p=p, q=1.0, walk_length=30, num_walks=2, first_walk=1, seed=77, threads=8)
        paths, lens, st = eng.walk(p=p, q=1.0, walk_length=30, num_walks=2, first_walk=1, seed=77)
        assert np.array_equal(lens, ref[1]) and np.array_equal(paths, ref[0]) and st["n_steps"] == ref[2], (case, p)
        ss = st["strategy_steps"]
        if case in ("rmat12", "rmat12w", "rmat11wd", "rmat12f", "karate", "hub"):
            assert ss["q1_lane"] > 0.5 * st["n_steps"], (case, st)
        if case == "rmat12x":
            assert ss["q1_lane"] == 0, st                       # uncertified rows: the whole call stays with the general kernel
        # draws exactly on CDF boundaries: constant r runs the general kernel (no lattice stream), must agree as well
        refc = g.walk(p=p, q=1.0, walk_length=8, rng="const", const_r=0.5, threads=8)
        pc, lc, _ = eng.walk(p=p, q=1.0, walk_length=8, rng="const", const_r=0.5)
        assert np.array_equal(pc, refc[0]) and np.array_equal(lc, refc[1])


def test_non_ascii_digits_reach_the_host_tokenizer(eng, oracle, tmp_path):
    """A file with Unicode decimal digits (which Integer.parseInt accepts) is not for the device tokenizer: it must hand the
    file over to the host tokenizer, and the graph must be the oracle's."""
    f = tmp_path / "u.txt"
    f.write_bytes("1 2\n２ ٣\n3 1\n٣ ४\n".encode("utf-8"))
    eng.load_edgelist(str(f), weighted=False)
    g = oracle.Graph.load(str(f), weighted=False)
    assert eng.stats() == (g.num_vertices, g.num_entries) == (4, 8)
    assert eng.vertices().tolist() == [1, 2, 3, 4]
    paths, lens, _ = eng.walk(walk_length=6, seed=5, p=0.5, q=2.0)
    rp, rl, _ = g.walk(walk_length=6, seed=5, p=0.5, q=2.0)
    assert np.array_equal(paths, rp) and np.array_equal(lens, rl)


def test_input_directory_and_byte_order_mark(eng, oracle, tmp_path):
    """--input naming a directory of part files (sc.textFile semantics: hidden files skipped, files in name order) and a file
    that starts with a UTF-8 byte order mark (skipped by Hadoop's LineRecordReader): both leave the device tokenizer for the
    host tokenizer, and the graph is the oracle's graph of the concatenated lines."""
    lines = open(KARATE).read().splitlines()
    d = tmp_path / "in"
    d.mkdir()
    (d / "part-00000").write_text("\n".join(lines[:30]) + "\n")
    (d / "part-00001").write_text("\n".join(lines[30:]))                 # no final newline
    (d / "_SUCCESS").write_text("")
    (d / ".part-00000.crc").write_bytes(b"crc\x00junk")
    g = oracle.Graph.load(KARATE, weighted=False)
    rp, rl, _ = g.walk(walk_length=10, seed=3, p=0.25, q=4.0)
    eng.load_edgelist(str(d), weighted=False)
    assert eng.stats() == (34, 156)
    paths, lens, _ = eng.walk(walk_length=10, seed=3, p=0.25, q=4.0)
    assert np.array_equal(paths, rp) and np.array_equal(lens, rl)
    bom = tmp_path / "bom.txt"
    bom.write_bytes(b"\xef\xbb\xbf" + open(KARATE, "rb").read())
    eng.load_edgelist(str(bom), weighted=False)
    assert eng.stats() == (34, 156)
    paths, lens, _ = eng.walk(walk_length=10, seed=3, p=0.25, q=4.0)
    assert np.array_equal(paths, rp) and np.array_equal(lens, rl)


def test_gz_input_through_the_cli(oracle, tmp_path):
    """stellar-rw --input karate.txt.gz: the device tokenizer declines, the host tokenizer inflates; the part file is the
    oracle's."""
    import gzip, subprocess
    from conftest import ROOT
    gz = tmp_path / "karate.txt.gz"
    gz.write_bytes(gzip.compress(open(KARATE, "rb").read()))
    out = tmp_path / "out"
    cli = os.path.join(ROOT, "stellar-random-walk_amd", "stellar-rw")
    r = subprocess.run([cli, "--cmd", "randomwalk", "--input", str(gz), "--output", str(out), "--walkLength", "12", "--numWalks", "2",
                        "--p", "0.5", "--q", "2.0", "--weighted", "false", "--seed", "9"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "vertices: 34" in r.stdout and "edges: 156" in r.stdout
    g = oracle.Graph.load(KARATE, weighted=False)
    rp, rl, _ = g.walk(walk_length=12, num_walks=2, seed=9, p=0.5, q=2.0)
    want = sorted("\t".join(str(int(x)) for x in p_[:n]) for p_, n in zip(rp, rl))
    got = []
    for name in sorted(os.listdir(out / "path")):
        if name.startswith("part-"):
            got += (out / "path" / name).read_text().splitlines()
    assert sorted(got) == want


def test_wide_id_range_is_compacted(eng, oracle):
    """The reference's GraphMap is a HashMap: any int32 ids load.  A sparse id space is compacted at load (slot = rank
    among the sorted distinct ids) — the whole int32 range included, with and without partition ids — and
    the handle stays usable for a dense graph afterwards.  tests/test_sparse_ids.py holds the parity tests proper."""
    for s, d, pid in [(np.array([2147483647], np.int32), np.array([-2147483648], np.int32), None),
                      (np.array([2147483647], np.int32), np.array([-2147483648], np.int32), np.zeros(1, np.int32)),
                      (np.array([0, 2000000000], np.int32), np.array([1, 3], np.int32), None)]:
        eng.load_coo(s, d, pid=pid)
        g = oracle.Graph.from_coo(s, d, None)
        assert eng.stats() == (g.num_vertices, g.num_entries)
        paths, lens, _ = eng.walk(walk_length=5, seed=3)
        rp, rl, _ = g.walk(walk_length=5, seed=3)
        assert np.array_equal(paths, rp) and np.array_equal(lens, rl)
    eng.load_edgelist(KARATE)
    assert eng.stats() == (34, 156)
    eng.load_edgelist(KARATE)
    assert eng.stats() == (34, 156)


def test_num_walks_zero_writes_an_empty_job(eng, tmp_path):
    """--numWalks 0: the reference's `0 until numWalks` loop runs zero times and still writes path/ + _SUCCESS."""
    eng.load_edgelist(KARATE)
    st = eng.walk(fetch=False, num_walks=0, walk_length=5)
    assert st["n_walkers"] == 0 and st["n_steps"] == 0
    out = tmp_path / "out"
    st, dead = eng.walk_and_save(str(out), n_parts=1, num_walks=0, walk_length=5)
    assert (out / "path" / "_SUCCESS").exists() and (out / "path" / "part-00000").read_bytes() == b""
    with pytest.raises(pkg().SrwError):
        eng.walk(fetch=False, num_walks=-1)


def test_randomized_differential_fuzz(eng):
    # ~8 s of tests/fuzz_parity.py: random multigraphs / hub graphs / RMATs, every sampler variant vs the CPU oracle
    import fuzz_parity
    assert fuzz_parity.run(budget=8.0, seed=20260928, eng=eng)


def _graph_snapshot(eng):
    verts = eng.vertices()
    nb = [eng.neighbors(int(v)) for v in verts[:: max(1, len(verts) // 50)]]
    return eng.stats(), verts.tolist(), [(a.tolist(), b.tolist()) for a, b in nb]


def test_device_tokenizer_equals_host_tokenizer(eng, oracle, tmp_path, monkeypatch):
    # two-column integer files go through the device tokenizer (edgelist_device.hip); everything else falls back to
    # the host tokenizer.  Same graph either way, and the same errors for malformed input.
    rng = np.random.default_rng(8)
    n = 5000
    a = rng.integers(-300, 900, n); b = rng.integers(-300, 900, n)
    sep = rng.choice([" ", "\t", "  ", " \t "], n); tail = rng.choice(["", "", " ", "\t\t"], n)
    sign = rng.choice(["", "", "+"], n)
    body = "\n".join("%s%d%s%d%s" % (sg if x >= 0 else "", x, s_, y, t) for x, y, s_, t, sg in zip(a, b, sep, tail, sign))
    cases = {
        "plain": body + "\n",
        "no_final_newline": body,
        "one_line": "7 9",
        "signs": "1000000 -1000000\n0 1\n-5 +5\n+0 -0\n",
        "leading_zeros": "007 0012\n3 4\n",
    }
    for name, text in cases.items():
        f = tmp_path / (name + ".txt")
        f.write_text(text)
        for directed in (False, True):
            monkeypatch.delenv("SRW_HOST_TOKENIZER", raising=False)
            eng.load_edgelist(str(f), directed=directed)
            dev = _graph_snapshot(eng)
            monkeypatch.setenv("SRW_HOST_TOKENIZER", "1")
            eng.load_edgelist(str(f), directed=directed)
            assert _graph_snapshot(eng) == dev, (name, directed)
        g = oracle.Graph.load(str(f), directed=False)
        monkeypatch.delenv("SRW_HOST_TOKENIZER", raising=False)
        eng.load_edgelist(str(f), directed=False)
        assert eng.stats() == (g.num_vertices, g.num_entries), name
        # a file larger than what the device tokenizer may take next to its reserve (a third of the free HBM; here: the test's cap of
        # 0 MB) goes to the host tokenizer by itself
        monkeypatch.setenv("SRW_DEVICE_TOKENIZER_MAX_MB", "0")
        eng.load_edgelist(str(f), directed=False)
        assert eng.stats() == (g.num_vertices, g.num_entries), name
        monkeypatch.delenv("SRW_DEVICE_TOKENIZER_MAX_MB", raising=False)
        # files of 4 GiB and more count their newlines in 64 bits before the 32-bit scan (a wrapped count would pass the 2^31-line guard);
        # the switch takes that branch on this small file
        monkeypatch.setenv("SRW_TOKENIZER_COUNT64", "1")
        eng.load_edgelist(str(f), directed=False)
        assert eng.stats() == (g.num_vertices, g.num_entries), name
        monkeypatch.delenv("SRW_TOKENIZER_COUNT64", raising=False)
    # weighted files with short decimal weights are tokenized on the device too: bitwise the same weights as strtof
    wrng = np.random.default_rng(21)
    def wtoken(i):
        kind = i % 6
        if kind == 0: return str(int(wrng.integers(0, 9999999)))
        if kind == 1: return "%.*f" % (int(wrng.integers(1, 7)), wrng.random() * 10 ** int(wrng.integers(0, 2)))
        if kind == 2: return "0." + "0" * int(wrng.integers(0, 4)) + str(int(wrng.integers(1, 9999999)))[: 10 - 4]
        if kind == 3: return ("-" if wrng.random() < 0.5 else "+") + "%.3f" % (wrng.random() * 100)
        if kind == 4: return str(int(wrng.integers(0, 999))) + "."
        return "." + str(int(wrng.integers(0, 9999999)))
    wlines = ["%d %d %s" % (x, y, wtoken(i)) for i, (x, y) in enumerate(zip(a[:3000], b[:3000]))]
    wlines[5] = "%d %d" % (a[5], b[5])                      # a two-column line in a weighted file: weight 1.0
    fw = tmp_path / "weighted_simple.txt"
    fw.write_text("\n".join(wlines) + "\n")
    for directed in (False, True):
        monkeypatch.delenv("SRW_HOST_TOKENIZER", raising=False)
        eng.load_edgelist(str(fw), directed=directed, weighted=True)
        dev = _graph_snapshot(eng)
        devw = np.concatenate([eng.neighbors(int(v))[1] for v in eng.vertices()]).view(np.uint32)
        monkeypatch.setenv("SRW_HOST_TOKENIZER", "1")
        eng.load_edgelist(str(fw), directed=directed, weighted=True)
        hostw = np.concatenate([eng.neighbors(int(v))[1] for v in eng.vertices()]).view(np.uint32)
        assert _graph_snapshot(eng) == dev and np.array_equal(devw, hostw)
    monkeypatch.delenv("SRW_HOST_TOKENIZER", raising=False)
    # not the fast shape: host tokenizer decides (weights parsed, CRLF accepted, errors raised as before)
    monkeypatch.delenv("SRW_HOST_TOKENIZER", raising=False)
    ok_cases = {"weighted": "1 2 0.5\n2 3 1.5\n", "crlf": "1 2\r\n2 3\r\n", "three_ints": "1 2 3\n4 5 6\n",
                "exp_weights": "1 2 1e-3\n2 3 2.5E2\n", "long_weights": "1 2 0.123456789012\n2 3 16777217\n",
                "odd_weights": "1 2 NaN\n2 3 0x1p3\n3 4 1.5f\n4 5 abc\n5 6 .\n", "four_cols": "1 2 9 0.25\n2 3 9 4\n"}
    for name, text in ok_cases.items():
        f = tmp_path / (name + ".txt")
        f.write_text(text)
        g = oracle.Graph.load(str(f), directed=False)
        eng.load_edgelist(str(f), directed=False)
        assert eng.stats() == (g.num_vertices, g.num_entries), name
        nb = eng.neighbors(2)
        onb = g.neighbors(2)
        assert nb[0].tolist() == onb[0].tolist(), name
        assert np.array_equal(np.asarray(nb[1], np.float32).view(np.uint32), np.asarray(onb[1], np.float32).view(np.uint32)), name
    bad_cases = {"leading_blank": " 1 2\n", "empty_line": "1 2\n\n3 4\n", "overflow": "1 2147483648\n", "garbage": "1 x\n",
                 "glued": "12-3 4\n", "lonely_sign": "- 4\n"}
    for name, text in bad_cases.items():
        f = tmp_path / (name + ".txt")
        f.write_text(text)
        with pytest.raises(pkg().SrwError) as ei:
            eng.load_edgelist(str(f), directed=False)
        assert ei.value.code == pkg().ERR_PARSE, name


def test_device_tokenizer_fuzz(eng, tmp_path, monkeypatch):
    # random files over an alphabet that mixes the fast shape with everything that must fall back or fail:
    # the default loader (device tokenizer first) and the host tokenizer alone agree on graph or error, file by file
    rng = np.random.default_rng(99)
    toks = ["1", "2", "33", "-4", "+5", "007", "2147483647", "-2147483648", "2147483648", "-", "+", "1-2", "x", "1.5", "", "9",
            "0.25", ".5", "5.", "1e3", "12345678", "0.00000000001", "-0", "1..2", "3.1415927"]
    seps = [" ", "\t", "  ", " \t"]
    ends = ["\n", "\n", "\n", "\r\n", " \n", "\n\n"]
    def outcome():
        try:
            eng.load_edgelist(str(f), directed=bool(k & 1))
            return ("ok",) + _graph_snapshot(eng)
        except pkg().SrwError as e:
            return ("err", e.code)
    for k in range(80):
        nl = int(rng.integers(1, 6))
        lines = []
        for _ in range(nl):
            nt = int(rng.choice([2, 2, 2, 2, 3, 1]))
            p_weird = 0.06
            parts = [str(rng.choice(toks)) if rng.random() < p_weird else str(int(rng.integers(-50, 60))) for _ in range(nt)]
            lead = " " if rng.random() < 0.03 else ""
            lines.append(lead + str(rng.choice(seps)).join(parts) + str(rng.choice(ends)))
        text = "".join(lines)
        if rng.random() < 0.3:
            text = text.rstrip("\n")
        f = tmp_path / ("f%d.txt" % k)
        f.write_text(text)
        monkeypatch.delenv("SRW_HOST_TOKENIZER", raising=False)
        a = outcome()
        monkeypatch.setenv("SRW_HOST_TOKENIZER", "1")
        b = outcome()
        assert a == b, (k, text)


def test_entry_points_refuse_while_population_1_is_selected():
    """srw_shard_select(h, 1) swaps the handle's stream, counters and cursors with the second population's; every entry point outside
    the super-step family refuses until population 0 is selected again, instead of running with the wrong context."""
    P = pkg()
    with P.Engine(device=0) as e:
        e.load_edgelist(KARATE, directed=False)
        ref = e.walk(walk_length=5, seed=1)
        e._ck(P.lib().srw_shard_select(e.h, 1))
        with pytest.raises(P.SrwError, match="population 1 is selected"):
            e.walk(walk_length=5, seed=1)
        with pytest.raises(P.SrwError, match="population 1 is selected"):
            e.load_edgelist(KARATE, directed=False)
        e._ck(P.lib().srw_shard_select(e.h, 0))
        again = e.walk(walk_length=5, seed=1)
        assert np.array_equal(again[0], ref[0]) and np.array_equal(again[1], ref[1])
