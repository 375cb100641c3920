"""Pins the CPU oracle against every known answer the reference's own tests hold for the hot path.

Each test names the reference test it restates (T/ = randomwalk/src/test/scala/au/csiro/data61/randomwalk/algorithm/).
Fixtures tests/golden/karate.txt and testgraph.txt are the reference's test resources (data, copied verbatim).
"""
import hashlib

import numpy as np
import pytest

from conftest import KARATE, TESTGRAPH


# --- T/RandomSampleTest.scala:9-24 "Test random sample function" ---------------------------------
@pytest.mark.parametrize("r,expect", [(0.1, 0), (0.4, 1), (0.7, 2)])
def test_random_sample_function(oracle, r, expect):
    assert oracle.sample_index([1.0, 1.0, 1.0], r) == expect


# --- T/RandomSampleTest.scala:26-94 "Test second order random selection" ---------------------------
CURR = ([1, 3, 4], [1.0, 1.0, 1.0])          # e21, e23, e24
PREV3 = [2, 4, 5]                             # e12, e14, e15
PREV2 = [2, 5]                                # e12, e15


def test_second_order_weights_p1_q1_unchanged(oracle):                       # :41-44
    w = oracle.second_order_weights(1.0, 1.0, 1, PREV3, *CURR)
    assert w.tolist() == [1.0, 1.0, 1.0]


@pytest.mark.parametrize("r,expect_id", [(0.1, 1), (0.4, 3), (0.7, 4)])       # :46-56
def test_second_order_sample_p1_q1(oracle, r, expect_id):
    k = oracle.second_order_sample_index(1.0, 1.0, 1, PREV3, *CURR, r)
    assert CURR[0][k] == expect_id


def test_second_order_weights_p2_q2(oracle):
    assert oracle.second_order_weights(2.0, 2.0, 1, PREV2, *CURR).tolist() == [0.5, 0.5, 0.5]   # :58-66
    assert oracle.second_order_weights(2.0, 2.0, 1, PREV3, *CURR).tolist() == [0.5, 0.5, 1.0]   # :68-75


@pytest.mark.parametrize("r,expect_id", [(0.24, 1), (0.26, 3), (0.51, 4), (0.99, 4)])  # :69-88
def test_second_order_sample_p2_q2(oracle, r, expect_id):
    k = oracle.second_order_sample_index(2.0, 2.0, 1, PREV3, *CURR, r)
    assert CURR[0][k] == expect_id


def test_second_order_inputs_not_mutated(oracle):                              # :90-93
    ids = np.array(CURR[0], dtype=np.int32)
    w = np.array(CURR[1], dtype=np.float32)
    oracle.second_order_sample_index(2.0, 2.0, 1, PREV3, ids, w, 0.5)
    assert w.tolist() == [1.0, 1.0, 1.0]


# --- T/GraphMapTest.scala:7-33 ------------------------------------------------------------------
def test_graphmap_data_structure(oracle):
    e1, e2, e3, e4 = [(2, 1.0)], [(3, 1.0)], [(3, 1.0)], [(1, 1.0)]
    g = oracle.GraphMap()
    g.add_vertex(1, e1)
    g.add_vertex(2)
    assert g.num_edges == 1 and g.num_vertices == 2
    assert g.get_neighbors(1) == e1
    g.reset()
    g.add_vertex(1, e1 + e2)
    g.add_vertex(2)
    g.add_vertex(3)
    assert g.get_neighbors(1) == e1 + e2
    g.reset()
    g.add_vertex(2, e3 + e4)
    g.add_vertex(1, e1 + e2)
    g.add_vertex(3)
    assert g.get_neighbors(1) == e1 + e2 and g.get_neighbors(2) == e3 + e4
    # GraphMap.scala:109-120: -1 row -> empty array, unknown -> null; :24-25,37 first add wins
    assert g.get_neighbors(3) == [] and g.get_neighbors(99) is None
    g.add_vertex(1, e4)
    assert g.get_neighbors(1) == e1 + e2
    # GraphMap.scala:28-32,66-68: dst -> pId, last put wins
    g.add_vertex(7, [(8, 3, 1.0), (9, 4, 2.0), (8, 5, 1.0)])
    assert g.get_partition(8) == 5 and g.get_partition(9) == 4 and g.get_partition(1) is None


# --- T/UniformRandomWalkTest.scala:33-67 load graph ----------------------------------------------
def test_load_graph_undirected(oracle):
    g = oracle.Graph.load(KARATE, directed=False)
    assert g.num_entries == 156 and g.num_vertices == 34


def test_load_graph_directed(oracle):
    g = oracle.Graph.load(KARATE, directed=True)
    assert g.num_entries == 78 and g.num_vertices == 34


def test_neighbor_order_is_file_order(oracle):
    # SURVEY §8c cross-check of the canonical (file-order) adjacency
    g = oracle.Graph.load(KARATE, directed=False)
    assert g.neighbors(1)[0].tolist() == [32, 22, 20, 18, 14, 13, 12, 11, 9, 8, 7, 6, 5, 4, 3, 2]
    assert g.neighbors(9)[0].tolist() == [1, 3, 34, 33, 33]
    assert g.neighbors(34)[0].tolist() == [9, 10, 14, 15, 16, 19, 20, 21, 23, 24, 27, 28, 29, 30, 31, 32, 33]


# --- T/UniformRandomWalkTest.scala:69-86 "the first step of Random Walk" --------------------------
def test_first_step_on_testgraph(oracle):
    g = oracle.Graph.load(TESTGRAPH, directed=True)
    assert g.num_vertices == 2
    paths, lens, _ = g.walk(walk_length=0, rng="const", const_r=0.3)
    got = {int(p[0]): p[:n].tolist() for p, n in zip(paths, lens)}
    assert got == {1: [1, 2], 2: [2]}


# --- T/UniformRandomWalkTest.scala:181-291 + 293-321: job walk == sequential test-oracle walk -----
CASES = [("undirected1", False, 1, 0.1), ("undirected2", False, 50, 0.1), ("undirected3", False, 50, 0.9),
         ("undirected4", False, 50, 0.1), ("directed1", True, 50, 0.9), ("directed2", True, 50, 0.1)]


@pytest.mark.parametrize("name,directed,L,r", CASES)
def test_second_order_random_walk_matches_sequential(oracle, name, directed, L, r):
    g = oracle.Graph.load(KARATE, directed=directed)
    for faithful in (True, False):
        paths, lens, steps = g.walk(walk_length=L, num_walks=1, rng="const", const_r=r, faithful=faithful)
        assert len(paths) == g.num_vertices                      # "a path per vertex"
        for p, n in zip(paths, lens):
            p2 = g.seq_walk(int(p[0]), walk_length=L, rng="const", const_r=r)
            assert p[:n].tolist() == p2.tolist()
        assert steps == int((lens - 1).sum())


def _digest(paths, lens):
    lines = sorted("\t".join(str(int(x)) for x in p[:n]) + "\n" for p, n in zip(paths, lens))
    return hashlib.sha256("".join(lines).encode()).hexdigest()[:16]


# --- SURVEY §8c derived goldens (independent throw-away restatement; file-order rule, constant r) --
DERIVED = [
    (False, 1, 0.1, 1.0, 1.0, {1: [1, 22, 1], 2: [2, 1, 22], 9: [9, 1, 22], 34: [34, 10, 3]}, 68, "25b2f0fffab1481e"),
    (False, 50, 0.1, 1.0, 1.0, {34: [34, 10, 3, 2, 1, 22, 1, 22]}, 1734, "7780f0ee2a73f2b9"),
    (False, 50, 0.9, 1.0, 1.0, {1: [1, 3, 8, 4, 8, 4], 9: [9, 33, 32, 33, 32]}, 1734, "fe0f7f858ba84ae3"),
    (False, 10, 0.5, 0.25, 4.0, {1: [1, 11, 1, 11], 9: [9, 34, 14, 34, 14]}, 374, "8b314af7cd74348c"),
    (False, 10, 0.3, 4.0, 0.5, {1: [1, 14, 3, 10, 34, 19, 33, 15, 34, 19, 33, 15],
                                2: [2, 22, 1, 13, 4, 2, 22, 1, 13, 4, 2, 22]}, 374, "b42487837425aa3a"),
    (True, 50, 0.1, 1.0, 1.0, {1: [1, 22], 2: [2, 31, 34], 34: [34]}, 32, "c21670d58f359860"),
    (True, 50, 0.9, 1.0, 1.0, {1: [1, 3, 4, 8], 9: [9, 33, 34]}, 54, "bdef19b134fd9227"),
]


@pytest.mark.parametrize("directed,L,r,p,q,prefixes,steps,digest", DERIVED)
def test_survey_derived_goldens(oracle, directed, L, r, p, q, prefixes, steps, digest):
    g = oracle.Graph.load(KARATE, directed=directed)
    paths, lens, nsteps = g.walk(walk_length=L, p=p, q=q, rng="const", const_r=r, faithful=True)
    got = {int(pp[0]): pp[:n].tolist() for pp, n in zip(paths, lens)}
    for v, pre in prefixes.items():
        assert got[v][:len(pre)] == pre
    assert nsteps == steps
    assert _digest(paths, lens) == digest


# --- rounding known-answers: the sequential f64 sum decides lattice ties -------------
@pytest.mark.parametrize("deg,r,expect_index", [(12, 0.5, 6), (14, 0.5, 7), (6, 0.5, 2)])
def test_rounding_known_answers(oracle, deg, r, expect_index):
    assert oracle.sample_index([1.0] * deg, r) == expect_index


def test_sample_fallback_head(oracle):
    # RandomSample.scala:24: no acc >= p (sum == 0 -> NaN) falls back to edges.head
    assert oracle.sample_index([0.0, 0.0, 0.0], 0.5) == 0
    assert oracle.sample_index([1.0, 1.0], 1.5) == 0


# --- RNG known answers -------------------------------------------------------
def test_philox_kat(oracle):
    assert oracle.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert oracle.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert oracle.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_walk_stream_kat(oracle):
    assert oracle.walk_uniform(42, 0, 1, 1) == 0.2174382209777832
    assert oracle.walk_uniform(42, 0, 1, 2) == 0.21827632188796997
    assert oracle.walk_uniform(42, 0, 1, 3) == 0.9992966651916504


def test_java_random_kat(oracle):
    assert oracle.java_random_floats(42, 3).tolist() == [np.float32(0.7275636792182922),
                                                         np.float32(0.054665207862854004),
                                                         np.float32(0.6832234263420105)]
    assert oracle.java_random_floats(0, 3).tolist() == [np.float32(0.7309677600860596),
                                                        np.float32(0.8314409852027893),
                                                        np.float32(0.2405363917350769)]
