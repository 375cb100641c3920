"""Property tests (hypothesis) on random small multigraphs — self-loops, duplicate lines, zero-degree vertices (directed
sinks), ids that do not occur, negative and sparse ids, weights from exact small integers to 2^±30 — as SURVEY.md §8c
asks.  CPU part: properties of the oracle itself (fast variant == faithful HashMap-shaped variant == the sequential walk
of T/UniformRandomWalkTest.scala:293-321; writer -> text -> tokenizer round trip through the C ABI's host entry points).
GPU part: the HIP path through the C ABI == the oracle, bit for bit, for the (p, q) set of §8c."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from helpers import pkg

PQ = [(1.0, 1.0), (0.25, 4.0), (4.0, 0.5), (2.0, 2.0), (0.5, 1.0)]
WEIGHTS = [0.25, 0.5, 1.0, 1.0, 2.0, 3.0, 7.0, 16.0, 1000.0, 0.001, float(2.0 ** -30), float(2.0 ** 30)]


@st.composite
def multigraphs(draw, max_vertices=24, max_lines=120):
    """(src, dst, w or None, directed).  The id universe is dense, shifted negative, or scattered over all of int32."""
    nv = draw(st.integers(1, max_vertices))
    space = draw(st.sampled_from(["dense", "negative", "sparse"]))
    if space == "sparse":
        ids = draw(st.lists(st.integers(-2147483648, 2147483647), min_size=nv, max_size=nv, unique=True))
    else:
        lo = 0 if space == "dense" else -draw(st.integers(1, 40))
        ids = [lo + 2 * k for k in range(nv)]          # every other id: the ids in between do not occur
    nl = draw(st.integers(1, max_lines))
    pick = st.integers(0, nv - 1)
    lines = draw(st.lists(st.tuples(pick, pick), min_size=nl, max_size=nl))
    # self-loops and duplicate lines on purpose
    if draw(st.booleans()):
        lines += [(a, a) for a, _ in lines[:3]]
    if draw(st.booleans()):
        lines += lines[:5]
    s = np.array([ids[a] for a, _ in lines], np.int32)
    d = np.array([ids[b] for _, b in lines], np.int32)
    w = None
    if draw(st.booleans()):
        w = np.array(draw(st.lists(st.sampled_from(WEIGHTS), min_size=len(lines), max_size=len(lines))), np.float32)
    return s, d, w, draw(st.booleans())


@st.composite
def walk_args(draw):
    p, q = draw(st.sampled_from(PQ))
    kw = dict(p=p, q=q, walk_length=draw(st.integers(0, 20)), num_walks=draw(st.integers(1, 3)), seed=draw(st.integers(0, 2 ** 31 - 1)))
    if draw(st.integers(0, 3)) == 0:        # the reference's own determinism hook: nextFloat = () => r
        kw.update(rng="const", const_r=draw(st.sampled_from([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99999994])))
    return kw


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(g=multigraphs(), kw=walk_args())
def test_oracle_variants_agree(oracle, g, kw):
    s, d, w, directed = g
    og = oracle.Graph.from_coo(s, d, w, directed=directed)
    fast = og.walk(**kw)
    slow = og.walk(faithful=True, **kw)
    assert np.array_equal(fast[0], slow[0]) and np.array_equal(fast[1], slow[1]) and fast[2] == slow[2]
    verts = og.vertices()
    assert verts.tolist() == sorted(set(s.tolist()) | set(d.tolist()))
    # a path never leaves the graph, starts at its source, is as long as it says, and only stops early at a dead end
    L, nv = kw["walk_length"], len(verts)
    for i, (path, n) in enumerate(zip(fast[0], fast[1])):
        assert path[0] == verts[i % nv] and 1 <= n <= L + 2 and (path[n:] == -1).all()
        if n < L + 2:
            assert og.degree(int(path[n - 1])) == 0
        for a, b in zip(path[:n - 1], path[1:n]):
            assert int(b) in og.neighbors(int(a))[0].tolist()
    # the reference's own ground truth: the sequential walk of one source
    if kw["num_walks"] == 1 and nv:
        v = int(verts[len(verts) // 2])
        i = len(verts) // 2
        assert og.seq_walk(v, **kw).tolist() == fast[0][i][:fast[1][i]].tolist()
    assert fast[2] == int((fast[1] - 1).sum())


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(g=multigraphs(), kw=walk_args(), n_parts=st.integers(1, 4))
def test_writer_text_round_trip_through_the_c_abi(oracle, tmp_path_factory, g, kw, n_parts):
    """paths -> srw_save_paths (part files) -> text -> the same lines the oracle's writer prints; the edge list written as
    text and read back by srw_parse_edgelist gives the same lines (host entry points of the C ABI: no GPU involved)."""
    P = pkg()
    s, d, w, directed = g
    og = oracle.Graph.from_coo(s, d, w, directed=directed)
    paths, lens, _ = og.walk(**kw)
    out = tmp_path_factory.mktemp("rt")
    P.save_paths(paths, lens, str(out / "a"), n_parts=n_parts)
    oracle.write_paths(paths, lens, str(out / "b"), n_parts=n_parts)
    for name in sorted(os.listdir(out / "b" / "path")):
        if name.startswith("part-") or name == "_SUCCESS":
            assert (out / "a" / "path" / name).read_bytes() == (out / "b" / "path" / name).read_bytes(), name
    f = out / "edges.txt"
    f.write_text("".join(("%d\t%d\t%r\n" % (a, b, float(x))) for a, b, x in zip(s, d, w if w is not None else np.ones(len(s)))))
    ps, pd, pw, _ = P.parse_edgelist(str(f), weighted=True)
    assert np.array_equal(ps, s) and np.array_equal(pd, d)
    assert np.array_equal(pw, w if w is not None else np.ones(len(s), np.float32))


@pytest.fixture(scope="module")
def eng():
    e = pkg().Engine(device=0)
    yield e
    e.close()


@pytest.mark.gpu
@settings(max_examples=120, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(g=multigraphs(max_vertices=40, max_lines=400), kw=walk_args(), general=st.booleans())
def test_gpu_equals_oracle(oracle, eng, g, kw, general):
    s, d, w, directed = g
    og = oracle.Graph.from_coo(s, d, w, directed=directed)
    eng.load_coo(s, d, w, directed=directed)
    assert eng.stats() == (og.num_vertices, og.num_entries)
    assert np.array_equal(eng.vertices(), og.vertices())
    paths, lens, stt = eng.walk(force_general=general, **kw)
    rp, rl, rs = og.walk(**kw)
    assert np.array_equal(lens, rl) and np.array_equal(paths, rp)
    assert stt["n_steps"] == rs


@pytest.mark.gpu
@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(g=multigraphs(max_vertices=40, max_lines=400), kw=walk_args())
def test_gpu_mode_a_equals_oracle_mode_a(oracle, eng, g, kw):
    if kw.get("rng") == "const":
        kw = {k: v for k, v in kw.items() if k not in ("rng", "const_r")}     # Mode A draws from Philox only
    s, d, w, directed = g
    og = oracle.Graph.from_coo(s, d, w, directed=directed)
    eng.load_coo(s, d, w, directed=directed)
    paths, lens, _ = eng.walk(sampler="alias", **kw)
    rp, rl, _ = og.walk(sampler=1, **kw)
    assert np.array_equal(lens, rl) and np.array_equal(paths, rp)
