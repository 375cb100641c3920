"""Sparse vertex-id spaces.  The reference keys vertices in a HashMap (M/algorithm/GraphMap.scala:13-15) and takes any
int32 ids; the engine compacts a sparse id space at load (slot = rank among the sorted distinct ids,
graph_build.hip:compact_ids) and translates at the boundary.  CPU part: the oracle's own rank-indexed fast graph against
its faithful HashMap-shaped variant.  GPU part: the HIP path against the oracle through the C ABI, bit for bit."""
import os

import numpy as np
import pytest

from helpers import pkg, random_multigraph

I32_MIN, I32_MAX = -2147483648, 2147483647


def sparse_multigraph(seed, n_vertices=70, n_lines=500, weighted=True):
    """random_multigraph over ids scattered across the whole int32 range (both extremes included)."""
    rng = np.random.default_rng(seed)
    s, d, w = random_multigraph(rng, n_vertices, n_lines, weighted)
    ids = np.unique(np.concatenate([rng.integers(I32_MIN, I32_MAX, size=n_vertices - 4, dtype=np.int64),
                                    np.array([I32_MIN, I32_MAX, -1, 0], dtype=np.int64)]))
    while ids.size < n_vertices:
        ids = np.unique(np.concatenate([ids, rng.integers(I32_MIN, I32_MAX, size=4, dtype=np.int64)]))
    ids = rng.permutation(ids).astype(np.int32)      # NOT monotone in the dense ids: the order of the ids is new too
    return ids[s], ids[d], w


@pytest.mark.parametrize("seed,weighted", [(1, True), (2, False)])
def test_oracle_rank_index_equals_faithful_hashmap_variant(oracle, seed, weighted):
    s, d, w = sparse_multigraph(seed, weighted=weighted)
    g = oracle.Graph.from_coo(s, d, w)
    assert g is not None and g.num_vertices == np.unique(np.concatenate([s, d])).size
    assert g.vertices().tolist() == sorted(set(s.tolist()) | set(d.tolist()))
    for p, q in [(1.0, 1.0), (0.25, 4.0), (4.0, 0.5)]:
        fast = g.walk(p=p, q=q, walk_length=12, seed=7)
        slow = g.walk(p=p, q=q, walk_length=12, seed=7, faithful=True)
        assert np.array_equal(fast[0], slow[0]) and np.array_equal(fast[1], slow[1]) and fast[2] == slow[2]
    v = int(g.vertices()[0])
    assert v == I32_MIN and g.seq_walk(v, walk_length=12, seed=7, p=0.25, q=4.0).tolist() == \
        g.walk(p=0.25, q=4.0, walk_length=12, seed=7)[0][0][:g.walk(p=0.25, q=4.0, walk_length=12, seed=7)[1][0]].tolist()


# ---- GPU -------------------------------------------------------------------------------------------------
CONFIGS = [dict(p=1.0, q=1.0), dict(p=0.25, q=4.0), dict(p=4.0, q=0.5), dict(p=0.5, q=1.0), dict(p=2.0, q=1.0, rng="const", const_r=0.37),
           dict(p=1.0, q=1.0, force_general=True)]


def walk_kw(kw):
    return {k: v for k, v in kw.items() if k != "force_general"}


@pytest.mark.gpu
@pytest.mark.parametrize("seed,weighted", [(1, True), (2, False), (3, True)])
def test_sparse_ids_walks_equal_oracle(oracle, seed, weighted):
    """ids from the whole int32 range (INT_MIN and INT_MAX among them): loaded without a flag, every kernel family."""
    P = pkg()
    s, d, w = sparse_multigraph(seed, weighted=weighted)
    g = oracle.Graph.from_coo(s, d, w)
    with P.Engine(device=0) as eng:
        eng.load_coo(s, d, w)
        assert eng.stats() == (g.num_vertices, g.num_entries)
        assert eng.vertices().tolist() == g.vertices().tolist()
        for v in g.vertices().tolist():
            (gi, gw), (oi, ow) = eng.neighbors(v), g.neighbors(v)
            assert np.array_equal(gi, oi) and np.array_equal(gw, ow)
        for v in (5, 123456789, I32_MAX - 1):
            if v not in set(g.vertices().tolist()):
                assert eng.neighbors(v) is None
        for kw in CONFIGS:
            for nw in (1, 2):
                paths, lens, st = eng.walk(walk_length=15, num_walks=nw, seed=11, **kw)
                rp, rl, rs = g.walk(walk_length=15, num_walks=nw, seed=11, **walk_kw(kw))
                assert np.array_equal(lens, rl) and np.array_equal(paths, rp), kw
                assert st["n_steps"] == rs
        # Mode A (alias + rejection) against the oracle's Mode A
        for kw in (dict(p=1.0, q=1.0), dict(p=0.25, q=4.0)):
            paths, lens, _ = eng.walk(walk_length=15, seed=11, sampler="alias", **kw)
            rp, rl, _ = g.walk(walk_length=15, seed=11, sampler=1, **kw)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp), kw


@pytest.mark.gpu
def test_forced_compaction_changes_nothing(oracle):
    """SRW_CFG_COMPACT_IDS on a dense graph: same vertices, same neighbors, same paths as the dense index (the Philox
    stream stays keyed by the input's ids), for every kernel family, at a size where the accelerators fire."""
    P = pkg()
    with P.Engine(device=0) as dense, P.Engine(device=0, compact_ids=True) as comp:
        n = 16 << 14
        s, d = oracle.rmat_edges(14, n, seed=5)
        from helpers import rmat_weights_np
        w = rmat_weights_np(s, d, 5)
        s = (s.astype(np.int64) * 3 - 20000).astype(np.int32)     # monotone but not the identity, negative ids included
        d = (d.astype(np.int64) * 3 - 20000).astype(np.int32)
        dense.load_coo(s, d, w); comp.load_coo(s, d, w)
        assert dense.stats() == comp.stats()
        assert np.array_equal(dense.vertices(), comp.vertices())
        for kw in CONFIGS + [dict(p=0.25, q=4.0, sampler="alias")]:
            a = dense.walk(walk_length=30, seed=3, **kw)
            b = comp.walk(walk_length=30, seed=3, **kw)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0]), kw
            assert a[2]["n_steps"] == b[2]["n_steps"]


@pytest.mark.gpu
@pytest.mark.parametrize("host_tokenizer", [False, True])
def test_sparse_ids_edge_list_file_to_part_files(oracle, tmp_path, host_tokenizer, monkeypatch):
    """Text in, text out: both tokenizers, walk_and_save with the device formatter and the host formatter."""
    P = pkg()
    s, d, w = sparse_multigraph(9, weighted=False)
    f = tmp_path / "edges.txt"
    f.write_text("".join("%d %d\n" % (a, b) for a, b in zip(s, d)))
    if host_tokenizer:
        monkeypatch.setenv("SRW_HOST_TOKENIZER", "1")
    g = oracle.Graph.load(str(f), weighted=False)
    rp, rl, _ = g.walk(walk_length=10, num_walks=2, seed=5, p=0.5, q=2.0)
    expect = sorted("\t".join(str(int(x)) for x in p[:n]) for p, n in zip(rp, rl))
    with P.Engine(device=0) as eng:
        eng.load_edgelist(str(f), weighted=False)
        for k, fmt in enumerate(("device", "host")):
            out = tmp_path / ("out%d" % k)
            eng.walk_and_save(str(out), walk_length=10, num_walks=2, seed=5, p=0.5, q=2.0, device_format=(fmt == "device"))
            lines = []
            for name in sorted(os.listdir(out / "path")):
                if name.startswith("part-"):
                    lines += (out / "path" / name).read_text().splitlines()
            assert sorted(lines) == expect, fmt
        # the resident result + srw_write_paths
        eng.walk(fetch=False, walk_length=10, num_walks=2, seed=5, p=0.5, q=2.0)
        eng.write_paths(str(tmp_path / "out_w"))
        lines = []
        for name in sorted(os.listdir(tmp_path / "out_w" / "path")):
            if name.startswith("part-"):
                lines += (tmp_path / "out_w" / "path" / name).read_text().splitlines()
        assert sorted(lines) == expect


@pytest.mark.gpu
def test_sparse_ids_partitioned_load_and_adjacency_surface(oracle):
    """VCut partition ids and the GraphMap surface (srw_load_adjacency) over sparse ids."""
    P = pkg()
    s, d, w = sparse_multigraph(4, n_vertices=30, n_lines=120, weighted=True)
    pid = (np.arange(len(s)) % 7).astype(np.int32)
    with P.Engine(device=0) as eng:
        eng.load_coo(s, d, w, pid=pid)
        last = {}
        for a, b, p_ in zip(s.tolist(), d.tolist(), pid.tolist()):      # vertexPartitionMap.put(dst, pId), last wins; both ends (undirected)
            last[b] = p_; last[a] = p_
        for v, p_ in last.items():
            assert eng.partition(v) == p_
        assert eng.partition(12345) is None
        # GraphMap surface: rows handed over one by one, a neighbor without a row of its own among them
        rows = [(I32_MAX, [(I32_MIN, 1.0), (7, 2.0)]), (I32_MIN, [(I32_MAX, 1.0)]), (1000000007, [(I32_MAX, 0.5), (I32_MIN, 0.25)])]
        eng.load_adjacency(rows)
        gm = oracle.GraphMap()
        for v, nb in rows:
            gm.add_vertex(v, nb)
        assert eng.num_vertices == gm.num_vertices
        for v in (I32_MAX, I32_MIN, 1000000007):
            gi, gw = eng.neighbors(v)
            oi = gm.get_neighbors(v)
            assert [(int(a), float(b)) for a, b in zip(gi, gw)] == [(int(a), float(b)) for a, b in oi]
        assert eng.neighbors(7) is None or len(eng.neighbors(7)[0]) == 0
        paths, lens, _ = eng.walk(walk_length=6, seed=2, p=0.5, q=2.0)
        assert set(paths[:, 0].tolist()) == {I32_MAX, I32_MIN, 1000000007}
        assert all(int(x) in (I32_MAX, I32_MIN, 1000000007, 7) for p_, n in zip(paths, lens) for x in p_[:n])


@pytest.mark.gpu
@pytest.mark.parametrize("world,p,q,directed", [(2, 1.0, 1.0, False), (3, 0.25, 4.0, False), (4, 4.0, 0.5, True), (2, 0.5, 1.0, False)])
def test_sparse_ids_vertex_sharded_cluster(oracle, tmp_path, world, p, q, directed):
    """Vertex-sharded handles compact the same way (every shard sees the whole edge list and computes the same ranks;
    owner(v) is taken over the rank): the in-process cluster equals the oracle for every batching, and its part files too."""
    P = pkg()
    s, d, w = sparse_multigraph(20 + world, n_vertices=90, n_lines=700, weighted=True)
    g = oracle.Graph.from_coo(s, d, w, directed=directed)
    with P.Cluster([0] * world) as cl:
        cl.load_coo(s, d, w, directed=directed)
        assert cl.stats() == (g.num_vertices, g.num_entries)
        rp, rl, rs = g.walk(p=p, q=q, walk_length=11, num_walks=3, first_walk=2, seed=9)
        for batch in (0, 1, 2):
            paths, lens, st = cl.walk(p=p, q=q, walk_length=11, num_walks=3, first_walk=2, seed=9, batch=batch)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp), (world, batch)
            assert st["n_steps"] == rs
        out = tmp_path / "out"
        cl.walk_and_save(str(out), n_parts=2, p=p, q=q, walk_length=11, num_walks=3, first_walk=2, seed=9)
        ref = tmp_path / "ref"
        oracle.write_paths(rp, rl, str(ref), n_parts=2)
        for f in sorted(os.listdir(ref / "path")):
            assert (out / "path" / f).read_bytes() == (ref / "path" / f).read_bytes(), f


@pytest.mark.gpu
def test_sparse_ids_sharded_by_user_partitions(oracle):
    """SRW_CFG_OWNER_FROM_PARTITIONS over compacted ids: the owner table is indexed by rank."""
    P = pkg()
    s, d, w = sparse_multigraph(31, n_vertices=60, n_lines=400, weighted=False)
    pid = (np.abs(d.astype(np.int64)) % 5).astype(np.int32)
    g = oracle.Graph.from_coo(s, d, w)
    with P.Cluster([0, 0, 0], owner_from_partitions=True) as cl:
        cl.load_coo(s, d, w, pid=pid)
        rp, rl, rs = g.walk(p=0.5, q=2.0, walk_length=8, num_walks=2, seed=4)
        paths, lens, st = cl.walk(p=0.5, q=2.0, walk_length=8, num_walks=2, seed=4)
        assert np.array_equal(lens, rl) and np.array_equal(paths, rp) and st["n_steps"] == rs
