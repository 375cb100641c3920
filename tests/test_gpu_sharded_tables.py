"""The bit-exact second-order walk on VERTEX SHARDS with the per-edge bias tables (north_star's biased multi-GPU config:
directed, p = 4, q = .5, graph sharded by source vertex — replacing RandomWalk.scala:92-139, whose shuffle ships N(prev)
with every walker so that the receiving partition can recompute RandomSample.computeSecondOrderWeights, :27-44).
Each shard holds the tables of the pairs (prev -> curr) into ITS rows, found through a pair hash; the walk must stay
bit-identical to the oracle for every world size, and the table / mask steps must actually fire on the shards."""
import os

import numpy as np
import pytest

from helpers import pkg, rmat_lines, rmat_weights_np

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("p,q,directed", [(0.25, 4.0, False), (4.0, 0.5, True), (2.0, 2.0, False)])
def test_shard_tables_equal_oracle(oracle, world, p, q, directed):
    s, d, w = rmat_lines(oracle, 11, edge_factor=16, weighted=True)
    g = oracle.Graph.from_coo(s, d, w, directed=directed)
    rp, rl, rs = g.walk(p=p, q=q, walk_length=12, num_walks=2, first_walk=1, seed=5, threads=8)
    with pkg().Cluster([0] * world) as cl:
        cl.load_coo(s, d, w, directed=directed)
        assert cl.stats() == (g.num_vertices, g.num_entries)
        # default selection (masks for rows < 256 candidates, chunk-prefix tables beyond), every table pair with chunks of
        # 4 candidates, hub bitmaps on rows of any degree, and the on-the-fly samplers alone: the same paths
        for kw in (dict(), dict(edge_tables_all=True), dict(binned_tune=8), dict(edge_tables=False)):
            for batch in (0, 1):
                paths, lens, st = cl.walk(p=p, q=q, walk_length=12, num_walks=2, first_walk=1, seed=5, batch=batch, **kw)
                assert np.array_equal(lens, rl) and np.array_equal(paths, rp), (world, kw, batch)
                assert st["n_steps"] == rs
            ss = st["strategy_steps"]
            if kw.get("edge_tables", True):
                assert st["edge_tables"] > 0 and ss["edge_table"] > 0, (kw, st)
                if not kw.get("edge_tables_all"):
                    assert ss["edge_mask"] > 0 and ss["handed_over_walkers"] == 0, (kw, st)   # a complete set: nothing left to the general step
                    assert ss["p1"] + ss["p2"] + ss["w"] + ss["p3"] == 0, st
            else:
                assert st["edge_tables"] == 0 and ss["edge_table"] == 0 and ss["edge_mask"] == 0, st
        # constant r (the reference tests' injection) lands on CDF boundaries: the exact chain behind the tables
        rp2, rl2, rs2 = g.walk(p=p, q=q, walk_length=6, rng="const", const_r=0.5, threads=8)
        paths, lens, st = cl.walk(p=p, q=q, walk_length=6, rng="const", const_r=0.5)
        assert np.array_equal(lens, rl2) and np.array_equal(paths, rp2) and st["n_steps"] == rs2


def test_shard_tables_new_pq_rebuilds(oracle):
    """The standing tables belong to one (p, q): another pair rebuilds them, q = 1 drops them, and a handle walked whole
    (srw_walk, tables keyed by entry) and sharded (srw_shard_*, tables keyed by the pair hash) in turn stays exact."""
    s, d, w = rmat_lines(oracle, 10, edge_factor=16, weighted=True)
    g = oracle.Graph.from_coo(s, d, w)
    with pkg().Cluster([0]) as cl:
        cl.load_coo(s, d, w)
        eng = cl.shard(0)
        for p, q in [(0.25, 4.0), (4.0, 0.5), (0.5, 1.0), (0.25, 4.0), (1.0, 1.0)]:
            rp, rl, rs = g.walk(p=p, q=q, walk_length=10, seed=3, threads=8)
            paths, lens, st = cl.walk(p=p, q=q, walk_length=10, seed=3)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp), (p, q)
            assert (st["edge_tables"] > 0) == (q != 1.0)
            paths, lens, st = eng.walk(p=p, q=q, walk_length=10, seed=3)        # the same handle through srw_walk
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp), (p, q)


@pytest.mark.parametrize("world,scale,ef,weighted,directed,p,q", [(4, 16, 16, True, False, 0.25, 4.0), (3, 18, 27, False, True, 4.0, 0.5)])
def test_shard_tables_at_scale_equal_replicated(world, scale, ef, weighted, directed, p, q):
    """The shapes of tests/test_gpu_scale.py (which pins the replicated kernels to the oracle at these sizes) through the
    sharded protocol: every walker of the graph, default strategy selection, bit-identical; table and mask steps on the shards."""
    P = pkg()
    with P.Engine(device=0) as eng:
        eng.generate_rmat(scale, ef << scale, seed=7, weighted=weighted, directed=directed)
        paths, lens, st = eng.walk(p=p, q=q, walk_length=24, seed=99)
        nv = eng.stats()
    with P.Cluster([0] * world) as cl:
        cl.generate_rmat(scale, ef << scale, seed=7, weighted=weighted, directed=directed)
        assert cl.stats() == nv
        cp, clens, cst = cl.walk(p=p, q=q, walk_length=24, seed=99)
        assert np.array_equal(clens, lens)
        bad = np.nonzero((cp != paths).any(axis=1))[0]
        assert bad.size == 0, (bad[:5], cp[bad[0]], paths[bad[0]])
        assert cst["n_steps"] == st["n_steps"]
        ss = cst["strategy_steps"]
        assert cst["edge_tables"] > 0 and ss["edge_table"] > 0 and ss["edge_mask"] > 0, cst
        assert ss["edge_table"] + ss["edge_mask"] + ss["scan"] >= 0.999 * cst["n_steps"], cst     # (scan: the first steps)


@pytest.mark.parametrize("world,scale,ef,weighted,directed,p,q", [(1, 15, 16, True, False, 0.25, 4.0), (2, 15, 27, False, True, 4.0, 0.5)])
def test_table_step_variants_give_the_same_paths(monkeypatch, world, scale, ef, weighted, directed, p, q):
    """k_sh_step_tab serves a grab of records at once (lane l: record r0 + l, its Philox draw, its pair-hash probe, its rows; the
    sampled records stored once per grab).  The variants behind SRW_SH_BATCH (0: one record at a time, 1, 2 = default) and every grab
    size — the kernel takes fewer records per grab when a super-step has under 4 grabs per wave, so a small grid (SRW_SH_BLOCKS=1)
    makes grabs of up to 64 real at this size — give the paths of the replicated kernel, ties through the chain kernels included."""
    P = pkg()
    kw = dict(p=p, q=q, walk_length=10, num_walks=4, seed=11)
    with P.Engine(device=0) as eng:
        eng.generate_rmat(scale, ef << scale, seed=3, weighted=weighted, directed=directed)
        paths, lens, st = eng.walk(**kw)
        cpaths, clens, cst = eng.walk(rng="const", const_r=0.5, **dict(kw, num_walks=1))
    with P.Cluster([0] * world) as cl:
        cl.generate_rmat(scale, ef << scale, seed=3, weighted=weighted, directed=directed)
        for blocks, batch, grab in [("", "", ""), ("1", "0", "16"), ("1", "1", "16"), ("1", "2", "5"), ("1", "2", "32"), ("1", "2", "64"), ("2", "2", "33")]:
            for k, v in (("SRW_SH_BLOCKS", blocks), ("SRW_SH_BATCH", batch), ("SRW_SH_GRAB", grab)):
                if v:
                    monkeypatch.setenv(k, v)
                else:
                    monkeypatch.delenv(k, raising=False)
            sp, sl, sst = cl.walk(batch=4, **kw)               # one population of 4 iterations: 4 x 2^15 records per super-step
            assert np.array_equal(sl, lens), (blocks, batch, grab)
            bad = np.nonzero((sp != paths).any(axis=1))[0]
            assert bad.size == 0, (blocks, batch, grab, bad[:5], sp[bad[0]], paths[bad[0]])
            assert sst["n_steps"] == st["n_steps"]
            ss = sst["strategy_steps"]
            assert ss["edge_table"] > 0 and ss["edge_mask"] > 0, sst
            sp, sl, sst = cl.walk(rng="const", const_r=0.5, **dict(kw, num_walks=1))      # CDF boundaries: ties
            assert np.array_equal(sl, clens) and np.array_equal(sp, cpaths), (blocks, batch, grab)


@pytest.mark.parametrize("p,q", [(1.0, 1.0), (0.5, 2.0)])
def test_all_vertices_in_one_partition(oracle, tmp_path, p, q):
    """SRW_CFG_OWNER_FROM_PARTITIONS with fewer partitions than shards: one shard owns every vertex, its seeds exceed a chunk
    sized for an even split (nV / world^2).  The surplus must raise the overflow flag (retry with more slack), never be
    written past the chunk or silently dropped."""
    s, d, w = rmat_lines(oracle, 10, edge_factor=8, weighted=True)
    f = tmp_path / "vcut.txt"
    f.write_text("".join("%d %d %d %g\n" % (a, b, 2, x) for a, b, x in zip(s, d, w)))      # every edge in partition 2
    g = oracle.Graph.load(str(f), partitioned=True)
    with pkg().Cluster([0] * 4, owner_from_partitions=True) as cl:
        cl.load_edgelist(str(f), partitioned=True)
        assert [len(cl.shard(r).vertices()) for r in range(4)] == [0, 0, g.num_vertices, 0]
        rp, rl, rs = g.walk(p=p, q=q, walk_length=10, num_walks=3, seed=12, threads=8)
        for batch in (0, 1, 3):
            paths, lens, st = cl.walk(p=p, q=q, walk_length=10, num_walks=3, seed=12, batch=batch)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp) and st["n_steps"] == rs, batch


@pytest.mark.parametrize("world,directed", [(2, False), (3, True)])
def test_shard_build_in_blocks(oracle, monkeypatch, world, directed):
    """A shard builds its rows from the line stream in blocks and keeps only what it owns (graph_build.hip:
    build_graph_blocked — the reference's partitionBy, UniformRandomWalk.scala:41-42): blocks of 777 lines must give the rows,
    in input-line order, that the whole-stream build gives, from host lines (load_coo) and from the device generator."""
    monkeypatch.setenv("SRW_BUILD_BLOCK_LINES", "777")
    s, d, w = rmat_lines(oracle, 10, edge_factor=8, weighted=True)
    g = oracle.Graph.from_coo(s, d, w, directed=directed)
    P = pkg()
    with P.Cluster([0] * world) as cl:
        cl.load_coo(s, d, w, directed=directed)
        assert cl.stats() == (g.num_vertices, g.num_entries)
        owned = 0
        for r in range(world):
            e = cl.shard(r)
            for v in e.vertices()[::7]:
                ids, ws = e.neighbors(int(v))
                oi, ow = g.neighbors(int(v))
                assert np.array_equal(ids, oi) and np.array_equal(ws, ow), (r, int(v))
            owned += len(e.vertices())
        assert owned == g.num_vertices
        for p, q in ((1.0, 1.0), (0.25, 4.0)):
            rp, rl, rs = g.walk(p=p, q=q, walk_length=9, num_walks=2, seed=4, threads=8)
            paths, lens, st = cl.walk(p=p, q=q, walk_length=9, num_walks=2, seed=4)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp) and st["n_steps"] == rs
        # the device generator, block by block, against the whole-stream build of the same (seed, edge index) stream
        cl.generate_rmat(12, 16 << 12, seed=9, weighted=True, directed=directed)
        a = cl.walk(p=0.5, q=2.0, walk_length=7, seed=1)
    monkeypatch.setenv("SRW_BUILD_WHOLE", "1")
    with P.Cluster([0] * world) as cl:
        cl.generate_rmat(12, 16 << 12, seed=9, weighted=True, directed=directed)
        b = cl.walk(p=0.5, q=2.0, walk_length=7, seed=1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2]["n_steps"] == b[2]["n_steps"]


@pytest.mark.parametrize("world,directed,membership,max_ret", [(1, False, True, None), (2, False, False, 1), (3, True, True, 0), (8, False, False, None)])
def test_shard_q1_per_lane_step(oracle, monkeypatch, world, directed, membership, max_ret):
    """p != 1 with q == 1 on shards: one record per lane (k_sh_step_q1) over the shard's compact records, exact prefix sums and
    the return-edge hash of the pairs into its rows — also on handles without the replicated membership structure, which is what
    q == 1 jobs create.  Multigraph with parallel return edges, weighted RMAT; bit-identical to the oracle.  max_ret: pairs with
    more parallel return edges than this go one wave per record (k_sh_step_q1w; default 16) — 0 / 1 send most second-order steps there."""
    from helpers import random_multigraph
    if max_ret is not None:
        monkeypatch.setenv("SRW_Q1_MAX_RET", str(max_ret))
    s, d, w = rmat_lines(oracle, 11, edge_factor=16, weighted=True)
    ms, md, mw = random_multigraph(np.random.default_rng(7), 300, 6000, True, id_lo=5000)     # duplicates, self-loops, unused ids
    s, d, w = np.concatenate([s, ms]), np.concatenate([d, md]), np.concatenate([w, mw])
    g = oracle.Graph.from_coo(s, d, w, directed=directed)
    with pkg().Cluster([0] * world, membership=membership) as cl:
        cl.load_coo(s, d, w, directed=directed)
        for p in (0.5, 4.0, 0.25):
            rp, rl, rs = g.walk(p=p, q=1.0, walk_length=14, num_walks=2, first_walk=1, seed=9, threads=8)
            for batch in (0, 1):
                paths, lens, st = cl.walk(p=p, q=1.0, walk_length=14, num_walks=2, first_walk=1, seed=9, batch=batch)
                assert np.array_equal(lens, rl) and np.array_equal(paths, rp), (world, p, batch)
                assert st["n_steps"] == rs
            ss = st["strategy_steps"]
            # per lane, or per wave for the pairs with many parallel return edges ("prefix"); the rest: irregular rows, handed over
            assert ss["q1_lane"] + ss["prefix"] > 0.95 * st["n_steps"], (p, ss)
            assert (ss["q1_lane"] > 0.5 * st["n_steps"]) if max_ret is None else (ss["prefix"] > 0), (p, ss)
            assert ss["handed_over_walkers"] < 0.05 * st["n_steps"], (p, ss)
        rp, rl, rs = g.walk(p=0.5, q=1.0, walk_length=5, rng="const", const_r=0.5, threads=8)      # constant r: the general step
        paths, lens, st = cl.walk(p=0.5, q=1.0, walk_length=5, rng="const", const_r=0.5)
        assert np.array_equal(lens, rl) and np.array_equal(paths, rp) and st["strategy_steps"]["q1_lane"] == 0


@pytest.mark.parametrize("world,kind", [(1, 0), (3, 1), (2, 2), (1, 3)])
def test_shard_chain_kernels_long_rows(oracle, monkeypatch, world, kind):
    """k_chain_d / k_chain_scan / k_chain_u / k_chain_seq on LONG rows: a directed hub of 2^15 (or 40 000) out-edges reached from a few
    sources, constant r so that every table step at the hub sits on a CDF boundary (unit weights, power-of-two weights: rounding ties in
    every binade, integers, a wide dynamic range) — the chain's units as integer increments, binade crossings, the answer's unit —
    against the oracle's plain left-to-right loop."""
    rng = np.random.default_rng(40 + kind)
    n = 1 << 15 if kind < 2 else 40000
    hub, t0, s0, n_src = 0, 1, n + 1, 48
    if kind == 0:
        hw = np.ones(n, dtype=np.float32)
    elif kind == 1:
        hw = np.float32(2.0) ** rng.integers(-6, 6, n).astype(np.float32)
    elif kind == 2:
        hw = rng.integers(1, 1000, n).astype(np.float32)
    else:
        hw = np.exp(rng.random(n) * 16.0 - 8.0).astype(np.float32)
    src = [np.full(n, hub)]; dst = [np.arange(t0, t0 + n)]; w = [hw]
    tt = np.arange(t0, t0 + n)
    src.append(tt); dst.append(np.where(tt + 1 < t0 + n, tt + 1, t0)); w.append(np.ones(n, dtype=np.float32))          # the targets' ring
    back = tt[::97]
    src.append(back); dst.append(s0 + (np.arange(len(back)) % n_src)); w.append(np.full(len(back), 2.0, dtype=np.float32))   # some lead to a source
    src.append(back[::3]); dst.append(np.full(len(back[::3]), hub)); w.append(np.full(len(back[::3]), 0.5, dtype=np.float32))   # and to the hub (members of N(prev))
    ss = np.arange(s0, s0 + n_src)
    src.append(ss); dst.append(np.full(n_src, hub)); w.append(np.full(n_src, 3.0, dtype=np.float32))
    src.append(ss); dst.append(np.where(ss + 1 < s0 + n_src, ss + 1, s0)); w.append(np.ones(n_src, dtype=np.float32))
    src.append(ss); dst.append(t0 + (np.arange(n_src) * 131) % n); w.append(np.ones(n_src, dtype=np.float32))                # members: a source's targets
    s, d, w = (np.concatenate(x) for x in (src, dst, w))
    s, d, w = s.astype(np.int32), d.astype(np.int32), w.astype(np.float32)
    g = oracle.Graph.from_coo(s, d, w, directed=True)
    with pkg().Cluster([0] * world) as cl:
        cl.load_coo(s, d, w, directed=True)
        for r in (0.5, 0.25, 0.75):
            rp, rl, rs = g.walk(p=0.5, q=2.0, walk_length=6, rng="const", const_r=r, threads=8)
            paths, lens, st = cl.walk(p=0.5, q=2.0, walk_length=6, rng="const", const_r=r)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp), (world, kind, r)
            assert st["n_steps"] == rs
            if kind == 0:
                assert st["strategy_steps"]["chain"] > 0, st            # ties at the hub went through the chain kernels
        # Philox draws, and every sharded step on a row of >= 1000 candidates treated as a draw on a boundary (the debug switch of
        # run_shard_superstep): the chain kernels over arbitrary quotients, q != 1 (table steps) and q == 1 (per-lane steps)
        for q in (2.0, 1.0):
            rp, rl, rs = g.walk(p=0.5, q=q, walk_length=6, seed=5, threads=8)
            monkeypatch.delenv("SRW_DEBUG_CHAIN_DEG", raising=False)
            paths, lens, st = cl.walk(p=0.5, q=q, walk_length=6, seed=5)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp), (world, kind, q, "philox")
            monkeypatch.setenv("SRW_DEBUG_CHAIN_DEG", "1000")
            paths, lens, st = cl.walk(p=0.5, q=q, walk_length=6, seed=5)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp), (world, kind, q, "philox, forced chain")
            assert st["strategy_steps"]["chain"] > 0, st
            monkeypatch.delenv("SRW_DEBUG_CHAIN_DEG")
    # the whole-graph walk's ties take the same kernels (k_walk_tables records them, k_walk_general takes the resolved step)
    with pkg().Engine(device=0) as eng:
        eng.load_coo(s, d, w, directed=True)
        rp, rl, rs = g.walk(p=0.5, q=2.0, walk_length=6, seed=5, threads=8)
        monkeypatch.setenv("SRW_DEBUG_CHAIN_DEG", "1000")
        paths, lens, st = eng.walk(p=0.5, q=2.0, walk_length=6, seed=5)
        monkeypatch.delenv("SRW_DEBUG_CHAIN_DEG")
        assert np.array_equal(lens, rl) and np.array_equal(paths, rp), (kind, "whole graph, forced chain")
        ss = st["strategy_steps"]
        assert ss["ties_resolved"] <= ss["handed_over_walkers"], ss
        if kind < 3 and not os.environ.get("SRW_NO_TIE_KERNELS"):   # (kind 3's hub row has no exact prefix sums, hence no table: its steps are the general kernel's own)
            assert ss["ties_resolved"] > 0, ss


def test_shard_tables_built_over_a_progressively_mapped_buffer(oracle, monkeypatch):
    """The shards' table buffer as a virtual range mapped chunk by chunk while the build runs one segment of its work list per chunk
    (vm_buf.h, edge_tables.hip:k_eb_segments) — forced onto a small graph with 8 MiB chunks — against the oracle, worlds 1 and 3."""
    monkeypatch.setenv("SRW_EB_VMM_CHUNK_MB", "8")
    s, d, w = rmat_lines(oracle, 14, edge_factor=16, weighted=True)
    g = oracle.Graph.from_coo(s, d, w, directed=False)
    rp, rl, rs = g.walk(p=0.25, q=4.0, walk_length=10, seed=8, threads=8)
    for world in (1, 3):
        with pkg().Cluster([0] * world) as cl:
            cl.load_coo(s, d, w, directed=False)
            paths, lens, st = cl.walk(p=0.25, q=4.0, walk_length=10, seed=8)
            assert np.array_equal(lens, rl) and np.array_equal(paths, rp) and st["n_steps"] == rs, world
            assert st["edge_tables"] > 0 and st["edge_table_bytes"] > (32 << 20), st
