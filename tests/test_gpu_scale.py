"""GPU parity AT SCALE (GPU == oracle on RMAT-10..16 and beyond): weighted RMAT-16 and RMAT-20 generated on
the device, the same graph rebuilt in the CPU oracle from the same (seed, edge index) stream, ~2 000 sampled walkers
including the 20 highest-degree hubs compared BIT FOR BIT for the biased configs of BASELINE.json, with the DEFAULT
strategy selection — hub bitmaps, the edge hash set, the per-edge bias tables and the on-the-fly searches must fire on
their own (asserted through srw_walk_stats.strategy_steps)."""
import os

import numpy as np
import pytest

from helpers import pkg, rmat_weights_np

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = pkg().Engine(device=0)
    yield e
    e.close()


def _sources(g, verts, n_sample, seed):
    deg = np.array([g.degree(int(v)) for v in verts[:: max(1, len(verts) // 200000)]])   # sampled degrees are enough to find hubs
    sub = verts[:: max(1, len(verts) // 200000)]
    hubs = sub[np.argsort(-deg)[:20]]
    rng = np.random.default_rng(seed)
    rest = rng.choice(verts, size=n_sample, replace=False)
    return np.unique(np.concatenate([hubs, rest])).astype(np.int32)


@pytest.mark.parametrize("scale,n_sample,L", [(16, 2000, 40), (20, 1500, 24)])
def test_weighted_rmat_biased_walks_equal_oracle(eng, oracle, scale, n_sample, L):
    n_edges = 16 << scale
    s, d = oracle.rmat_edges(scale, n_edges, seed=42)
    w = rmat_weights_np(s, d, 42)
    assert np.array_equal(w[:256], np.array([oracle.rmat_weight(a, b, 42) for a, b in zip(s[:256], d[:256])], dtype=np.float32))
    g = oracle.Graph.from_coo(s, d, w, directed=False)
    eng.generate_rmat(scale, n_edges, seed=42, weighted=True)          # same stream, generated on the device
    assert eng.stats() == (g.num_vertices, g.num_entries)
    verts = eng.vertices()
    src = _sources(g, verts, n_sample, scale)
    idx = np.searchsorted(verts, src)
    hub = int(src[np.argmax([g.degree(int(v)) for v in src])])
    ids, ws = eng.neighbors(hub)
    oi, ow = g.neighbors(hub)
    assert np.array_equal(ids, oi) and np.array_equal(ws, ow)            # neighbor order of the biggest row
    for p, q in [(0.25, 4.0), (4.0, 0.5), (0.5, 1.0)]:
        rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=L, seed=1234, threads=8)
        # default strategy selection (per-edge tables on)
        paths, lens, st = eng.walk(p=p, q=q, walk_length=L, seed=1234)
        assert np.array_equal(lens[idx], rl), (scale, p, q)
        bad = np.nonzero((paths[idx] != rp).any(axis=1))[0]
        assert bad.size == 0, (scale, p, q, int(src[bad[0]]), paths[idx][bad[0]], rp[bad[0]])
        ss = st["strategy_steps"]
        if q != 1.0:
            assert st["edge_tables"] > 0 and ss["edge_table"] > 0 and ss["edge_mask"] > 0, st
            if scale == 16:     # chunk prefixes as 16-bit chunk masses / floats where they are exact: the f64 layout must give the same paths
                os.environ["SRW_EB_NO_F32"] = "1"; os.environ["SRW_EB_NO_U16"] = "1"
                try:
                    p64, l64, st64 = eng.walk(p=p, q=q, walk_length=L, seed=1234)
                finally:
                    del os.environ["SRW_EB_NO_F32"]; del os.environ["SRW_EB_NO_U16"]
                assert np.array_equal(p64, paths) and np.array_equal(l64, lens)
                assert st64["edge_table_bytes"] > st["edge_table_bytes"], (st64["edge_table_bytes"], st["edge_table_bytes"])
                eng.walk(p=p, q=q, walk_length=1, seed=1234)       # back to the default layout for what follows
            # without the tables the on-the-fly strategies carry the hub steps: they must fire on their own
            paths2, lens2, st2 = eng.walk(p=p, q=q, walk_length=L, seed=1234, edge_tables=False)
            assert np.array_equal(paths2, paths) and np.array_equal(lens2, lens)
            s2 = st2["strategy_steps"]
            assert s2["edge_table"] == 0 and s2["p1"] > 0 and (s2["p3"] > 0 or scale < 20) and s2["scan"] > 0, st2
            if scale >= 20:
                assert s2["w"] + s2["p2"] + s2["p3"] > 0, st2
        else:
            assert ss["q1_lane"] > 0.99 * st["n_steps"] and st["edge_tables"] == 0, st      # one walker per lane
            os.environ["SRW_NO_Q1_KERNEL"] = "1"
            try:
                pw, lw, stw = eng.walk(p=p, q=q, walk_length=L, seed=1234)                   # one walker per wave
            finally:
                del os.environ["SRW_NO_Q1_KERNEL"]
            assert np.array_equal(pw, paths) and np.array_equal(lw, lens) and stw["strategy_steps"]["prefix"] > 0


def test_directed_rmat_biased_walk_equals_oracle(eng, oracle):
    """Config 5's shape (directed, p = 4, q = .5) at RMAT-18 ef 27."""
    scale, n_edges = 18, 27 << 18
    s, d = oracle.rmat_edges(scale, n_edges, seed=7)
    g = oracle.Graph.from_coo(s, d, None, directed=True)
    eng.generate_rmat(scale, n_edges, seed=7, weighted=False, directed=True)
    assert eng.stats() == (g.num_vertices, g.num_entries)
    verts = eng.vertices()
    src = _sources(g, verts, 1500, 5)
    idx = np.searchsorted(verts, src)
    rp, rl, _ = g.walk(sources=src, p=4.0, q=0.5, walk_length=40, seed=99, threads=8)
    paths, lens, st = eng.walk(p=4.0, q=0.5, walk_length=40, seed=99)
    assert np.array_equal(lens[idx], rl) and np.array_equal(paths[idx], rp)
    assert st["edge_tables"] > 0 and st["strategy_steps"]["edge_table"] > 0, st


@pytest.mark.parametrize("world,weighted", [(4, False), (3, True)])
def test_sharded_linked_first_order_equals_replicated_at_scale(eng, world, weighted):
    """RMAT-18 through the in-process cluster (row links across the shards + the fused sample-and-bucket kernel) must give
    the replicated single-launch kernel's paths bit for bit — which the tests above pin to the oracle."""
    scale = 18
    eng.generate_rmat(scale, 16 << scale, seed=7, weighted=weighted)
    paths, lens, st = eng.walk(walk_length=40, num_walks=2, first_walk=3, seed=99)
    with pkg().Cluster([0] * world) as cl:
        cl.generate_rmat(scale, 16 << scale, seed=7, weighted=weighted)
        assert cl.stats() == eng.stats()
        for batch in (1, 2):
            cp, clens, cst = cl.walk(walk_length=40, num_walks=2, first_walk=3, seed=99, batch=batch)
            assert np.array_equal(clens, lens) and np.array_equal(cp, paths), (world, batch)
            assert cst["n_steps"] == st["n_steps"]


@pytest.mark.parametrize("fail_above,expect_tables", [("32", True), ("0", False)])
def test_table_build_failure_degrades_instead_of_failing(oracle, monkeypatch, fail_above, expect_tables):
    """An allocation of the per-edge table build that fails (simulated: SRW_EB_FAIL_ABOVE) must not fail the walk:
    prepare_tables retries with 32 chunks, then walks without tables — same paths either way."""
    monkeypatch.setenv("SRW_EB_FAIL_ABOVE", fail_above)
    scale, L = 16, 24
    with pkg().Engine(device=0) as e:
        e.generate_rmat(scale, 16 << scale, seed=11, weighted=True)
        s, d = oracle.rmat_edges(scale, 16 << scale, seed=11)
        from helpers import rmat_weights_np
        g = oracle.Graph.from_coo(s, d, rmat_weights_np(s, d, 11))
        verts = e.vertices()
        src = np.unique(np.random.default_rng(1).choice(verts, 600, replace=False)).astype(np.int32)
        idx = np.searchsorted(verts, src)
        paths, lens, st = e.walk(p=0.25, q=4.0, walk_length=L, seed=77)
        rp, rl, _ = g.walk(sources=src, p=0.25, q=4.0, walk_length=L, seed=77, threads=16)
        assert np.array_equal(paths[idx], rp) and np.array_equal(lens[idx], rl)
        assert (st["edge_tables"] > 0) == expect_tables, st["edge_tables"]


_GEOMETRIES = [
    {"SRW_EB_NO_U16": "1"},                                                       # absolute prefixes at level 0
    {"SRW_EB_CM_MAX": "0"},                                                       # no chunk masks: every located chunk probes
    {"SRW_EB_CM_MIN_DU": "0"},                                                    # chunk masks on every pair into a row of up to 16 384 candidates
    {"SRW_EB_CM_MAX": "0", "SRW_EB_FINE_CAP": "4096", "SRW_EB_FINE_MIN_DU": "0"},  # finer tables everywhere: two-level trees, HBM-scratch bins in the build
    {"SRW_EB_CM_MAX": "4096", "SRW_EB_FINE_CAP": "1024", "SRW_EB_NO_U16": "1"},
    {"SRW_EB_CHUNKS": "32", "SRW_EB_MIN_SH": "8"},                                # the coarse complete set of a graph that fills the GPU
    {"SRW_EB_VMM_CHUNK_MB": "64"},                                                # the table buffer mapped in 64 MiB chunks while the build fills it, one build segment per chunk (vm_buf.h; default: 4 GiB chunks, tables >= 8 GiB)
    {"SRW_EB_VMM_CHUNK_MB": "16", "SRW_EB_FINE_CAP": "4096", "SRW_EB_FINE_MIN_DU": "0"},
]


def test_table_geometries_give_the_same_paths(monkeypatch):
    """The per-edge tables' geometry (chunk size by (deg(curr), deg(prev)), chunk masks, finer tables, 16-bit level 0, tree depth) is a
    memory / request trade: every choice must give the paths of the default selection — which the tests above pin to the oracle."""
    scale = 18

    def walk(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        try:
            with pkg().Engine(device=0) as e:
                e.generate_rmat(scale, 16 << scale, seed=5, weighted=True)
                out = []
                for p, q in ((0.25, 4.0), (4.0, 0.5)):
                    paths, lens, st = e.walk(p=p, q=q, walk_length=30, seed=4321)
                    assert st["strategy_steps"]["edge_table"] > 0, (env, st)
                    out.append((paths, lens, st["edge_table_bytes"]))
                return out
        finally:
            for k in env:
                monkeypatch.delenv(k, raising=False)

    ref = walk({})
    sizes = {"default": [r[2] for r in ref]}
    for env in _GEOMETRIES:
        got = walk(env)
        for (rp, rl, _), (gp, gl, gb) in zip(ref, got):
            assert np.array_equal(gl, rl) and np.array_equal(gp, rp), env
        sizes[str(env)] = [r[2] for r in got]
    assert sizes[str(_GEOMETRIES[0])][0] > sizes["default"][0], sizes      # the 16-bit level 0 is what makes the default smaller
    assert sizes[str(_GEOMETRIES[2])][0] > sizes["default"][0], sizes      # masks on every pair cost bytes


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tables_on_hubs_with_heavy_multi_edges(oracle, seed):
    """Multi-edges through the per-edge tables against the oracle: eight hubs of ~6 000 entries over 3 000 shared leaves, every edge with a
    random multiplicity 1 .. 5 (runs of equal ids in both sorted rows, some across the 1 024-id boundary of the staged chunks), hub <-> hub
    multi-edges, self-loops on hubs and leaves, small integer weights.  A multi-edge is ONE pair: its entries share one table
    (edge_tables.hip:eb_enum, k_eb_dups)."""
    rng = np.random.default_rng(seed)
    n_hub, n_leaf = 8, 3000
    s, d = [], []
    for h in range(n_hub):
        leaves = n_hub + rng.choice(n_leaf, 2000, replace=False)
        mult = rng.integers(1, 6, len(leaves))
        s.append(np.full(int(mult.sum()), h)); d.append(np.repeat(leaves, mult))
        for h2 in range(n_hub):                                # hub <-> hub, incl. self-loops (h2 == h)
            m = int(rng.integers(0, 4))
            s.append(np.full(m, h)); d.append(np.full(m, h2))
    a = n_hub + rng.integers(0, n_leaf, 4000); b = n_hub + rng.integers(0, n_leaf, 4000)     # leaf - leaf, some self-loops
    s.append(a); d.append(b)
    s = np.concatenate(s).astype(np.int32); d = np.concatenate(d).astype(np.int32)
    perm = rng.permutation(len(s)); s, d = s[perm], d[perm]                                  # input order is not sorted order
    w = rng.integers(1, 9, len(s)).astype(np.float32)
    g = oracle.Graph.from_coo(s, d, w, directed=False)
    with pkg().Engine(device=0) as e:
        e.load_coo(s, d, w, directed=False)
        assert e.stats() == (g.num_vertices, g.num_entries)
        verts = e.vertices()
        for p, q in ((0.25, 4.0), (4.0, 0.5), (0.5, 2.0)):
            rp, rl, _ = g.walk(p=p, q=q, walk_length=20, seed=77, threads=8)
            paths, lens, st = e.walk(p=p, q=q, walk_length=20, seed=77)
            assert np.array_equal(lens, rl), (p, q)
            bad = np.nonzero((paths != rp).any(axis=1))[0]
            assert bad.size == 0, (p, q, int(verts[bad[0]]), paths[bad[0]], rp[bad[0]])
            assert st["strategy_steps"]["edge_table"] > 0, st


def test_plan_walks_steers_the_tables_not_the_paths():
    """srw_plan_walks (the job's --numWalks, Params.scala:7-23): the finer per-edge tables are built only for a job long enough to
    pay for them (>= 64 planned iterations; default: the reference's 10).  The table set changes, the paths never do."""
    scale = 18
    res = []
    for planned in (0, 10, 200):
        with pkg().Engine(device=0) as e:
            e.generate_rmat(scale, 16 << scale, seed=5, weighted=True)
            if planned:
                e.plan_walks(planned)
            paths, lens, st = e.walk(p=0.25, q=4.0, walk_length=30, seed=4321)
            assert st["strategy_steps"]["edge_table"] > 0, st
            res.append((paths, lens, st["edge_table_bytes"]))
    for paths, lens, _ in res[1:]:
        assert np.array_equal(lens, res[0][1]) and np.array_equal(paths, res[0][0])
    assert res[0][2] == res[1][2] and res[2][2] > res[1][2], [r[2] for r in res]      # 200 planned iterations: the finer tables are in
    with pkg().Engine(device=0) as e:
        with pytest.raises(pkg().SrwError):
            e.plan_walks(-1)


def test_long_chunk_masks_on_hub_rows(oracle, monkeypatch):
    """Round 6: pairs (prev -> curr) into rows of MORE than 16 384 candidates whose N(prev) is too long for the LDS staging get their membership
    mask too — the builder ORs the bits straight into the pair's table block (sampling.h:eb_pair_geometry, edge_tables.hip:k_eb_build).  Three
    hubs of 40 000 / 70 000 / 100 000 leaves sharing many of them, joined pairwise; against the oracle, with the long masks (SRW_EB_CM_MAX: the planner
    never takes them by itself — they do not fit where they would matter, profiles/r06_long_masks.md) and without."""
    rng = np.random.default_rng(5)
    n_leaf = 120000
    leaves = np.arange(8, 8 + n_leaf, dtype=np.int32)
    s, d = [], []
    for hub, k in ((0, 40000), (1, 70000), (2, 100000)):
        s.append(np.full(k, hub, np.int32)); d.append(rng.choice(leaves, k, replace=False).astype(np.int32))
    s.append(np.array([0, 1, 2, 0, 1], np.int32)); d.append(np.array([1, 2, 0, 2, 2], np.int32))      # hub <-> hub edges (one doubled)
    a = rng.integers(8, 8 + n_leaf, 60000).astype(np.int32); b = rng.integers(8, 8 + n_leaf, 60000).astype(np.int32)
    s.append(a); d.append(b)
    s = np.concatenate(s); d = np.concatenate(d)
    w = (1 + rng.integers(0, 8, len(s))).astype(np.float32)
    g = oracle.Graph.from_coo(s, d, w, directed=False)
    res = {}
    for long_masks in (True, False):
        if long_masks: monkeypatch.setenv("SRW_EB_CM_MAX", str(1 << 30))
        with pkg().Engine(device=0) as e:
            e.load_coo(s, d, w, directed=False)
            verts = e.vertices()
            src = np.unique(np.concatenate([[0, 1, 2], rng.choice(verts, 4000, replace=False)])).astype(np.int32)
            idx = np.searchsorted(verts, src)
            for p, q in ((0.25, 4.0), (2.0, 0.5)):
                rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=16, seed=41, threads=8)
                for env in ({}, {"SRW_TABLE_LANES": "-1"}, {"SRW_TABLE_LANES": "2", "SRW_LANE_CSH": "8"}):
                    for k_, v_ in env.items(): monkeypatch.setenv(k_, v_)
                    paths, lens, st = e.walk(p=p, q=q, walk_length=16, seed=41)
                    for k_ in env: monkeypatch.delenv(k_)
                    assert np.array_equal(lens[idx], rl) and np.array_equal(paths[idx], rp), (long_masks, p, q, env)
                    assert st["strategy_steps"]["edge_table"] > 0, st
                res[(long_masks, p)] = st["edge_table_bytes"]
        if long_masks: monkeypatch.delenv("SRW_EB_CM_MAX")
    assert res[(True, 0.25)] > res[(False, 0.25)]          # the hub -> hub pairs' masks are there


def test_three_level_table_on_a_long_row(oracle, monkeypatch):
    """A row of 300 000 candidates with chunks of 64: 4 688 chunks, a three-level tree (4 688 -> 74 -> 2), built through the HBM-scratch
    bins; against the oracle."""
    monkeypatch.setenv("SRW_EB_FINE_CAP", "32768")
    monkeypatch.setenv("SRW_EB_FINE_MIN_DU", "0")
    monkeypatch.setenv("SRW_EB_CM_MAX", "0")
    rng = np.random.default_rng(12)
    n_leaf = 300000
    hub, hub2 = 0, 1
    leaves = np.arange(2, 2 + n_leaf, dtype=np.int32)
    s = [np.full(n_leaf, hub, np.int32), np.full(n_leaf // 3, hub2, np.int32)]
    d = [leaves, rng.choice(leaves, n_leaf // 3, replace=False).astype(np.int32)]
    a = rng.integers(2, 2 + n_leaf, 200000).astype(np.int32); b = rng.integers(2, 2 + n_leaf, 200000).astype(np.int32)
    s.append(a); d.append(b)
    s.append(np.array([hub], np.int32)); d.append(np.array([hub2], np.int32))
    s = np.concatenate(s); d = np.concatenate(d)
    w = (1 + (rng.integers(0, 8, len(s)))).astype(np.float32)
    g = oracle.Graph.from_coo(s, d, w, directed=False)
    with pkg().Engine(device=0) as e:
        e.load_coo(s, d, w, directed=False)
        verts = e.vertices()
        src = np.unique(np.concatenate([[hub, hub2], rng.choice(verts, 3000, replace=False)])).astype(np.int32)
        idx = np.searchsorted(verts, src)
        for p, q in ((0.25, 4.0), (2.0, 0.5)):
            rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=12, seed=31, threads=8)
            paths, lens, st = e.walk(p=p, q=q, walk_length=12, seed=31)
            assert np.array_equal(lens[idx], rl) and np.array_equal(paths[idx], rp), (p, q)
            assert st["strategy_steps"]["edge_table"] > 0, st


@pytest.mark.gpu
@pytest.mark.parametrize("weighted", [False, True])
def test_non_dyadic_pq_through_tables_and_masks(eng, oracle, weighted):
    """p, q that are not powers of two: the biased weights are real f32 divides (RandomSample.scala:33,35), one per candidate with the
    divisor selected per lane (sampling.h:BiasDiv) — on the table steps, the mask steps and the on-the-fly samplers alike."""
    scale, L = 14, 30
    n_edges = 16 << scale
    s, d = oracle.rmat_edges(scale, n_edges, seed=9)
    w = rmat_weights_np(s, d, 9) if weighted else None
    g = oracle.Graph.from_coo(s, d, w, directed=False)
    eng.generate_rmat(scale, n_edges, seed=9, weighted=weighted)
    assert eng.stats() == (g.num_vertices, g.num_entries)
    verts = eng.vertices()
    src = _sources(g, verts, 1500, scale)
    idx = np.searchsorted(verts, src)
    for p, q in [(0.3, 0.7), (3.0, 7.0), (1.1, 0.9), (0.25, 3.0)]:
        rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=L, seed=4321, threads=8)
        paths, lens, st = eng.walk(p=p, q=q, walk_length=L, seed=4321)
        assert np.array_equal(lens[idx], rl), (p, q)
        bad = np.nonzero((paths[idx] != rp).any(axis=1))[0]
        assert bad.size == 0, (p, q, int(src[bad[0]]), paths[idx][bad[0]], rp[bad[0]])
        paths2, lens2, _ = eng.walk(p=p, q=q, walk_length=L, seed=4321, edge_tables=False)
        assert np.array_equal(paths2, paths) and np.array_equal(lens2, lens), (p, q)


def test_a_refused_mapping_call_falls_back_to_one_allocation(monkeypatch):
    """vm_buf.h: the table buffer is mapped chunk by chunk while the build fills it; if a mapping call refuses for another reason than
    memory (simulated at the third chunk, with segments of the build already running over the first two), the tables are built once
    more over one hipMalloc and the walk goes on — same paths, same tables."""
    import subprocess, sys
    from conftest import ROOT
    code = ("import sys, hashlib; sys.path.insert(0, %r); import _pkg; P = _pkg.load(); e = P.Engine(0); "
            "e.generate_rmat(16, 16 << 16, seed=5, weighted=True); "
            "p, l, st = e.walk(p=0.25, q=4.0, walk_length=12, seed=3); "
            "print('RESULT', hashlib.sha256(p.tobytes() + l.tobytes()).hexdigest(), st['edge_tables'], st['strategy_steps']['edge_table'])" % ROOT)

    def run(env):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SRW_TIMING="1", **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0], r.stderr

    ref, _ = run({"SRW_EB_NO_VMM": "1"})
    mapped, err = run({"SRW_EB_VMM_CHUNK_MB": "8"})
    assert mapped == ref and "mapped in chunks" in err
    fell_back, err = run({"SRW_EB_VMM_CHUNK_MB": "8", "SRW_EB_VMM_FAIL_AT": "2"})
    assert fell_back == ref and "once more with one allocation" in err and "simulated failure" in err
    assert int(ref.split()[2]) > 0 and int(ref.split()[3]) > 0


def test_mapped_table_buffer_gives_its_memory_back():
    """vm_buf.h: a table buffer mapped in N chunks is unmapped chunk by chunk — after a second (p, q) rebuilt the tables in the
    same process and after the engine is closed the device's free memory is back where it was, and no unmap / release call complained."""
    import subprocess, sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); import torch; torch.cuda.init(); import _pkg; P = _pkg.load(); "
            "free = lambda: torch.cuda.mem_get_info(0)[0]; f0 = free(); e = P.Engine(0); "
            "e.generate_rmat(16, 16 << 16, seed=5, weighted=True); f1 = free(); "
            "st = e.walk(fetch=False, p=0.25, q=4.0, walk_length=12, seed=3); f2 = free(); "
            "st2 = e.walk(fetch=False, p=4.0, q=0.5, walk_length=12, seed=3); f3 = free(); "
            "st3 = e.walk(fetch=False, p=0.25, q=4.0, walk_length=12, seed=3); f4 = free(); e.close(); f5 = free(); "
            "print('RESULT', f0, f1, f2, f3, f4, f5, st['edge_tables'], st2['edge_tables'])" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SRW_TIMING="1", SRW_EB_VMM_CHUNK_MB="8"), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "mapped in chunks" in r.stderr and "VmBuf::release" not in r.stderr, r.stderr[-2000:]
    f0, f1, f2, f3, f4, f5, t1, t2 = (int(x) for x in [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()[1:])
    assert t1 > 0 and t2 > 0
    slack = 64 << 20
    assert abs(f4 - f2) <= slack, (f2, f3, f4)          # the third build (the first (p, q) again) stands where the first stood: nothing of the second is left
    assert f5 >= f0 - (1 << 30), (f0, f5)               # everything is back once the engine is closed (what stays: the runtime's own pools — kernel scratch, code objects)


@pytest.mark.parametrize("kernel", [{}, {"SRW_TABLE_LANES": "3"}, {"SRW_TABLE_LANES": "0"}, {"SRW_TABLE_LANES": "2", "SRW_LANE_CSH": "8"}, {"SRW_TABLE_GROUPS": "1", "SRW_TABLE_LANES": "-1"},
                                    {"SRW_TABLE_ROUNDS": "1"}, {"SRW_TABLE_ROUNDS": "1", "SRW_LANE_CSH": "8"}])
def test_table_kernels_walk_every_walker_alike(oracle, monkeypatch, kernel):
    """The table walk has three kernels: one walker per wave (k_walk_tables, SRW_TABLE_LANES=-1: the reference here), one per lane
    (walk_lanes.hip: the default — {} — where the tables' chunks are 64 candidates; modes: rows + tables per lane, every step served by the
    wave, located chunks up to 256 candidates per lane) and one per 16 lanes (walk_groups.hip, opt-in) — profiles/r06_lane_kernel.md,
    r06_group_kernel.md.  EVERY walker as the wave kernel walks it — weighted and unit-weight, undirected and directed, three-level tables, chunk masks off, and with every
    table step on a long row treated as a boundary draw (the tie list + the chain kernels) — and a sample against the oracle."""
    rng = np.random.default_rng(77)
    cases = [("w", False, 14, (0.25, 4.0), {}), ("", False, 14, (4.0, 0.5), {}), ("w", True, 13, (0.5, 2.0), {}),
             ("w", False, 13, (0.25, 4.0), {"SRW_EB_CM_MAX": "0", "SRW_EB_FINE_CAP": "32768", "SRW_EB_FINE_MIN_DU": "0"}),
             ("w", False, 13, (0.25, 4.0), {"SRW_DEBUG_CHAIN_DEG": "600"}), ("w", False, 12, (3.0, 0.7), {})]
    for spec, directed, scale, (p, q), env in cases:
        for k, v in env.items(): monkeypatch.setenv(k, v)
        with pkg().Engine(device=0) as e:
            e.generate_rmat(scale, 16 << scale, seed=9, weighted=bool(spec), directed=directed)
            monkeypatch.setenv("SRW_TABLE_LANES", "-1")
            ref_p, ref_l, st0 = e.walk(p=p, q=q, walk_length=40, num_walks=2, seed=5)
            monkeypatch.delenv("SRW_TABLE_LANES")
            assert st0["strategy_steps"]["edge_table"] > 0 and st0["strategy_steps"]["edge_mask"] > 0, st0
            for k, v in kernel.items(): monkeypatch.setenv(k, v)
            got_p, got_l, st = e.walk(p=p, q=q, walk_length=40, num_walks=2, seed=5)
            for k in kernel: monkeypatch.delenv(k)
            assert np.array_equal(got_l, ref_l) and np.array_equal(got_p, ref_p), (kernel, spec, directed, p, q, env)
            for key in ("edge_table", "edge_mask", "scan", "handed_over_walkers"):
                assert st["strategy_steps"][key] == st0["strategy_steps"][key], (kernel, key, st, st0)
            assert st["ent_reads"] == st0["ent_reads"] and st["n_steps"] == st0["n_steps"]
            assert abs(st["trials"] - st0["trials"]) <= 1e-4 * st0["trials"]      # (bytes-read accounting: a diagnostic, counted per kernel form)
            if env.get("SRW_DEBUG_CHAIN_DEG"): assert st["strategy_steps"]["handed_over_walkers"] > 0
        for k in env: monkeypatch.delenv(k)
    # ... and the paths are the oracle's (the first case, sampled)
    from helpers import rmat_weights_np as wts
    s_, d_ = oracle.rmat_edges(14, 16 << 14, seed=9)
    g = oracle.Graph.from_coo(s_, d_, wts(s_, d_, 9), directed=False)
    with pkg().Engine(device=0) as e:
        e.generate_rmat(14, 16 << 14, seed=9, weighted=True)
        for k, v in kernel.items(): monkeypatch.setenv(k, v)
        paths, lens, _ = e.walk(p=0.25, q=4.0, walk_length=40, seed=5)
        verts = e.vertices()
        pick = rng.choice(len(verts), 400, replace=False)
        rp, rl, _ = g.walk(sources=verts[pick].astype(np.int32), p=0.25, q=4.0, walk_length=40, seed=5, threads=8)
        assert np.array_equal(paths[pick], rp) and np.array_equal(lens[pick], rl), kernel
