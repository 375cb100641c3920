"""Randomized differential test: HIP path vs CPU oracle (test infrastructure; run on the GPU box).
usage: python tests/fuzz_parity.py [seconds] [seed] [giant] [kernels]; test_gpu_parity.py runs 8 s of it.  kernels: every table-walk kernel form (waves / lanes modes / groups / rounds)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import _pkg
import oracle_py as oracle
pkg = _pkg.load()
def run(budget=60.0, seed=1, eng=None, giant=False, kernels=False):
    rng = np.random.default_rng(seed)
    eng = eng or pkg.Engine(0)
    t0, n_graphs, n_walks = time.time(), 0, 0
    WCHOICES = [
        lambda n: None,
        lambda n: rng.integers(1, 17, n).astype(np.float32),
        lambda n: (0.5 + 1.5 * rng.random(n)).astype(np.float32),
        lambda n: rng.choice(np.array([0.25, 0.5, 1, 2, 4, 1000, 0.001], np.float32), n),
        lambda n: (2.0 ** rng.integers(-30, 31, n)).astype(np.float32),
    ]
    while time.time() - t0 < budget:
        kind = rng.integers(0, 3)
        if giant and rng.integers(0, 12) == 0:
            kind = 3
        if kind == 0:      # small dense multigraph
            nv, nl = int(rng.integers(2, 40)), int(rng.integers(1, 400))
            s = rng.integers(-5, nv, nl).astype(np.int32); d = rng.integers(-5, nv, nl).astype(np.int32)
        elif kind == 1:    # hubs: a few vertices with hundreds to thousands of neighbors sharing many of them
            nh, nleaf = int(rng.integers(2, 6)), int(rng.integers(200, 6000))
            hubs = rng.integers(0, nh, nleaf * 2).astype(np.int32)
            leaves = (100 + rng.integers(0, nleaf, nleaf * 2)).astype(np.int32)
            extra_s = rng.integers(0, nh, 8).astype(np.int32); extra_d = rng.integers(0, nh, 8).astype(np.int32)
            s = np.concatenate([hubs, extra_s]); d = np.concatenate([leaves, extra_d])
        elif kind == 3:    # two or three giant hubs (rows beyond 65536 entries: wide chunks, multi-segment bitmaps)
            nh, nleaf = int(rng.integers(2, 4)), int(rng.integers(70000, 160000))
            s = np.concatenate([np.full(nleaf, h, np.int32) for h in range(nh)] + [np.arange(nh, dtype=np.int32)])
            d = np.concatenate([(10 + rng.permutation(nleaf + nleaf // 2)[:nleaf]).astype(np.int32) for h in range(nh)]
                               + [np.roll(np.arange(nh, dtype=np.int32), 1)])
        else:              # rmat
            sc = int(rng.integers(6, 12))
            s, d = oracle.rmat_edges(sc, int(rng.integers(4, 24)) << sc, seed=int(rng.integers(1, 1 << 30)))
        w = WCHOICES[int(rng.integers(0, len(WCHOICES)))](len(s))
        directed = bool(rng.integers(0, 2))
        g = oracle.Graph.from_coo(s, d, w, directed=directed)
        eng.load_coo(s, d, w, directed=directed)
        assert eng.stats() == (g.num_vertices, g.num_entries)
        n_graphs += 1
        for _ in range(4):
            p, q = [float(x) for x in rng.choice([0.25, 0.5, 1.0, 2.0, 4.0], 2)]
            L, nw, seed = int(rng.integers(0, 40)), int(rng.integers(1, 4)), int(rng.integers(0, 1 << 30))
            if kind == 3:
                L, nw = int(rng.integers(2, 7)), 1
            kw = dict(p=p, q=q, walk_length=L, num_walks=nw, seed=seed, first_walk=int(rng.integers(0, 5)))
            if rng.integers(0, 4) == 0:
                kw.update(rng="const", const_r=float(rng.choice([0.0, 0.25, 0.5, 0.75, 0.99999994, float(rng.random())])))
            sel = None
            if kind == 3:          # the CPU oracle is O(deg) per step: compare the hubs and a sample of the leaves only
                verts = eng.vertices()
                src = np.unique(np.concatenate([np.arange(nh, dtype=np.int32), rng.choice(verts, 40).astype(np.int32)]))
                sel = np.searchsorted(verts, src)
                ref = g.walk(sources=src, threads=8, **kw)
            else:
                ref = g.walk(threads=8, **kw)
            # default = per-edge tables (masks for short rows, chunk prefixes for long ones); the on-the-fly strategies
            # only run with the tables off
            variants = [dict(), dict(force_general=True), dict(edge_tables_all=True), dict(edge_tables=False),
                        dict(binned_tune=8 | int(rng.integers(1, 5)), edge_tables=False), dict(hub_bitmaps=False, edge_tables=False),
                        dict(binned=False), dict(prefix=False), dict(compact=False)]
            # the table walk's kernels (round 6): the default picks by the tables' geometry; the others are forced through their switches,
            # with the default tables and with a table for every certified pair (chunks of 4 candidates)
            if kernels:
                for env in ({"SRW_TABLE_LANES": "-1"}, {"SRW_TABLE_LANES": "3"}, {"SRW_TABLE_LANES": "2", "SRW_LANE_CSH": "8"}, {"SRW_TABLE_LANES": "0"},
                            {"SRW_TABLE_GROUPS": "1", "SRW_TABLE_LANES": "-1"}, {"SRW_TABLE_ROUNDS": "1"}):
                    variants.append(dict(_env=env))
                    variants.append(dict(_env=env, edge_tables_all=True))
            for v in variants:
                v = dict(v)
                env = v.pop("_env", None) or {}
                for k_, x_ in env.items(): os.environ[k_] = x_
                try:
                    got = eng.walk(**kw, **v)
                finally:
                    for k_ in env: os.environ.pop(k_, None)
                if env: v["_env"] = env
                n_walks += 1
                gp, gl = (got[0], got[1]) if sel is None else (got[0][sel], got[1][sel])
                if not (np.array_equal(gp, ref[0]) and np.array_equal(gl, ref[1]) and (sel is not None or got[2]["n_steps"] == ref[2])):
                    bad = np.nonzero((gp != ref[0]).any(axis=1))[0]
                    print("MISMATCH", dict(kind=int(kind), directed=directed, weights=None if w is None else w[:8]), kw, v,
                          "walker", bad[:3], gp[bad[0]] if len(bad) else None, ref[0][bad[0]] if len(bad) else None)
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True); np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_fail.npz"), s=s, d=d, w=w if w is not None else np.zeros(0))
                    return False
            if "rng" not in kw:     # Mode A against its own oracle
                a = eng.walk(sampler="alias", **kw)
                r = g.walk(threads=8, sampler=1, **kw) if sel is None else g.walk(sources=src, threads=8, sampler=1, **kw)
                n_walks += 1
                ap, al = (a[0], a[1]) if sel is None else (a[0][sel], a[1][sel])
                if not (np.array_equal(ap, r[0]) and np.array_equal(al, r[1])):
                    print("MODE-A MISMATCH", kw); return False
    print("fuzz ok: %d graphs, %d device walks in %.0f s" % (n_graphs, n_walks, time.time() - t0))
    return True


if __name__ == "__main__":
    ok = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1, giant=len(sys.argv) > 3 and "giant" in sys.argv[3:],
             kernels="kernels" in sys.argv[3:])
    sys.exit(0 if ok else 1)
