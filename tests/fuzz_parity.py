"""Randomized differential test: HIP path vs CPU oracle (test infrastructure; run on the GPU box).
usage: python tests/fuzz_parity.py [seconds] [seed]; test_gpu_parity.py runs 8 s of it."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import _pkg
import oracle_py as oracle
pkg = _pkg.load()
def run(budget=60.0, seed=1, eng=None):
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
        if kind == 0:      # small dense multigraph
            nv, nl = int(rng.integers(2, 40)), int(rng.integers(1, 400))
            s = rng.integers(-5, nv, nl).astype(np.int32); d = rng.integers(-5, nv, nl).astype(np.int32)
        elif kind == 1:    # hubs: a few vertices with hundreds to thousands of neighbors sharing many of them
            nh, nleaf = int(rng.integers(2, 6)), int(rng.integers(200, 6000))
            hubs = rng.integers(0, nh, nleaf * 2).astype(np.int32)
            leaves = (100 + rng.integers(0, nleaf, nleaf * 2)).astype(np.int32)
            extra_s = rng.integers(0, nh, 8).astype(np.int32); extra_d = rng.integers(0, nh, 8).astype(np.int32)
            s = np.concatenate([hubs, extra_s]); d = np.concatenate([leaves, extra_d])
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
            kw = dict(p=p, q=q, walk_length=L, num_walks=nw, seed=seed, first_walk=int(rng.integers(0, 5)))
            if rng.integers(0, 4) == 0:
                kw.update(rng="const", const_r=float(rng.choice([0.0, 0.25, 0.5, 0.75, 0.99999994, float(rng.random())])))
            ref = g.walk(threads=8, **kw)
            variants = [dict(), dict(force_general=True), dict(binned_tune=4 | int(rng.integers(1, 4))), dict(binned=False),
                        dict(prefix=False), dict(compact=False)]
            for v in variants:
                got = eng.walk(**kw, **v)
                n_walks += 1
                if not (np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and got[2]["n_steps"] == ref[2]):
                    bad = np.nonzero((got[0] != ref[0]).any(axis=1))[0]
                    print("MISMATCH", dict(kind=int(kind), directed=directed, weights=None if w is None else w[:8]), kw, v,
                          "walker", bad[:3], got[0][bad[0]] if len(bad) else None, ref[0][bad[0]] if len(bad) else None)
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True); np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_fail.npz"), s=s, d=d, w=w if w is not None else np.zeros(0))
                    return False
            if "rng" not in kw:     # Mode A against its own oracle
                a = eng.walk(sampler="alias", **kw)
                r = g.walk(threads=8, sampler=1, **kw)
                n_walks += 1
                if not (np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1])):
                    print("MODE-A MISMATCH", kw); return False
    print("fuzz ok: %d graphs, %d device walks in %.0f s" % (n_graphs, n_walks, time.time() - t0))
    return True


if __name__ == "__main__":
    ok = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sys.exit(0 if ok else 1)
