"""One-off verification of BASELINE config 5's stand-in at FULL size (directed unweighted RMAT-26 ef 27, p = 4 q = .5, Mode R, default sampler
selection: per-edge tables + the lean kernel): the device-generated graph is rebuilt in the CPU oracle from the same
(seed, edge index) stream and ~1 500 sampled walkers (incl. the 20 highest-degree hubs) are compared bit for bit.
Not collected by pytest (≈10 minutes of host time, ≈60 GB of host memory):  python tests/big_c5_check.py [scale] [edge factor]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _pkg
import oracle_py as oracle

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 27
n_edges = ef << scale
t = time.time()
s, d = oracle.rmat_edges(scale, n_edges, seed=42)
g = oracle.Graph.from_coo(s, d, None, directed=True)
del s, d
print("oracle graph: %d vertices, %d entries, %.0f s" % (g.num_vertices, g.num_entries, time.time() - t), flush=True)
pkg = _pkg.load()
eng = pkg.Engine(0)
eng.generate_rmat(scale, n_edges, seed=42, weighted=False, directed=True)
assert eng.stats() == (g.num_vertices, g.num_entries)
verts = eng.vertices()
sub = verts[:: max(1, len(verts) // 200000)]
deg = np.array([g.degree(int(v)) for v in sub])
hubs = sub[np.argsort(-deg)[:20]]
src = np.unique(np.concatenate([hubs, np.random.default_rng(3).choice(verts, 1500, replace=False)])).astype(np.int32)
idx = np.searchsorted(verts, src)
ok = True
for p, q, L in ((4.0, 0.5, 20),):
    t = time.time()
    rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=L, seed=2026, threads=min(64, os.cpu_count() or 8))
    t_or = time.time() - t
    paths, lens, st = eng.walk(p=p, q=q, walk_length=L, seed=2026)
    same = bool(np.array_equal(paths[idx], rp) and np.array_equal(lens[idx], rl))
    ok &= same
    ss = {k: v for k, v in st["strategy_steps"].items() if v}
    print("p=%g q=%g L=%d: %d sampled walkers (max degree %d) %s; oracle %.0f s; device kernel %.0f ms, setup %.0f ms, %s"
          % (p, q, L, len(src), int(deg.max()), "IDENTICAL" if same else "MISMATCH", t_or, st["kernel_ms"], st["setup_ms"], ss), flush=True)
print("config 5 stand-in at full size:", "parity OK" if ok else "PARITY FAILED")
sys.exit(0 if ok else 1)
