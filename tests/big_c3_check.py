"""Verification of BASELINE config 3 at FULL size (weighted RMAT-24 ef 16, p = .25 q = 4, Mode R, default sampler selection: per-edge
tables + the lean kernel; also (4, .5) and (.25, 1)): 20 000 sampled walkers at walkLength 80 for the configuration's own (p, q) (~1 500 for the other two; incl. the 20 highest-degree starts of
a 2 000-vertex sample) are compared bit for bit with the CPU ORACLE.  The oracle's graph holds the adjacency rows of every vertex on those walkers'
DEVICE paths, rebuilt on the host from the same (seed, edge index) stream (the oracle's generator, 256 M lines at a time, lines kept in
stream order — rows of prev and curr alike, so computeSecondOrderWeights sees what the reference would).  A deviation sends the oracle's
walker into a row that was not collected (partial or empty), which shows as a mismatch: the check cannot pass on the device's say-so.
Run by tests/test_gpu_full_size.py:  python tests/big_c3_check.py [scale]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _pkg
import oracle_py as oracle
from helpers import rmat_weights_np

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_edges = 16 << scale
# (p, q, walkLength, sampled walkers): the configuration's own case at its walkLength 80 over 20 000 walkers (the rare paths
# — boundary draws, hand-overs — must be among the compared walkers), the other two as before
CASES = ((0.25, 4.0, 80, 20000), (4.0, 0.5, 16, 1500), (0.25, 1.0, 24, 1500))
pkg = _pkg.load()
t = time.time()
eng = pkg.Engine(0)
eng.generate_rmat(scale, n_edges, seed=42, weighted=True)
nv, ne = eng.stats()
verts = eng.vertices()
rng = np.random.default_rng(3)
cand = np.arange(0, len(verts), max(1, len(verts) // 2000))
degs = np.array([len(eng.neighbors(int(verts[i]))[0]) for i in cand])
picks = [np.unique(np.concatenate([cand[np.argsort(-degs)[:20]], rng.choice(len(verts), n, replace=False)])) for _, _, _, n in CASES]
print("device graph: %d vertices, %d entries, %.0f s" % (nv, ne, time.time() - t), flush=True)
# the device walks first: their paths say which rows the oracle needs
dev = []
on_path = np.zeros(1 << scale, dtype=bool)
for (p, q, L, _), pick in zip(CASES, picks):
    paths, lens, st = eng.walk(p=p, q=q, walk_length=L, seed=2026)
    sp, sl = paths[pick].copy(), lens[pick].copy()
    on_path[sp[sp >= 0]] = True
    dev.append((sp, sl, st))
    del paths, lens
t = time.time()
fs, fd = [], []
BLOCK = 1 << 28
for lo in range(0, n_edges, BLOCK):
    s_, d_ = oracle.rmat_edges(scale, min(BLOCK, n_edges - lo), seed=42, first=lo)
    keep = on_path[s_] | on_path[d_]
    fs.append(s_[keep]); fd.append(d_[keep])
    del s_, d_, keep
fs = np.concatenate(fs); fd = np.concatenate(fd)
g = oracle.Graph.from_coo(fs, fd, rmat_weights_np(fs, fd, 42), directed=False)
print("oracle rows of %d path vertices rebuilt from the edge stream (%d of %d lines kept), %.0f s" % (int(on_path.sum()), len(fs), n_edges, time.time() - t), flush=True)
ok = True
for (p, q, L, _), pick, (sp, sl, st) in zip(CASES, picks, dev):
    t = time.time()
    src = verts[pick].astype(np.int32)
    rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=L, seed=2026, threads=min(128, os.cpu_count() or 8))
    same = bool(np.array_equal(sp, rp) and np.array_equal(sl, rl))
    ok &= same
    ss = {k: v for k, v in st["strategy_steps"].items() if v}
    print("p=%g q=%g L=%d: %d sampled walkers (longest start row %d) %s; oracle %.0f s; device kernel %.0f ms, setup %.0f ms, %s"
          % (p, q, L, len(src), int(degs.max()), "IDENTICAL" if same else "MISMATCH", time.time() - t, st["kernel_ms"], st["setup_ms"], ss), flush=True)
print("config 3 at full size:", "parity OK" if ok else "PARITY FAILED")
sys.exit(0 if ok else 1)
