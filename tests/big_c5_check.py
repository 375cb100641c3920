"""Verification of BASELINE config 5's stand-in at FULL size (directed unweighted RMAT-26 ef 27, p = 4 q = .5, Mode R, default sampler
selection: per-edge tables + the lean kernel): 20 000 sampled walkers at walkLength 80 (incl. the 20 highest-degree starts of a 2 000-vertex sample) are
compared bit for bit with the CPU ORACLE over the out-rows of every vertex on those walkers' DEVICE paths, rebuilt on the host from the
same (seed, edge index) stream (256 M lines at a time, stream order kept) — as tests/big_c3_check.py; a deviation walks the oracle into
a row that was not collected and shows as a mismatch.
Run by tests/test_gpu_full_size.py:  python tests/big_c5_check.py [scale] [edge factor]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _pkg
import oracle_py as oracle

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 27
n_edges = ef << scale
p, q, L = 4.0, 0.5, 80                   # the configuration's walkLength, 20 000 sampled walkers
N_SAMPLE = int(os.environ.get("SRW_C5_SAMPLE", "20000"))
pkg = _pkg.load()
t = time.time()
eng = pkg.Engine(0)
eng.generate_rmat(scale, n_edges, seed=42, weighted=False, directed=True)
nv, ne = eng.stats()
verts = eng.vertices()
rng = np.random.default_rng(3)
cand = np.arange(0, len(verts), max(1, len(verts) // 2000))
degs = np.array([len(eng.neighbors(int(verts[i]))[0]) for i in cand])
pick = np.unique(np.concatenate([cand[np.argsort(-degs)[:20]], rng.choice(len(verts), N_SAMPLE, replace=False)]))
src = verts[pick].astype(np.int32)
print("device graph: %d vertices, %d entries, %.0f s" % (nv, ne, time.time() - t), flush=True)
paths, lens, st = eng.walk(p=p, q=q, walk_length=L, seed=2026)
sp, sl = paths[pick].copy(), lens[pick].copy()
del paths, lens
on_path = np.zeros(1 << scale, dtype=bool)
on_path[sp[sp >= 0]] = True
t = time.time()
fs, fd = [], []
BLOCK = 1 << 28
for lo in range(0, n_edges, BLOCK):
    s_, d_ = oracle.rmat_edges(scale, min(BLOCK, n_edges - lo), seed=42, first=lo)
    keep = on_path[s_]                       # directed: a row is its vertex's out-lines
    fs.append(s_[keep]); fd.append(d_[keep])
    del s_, d_, keep
fs = np.concatenate(fs); fd = np.concatenate(fd)
g = oracle.Graph.from_coo(fs, fd, None, directed=True)
print("oracle out-rows of %d path vertices rebuilt from the edge stream (%d of %d lines kept), %.0f s" % (int(on_path.sum()), len(fs), n_edges, time.time() - t), flush=True)
t = time.time()
rp, rl, _ = g.walk(sources=src, p=p, q=q, walk_length=L, seed=2026, threads=min(128, os.cpu_count() or 8))
ok = bool(np.array_equal(sp, rp) and np.array_equal(sl, rl))
ss = {k: v for k, v in st["strategy_steps"].items() if v}
print("p=%g q=%g L=%d: %d sampled walkers (longest start row %d) %s; oracle %.0f s; device kernel %.0f ms, setup %.0f ms, %s"
      % (p, q, L, len(src), int(degs.max()), "IDENTICAL" if ok else "MISMATCH", time.time() - t, st["kernel_ms"], st["setup_ms"], ss), flush=True)
print("config 5 stand-in at full size:", "parity OK" if ok else "PARITY FAILED")
sys.exit(0 if ok else 1)
