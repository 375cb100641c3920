"""One-off check of BASELINE config 4's shape (RMAT scale-27 / scale-26, vertex-partitioned, p = q = 1) through the
vertex-sharded protocol on ONE device: `world` sharded handles ("virtual shards") driven by srw_cluster_* — chunks, row
links across the shards, home-shard paths, the fused sample-and-bucket kernel, exactly the code an 8-GPU run executes,
minus the xGMI peer stores — must give the replicated single-launch kernel's paths bit for bit — and ~1 500 sampled walkers (incl. the 20 highest-degree
starts of a 200 000-vertex sample) are compared with the CPU ORACLE: the adjacency rows of every vertex on those walkers'
device paths are rebuilt on the host from the same (seed, edge index) stream (the oracle's generator, 256 M lines at a time,
lines kept in stream order), and the oracle walks the same sources over them.  A deviation sends the oracle's walker into a row
that was not collected (partial or empty), which shows as a mismatch: the check cannot pass on the device's say-so.
Run by tests/test_gpu_full_size.py at BASELINE config 4's own size (RMAT-27, world 8):
    python tests/big_c4_check.py [scale] [world] [L]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import _pkg
import oracle_py as oracle

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = int(sys.argv[3]) if len(sys.argv) > 3 else 16
n_edges = 16 << scale
pkg = _pkg.load()
t = time.time()
eng = pkg.Engine(0)
eng.generate_rmat(scale, n_edges, seed=42)
nv, ne = eng.stats()
print("replicated: %d vertices, %d entries, graph %.1f s" % (nv, ne, time.time() - t), flush=True)
paths, lens, st = eng.walk(walk_length=L, seed=2026, first_walk=1)
print("replicated walk L=%d: %d steps, kernel %.1f ms" % (L, st["n_steps"], st["kernel_ms"]), flush=True)
# ---- the sampled walkers against the oracle -----------------------------------------------------------------------------
t = time.time()
verts = eng.vertices()
rng = np.random.default_rng(3)
# the highest-degree starts of a 2 000-vertex sample (row lengths read off the device) + 1 500 random ones
cand = np.arange(0, len(verts), max(1, len(verts) // 2000))
degs = np.array([len(eng.neighbors(int(verts[i]))[0]) for i in cand])
pick = np.unique(np.concatenate([cand[np.argsort(-degs)[:20]], rng.choice(len(verts), 1500, replace=False)]))
src = verts[pick].astype(np.int32)
on_path = np.zeros(1 << scale, dtype=bool)
for i in pick:
    on_path[paths[i, : lens[i]]] = True
fs, fd = [], []
BLOCK = 1 << 28
for lo in range(0, n_edges, BLOCK):
    s_, d_ = oracle.rmat_edges(scale, min(BLOCK, n_edges - lo), seed=42, first=lo)
    keep = on_path[s_] | on_path[d_]
    fs.append(s_[keep]); fd.append(d_[keep])
    del s_, d_, keep
fs = np.concatenate(fs); fd = np.concatenate(fd)
g = oracle.Graph.from_coo(fs, fd, None, directed=False)
rp, rl, _ = g.walk(sources=src, p=1.0, q=1.0, walk_length=L, seed=2026, first_walk=1, threads=min(64, os.cpu_count() or 8))
same_or = bool(np.array_equal(paths[pick], rp) and np.array_equal(lens[pick], rl))
ok = same_or
print("oracle: %d sampled walkers (longest start row %d), rows of %d path vertices rebuilt from the edge stream (%d of %d lines kept): %s (%.0f s)"
      % (len(src), int(degs.max()), int(on_path.sum()), len(fs), n_edges, "IDENTICAL" if same_or else "MISMATCH", time.time() - t), flush=True)
del g, fs, fd, on_path, rp, rl
quick = bool(os.environ.get("SRW_CHECK_QUICK"))            # the driver's suite: parity only, no L = 80 timing walks
if not quick:
    st80 = eng.walk(fetch=False, walk_length=80, seed=2026)
    st80 = eng.walk(fetch=False, walk_length=80, seed=2026, first_walk=1)
    print("replicated walk L=80: %.2f G steps/s (kernel %.1f ms)" % (st80["n_steps"] / st80["kernel_ms"] / 1e6, st80["kernel_ms"]), flush=True)
eng.close(); del eng
t = time.time()
with pkg.Cluster([0] * world, membership=False) as cl:      # config 4 is p = q = 1: SRW_CFG_NO_MEMBERSHIP
    cl.generate_rmat(scale, n_edges, seed=42)
    assert cl.stats() == (nv, ne), (cl.stats(), nv, ne)
    print("cluster of %d virtual shards: graph %.1f s" % (world, time.time() - t), flush=True)
    cp, clens, cst = cl.walk(walk_length=L, seed=2026, first_walk=1)
    same = bool(np.array_equal(clens, lens) and np.array_equal(cp, paths) and cst["n_steps"] == st["n_steps"])
    ok &= same
    print("sharded walk L=%d, world %d: %s (%d walkers, %d steps; super-steps %.1f ms; overflow retries %s)"
          % (L, world, "IDENTICAL" if same else "MISMATCH", len(clens), cst["n_steps"], cst["kernel_ms"], cst.get("overflow_retries")), flush=True)
    del cp, clens, paths, lens
    try:
        if quick:
            raise pkg.SrwError(0, "quick mode")
        c80 = cl.walk(fetch=False, walk_length=80, seed=2026, num_walks=2, batch=2)
        c80 = cl.walk(fetch=False, walk_length=80, seed=2026, num_walks=2, first_walk=2, batch=2)
        print("sharded walk L=80, 2 iterations as one population: %.2f G steps/s on ONE device (the %d shards' kernels run one after the other)"
              % (c80["n_steps"] / c80["kernel_ms"] / 1e6, world), flush=True)
    except pkg.SrwError as ex:      # all the shards' tables, buffers and two iterations of paths on ONE device: timing only
        print("sharded walk L=80 (timing only) skipped: %s" % str(ex)[:120], flush=True)
print("config 4 shape (RMAT-%d, world %d):" % (scale, world), "parity OK" if ok else "PARITY FAILED")
sys.exit(0 if ok else 1)
