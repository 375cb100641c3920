"""One-off check of BASELINE config 4's shape (RMAT scale-27 / scale-26, vertex-partitioned, p = q = 1) through the
vertex-sharded protocol on ONE device: `world` sharded handles ("virtual shards") driven by srw_cluster_* — chunks, row
links across the shards, home-shard paths, the fused sample-and-bucket kernel, exactly the code an 8-GPU run executes,
minus the xGMI peer stores — must give the replicated single-launch kernel's paths bit for bit (which the -m gpu tests pin
to the oracle up to RMAT-20/24).  Not collected by pytest (tens of GB of HBM, ~10 GB of host memory):
    python tests/big_c4_check.py [scale] [world] [L]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _pkg

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
st80 = eng.walk(fetch=False, walk_length=80, seed=2026)
st80 = eng.walk(fetch=False, walk_length=80, seed=2026, first_walk=1)
print("replicated walk L=80: %.2f G steps/s (kernel %.1f ms)" % (st80["n_steps"] / st80["kernel_ms"] / 1e6, st80["kernel_ms"]), flush=True)
eng.close(); del eng
t = time.time()
ok = True
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
        c80 = cl.walk(fetch=False, walk_length=80, seed=2026, num_walks=2, batch=2)
        c80 = cl.walk(fetch=False, walk_length=80, seed=2026, num_walks=2, first_walk=2, batch=2)
        print("sharded walk L=80, 2 iterations as one population: %.2f G steps/s on ONE device (the %d shards' kernels run one after the other)"
              % (c80["n_steps"] / c80["kernel_ms"] / 1e6, world), flush=True)
    except pkg.SrwError as ex:      # all the shards' tables, buffers and two iterations of paths on ONE device: timing only
        print("sharded walk L=80 (timing only) skipped: %s" % str(ex)[:120], flush=True)
print("config 4 shape (RMAT-%d, world %d):" % (scale, world), "parity OK" if ok else "PARITY FAILED")
sys.exit(0 if ok else 1)
