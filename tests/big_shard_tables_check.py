"""One-off check at size: the vertex-sharded biased walk (per-edge tables on the shards: tree tables, 16-bit level 0, chunk masks,
finer tables, two walker populations) against the replicated kernel on the same graph — every walker, bit for bit — which
tests/big_c3_check.py / big_c5_check.py pin to the oracle.  Not collected by pytest (tens of GB of HBM per shard set):
    python tests/big_shard_tables_check.py [scale] [ef] [weighted] [directed] [p] [q] [worlds...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _pkg
P = _pkg.load()
a = sys.argv[1:]
scale, ef, weighted, directed = int(a[0]) if a else 22, int(a[1]) if len(a) > 1 else 16, int(a[2]) if len(a) > 2 else 1, int(a[3]) if len(a) > 3 else 0
p, q = float(a[4]) if len(a) > 4 else 0.25, float(a[5]) if len(a) > 5 else 4.0
worlds = [int(x) for x in a[6:]] or [1, 2]
L, NW = 20, 2
with P.Engine(device=0) as e:
    e.generate_rmat(scale, ef << scale, seed=42, weighted=bool(weighted), directed=bool(directed))
    paths, lens, st = e.walk(p=p, q=q, walk_length=L, num_walks=NW, seed=5)
    print("replicated: %d walkers, %d steps, kernel %.0f ms, tables %.1f GB" % (len(lens), st["n_steps"], st["kernel_ms"], st["edge_table_bytes"] / 1e9), flush=True)
ok = True
# SRW_CHECK_QUICK=1 (the driver's suite): one population at world 1, two at the larger worlds — both drivers' forms, half the table builds
quick = bool(os.environ.get("SRW_CHECK_QUICK"))
for w in worlds:
    for pops in ((("1",) if w == 1 else ("2",)) if quick else ("1", "2")):
        os.environ["SRW_CLUSTER_POPULATIONS"] = pops
        with P.Cluster([0] * w) as cl:
            cl.generate_rmat(scale, ef << scale, seed=42, weighted=bool(weighted), directed=bool(directed))
            t0 = time.time()
            cp, cl_lens, cst = cl.walk(p=p, q=q, walk_length=L, num_walks=NW, seed=5, batch=NW)
            same = bool(np.array_equal(cl_lens, lens) and np.array_equal(cp, paths) and cst["n_steps"] == st["n_steps"])
            ok &= same
            ss = cst["strategy_steps"]
            print("world %d, %s population(s): %s (%.0f s; tables %.1f GB on the shards; table steps %d, mask steps %d)" % (
                w, pops, "IDENTICAL" if same else "MISMATCH", time.time() - t0, cst["edge_table_bytes"] / 1e9, ss["edge_table"], ss["edge_mask"]), flush=True)
print("sharded tables at RMAT-%d:" % scale, "parity OK" if ok else "PARITY FAILED")
sys.exit(0 if ok else 1)
