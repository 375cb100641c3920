"""Vertex-sharded biased walk (per-edge tables on the shards) against the replicated kernel on the same graph.
usage: shard_tables_bench.py scale ef weighted directed p q [worlds...]   (all shards on device 0)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
P = _pkg.load()
scale, ef, weighted, directed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
p, q = float(sys.argv[5]), float(sys.argv[6])
worlds = [int(x) for x in sys.argv[7:]] or [1]
L = 80
if not os.environ.get("SKIP_REPLICATED"):
    with P.Engine(device=0) as e:
        e.generate_rmat(scale, ef << scale, seed=42, weighted=bool(weighted), directed=bool(directed))
        st = e.walk(fetch=False, p=p, q=q, walk_length=L, seed=1)
        st = e.walk(fetch=False, p=p, q=q, walk_length=L, seed=2)
        print("replicated: %.3e steps/s (kernel %.1f ms, setup of the first call %.1f ms) tables %d" % (
            st["n_steps"] / st["kernel_ms"] * 1e3, st["kernel_ms"], st["setup_ms"], st["edge_tables"]), flush=True)
for w in worlds:
    with P.Cluster([0] * w) as cl:
        t0 = time.perf_counter()
        cl.generate_rmat(scale, ef << scale, seed=42, weighted=bool(weighted), directed=bool(directed))
        t1 = time.perf_counter()
        st = cl.walk(fetch=False, p=p, q=q, walk_length=1, seed=1)          # tables outside the timed walks
        t2 = time.perf_counter()
        for v in [x for x in os.environ.get("SRW_AB", "").split(",") if x]:      # A / B of the table step's variants on one set of tables
            os.environ["SRW_SH_BATCH"], os.environ["SRW_SH_GRAB"] = (v.split("/") + ["16"])[:2]      # "variant" or "variant/records per grab"
            st = cl.walk(fetch=False, p=p, q=q, walk_length=L, num_walks=1, seed=2, batch=1)
            print("world %d SRW_SH_BATCH=%s: %.3e steps/s (%.1f ms per iteration) steps %d" % (w, v, st["n_steps"] / st["kernel_ms"] * 1e3, st["kernel_ms"], st["n_steps"]), flush=True)
        if os.environ.get("SRW_AB"):
            continue
        for B in (1, 2):
            st = cl.walk(fetch=False, p=p, q=q, walk_length=L, num_walks=B, seed=2, batch=B)
            ss = st["strategy_steps"]
            print("world %d batch %d: %.3e steps/s (%.1f ms per iteration); load %.1f s, tables %.1f s; tables %d (%.2f GB), table steps %d mask steps %d handed %d fallbacks %d" % (
                w, B, st["n_steps"] / st["kernel_ms"] * 1e3, st["kernel_ms"] / B, t1 - t0, t2 - t1, st["edge_tables"], st["edge_table_bytes"] / 1e9,
                ss["edge_table"], ss["edge_mask"], ss["handed_over_walkers"], st["fallbacks"]), flush=True)
