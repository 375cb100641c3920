"""Scratch: the vertex-sharded walk on `world` shards of device 0 (for rocprofv3 passes).
usage: one_shard_walk.py SCALE[w][d] P Q WORLD [iters] [ef] [batch]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _pkg
pkg = _pkg.load()
spec, p, q, world = sys.argv[1], float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ef = int(sys.argv[6]) if len(sys.argv) > 6 else 16
batch = int(sys.argv[7]) if len(sys.argv) > 7 else 1
sc = int(spec.rstrip("wd"))
cl = pkg.Cluster([0] * world, membership=(q != 1.0))
cl.generate_rmat(sc, ef << sc, seed=42, weighted="w" in spec, directed="d" in spec)
for it in range(iters):
    st = cl.walk(fetch=False, walk_length=80, num_walks=batch, first_walk=it * batch, seed=42, p=p, q=q, batch=batch)
    print(f"iter {it}: {st['n_steps']/st['kernel_ms']/1e3:.1f} Msteps/s wall {st['kernel_ms']:.1f} ms steps {st['n_steps']} fallbacks {st['fallbacks']} strategies {st['strategy_steps']}", flush=True)
