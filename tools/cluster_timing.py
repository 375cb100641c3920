"""Scratch: vertex-sharded walk (in-process cluster, virtual shards on one GPU) vs the single-launch replicated kernel."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _pkg
pkg = _pkg.load()
sc = int(sys.argv[1]); worlds = [int(x) for x in sys.argv[2].split(",")]
p, q = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (1.0, 1.0)
with pkg.Engine(0) as eng:
    eng.generate_rmat(sc, 16 << sc, seed=42)
    eng.walk(fetch=False, walk_length=80, seed=1, p=p, q=q)
    t = time.time(); st = eng.walk(fetch=False, walk_length=80, num_walks=4, seed=1, p=p, q=q); dt = time.time() - t
    print(f"replicated kernel: scale {sc} p={p} q={q}: {st['n_steps']/dt/1e9:.2f} G steps/s ({dt*1e3:.0f} ms for 4 iterations)", flush=True)
for w in worlds:
    with pkg.Cluster([0] * w) as cl:
        cl.generate_rmat(sc, 16 << sc, seed=42)
        cl.walk(fetch=False, walk_length=80, num_walks=1, seed=1, p=p, q=q)
        for batch in (1, 4):
            t = time.time(); st = cl.walk(fetch=False, walk_length=80, num_walks=4, seed=1, p=p, q=q, batch=batch); dt = time.time() - t
            print(f"cluster world {w} batch {batch}: {st['n_steps']/dt/1e9:.2f} G steps/s ({dt*1e3:.0f} ms for 4 iterations)", flush=True)
