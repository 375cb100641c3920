"""Scratch: world-1 in-process cluster walk only (batch 1 and 4), for library variants (SRW_LIB)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _pkg
pkg = _pkg.load()
sc = int(sys.argv[1]); worlds = [int(x) for x in sys.argv[2].split(",")]
for w in worlds:
    with pkg.Cluster([0] * w) as cl:
        cl.generate_rmat(sc, 16 << sc, seed=42)
        cl.walk(fetch=False, walk_length=80, num_walks=1, seed=1)
        for batch in (1, 4):
            t = time.time(); st = cl.walk(fetch=False, walk_length=80, num_walks=4, seed=1, batch=batch); dt = time.time() - t
            print(f"{os.environ.get('SRW_LIB', 'default')[-12:]} world {w} batch {batch}: {st['n_steps']/dt/1e9:.2f} G steps/s ({dt*1e3:.0f} ms for 4 iterations)", flush=True)
