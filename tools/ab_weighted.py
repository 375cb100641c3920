"""A/B: weighted p=q=1 through the 16-byte compact records vs the 32-byte exact records."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
sc = int(sys.argv[1]) if len(sys.argv) > 1 else 24
e = pkg.Engine(0); e.generate_rmat(sc, 16 << sc, seed=42, weighted=True)
for rep in range(3):
    for compact in (True, False):
        st = e.walk(fetch=False, walk_length=80, first_walk=rep, compact=compact)
        print("rep %d %s: %.2f ms  %.2f Gsteps/s  reads/step %.3f" % (rep, "compact 16B" if compact else "exact 32B  ", st["kernel_ms"],
              st["n_steps"] / st["kernel_ms"] / 1e6, st["ent_reads"] / st["n_steps"]), flush=True)
