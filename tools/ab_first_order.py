"""Within-process interleaved A/B of first-order kernel variants (scratch tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
sc = int(sys.argv[1]); rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = pkg.Engine(0); eng.generate_rmat(sc, 16 << sc, seed=42)
variants = [dict(nt_loads=True, occ=0), dict(nt_loads=True, occ=7), dict(nt_loads=True, occ=8), dict(nt_loads=False, occ=0)]
eng.walk(fetch=False, walk_length=80, seed=1)
res = {i: [] for i in range(len(variants))}
for r in range(rounds):
    for i, v in enumerate(variants):
        st = eng.walk(fetch=False, walk_length=80, num_walks=1, first_walk=r, seed=1, **v)
        res[i].append(st["kernel_ms"])
for i, v in enumerate(variants):
    ms = sorted(res[i]); print(v, "median %.2f ms min %.2f  -> %.2f Gsteps/s" % (ms[len(ms)//2], ms[0], st["n_steps"] / ms[len(ms)//2] / 1e6))
