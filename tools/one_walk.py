"""Scratch: one graph, a few walk iterations of one kernel (for rocprofv3 passes).
usage: one_walk.py SCALE[w][d] P Q [reference|alias] [iters] [ef]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _pkg
pkg = _pkg.load()
spec, p, q = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
sampler = sys.argv[4] if len(sys.argv) > 4 else "reference"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 2
ef = int(sys.argv[6]) if len(sys.argv) > 6 else 16
import json
extra = json.loads(os.environ.get("SRW_ONE_WALK_KW", "{}"))      # e.g. {"edge_hash": false}
sc = int(spec.rstrip("wd"))
eng = pkg.Engine(0)
eng.generate_rmat(sc, ef << sc, seed=42, weighted="w" in spec, directed="d" in spec)
for it in range(iters):
    st = eng.walk(fetch=False, walk_length=80, num_walks=1, first_walk=it, seed=42, p=p, q=q, sampler=sampler, **extra)
    print(f"iter {it}: {st['n_steps']/st['kernel_ms']/1e3:.1f} Msteps/s kernel {st['kernel_ms']:.1f} ms setup {st['setup_ms']:.0f} ms steps {st['n_steps']} trials {st['trials']} reads {st['ent_reads']} handed {st['strategy_steps'].get('handed_over_walkers')} mix {st['strategy_steps']}", flush=True)
