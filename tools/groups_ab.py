"""Scratch (round 6): the group table kernel (walk_groups.hip) against the one-walker-per-wave kernel on one graph, in ONE process
(the tables are built once): every walker of one iteration compared, then kernel times alternated.
usage: groups_ab.py SCALE[w][d] P Q [iters] [ef] [L]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import _pkg
pkg = _pkg.load()
spec, p, q = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ef = int(sys.argv[5]) if len(sys.argv) > 5 else 16
L = int(sys.argv[6]) if len(sys.argv) > 6 else 80
sc = int(spec.rstrip("wd"))
eng = pkg.Engine(0)
eng.generate_rmat(sc, ef << sc, seed=42, weighted="w" in spec, directed="d" in spec)
def run(groups, fetch, it):
    os.environ["SRW_TABLE_GROUPS"] = "1" if groups else "0"
    return eng.walk(fetch=fetch, walk_length=L, num_walks=1, first_walk=it, seed=42, p=p, q=q)
t = time.time()
pa, la, sa = run(True, True, 0)
print("groups: setup %.0f ms kernel %.1f ms steps %d handed %s mix %s" % (sa["setup_ms"], sa["kernel_ms"], sa["n_steps"], sa["strategy_steps"].get("handed_over_walkers"), {k: v for k, v in sa["strategy_steps"].items() if v}), flush=True)
pb, lb, sb = run(False, True, 0)
print("waves : setup %.0f ms kernel %.1f ms steps %d handed %s mix %s" % (sb["setup_ms"], sb["kernel_ms"], sb["n_steps"], sb["strategy_steps"].get("handed_over_walkers"), {k: v for k, v in sb["strategy_steps"].items() if v}), flush=True)
same = bool(np.array_equal(la, lb) and np.array_equal(pa, pb))
print("every walker (%d) of iteration 0: %s; trials %d / %d, reads %d / %d" % (len(la), "IDENTICAL" if same else "MISMATCH", sa["trials"], sb["trials"], sa["ent_reads"], sb["ent_reads"]), flush=True)
if not same:
    bad = np.nonzero((pa != pb).any(axis=1) | (la != lb))[0]
    print("first differing walkers:", bad[:8], "of", len(bad))
    for w in bad[:3]:
        d = np.nonzero(pa[w] != pb[w])[0]
        print(" walker", w, "first diff at slot", d[:1], "groups", pa[w][max(0, d[0] - 2):d[0] + 3], "waves", pb[w][max(0, d[0] - 2):d[0] + 3], "lens", la[w], lb[w])
del pa, pb
for it in range(1, iters + 1):
    a = run(True, False, it); b = run(False, False, it)
    print("iter %d: groups %.1f ms (%.3e steps/s)   waves %.1f ms (%.3e steps/s)" % (it, a["kernel_ms"], a["n_steps"] / a["kernel_ms"] * 1e3, b["kernel_ms"], b["n_steps"] / b["kernel_ms"] * 1e3), flush=True)
os.environ.pop("SRW_TABLE_GROUPS", None)
sys.exit(0 if same else 1)
