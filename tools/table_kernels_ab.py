"""Scratch (round 6): the table-walk kernels against each other on one graph, in ONE process (the tables are built once): every walker of
one iteration compared with the one-walker-per-wave kernel, then kernel times alternated.
usage: table_kernels_ab.py SCALE[w][d] P Q [iters] [ef] [L] [variants, comma separated: waves,groups,rounds,lanes0,lanes1,lanes2,lanes3]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import _pkg
pkg = _pkg.load()
spec, p, q = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ef = int(sys.argv[5]) if len(sys.argv) > 5 else 16
L = int(sys.argv[6]) if len(sys.argv) > 6 else 80
variants = (sys.argv[7] if len(sys.argv) > 7 else "waves,lanes0,lanes3").split(",")
sc = int(spec.rstrip("wd"))
eng = pkg.Engine(0)
eng.generate_rmat(sc, ef << sc, seed=42, weighted="w" in spec, directed="d" in spec)
def run(v, fetch, it):
    for k in ("SRW_TABLE_GROUPS", "SRW_TABLE_LANES", "SRW_TABLE_ROUNDS"): os.environ.pop(k, None)
    if v == "groups": os.environ["SRW_TABLE_GROUPS"] = "1"
    elif v == "rounds": os.environ["SRW_TABLE_ROUNDS"] = "1"
    elif v.startswith("lanes"): os.environ["SRW_TABLE_LANES"] = v[5:]
    elif v == "waves": os.environ["SRW_TABLE_LANES"] = "-1"
    r = eng.walk(fetch=fetch, walk_length=L, num_walks=1, first_walk=it, seed=42, p=p, q=q)
    for k in ("SRW_TABLE_GROUPS", "SRW_TABLE_LANES", "SRW_TABLE_ROUNDS"): os.environ.pop(k, None)
    return r
ref = None; ok = True
for v in ["waves"] + [x for x in variants if x != "waves"]:
    pa, la, sa = run(v, True, 0)
    print("%-7s setup %.0f ms kernel %.1f ms steps %d handed %s mix %s trials %d reads %d" % (v, sa["setup_ms"], sa["kernel_ms"], sa["n_steps"], sa["strategy_steps"].get("handed_over_walkers"), {k: x for k, x in sa["strategy_steps"].items() if x}, sa["trials"], sa["ent_reads"]), flush=True)
    if ref is None: ref = (pa, la)
    else:
        same = bool(np.array_equal(la, ref[1]) and np.array_equal(pa, ref[0]))
        ok &= same
        print("  every walker (%d) of iteration 0 against waves: %s" % (len(la), "IDENTICAL" if same else "MISMATCH"), flush=True)
        if not same:
            bad = np.nonzero((pa != ref[0]).any(axis=1) | (la != ref[1]))[0]
            print("  differing walkers:", bad[:8], "of", len(bad))
            for w in bad[:3]:
                d = np.nonzero(pa[w] != ref[0][w])[0]
                print("   walker", w, "first diff at slot", d[:1], v, pa[w][max(0, d[0] - 2):d[0] + 3], "waves", ref[0][w][max(0, d[0] - 2):d[0] + 3], "lens", la[w], ref[1][w])
        del pa, la
for it in range(1, iters + 1):
    print("iter %d: " % it + "   ".join("%s %.1f ms" % (v, run(v, False, it)["kernel_ms"]) for v in variants), flush=True)
sys.exit(0 if ok else 1)
