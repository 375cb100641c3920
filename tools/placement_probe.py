"""Scratch: does the 64 / 74 ms spread of the headline kernel follow the PROCESS or the ALLOCATION?  One process, the graph and
its tables rebuilt several times (new hipMalloc's each time), the first-order kernel timed after each rebuild.
usage: placement_probe.py [rebuilds] [scale]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _pkg
pkg = _pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sc = int(sys.argv[2]) if len(sys.argv) > 2 else 26
for k in range(n):
    eng = pkg.Engine(0)
    eng.generate_rmat(sc, 16 << sc, seed=42)
    ms = []
    for it in range(5):
        st = eng.walk(fetch=False, walk_length=80, num_walks=1, first_walk=it, seed=42)
        ms.append(st["kernel_ms"])
    print("rebuild %d: kernel_ms %s" % (k, " ".join("%.1f" % x for x in ms[1:])), flush=True)
    eng.close()
