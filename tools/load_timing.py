"""Scratch: srw_load_edgelist of a big text file, timed.  usage: load_timing.py FILE [repeat]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _pkg
pkg = _pkg.load()
fn = sys.argv[1]; rep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
size = os.path.getsize(fn)
for r in range(rep):
    eng = pkg.Engine(0)
    t0 = time.time(); eng.load_edgelist(fn, directed=False); dt = time.time() - t0
    print("load %d: %.2f s = %.2f GB/s of text, stats %s" % (r, dt, size / dt / 1e9, eng.stats()), flush=True)
    eng.close()
