"""Scratch: the embedding stage behind a walk, timed (words per second of one training iteration).
usage: w2v_timing.py SCALE [num_walks] [iterations] [dim] [window]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _pkg
pkg = _pkg.load()
sc = int(sys.argv[1]); nw = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dim = int(sys.argv[4]) if len(sys.argv) > 4 else 128
window = int(sys.argv[5]) if len(sys.argv) > 5 else 10
eng = pkg.Engine(0)
eng.generate_rmat(sc, 16 << sc, seed=42)
st = eng.walk(fetch=False, walk_length=80, num_walks=nw, seed=42, p=1.0, q=1.0)
words = st["n_steps"] + st["n_walkers"] if "n_walkers" in st else st["n_steps"]
for k in (0, iters):
    t0 = time.time()
    ids, vec = eng.w2v_fit_device(dim=dim, window=window, iterations=k, seed=7)
    dt = time.time() - t0
    print(f"scale {sc} numWalks {nw}: fit with {k} iterations {dt:.2f} s, vocab {len(ids)}, ~{words/1e6:.1f} M words", flush=True)
    if k == 0: base = dt
if iters: print(f"one training iteration: {(dt - base)/iters:.2f} s = {words*iters/(dt-base)/1e6:.1f} M words/s", flush=True)
