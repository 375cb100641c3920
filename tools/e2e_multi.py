import os, sys, time, shutil, tempfile
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import _pkg
pkg = _pkg.load()
eng = pkg.Engine(0)
eng.generate_rmat(25, 16 << 25, seed=42)
eng.walk(fetch=False, walk_length=80, num_walks=1, seed=42)
for parts, nw in ((200, 1), (200, 1), (200, 3), (1, 3), (200, 3)):
    d = tempfile.mkdtemp(prefix="srw_e2e_", dir="/tmp")
    t0 = time.time(); st, _ = eng.walk_and_save(os.path.join(d, "out"), n_parts=parts, walk_length=80, num_walks=nw, seed=42, first_walk=1, device_format=True); dt = time.time() - t0
    nb = sum(os.path.getsize(os.path.join(d, "out", "path", f)) for f in os.listdir(os.path.join(d, "out", "path")))
    print("parts %3d numWalks %d: %.2f s (%.2f s per iteration), %.1f GB of text, %.1f GB/s, %.2e walk-steps/s" % (parts, nw, dt, dt / nw, nb / 1e9, nb / dt / 1e9, st["n_steps"] / dt), flush=True)
    shutil.rmtree(d)
