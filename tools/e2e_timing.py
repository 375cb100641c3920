"""Scratch: srw_walk_and_save end to end on the headline graph by number of part files (SRW_TIMING=1 for the phases).
usage: e2e_timing.py [scale] [parts ...]"""
import os, sys, time, shutil, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _pkg
pkg = _pkg.load()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
eng = pkg.Engine(0)
eng.generate_rmat(scale, 16 << scale, seed=42)
eng.walk(fetch=False, walk_length=80, num_walks=1, seed=42)
for parts in [int(x) for x in sys.argv[2:]] or [1, 200]:
    d = tempfile.mkdtemp(prefix="srw_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    t0 = time.time(); st, _ = eng.walk_and_save(os.path.join(d, "out"), n_parts=parts, walk_length=80, num_walks=1, seed=42, first_walk=1, device_format=True); dt = time.time() - t0
    nb = sum(os.path.getsize(os.path.join(d, "out", "path", f)) for f in os.listdir(os.path.join(d, "out", "path")))
    print("parts %3d: %.2f s, %.1f GB of text, %.1f GB/s, %.2e walk-steps/s" % (parts, dt, nb / 1e9, nb / dt / 1e9, st["n_steps"] / dt), flush=True)
    shutil.rmtree(d)
