"""Scratch: does BASELINE config 4's graph (RMAT-27, 2.1 B edge lines, 4.3 B entries) fit and run on ONE MI355X?"""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
sc = int(sys.argv[1]) if len(sys.argv) > 1 else 27
eng = pkg.Engine(0)
t = time.time(); eng.generate_rmat(sc, 16 << sc, seed=42); print("build %.1f s" % (time.time() - t), eng.stats(), flush=True)
print(subprocess.run("rocm-smi --showmeminfo vram | grep Used", shell=True, capture_output=True, text=True).stdout, flush=True)
t = time.time(); st = eng.walk(fetch=False, walk_length=80, seed=1); print("first walk (tables) %.1f s" % (time.time() - t), flush=True)
for it in range(3):
    st = eng.walk(fetch=False, walk_length=80, first_walk=it, seed=1)
    print("iter %d: %d steps %.1f ms -> %.2f Gsteps/s" % (it, st["n_steps"], st["kernel_ms"], st["n_steps"] / st["kernel_ms"] / 1e6), flush=True)
print(subprocess.run("rocm-smi --showmeminfo vram | grep Used", shell=True, capture_output=True, text=True).stdout, flush=True)
