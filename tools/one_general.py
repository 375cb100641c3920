"""Scratch: one general-kernel walk (for rocprofv3 counter passes).  usage: one_general.py SCALE P Q [w]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
sc, p, q = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
eng = pkg.Engine(0); eng.generate_rmat(sc, 16 << sc, seed=42, weighted=len(sys.argv) > 4)
st = eng.walk(fetch=False, walk_length=80, num_walks=1, seed=1, p=p, q=q, force_general=True)
print(f"p={p} q={q}: {st['n_steps']/st['kernel_ms']/1e3:.1f} Msteps/s kernel {st['kernel_ms']:.0f} ms")
