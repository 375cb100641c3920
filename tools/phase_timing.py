"""Scratch (needs a library built with -DSRW_PHASE_TIMING): where the general kernel's wave-cycles go."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
sc = int(sys.argv[1]); weighted = len(sys.argv) > 2
eng = pkg.Engine(0); eng.generate_rmat(sc, 16 << sc, seed=42, weighted=weighted)
eng.walk(fetch=False, walk_length=2, seed=1, force_general=True)
for (p, q) in ((0.25, 4.0), (4.0, 0.5)):
    st = eng.walk(fetch=False, walk_length=80, num_walks=1, seed=1, p=p, q=q, force_general=True)
    ms = st["kernel_ms"]
    # wall_clock64 ticks at 100 MHz; counters were >> 10
    tick = 1024 / 100e6 * 1e3  # ms of one wave per counter unit
    print(f"p={p} q={q}: {st['n_steps']/ms/1e3:.1f} Msteps/s, kernel {ms:.0f} ms", flush=True)
