"""EVERY walker of one iteration at walkLength 80, at FULL size, through independent samplers: the default
table path (per-edge tables + the table kernel the geometry selects: one walker per lane at config 3, one per wave on config 5's stand-in),
the other table kernel, and the ON-THE-FLY samplers of the general kernel with the tables off (edge_tables=False: intersections of the
two rows at every step — what the tables precompute) — all walkers, all 82 slots compared.  The rare paths are asserted to be among the
compared walkers: draws within rounding distance of a CDF boundary (strategy 'chain' > 0: the tie list + chain kernels, hand-overs).
The paths themselves are pinned to the CPU oracle by tests/big_c3_check.py / big_c5_check.py (20 000 sampled walkers at the same L and seed).
Run by tests/test_gpu_full_size.py:  python tests/big_every_walker_check.py c3|c5 [scale]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import _pkg

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
if cfg == "c3":
    scale, ef, weighted, directed, p, q = (int(sys.argv[2]) if len(sys.argv) > 2 else 24), 16, True, False, 0.25, 4.0
else:
    scale, ef, weighted, directed, p, q = (int(sys.argv[2]) if len(sys.argv) > 2 else 26), 27, False, True, 4.0, 0.5
L = 80
pkg = _pkg.load()
t = time.time()
eng = pkg.Engine(0)
eng.generate_rmat(scale, ef << scale, seed=42, weighted=weighted, directed=directed)
print("device graph: %d vertices, %d entries, %.0f s" % (*eng.stats(), time.time() - t), flush=True)


def walk(env=None, **kw):
    for k, v in (env or {}).items(): os.environ[k] = v
    try:
        t0 = time.time()
        out = eng.walk(p=p, q=q, walk_length=L, seed=2026, **kw)
        return out + (time.time() - t0,)
    finally:
        for k in (env or {}): os.environ.pop(k, None)


ref_p, ref_l, st, dt = walk()
mix = {k: v for k, v in st["strategy_steps"].items() if v}
print("default table path: %d walkers, %d steps, kernel %.0f ms (setup %.0f ms), %s" % (len(ref_l), st["n_steps"], st["kernel_ms"], st["setup_ms"], mix), flush=True)
ok = st["strategy_steps"]["edge_table"] > 0 and st["strategy_steps"]["edge_mask"] > 0
if not (st["strategy_steps"]["chain"] > 0 and st["strategy_steps"]["handed_over_walkers"] > 0):
    print("no boundary draw among the walkers: the tie paths are NOT covered"); ok = False
# the wave kernel's mode switch is read per call: -1 = one walker per wave, 2 = one walker per lane (table steps per lane)
for name, env, kw in (("one walker per wave (k_walk_tables)", {"SRW_TABLE_LANES": "-1"}, {}),
                      ("one walker per lane (k_walk_tables_lanes, mode 2)", {"SRW_TABLE_LANES": "2"}, {}),
                      ("tables off: the on-the-fly samplers of k_walk_general", {}, {"edge_tables": False})):
    pp, ll, s2, dt = walk(env, **kw)
    same = bool(np.array_equal(ll, ref_l) and np.array_equal(pp, ref_p))
    ok &= same
    if kw: ok &= s2["strategy_steps"]["edge_table"] == 0 and s2["edge_tables"] == 0
    print("%s: every walker (%d) x %d slots %s; kernel %.0f ms, setup %.0f ms, %s" % (name, len(ll), L + 2, "IDENTICAL" if same else "MISMATCH", s2["kernel_ms"], s2["setup_ms"],
                                                                                   {k: v for k, v in s2["strategy_steps"].items() if v}), flush=True)
    if not same:
        bad = np.nonzero((pp != ref_p).any(axis=1) | (ll != ref_l))[0]
        print("  differing walkers:", bad[:8], "of", len(bad))
    del pp, ll
print("every walker at full size (%s):" % cfg, "parity OK" if ok else "PARITY FAILED")
sys.exit(0 if ok else 1)
