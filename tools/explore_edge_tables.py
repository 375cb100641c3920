"""Scratch: general kernel with / without the per-edge bias tables.  usage: explore_edge_tables.py SCALE[w][d] P Q [ef]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("SRW_TIMING", "1")
import _pkg
pkg = _pkg.load()
spec, p, q = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
ef = int(sys.argv[4]) if len(sys.argv) > 4 else 16
sc = int(spec.rstrip("wd")); weighted = "w" in spec; directed = "d" in spec
eng = pkg.Engine(0)
t = time.time(); eng.generate_rmat(sc, ef << sc, seed=42, weighted=weighted, directed=directed)
print(f"scale {sc} ef {ef} weighted={weighted} directed={directed}: {eng.stats()} built in {time.time()-t:.1f} s", flush=True)
for label, kw in (("tables", {}), ("tables again", {}), ("no tables", {"edge_tables": False})):
    if label == "no tables" and len(sys.argv) > 5 and sys.argv[5] == "skip": continue
    t = time.time()
    st = eng.walk(fetch=False, walk_length=80, num_walks=1, seed=1, p=p, q=q, **kw)
    wall = time.time() - t
    ss = {k: v for k, v in st["strategy_steps"].items() if v}
    print(f"[{label}] p={p} q={q}: {st['n_steps']/st['kernel_ms']/1e3:.1f} Msteps/s kernel {st['kernel_ms']:.0f} ms setup {st['setup_ms']:.0f} ms wall {wall:.1f} s "
          f"tables {st['edge_tables']} ({st['edge_table_bytes']/1e9:.2f} GB) fb {st['fallbacks']} {ss}", flush=True)
