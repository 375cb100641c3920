"""Scratch exploration: build + walk timings per RMAT scale (not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
scales = [int(x) for x in sys.argv[1].split(",")]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 80
for sc in scales:
    eng = pkg.Engine(0)
    t0 = time.time(); eng.generate_rmat(sc, 16 << sc, seed=42); t1 = time.time()
    nv, ne = eng.stats()
    st = eng.walk(fetch=False, walk_length=L, num_walks=1, seed=1); t2 = time.time()
    print(f"scale {sc}: V={nv} E={ne} build={t1-t0:.2f}s first_walk_call(incl tables)={t2-t1:.2f}s", flush=True)
    for it in range(6):
        st = eng.walk(fetch=False, walk_length=L, num_walks=1, first_walk=it, seed=1, nt_loads=bool(it & 1))
        print(f"   fo  nt={it&1} iter{it}: steps={st['n_steps']} ms={st['kernel_ms']:.2f} -> {st['n_steps']/st['kernel_ms']/1e6:.3f} Gsteps/s reads/step={st['ent_reads']/max(st['n_steps'],1):.2f}", flush=True)
    if sc <= 22 and len(sys.argv) > 3:
        for (p, q) in ((1.0, 1.0), (0.25, 4.0)):
            t = time.time()
            st = eng.walk(fetch=False, walk_length=L, num_walks=1, seed=1, p=p, q=q, force_general=True)
            print(f"   gen p={p} q={q}: steps={st['n_steps']} ms={st['kernel_ms']:.1f} -> {st['n_steps']/st['kernel_ms']/1e3:.1f} Msteps/s  sumdeg={st['sum_deg_curr']} ({st['sum_deg_curr']/max(st['n_steps'],1):.0f}/step) fb={st['fallbacks']}", flush=True)
    eng.close()
