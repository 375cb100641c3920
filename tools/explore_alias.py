"""Scratch: Mode A throughput per RMAT scale."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
for sc in [int(x) for x in sys.argv[1].split(",")]:
    eng = pkg.Engine(0); eng.generate_rmat(sc, 16 << sc, seed=42, weighted=True)
    nv, ne = eng.stats()
    t = time.time(); eng.walk(fetch=False, walk_length=1, seed=1, sampler="alias"); tb = time.time() - t
    print(f"scale {sc} weighted: V={nv} E={ne} alias build+first walk {tb:.2f}s", flush=True)
    for (p, q) in ((1.0, 1.0), (0.25, 4.0), (4.0, 0.5)):
        for nt in (False, True):
            st = eng.walk(fetch=False, walk_length=80, num_walks=1, seed=1, p=p, q=q, sampler="alias", nt_loads=nt)
            print(f"   alias p={p} q={q} nt={int(nt)}: steps={st['n_steps']} ms={st['kernel_ms']:.1f} -> {st['n_steps']/st['kernel_ms']/1e6:.3f} Gsteps/s  "
                  f"trials/step={st['trials']/max(st['n_steps'],1):.2f} reads/step={st['ent_reads']/max(st['n_steps'],1):.2f} fb={st['fallbacks']}", flush=True)
    eng.close()
