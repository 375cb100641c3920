"""Scratch: throughput of the general (any p,q) kernel per RMAT scale."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
for sc in [int(x) for x in sys.argv[1].split(",")]:
    eng = pkg.Engine(0); eng.generate_rmat(sc, 16 << sc, seed=42, weighted=len(sys.argv) > 2)
    nv, ne = eng.stats()
    print(f"scale {sc}: V={nv} E={ne}", flush=True)
    eng.walk(fetch=False, walk_length=2, seed=1, force_general=True)
    for (p, q) in ((1.0, 1.0), (0.5, 1.0), (0.25, 4.0), (4.0, 0.5)):
        st = eng.walk(fetch=False, walk_length=80, num_walks=1, seed=1, p=p, q=q, force_general=True)
        print(f"   gen p={p} q={q}: steps={st['n_steps']} ms={st['kernel_ms']:.1f} -> {st['n_steps']/st['kernel_ms']/1e3:.1f} Msteps/s  "
              f"{st['sum_deg_curr']/st['kernel_ms']/1e6:.1f} Gentries/s ({st['sum_deg_curr']/max(st['n_steps'],1):.0f}/step) "
              f"alg {(st['n_steps']*20+st['sum_deg_curr']*8+st['sum_deg_prev']*4)/st['kernel_ms']/1e6:.0f} GB/s fb={st['fallbacks']}", flush=True)
    eng.close()
