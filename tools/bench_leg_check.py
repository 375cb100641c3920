import sys, json
sys.path.insert(0, "/root/repo")
import torch, _pkg, bench
pkg = _pkg.load(); torch.zeros(1, device="cuda")
for (n, p, q) in [("c3q1", 0.25, 1.0), ("c3", 0.25, 4.0)]:
    o = bench.run_sharded_biased_world1(pkg, 0, n, 24, 16, True, False, p, q, None, batch2=True)
    print(n, "%.3e" % o["value"], round(o["ms_per_step"], 1), json.dumps(o.get("two_iterations_per_population")), flush=True)
