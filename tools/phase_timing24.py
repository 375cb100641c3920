import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import _pkg
pkg = _pkg.load()
eng = pkg.Engine(0); eng.generate_rmat(24, 16 << 24, seed=42, weighted=True)
st = eng.walk(fetch=False, walk_length=80, num_walks=1, seed=1, p=0.25, q=4.0)
print(f"p=.25 q=4: {st['n_steps']/st['kernel_ms']/1e3:.2f} Msteps/s, kernel {st['kernel_ms']:.0f} ms", flush=True)
