"""Scratch (GPU box): device formatter vs host formatter on an output beyond 2^31 bytes (RMAT-23, one iteration)."""
import sys, os, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg
pkg = _pkg.load()
eng = pkg.Engine(0); eng.generate_rmat(23, 16 << 23, seed=42)
for name, dev in (("h", False), ("d", True)):
    out = "/tmp/big_" + name
    subprocess.run(["rm", "-rf", out])
    t = time.time()
    st, dead = eng.walk_and_save(out, n_parts=3, walk_length=80, num_walks=1, seed=5, device_format=dev)
    print(name, "%.2f s" % (time.time() - t), st["n_steps"], [os.path.getsize(out + "/path/part-%05d" % k) for k in range(3)], flush=True)
rc = [subprocess.run(["cmp", "/tmp/big_h/path/part-%05d" % k, "/tmp/big_d/path/part-%05d" % k]).returncode for k in range(3)]
print("identical" if rc == [0, 0, 0] else "DIFFERENT %s" % rc)
subprocess.run(["rm", "-rf", "/tmp/big_h", "/tmp/big_d"])
