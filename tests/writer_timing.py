import sys, time, os, shutil
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, _pkg
pkg = _pkg.load()
n, stride = 6458340, 82
rng = np.random.default_rng(0)
paths = rng.integers(0, 1 << 20, size=(n, stride), dtype=np.int32); lens = np.full(n, 82, dtype=np.int32)
for parts in (1, 8):
    shutil.rmtree("/tmp/wt_out", ignore_errors=True)
    t = time.time(); pkg.save_paths(paths, lens, "/tmp/wt_out", n_parts=parts); dt = time.time() - t
    sz = sum(os.path.getsize(os.path.join("/tmp/wt_out/path", f)) for f in os.listdir("/tmp/wt_out/path"))
    print("parts=%d: %.2f s for %.2f GB -> %.2f GB/s" % (parts, dt, sz / 1e9, sz / 1e9 / dt))
t = time.time()
with open("/tmp/wt_out/raw.bin", "wb") as f: f.write(paths.tobytes()[: 3340000000 // 2]); 
print("raw write of 1.67 GB: %.2f s" % (time.time() - t))
