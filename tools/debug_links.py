"""Scratch: hub graph through the in-process cluster, linked vs unlinked."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import _pkg
pkg = _pkg.load()
for n in (50, 200, 3000, 20000):
    s = np.concatenate([np.zeros(n, np.int32), np.arange(1, n + 1, dtype=np.int32)])
    d = np.concatenate([np.arange(1, n + 1, dtype=np.int32), np.roll(np.arange(1, n + 1, dtype=np.int32), 1)])
    for world in (1, 2, 4):
        with pkg.Cluster([0] * world) as cl:
            cl.load_coo(s, d, None, directed=True)
            paths, lens, st = cl.walk(walk_length=6, num_walks=2, seed=5)
            bad = np.nonzero(lens != 8)[0]
            print(n, world, "short walkers:", bad[:10], "of", len(lens), "path0", paths[0], flush=True)
