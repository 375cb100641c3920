"""The HIP shard kernels across a REAL process boundary (until now they had only exchanged chunks between
shard handles of one process, or with themselves at world 1).  Two processes, each with its own HIP step engine on device 0,
run stellar_random_walk_amd.distributed.ShardedWalker; the per-super-step all-to-all, the row-table all-reduce of the linked
first-order walk and the overflow vote travel through gloo, staged through host memory (HipShardEngine.all_to_all) — the
kernels, the chunk format and the driver are exactly what runs under RCCL, only the wire differs.  The result must equal the
single-process oracle walk bit for bit: p = q = 1 with and without row links, return-edge bias, and q != 1 with the per-edge
tables on the shards (replaces the shuffle of RandomWalk.scala:92-93,186-192)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from helpers import rmat_lines

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [dict(p=1.0, q=1.0), dict(p=1.0, q=1.0, no_links=True), dict(p=0.5, q=1.0), dict(p=0.25, q=4.0), dict(p=4.0, q=0.5),
         dict(p=0.25, q=4.0, edge_tables=False)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, directed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    import _pkg
    _pkg.load()
    from importlib import import_module
    sharded = import_module("stellar_random_walk_amd.distributed")
    torch.zeros(1, device="cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        coo = np.load(os.path.join(out_dir, "coo.npz"))
        drv = sharded.ShardedWalker(device=0, rank=rank, world=world)
        drv.EXCHANGE_WINDOW, drv.EXCHANGE_PAD = 4, 8      # the variable-occupancy exchange at this graph's size (defaults: 8 super-steps, 256 records)
        drv.load_coo(coo["s"], coo["d"], coo["w"], directed=directed)
        for ci, case in enumerate(CASES):
            kw = dict(case)
            if kw.pop("no_links", False):
                os.environ["SRW_SHARD_NO_LINKS"] = "1"
                drv._linked = None
            else:
                os.environ.pop("SRW_SHARD_NO_LINKS", None)
            extra = {}
            if "edge_tables" in kw:
                extra["flags_kw"] = dict(edge_tables=kw.pop("edge_tables"))
            paths, lens, stats = drv.walk(num_walks=3, first_walk=2, batch=2, walk_length=14, seed=17, **kw, **extra)
            if rank == 0:
                np.savez(os.path.join(out_dir, "res_%d.npz" % ci), paths=paths, lens=lens,
                         steps=sum(s["n_steps_global"] for s in stats), linked=int(bool(drv._linked)),
                         tables=sum(s.get("edge_tables", 0) for s in stats),
                         ex_bytes=sum(s["exchange_bytes_per_superstep"] for s in stats), ex_cap=sum(s["exchange_bytes_capacity"] for s in stats),
                         ex_retries=getattr(drv, "exchange_retries", 0),
                         table_steps=sum(s["strategy_steps"]["edge_table"] + s["strategy_steps"]["edge_mask"] for s in stats))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("directed", [False, True])
def test_two_processes_hip_engines_equal_oracle(oracle, tmp_path, directed):
    s, d, w = rmat_lines(oracle, 10, edge_factor=16, weighted=True)
    np.savez(tmp_path / "coo.npz", s=s, d=d, w=w)
    g = oracle.Graph.from_coo(s, d, w, directed=directed)
    mp.spawn(_worker, args=(2, _free_port(), directed, str(tmp_path)), nprocs=2, join=True)
    for ci, case in enumerate(CASES):
        r = np.load(tmp_path / ("res_%d.npz" % ci))
        rp, rl, rs = g.walk(p=case["p"], q=case["q"], walk_length=14, num_walks=3, first_walk=2, seed=17, threads=8)
        assert np.array_equal(r["lens"], rl) and np.array_equal(r["paths"], rp), case
        assert int(r["steps"]) == rs, case
        if case["p"] == 1.0 and case["q"] == 1.0:
            assert int(r["linked"]) == (0 if case.get("no_links") else 1), case
        # the exchange shipped the live prefix of the chunks, not whole chunks; on the directed graph walkers die and it shrinks further
        assert 0 < int(r["ex_bytes"]) < int(r["ex_cap"]), (case, int(r["ex_bytes"]), int(r["ex_cap"]))
        if case["q"] != 1.0:        # the shards' own per-edge tables served the second-order steps (or were switched off)
            assert (int(r["table_steps"]) > 0) == case.get("edge_tables", True), case
