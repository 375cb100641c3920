"""world_size-2 (and 3) gloo tests of the vertex-sharded walk driver on CPU.

The product's step engine is HIP-only; here the CPU oracle is plugged in as the step engine (same chunk format as the HIP
kernels) so that the driver's protocol — one equal-split all_to_all_single of fixed-capacity chunks per super-step, path
returns to the home rank, the flush after the last super-step, the overflow retry, the canonical assembly — is exercised
for real across processes.  The sharded result must equal the single-process oracle walk bit for bit (keyed RNG => the
result does not depend on the number of shards)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import KARATE

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, directed, p, q, L, rng, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    sys.path.insert(0, HERE)
    import _pkg
    _pkg.load()
    from importlib import import_module
    sharded = import_module("stellar_random_walk_amd.distributed")
    import oracle_py
    from oracle_shard_engine import OracleShardEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = oracle_py.Graph.load(KARATE, directed=directed)
        drv = sharded.ShardedWalker(rank=rank, world=world, step_engine=OracleShardEngine(g, rank, world))
        paths, lens, stats = drv.walk(num_walks=2, first_walk=3, p=p, q=q, walk_length=L, seed=11, rng=rng, const_r=0.4)
        # both iterations shared their super-steps (one population); one iteration per population must give the same
        p1, l1, _ = drv.walk(num_walks=2, first_walk=3, batch=1, p=p, q=q, walk_length=L, seed=11, rng=rng, const_r=0.4)
        assert len(stats) == 1 and np.array_equal(p1, paths) and np.array_equal(l1, lens)
        # chunks deliberately too small at first: every rank must see the overflow and retry together with more slack
        drv2 = sharded.ShardedWalker(rank=rank, world=world, step_engine=OracleShardEngine(g, rank, world, tiny_chunks=True))
        p2, l2, _ = drv2.walk(num_walks=2, first_walk=3, p=p, q=q, walk_length=L, seed=11, rng=rng, const_r=0.4)
        assert np.array_equal(p2, paths) and np.array_equal(l2, lens)
        # variable-occupancy exchange: after a window of super-steps only the live prefix of every chunk section is shipped —
        # fewer bytes than whole chunks, the same paths; a window whose prefix a chunk outgrows (margin and pad forced to nothing)
        # raises the vote and every rank redoes the batch with whole chunks
        drv3 = sharded.ShardedWalker(rank=rank, world=world, step_engine=OracleShardEngine(g, rank, world))
        drv3.EXCHANGE_WINDOW = 2
        p3, l3, st3 = drv3.walk(num_walks=2, first_walk=3, p=p, q=q, walk_length=L, seed=11, rng=rng, const_r=0.4)
        assert np.array_equal(p3, paths) and np.array_equal(l3, lens)
        assert all(s_["exchange_bytes_per_superstep"] < 0.5 * s_["exchange_bytes_capacity"] for s_ in st3), st3
        drv3.EXCHANGE_WINDOW, drv3.EXCHANGE_MARGIN, drv3.EXCHANGE_PAD = 1, 0.0, 0
        p4, l4, st4 = drv3.walk(num_walks=2, first_walk=3, p=p, q=q, walk_length=L, seed=11, rng=rng, const_r=0.4)
        assert np.array_equal(p4, paths) and np.array_equal(l4, lens) and getattr(drv3, "exchange_retries", 0) >= 1
        assert all(s_["exchange_bytes_per_superstep"] == s_["exchange_bytes_capacity"] for s_ in st4), st4
        drv0 = sharded.ShardedWalker(rank=rank, world=world, step_engine=OracleShardEngine(g, rank, world))
        drv0.EXCHANGE_WINDOW = 0                                # whole chunks always
        p5, l5, st5 = drv0.walk(num_walks=2, first_walk=3, p=p, q=q, walk_length=L, seed=11, rng=rng, const_r=0.4)
        assert np.array_equal(p5, paths) and all(s_["exchange_bytes_per_superstep"] == s_["exchange_bytes_capacity"] for s_ in st5)
        # memory: a rank holds the paths of ITS walkers only
        pl, ll, _ = drv.walk_batch(iteration=3, num_walks=2, p=p, q=q, walk_length=L, seed=11, rng=rng, const_r=0.4)
        assert pl.shape[0] == 2 * drv.se.capacity()[0]
        np.save(os.path.join(out_dir, "paths_%d.npy" % rank), paths)
        np.save(os.path.join(out_dir, "lens_%d.npy" % rank), lens)
        np.save(os.path.join(out_dir, "steps_%d.npy" % rank), np.array([sum(s["n_steps_global"] for s in stats),
                                                                          sum(s["exchange_bytes_per_superstep"] for s in stats)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,directed,p,q,rng", [(2, False, 1.0, 1.0, "philox"), (2, True, 0.5, 1.0, "philox"),
                                                     (3, False, 4.0, 1.0, "const"), (2, False, 0.25, 4.0, "philox")])
def test_sharded_equals_single(oracle, tmp_path, world, directed, p, q, rng):
    L = 12
    mp.spawn(_worker, args=(world, _free_port(), directed, p, q, L, rng, str(tmp_path)), nprocs=world, join=True)
    g = oracle.Graph.load(KARATE, directed=directed)
    ref_paths, ref_lens, ref_steps = g.walk(p=p, q=q, walk_length=L, num_walks=2, first_walk=3, seed=11, rng=rng,
                                            const_r=0.4)
    for r in range(world):
        paths = np.load(tmp_path / ("paths_%d.npy" % r))
        lens = np.load(tmp_path / ("lens_%d.npy" % r))
        steps, exchanged = np.load(tmp_path / ("steps_%d.npy" % r))
        assert np.array_equal(lens, ref_lens)
        assert np.array_equal(paths, ref_paths)
        assert steps == ref_steps
    assert exchanged > 0  # walkers really crossed shards
