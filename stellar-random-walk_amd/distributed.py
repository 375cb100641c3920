"""Vertex-sharded multi-GPU walk: one process per GPU, walkers exchanged by all-to-all each super-step.

Replaces the Spark shuffle of the reference's super-step loop
(M/algorithm/RandomWalk.scala:91-162: prepareWalkersToTransfer -> partitionBy(HashPartitioner) -> zipPartitions,
UniformRandomWalk.scala:103-112): the graph is sharded by source vertex, owner(v) = nonNegativeMod(v, world)
(RandomWalk.scala:16), and a walker standing on v is processed by owner(v).

What moves (xGMI, RCCL `all_to_all_single`): fixed 16-byte records {wid, src, prev, curr} — not the path and
not N(prev) as in the reference (RandomWalk.scala:135).  Every rank writes the vertices it samples into its own
copy of the path matrix (slot (wid, step) is written by exactly one rank); one MAX all-reduce per walk
iteration assembles the paths.  Because the RNG is keyed by (iteration, source vertex, step), the result is
bit-identical to the single-GPU walk for any world size — tests assert exactly that.

q != 1 needs N(prev), which lives on owner(prev): every shard therefore also keeps a replicated *membership
structure* of the whole graph (row boundaries + sorted neighbor ids, 4 B/entry; graph_build.hip), so the p/q bias is
evaluated locally and the exchanged record stays 16 bytes — see DESIGN.md §6.

The step engine is injectable so that the exchange protocol can be tested with the gloo backend on CPU (the
tests plug the CPU oracle in; the product default is the HIP engine, which needs a GPU).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import Engine, WalkStats, lib, OK, SrwError

UNWRITTEN = -(2 ** 31)  # path-slot filler for the MAX-combine (every real id is larger)


class HipShardEngine:
    """Thin adapter: srw_shard_* on torch CUDA tensors (device pointers), kernels on torch's current stream."""

    def __init__(self, device, rank, world, owner_from_partitions=False):
        self.engine = Engine(device=device, rank=rank, world=world, owner_from_partitions=owner_from_partitions)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.engine.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def capacity(self):
        return self.engine.shard_capacity()

    def seed(self, iter_in_call, out, paths, stride):
        n = C.c_int64(0)
        self.engine._ck(lib().srw_shard_seed(self.engine.h, iter_in_call, C.c_void_p(out.data_ptr()), C.byref(n),
                                             C.c_void_p(paths.data_ptr()), stride))
        return n.value

    def step(self, params, iteration, step, recs_in, n_in, recs_out, paths, stride, world):
        counts = (C.c_int64 * world)()
        st = WalkStats()
        self.engine._ck(lib().srw_shard_step(self.engine.h, C.byref(params), iteration, step,
                                             C.c_void_p(recs_in.data_ptr()), n_in, C.c_void_p(recs_out.data_ptr()),
                                             counts, C.c_void_p(paths.data_ptr()), stride, C.byref(st)))
        return list(counts), st.as_dict()


class ShardedWalker:
    def __init__(self, device=0, rank=None, world=None, step_engine=None, group=None, owner_from_partitions=False):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.se = step_engine if step_engine is not None else HipShardEngine(device, self.rank, self.world,
                                                                            owner_from_partitions)
        self.engine = getattr(self.se, "engine", None)
        self.device = self.se.device

    # ---- graph (each rank keeps only the rows it owns) ----
    def generate_rmat(self, scale, n_edges=None, seed=42, weighted=False, directed=False):
        self.engine.generate_rmat(scale, n_edges, seed=seed, weighted=weighted, directed=directed)
        return self

    def load_edgelist(self, path, **kw):
        self.engine.load_edgelist(path, **kw)
        return self

    def load_coo(self, src, dst, w=None, pid=None, directed=False):
        self.engine.load_coo(src, dst, w, pid=pid, directed=directed)
        return self

    # ---- walk_length + 1 super-steps over the walkers of `num_walks` consecutive walk iterations ----
    def walk_iteration(self, iteration=0, p=1.0, q=1.0, walk_length=80, num_walks=1, seed=42, rng="philox",
                       const_r=0.0, gather=False):
        """Walk iterations iteration .. iteration + num_walks - 1 as ONE walker population (walker id =
        iteration-in-call * nVertices + rank of the source vertex; the step kernels derive the RNG's iteration word
        from it), so the per-super-step costs (launches, one host sync, two collectives) are paid once per batch."""
        world = self.world
        n_local, n_global = self.se.capacity()
        if num_walks * n_global >= 2 ** 31:
            raise SrwError(-1, "num_walks * nVertices must stay below 2^31 per batch")
        stride = walk_length + 2
        P = Engine.params(p=p, q=q, walk_length=walk_length, num_walks=1, first_walk=iteration, rng=rng,
                          const_r=const_r, seed=seed)
        dev = self.device
        n_walkers = num_walks * n_global
        paths = torch.full((n_walkers, stride), UNWRITTEN, dtype=torch.int32, device=dev)
        cap = max(n_walkers, 1)
        cur = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        out = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        n = 0
        for b in range(num_walks):
            n += self.se.seed(b, cur[n:], paths, stride)
        tot = {"n_steps": 0, "dead_ends": 0, "kernel_ms": 0.0, "sum_deg_curr": 0, "exchanged": 0}
        for step in range(1, walk_length + 2):
            counts, st = self.se.step(P, iteration, step, cur, n, out, paths, stride, world)
            tot["n_steps"] += st["n_steps"]
            tot["dead_ends"] += st["dead_ends"]
            tot["kernel_ms"] += st["kernel_ms"]
            tot["sum_deg_curr"] += st["sum_deg_curr"]
            # 1) counts: who sends how many records to whom
            send = torch.tensor(counts, dtype=torch.int64, device=dev)
            recv = torch.empty(world, dtype=torch.int64, device=dev)
            dist.all_to_all_single(recv, send, group=self.group)
            recv_counts = recv.tolist()
            n_next = int(sum(recv_counts))
            tot["exchanged"] += int(sum(counts)) - counts[self.rank]
            # 2) records: all-to-all-v of 16-byte rows (out is grouped by owner(next) in rank order)
            n_send = int(sum(counts))
            dist.all_to_all_single(cur[:n_next], out[:n_send], output_split_sizes=recv_counts,
                                   input_split_sizes=counts, group=self.group)
            n = n_next
        # assemble: slot (wid, step) was written by exactly one rank
        dist.all_reduce(paths, op=dist.ReduceOp.MAX, group=self.group)
        t = torch.tensor([tot["n_steps"], tot["dead_ends"]], dtype=torch.int64, device=dev)
        dist.all_reduce(t, group=self.group)
        tot["n_steps_global"], tot["dead_ends_global"] = int(t[0]), int(t[1])
        if gather:
            written = paths != UNWRITTEN
            lens = written.sum(dim=1).to(torch.int32)
            paths = torch.where(written, paths, torch.full_like(paths, -1))
            return paths.cpu().numpy(), lens.cpu().numpy(), tot
        return tot

    def walk(self, num_walks=1, first_walk=0, batch=None, **kw):
        """num_walks iterations; returns (paths [num_walks * nV, L + 2], lens, stats) on every rank.  `batch` walk
        iterations share their super-steps (default: as many as keep the path matrix under 8 GiB)."""
        _, n_global = self.se.capacity()
        stride = kw.get("walk_length", 80) + 2
        if batch is None:
            batch = max(1, min(num_walks, (8 << 30) // max(1, n_global * stride * 4), (2 ** 31 - 1) // max(1, n_global)))
        ps, ls, stats = [], [], []
        for it in range(0, num_walks, batch):
            b = min(batch, num_walks - it)
            pth, ln, st = self.walk_iteration(iteration=first_walk + it, num_walks=b, gather=True, **kw)
            ps.append(pth)
            ls.append(ln)
            stats.append(st)
        return np.concatenate(ps), np.concatenate(ls), stats
