"""CPU stand-in for the HIP step engine of stellar_random_walk_amd.distributed — TEST ONLY.

Implements capacity/seed/step with the CPU oracle so that the exchange protocol of ShardedWalker (counts
all-to-all, record all-to-all-v, MAX-combine of the path matrix) can run under gloo without a GPU."""
import numpy as np
import torch

import oracle_py


def owner(v, world):
    return int(v) % world  # python % is non-negative for positive world = Utils.nonNegativeMod


class OracleShardEngine:
    def __init__(self, graph, rank, world):
        self.g, self.rank, self.world = graph, rank, world
        self.device = torch.device("cpu")
        self.all_verts = graph.vertices()
        self.local = [(i, int(v)) for i, v in enumerate(self.all_verts) if owner(v, world) == rank]

    def capacity(self):
        return len(self.local), len(self.all_verts)

    def seed(self, iter_in_call, out, paths, stride):
        nv = len(self.all_verts)
        for k, (grank, v) in enumerate(self.local):
            wid = iter_in_call * nv + grank
            out[k] = torch.tensor([wid, v, v, v], dtype=torch.int32)
            paths[wid, 0] = v
        return len(self.local)

    def step(self, params, iteration, step, recs_in, n_in, recs_out, paths, stride, world):
        buckets = [[] for _ in range(world)]
        steps = dead = 0
        for i in range(n_in):
            wid, src, prev, curr = (int(x) for x in recs_in[i])
            assert owner(curr, world) == self.rank, "record delivered to the wrong rank"
            nb = self.g.neighbors(curr)
            if nb is None or len(nb[0]) == 0:
                dead += step > 1
                continue
            ids, w = nb
            it = params.first_walk + wid // len(self.all_verts)      # batched iterations: the walker id names its iteration
            r = params.const_r if params.rng_mode == 0 else oracle_py.walk_uniform(params.seed, it, src, step)
            if step == 1:
                k = oracle_py.sample_index(w, r)
            else:
                k = oracle_py.second_order_sample_index(params.p, params.q, prev, self.g.neighbors(prev)[0], ids, w, r)
            nxt = int(ids[k])
            paths[wid, step] = nxt
            buckets[owner(nxt, world)].append((wid, src, curr, nxt))
            steps += 1
        counts, pos = [], 0
        for b in buckets:
            for rec in b:
                recs_out[pos] = torch.tensor(rec, dtype=torch.int32)
                pos += 1
            counts.append(len(b))
        return counts, {"n_steps": steps, "dead_ends": dead, "kernel_ms": 0.0, "sum_deg_curr": 0}
