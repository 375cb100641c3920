"""CPU stand-in for the HIP step engine of stellar_random_walk_amd.distributed — TEST ONLY.

Implements the srw_shard_* protocol (capacity / vertex_ranks / layout / begin / superstep / flush / finish) on CPU
tensors with the CPU oracle as the sampler, in the SAME chunk format the HIP kernels use
    chunk = { n_walkers, n_rets, 0, 0 } | {lw, src, prev, curr}[cap_w] | {lw, v}[cap_r]        (24 bytes per walker-step)
so that ShardedWalker's exchange (one equal-split all_to_all_single per super-step, flush, overflow retry, canonical
assembly) runs for real under gloo without a GPU."""
import numpy as np
import torch

import oracle_py


def owner(v, world):
    """The product's owner function (csrc/device_common.h:owner_of): mix32(v) mod world — the role of HashPartitioner's
    nonNegativeMod(id, n) (RandomWalk.scala:16) with the ids mixed first.  Restated here so that the protocol test routes
    walkers exactly as the HIP kernels do."""
    h = (int(v) & 0xFFFFFFFF) * 0x9E3779B1 & 0xFFFFFFFF
    h ^= h >> 15
    h = h * 0x85EBCA6B & 0xFFFFFFFF
    h ^= h >> 13
    return h % world


class Layout:
    def __init__(self, cap, chunk_bytes):
        self.cap_walkers = self.cap_rets = cap
        self.chunk_bytes = chunk_bytes


class OracleShardEngine:
    def __init__(self, graph, rank, world, tiny_chunks=False):
        self.g, self.rank, self.world = graph, rank, world
        self.device = torch.device("cpu")
        self.all_verts = graph.vertices()
        self.local = [(i, int(v)) for i, v in enumerate(self.all_verts) if owner(v, world) == rank]
        self.local_index = {}                      # vertex -> local index on ITS home rank
        cnt = [0] * world
        for v in self.all_verts:
            o = owner(v, world)
            self.local_index[int(v)] = cnt[o]
            cnt[o] += 1
        self.tiny = tiny_chunks                    # first layout deliberately too small: exercises the overflow retry
        self.steps = self.dead = 0
        self.overflow = 0

    def capacity(self):
        return len(self.local), len(self.all_verts)

    def vertex_ranks(self):
        return np.array([g for g, _ in self.local], dtype=np.int32)

    def layout(self, batch, slack):
        per_pair = batch * len(self.all_verts) / (self.world * self.world)
        cap = int(per_pair * slack) + (1 if self.tiny and slack < 2 else 4096)
        return Layout(cap, 16 + cap * 16 + cap * 8)

    # ---- chunk views over a uint8 tensor ----
    def _views(self, buf, lay, c):
        a = buf.numpy()
        base = c * lay.chunk_bytes
        hdr = a[base:base + 16].view(np.int32)
        w = a[base + 16:base + 16 + lay.cap_walkers * 16].view(np.int32).reshape(-1, 4)
        r = a[base + 16 + lay.cap_walkers * 16:base + lay.chunk_bytes].view(np.int32).reshape(-1, 2)
        return hdr, w, r

    def begin(self, P, batch, lay, recv, paths, lens):
        self.steps = self.dead = 0
        self.overflow = 0
        n = len(self.local) * batch
        paths.fill_(-1)
        for c in range(self.world):
            self._views(recv, lay, c)[0][:] = 0
        for i in range(n):
            v = self.local[i // batch][1]
            c = i % self.world
            hdr, w, _ = self._views(recv, lay, c)
            w[i // self.world] = (i, v, v, v)
            hdr[0] += 1
            paths[i, 0] = v
            lens[i] = P.walk_length + 2

    def _apply(self, lay, recv, paths, lens, slot):
        """returns produced by super-step `slot`: path slot `slot` of their walker; death notice: the path has `slot` entries"""
        for c in range(self.world):
            hdr, _, r = self._views(recv, lay, c)
            for i in range(min(int(hdr[1]), lay.cap_rets)):
                lw, v = int(r[i, 0]), int(r[i, 1])
                if lw < 0:
                    lens[lw & 0x7FFFFFFF] = slot
                else:
                    paths[lw, slot] = v

    def superstep(self, P, batch, step, lay, recv, send, paths, lens):
        if step > 1:
            self._apply(lay, recv, paths, lens, step - 1)
        last = step == P.walk_length + 1
        out_w = [[] for _ in range(self.world)]
        out_r = [[] for _ in range(self.world)]
        for c in range(self.world):
            hdr, w, _ = self._views(recv, lay, c)
            for i in range(min(int(hdr[0]), lay.cap_walkers)):
                lw, src, prev, curr = (int(x) for x in w[i])
                assert owner(curr, self.world) == self.rank, "record delivered to the wrong rank"
                nb = self.g.neighbors(curr)
                if nb is None or len(nb[0]) == 0:
                    self.dead += step > 1
                    out_r[owner(src, self.world)].append((lw - (1 << 31), 0))
                    continue
                ids, wt = nb
                it = P.first_walk + lw % batch
                r = P.const_r if P.rng_mode == 0 else oracle_py.walk_uniform(P.seed, it, src, step)
                if step == 1:
                    k = oracle_py.sample_index(wt, r)
                else:
                    k = oracle_py.second_order_sample_index(P.p, P.q, prev, self.g.neighbors(prev)[0], ids, wt, r)
                nxt = int(ids[k])
                self.steps += 1
                out_r[owner(src, self.world)].append((lw, nxt))
                if not last:
                    out_w[owner(nxt, self.world)].append((lw, src, curr, nxt))
        for d in range(self.world):
            hdr, w, r = self._views(send, lay, d)
            if len(out_w[d]) > lay.cap_walkers or len(out_r[d]) > lay.cap_rets:
                self.overflow = 1
            nw, nr = min(len(out_w[d]), lay.cap_walkers), min(len(out_r[d]), lay.cap_rets)
            hdr[:] = (nw, nr, 0, 0)
            if nw:
                w[:nw] = np.array(out_w[d][:nw], dtype=np.int64).astype(np.int32)
            if nr:
                r[:nr] = np.array(out_r[d][:nr], dtype=np.int64).astype(np.int32)

    def flush(self, P, batch, lay, recv, paths, lens):
        self._apply(lay, recv, paths, lens, P.walk_length + 1)

    def finish(self):
        return {"n_steps": self.steps, "dead_ends": self.dead, "kernel_ms": 0.0, "sum_deg_curr": 0}, self.overflow
