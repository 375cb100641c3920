"""Vertex-sharded multi-GPU walk, one process per GPU: walkers exchanged by ONE all-to-all per super-step.

Replaces the Spark shuffle of the reference's super-step loop
(M/algorithm/RandomWalk.scala:91-162: prepareWalkersToTransfer -> partitionBy(HashPartitioner) -> zipPartitions,
UniformRandomWalk.scala:103-112): the graph is sharded by source vertex, owner(v) = mix32(v) mod world
(RandomWalk.scala:16) or the VCut partition ids, and a walker standing on v is processed by owner(v).

What moves (xGMI, RCCL `all_to_all_single`, equal splits): per (sender, receiver) pair one fixed-capacity chunk
    { n_walkers, n_rets, 0, 0 } | {lw, src, prev, curr}[cap] | {lw, v}[cap]
— 16-byte walker records (not the path and not N(prev) as in the reference, RandomWalk.scala:135) and 8-byte path returns
(the vertex just sampled, to the walker's HOME rank, which alone stores its path; the slot is implicit: super-step s
produces slot s): 24 bytes per walker-step.  Memory per rank is 1/world of the paths plus 24 B per resident walker of chunk
buffers.  The counts travel in the chunk headers, so a super-step is: kernels -> one collective -> kernels, with no host
synchronisation; an overflowing chunk raises a flag that is read once per batch (every rank then retries together with more slack).
Because the RNG is keyed by (iteration, source vertex, step), the result is bit-identical to the single-GPU walk for any
world size — tests assert exactly that.

q != 1 needs N(prev), which lives on owner(prev): every shard therefore also keeps a replicated *membership
structure* of the whole graph (row boundaries + sorted neighbor ids, 4 B/entry; graph_build.hip), so the p/q bias is
evaluated locally and no second query/response exchange is needed — see DESIGN.md §6.

The same kernels serve the single-process form (csrc/cluster.cpp: peer stores instead of the collective).  The step
engine is injectable so that this driver's protocol can be tested with the gloo backend on CPU (the tests plug the CPU
oracle in; the product default is the HIP engine, which needs a GPU).
"""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import Engine, ShardLayout, SrwError, WalkStats, lib


class HipShardEngine:
    """Thin adapter: srw_shard_* on torch CUDA tensors (device pointers), kernels on torch's current stream."""

    def __init__(self, device, rank, world, owner_from_partitions=False, membership=True):
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        torch.zeros(1, device=self.device)       # torch's device context first (it ships its own HIP runtime)
        self.engine = Engine(device=device, rank=rank, world=world, owner_from_partitions=owner_from_partitions,
                             membership=membership)     # False: SRW_CFG_NO_MEMBERSHIP (q == 1 walks only)
        # population 0 runs on the stream that is current NOW — the object is kept: select() hands THIS stream back, whatever stream is
        # current by the time of the first select (kernels and their all_to_all must share one stream)
        self._streams = [torch.cuda.current_stream(self.device), None]
        self.engine.set_stream(self._streams[0].cuda_stream)
        self.world = world

    def select(self, population):
        """srw_shard_select: population 0 / 1 of this handle, each with its own super-step context and its own torch stream
        (returned: run the population's super-step and its collective under `with torch.cuda.stream(...)`)."""
        self.engine._ck(lib().srw_shard_select(self.engine.h, int(population)))
        if population == 1 and self._streams[1] is None:
            self._streams[1] = torch.cuda.Stream(device=self.device)
            self.engine.set_stream(self._streams[1].cuda_stream)        # (the selected context's stream)
        return self._streams[population]

    def capacity(self):
        return self.engine.shard_capacity()

    def vertex_ranks(self):
        n = self.capacity()[0]
        out = np.zeros(max(n, 1), dtype=np.int32)
        self.engine._ck(lib().srw_shard_vertex_ranks(self.engine.h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out[:n]

    def layout(self, batch, slack):
        lay = ShardLayout()
        self.engine._ck(lib().srw_shard_layout_for(self.engine.h, batch, float(slack), C.byref(lay)))
        return lay

    def begin(self, P, batch, lay, recv, paths, lens):
        self.engine._ck(lib().srw_shard_begin(self.engine.h, C.byref(P), batch, C.byref(lay), C.c_void_p(recv.data_ptr()),
                                              C.c_void_p(paths.data_ptr()), C.c_void_p(lens.data_ptr())))

    def superstep(self, P, batch, step, lay, recv, send, paths, lens):
        dst = (C.c_void_p * self.world)(*[C.c_void_p(send.data_ptr() + d * lay.chunk_bytes) for d in range(self.world)])
        self.engine._ck(lib().srw_shard_superstep(self.engine.h, C.byref(P), batch, step, C.byref(lay),
                                                  C.c_void_p(recv.data_ptr()), dst, C.c_void_p(paths.data_ptr()),
                                                  C.c_void_p(lens.data_ptr())))

    def flush(self, P, batch, lay, recv, paths, lens):
        self.engine._ck(lib().srw_shard_flush(self.engine.h, C.byref(P), batch, C.byref(lay), C.c_void_p(recv.data_ptr()),
                                              C.c_void_p(paths.data_ptr()), C.c_void_p(lens.data_ptr())))

    def finish(self):
        st, of = WalkStats(), C.c_int32(0)
        self.engine._ck(lib().srw_shard_finish(self.engine.h, C.byref(st), C.byref(of)))
        return st.as_dict(), of.value

    # Collectives on the engine's device buffers.  Backend nccl (= RCCL over xGMI): as they are.  Backend gloo (the two-process
    # GPU test on a single-GPU box, tests/test_gpu_two_process.py): staged through host memory — the kernels and the chunk
    # protocol are the real ones, only the wire is not.
    @staticmethod
    def _staged(group):
        return dist.get_backend(group) != "nccl"

    def all_reduce(self, t, op, group=None):
        if t.is_cuda and self._staged(group):
            torch.cuda.current_stream(self.device).synchronize()
            c = t.cpu()
            dist.all_reduce(c, op=op, group=group)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op, group=group)

    def all_to_all(self, recv, send, group=None):
        if recv.is_cuda and self._staged(group):
            torch.cuda.current_stream(self.device).synchronize()
            s = send.view(torch.int64).cpu()
            r = torch.empty_like(s)
            dist.all_to_all_single(r, s, group=group)
            recv.view(torch.int64).copy_(r)
        else:
            dist.all_to_all_single(recv.view(torch.int64), send.view(torch.int64), group=group)

    def link_rows(self, group=None):
        """Row descriptors across the shards (srw_shard_rows_*): one all-reduce MAX of the row tables, then every rank
        derives first-order records whose links point into the owners' tables.  All ranks or none.  Returns linked?"""
        if os.environ.get("SRW_SHARD_NO_LINKS"):
            return False
        L, h = lib(), self.engine.h
        n = C.c_int64(0)
        self.engine._ck(L.srw_shard_rows_count(h, C.byref(n)))
        rows = torch.zeros(max(n.value, 1) * 2, dtype=torch.int64, device=self.device)
        self.engine._ck(L.srw_shard_rows_export(h, C.c_void_p(rows.data_ptr()), n.value))
        torch.cuda.current_stream(self.device).synchronize()
        self.all_reduce(rows, dist.ReduceOp.MAX, group)
        # a failing commit on one rank must not leave the others waiting in the MIN all-reduce: take part with
        # ok = 0, release everywhere, raise afterwards
        linked, err = C.c_int32(0), None
        try:
            self.engine._ck(L.srw_shard_rows_commit(h, C.c_void_p(rows.data_ptr()), n.value, C.byref(linked)))
        except Exception as ex:      # noqa: BLE001
            err, linked = ex, C.c_int32(0)
        # [linked everywhere?, nobody failed?]: the error flag travels separately, so that EVERY rank raises when one rank's commit
        # failed — a rank that only saw "not linked" would walk on into the next collective and wait there for the departed one
        # until the process-group timeout
        ok = torch.tensor([linked.value, 0 if err is not None else 1], dtype=torch.int64, device=self.device)
        self.all_reduce(ok, dist.ReduceOp.MIN, group)
        all_linked, nobody_failed = int(ok[0]) != 0, int(ok[1]) != 0
        if not (all_linked and nobody_failed):
            self.engine._ck(L.srw_shard_rows_release(h))
        if err is not None:
            raise err
        if not nobody_failed:
            raise SrwError(3, "srw_shard_rows_commit failed on another rank: the row links were released on every rank")
        return all_linked


class _DeviceBuffer:
    """One hipMalloc through the library, seen by torch through __cuda_array_interface__ (zero copy).  The exchange
    buffers are NOT taken from torch's caching allocator: RCCL faulted (memory access fault inside the collective) on
    multi-GB buffers that were sub-blocks of a cached segment freed by an earlier, smaller population
    (profiles/r02i_rccl_driver_buffers.md)."""

    def __init__(self, engine, nbytes, device):
        self.engine, self.nbytes = engine, int(nbytes)
        p = C.c_void_p()
        engine._ck(lib().srw_device_alloc(engine.h, self.nbytes, C.byref(p)))
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 2,
                                         "strides": None}
        self.tensor = torch.as_tensor(self, device=device)

    def free(self):
        if self.ptr:
            self.tensor = None
            lib().srw_device_free(self.engine.h, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ShardedWalker:
    def __init__(self, device=0, rank=None, world=None, step_engine=None, group=None, owner_from_partitions=False,
                 membership=True):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.se = step_engine if step_engine is not None else HipShardEngine(device, self.rank, self.world,
                                                                            owner_from_partitions, membership)
        self.engine = getattr(self.se, "engine", None)
        self.device = self.se.device
        self._bufs = {}
        self._linked = None          # row links across the shards: not tried yet for the loaded graph

    def _a2a(self, recv, send):
        if hasattr(self.se, "all_to_all"):
            self.se.all_to_all(recv, send, self.group)
        else:
            dist.all_to_all_single(recv.view(torch.int64), send.view(torch.int64), group=self.group)

    # ---- variable-occupancy exchange ("counts first, then all-to-all-v of what is live") ------------------------------------------
    # The chunk of a (sender, receiver) pair has room for the walkers a population STARTS with (x slack); on a directed graph walkers die
    # (config 5: ~40 % alive on average), on any graph the last super-steps of a batch carry fewer — shipping `world x chunk_bytes` every
    # super-step moves mostly empty slots then.  What is exchanged instead is the PREFIX of each section that holds records: the header +
    # `tw` walker records + `tr` path returns per pair, packed into one contiguous equal-split message (two strided device copies either
    # side of the one collective: no second collective, no per-step host synchronisation, any backend).  tw / tr follow the counts the
    # chunk headers carry: every EXCHANGE_WINDOW super-steps the largest header count of the window goes through one small all-reduce
    # (MAX) and the next window ships that x EXCHANGE_MARGIN (+ a pad); a chunk that outgrows its window's prefix sets a flag that joins
    # the overflow vote at the end of the batch — every rank then redoes the batch with whole chunks, as for a capacity overflow.
    EXCHANGE_WINDOW = int(os.environ.get("SRW_EXCHANGE_WINDOW", "8"))       # super-steps per count refresh; 0: always whole chunks
    EXCHANGE_MARGIN = float(os.environ.get("SRW_EXCHANGE_MARGIN", "1.25"))
    EXCHANGE_PAD = 256                                                      # records on top of margin x the window's largest count

    class _Exchange:
        """The per-batch state of the variable exchange over one (recv, send) buffer pair of layout `lay`."""

        def __init__(self, walker, lay, recv, send, variable=True):
            self.w, self.lay, self.recv, self.send = walker, lay, recv, send
            self.cap_w, self.cap_r = int(lay.cap_walkers), int(lay.cap_rets)
            self.tw, self.tr = self.cap_w, self.cap_r          # whole chunks until the first refresh
            self.off_r = 16 + self.cap_w * 16
            self.window = walker.EXCHANGE_WINDOW if variable else 0
            self.seen = torch.zeros(2, dtype=torch.int32, device=recv.device)       # largest header counts of the window (this rank's sends)
            self.trunc = torch.zeros(1, dtype=torch.int32, device=recv.device)      # a chunk outgrew the prefix that was shipped
            self.bytes = 0
            self.steps = 0

        def run(self, step):
            w, world, cb = self.w, self.w.world, int(self.lay.chunk_bytes)
            S = self.send.view(world, cb)
            if self.window > 0:
                hdr = S[:, :8].view(torch.int32)                    # (world, 2): n_walkers, n_rets of the chunks this rank sends
                mx = hdr.max(dim=0).values
                self.seen = torch.maximum(self.seen, mx)
                self.trunc |= ((mx[0] > self.tw) | (mx[1] > self.tr)).to(torch.int32)
            if self.tw >= self.cap_w and self.tr >= self.cap_r:
                w._a2a(self.recv, self.send)
                self.bytes += world * cb
            else:
                nw, nr = 16 + self.tw * 16, self.tr * 8
                pack = torch.cat([S[:, :nw], S[:, self.off_r:self.off_r + nr]], dim=1).contiguous()
                ph = pack[:, :8].view(torch.int32)                  # (a chunk that outgrew the prefix: the receiver must not read past what was
                ph[:, 0].clamp_(max=self.tw); ph[:, 1].clamp_(max=self.tr)      #  shipped — the batch is redone anyway, see truncated())
                rpack = torch.empty_like(pack)
                w._a2a(rpack.view(-1), pack.view(-1))
                R = self.recv.view(world, cb)
                R[:, :nw] = rpack[:, :nw]
                R[:, self.off_r:self.off_r + nr] = rpack[:, nw:]
                self.bytes += world * (nw + nr)
            self.steps += 1
            if self.window > 0 and step % self.window == 0:
                m = self.seen.clone()
                w._ar(m, dist.ReduceOp.MAX)                         # (one small collective + one read per window)
                mw, mr = (int(x) for x in m.tolist())
                # (the prefixes stay multiples of 8 records: the packed message stays 8-byte granular for the int64 view of the collective)
                self.tw = min(self.cap_w, (int(mw * w.EXCHANGE_MARGIN) + w.EXCHANGE_PAD + 7) & ~7)
                self.tr = min(self.cap_r, (int(mr * w.EXCHANGE_MARGIN) + w.EXCHANGE_PAD + 7) & ~7)
                self.seen.zero_()

        def truncated(self):
            return int(self.trunc.item()) if self.window > 0 else 0

    def _ar(self, t, op):
        if hasattr(self.se, "all_reduce"):
            self.se.all_reduce(t, op, self.group)
        else:
            dist.all_reduce(t, op=op, group=self.group)

    # ---- graph (each rank keeps only the rows it owns) ----
    def generate_rmat(self, scale, n_edges=None, seed=42, weighted=False, directed=False):
        self.engine.generate_rmat(scale, n_edges, seed=seed, weighted=weighted, directed=directed)
        self._linked = None
        return self

    def load_edgelist(self, path, **kw):
        self.engine.load_edgelist(path, **kw)
        self._linked = None
        return self

    def load_coo(self, src, dst, w=None, pid=None, directed=False):
        self.engine.load_coo(src, dst, w, pid=pid, directed=directed)
        self._linked = None
        return self

    # One (sender, receiver) message of the equal-split all-to-all; RCCL / torch fault on messages beyond 2 GiB
    # (observed: world 1, 2.5 GB chunk — memory access fault inside the collective), so the driver refuses them.
    MAX_MESSAGE_BYTES = int(os.environ.get("SRW_MAX_MESSAGE_BYTES", (1 << 31) - 4096))

    def max_batch(self, want, slack=1.25):
        """Largest number of walk iterations <= want that one population may hold under MAX_MESSAGE_BYTES."""
        b = max(1, int(want))
        while b > 1 and self.se.layout(b, slack).chunk_bytes > self.MAX_MESSAGE_BYTES:
            b -= 1
        return b

    def _buffers(self, nbytes, key=""):
        b = self._bufs.get("x" + key)
        if b is None or b[0].numel() < nbytes:
            if self.engine is not None and self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
                old = self._bufs.pop("mem" + key, None)
                if old:
                    for m in old:
                        m.free()
                mem = (_DeviceBuffer(self.engine, nbytes, self.device), _DeviceBuffer(self.engine, nbytes, self.device))
                self._bufs["mem" + key] = mem
                b = (mem[0].tensor, mem[1].tensor)
            else:                           # the CPU protocol tests (gloo + the oracle as step engine)
                b = (torch.empty(nbytes, dtype=torch.uint8, device=self.device), torch.empty(nbytes, dtype=torch.uint8, device=self.device))
            self._bufs["x" + key] = b
        return b[0][:nbytes], b[1][:nbytes]

    # ---- walk_length + 1 super-steps over the walkers of `num_walks` consecutive walk iterations ----
    def walk_batch(self, iteration=0, p=1.0, q=1.0, walk_length=80, num_walks=1, seed=42, rng="philox", const_r=0.0,
                   slack=1.25, flags_kw=None):
        """Walk iterations iteration .. iteration + num_walks - 1 as ONE walker population: they share their super-steps,
        so the per-super-step costs (kernel launches, one collective) are paid once per batch.  Returns (paths, lens, stats)
        of THIS rank's walkers (device tensors; row lw = local vertex lw // num_walks, iteration lw % num_walks)."""
        world, B = self.world, num_walks
        n_local, n_global = self.se.capacity()
        stride = walk_length + 2
        if self._linked is None and p == 1.0 and q == 1.0 and rng == "philox":
            self._linked = bool(self.se.link_rows(self.group)) if hasattr(self.se, "link_rows") else False
        P = Engine.params(p=p, q=q, walk_length=walk_length, num_walks=B, first_walk=iteration, rng=rng, const_r=const_r, seed=seed,
                          **(flags_kw or {}))                 # flags_kw: Engine.params switches (edge_tables=False, ...; tests)
        dev = self.device
        paths = torch.empty((max(B * n_local, 1), stride), dtype=torch.int32, device=dev)
        lens = torch.empty(max(B * n_local, 1), dtype=torch.int32, device=dev)
        variable = True
        while True:
            lay = self.se.layout(B, slack)
            if lay.chunk_bytes > self.MAX_MESSAGE_BYTES:
                raise ValueError("vertex-sharded walk: a chunk of %d bytes per peer exceeds the %d-byte exchange message limit "
                                 "(batch %d, slack %.2f): walk fewer iterations per population (max_batch())"
                                 % (lay.chunk_bytes, self.MAX_MESSAGE_BYTES, B, slack))
            recv, send = self._buffers(world * lay.chunk_bytes)
            ex = self._Exchange(self, lay, recv, send, variable=variable)
            self.se.begin(P, B, lay, recv, paths, lens)
            for step in range(1, walk_length + 2):
                self.se.superstep(P, B, step, lay, recv, send, paths, lens)
                ex.run(step)                                  # chunk (me -> d) -> rank d's slot `me` (the live prefix of it)
            self.se.flush(P, B, lay, recv, paths, lens)
            st, overflow = self.se.finish()
            trunc = ex.truncated()
            t = torch.tensor([st["n_steps"], st["dead_ends"], overflow, trunc], dtype=torch.int64, device=dev)
            tot = t.clone()
            self._ar(tot, dist.ReduceOp.SUM)
            if int(tot[2]) == 0 and int(tot[3]) == 0:
                break
            if int(tot[3]):
                variable = False                          # a chunk outgrew its window's prefix somewhere: every rank redoes the batch with whole chunks
                self.exchange_retries = getattr(self, "exchange_retries", 0) + 1
            if int(tot[2]):
                slack *= 2.0                              # a chunk was too small somewhere: every rank retries together
                if slack > 64:
                    raise RuntimeError("vertex-sharded walk: chunk overflow persists at 64x slack")
        st["n_steps_global"], st["dead_ends_global"] = int(tot[0]), int(tot[1])
        st["exchange_bytes_per_superstep"] = ex.bytes // max(ex.steps, 1)        # what crossed the wire per super-step (mean)
        st["exchange_bytes_capacity"] = world * lay.chunk_bytes                    # ... against whole chunks
        return paths[:B * n_local], lens[:B * n_local], st

    def walk_populations(self, iteration=0, p=1.0, q=1.0, walk_length=80, num_walks=2, seed=42, rng="philox", const_r=0.0,
                         slack=1.25, flags_kw=None):
        """Iterations iteration .. iteration + num_walks - 1 as TWO walker populations (num_walks // 2 and the rest) whose super-steps
        are interleaved, each on its own stream with its own exchange buffers: population B's kernels run while population A's chunks
        are in the all-to-all — the overlap the reference's shuffle / count rhythm (RandomWalk.scala:91-162) does not have.  Identical
        paths (a population is defined by its iterations).  Returns [(first iteration, n, paths, lens, stats)] per population.  Falls
        back to ONE population when the step engine has no second context (the CPU protocol tests), num_walks < 2 or
        SRW_SHARD_POPULATIONS=1."""
        pops_env = os.environ.get("SRW_SHARD_POPULATIONS", "")
        # (world 1 has no exchange to hide: one population there unless SRW_SHARD_POPULATIONS=2 asks for two)
        two = hasattr(self.se, "select") and num_walks >= 2 and (pops_env == "2" or (pops_env != "1" and self.world > 1))
        kw = dict(p=p, q=q, walk_length=walk_length, seed=seed, rng=rng, const_r=const_r, slack=slack, flags_kw=flags_kw)
        if not two:
            pth, ln, st = self.walk_batch(iteration=iteration, num_walks=num_walks, **kw)
            return [(iteration, num_walks, pth, ln, st)]
        world = self.world
        n_local, n_global = self.se.capacity()
        stride = walk_length + 2
        if self._linked is None and p == 1.0 and q == 1.0 and rng == "philox":
            self._linked = bool(self.se.link_rows(self.group)) if hasattr(self.se, "link_rows") else False
        dev = self.device
        variable = True
        sizes = [num_walks // 2, num_walks - num_walks // 2]
        firsts = [iteration, iteration + sizes[0]]
        Ps = [Engine.params(p=p, q=q, walk_length=walk_length, num_walks=b, first_walk=f, rng=rng, const_r=const_r, seed=seed, **(flags_kw or {}))
              for b, f in zip(sizes, firsts)]
        paths = [torch.empty((max(b * n_local, 1), stride), dtype=torch.int32, device=dev) for b in sizes]
        lens = [torch.empty(max(b * n_local, 1), dtype=torch.int32, device=dev) for b in sizes]
        try:
            while True:
                lays = [self.se.layout(b, slack) for b in sizes]
                for lay in lays:
                    if lay.chunk_bytes > self.MAX_MESSAGE_BYTES:
                        raise ValueError("vertex-sharded walk: a chunk of %d bytes per peer exceeds the %d-byte exchange message limit" % (lay.chunk_bytes, self.MAX_MESSAGE_BYTES))
                bufs = [self._buffers(world * lay.chunk_bytes, key=str(k)) for k, lay in enumerate(lays)]
                exs = [self._Exchange(self, lays[k], bufs[k][0], bufs[k][1], variable=variable) for k in (0, 1)]
                torch.cuda.synchronize(dev)
                streams = [self.se.select(k) for k in (0, 1)]
                for k in (0, 1):
                    self.se.select(k)
                    with torch.cuda.stream(streams[k]):
                        self.se.begin(Ps[k], sizes[k], lays[k], bufs[k][0], paths[k], lens[k])
                for step in range(1, walk_length + 2):
                    for k in (0, 1):
                        self.se.select(k)
                        with torch.cuda.stream(streams[k]):
                            self.se.superstep(Ps[k], sizes[k], step, lays[k], bufs[k][0], bufs[k][1], paths[k], lens[k])
                            exs[k].run(step)
                sts, overflow = [], 0
                for k in (0, 1):
                    self.se.select(k)
                    with torch.cuda.stream(streams[k]):
                        self.se.flush(Ps[k], sizes[k], lays[k], bufs[k][0], paths[k], lens[k])
                        st, of = self.se.finish()
                    sts.append(st); overflow |= of
                self.se.select(0)
                torch.cuda.synchronize(dev)
                trunc = exs[0].truncated() | exs[1].truncated()
                t = torch.tensor([sts[0]["n_steps"] + sts[1]["n_steps"], sts[0]["dead_ends"] + sts[1]["dead_ends"], overflow, trunc], dtype=torch.int64, device=dev)
                tot = t.clone()
                self._ar(tot, dist.ReduceOp.SUM)
                if int(tot[2]) == 0 and int(tot[3]) == 0:
                    break
                if int(tot[3]):
                    variable = False
                    self.exchange_retries = getattr(self, "exchange_retries", 0) + 1
                if int(tot[2]):
                    slack *= 2.0
                    if slack > 64:
                        raise RuntimeError("vertex-sharded walk: chunk overflow persists at 64x slack")
        finally:
            self.se.select(0)
        out = []
        for k in (0, 1):
            sts[k]["n_steps_global"], sts[k]["dead_ends_global"] = (int(tot[0]), int(tot[1])) if k == 0 else (0, 0)
            sts[k]["exchange_bytes_per_superstep"] = exs[k].bytes // max(exs[k].steps, 1)
            sts[k]["exchange_bytes_capacity"] = world * lays[k].chunk_bytes
            out.append((firsts[k], sizes[k], paths[k][:sizes[k] * n_local], lens[k][:sizes[k] * n_local], sts[k]))
        return out

    def profile_batch(self, iteration=0, p=1.0, q=1.0, walk_length=80, num_walks=1, seed=42, slack=1.25):
        """One single-population batch with device events around every super-step's kernels and around its all-to-all: what a
        super-step costs in kernels and in exchange when nothing overlaps (bench.py prints it next to the overlapped rate, so that a
        first multi-GPU run can be read).  Returns {"kernels_ms", "exchange_ms"} (means per super-step) or None on CPU engines."""
        if self.device.type != "cuda":
            return None
        n_local, _ = self.se.capacity()
        P = Engine.params(p=p, q=q, walk_length=walk_length, num_walks=num_walks, first_walk=iteration, seed=seed)
        lay = self.se.layout(num_walks, slack)
        recv, send = self._buffers(self.world * lay.chunk_bytes)
        paths = torch.empty((max(num_walks * n_local, 1), walk_length + 2), dtype=torch.int32, device=self.device)
        lens = torch.empty(max(num_walks * n_local, 1), dtype=torch.int32, device=self.device)
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(walk_length + 1)]
        self.se.begin(P, num_walks, lay, recv, paths, lens)
        for i, step in enumerate(range(1, walk_length + 2)):
            ev[i][0].record()
            self.se.superstep(P, num_walks, step, lay, recv, send, paths, lens)
            ev[i][1].record()
            self._a2a(recv, send)
            ev[i][2].record()
        self.se.flush(P, num_walks, lay, recv, paths, lens)
        self.se.finish()
        torch.cuda.synchronize(self.device)
        k = [e[0].elapsed_time(e[1]) for e in ev]
        x = [e[1].elapsed_time(e[2]) for e in ev]
        return {"kernels_ms": sum(k) / len(k), "exchange_ms": sum(x) / len(x), "super_steps": len(k), "exchange_bytes": self.world * lay.chunk_bytes}

    def walk(self, num_walks=1, first_walk=0, batch=None, **kw):
        """num_walks iterations; returns (paths [num_walks * nV, L + 2], lens, stats) in canonical order on every rank
        (all-gathered: tests and small graphs; production keeps the paths on their home ranks, see walk_batch)."""
        n_local, n_global = self.se.capacity()
        stride = kw.get("walk_length", 80) + 2
        if batch is None:
            batch = max(1, min(num_walks, (1 << 30) // max(1, (n_global // self.world + 1) * 30)))
        vr = torch.as_tensor(self.se.vertex_ranks().astype(np.int64), device=self.device)
        out_p = torch.full((num_walks * n_global, stride), -1, dtype=torch.int32, device=self.device)
        out_l = torch.zeros(num_walks * n_global, dtype=torch.int32, device=self.device)
        stats = []
        for it0 in range(0, num_walks, batch):
            for (f, b, pth, ln, st) in self.walk_populations(iteration=first_walk + it0, num_walks=min(batch, num_walks - it0), **kw):
                it = f - first_walk
                lw = torch.arange(b * n_local, device=self.device, dtype=torch.int64)
                canon = (it + lw % b) * n_global + vr[lw // b] if n_local else lw
                out_p[canon] = pth
                out_l[canon] = ln
                stats.append(st)
        # every canonical row is owned by exactly one rank; rows of other ranks are (-1.., 0) here
        self._ar(out_l, dist.ReduceOp.SUM)
        out_p += 1                                        # -1 filler -> 0, ids shifted by one: SUM assembles, no sentinel id
        wide = out_p.to(torch.int64)
        self._ar(wide, dist.ReduceOp.SUM)
        # rows owned by nobody else contributed 0 = (-1 + 1) from the other ranks
        paths = (wide - 1).to(torch.int32)
        return paths.cpu().numpy(), out_l.cpu().numpy(), stats


def bench_vertex_sharded(dist_mod, local_rank, rank, world, scale, n_edges, weighted, directed, walk_kw, K, W, barrier_sync):
    """bench.py's vertex-sharded leg: K walk iterations (strong scaling: every rank works on every iteration)."""
    drv = ShardedWalker(device=local_rank, rank=rank, world=world, membership=(walk_kw.get("q", 1.0) != 1.0))
    t0 = time.perf_counter()
    drv.generate_rmat(scale, n_edges, seed=42, weighted=weighted, directed=directed)
    nv, ne = drv.engine.stats()
    n_local, _ = drv.se.capacity()
    t_graph = time.perf_counter() - t0
    kw = {k: v for k, v in walk_kw.items() if k in ("p", "q", "walk_length", "seed")}
    B = drv.max_batch(max(1, min(K, 4)))
    for it in range(0, W, B):
        drv.walk_batch(iteration=it, num_walks=min(B, W - it), **kw)
    if W == 0:
        drv.walk_batch(iteration=0, num_walks=1, **dict(kw, walk_length=1))     # tables outside the timed region
    barrier_sync()
    t0 = time.perf_counter()
    steps = 0
    st = None
    for it in range(W, W + K, B):
        for (_, _, _, _, st) in drv.walk_populations(iteration=it, num_walks=min(B, W + K - it), **kw):
            steps += st["n_steps_global"]
    barrier_sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist_mod.get_backend() == "nccl" else "cpu")
    dist_mod.all_reduce(t, op=dist_mod.ReduceOp.MAX)
    max_dt = float(t.item())
    try:          # outside the timed region: one unoverlapped batch, kernels and exchange timed apart
        prof = drv.profile_batch(iteration=W + K, num_walks=max(1, B // 2), **kw)
    except Exception as ex:      # noqa: BLE001
        prof = {"error": str(ex)[:200]}
    return {"value": steps / max_dt, "unit": "walk-steps/s", "ms_per_step": max_dt / max(K, 1) * 1e3, "scaling": "strong",
            "steps": K, "warmup": W, "iterations_per_batch": B, "populations_per_batch": 2 if (B >= 2 and (os.environ.get("SRW_SHARD_POPULATIONS", "") == "2" or (os.environ.get("SRW_SHARD_POPULATIONS", "") != "1" and world > 1))) else 1,
            "workload": "RMAT scale-%d (%d edge lines, %d adjacency entries, %d vertices), p=%g q=%g walkLength=%d" % (
                scale, n_edges, ne, nv, kw.get("p", 1.0), kw.get("q", 1.0), kw.get("walk_length", 80)),
            "parallelism": "graph sharded by source vertex x%d (owner = mix32(id) mod world), 1 RCCL all_to_all_single per super-step, "
                           "paths on the home GPU" % world,
            "local_vertices_rank0": n_local, "exchange_bytes_per_superstep_per_rank": st["exchange_bytes_per_superstep"] if st else 0,
            "per_superstep_unoverlapped": prof,
            "setup_s": {"graph_generate_and_csr": t_graph}}
