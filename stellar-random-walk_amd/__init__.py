"""stellar_random_walk_amd — Python binding (ctypes) of libstellar_rw.so, the MI355X-native engine behind the
`--cmd randomwalk` path of data61/stellar-random-walk.

This module is plumbing: every computation happens in hand-written HIP kernels behind the C ABI declared in
include/stellar_rw.h.  There is no CPU fallback — importing works anywhere (so the symbols can be checked), but
creating an Engine without a usable gfx950 device raises.
"""
import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRW_LIB") or os.path.join(_DIR, "libstellar_rw.so")   # SRW_LIB: an instrumented build (tools/)
CLI_PATH = os.path.join(_DIR, "stellar-rw")

OK, ERR_INVALID, ERR_IO, ERR_PARSE, ERR_HIP, ERR_EXISTS, ERR_NOMEM = range(7)
SAMPLER_REFERENCE, SAMPLER_ALIAS = 0, 1
RNG_CONST, RNG_PHILOX = 0, 1
CFG_OWNER_FROM_PARTITIONS = 1
CFG_COMPACT_IDS = 2
CFG_NO_MEMBERSHIP = 4
CFG_OWNER_HASH_PARTITIONER = 8
WALK_FORCE_GENERAL = 1
WALK_NT_LOADS = 2
WALK_CACHED_LOADS = 4
WALK_NO_COMPACT = 16
WALK_NO_PREFIX = 32
WALK_NO_EDGE_HASH = 64
WALK_NO_BINNED = 128
WALK_NO_HUB_BITMAPS = 65536
WALK_DEVICE_FORMAT = 131072
WALK_NO_EDGE_TABLES = 262144
WALK_EDGE_TABLES_ALL = 524288


class W2vParams(C.Structure):
    _fields_ = [("dim", C.c_int32), ("window", C.c_int32), ("iterations", C.c_int32), ("learning_rate", C.c_float),
                ("seed", C.c_uint32), ("threads", C.c_int32)]


def w2v_save(vocab_ids, vectors, output_dir, n_parts=1):
    """<output_dir>/vec/part-* ("id\\tv0\\t..." lines, Main.scala:88-91) + <output_dir>/bin."""
    ids = np.ascontiguousarray(vocab_ids, dtype=np.int32); vec = np.ascontiguousarray(vectors, dtype=np.float32)
    rc = lib().srw_w2v_save(_i32(ids), _f32(vec), len(ids), vec.shape[1] if vec.ndim == 2 else 1, os.fsencode(output_dir), n_parts)
    if rc != 0:
        raise SrwError(rc, "srw_w2v_save: %s" % lib().srw_last_error(None).decode())


class SrwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("srw error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("flags", C.c_int32)]


class WalkParams(C.Structure):
    _fields_ = [("p", C.c_float), ("q", C.c_float), ("walk_length", C.c_int32), ("num_walks", C.c_int32),
                ("first_walk", C.c_int32), ("rng_mode", C.c_int32), ("const_r", C.c_float), ("seed", C.c_uint32),
                ("sampler", C.c_int32), ("flags", C.c_int32)]


class WalkStats(C.Structure):
    _fields_ = [("n_walkers", C.c_int64), ("n_steps", C.c_int64), ("dead_ends", C.c_int64),
                ("sum_deg_curr", C.c_int64), ("sum_deg_prev", C.c_int64), ("ent_reads", C.c_int64),
                ("fallbacks", C.c_int64), ("trials", C.c_int64), ("kernel_ms", C.c_double), ("kernel_kind", C.c_int32),
                ("record_bytes", C.c_int32), ("strategy_steps", C.c_int64 * 12), ("edge_tables", C.c_int64),
                ("edge_table_bytes", C.c_int64), ("setup_ms", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["strategy_steps"] = dict(zip(STRATEGIES, list(self.strategy_steps)))
        return d


class ShardLayout(C.Structure):
    _fields_ = [("cap_walkers", C.c_int64), ("cap_rets", C.c_int64), ("chunk_bytes", C.c_int64)]


STRATEGIES = ("edge_table", "p1", "p2", "w", "p3", "scan", "prefix", "chain", "edge_mask", "q1_lane", "handed_over_walkers", "ties_resolved")   # SRW_STRAT_*


# every symbol include/stellar_rw.h declares
EXPORTS = [
    "srw_create", "srw_destroy", "srw_last_error", "srw_set_stream", "srw_plan_walks", "srw_load_edgelist", "srw_load_coo",
    "srw_load_adjacency", "srw_generate_rmat", "srw_graph_stats", "srw_graph_vertices", "srw_graph_neighbors",
    "srw_graph_partition", "srw_alias_row", "srw_walk", "srw_walk_to_host", "srw_walk_and_save", "srw_host_alloc", "srw_host_free", "srw_fetch_paths", "srw_device_paths", "srw_write_paths",
    "srw_shard_capacity", "srw_shard_vertex_ranks", "srw_shard_layout_for", "srw_shard_begin", "srw_shard_superstep",
    "srw_shard_flush", "srw_shard_finish", "srw_shard_rows_count", "srw_shard_rows_export", "srw_shard_rows_merge",
    "srw_shard_rows_commit", "srw_shard_rows_release", "srw_device_alloc", "srw_device_free", "srw_cluster_create", "srw_cluster_destroy", "srw_cluster_last_error",
    "srw_cluster_shard", "srw_cluster_load_edgelist", "srw_cluster_load_coo", "srw_cluster_generate_rmat",
    "srw_cluster_graph_stats", "srw_cluster_walk", "srw_cluster_fetch_paths", "srw_cluster_walk_and_save",
    "srw_shard_select", "srw_w2v_fit", "srw_w2v_fit_device", "srw_w2v_huffman", "srw_w2v_save", "srw_w2v_save_words", "srw_probe_request_rate", "srw_result_scan_sums", "srw_sample", "srw_second_order_weights",
    "srw_second_order_sample", "srw_rng_uniform", "srw_parse_edgelist", "srw_free", "srw_save_paths", "srw_table_geometry", "srw_version",
]

_lib = None


def lib():
    """Loads libstellar_rw.so.  Fails loudly if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libstellar_rw.so is missing at %s — build it with `make -C %s/csrc` "
                          "(there is no CPU fallback)" % (LIB_PATH, _DIR))
    L = C.CDLL(LIB_PATH)
    vp, i32p, i64p, f32p, u32p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_float), \
        C.POINTER(C.c_uint32)
    L.srw_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.srw_destroy.argtypes = [vp]
    L.srw_destroy.restype = None
    L.srw_last_error.argtypes = [vp]
    L.srw_last_error.restype = C.c_char_p
    L.srw_set_stream.argtypes = [vp, vp]
    L.srw_plan_walks.argtypes = [vp, C.c_int64]
    L.srw_load_edgelist.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.srw_load_coo.argtypes = [vp, i32p, i32p, f32p, i32p, C.c_int64, C.c_int32]
    L.srw_load_adjacency.argtypes = [vp, i32p, i64p, C.c_int64, i32p, f32p, i32p]
    L.srw_generate_rmat.argtypes = [vp, C.c_int32, C.c_int64, C.c_uint32, C.c_int32, C.c_int32]
    L.srw_graph_stats.argtypes = [vp, i64p, i64p]
    L.srw_graph_vertices.argtypes = [vp, i32p]
    L.srw_graph_neighbors.argtypes = [vp, C.c_int32, i32p, f32p, C.c_int64, i64p]
    L.srw_graph_partition.argtypes = [vp, C.c_int32, i32p, i32p]
    L.srw_alias_row.argtypes = [vp, C.c_int32, f32p, i32p, C.c_int64, i64p, i32p]
    L.srw_walk.argtypes = [vp, C.POINTER(WalkParams), C.POINTER(WalkStats)]
    L.srw_walk_to_host.argtypes = [vp, C.POINTER(WalkParams), i32p, i32p, C.POINTER(WalkStats)]
    L.srw_walk_and_save.argtypes = [vp, C.POINTER(WalkParams), C.c_char_p, C.c_int32, C.c_int32, C.POINTER(WalkStats), i64p]
    L.srw_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.srw_host_free.argtypes = [vp]
    L.srw_host_free.restype = None
    L.srw_fetch_paths.argtypes = [vp, i32p, i32p]
    L.srw_device_paths.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), i64p, i32p]
    L.srw_write_paths.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32]
    L.srw_shard_capacity.argtypes = [vp, i64p, i64p]
    L.srw_shard_vertex_ranks.argtypes = [vp, i32p]
    L.srw_shard_layout_for.argtypes = [vp, C.c_int32, C.c_double, C.POINTER(ShardLayout)]
    L.srw_shard_begin.argtypes = [vp, C.POINTER(WalkParams), C.c_int32, C.POINTER(ShardLayout), vp, vp, vp]
    L.srw_shard_superstep.argtypes = [vp, C.POINTER(WalkParams), C.c_int32, C.c_int32, C.POINTER(ShardLayout), vp,
                                      C.POINTER(vp), vp, vp]
    L.srw_shard_flush.argtypes = [vp, C.POINTER(WalkParams), C.c_int32, C.POINTER(ShardLayout), vp, vp, vp]
    L.srw_shard_finish.argtypes = [vp, C.POINTER(WalkStats), i32p]
    L.srw_shard_rows_count.argtypes = [vp, i64p]
    L.srw_shard_rows_export.argtypes = [vp, vp, C.c_int64]
    L.srw_shard_rows_merge.argtypes = [vp, vp, vp, C.c_int64]
    L.srw_shard_rows_commit.argtypes = [vp, vp, C.c_int64, i32p]
    L.srw_shard_rows_release.argtypes = [vp]
    L.srw_device_alloc.argtypes = [vp, C.c_int64, C.POINTER(vp)]
    L.srw_device_free.argtypes = [vp, vp]
    L.srw_cluster_create.argtypes = [i32p, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.srw_cluster_destroy.argtypes = [vp]
    L.srw_cluster_destroy.restype = None
    L.srw_cluster_last_error.argtypes = [vp]
    L.srw_cluster_last_error.restype = C.c_char_p
    L.srw_cluster_shard.argtypes = [vp, C.c_int32]
    L.srw_cluster_shard.restype = vp
    L.srw_cluster_load_edgelist.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.srw_cluster_load_coo.argtypes = [vp, i32p, i32p, f32p, i32p, C.c_int64, C.c_int32]
    L.srw_cluster_generate_rmat.argtypes = [vp, C.c_int32, C.c_int64, C.c_uint32, C.c_int32, C.c_int32]
    L.srw_cluster_graph_stats.argtypes = [vp, i64p, i64p]
    L.srw_cluster_walk.argtypes = [vp, C.POINTER(WalkParams), C.c_int32, C.POINTER(WalkStats)]
    L.srw_cluster_fetch_paths.argtypes = [vp, i32p, i32p]
    L.srw_cluster_walk_and_save.argtypes = [vp, C.POINTER(WalkParams), C.c_char_p, C.c_int32, C.c_int32, C.POINTER(WalkStats)]
    L.srw_shard_select.argtypes = [vp, C.c_int32]
    L.srw_w2v_fit.argtypes = [vp, i32p, i32p, C.c_int64, C.c_int64, C.POINTER(W2vParams), C.POINTER(i32p), C.POINTER(f32p), C.POINTER(C.c_int64)]
    L.srw_w2v_fit_device.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.POINTER(W2vParams), C.POINTER(i32p), C.POINTER(f32p), C.POINTER(C.c_int64)]
    L.srw_w2v_save_words.argtypes = [C.POINTER(C.c_char_p), f32p, C.c_int64, C.c_int32, C.c_char_p, C.c_int32]
    L.srw_w2v_huffman.argtypes = [C.POINTER(C.c_int64), C.c_int64, i32p, C.POINTER(C.c_uint8), i32p]
    L.srw_w2v_save.argtypes = [i32p, f32p, C.c_int64, C.c_int32, C.c_char_p, C.c_int32]
    L.srw_probe_request_rate.argtypes = [vp, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.srw_result_scan_sums.argtypes = [vp, i64p]
    L.srw_sample.argtypes = [vp, f32p, C.c_int64, C.c_float, i64p]
    L.srw_second_order_weights.argtypes = [vp, C.c_float, C.c_float, C.c_int32, i32p, C.c_int64, i32p, f32p,
                                           C.c_int64, f32p]
    L.srw_second_order_sample.argtypes = [vp, C.c_float, C.c_float, C.c_int32, i32p, C.c_int64, i32p, f32p,
                                          C.c_int64, C.c_float, i64p]
    L.srw_rng_uniform.argtypes = [vp, C.c_uint32, u32p, u32p, u32p, C.c_int64, f32p]
    L.srw_parse_edgelist.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(i32p), C.POINTER(i32p),
                                     C.POINTER(f32p), C.POINTER(i32p), i64p, C.c_char_p, C.c_size_t]
    L.srw_free.argtypes = [vp]
    L.srw_free.restype = None
    L.srw_save_paths.argtypes = [i32p, i32p, C.c_int64, C.c_int64, C.c_char_p, C.c_int32, C.c_int32]
    L.srw_table_geometry.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.srw_version.restype = C.c_char_p
    _lib = L
    return L


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def version():
    return lib().srw_version().decode()


def parse_edgelist(path, weighted=True, partitioned=False):
    """Host-only: the edge-list tokenizer (UniformRandomWalk.scala:26-34 / VCutRandomWalk.scala:21-34 rules).
    Returns (src, dst, w, pid) numpy arrays in file order; raises SrwError(ERR_PARSE) where the reference throws."""
    L = lib()
    s, d, p = (C.POINTER(C.c_int32)() for _ in range(3))
    w = C.POINTER(C.c_float)()
    n = C.c_int64(0)
    err = C.create_string_buffer(512)
    rc = L.srw_parse_edgelist(os.fsencode(path), int(weighted), int(partitioned), C.byref(s), C.byref(d), C.byref(w),
                              C.byref(p), C.byref(n), err, 512)
    if rc != OK:
        raise SrwError(rc, err.value.decode())
    k = n.value
    out = tuple(np.ctypeslib.as_array(x, shape=(max(k, 1),))[:k].copy() for x in (s, d, w, p))
    for x in (s, d, w, p):
        L.srw_free(x)
    return out


def save_paths(paths, lens, output_dir, n_parts=1, write_crc=False):
    paths = np.ascontiguousarray(paths, dtype=np.int32)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    rc = lib().srw_save_paths(_i32(paths), _i32(lens), len(lens), paths.shape[1] if paths.ndim == 2 else 0,
                              os.fsencode(output_dir), n_parts, int(write_crc))
    if rc != OK:
        raise SrwError(rc, lib().srw_last_error(None).decode())


class Engine:
    """One handle = one GPU.  Mirrors the life of the reference's SparkContext + GraphMap + RandomWalk object."""

    def __init__(self, device=0, rank=0, world=1, owner_from_partitions=False, compact_ids=False, membership=True, hash_partitioner=False):
        self.h = C.c_void_p()
        cfg = Config(device, rank, world, (CFG_OWNER_FROM_PARTITIONS if owner_from_partitions else 0) |
                     (CFG_COMPACT_IDS if compact_ids else 0) | (0 if membership else CFG_NO_MEMBERSHIP) |
                     (CFG_OWNER_HASH_PARTITIONER if hash_partitioner else 0))
        rc = lib().srw_create(C.byref(cfg), C.byref(self.h))
        if rc != OK:
            self.h = None
            raise SrwError(rc, lib().srw_last_error(None).decode())
        self.rank, self.world = rank, world

    def close(self):
        if getattr(self, "h", None):
            if getattr(self, "_owned", True):
                lib().srw_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: module globals may already be gone
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != OK:
            raise SrwError(rc, lib().srw_last_error(self.h).decode())

    def set_stream(self, stream_ptr):
        self._ck(lib().srw_set_stream(self.h, C.c_void_p(stream_ptr)))

    def plan_walks(self, num_walks):
        """The job's --numWalks (total walk iterations over the tables the next walk builds): steers what is worth building."""
        self._ck(lib().srw_plan_walks(self.h, int(num_walks)))
        return self

    # ---- graph ----
    def load_edgelist(self, path, directed=False, weighted=True, partitioned=False, rdd_partitions=200):
        self._ck(lib().srw_load_edgelist(self.h, os.fsencode(path), int(directed), int(weighted), int(partitioned),
                                         rdd_partitions))
        return self

    def load_coo(self, src, dst, w=None, pid=None, directed=False):
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.ascontiguousarray(dst, dtype=np.int32)
        wp = pp = None
        if w is not None:
            w = np.ascontiguousarray(w, dtype=np.float32)
            wp = _f32(w)
        if pid is not None:
            pid = np.ascontiguousarray(pid, dtype=np.int32)
            pp = _i32(pid)
        self._ck(lib().srw_load_coo(self.h, _i32(src), _i32(dst), wp, pp, len(src), int(directed)))
        return self

    def load_adjacency(self, rows):
        """rows: list of (vid, [(dst, w)] or [(dst, pid, w)]) — the GraphMap.addVertex surface."""
        vids = np.ascontiguousarray([r[0] for r in rows], dtype=np.int32)
        offs = np.zeros(len(rows) + 1, dtype=np.int64)
        ids, ws, pids = [], [], []
        has_pid = any(len(e) == 3 for _, nb in rows for e in nb)
        for i, (_, nb) in enumerate(rows):
            for e in nb:
                ids.append(e[0])
                ws.append(e[-1])
                pids.append(e[1] if len(e) == 3 else -1)
            offs[i + 1] = len(ids)
        ids = np.ascontiguousarray(ids if ids else [0], dtype=np.int32)
        ws = np.ascontiguousarray(ws if ws else [0], dtype=np.float32)
        pids = np.ascontiguousarray(pids if pids else [0], dtype=np.int32)
        self._ck(lib().srw_load_adjacency(self.h, _i32(vids), offs.ctypes.data_as(C.POINTER(C.c_int64)), len(rows),
                                          _i32(ids), _f32(ws), _i32(pids) if has_pid else None))
        return self

    def generate_rmat(self, scale, n_edges=None, seed=42, weighted=False, directed=False):
        if n_edges is None:
            n_edges = 16 << scale
        self._ck(lib().srw_generate_rmat(self.h, scale, n_edges, seed, int(weighted), int(directed)))
        return self

    def stats(self):
        v, e = C.c_int64(0), C.c_int64(0)
        self._ck(lib().srw_graph_stats(self.h, C.byref(v), C.byref(e)))
        return v.value, e.value

    @property
    def num_vertices(self):
        return self.stats()[0]

    @property
    def num_entries(self):
        return self.stats()[1]

    def shard_capacity(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self._ck(lib().srw_shard_capacity(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def vertices(self):
        n = self.shard_capacity()[0]
        out = np.zeros(max(n, 1), dtype=np.int32)
        self._ck(lib().srw_graph_vertices(self.h, _i32(out)))
        return out[:n]

    def neighbors(self, v):
        """GraphMap.getNeighbors: None for an unknown vertex, else (ids, w)."""
        n = C.c_int64(0)
        self._ck(lib().srw_graph_neighbors(self.h, v, None, None, 0, C.byref(n)))
        if n.value < 0:
            return None
        ids = np.zeros(max(n.value, 1), dtype=np.int32)
        w = np.zeros(max(n.value, 1), dtype=np.float32)
        self._ck(lib().srw_graph_neighbors(self.h, v, _i32(ids), _f32(w), n.value, C.byref(n)))
        return ids[:n.value], w[:n.value]

    def partition(self, v):
        pid, known = C.c_int32(0), C.c_int32(0)
        self._ck(lib().srw_graph_partition(self.h, v, C.byref(pid), C.byref(known)))
        return pid.value if known.value else None

    def alias_row(self, v):
        """Mode A table of vertex v: None if absent, else (regular, prob, alias)."""
        n, reg = C.c_int64(0), C.c_int32(0)
        self._ck(lib().srw_alias_row(self.h, v, None, None, 0, C.byref(n), C.byref(reg)))
        if n.value < 0:
            return None
        prob = np.zeros(max(n.value, 1), dtype=np.float32)
        alias = np.zeros(max(n.value, 1), dtype=np.int32)
        self._ck(lib().srw_alias_row(self.h, v, _f32(prob), _i32(alias), n.value, C.byref(n), C.byref(reg)))
        return reg.value, prob[:n.value], alias[:n.value]

    # ---- walk ----
    @staticmethod
    def params(p=1.0, q=1.0, walk_length=80, num_walks=1, first_walk=0, rng="philox", const_r=0.0, seed=42,
               sampler=SAMPLER_REFERENCE, force_general=False, nt_loads=None, occ=0, compact=True, prefix=True, edge_hash=True, binned=True, binned_tune=0, hub_bitmaps=True, device_format=False, edge_tables=True, edge_tables_all=False):
        if sampler == "alias":
            sampler = SAMPLER_ALIAS
        elif sampler == "reference":
            sampler = SAMPLER_REFERENCE
        return WalkParams(np.float32(p), np.float32(q), walk_length, num_walks, first_walk,
                          RNG_CONST if rng == "const" else RNG_PHILOX, np.float32(const_r), seed, sampler,
                          (WALK_FORCE_GENERAL if force_general else 0) | (0 if nt_loads is None else WALK_NT_LOADS if nt_loads else WALK_CACHED_LOADS) | (occ << 8) | (0 if compact else WALK_NO_COMPACT) | (0 if prefix else WALK_NO_PREFIX) | (0 if edge_hash else WALK_NO_EDGE_HASH) | (0 if binned else WALK_NO_BINNED) | (0 if hub_bitmaps else WALK_NO_HUB_BITMAPS) | (WALK_DEVICE_FORMAT if device_format else 0) | (binned_tune << 12) | (0 if edge_tables else WALK_NO_EDGE_TABLES) | (WALK_EDGE_TABLES_ALL if edge_tables_all else 0))

    def walk(self, fetch=True, **kw):
        """Runs srw_walk.  Returns (paths [nWalkers, L+2] int32 (-1 tail), lens, stats dict) or just stats."""
        P = self.params(**kw)
        st = WalkStats()
        self._ck(lib().srw_walk(self.h, C.byref(P), C.byref(st)))
        if not fetch:
            return st.as_dict()
        paths = np.empty((max(st.n_walkers, 1), P.walk_length + 2), dtype=np.int32)
        lens = np.empty(max(st.n_walkers, 1), dtype=np.int32)
        self._ck(lib().srw_fetch_paths(self.h, _i32(paths), _i32(lens)))
        return paths[:st.n_walkers], lens[:st.n_walkers], st.as_dict()

    def walk_to_host(self, pinned=True, **kw):
        """srw_walk_to_host: all num_walks iterations streamed into host buffers (kernel i overlaps the copy of i-1).
        Returns (paths, lens, stats); with pinned=True the arrays are copies of pinned staging memory."""
        P = self.params(**kw)
        nv = self.num_vertices
        n, stride = P.num_walks * nv, P.walk_length + 2
        st = WalkStats()
        if pinned:
            pp, pl = C.c_void_p(), C.c_void_p()
            if lib().srw_host_alloc(max(n * stride * 4, 4), C.byref(pp)) != OK or \
                    lib().srw_host_alloc(max(n * 4, 4), C.byref(pl)) != OK:
                raise SrwError(ERR_NOMEM, "pinned host allocation failed")
            try:
                self._ck(lib().srw_walk_to_host(self.h, C.byref(P), C.cast(pp, C.POINTER(C.c_int32)),
                                                C.cast(pl, C.POINTER(C.c_int32)), C.byref(st)))
                paths = np.ctypeslib.as_array(C.cast(pp, C.POINTER(C.c_int32)), shape=(max(n, 1), stride))[:n].copy()
                lens = np.ctypeslib.as_array(C.cast(pl, C.POINTER(C.c_int32)), shape=(max(n, 1),))[:n].copy()
            finally:
                lib().srw_host_free(pp)
                lib().srw_host_free(pl)
        else:
            paths = np.empty((max(n, 1), stride), dtype=np.int32)
            lens = np.empty(max(n, 1), dtype=np.int32)
            self._ck(lib().srw_walk_to_host(self.h, C.byref(P), _i32(paths), _i32(lens), C.byref(st)))
            paths, lens = paths[:n], lens[:n]
        return paths, lens, st.as_dict()

    def walk_and_save(self, output_dir, n_parts=1, write_crc=False, **kw):
        """srw_walk_and_save: Main.doRandomWalk fused and streamed.  Returns (stats, dead_ends_per_iteration)."""
        P = self.params(**kw)
        st = WalkStats()
        dead = (C.c_int64 * max(P.num_walks, 1))()
        self._ck(lib().srw_walk_and_save(self.h, C.byref(P), os.fsencode(output_dir), n_parts, int(write_crc),
                                         C.byref(st), dead))
        return st.as_dict(), list(dead)[:P.num_walks]

    def device_paths(self):
        dp, dl, n, s = C.c_void_p(), C.c_void_p(), C.c_int64(0), C.c_int32(0)
        self._ck(lib().srw_device_paths(self.h, C.byref(dp), C.byref(dl), C.byref(n), C.byref(s)))
        return dp.value, dl.value, n.value, s.value

    def write_paths(self, output_dir, n_parts=1, write_crc=False):
        self._ck(lib().srw_write_paths(self.h, os.fsencode(output_dir), n_parts, int(write_crc)))

    # ---- the embedding stage (--cmd node2vec / embedding; include/stellar_rw.h: parity unpinned) ----
    def w2v_fit(self, paths, lens, dim=128, window=10, iterations=10, lr=0.025, seed=1, threads=0):
        """Skip-gram + hierarchical softmax over host paths [n, stride] on the GPU: (vocab ids by descending count, vectors [vocab, dim])."""
        paths = np.ascontiguousarray(paths, dtype=np.int32); lens = np.ascontiguousarray(lens, dtype=np.int32)
        P = W2vParams(dim, window, iterations, lr, seed, threads)
        ids, vec, nv = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)(), C.c_int64(0)
        n, stride = (paths.shape[0], paths.shape[1]) if paths.ndim == 2 else (0, 1)
        self._ck(lib().srw_w2v_fit(self.h, _i32(paths), _i32(lens), n, stride, C.byref(P), C.byref(ids), C.byref(vec), C.byref(nv)))
        k = nv.value
        out_ids = np.ctypeslib.as_array(ids, shape=(max(k, 1),))[:k].copy()
        out_vec = np.ctypeslib.as_array(vec, shape=(max(k * dim, 1),))[:k * dim].copy().reshape(k, dim)
        lib().srw_free(ids); lib().srw_free(vec)
        return out_ids, out_vec

    def w2v_fit_device(self, dim=128, window=10, iterations=10, lr=0.025, seed=1, threads=0):
        """The same over the LAST WALK's paths where they are — in HBM (srw_w2v_fit_device with NULL pointers): no PCIe round trip."""
        P = W2vParams(dim, window, iterations, lr, seed, threads)
        ids, vec, nv = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)(), C.c_int64(0)
        self._ck(lib().srw_w2v_fit_device(self.h, None, None, 0, 1, C.byref(P), C.byref(ids), C.byref(vec), C.byref(nv)))
        k = nv.value
        out_ids = np.ctypeslib.as_array(ids, shape=(max(k, 1),))[:k].copy()
        out_vec = np.ctypeslib.as_array(vec, shape=(max(k * dim, 1),))[:k * dim].copy().reshape(k, dim)
        lib().srw_free(ids); lib().srw_free(vec)
        return out_ids, out_vec

    # ---- measurement hooks (bench.py's roofline object) ----
    def probe_request_rate(self, table_bytes=0):
        """(dependent random 16-byte reads per second on this GPU, GiB of table used) — csrc/probe.hip."""
        r, g = C.c_double(0.0), C.c_double(0.0)
        self._ck(lib().srw_probe_request_rate(self.h, int(table_bytes), C.byref(r), C.byref(g)))
        return r.value, g.value

    def result_scan_sums(self):
        """(sum of deg(curr) over the steps, sum of deg(prev) over the second-order steps, steps) of the last srw_walk."""
        out = (C.c_int64 * 3)()
        self._ck(lib().srw_result_scan_sums(self.h, out))
        return int(out[0]), int(out[1]), int(out[2])

    # ---- unit hooks: RandomSample on the GPU ----
    def sample(self, w, r):
        w = np.ascontiguousarray(w, dtype=np.float32)
        k = C.c_int64(0)
        self._ck(lib().srw_sample(self.h, _f32(w), len(w), C.c_float(r), C.byref(k)))
        return k.value

    def second_order_weights(self, p, q, prev_id, prev_ids, curr_ids, curr_w):
        prev_ids = np.ascontiguousarray(prev_ids, dtype=np.int32)
        curr_ids = np.ascontiguousarray(curr_ids, dtype=np.int32)
        curr_w = np.ascontiguousarray(curr_w, dtype=np.float32)
        out = np.empty_like(curr_w)
        self._ck(lib().srw_second_order_weights(self.h, C.c_float(p), C.c_float(q), prev_id, _i32(prev_ids),
                                                len(prev_ids), _i32(curr_ids), _f32(curr_w), len(curr_ids), _f32(out)))
        return out

    def second_order_sample(self, p, q, prev_id, prev_ids, curr_ids, curr_w, r):
        prev_ids = np.ascontiguousarray(prev_ids, dtype=np.int32)
        curr_ids = np.ascontiguousarray(curr_ids, dtype=np.int32)
        curr_w = np.ascontiguousarray(curr_w, dtype=np.float32)
        k = C.c_int64(0)
        self._ck(lib().srw_second_order_sample(self.h, C.c_float(p), C.c_float(q), prev_id, _i32(prev_ids),
                                               len(prev_ids), _i32(curr_ids), _f32(curr_w), len(curr_ids),
                                               C.c_float(r), C.byref(k)))
        return k.value

    def rng_uniform(self, seed, it, src, step):
        it, src, step = (np.ascontiguousarray(x, dtype=np.uint32) for x in (it, src, step))
        out = np.empty(len(it), dtype=np.float32)
        u32p = C.POINTER(C.c_uint32)
        self._ck(lib().srw_rng_uniform(self.h, seed, it.ctypes.data_as(u32p), src.ctypes.data_as(u32p),
                                       step.ctypes.data_as(u32p), len(it), _f32(out)))
        return out


def w2v_huffman(counts):
    """srw_w2v_huffman (host only): [(code bits, syn1 rows)] per word of a vocabulary given by its counts in descending order."""
    cn = np.ascontiguousarray(counts, dtype=np.int64)
    V = len(cn)
    cl = np.zeros(max(V, 1), np.int32); codes = np.zeros((max(V, 1), 40), np.uint8); points = np.zeros((max(V, 1), 40), np.int32)
    rc = lib().srw_w2v_huffman(cn.ctypes.data_as(C.POINTER(C.c_int64)), V, _i32(cl), codes.ctypes.data_as(C.POINTER(C.c_uint8)), _i32(points))
    if rc != OK:
        raise SrwError(rc, lib().srw_last_error(None).decode())
    return [(codes[a, :cl[a]].tolist(), points[a, :cl[a]].tolist()) for a in range(V)]


class Cluster:
    """srw_cluster_*: the vertex-sharded walk inside one process over several devices (peer stores over xGMI, no
    collective).  `devices` may repeat an ordinal: several shards on one GPU (how single-GPU boxes test the protocol)."""

    def __init__(self, devices, owner_from_partitions=False, membership=True, hash_partitioner=False):
        """membership=False (SRW_CFG_NO_MEMBERSHIP): the shards skip the replicated neighbor-id structure; q == 1 walks only."""
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        self.h = C.c_void_p()
        rc = lib().srw_cluster_create(_i32(devs), len(devs), (CFG_OWNER_FROM_PARTITIONS if owner_from_partitions else 0) |
                                      (0 if membership else CFG_NO_MEMBERSHIP) | (CFG_OWNER_HASH_PARTITIONER if hash_partitioner else 0), C.byref(self.h))
        if rc != OK:
            raise SrwError(rc, lib().srw_last_error(None).decode())
        self.world = len(devs)

    def close(self):
        if self.h:
            lib().srw_cluster_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != OK:
            raise SrwError(rc, lib().srw_cluster_last_error(self.h).decode())

    def plan_walks(self, num_walks):
        """srw_plan_walks on every shard: the job's --numWalks."""
        for r in range(self.world):
            rc = lib().srw_plan_walks(C.c_void_p(lib().srw_cluster_shard(self.h, r)), int(num_walks))
            if rc != OK:
                raise SrwError(rc, "srw_plan_walks on shard %d" % r)
        return self

    def load_edgelist(self, path, directed=False, weighted=True, partitioned=False, rdd_partitions=200):
        self._ck(lib().srw_cluster_load_edgelist(self.h, os.fsencode(path), int(directed), int(weighted), int(partitioned),
                                                 rdd_partitions))
        return self

    def load_coo(self, src, dst, w=None, pid=None, directed=False):
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.ascontiguousarray(dst, dtype=np.int32)
        w = None if w is None else np.ascontiguousarray(w, dtype=np.float32)
        pid = None if pid is None else np.ascontiguousarray(pid, dtype=np.int32)
        self._ck(lib().srw_cluster_load_coo(self.h, _i32(src), _i32(dst), None if w is None else _f32(w),
                                            None if pid is None else _i32(pid), len(src), int(directed)))
        return self

    def generate_rmat(self, scale, n_edges=None, seed=42, weighted=False, directed=False):
        self._ck(lib().srw_cluster_generate_rmat(self.h, scale, (16 << scale) if n_edges is None else n_edges, seed,
                                                 int(weighted), int(directed)))
        return self

    def stats(self):
        v, e = C.c_int64(0), C.c_int64(0)
        self._ck(lib().srw_cluster_graph_stats(self.h, C.byref(v), C.byref(e)))
        return v.value, e.value

    def shard(self, rank):
        """Non-owning Engine view of one shard's handle (graph queries, tests)."""
        e = Engine.__new__(Engine)
        e.h = C.c_void_p(lib().srw_cluster_shard(self.h, rank))
        e._owned = False
        e.rank, e.world = rank, self.world
        return e

    def walk(self, fetch=True, batch=0, **kw):
        P = Engine.params(**kw)
        st = WalkStats()
        self._ck(lib().srw_cluster_walk(self.h, C.byref(P), batch, C.byref(st)))
        if not fetch:
            return st.as_dict()
        nv = self.stats()[0]
        n = P.num_walks * nv
        paths = np.empty((max(n, 1), P.walk_length + 2), dtype=np.int32)
        lens = np.empty(max(n, 1), dtype=np.int32)
        self._ck(lib().srw_cluster_fetch_paths(self.h, _i32(paths), _i32(lens)))
        return paths[:n], lens[:n], st.as_dict()

    def walk_and_save(self, output_dir, n_parts=1, write_crc=False, **kw):
        P = Engine.params(**kw)
        st = WalkStats()
        self._ck(lib().srw_cluster_walk_and_save(self.h, C.byref(P), os.fsencode(output_dir), n_parts, int(write_crc), C.byref(st)))
        return st.as_dict()
