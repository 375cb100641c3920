"""ctypes wrapper around oracle/libsrw_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (stellar-random-walk_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libsrw_oracle.so")

RNG_CONST, RNG_PHILOX = 0, 1


def build(force=False):
    src = [os.path.join(_DIR, f) for f in ("srw_oracle.c", "srw_oracle.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return _SO


class WalkParams(C.Structure):
    _fields_ = [("p", C.c_float), ("q", C.c_float), ("walk_length", C.c_int32), ("num_walks", C.c_int32),
                ("rng_mode", C.c_int32), ("const_r", C.c_float), ("seed", C.c_uint32),
                ("first_walk", C.c_int32), ("faithful", C.c_int32), ("threads", C.c_int32), ("sampler", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    i32p, f32p, u32p = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
    L.orc_walk_uniform.restype = C.c_float
    L.orc_walk_uniform.argtypes = [C.c_uint32] * 4
    L.orc_java_random_floats.argtypes = [C.c_int64, C.c_int, f32p]
    L.orc_sample_index.restype = C.c_int64
    L.orc_sample_index.argtypes = [f32p, C.c_int64, C.c_float]
    L.orc_second_order_weights.argtypes = [C.c_float, C.c_float, C.c_int32, i32p, C.c_int64, i32p, f32p,
                                           C.c_int64, f32p]
    L.orc_second_order_sample_index.restype = C.c_int64
    L.orc_second_order_sample_index.argtypes = [C.c_float, C.c_float, C.c_int32, i32p, C.c_int64, i32p, f32p,
                                                C.c_int64, C.c_float]
    L.orc_graphmap_new.restype = C.c_void_p
    for n in ("free", "reset"):
        getattr(L, "orc_graphmap_" + n).argtypes = [C.c_void_p]
    L.orc_graphmap_add_vertex.argtypes = [C.c_void_p, C.c_int32, i32p, f32p, C.c_int64]
    L.orc_graphmap_add_vertex_p.argtypes = [C.c_void_p, C.c_int32, i32p, i32p, f32p, C.c_int64]
    L.orc_graphmap_num_vertices.restype = C.c_int64
    L.orc_graphmap_num_vertices.argtypes = [C.c_void_p]
    L.orc_graphmap_num_edges.restype = C.c_int64
    L.orc_graphmap_num_edges.argtypes = [C.c_void_p]
    L.orc_graphmap_get_neighbors.restype = C.c_int64
    L.orc_graphmap_get_neighbors.argtypes = [C.c_void_p, C.c_int32, i32p, f32p, C.c_int64]
    L.orc_graphmap_get_partition.argtypes = [C.c_void_p, C.c_int32, i32p]
    L.orc_graph_load_edgelist.restype = C.c_void_p
    L.orc_graph_load_edgelist.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    L.orc_graph_from_coo.restype = C.c_void_p
    L.orc_graph_from_coo.argtypes = [i32p, i32p, f32p, C.c_int64, C.c_int]
    L.orc_graph_free.argtypes = [C.c_void_p]
    for n in ("num_vertices", "num_entries", "num_lines"):
        f = getattr(L, "orc_graph_" + n)
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p]
    L.orc_graph_vertices.argtypes = [C.c_void_p, i32p]
    L.orc_graph_degree.restype = C.c_int64
    L.orc_graph_degree.argtypes = [C.c_void_p, C.c_int32]
    L.orc_graph_neighbors.restype = C.c_int64
    L.orc_graph_neighbors.argtypes = [C.c_void_p, C.c_int32, i32p, f32p, C.c_int64]
    L.orc_graph_lines.argtypes = [C.c_void_p, i32p, i32p, f32p, i32p]
    L.orc_walk.restype = C.c_int64
    L.orc_walk.argtypes = [C.c_void_p, C.POINTER(WalkParams), i32p, C.c_int64, i32p, i32p]
    L.orc_seq_walk.restype = C.c_int32
    L.orc_seq_walk.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(WalkParams), i32p]
    L.orc_write_paths.argtypes = [i32p, i32p, C.c_int64, C.c_int64, C.c_char_p, C.c_int]
    L.orc_alias_row.argtypes = [f32p, C.c_int64, f32p, i32p]
    L.orc_graph_alias_row.argtypes = [C.c_void_p, C.c_int32, f32p, i32p]
    L.orc_w2v_fit.argtypes = [i32p, i32p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint32,
                              C.POINTER(i32p), C.POINTER(f32p), C.POINTER(C.c_int64)]
    L.orc_free.argtypes = [C.c_void_p]
    u8p = C.POINTER(C.c_uint8)
    L.orc_w2v_huffman.argtypes = [C.POINTER(C.c_int64), C.c_int64, i32p, u8p, i32p]
    L.orc_w2v_exp_table.argtypes = [f32p]
    L.orc_w2v_pair_update.argtypes = [C.c_int32, f32p, f32p, u8p, C.c_int32, C.c_float]
    L.orc_rmat_edges.argtypes = [C.c_int, C.c_uint32, C.c_int64, C.c_int64, i32p, i32p]
    L.orc_rmat_weight.restype = C.c_float
    L.orc_rmat_weight.argtypes = [C.c_int32, C.c_int32, C.c_uint32]
    _lib = L
    return L


def _i32(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def walk_uniform(seed, it, src, step):
    return float(lib().orc_walk_uniform(seed, it, src & 0xFFFFFFFF, step))


def java_random_floats(seed, n):
    out = np.zeros(n, dtype=np.float32)
    lib().orc_java_random_floats(seed, n, _f32(out))
    return out


def sample_index(w, r):
    w = np.ascontiguousarray(w, dtype=np.float32)
    return int(lib().orc_sample_index(_f32(w), len(w), C.c_float(r)))


def second_order_weights(p, q, prev_id, prev_ids, curr_ids, curr_w):
    prev_ids = np.ascontiguousarray(prev_ids, dtype=np.int32)
    curr_ids = np.ascontiguousarray(curr_ids, dtype=np.int32)
    curr_w = np.ascontiguousarray(curr_w, dtype=np.float32)
    out = np.empty_like(curr_w)
    lib().orc_second_order_weights(C.c_float(p), C.c_float(q), prev_id, _i32(prev_ids), len(prev_ids),
                                   _i32(curr_ids), _f32(curr_w), len(curr_ids), _f32(out))
    return out


def second_order_sample_index(p, q, prev_id, prev_ids, curr_ids, curr_w, r):
    prev_ids = np.ascontiguousarray(prev_ids, dtype=np.int32)
    curr_ids = np.ascontiguousarray(curr_ids, dtype=np.int32)
    curr_w = np.ascontiguousarray(curr_w, dtype=np.float32)
    return int(lib().orc_second_order_sample_index(C.c_float(p), C.c_float(q), prev_id, _i32(prev_ids),
                                                   len(prev_ids), _i32(curr_ids), _f32(curr_w), len(curr_ids),
                                                   C.c_float(r)))


class GraphMap:
    """Restatement of the GraphMap singleton (M/algorithm/GraphMap.scala)."""

    def __init__(self):
        self.h = lib().orc_graphmap_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_graphmap_free(self.h)
            self.h = None

    def reset(self):
        lib().orc_graphmap_reset(self.h)

    def add_vertex(self, v, neighbors=()):
        ids = np.ascontiguousarray([e[0] for e in neighbors], dtype=np.int32)
        if neighbors and len(neighbors[0]) == 3:
            pids = np.ascontiguousarray([e[1] for e in neighbors], dtype=np.int32)
            w = np.ascontiguousarray([e[2] for e in neighbors], dtype=np.float32)
            lib().orc_graphmap_add_vertex_p(self.h, v, _i32(ids), _i32(pids), _f32(w), len(ids))
        else:
            w = np.ascontiguousarray([e[1] for e in neighbors], dtype=np.float32)
            lib().orc_graphmap_add_vertex(self.h, v, _i32(ids), _f32(w), len(ids))

    @property
    def num_vertices(self):
        return int(lib().orc_graphmap_num_vertices(self.h))

    @property
    def num_edges(self):
        return int(lib().orc_graphmap_num_edges(self.h))

    def get_neighbors(self, v):
        n = int(lib().orc_graphmap_get_neighbors(self.h, v, None, None, 0))
        if n < 0:
            return None
        ids = np.zeros(max(n, 1), dtype=np.int32)
        w = np.zeros(max(n, 1), dtype=np.float32)
        lib().orc_graphmap_get_neighbors(self.h, v, _i32(ids), _f32(w), n)
        return [(int(ids[k]), float(w[k])) for k in range(n)]

    def get_partition(self, v):
        p = C.c_int32(0)
        return int(p.value) if lib().orc_graphmap_get_partition(self.h, v, C.byref(p)) else None


class Graph:
    def __init__(self, handle):
        if not handle:
            raise ValueError("oracle graph construction failed")
        self.h = handle

    @classmethod
    def load(cls, path, directed=False, weighted=True, partitioned=False):
        err = C.create_string_buffer(512)
        h = lib().orc_graph_load_edgelist(os.fsencode(path), int(directed), int(weighted), int(partitioned),
                                          err, 512)
        if not h:
            raise ValueError(err.value.decode())
        return cls(h)

    @classmethod
    def from_coo(cls, src, dst, w=None, directed=False):
        src = np.ascontiguousarray(src, dtype=np.int32)
        dst = np.ascontiguousarray(dst, dtype=np.int32)
        wp = None
        if w is not None:
            w = np.ascontiguousarray(w, dtype=np.float32)
            wp = _f32(w)
        return cls(lib().orc_graph_from_coo(_i32(src), _i32(dst), wp, len(src), int(directed)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_graph_free(self.h)
            self.h = None

    @property
    def num_vertices(self):
        return int(lib().orc_graph_num_vertices(self.h))

    @property
    def num_entries(self):
        return int(lib().orc_graph_num_entries(self.h))

    @property
    def num_lines(self):
        return int(lib().orc_graph_num_lines(self.h))

    def vertices(self):
        out = np.zeros(max(self.num_vertices, 1), dtype=np.int32)
        lib().orc_graph_vertices(self.h, _i32(out))
        return out[:self.num_vertices]

    def degree(self, v):
        return int(lib().orc_graph_degree(self.h, v))

    def neighbors(self, v):
        n = self.degree(v)
        if n < 0:
            return None
        ids = np.zeros(max(n, 1), dtype=np.int32)
        w = np.zeros(max(n, 1), dtype=np.float32)
        lib().orc_graph_neighbors(self.h, v, _i32(ids), _f32(w), n)
        return ids[:n], w[:n]

    def alias_row(self, v):
        n = self.degree(v)
        if n < 0:
            return None
        prob = np.zeros(max(n, 1), dtype=np.float32)
        alias = np.zeros(max(n, 1), dtype=np.int32)
        reg = int(lib().orc_graph_alias_row(self.h, v, _f32(prob), _i32(alias)))
        return reg, prob[:n], alias[:n]

    def lines(self):
        n = self.num_lines
        s, d, p = (np.zeros(max(n, 1), dtype=np.int32) for _ in range(3))
        w = np.zeros(max(n, 1), dtype=np.float32)
        lib().orc_graph_lines(self.h, _i32(s), _i32(d), _f32(w), _i32(p))
        return s[:n], d[:n], w[:n], p[:n]

    def params(self, p=1.0, q=1.0, walk_length=80, num_walks=1, rng="philox", const_r=0.0, seed=42,
               first_walk=0, faithful=False, threads=1, sampler=0):
        return WalkParams(np.float32(p), np.float32(q), walk_length, num_walks,
                          RNG_CONST if rng == "const" else RNG_PHILOX, np.float32(const_r), seed, first_walk,
                          int(faithful), threads, int(sampler))

    def walk(self, sources=None, **kw):
        """Returns (paths [nWalkers, L+2] int32 with -1 padding, lens, steps)."""
        P = self.params(**kw)
        ns = self.num_vertices if sources is None else len(sources)
        nw = P.num_walks * ns
        paths = np.full((max(nw, 1), P.walk_length + 2), -1, dtype=np.int32)
        lens = np.zeros(max(nw, 1), dtype=np.int32)
        sp = None
        if sources is not None:
            sources = np.ascontiguousarray(sources, dtype=np.int32)
            sp = _i32(sources)
        steps = int(lib().orc_walk(self.h, C.byref(P), sp, ns, _i32(paths), _i32(lens)))
        return paths[:nw], lens[:nw], steps

    def seq_walk(self, src, it=0, **kw):
        P = self.params(**kw)
        out = np.full(P.walk_length + 2, -1, dtype=np.int32)
        n = int(lib().orc_seq_walk(self.h, src, it, C.byref(P), _i32(out)))
        return out[:n]


def alias_row(w):
    """Mode A exact-integer alias table of one weight list: (regular, prob, alias)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    prob = np.zeros(max(len(w), 1), dtype=np.float32)
    alias = np.zeros(max(len(w), 1), dtype=np.int32)
    reg = int(lib().orc_alias_row(_f32(w), len(w), _f32(prob), _i32(alias)))
    return reg, prob[:len(w)], alias[:len(w)]


def write_paths(paths, lens, output_dir, n_parts=1):
    paths = np.ascontiguousarray(paths, dtype=np.int32)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    return int(lib().orc_write_paths(_i32(paths), _i32(lens), len(lens), paths.shape[1], os.fsencode(output_dir),
                                     n_parts))


def rmat_edges(scale, n_edges, seed=42, first=0):
    s = np.zeros(n_edges, dtype=np.int32)
    d = np.zeros(n_edges, dtype=np.int32)
    L = lib()
    if n_edges < (1 << 22):
        L.orc_rmat_edges(scale, seed, first, n_edges, _i32(s), _i32(d))
        return s, d
    # edge i is a pure function of (seed, i): slices on threads (ctypes releases the GIL)
    import concurrent.futures as cf
    nt = min(64, os.cpu_count() or 1)
    step = -(-n_edges // (nt * 4))
    def work(lo):
        n = min(step, n_edges - lo)
        L.orc_rmat_edges(scale, seed, first + lo, n, _i32(s[lo:lo + n]), _i32(d[lo:lo + n]))
    with cf.ThreadPoolExecutor(nt) as ex:
        list(ex.map(work, range(0, n_edges, step)))
    return s, d


def rmat_weight(u, v, seed=42):
    return float(lib().orc_rmat_weight(int(u), int(v), seed))


def w2v_fit(paths, lens, dim=128, window=10, iterations=10, lr=0.025, seed=1):
    """Sequential restatement of the embedding stage (PARITY UNPINNED, see srw_oracle.c): (vocab ids, vectors [vocab, dim])."""
    paths = np.ascontiguousarray(paths, dtype=np.int32); lens = np.ascontiguousarray(lens, dtype=np.int32)
    ids, vec, nv = C.POINTER(C.c_int32)(), C.POINTER(C.c_float)(), C.c_int64(0)
    lib().orc_w2v_fit(_i32(paths), _i32(lens), paths.shape[0], paths.shape[1], dim, window, iterations, C.c_float(lr), seed,
                      C.byref(ids), C.byref(vec), C.byref(nv))
    n = nv.value
    out_ids = np.ctypeslib.as_array(ids, shape=(max(n, 1),))[:n].copy()
    out_vec = np.ctypeslib.as_array(vec, shape=(max(n * dim, 1),))[:n * dim].copy().reshape(n, dim)
    lib().orc_free(ids); lib().orc_free(vec)
    return out_ids, out_vec


def w2v_huffman(counts):
    """word2vec.c CreateBinaryTree over counts in descending order -> list of (code bits, syn1 rows) per word, root first."""
    cn = np.ascontiguousarray(counts, dtype=np.int64)
    V = len(cn)
    cl = np.zeros(V, np.int32); codes = np.zeros((V, 40), np.uint8); points = np.zeros((V, 40), np.int32)
    lib().orc_w2v_huffman(cn.ctypes.data_as(C.POINTER(C.c_int64)), V, _i32(cl), codes.ctypes.data_as(C.POINTER(C.c_uint8)), _i32(points))
    return [(codes[a, :cl[a]].tolist(), points[a, :cl[a]].tolist()) for a in range(V)]


def w2v_exp_table():
    t = np.zeros(1000, np.float32)
    lib().orc_w2v_exp_table(_f32(t))
    return t


def w2v_pair_update(syn0_row, syn1_rows, code, alpha):
    """One (centre, context) pair of skip-gram + hierarchical softmax: returns the updated (syn0 row, syn1 rows)."""
    r0 = np.ascontiguousarray(syn0_row, dtype=np.float32).copy()
    r1 = np.ascontiguousarray(syn1_rows, dtype=np.float32).copy()
    cb = np.ascontiguousarray(code, dtype=np.uint8)
    lib().orc_w2v_pair_update(r0.shape[0], _f32(r0), _f32(r1), cb.ctypes.data_as(C.POINTER(C.c_uint8)), r1.shape[0], C.c_float(alpha))
    return r0, r1
