import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _pkg  # noqa: E402


def pkg():
    return _pkg.load()


def digest(paths, lens):
    lines = sorted("\t".join(str(int(x)) for x in p[:n]) + "\n" for p, n in zip(paths, lens))
    return hashlib.sha256("".join(lines).encode()).hexdigest()[:16]


def rmat_lines(oracle, scale, edge_factor=8, seed=42, weighted=False):
    n = edge_factor << scale
    s, d = oracle.rmat_edges(scale, n, seed=seed)
    w = None
    if weighted:
        w = np.array([oracle.rmat_weight(a, b, seed) for a, b in zip(s, d)], dtype=np.float32)
    return s, d, w


def random_multigraph(rng, n_vertices, n_lines, weighted, id_lo=0):
    """Random small multigraph with self-loops, duplicate lines and unused ids."""
    ids = rng.integers(id_lo, id_lo + n_vertices, size=(n_lines, 2)).astype(np.int32)
    # force some self loops and duplicates
    for k in range(0, n_lines, 7):
        ids[k, 1] = ids[k, 0]
    for k in range(3, n_lines, 5):
        ids[k] = ids[k - 1]
    w = None
    if weighted:
        w = rng.choice(np.array([0.5, 1.0, 1.5, 2.0, 3.25, 7.0, 0.001, 1000.0], dtype=np.float32), size=n_lines)
    return ids[:, 0].copy(), ids[:, 1].copy(), w


def rmat_weights_np(s, d, seed=42):
    """Vectorised oracle.rmat_weight (oracle/srw_oracle.c:orc_rmat_weight): w = 1 + (mix32(min, max, seed) & 15)."""
    a = np.minimum(s, d).astype(np.uint32)
    b = np.maximum(s, d).astype(np.uint32)
    with np.errstate(over="ignore"):
        x = b * np.uint32(0x85EBCA77)
        h = np.uint32(seed) ^ (a * np.uint32(0x9E3779B1)) ^ ((x << np.uint32(13)) | (x >> np.uint32(19)))
        h ^= h >> np.uint32(16); h *= np.uint32(0x85EBCA6B); h ^= h >> np.uint32(13)
        h *= np.uint32(0xC2B2AE35); h ^= h >> np.uint32(16)
    return (np.uint32(1) + (h & np.uint32(15))).astype(np.float32)
