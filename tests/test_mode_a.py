"""Mode A (alias tables + rejection) — build-defined, NOT in the reference.

CPU part: the oracle's exact-integer alias construction reproduces the weights exactly as a distribution.
GPU part (-m gpu): device-built tables equal the oracle's bit for bit (float construction tolerance asked by
north_star: 1e-6; achieved: 0), Mode A walks equal the oracle's Mode A walks bit for bit, and Mode A's transition
statistics agree with the reference-exact Mode R (chi-square)."""
from fractions import Fraction

import numpy as np
import pytest

from conftest import KARATE
from helpers import pkg, random_multigraph, rmat_lines


def _implied(prob, alias):
    n = len(prob)
    P = [Fraction(0)] * n
    for j in range(n):
        pj = Fraction(float(prob[j]))
        P[j] += pj / n
        P[int(alias[j])] += (1 - pj) / n
    return P


@pytest.mark.parametrize("w", [[1, 1, 1], [1, 2, 3, 4], [0.5, 0.25, 8, 1, 1], [3], [1, 0, 1], [7, 1, 1, 1, 1, 1, 1, 1],
                               [1e-3, 1, 1000], list(range(1, 40)), [5] * 64 + [1] * 64])
def test_oracle_alias_table_is_the_distribution(oracle, w):
    reg, prob, alias = oracle.alias_row(w)
    assert reg == 1
    S = sum(Fraction(float(np.float32(x))) for x in w)
    for k, pk in enumerate(_implied(prob, alias)):
        assert abs(float(pk - Fraction(float(np.float32(w[k]))) / S)) < 1e-6   # f32 rounding of prob only
    assert all(0.0 <= p <= 1.0 for p in prob) and all(0 <= a < len(w) for a in alias)


def test_oracle_alias_irregular_rows(oracle):
    for w in ([0, 0], [1, -1], [float("nan"), 1], [float("inf"), 1], [1e-30, 1e30]):
        assert oracle.alias_row(w)[0] == 0


def test_oracle_alias_uniform_rows_are_trivial(oracle):
    reg, prob, alias = oracle.alias_row([2.5] * 17)
    assert reg == 1 and prob.tolist() == [1.0] * 17 and alias.tolist() == list(range(17))


def test_oracle_mode_a_walk_is_deterministic_and_valid(oracle):
    g = oracle.Graph.load(KARATE)
    a = g.walk(walk_length=20, p=0.25, q=4.0, sampler=1, seed=3, num_walks=2)
    b = g.walk(walk_length=20, p=0.25, q=4.0, sampler=1, seed=3, num_walks=2, threads=4)
    assert np.array_equal(a[0], b[0]) and (a[1] == 22).all()
    for path in a[0][:10]:
        for x, y in zip(path[:-1], path[1:]):
            assert int(y) in g.neighbors(int(x))[0].tolist()


# ---------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = pkg().Engine(device=0)
    yield e
    e.close()


def _same_tables(eng, g, verts):
    for v in verts:
        a, b = eng.alias_row(int(v)), g.alias_row(int(v))
        assert a[0] == b[0], v
        if b[0]:
            assert a[1].view(np.uint32).tolist() == b[1].view(np.uint32).tolist(), v
            assert a[2].tolist() == b[2].tolist(), v


@gpu
def test_gpu_alias_tables_equal_oracle(eng, oracle):
    g = oracle.Graph.load(KARATE)
    eng.load_edgelist(KARATE)
    _same_tables(eng, g, g.vertices())
    s, d, w = rmat_lines(oracle, 12, edge_factor=16, weighted=True)
    g = oracle.Graph.from_coo(s, d, w)
    eng.load_coo(s, d, w)
    verts = g.vertices()
    degs = np.array([g.degree(int(v)) for v in verts])
    big = verts[np.argsort(-degs)[:12]]                     # hubs: > 2048 entries -> HBM staging path
    assert degs.max() > 2048
    _same_tables(eng, g, list(big) + list(verts[::97]))
    rng = np.random.default_rng(3)
    s, d, w = random_multigraph(rng, 50, 400, True)
    w[::11] = 0.0                                            # zero weights stay regular
    g = oracle.Graph.from_coo(s, d, w, directed=True)
    eng.load_coo(s, d, w, directed=True)
    _same_tables(eng, g, g.vertices())


@gpu
def test_gpu_alias_irregular_rows(eng, oracle):
    s = np.array([1, 1, 1, 2, 2, 3, 3, 4, 4, 4], dtype=np.int32)
    d = np.array([2, 3, 4, 1, 3, 1, 4, 1, 2, 3], dtype=np.int32)
    w = np.array([1.0, -1.0, 2.0, 0.0, 0.0, np.nan, 1.0, 3.0, 1.0, 1.0], dtype=np.float32)
    g = oracle.Graph.from_coo(s, d, w, directed=True)
    eng.load_coo(s, d, w, directed=True)
    _same_tables(eng, g, g.vertices())
    assert [eng.alias_row(v)[0] for v in (1, 2, 3, 4)] == [0, 0, 0, 1]
    for p, q in [(1.0, 1.0), (0.25, 4.0)]:
        a = eng.walk(walk_length=12, num_walks=5, seed=8, p=p, q=q, sampler="alias")
        b = g.walk(walk_length=12, num_walks=5, seed=8, p=p, q=q, sampler=1)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


PQ = [(1.0, 1.0), (0.25, 4.0), (4.0, 0.5), (2.0, 2.0), (0.5, 1.0)]


@gpu
@pytest.mark.parametrize("p,q", PQ)
def test_gpu_mode_a_walk_equals_oracle(eng, oracle, p, q):
    for directed in (False, True):
        g = oracle.Graph.load(KARATE, directed=directed)
        eng.load_edgelist(KARATE, directed=directed)
        a = eng.walk(walk_length=40, num_walks=3, first_walk=1, seed=5, p=p, q=q, sampler="alias")
        b = g.walk(walk_length=40, num_walks=3, first_walk=1, seed=5, p=p, q=q, sampler=1)
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
        assert a[2]["kernel_kind"] == 3 and a[2]["n_steps"] == b[2] and a[2]["trials"] >= a[2]["n_steps"]
    s, d, w = rmat_lines(oracle, 11, edge_factor=8, weighted=True)
    g = oracle.Graph.from_coo(s, d, w)
    eng.load_coo(s, d, w)
    for nt in (True, False):
        for eh in (True, False):        # membership through the edge hash set / through the sorted rows
            a = eng.walk(walk_length=20, seed=9, p=p, q=q, sampler="alias", nt_loads=nt, edge_hash=eh)
            b = g.walk(walk_length=20, seed=9, p=p, q=q, sampler=1, threads=8)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])


@gpu
@pytest.mark.parametrize("p,q", [(0.25, 4.0), (4.0, 0.5), (1.0, 1.0)])
def test_mode_a_statistics_match_mode_r(eng, p, q):
    # two-sample chi-square on (v1, v2, v3) triples: Mode A (alias + rejection) vs Mode R (reference-exact)
    eng.load_edgelist(KARATE)
    nw = 4000
    ra = eng.walk(walk_length=1, num_walks=nw, seed=11, p=p, q=q, sampler="alias")[0]
    rr = eng.walk(walk_length=1, num_walks=nw, seed=12, p=p, q=q, sampler="reference")[0]
    def hist(paths):
        key = paths[:, 0].astype(np.int64) * 10000 + paths[:, 1].astype(np.int64) * 100 + paths[:, 2]
        u, c = np.unique(key, return_counts=True)
        return dict(zip(u.tolist(), c.tolist()))
    ha, hr = hist(ra), hist(rr)
    stat, dof = 0.0, 0
    for k in set(ha) | set(hr):
        a, r = ha.get(k, 0), hr.get(k, 0)
        if a + r >= 40:
            stat += (a - r) ** 2 / (a + r)
            dof += 1
    assert dof > 100
    assert stat < dof + 5.0 * np.sqrt(2.0 * dof), (stat, dof)


def test_oracle_alias_fuzz(oracle):
    rng = np.random.default_rng(0)
    for trial in range(400):
        n = int(rng.integers(1, 60))
        kind = trial % 4
        if kind == 0:
            w = rng.integers(1, 6, n).astype(np.float32)
        elif kind == 1:
            w = rng.integers(0, 4, n).astype(np.float32)
            w[0] = max(w[0], 1)
        elif kind == 2:
            w = (rng.random(n) * 8).astype(np.float32)
        else:
            w = rng.choice(np.array([0.25, 0.5, 1, 2, 4], dtype=np.float32), n)   # many exact ties (D_i == E_j)
        reg, prob, alias = oracle.alias_row(w)
        if not reg:
            continue
        S = sum(Fraction(float(x)) for x in w)
        for k, pk in enumerate(_implied(prob, alias)):
            assert abs(float(pk - Fraction(float(w[k])) / S)) < 1e-6, (w, prob, alias)
