"""The embedding stage on the GPU (`--cmd node2vec` / `--cmd embedding`; csrc/embedding.hip).  MLlib's Word2Vec
is a dependency that is absent from the reference tree and seeds itself from the clock: PARITY UNPINNED.  What is checked: the GPU
trainer's sequential mode against the build's CPU restatement (oracle: orc_w2v_fit) within a float tolerance (the dot products are
reduced across the wave, not left to right), the Hogwild mode statistically, and the CLI's files."""
import os
import subprocess

import numpy as np
import pytest

from helpers import pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KARATE = os.path.join(ROOT, "tests", "golden", "karate.txt")


@pytest.fixture(scope="module")
def eng():
    e = pkg().Engine(device=0)
    yield e
    e.close()


def _cos_split(oracle_graph, ids, vec):
    vn = vec / np.maximum(np.linalg.norm(vec, axis=1, keepdims=True), 1e-30)
    idx = {int(v): i for i, v in enumerate(ids)}
    nb, nn = [], []
    for a in ids.tolist():
        na = set(oracle_graph.neighbors(a)[0].tolist())
        for b in ids.tolist():
            if a < b:
                (nb if b in na else nn).append(float(vn[idx[a]] @ vn[idx[b]]))
    return float(np.mean(nb)), float(np.mean(nn))


# (16, 40): a window wider than a register holds (the tokens come from memory); 200 / 300 / 600: 4, 8, 16 floats per lane — with 16 the
# kernel keeps 4 rows in registers and karate's 5-7 node codes run their tail through memory
@pytest.mark.parametrize("dim,window", [(16, 5), (128, 10), (70, 3), (16, 40), (200, 4), (300, 3), (600, 2)])
def test_sequential_mode_matches_the_cpu_restatement(eng, oracle, dim, window):
    g = oracle.Graph.load(KARATE)
    eng.load_edgelist(KARATE, directed=False)
    paths, lens, _ = eng.walk(p=0.5, q=2.0, walk_length=20, num_walks=4, seed=3)
    rp, rl, _ = g.walk(p=0.5, q=2.0, walk_length=20, num_walks=4, seed=3)
    assert np.array_equal(paths, rp) and np.array_equal(lens, rl)
    ids, vec = eng.w2v_fit(paths, lens, dim=dim, window=window, iterations=3, lr=0.025, seed=11, threads=1)
    oids, ovec = oracle.w2v_fit(paths, lens, dim=dim, window=window, iterations=3, lr=0.025, seed=11)
    assert np.array_equal(ids, oids)
    assert np.allclose(vec, ovec, rtol=2e-3, atol=2e-4), float(np.abs(vec - ovec).max())
    # zero iterations: the initial vectors, bit for bit
    i0, v0 = eng.w2v_fit(paths, lens, dim=dim, window=window, iterations=0, seed=11)
    o0, ov0 = oracle.w2v_fit(paths, lens, dim=dim, window=window, iterations=0, seed=11)
    assert np.array_equal(i0, o0) and np.array_equal(v0, ov0)


def test_vocabulary_built_in_chunks_is_the_same_vocabulary(eng, oracle, monkeypatch):
    """more than 2^32 tokens in one fit (numWalks 10 x walkLength 80 from ~5.3 M vertices on) — the vocabulary is built chunk by
    chunk (sort + run-length encode per chunk of 2^30 tokens, runs merged by id, 64-bit counts).  Here: chunks of 37 / 1 000 tokens."""
    g = oracle.Graph.load(KARATE)
    eng.load_edgelist(KARATE, directed=False)
    paths, lens, _ = eng.walk(p=0.5, q=2.0, walk_length=30, num_walks=6, seed=5)
    ref_ids, ref_vec = eng.w2v_fit(paths, lens, dim=32, window=5, iterations=2, lr=0.025, seed=11, threads=1)
    oids, _ = oracle.w2v_fit(paths, lens, dim=32, window=5, iterations=0, seed=11)
    assert np.array_equal(ref_ids, oids)
    for chunk in ("37", "1000"):
        monkeypatch.setenv("SRW_W2V_VOCAB_CHUNK", chunk)
        ids, vec = eng.w2v_fit(paths, lens, dim=32, window=5, iterations=2, lr=0.025, seed=11, threads=1)
        monkeypatch.delenv("SRW_W2V_VOCAB_CHUNK")
        assert np.array_equal(ids, ref_ids) and np.array_equal(vec, ref_vec), chunk


@pytest.mark.parametrize("dim,n_words", [(96, 27), (160, 26), (40, 20)])
def test_deep_huffman_codes(eng, oracle, dim, n_words):
    """Fibonacci counts give the deepest tree a vocabulary can have (code length n_words - 1): the rare words' codes are longer than the
    rows the training kernel keeps in registers (24 at dim <= 128 and <= 256, 32 at dim <= 64), so their last nodes take the memory
    path; the frequent words' codes are one or two nodes (groups with empty slots)."""
    fib = [1, 1]
    while len(fib) < n_words:
        fib.append(fib[-1] + fib[-2])
    rng = np.random.default_rng(n_words)
    toks = np.repeat(np.arange(n_words, dtype=np.int32) + 100, fib)
    rng.shuffle(toks)
    stride = 50
    n = (len(toks) + stride - 1) // stride
    paths = np.full((n, stride), -1, np.int32)
    paths.reshape(-1)[:len(toks)] = toks
    lens = np.full(n, stride, np.int32); lens[-1] = len(toks) - (n - 1) * stride
    ids, vec = eng.w2v_fit(paths, lens, dim=dim, window=2, iterations=1, lr=0.025, seed=5, threads=1)
    oids, ovec = oracle.w2v_fit(paths, lens, dim=dim, window=2, iterations=1, lr=0.025, seed=5)
    assert np.array_equal(ids, oids) and ids[0] == 100 + n_words - 1
    codes = pkg().w2v_huffman(np.sort(np.array(fib, np.int64))[::-1])
    assert max(len(c) for c, _ in codes) == n_words - 1
    err = float(np.abs(vec - ovec).max())
    assert np.allclose(vec, ovec, rtol=5e-3, atol=5e-4), err
    # and the Hogwild launch over the same sentences trains (finite, moved away from the initial vectors)
    _, hv = eng.w2v_fit(paths, lens, dim=dim, window=2, iterations=1, lr=0.025, seed=5)
    _, v0 = eng.w2v_fit(paths, lens, dim=dim, window=2, iterations=0, seed=5)
    assert np.isfinite(hv).all() and float(np.abs(hv - v0).max()) > 1e-5


def test_hogwild_mode_embeds_the_graph(eng, oracle):
    scale = 12
    s, d = oracle.rmat_edges(scale, 8 << scale, seed=5)
    g = oracle.Graph.from_coo(s, d, None, directed=False)
    eng.load_coo(s, d, None, directed=False)
    paths, lens, _ = eng.walk(p=1.0, q=1.0, walk_length=40, num_walks=4, seed=9)
    ids, vec = eng.w2v_fit(paths, lens, dim=64, window=5, iterations=3, lr=0.025, seed=2)
    assert len(ids) == eng.stats()[0] and np.isfinite(vec).all()
    sub = ids[:: max(1, len(ids) // 300)]
    pos = {int(v): i for i, v in enumerate(ids)}
    nb, nn = _cos_split(g, sub, vec[[pos[int(v)] for v in sub]])
    assert nb > nn + 0.05, (nb, nn)


def test_cli_node2vec_and_embedding(tmp_path):
    cli = os.path.join(ROOT, "stellar-random-walk_amd", "stellar-rw")
    out = str(tmp_path / "n2v")
    r = subprocess.run([cli, "--cmd", "node2vec", "--input", KARATE, "--output", out, "--weighted", "false", "--walkLength", "10",
                        "--numWalks", "3", "--dim", "8", "--iter", "2", "--window", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    path_lines = open(os.path.join(out, "path", "part-00000")).read().splitlines()
    assert len(path_lines) == 3 * 34
    vec_lines = open(os.path.join(out, "vec", "part-00000")).read().splitlines()
    assert len(vec_lines) == 34 and all(len(l.split("\t")) == 9 for l in vec_lines)
    assert sorted(int(l.split("\t")[0]) for l in vec_lines) == list(range(1, 35))
    assert os.path.exists(os.path.join(out, "vec", "_SUCCESS")) and os.path.exists(os.path.join(out, "bin", "metadata", "part-00000"))
    # --cmd embedding on those paths: the same vocabulary
    out2 = str(tmp_path / "emb")
    r = subprocess.run([cli, "--cmd", "embedding", "--input", os.path.join(out, "path", "part-00000"), "--output", out2, "--dim", "8",
                        "--iter", "2", "--window", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    v2 = open(os.path.join(out2, "vec", "part-00000")).read().splitlines()
    assert [l.split("\t")[0] for l in v2] == [l.split("\t")[0] for l in vec_lines]
    # an existing <output>/vec: the job fails like saveAsTextFile does
    r = subprocess.run([cli, "--cmd", "embedding", "--input", os.path.join(out, "path", "part-00000"), "--output", out2, "--dim", "8"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "already exists" in r.stderr


def test_fit_from_the_device_resident_walk(eng, oracle):
    """srw_w2v_fit_device on the handle's last walk (paths never leave HBM: M/Main.scala:113-117 hands randomWalk's RDD to Word2Vec.fit)
    == srw_w2v_fit on the fetched paths (one upload, then the same device code) == the oracle's restatement; vocabulary built on the
    device (radix sort + run-length encode + stable sort by count)."""
    scale = 11
    s, d = oracle.rmat_edges(scale, 8 << scale, seed=6)
    eng.load_coo(s, d, None, directed=False)
    paths, lens, _ = eng.walk(p=1.0, q=1.0, walk_length=12, num_walks=2, seed=4)
    ids_d, vec_d = eng.w2v_fit_device(dim=32, window=4, iterations=2, lr=0.025, seed=5, threads=1)
    ids_h, vec_h = eng.w2v_fit(paths, lens, dim=32, window=4, iterations=2, lr=0.025, seed=5, threads=1)
    assert np.array_equal(ids_d, ids_h) and np.array_equal(vec_d, vec_h)            # the same kernels on the same sentences: bit for bit
    oids, ovec = oracle.w2v_fit(paths, lens, dim=32, window=4, iterations=2, lr=0.025, seed=5)
    assert np.array_equal(ids_d, oids)
    assert np.allclose(vec_d, ovec, rtol=2e-3, atol=2e-4), float(np.abs(vec_d - ovec).max())
    # vocabulary order: counts descending, ties by ascending id
    flat = np.concatenate([paths[i, : lens[i]] for i in range(len(lens))])
    u, c = np.unique(flat, return_counts=True)
    order = np.lexsort((u, -c))
    assert np.array_equal(ids_d, u[order])


def test_fit_device_without_a_walk_fails_loudly():
    with pkg().Engine(device=0) as e:
        with pytest.raises(pkg().SrwError):
            e.w2v_fit_device(dim=8)


def _cli(*args, env=None):
    cli = os.path.join(ROOT, "stellar-random-walk_amd", "stellar-rw")
    return subprocess.run([cli, *args], capture_output=True, text=True, env=dict(os.environ, **(env or {})))


def test_cli_embedding_reads_a_directory_of_part_files(tmp_path):
    """context.textFile(input) takes the randomwalk stage's own <output>/path DIRECTORY (part files in name order, _SUCCESS and .crc
    files skipped); round 4 opened it as one file and wrote an empty model with exit code 0."""
    out = str(tmp_path / "rw")
    r = _cli("--cmd", "randomwalk", "--input", KARATE, "--output", out, "--weighted", "false", "--walkLength", "8", "--numWalks", "4",
             "--singleOutput", "false", "--rddPartitions", "3", "--crc", "true")
    assert r.returncode == 0, r.stderr
    names = sorted(os.listdir(os.path.join(out, "path")))
    assert "part-00002" in names and "_SUCCESS" in names and any(n.endswith(".crc") for n in names)
    det = {"SRW_W2V_DETERMINISTIC": "1"}
    r = _cli("--cmd", "embedding", "--input", os.path.join(out, "path"), "--output", str(tmp_path / "e_dir"), "--dim", "8", "--iter", "2", env=det)
    assert r.returncode == 0, r.stderr
    vec_dir = open(os.path.join(str(tmp_path / "e_dir"), "vec", "part-00000")).read()
    assert len(vec_dir.splitlines()) == 34
    # the same text as ONE file gives the same model (deterministic mode), and a second deterministic run is byte-identical
    cat = str(tmp_path / "all.txt")
    with open(cat, "w") as f:
        for n in names:
            if n.startswith("part-"):
                f.write(open(os.path.join(out, "path", n)).read())
    for tag in ("e_file", "e_file2"):
        r = _cli("--cmd", "embedding", "--input", cat, "--output", str(tmp_path / tag), "--dim", "8", "--iter", "2", env=det)
        assert r.returncode == 0, r.stderr
        assert open(os.path.join(str(tmp_path / tag), "vec", "part-00000")).read() == vec_dir
    # nothing to read: an error, not an empty model
    empty = tmp_path / "empty_dir"; empty.mkdir(); (empty / "_SUCCESS").write_text("")
    r = _cli("--cmd", "embedding", "--input", str(empty), "--output", str(tmp_path / "e_none"), "--dim", "8")
    assert r.returncode == 1 and "vocabulary size should be > 0" in r.stderr and not os.path.exists(str(tmp_path / "e_none" / "vec"))
    r = _cli("--cmd", "embedding", "--input", str(tmp_path / "nope"), "--output", str(tmp_path / "e_none2"), "--dim", "8")
    assert r.returncode == 1 and "Input path does not exist" in r.stderr


def test_cli_embedding_takes_words(tmp_path):
    """The reference's Word2Vec takes any token (Main.scala:119-124: textFile(input).map(_.split("\\\\s+"))): a text of words gets its
    vectors under the words; ids out of the int32 range and non-canonical numbers ("007") are words too."""
    txt = tmp_path / "corpus.txt"
    lines = ["the quick brown fox", "the lazy dog 007 99999999999", "quick\tquick  fox the"] * 5
    txt.write_text("\n".join(lines) + "\n")
    out = str(tmp_path / "w")
    r = _cli("--cmd", "embedding", "--input", str(txt), "--output", out, "--dim", "4", "--iter", "1", "--window", "2")
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in open(os.path.join(out, "vec", "part-00000")).read().splitlines()]
    words = [r_[0] for r_ in rows]
    assert sorted(words) == sorted(["the", "quick", "brown", "fox", "lazy", "dog", "007", "99999999999"]) and all(len(r_) == 5 for r_ in rows)
    assert words[:2] == ["quick", "the"]            # by descending count (quick 15, the 15: ties in dictionary order), then fox 10
    assert words[2] == "fox"
    # bin before vec, nothing overwritten: an existing <output>/bin alone fails the job before any vector is written
    out2 = tmp_path / "w2"; (out2 / "bin").mkdir(parents=True)
    r = _cli("--cmd", "embedding", "--input", str(txt), "--output", str(out2), "--dim", "4", "--iter", "1")
    assert r.returncode == 1 and "already exists" in r.stderr and not (out2 / "vec").exists()


def test_cli_embedding_cuts_sentences_at_1000_words(tmp_path):
    """MLlib's maxSentenceLength (1000, not changed by Main.configureWord2Vec): a line of 2 500 words trains as sentences of 1000, 1000
    and 500 — the same model as the three lines written out."""
    rng = np.random.default_rng(1)
    words = [str(int(x)) for x in rng.integers(0, 50, 2500)]
    one = tmp_path / "one.txt"; one.write_text(" ".join(words) + "\n")
    three = tmp_path / "three.txt"; three.write_text("\n".join(" ".join(words[i:i + 1000]) for i in (0, 1000, 2000)) + "\n")
    outs = []
    for name, f in (("a", one), ("b", three)):
        r = _cli("--cmd", "embedding", "--input", str(f), "--output", str(tmp_path / name), "--dim", "8", "--iter", "1", "--window", "3",
                 env={"SRW_W2V_DETERMINISTIC": "1"})
        assert r.returncode == 0, r.stderr
        outs.append(open(os.path.join(str(tmp_path / name), "vec", "part-00000")).read())
    assert outs[0] == outs[1] and len(outs[0].splitlines()) == 50


def test_cli_embedding_fast_id_parser_equals_the_general_one(tmp_path):
    """`--cmd embedding` over canonical int32 tokens (what the randomwalk stage writes) is parsed by all host threads straight into ids;
    anything else goes through the general string path.  Same sentences either way: CR LF line ends, TABs and runs of blanks, trailing
    blanks, negative ids, a line of 2 300 ids (three sentences), several files, a last line without a newline — byte-identical models
    in the deterministic mode; and a file with ONE non-canonical token ("+5") is words in both."""
    rng = np.random.default_rng(5)
    d = tmp_path / "in"; d.mkdir()
    def line(n, sep):
        return sep.join(str(int(x)) for x in rng.integers(-40, 60, n))
    (d / "part-00000").write_text("\n".join(line(int(rng.integers(1, 30)), " ") for _ in range(300)) + "\n")
    (d / "part-00001").write_text("\r\n".join(line(int(rng.integers(1, 30)), "\t") + "  " for _ in range(200)) + "\r\n")
    (d / "part-00002").write_text(line(2300, "  \t") + "\n" + line(7, " "))
    (d / "_SUCCESS").write_text("")
    outs = []
    for tag, env in (("fast", {}), ("general", {"SRW_EMBEDDING_GENERAL_PARSER": "1"})):
        r = _cli("--cmd", "embedding", "--input", str(d), "--output", str(tmp_path / tag), "--dim", "8", "--iter", "2", "--window", "3",
                 env=dict(env, SRW_W2V_DETERMINISTIC="1", SRW_TIMING="1"))
        assert r.returncode == 0, r.stderr
        assert ("parsed by the host threads" in r.stderr) == (tag == "fast")
        outs.append(open(os.path.join(str(tmp_path / tag), "vec", "part-00000")).read())
    assert outs[0] == outs[1] and len(outs[0].splitlines()) == 100
    (d / "part-00003").write_text("1 2 +5\n")
    r = _cli("--cmd", "embedding", "--input", str(d), "--output", str(tmp_path / "words"), "--dim", "8", "--iter", "1", env={"SRW_TIMING": "1"})
    assert r.returncode == 0 and "parsed by the host threads" not in r.stderr
    assert any(l.startswith("+5\t") for l in open(os.path.join(str(tmp_path / "words"), "vec", "part-00000")).read().splitlines())
