"""Known answers for the restatement of the embedding stage (M/Main.scala:36-44,113-124 hand the paths to MLlib's
Word2Vec, which is absent from /root/reference: PARITY UNPINNED against the reference).  What CAN be pinned is the published
algorithm MLlib ports — word2vec.c's CreateBinaryTree, its expTable and one skip-gram / hierarchical-softmax update — worked out BY
HAND below (not by running either implementation), and held against both the oracle's restatement (oracle/srw_oracle.c:w2v_build_tree,
w2v_exp_table, w2v_pair: the functions orc_w2v_fit itself runs) and the product's host-side Huffman builder (srw_w2v_huffman, the
function the GPU trainer takes its codes from).  No GPU needed."""
import math

import numpy as np

from helpers import pkg


# ---- CreateBinaryTree on counts (5, 4, 3, 2, 1), V = 5, by hand ------------------------------------------------------------------
# count[] = 5 4 3 2 1 | inf inf inf inf;  pos1 = 4, pos2 = 5
#  a = 0: min1 = 4 (1), min2 = 3 (2)            -> node 5 = 3,  binary[3] = 1
#  a = 1: count[2] = 3 < count[5] = 3 is FALSE  -> min1 = node 5, then min2 = 2 (3)   -> node 6 = 6,  binary[2] = 1
#  a = 2: min1 = 1 (4), min2 = 0 (5)            -> node 7 = 9,  binary[0] = 1
#  a = 3: min1 = node 6, min2 = node 7          -> node 8 = 15 (root = 2V - 2), binary[7] = 1
# parents: 4 -> 5, 3 -> 5, 5 -> 6, 2 -> 6, 1 -> 7, 0 -> 7, 6 -> 8, 7 -> 8.  Codes are read root first; points are node - V (rows of
# syn1), the root (V - 2 = 3) first.
HAND_TREE = [
    ([1, 1], [3, 2]),            # word 0 (count 5): root -> node 7 (bit 1) -> leaf (bit 1)
    ([1, 0], [3, 2]),            # word 1 (count 4)
    ([0, 1], [3, 1]),            # word 2 (count 3): root -> node 6 (bit 0) -> leaf (bit 1)
    ([0, 0, 1], [3, 1, 0]),      # word 3 (count 2): root -> node 6 -> node 5 -> leaf (bit 1)
    ([0, 0, 0], [3, 1, 0]),      # word 4 (count 1)
]


def test_huffman_five_words_oracle(oracle):
    assert oracle.w2v_huffman([5, 4, 3, 2, 1]) == HAND_TREE


def test_huffman_five_words_product():
    assert pkg().w2v_huffman([5, 4, 3, 2, 1]) == HAND_TREE


def test_huffman_two_words_and_ties(oracle):
    # V = 2: one inner node (row 0 of syn1), the SECOND minimum gets bit 1: word 1 (the rarer: min1) -> 0, word 0 -> 1
    two = [([1], [0]), ([0], [0])]
    assert oracle.w2v_huffman([7, 3]) == two and pkg().w2v_huffman([7, 3]) == two
    # equal counts (4 x 1): a = 0 joins leaves 3, 2 (node 4 = 2); a = 1: count[1] = 1 < 2 -> leaves 1, 0 (node 5 = 2); a = 2: nodes 4, 5.
    # every word: 2 bits; leaves 3 / 1 are min1 (bit 0), leaves 2 / 0 min2 (bit 1); node 4 is min1 of the root (bit 0), node 5 bit 1
    four = [([1, 1], [2, 1]), ([1, 0], [2, 1]), ([0, 1], [2, 0]), ([0, 0], [2, 0])]
    assert oracle.w2v_huffman([1, 1, 1, 1]) == four and pkg().w2v_huffman([1, 1, 1, 1]) == four


def test_huffman_product_equals_oracle_on_random_counts(oracle):
    rng = np.random.default_rng(5)
    for V in (3, 17, 1000):
        cn = np.sort(rng.integers(1, 50, V))[::-1].copy()
        assert pkg().w2v_huffman(cn) == oracle.w2v_huffman(cn)
    # a Zipf vocabulary whose rarest words sit deep in the tree (code lengths up to ~20)
    cn = np.maximum(1, (1e6 / np.arange(1, 5001) ** 1.2).astype(np.int64))
    a, b = pkg().w2v_huffman(cn), oracle.w2v_huffman(cn)
    assert a == b and max(len(c) for c, _ in a) >= 15
    # Kraft equality: a full binary tree
    assert abs(sum(2.0 ** -len(c) for c, _ in a) - 1.0) < 1e-9


def test_exp_table(oracle):
    t = oracle.w2v_exp_table()
    # expTable[i] = sigmoid((i / 1000 * 2 - 1) * 6): closed forms at the ends and in the middle
    assert abs(t[0] - 1.0 / (1.0 + math.exp(6.0))) < 1e-9                  # 0.0024726...
    assert t[500] == np.float32(0.5)
    assert abs(t[999] - 1.0 / (1.0 + math.exp(-(999 / 1000 * 2 - 1) * 6))) < 1e-6
    x = (np.arange(1000) / 1000.0 * 2 - 1) * 6
    assert np.allclose(t, 1.0 / (1.0 + np.exp(-x)), rtol=0, atol=2e-7) and np.all(np.diff(t) > 0)


def test_one_pair_update_by_hand(oracle):
    """dim 2, one node: r0 = (0.1, -0.2), r1 = (0.3, 0.4), code bit 1, alpha = 0.025.
    f = 0.1 * 0.3 - 0.2 * 0.4 = -0.05; index = int((f + 6) * (1000 / 12)) = int(495.83) = 495; expTable[495] = sigmoid(-0.06) =
    0.48500450; g = (1 - 1 - 0.48500450) * 0.025 = -0.01212511; neu = g * r1 = (-0.00363753, -0.00485004);
    r1 += g * r0 -> (0.29878749, 0.40242502); r0 += neu -> (0.09636247, -0.20485004)."""
    f32 = np.float32
    r0, r1 = np.array([0.1, -0.2], f32), np.array([[0.3, 0.4]], f32)
    n0, n1 = oracle.w2v_pair_update(r0, r1, [1], 0.025)
    sig = 1.0 / (1.0 + math.exp(0.06))
    g = (1.0 - 1.0 - sig) * 0.025
    assert abs(sig - 0.4850045) < 1e-7 and abs(g + 0.01212511) < 1e-8
    assert np.allclose(n1, [[0.3 + g * 0.1, 0.4 + g * -0.2]], rtol=0, atol=1e-7)
    assert np.allclose(n0, [0.1 + g * 0.3, -0.2 + g * 0.4], rtol=0, atol=1e-7)
    assert np.allclose(n0, [0.09636247, -0.20485004], atol=1e-7) and np.allclose(n1, [[0.29878749, 0.40242502]], atol=1e-7)


def test_pair_update_two_nodes_uses_the_old_syn1_for_neu(oracle):
    """Two nodes: neu1e accumulates g_d * syn1[d] with the row as it was BEFORE its own update (word2vec.c order: neu1e first, then
    syn1 += g * syn0), and syn0 is only touched after the last node — so the second node's f still sees the old syn0."""
    f32 = np.float32
    r0 = np.array([0.5, 0.25], f32); r1 = np.array([[1.0, -1.0], [0.5, 0.5]], f32)
    n0, n1 = oracle.w2v_pair_update(r0, r1, [0, 1], 0.1)
    t = oracle.w2v_exp_table()
    exp_r0 = r0.astype(np.float64).copy(); neu = np.zeros(2)
    exp_r1 = r1.astype(np.float64).copy()
    for d, bit in enumerate([0, 1]):
        f = float(np.dot(r0.astype(np.float64), exp_r1[d]))
        g = (1.0 - bit - float(t[int((f + 6.0) * (1000.0 / 12.0))])) * 0.1
        neu += g * exp_r1[d]
        exp_r1[d] += g * r0
    exp_r0 += neu
    assert np.allclose(n0, exp_r0, atol=1e-6) and np.allclose(n1, exp_r1, atol=1e-6)
    # f of node 0 = 0.25 -> index 520; f of node 1 = 0.375 -> index 531 (both from the ORIGINAL r0)
    assert int((0.25 + 6.0) * (1000.0 / 12.0)) == 520 and int((0.375 + 6.0) * (1000.0 / 12.0)) == 531


def test_out_of_range_dot_product_skips_the_node(oracle):
    r0 = np.array([4.0, 0.0], np.float32); r1 = np.array([[2.0, 0.0]], np.float32)        # f = 8 >= MAX_EXP: no update at all
    n0, n1 = oracle.w2v_pair_update(r0, r1, [1], 0.025)
    assert np.array_equal(n0, r0) and np.array_equal(n1, r1)
