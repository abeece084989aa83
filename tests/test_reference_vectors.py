"""The reference's own seeded unit tests for the structures on the k-mer counting path, replayed
against the oracle (CPU; the GPU suite replays them against the product in
test_gpu_reference_vectors.py).

The inputs are regenerated with the generator the reference's tests use (std::mt19937 +
libstdc++ distributions: tests/golden/gen_reference_inputs.cpp, own code) and pinned by the
digests in tests/golden/reference_kat.json; what is asserted is what the reference's test
asserts (file:line in reference_kat.json "seeded_test_inputs"), through the oracle's writer and
its restatement of the reference's readers.  No reference test pins file BYTES (SURVEY.md
section 0, fact 7): these replays pin behaviour -- every access / rank / select / iterator
answer the reference's tests check."""
import json
import os

import pytest

import refvec

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_kat.json")) as f:
    KAT = json.load(f)


def test_streams_match_the_committed_digests():
    """the regenerated std::mt19937 streams are the pinned ones (toolchain drift would show here)"""
    c = refvec.cases()
    pinned = KAT["seeded_test_inputs"]
    assert sorted(c) == sorted(pinned)
    for name in c:
        assert c[name]["sha256"] == pinned[name]["sha256"], name
        assert "test" in pinned[name]["reference"] and ".cc:" in pinned[name]["reference"]
    # independent restatement of the first stream: mt19937(17) through numpy, two 32-bit draws per
    # uniform_real_distribution<double> value (generate_canonical<double, 53>)
    import numpy as np
    raw = [int(x) for x in np.random.RandomState(17).randint(0, 2 ** 32, size=60, dtype=np.uint64)]
    ones = [i for i in range(30) if (raw[2 * i] + raw[2 * i + 1] * 2 ** 32) / 2.0 ** 64 < 0.1]
    assert ones == c["sparse_test1"]["ones"]
    assert c["sparse_test3"]["positions"][0] == (raw[0] << 32) | raw[1]


# ---- testSparseArray.cc -----------------------------------------------------------------

def test_sparse_termination(oracle):
    """testSparseArray.cc:27-37: Builder("x", fac, D = 8), nothing pushed, end(257): access(256) is false"""
    files = oracle.write_sparse_array([], 257, 0, base="x", N_end=257)      # N = 257, M = 0 gives D = 8
    r = oracle.SparseReader(files, "x")
    assert r.count() == 0 and r.access(256) is False
    assert oracle.replay_sparse(files, "x", [], universe=257) == 0


@pytest.mark.parametrize("name,N,M", [("sparse_test1", 30, 3), ("sparse_test2", 1000, 10)])
def test_sparse_small_universe(oracle, name, N, M):
    """testSparseArray.cc test1 / test2: every position of the universe -- access, rank,
    accessAndRank, select, iterator; rank beyond the universe stays at the count"""
    c = refvec.cases()[name]
    assert c["nbits"] == N
    files = oracle.write_sparse_array(c["ones"], N, M, base="x", N_end=N)
    assert oracle.replay_sparse(files, "x", c["ones"], universe=N) == 0
    r = oracle.SparseReader(files, "x")
    # rank(pos1, pos2) == (rank(pos1), rank(pos2)): testSparseArray.cc:100-112
    for i in range(0, N - 10):
        for j in range(1, 10):
            assert r.rank(i) <= r.rank(i + j) <= r.rank(i) + j
    assert r.rank(N) == len(c["ones"]) == r.rank(8 * N)


@pytest.mark.parametrize("name,bits", [("sparse_test3", 72), ("sparse_test4", 100)])
def test_sparse_wide_universe(oracle, name, bits):
    """testSparseArray.cc test3 / test4 / test5: 120 positions in a 2^72 / 2^100 universe"""
    c = refvec.cases()[name]
    pos = c["positions"]
    assert len(pos) == 120 and all(pos[i] < pos[i + 1] for i in range(119)) and pos[-1] < (1 << bits)
    files = oracle.write_sparse_array(pos, 1 << bits, 120, base="x", N_end=1 << bits)
    assert oracle.replay_sparse(files, "x", pos) == 0
    r = oracle.SparseReader(files, "x")
    assert [r.select(i) for i in range(120)] == pos           # Iterator / LazyIterator order


# ---- testDenseArray.cc ------------------------------------------------------------------

DENSE = [("dense_test1", False), ("dense_test2", False), ("dense_test3", False), ("dense_test4", True),
         ("dense_test5", True), ("dense_test6", False), ("dense_one_in_10", False), ("dense_one_in_100", False),
         ("dense_one_in_1000", False), ("dense_one_in_10000", False), ("dense_bug_over_256", False)]


def dense_case(name):
    c = refvec.cases()[name]
    ones = list(range(516)) if name == "dense_bug_over_256" else c["ones"]
    return ones, c["nbits"]


@pytest.mark.parametrize("name,invert", DENSE)
def test_dense_select_standalone(oracle, name, invert):
    """WordyBitVector + DenseSelect built exactly as the reference's test does; v.get(i) for every i,
    a.select(j) for every one (zero when the sense is inverted)"""
    ones, nbits = dense_case(name)
    files = oracle.write_bits_and_select(ones, nbits, invert)
    assert oracle.replay_dense_select(files, ones, nbits, invert) == 0
    # the pair form of the later cases (select(i, i+j), i += 113, j < 197) is two single selects
    targets = [p for p in range(nbits) if p not in set(ones)] if invert and nbits <= 1000 else ones
    if not invert:
        for i in range(0, max(0, len(targets) - 197), 113 * 7):
            for j in (1, 63, 64, 196):
                assert oracle.dense_select(files, i + j, invert) == targets[i + j]


@pytest.mark.parametrize("name,invert", DENSE)
def test_dense_select_inside_a_sparse_array(oracle, name, invert):
    """The same bit vector as the high-bits vector of a SparseArray with D = 24 (how the product
    emits DenseSelect): -d1 and -d0 are DenseSelect structures of both senses over it"""
    ones, nbits = dense_case(name)
    pos, N, M = embed_as_high_bits(ones, nbits)
    files = oracle.write_sparse_array(pos, N, M, base="x", N_end=N)
    assert oracle.SparseReader(files, "x").count() == len(ones)
    assert oracle.replay_sparse_highbits(files, "x", ones, nbits) == 0


EMBED_D = 24


def embed_as_high_bits(ones, nbits):
    """element i = (ones[i] - i) << 24 | i of a universe of (nbits - n) << 24 positions: with D = 24
    its high-bits vector has its ones exactly at `ones` (the low bits only keep equal high parts
    apart); M is an estimate that makes SparseArray::Builder choose D = 24 (SparseArray.cc:47-72)"""
    import oracle_lib as o
    n = len(ones)
    assert n < (1 << EMBED_D)
    pos = [((p - i) << EMBED_D) | i for i, p in enumerate(ones)]
    N = max(1, nbits - n) << EMBED_D
    M = int(max(1, nbits - n) / 1.4426950408889634)
    for M in range(max(0, M - 2), M + 4):
        if o.lib().go_sparse_d(o.key(N), M) == EMBED_D:
            return pos, N, M
    raise AssertionError("no estimate gives D = %d" % EMBED_D)


# ---- testWordyBitVector.cc --------------------------------------------------------------

def test_wordy_bit_vector_known_answers(oracle):
    c = refvec.cases()
    files = oracle.write_bits_sparse(c["wordy_test2"]["ones"])
    v = oracle.BitsReader(files)
    a = KAT["wordy_test2_answers"]
    assert all(v.get(p) for p in a["get_true"]) and not any(v.get(p) for p in a["get_false"])
    for frm, cnt, want in a["select1"]:
        assert v.select1(frm, cnt) == want
    for frm, cnt, want in a["select0"]:
        assert v.select0(frm, cnt) == want
    v3 = oracle.BitsReader(oracle.write_bits_sparse(c["wordy_test3"]["ones"]))
    for frm, cnt, want in KAT["wordy_test3_answers"]["select1"]:
        assert v3.select1(frm, cnt) == want
    # testWordyBitVector.cc:26-33 (test1): an empty file has no words
    assert oracle.BitsReader({"x": b""}).words() == 0


def test_wordy_bit_vector_popcount_and_iterator(oracle):
    """testWordyBitVector.cc test4 (:128-165: popcountRange(i, i+j), i += 123, j < 201) and test5
    (:168-197: Iterator1 visits the ones in order)"""
    c = refvec.cases()["dense_test2"]
    ones, nbits = c["ones"], c["nbits"]
    files = oracle.write_bits_and_select(ones, nbits, False)
    v = oracle.BitsReader(files, "v")
    import bisect
    for i in list(range(0, nbits - 201, 123 * 41)) + [max(0, p - 100) for p in ones]:
        for j in (0, 1, 100, 200):
            assert v.popcount_range(i, i + j) == bisect.bisect_left(ones, i + j) - bisect.bisect_left(ones, i)
    assert [v.select1(0, r) for r in range(len(ones))] == ones


# ---- testVariableByteArray.cc -----------------------------------------------------------

@pytest.mark.parametrize("name,num_items", [("vba_test1", 100), ("vba_test2", 10000), ("vba_test3", 1000), ("vba_test4", 100000)])
def test_variable_byte_array(oracle, name, num_items):
    values = refvec.cases()[name]["values"]
    files = oracle.write_vba(values, num_items)
    assert oracle.replay_vba(files, "x", values) == 0
    for i in range(0, len(values), max(1, len(values) // 50)):
        assert oracle.vba_get(files, "x", i) == values[i]


# ---- testGraph.cc -----------------------------------------------------------------------

def test_graph_five_edges_and_out_degrees(oracle):
    """testGraph.cc:79-124: K = 15, the 16-mers of four sequences that differ in the last base;
    count() == 5; out-degree (edges sharing the 15-mer prefix of an edge) histogram 0/1/0/0/4"""
    t = KAT["graph_test1"]
    K = t["K"]
    counts = {}
    for s in t["sequences"]:
        for j in range(len(s) - (K + 1) + 1):
            e = oracle.kmer_value(s[j:j + K + 1])
            counts[e] = counts.get(e, 0) + 1
    edges = sorted(counts)
    files = oracle.write_graph(edges, [counts[e] for e in edges], K, out="x")
    r = oracle.SparseReader(files, "x-edges")
    assert r.count() == t["count"]
    hist = [0] * 5
    for e in edges:
        node = e >> 2                                     # Graph::from: the edge without its last base
        hist[r.rank((node << 2) + 4) - r.rank(node << 2)] += 1      # Graph::outDegree: edges node<<2 .. node<<2 | 3
    assert hist == t["outdegree_hist"]
    for i, e in enumerate(edges):
        assert oracle.vba_get(files, "x-counts", i) == counts[e]


def test_biginteger_decimal_known_answers(oracle):
    """testBigInteger.cc:58-110: the two-word position type is a plain 128-bit unsigned integer --
    the oracle's (lo, hi) key round-trips the values whose decimal form the reference's test spells out"""
    t = KAT["biginteger_decimal"]
    v = 1
    for want in t["shift16"]:
        assert str(v & ((1 << 128) - 1)) == want
        k = oracle.key(v & ((1 << 128) - 1))
        assert (k.hi << 64) | k.lo == v & ((1 << 128) - 1)
        v <<= 16
    assert str(1 << 124) == t["pow124"] and str(1 << 125) == t["pow125"]
    assert str((1 << 124) + (1 << 125)) == t["sum"] and str(1 << 127) == t["pow127"]
