"""The reference's seeded unit tests (testSparseArray.cc, testDenseArray.cc, testVariableByteArray.cc,
testGraph.cc) replayed against the PRODUCT: the HIP writers build the structures from the
regenerated inputs (tests/golden/gen_reference_inputs.cpp, digests in reference_kat.json), and

  (i)   the product's files are byte-identical to the oracle's,
  (ii)  the assertions of the reference's test hold when the oracle's restatement of the
        reference's readers walks the PRODUCT's files (GPU writer -> CPU reader), and
  (iii) the device read side (goss_gpu_check_index: SparseArray select / rank / access through the
        -d1 / -d0 DenseSelect images) agrees with the element list on the ORACLE's files and on
        its own (CPU writer -> GPU reader, GPU writer -> GPU reader).

A DenseSelect only exists inside a SparseArray in the product, so the bit vectors of
testDenseArray.cc are embedded as the high-bits vector of a SparseArray (test_reference_vectors.py,
embed_as_high_bits): -d1 / -d0 are then DenseSelect structures of both senses over exactly that vector.
"""
import json
import os

import pytest

import gossamer_amd as g
import refvec
from test_reference_vectors import DENSE, dense_case, embed_as_high_bits

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_kat.json")) as f:
    KAT = json.load(f)


def _key_tensor(positions, words):
    import torch
    flat = []
    for p in positions:
        flat.append(p & 0xFFFFFFFFFFFFFFFF)
        if words == 2:
            flat.append(p >> 64)
    return torch.tensor([x - (1 << 64) if x >= (1 << 63) else x for x in flat] or [0], dtype=torch.int64, device="cuda")


def _product_sparse(oracle, positions, N, M, k):
    """(product files, oracle files, index reports of the device reader over both)"""
    import torch
    words = 1 if 2 * k <= 62 else 2
    t = _key_tensor(positions, words)
    ones = torch.ones(max(1, len(positions)), dtype=torch.int32, device="cuda")
    reports = []
    with g.Context(k, g.MODE_KMER_SET, hbm_budget=768 << 20) as ctx:
        if positions:
            ctx.push_run(t.data_ptr(), ones.data_ptr(), len(positions))
        ctx.finish()
        mine = ctx.emit_sparse_array(t.data_ptr(), words, len(positions), N, M)
        theirs = {n[1:]: b for n, b in oracle.write_sparse_array(positions, N, M, base="x", N_end=N).items()}
        for files in (mine, theirs):
            reports.append(ctx.check_index(files))
    return mine, theirs, reports


def _clean(rep):
    return rep["select"] == 0 and rep["rank"] == 0 and rep["access"] == 0 and rep["failures"] == 0


def _with_base(files, base="x"):
    return {base + n: b for n, b in files.items()}


@pytest.mark.parametrize("name,N,M", [("sparse_test1", 30, 3), ("sparse_test2", 1000, 10)])
def test_sparse_small_universe(oracle, name, N, M):
    c = refvec.cases()[name]
    mine, theirs, reports = _product_sparse(oracle, c["ones"], N, M, 25)
    assert mine == theirs
    assert oracle.replay_sparse(_with_base(mine), "x", c["ones"], universe=N) == 0
    assert all(_clean(r) for r in reports), reports


def test_sparse_termination(oracle):
    mine, theirs, reports = _product_sparse(oracle, [], 257, 0, 25)
    assert mine == theirs
    assert oracle.SparseReader(_with_base(mine), "x").access(256) is False


@pytest.mark.parametrize("name,bits", [("sparse_test3", 72), ("sparse_test4", 100)])
def test_sparse_wide_universe(oracle, name, bits):
    pos = refvec.cases()[name]["positions"]
    mine, theirs, reports = _product_sparse(oracle, pos, 1 << bits, 120, 63)
    assert mine == theirs
    assert oracle.replay_sparse(_with_base(mine), "x", pos) == 0
    assert all(_clean(r) for r in reports), reports


@pytest.mark.parametrize("name,invert", DENSE)
def test_dense_select_cases(oracle, name, invert):
    """every bit vector of testDenseArray.cc as the high-bits vector of a product-built SparseArray"""
    ones, nbits = dense_case(name)
    pos, N, M = embed_as_high_bits(ones, nbits)
    mine, theirs, reports = _product_sparse(oracle, pos, N, M, 31)
    assert mine == theirs
    assert oracle.replay_sparse_highbits(_with_base(mine), "x", ones, nbits) == 0
    assert all(_clean(r) for r in reports), reports


def _product_graph(keys, counts, K):
    import torch
    t = _key_tensor(keys, 1)
    c = torch.tensor(counts or [0], dtype=torch.int64, device="cuda").to(torch.int32)
    with g.Context(K, g.MODE_GRAPH, hbm_budget=768 << 20) as ctx:
        if keys:
            ctx.push_run(t.data_ptr(), c.data_ptr(), len(keys))
        ctx.finish()
        return ctx.emit()


@pytest.mark.parametrize("name", ["vba_test1", "vba_test2", "vba_test3", "vba_test4"])
def test_variable_byte_array(oracle, name):
    """the values of testVariableByteArray.cc as the multiplicities of a product-built graph: its
    -counts VariableByteArray must read back every value (and equal the oracle's bytes)"""
    values = refvec.cases()[name]["values"]
    K = 15
    keys = [7 * i + 3 for i in range(len(values))]          # any strictly increasing 16-mers
    mine = _product_graph(keys, values, K)
    theirs = {n[1:]: b for n, b in oracle.write_graph(keys, values, K, out="x").items()}
    assert sorted(mine) == sorted(theirs)
    for n in theirs:
        if n != "-counts-hist.txt":
            assert mine[n] == theirs[n], n
    assert oracle.replay_vba(_with_base(mine), "x-counts", values) == 0
    # the histogram lists every multiplicity with its frequency, ascending (Graph.cc:131-150)
    want = {}
    for v in values:
        want[v] = want.get(v, 0) + 1
    assert mine["-counts-hist.txt"] == "".join("%d\t%d\n" % (c, want[c]) for c in sorted(want)).encode()
    assert mine["-counts-hist.txt"] == theirs["-counts-hist.txt"]


@pytest.mark.parametrize("name", ["vba_test1", "vba_test4"])
def test_variable_byte_array_read_on_the_device(oracle, name):
    """the read side (VariableByteArray::operator[] / GeneralIterator, VariableByteArray.hh:120-247) on the device:
    the graph files of the previous test -- written by the ORACLE -- pushed back as a run
    (goss_gpu_push_run_graph, what merge-graphs does with its inputs) give every edge and every value"""
    values = refvec.cases()[name]["values"]
    K = 15
    keys = [7 * i + 3 for i in range(len(values))]
    theirs = {n[1:]: b for n, b in oracle.write_graph(keys, values, K, out="x").items()}
    with g.Context(K, g.MODE_GRAPH, hbm_budget=768 << 20) as ctx:
        ctx.push_run_graph(theirs, 2 * (K + 1))
        ctx.finish()
        got_keys, got_counts = ctx.result()
    assert got_keys == keys
    assert [int(x) for x in got_counts] == values


def test_graph_five_edges_and_out_degrees(oracle):
    t = KAT["graph_test1"]
    K = t["K"]
    counts = {}
    for s in t["sequences"]:
        for j in range(len(s) - (K + 1) + 1):
            e = oracle.kmer_value(s[j:j + K + 1])
            counts[e] = counts.get(e, 0) + 1
    edges = sorted(counts)
    mine = _product_graph(edges, [counts[e] for e in edges], K)
    theirs = {n[1:]: b for n, b in oracle.write_graph(edges, [counts[e] for e in edges], K, out="x").items()}
    assert mine == theirs
    r = oracle.SparseReader(_with_base(mine), "x-edges")
    assert r.count() == t["count"]
    hist = [0] * 5
    for e in edges:
        node = e >> 2
        hist[r.rank((node << 2) + 4) - r.rank(node << 2)] += 1
    assert hist == t["outdegree_hist"]


def test_graph_builder_refuses_k(oracle):
    """testGraph.cc:142-156: "unable to build a graph with k=12345678" -- the product's status for it"""
    with pytest.raises(g.binding.GossGpuError) as e:
        g.Context(KAT["graph_builder_k_error"]["K"], g.MODE_GRAPH, hbm_budget=64 << 20)
    assert e.value.status == -6          # GOSS_ERR_K_RANGE
