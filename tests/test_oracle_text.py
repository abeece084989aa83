"""Oracle restatements of dump-kmer-set / dump-graph / restore-graph against first principles."""
import random

import pytest


def test_dump_and_restore(oracle):
    rng = random.Random(31)
    genome = "".join(rng.choice("ACGT") for _ in range(2000))
    reads = [genome[s:s + 70] for s in (rng.randrange(0, 1930) for _ in range(300))]
    txt = "\n".join(reads) + "\n"
    k = 21
    ks, _ = oracle.build_kmer_set([(oracle.LINE, "r", txt)], k, out="ks")
    r = oracle.SparseReader(ks, "ks.kmers")
    keys = [r.select(j) for j in range(r.count())]

    def seq(x, n):
        return "".join("ACGT"[(x >> (2 * (n - 1 - i))) & 3] for i in range(n))

    exp = "#2011101701\n%d\t%d\n" % (k, len(keys)) + "".join(seq(x, k) + "\n" for x in keys)
    assert oracle.dump(ks, "ks", 0).decode() == exp

    gr, _ = oracle.build_graph([(oracle.LINE, "r", txt)], k, out="gr")
    edges, _, _ = oracle.collect([(oracle.LINE, "r", txt)], k + 1, 1)
    cnt = {}
    for x in edges:
        cnt[x] = cnt.get(x, 0) + 1
    exp = "#2011101014\n%d\t%d\t0\n" % (k, len(cnt)) + "".join("%s\t%d\n" % (seq(x, k + 1), cnt[x]) for x in sorted(cnt))
    text = oracle.dump(gr, "gr", 1)
    assert text.decode() == exp
    # restore(dump(g)) == g: the text's header carries the edge count, which is the estimate the
    # direct build used
    back = oracle.restore_graph(text, "gr")
    assert sorted(back) == sorted(gr)
    for name in gr:
        assert back[name] == gr[name], name
    # a last pair without a newline is dropped (the stream is no longer good after reading it)
    cut = oracle.restore_graph(text[:-1], "gr")
    rr = oracle.SparseReader(cut, "gr-edges")
    assert rr.count() == len(cnt) - 1
    # the asymmetric flag travels through the text header into the graph header
    flagged = text.replace(b"\t0\n", b"\t1\n", 1)
    fl = oracle.restore_graph(flagged, "gr")
    assert oracle.graph_header(fl, "gr") == (k, 1)
    with pytest.raises(oracle.OracleError, match="has wrong length"):
        oracle.restore_graph(b"#x\n21\t1\t0\nACGT\t1\n", "gr")
    with pytest.raises(oracle.OracleError, match="invalid sequence"):
        oracle.restore_graph(b"#x\n3\t1\t0\nACNT\t1\n", "gr")
    with pytest.raises(oracle.OracleError, match="unexpected end of file"):
        oracle.restore_graph(b"#x\n21\t1", "gr")


@pytest.mark.parametrize("k", [21, 31, 40])
def test_graph_to_kmer_set_against_first_principles(oracle, k):
    """A graph built from reads holds every (k+1)-mer and its reverse complement, so its normal
    edges are exactly the canonical (k+1)-mers of the reads: graph-to-kmer-set of the graph must
    hold the same k-mers as build-kmer-set with k+1 (the objects differ only in the SparseArray
    size estimate: the graph's edge count against the exact count)."""
    rng = random.Random(77)
    genome = "".join(rng.choice("ACGT") for _ in range(3000))
    reads = [genome[s:s + 90] for s in (rng.randrange(0, 2910) for _ in range(300))]
    reads.append("ACGT" * 20)                    # palindromic (k+1)-mers: their own reverse complement
    txt = "\n".join(reads) + "\n"
    gr, _ = oracle.build_graph([(oracle.LINE, "r", txt)], k, out="gr")
    ks = oracle.graph_to_kmer_set(gr, "gr", "ks")
    direct, _ = oracle.build_kmer_set([(oracle.LINE, "r", txt)], k + 1, out="ks")
    assert oracle.dump(ks, "ks", 0) == oracle.dump(direct, "ks", 0)
    assert ks["ks.header"] == direct["ks.header"]
    # the estimate is the number of edges, about twice the number of k-mers kept
    import struct
    n_edges = struct.unpack("<8Q", gr["gr-edges.header"])[7]
    n_kept = struct.unpack("<QQQ", ks["ks.header"])[2]
    assert n_kept <= n_edges <= 2 * n_kept
