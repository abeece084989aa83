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
