"""Config C5 at reduced scale on one GPU: two k-mer sets built from overlapping read sets, then
intersect / subtract / merge through the goss executable; size-independent properties of set
algebra on ~10^8 k-mers (the oracle-compared cases live in test_gpu_parity.py)."""
import os
import struct
import subprocess

import pytest

import gossamer_amd as g

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOSS = os.path.join(ROOT, "gossamer_amd", "goss")


def _count(path):
    with open(path + ".header", "rb") as f:
        version, k, n = struct.unpack("<QQQ", f.read(24))
    assert version == 2011101701
    return n


def _files(tmp_path, base):
    return {n[len(base):]: (tmp_path / n).read_bytes() for n in sorted(os.listdir(tmp_path)) if n.startswith(base + ".")}


def test_c5_set_algebra_properties(tmp_path):
    n, L, G = 300_000, 150, 20_000_000          # ~2x coverage: many k-mers of a set are not in the other
    for name, first in (("a", 0), ("b", n // 2)):
        (tmp_path / (name + ".txt")).write_bytes(g.synth_reads_host(n, L, G, seed=31, first_read=first))

    def run(args):
        p = subprocess.run([GOSS] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()

    P = lambda s: str(tmp_path / s)
    for name in ("a", "b"):
        run(["build-kmer-set", "-k", "25", "--line-in", P(name + ".txt"), "-O", P(name)])
    run(["intersect-kmer-sets", "-G", P("a"), "-G", P("b"), "-O", P("ab")])
    run(["intersect-kmer-sets", "-G", P("b"), "-G", P("a"), "-O", P("ba")])
    run(["subtract-kmer-set", "-G", P("a"), "-G", P("b"), "-O", P("a_b")])
    run(["subtract-kmer-set", "-G", P("b"), "-G", P("a"), "-O", P("b_a")])
    run(["subtract-kmer-set", "-G", P("a"), "-G", P("ab"), "-O", P("a_ab")])
    run(["merge-kmer-sets", "-G", P("a"), "-G", P("b"), "-O", P("u")])
    run(["merge-kmer-sets", "-G", P("a_b"), "-G", P("b_a"), "-G", P("ab"), "-O", P("u3")])
    na, nb, nab = _count(P("a")), _count(P("b")), _count(P("ab"))
    assert na > 10_000_000 and nb > 10_000_000 and 0 < nab < min(na, nb)
    # |A n B| + |A \ B| = |A|, inclusion-exclusion for the union
    assert nab + _count(P("a_b")) == na
    assert nab + _count(P("b_a")) == nb
    assert _count(P("u")) == na + nb - nab
    # commutativity and A \ B == A \ (A n B): byte-identical objects
    assert _files(tmp_path, "ab") == _files(tmp_path, "ba")
    assert _files(tmp_path, "a_b") == _files(tmp_path, "a_ab")
    # the three disjoint parts re-assemble the union: same k-mers (the merge estimate differs:
    # |A|+|B| against the exact size, so only the decoded content is compared, through dump)
    run(["dump-kmer-set", "-G", P("u"), "-o", P("u.txt")])
    run(["dump-kmer-set", "-G", P("u3"), "-o", P("u3.txt")])
    assert (tmp_path / "u.txt").read_bytes() == (tmp_path / "u3.txt").read_bytes()


def test_batchwise_build_then_merge_equals_one_build(tmp_path):
    """The reference's scale-out workflow (docs/goss.md: build graphs of batches, merge them): the
    merged graph must hold the same edges and multiplicities as one build over all reads (compared
    through dump-graph: the merged object's SparseArray is sized with the sum of the inputs' counts,
    so its files differ while its content may not), and it must pass lint-graph."""
    n, L, G = 400_000, 150, 3_000_000
    whole = g.synth_reads_host(2 * n, L, G, seed=33)
    half = len(whole) // 2
    (tmp_path / "all.txt").write_bytes(whole)
    (tmp_path / "a.txt").write_bytes(whole[:half])
    (tmp_path / "b.txt").write_bytes(whole[half:])

    def run(args):
        p = subprocess.run([GOSS] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert p.returncode == 0, p.stderr.decode()
        return p

    P = lambda s: str(tmp_path / s)
    for k in (27, 55):
        for name in ("all", "a", "b"):
            run(["build-graph", "-k", str(k), "--line-in", P(name + ".txt"), "-O", P("%s%d" % (name, k))])
        run(["merge-graphs", "-G", P("a%d" % k), "-G", P("b%d" % k), "-O", P("m%d" % k)])
        run(["dump-graph", "-G", P("all%d" % k), "-o", P("all%d.txt" % k)])
        run(["dump-graph", "-G", P("m%d" % k), "-o", P("m%d.txt" % k)])
        assert (tmp_path / ("all%d.txt" % k)).read_bytes() == (tmp_path / ("m%d.txt" % k)).read_bytes()
        assert (tmp_path / ("all%d-counts-hist.txt" % k)).read_bytes() == (tmp_path / ("m%d-counts-hist.txt" % k)).read_bytes()
        p = run(["lint-graph", "-G", P("m%d" % k)])
        assert b"warning" not in p.stderr
