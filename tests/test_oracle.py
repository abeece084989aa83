"""CPU tests of the oracle (oracle/goss_oracle.c) against every known answer the reference's
own tests hold for this path, plus the known-answer vectors recorded in SURVEY.md App. C, plus
writer -> reader round trips through the restated read side (SparseArray select/rank/access,
DenseSelect::select, VariableByteArray::operator[]) in the style of testSparseArray.cc /
testDenseArray.cc / testVariableByteArray.cc."""
import random
import struct

import pytest


def test_utils_known_answers(oracle):
    """testUtils.cc:23-57."""
    L = oracle.lib()
    assert [L.go_log2(x) for x in range(1, 10)] == [0, 1, 2, 2, 3, 3, 3, 3, 4]
    assert L.go_select1(0x5, 0) == 0 and L.go_select1(0x5, 1) == 2
    for i in range(64):
        assert L.go_select1(0xFFFFFFFFFFFFFFFF, i) == i
    for i in range(32):
        assert L.go_select1(0x5555555555555555, i) == 2 * i
        assert L.go_select1(0xAAAAAAAAAAAAAAAA, i) == 2 * i + 1
    for i in range(16):
        for j, w in enumerate((0x1111111111111111, 0x2222222222222222, 0x4444444444444444, 0x8888888888888888)):
            assert L.go_select1(w, i) == 4 * i + j


def test_vbyte_known_bytes(oracle):
    """testVByteCodec.cc:21-110."""
    assert oracle.vbyte_encode(0) == b"\x00"
    assert oracle.vbyte_encode(1) == b"\x01"
    assert oracle.vbyte_encode(128) == b"\x80\x80"
    for i in range(64):
        x = 1 << i
        assert oracle.vbyte_decode(oracle.vbyte_encode(x))[0] == x
    rng = random.Random(1)
    for _ in range(20000):
        x = rng.getrandbits(rng.randint(1, 64))
        b = oracle.vbyte_encode(x)
        assert oracle.vbyte_decode(b) == (x, len(b))
    for x in range(0, 1 << 14):
        assert oracle.vbyte_decode(oracle.vbyte_encode(x))[0] == x


def test_canonical_form_known_answers(oracle):
    """SURVEY.md App. C (captured from the reference's own code): value, FNV-1a hash,
    reverse complement and which strand normalize() keeps."""
    v = oracle.kmer_value("ACGTACGTACGTACGTACGTACGTA")
    assert v == 119212931312748 and oracle.fnv(v) == 3023600895719869485
    rc = oracle.revcomp(v, 25)
    assert rc == 874228162960155 and oracle.kmer_string(rc, 25) == "TACGTACGTACGTACGTACGTACGT"
    assert oracle.fnv(rc) == 2200065297453272164
    assert oracle.normalize(v, 25) == rc               # larger value, smaller hash
    t = oracle.kmer_value("T" * 25)
    assert t == 1125899906842623 and oracle.fnv(t) == 15861409801123372236
    assert oracle.fnv(0) == 9808874869469701221 and oracle.normalize(t, 25) == 0
    g = oracle.kmer_value("GATTACAGATTACAGATTACAGATT")
    assert g == 629233934386319 and oracle.fnv(g) == 12020582313314063284
    assert oracle.revcomp(g, 25) == 61232791124749 and oracle.fnv(61232791124749) == 14067249205108543165
    assert oracle.normalize(g, 25) == g                # forward: larger value, smaller hash
    c = oracle.kmer_value("CCCCCCCCCCCCCAAAAAAAAAAAA")
    assert c == 375299963355136 and oracle.fnv(c) == 6897489284872682045
    assert oracle.revcomp(c, 25) == 1125899884473002 and oracle.fnv(1125899884473002) == 17867811721608618404
    assert oracle.normalize(c, 25) == c
    assert oracle.fnv(27) == 4535195482310230718 and oracle.revcomp(27, 4) == 27     # ACGT palindrome
    w = oracle.kmer_value("ACGTTGCA" * 7)
    assert (w & (2**64 - 1), w >> 64) == (2009762000248511460, 30666534427620)
    assert oracle.fnv(w) == 7144136964027999792
    r = oracle.revcomp(w, 56)
    assert (r & (2**64 - 1), r >> 64) == (16436982073461040155, 250808442283035)
    assert oracle.kmer_string(r, 56) == "TGCAACGT" * 7


def test_sparse_d_table(oracle):
    """D(len, M)."""
    d = lambda ln, m: oracle.lib().go_sparse_d(oracle.key(4 ** ln), m)
    assert d(25, 126000000) == 23 and d(25, 10**9) == 20 and d(25, 3 * 10**9) == 18
    assert d(56, 2 * 10**8) == 84 and d(56, 4 * 10**9) == 80


def test_five_key_kmer_set_sizes(oracle):
    """five keys into KmerSet::Builder(25, "ks", fac, 5)."""
    files = oracle.write_kmer_set([3, 1000, 123456789, 2**40, 2**49 + 17], 25, 5)
    assert {k: len(v) for k, v in files.items()} == {
        "ks.header": 24, "ks.kmers.header": 64, "ks.kmers.high-bits": 8, "ks.kmers-d0": 4160,
        "ks.kmers-d1": 4144, "ks.kmers.low-bits.lwr": 20, "ks.kmers.low-bits.upr": 10}
    version, D, qD = struct.unpack("<3Q", files["ks.kmers.header"][:24])
    assert (version, D, qD) == (2012030501, 47, 48)
    assert struct.unpack("<3Q", files["ks.header"]) == (2011101701, 25, 5)


def test_reference_build_graph_cases(oracle):
    """testGossCmdBuildGraph.cc:114-179: polyA k=27 and the read with four Ns, k=15."""
    files, nwin = oracle.build_graph([(oracle.FASTA, "reads.fa", ">\nAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n")], 27)
    assert oracle.graph_header(files, "gr") == (27, 0)
    r = oracle.SparseReader(files, "gr-edges")
    assert r.count() == 2
    e = r.select(0)
    ep = oracle.revcomp(e, 28)
    assert r.rank(ep) == 1 and r.access(ep)
    assert oracle.vba_get(files, "gr-counts", 0) == oracle.vba_get(files, "gr-counts", 1)
    files, nwin = oracle.build_graph([(oracle.FASTA, "reads.fa",
        ">\nNACTTTTGATGCAATGTCAAATTCTCCNCGTCATTCGCAACTGAATACAAGNGAATTTGGAAGGAGAATNTGGTA\n")], 15)
    assert oracle.SparseReader(files, "gr-edges").count() == 42


def test_reverse_complement_adapter_count(oracle):
    """testReverseComplementAdapter.cc:26-53: two FASTA reads, rho = 15 -> 116 items."""
    keys, nreads, nwin = oracle.collect([(oracle.FASTA, "x.fa",
        ">1\nTTTT\n>2\nTTTTATGTACTATTATCTTATTTCTAAATATTAACTATAGTATCCCCTGGCGTTAATACAGCTCTAGAAATC\n")], 15, 1)
    assert len(keys) == 116 and nreads == 2
    for i in range(0, 116, 2):
        assert keys[i + 1] == oracle.revcomp(keys[i], 15)


def test_kmerize_restart_after_invalid(oracle):
    """GossReadBaseString.hh:52-103: windows restart after a non-ACGT character."""
    seq = "ACGTNACGTAC"
    assert oracle.kmerize(seq, 4) == [oracle.kmer_value("ACGT")] + [oracle.kmer_value(seq[i:i + 4]) for i in range(5, 8)]
    assert oracle.kmerize("acgtACGT", 8) == [oracle.kmer_value("ACGTACGT")]
    assert oracle.kmerize("ACG", 4) == [] and oracle.kmerize("", 4) == []
    rng = random.Random(5)
    for _ in range(200):
        s = "".join(rng.choice("ACGTNacgtn") for _ in range(rng.randint(0, 80)))
        k = rng.randint(1, 12)
        exp = [oracle.kmer_value(s[i:i + k]) for i in range(len(s) - k + 1) if all(c in "ACGTacgt" for c in s[i:i + k])]
        assert oracle.kmerize(s, k) == exp


def test_fastq_parser_behaviour(oracle):
    """FastqParser.hh:78-176 (the cases of testFastqParser.cc: '@' in quality, wrapped records,
    CRLF, empty sequence, truncation errors)."""
    fq = "@r1\nACGT\nAC\n+r1\nII@I\n+I\n@r2\r\nGGGG\r\n+\r\n@@@@\r\n@r3\n\n+\n"
    keys, nreads, nwin = oracle.collect([(oracle.FASTQ, "a.fq", fq)], 4, 0)
    assert nreads == 3
    assert nwin == 3 + 1
    with pytest.raises(oracle.OracleError, match="expected '@' at beginning of line 1"):
        oracle.collect([(oracle.FASTQ, "a.fq", "ACGT\n")], 4, 0)
    with pytest.raises(oracle.OracleError, match="expected sequence data or quality header at line 3"):
        oracle.collect([(oracle.FASTQ, "a.fq", "@r\nACGT\n")], 4, 0)
    with pytest.raises(oracle.OracleError, match="expected '\\+' at beginning of line 3"):
        oracle.collect([(oracle.FASTQ, "a.fq", "@r\nACGT\n@x\n")], 4, 0)
    with pytest.raises(oracle.OracleError, match="quality title does not match"):
        oracle.collect([(oracle.FASTQ, "a.fq", "@r\nACGT\n+q\nIIII\n")], 4, 0)
    with pytest.raises(oracle.OracleError, match="length mistmatch"):
        oracle.collect([(oracle.FASTQ, "a.fq", "@r\nACGT\n+\nII\n")], 4, 0)
    with pytest.raises(oracle.OracleError, match="No valid reads."):
        oracle.collect([(oracle.FASTQ, "a.fq", "")], 4, 0)
    # a last line without '\n' is kept; reads without any k-mer are fine
    keys, nreads, nwin = oracle.collect([(oracle.LINE, "l", "ACGTA\nAC")], 4, 0)
    assert (nreads, nwin) == (2, 2)
    keys, nreads, nwin = oracle.collect([(oracle.FASTA, "f", ">a\nAC\nGT\n>b\n")], 4, 0)
    assert (nreads, nwin) == (2, 1)
    with pytest.raises(oracle.OracleError, match="expected '>' at beginning of line 0"):
        oracle.collect([(oracle.FASTA, "f", "ACGT\n")], 4, 0)


def _check_sparse_round_trip(oracle, positions, N, M):
    files = oracle.write_sparse_array(positions, N, M, base="sa")
    r = oracle.SparseReader(files, "sa")
    assert r.count() == len(positions) and r.size() == N
    rng = random.Random(len(positions))
    idx = range(len(positions)) if len(positions) <= 3000 else sorted(rng.sample(range(len(positions)), 3000))
    for i in idx:
        assert r.select(i) == positions[i]
        assert r.rank(positions[i]) == i and r.access(positions[i])
        if positions[i] + 1 < N and (i + 1 == len(positions) or positions[i + 1] != positions[i] + 1):
            assert not r.access(positions[i] + 1)
            assert r.rank(positions[i] + 1) == i + 1
    return files


def test_sparse_array_round_trips(oracle):
    """testSparseArray.cc:27-307: empty set, small sets, 72- and 100-bit universes."""
    _check_sparse_round_trip(oracle, [], 1 << 20, 0)
    _check_sparse_round_trip(oracle, [0], 1 << 20, 1)
    _check_sparse_round_trip(oracle, [(1 << 20) - 1], 1 << 20, 1)
    rng = random.Random(9)
    for bits, n in [(20, 1000), (34, 20000), (50, 30000), (72, 4000), (100, 9000), (126, 500)]:
        pos = sorted({rng.getrandbits(bits) for _ in range(n)})
        _check_sparse_round_trip(oracle, pos, 1 << bits, len(pos))


def _ds_stats(data):
    h = struct.unpack("<16Q", data[:128])
    return {"blocks": h[8], "small": h[10], "intermediate": h[12], "large": h[14]}


def test_dense_select_block_kinds(oracle):
    """testDenseArray.cc:26-617 uses densities 1/10 .. 1/10000 to reach all three block
    kinds; here the kinds are asserted from the header statistics and every kind is read back
    through DenseSelect::select."""
    rng = random.Random(3)
    pos = sorted(rng.sample(range(1 << 26), 40000))
    f = _check_sparse_round_trip(oracle, pos, 1 << 26, len(pos))
    assert _ds_stats(f["sa-d1"])["small"] >= 4
    pos = sorted(rng.sample(range(1 << 40), 30000))
    f = _check_sparse_round_trip(oracle, pos, 1 << 40, 1 << 22)
    assert _ds_stats(f["sa-d1"])["intermediate"] >= 1
    pos = set()
    for c in range(6):
        base = rng.randrange(1 << 44)
        for _ in range(9000):
            pos.add(base + rng.randrange(1 << (8 + 3 * c)))
    pos = sorted(pos)
    f = _check_sparse_round_trip(oracle, pos, 1 << 46, 1 << 28)
    s1, s0 = _ds_stats(f["sa-d1"]), _ds_stats(f["sa-d0"])
    assert s1["large"] >= 2 and s0["small"] >= 1
    kinds = {k for s in (s0, s1) for k in ("small", "intermediate", "large") if s[k]}
    assert kinds == {"small", "intermediate", "large"}
    # every indexed zero / one position is found again
    r = oracle.SparseReader(f, "sa")
    D = struct.unpack("<Q", f["sa.header"][8:16])[0]
    ones = [(p >> D) + i for i, p in enumerate(pos)]
    for i in sorted(rng.sample(range(len(pos)), 2000)):
        assert r.d1_select(i) == ones[i]
    oneset = set(ones)
    zeros = [p for p in range(0, min(ones[-1] + 3, 200000)) if p not in oneset]
    for j in sorted(rng.sample(range(len(zeros)), min(2000, len(zeros)))):
        assert r.d0_select(j) == zeros[j]


def test_variable_byte_array_round_trip(oracle):
    """testVariableByteArray.cc:27-172: counts below 256, below 65536 and above."""
    rng = random.Random(4)
    counts = [rng.choice([1, 2, 3, 200, 255, 256, 300, 65535, 65536, 70000, 2**31 + 5, 2**32 - 1]) for _ in range(5000)]
    keys = sorted(rng.sample(range(1 << 40), len(counts)))
    files = oracle.write_graph(keys, counts, 19)
    for i in sorted(rng.sample(range(len(counts)), 400)):
        assert oracle.vba_get(files, "gr-counts", i) == counts[i]
    hist = {}
    for c in counts:
        hist[c] = hist.get(c, 0) + 1
    assert files["gr-counts-hist.txt"].decode() == "".join("%d\t%d\n" % (c, hist[c]) for c in sorted(hist))
    r = oracle.SparseReader(files, "gr-edges")
    assert [r.select(i) for i in range(0, len(keys), 97)] == keys[::97]


def test_integer_array_column_files(oracle):
    """IntegerArray::builder (IntegerArray.cc:259-357): the column files of every quantizedD
    the build commands can produce."""
    rng = random.Random(6)
    expect = {8: [""], 16: [""], 24: [".lwr", ".upr"], 32: [""], 40: [".lwr", ".upr"], 48: [".lwr", ".upr"],
              56: [".lwr.lwr", ".lwr.upr", ".upr"], 64: [""], 72: [".lwr", ".upr"], 80: [".lwr", ".upr"],
              88: [".lwr.lwr", ".lwr.upr", ".upr"], 96: [".lwr", ".upr"], 104: [".lwr.lwr", ".lwr.upr", ".upr"],
              112: [".lwr.lwr", ".lwr.upr", ".upr"], 120: [".lwr.lwr", ".lwr.upr", ".upr.lwr", ".upr.upr"]}
    for qd, cols in expect.items():
        # choose N, M so that D = qd exactly: D = ceil(log2(N / (1.44 (1+M))))
        M = 10
        N = 1 << min(126, qd + 4)
        pos = sorted({rng.getrandbits(min(126, qd + 4)) for _ in range(M)})
        files = oracle.write_sparse_array(pos, N, M, base="sa")
        D, q = struct.unpack("<2Q", files["sa.header"][8:24])
        if q != qd:
            continue
        got = sorted(n[len("sa.low-bits"):] for n in files if n.startswith("sa.low-bits"))
        assert got == sorted(cols), qd
        r = oracle.SparseReader(files, "sa")
        assert [r.select(i) for i in range(len(pos))] == pos


def test_build_commands_end_to_end(oracle):
    rng = random.Random(12)
    genome = "".join(rng.choice("ACGT") for _ in range(5000))
    reads = [genome[p:p + 100] for p in (rng.randrange(0, 4900) for _ in range(300))]
    fq = "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads))
    files, nwin = oracle.build_kmer_set([(oracle.FASTQ, "r.fq", fq)], 21)
    assert nwin == 300 * 80
    K, count = oracle.kmer_set_header(files, "ks")
    r = oracle.SparseReader(files, "ks.kmers")
    assert K == 21 and r.count() == count
    got = [r.select(i) for i in range(count)]
    exp = sorted({oracle.normalize(oracle.kmer_value(x[i:i + 21]), 21) for x in reads for i in range(80)})
    assert got == exp
    with pytest.raises(oracle.OracleError, match="unable to build a graph with k=64"):
        oracle.build_kmer_set([(oracle.FASTQ, "r.fq", fq)], 64)
    with pytest.raises(oracle.OracleError, match="unable to build a graph with k=63"):
        oracle.build_graph([(oracle.FASTQ, "r.fq", fq)], 63)


def test_merge_commands(oracle):
    """GossCmdMerge.tcc:151-326: union of the keys, counts summed, estimate M = sum of the input
    counts (so D differs from a direct build), groups of --max-merge first."""
    rng = random.Random(23)
    genome = "".join(rng.choice("ACGT") for _ in range(3000))
    parts = []
    for p in range(3):
        reads = [genome[s:s + 80] for s in (rng.randrange(0, 2920) for _ in range(150))]
        parts.append("\n".join(reads) + "\n")
    k = 21
    files = {}
    sets = []
    for i, txt in enumerate(parts):
        f, _ = oracle.build_kmer_set([(oracle.LINE, "r", txt)], k, out="ks%d" % i)
        files.update(f)
        r = oracle.SparseReader(f, "ks%d.kmers" % i)
        sets.append([r.select(j) for j in range(r.count())])
    merged = oracle.merge(files, ["ks0", "ks1", "ks2"], 0, "m")
    r = oracle.SparseReader(merged, "m.kmers")
    union = sorted(set().union(*sets))
    assert [r.select(j) for j in range(r.count())] == union
    assert oracle.kmer_set_header(merged, "m") == (k, len(union))
    D = struct.unpack("<Q", merged["m.kmers.header"][8:16])[0]
    tot = sum(len(s) for s in sets)
    assert D == oracle.lib().go_sparse_d(oracle.key(4 ** k), tot)
    # max-merge 2: ks0+ks1 first, then ks2 + temp: the estimate is |ks2| + |ks0 u ks1|
    merged2 = oracle.merge(files, ["ks0", "ks1", "ks2"], 0, "m", max_merge=2)
    r2 = oracle.SparseReader(merged2, "m.kmers")
    assert [r2.select(j) for j in range(r2.count())] == union
    D2 = struct.unpack("<Q", merged2["m.kmers.header"][8:16])[0]
    assert D2 == oracle.lib().go_sparse_d(oracle.key(4 ** k), len(sets[2]) + len(set(sets[0]) | set(sets[1])))
    # graphs: counts add up
    gfiles = {}
    expect = {}
    for i, txt in enumerate(parts):
        f, _ = oracle.build_graph([(oracle.LINE, "r", txt)], k, out="g%d" % i)
        gfiles.update(f)
        keys, _, _ = oracle.collect([(oracle.LINE, "r", txt)], k + 1, 1)
        for x in keys:
            expect[x] = expect.get(x, 0) + 1
    mg = oracle.merge(gfiles, ["g0", "g1", "g2"], 1, "mg")
    r = oracle.SparseReader(mg, "mg-edges")
    ek = sorted(expect)
    assert [r.select(j) for j in range(r.count())] == ek
    for j in sorted(rng.sample(range(len(ek)), 200)):
        assert oracle.vba_get(mg, "mg-counts", j) == expect[ek[j]]
    hist = {}
    for c in expect.values():
        hist[c] = hist.get(c, 0) + 1
    assert mg["mg-counts-hist.txt"].decode() == "".join("%d\t%d\n" % (c, hist[c]) for c in sorted(hist))
    with pytest.raises(oracle.OracleError, match="must have the same kmer-size"):
        f, _ = oracle.build_kmer_set([(oracle.LINE, "r", parts[0])], 19, out="other")
        files.update(f)
        oracle.merge(files, ["ks0", "other"], 0, "x")


def test_set_algebra_commands(oracle):
    """intersect-kmer-sets, subtract-kmer-set, merge-and-annotate-kmer-sets against Python sets:
    results are built with the exact count as the estimate (GossCmdIntersectKmerSets.cc:118-126,
    GossCmdSubtractKmerSet.cc:72-74, GossCmdMergeAndAnnotateKmerSets.cc:124)."""
    rng = random.Random(29)
    genome = "".join(rng.choice("ACGT") for _ in range(4000))
    k = 23
    files = {}
    sets = []
    for i in range(3):
        lo = 600 * i
        reads = [genome[s:s + 90] for s in (rng.randrange(lo, lo + 2500) for _ in range(200))]
        f, _ = oracle.build_kmer_set([(oracle.LINE, "r", "\n".join(reads) + "\n")], k, out="s%d" % i)
        files.update(f)
        r = oracle.SparseReader(f, "s%d.kmers" % i)
        sets.append([r.select(j) for j in range(r.count())])
    f, _ = oracle.build_kmer_set([(oracle.LINE, "r", "ACGT\n")], k, out="empty")      # no 23-mers: an empty set
    files.update(f)

    def keys_of(fs, base):
        r = oracle.SparseReader(fs, base + ".kmers")
        return [r.select(j) for j in range(r.count())]

    def check_exact(fs, base, keys):
        assert keys_of(fs, base) == keys
        assert oracle.kmer_set_header(fs, base) == (k, len(keys))
        D = struct.unpack("<Q", fs[base + ".kmers.header"][8:16])[0]
        assert D == oracle.lib().go_sparse_d(oracle.key(4 ** k), len(keys))

    inter = sorted(set(sets[0]) & set(sets[1]) & set(sets[2]))
    assert inter
    got = oracle.intersect_kmer_sets(files, ["s0", "s1", "s2"], "x")
    check_exact(got, "x", inter)
    # empty inputs are skipped by the reference's loop (invalid iterators are not kept)
    got = oracle.intersect_kmer_sets(files, ["s0", "empty", "s1"], "x")
    check_exact(got, "x", sorted(set(sets[0]) & set(sets[1])))
    # one input: a copy
    got = oracle.intersect_kmer_sets(files, ["s2"], "x")
    check_exact(got, "x", sets[2])
    assert oracle.intersect_kmer_sets(files, [], "x") == {}

    got = oracle.subtract_kmer_set(files, "s0", "s1", "d")
    check_exact(got, "d", sorted(set(sets[0]) - set(sets[1])))
    got = oracle.subtract_kmer_set(files, "s0", "empty", "d")
    check_exact(got, "d", sets[0])
    got = oracle.subtract_kmer_set(files, "s0", "s0", "d")
    check_exact(got, "d", [])

    got, stats = oracle.merge_and_annotate(files, "s0", "s1", "u")
    union = sorted(set(sets[0]) | set(sets[1]))
    check_exact(got, "u", union)
    assert stats == (len(sets[0]), len(sets[1]), len(set(sets[0]) & set(sets[1])))
    nw = (len(union) + 63) // 64
    for name, members in (("u.lhs-bits", set(sets[0])), ("u.rhs-bits", set(sets[1]))):
        words = struct.unpack("<%dQ" % nw, got[name])
        for i, x in enumerate(union):
            assert ((words[i // 64] >> (i % 64)) & 1) == (x in members)
    with pytest.raises(oracle.OracleError, match="nonsense"):
        oracle.merge_and_annotate(files, "s0", "empty", "u")


def test_threaded_build_gives_the_same_files(oracle):
    """go_build_kmer_set_mt (bench.py's CPU baseline at T threads) is the single-thread restatement
    with the counting sharded: identical files for any number of threads"""
    import gossamer_amd as g
    reads = g.synth_reads_host(3000, 150, 40000, seed=5)
    want, nwin = oracle.build_kmer_set([(oracle.LINE, "r", reads)], 25)
    for T in (1, 2, 3, 8):
        got, n = oracle.build_kmer_set_mt(reads, 25, T)
        assert n == nwin and got == want, T
    with pytest.raises(oracle.OracleError):
        oracle.build_kmer_set_mt(b"NNNN\n", 25, 4)
