"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Integer / byte work: the bar is bit-exact."""
import random
import struct

import pytest

import gossamer_amd as g

pytestmark = pytest.mark.gpu

MB = 1 << 20


def make_reads(rng, nreads, read_len, genome_len, n_rate=0.02, lower=False):
    genome = "".join(rng.choice("ACGT") for _ in range(genome_len))
    reads = []
    for _ in range(nreads):
        L = read_len if isinstance(read_len, int) else rng.randint(*read_len)
        p = rng.randint(0, max(0, genome_len - L))
        r = list(genome[p:p + L])
        for i in range(len(r)):
            if rng.random() < n_rate:
                r[i] = rng.choice("NnRY.-")
            elif lower and rng.random() < 0.3:
                r[i] = r[i].lower()
        reads.append("".join(r))
    return reads


def oracle_counts(oracle, reads, length, mode):
    ks, cs, nreads, nwin = oracle.count([(oracle.LINE, "r", "\n".join(reads) + "\n")], length, mode)
    return ks, cs, nwin


def gpu_counts(reads, k, mode, budget=256 * MB, pushes=1, path=0):
    with g.Context(k, mode, hbm_budget=budget) as ctx:
        ctx.set_path(path)
        per = (len(reads) + pushes - 1) // pushes
        for i in range(0, len(reads), per):
            ctx.push_host("\n".join(reads[i:i + per]) + "\n")
        c = ctx.finish()
        ks, cs = ctx.result()
        return ks, [int(x) for x in cs], c


@pytest.mark.parametrize("k,mode", [(25, 0), (15, 1), (31, 0), (32, 0), (30, 1), (31, 1), (55, 1), (63, 0), (62, 1), (1, 0), (4, 1)])
def test_keys_and_counts_match_oracle(oracle, k, mode):
    rng = random.Random(1000 + k * 2 + mode)
    reads = make_reads(rng, 300, (max(5, k - 3), 160), 3000, lower=True)
    length = k + 1 if mode == 1 else k
    ek, ec, nwin = oracle_counts(oracle, reads, length, mode)
    for path in (0, 1):          # segment-hash path (with its fallback) and LSD-only path
        ks, cs, c = gpu_counts(reads, k, mode, path=path)
        assert c.windows == nwin
        assert c.keys == nwin * (2 if mode else 1)
        assert c.distinct == len(ek)
        assert ks == ek
        assert cs == ec


def test_high_coverage_counts(oracle):
    rng = random.Random(7)
    reads = make_reads(rng, 4000, 150, 2000, n_rate=0.001)
    ek, ec, nwin = oracle_counts(oracle, reads, 25, 0)
    ks, cs, c = gpu_counts(reads, 25, 0)
    assert (ks, cs) == (ek, ec)
    assert max(cs) > 255


def test_segment_overflow_falls_back(oracle):
    """More distinct keys in one top-16-bit segment than the LDS table holds: the library must
    fall back to the full sort and still give the oracle's answer."""
    rng = random.Random(21)
    prefix = "ACGTACGT"
    reads = [prefix + "".join(rng.choice("ACGT") for _ in range(17)) for _ in range(6000)]
    reads += make_reads(rng, 200, 150, 3000)
    ek, ec, nwin = oracle_counts(oracle, reads, 25, 1)
    ks, cs, c = gpu_counts(reads, 24, 1)
    assert c.windows == nwin
    assert (ks, cs) == (ek, ec)


def test_multi_push_and_small_budget_merge(oracle):
    rng = random.Random(8)
    reads = make_reads(rng, 3000, 150, 20000)
    ek, ec, nwin = oracle_counts(oracle, reads, 25, 0)
    # 8 MiB budget forces several chunks per push plus run merging
    ks, cs, c = gpu_counts(reads, 25, 0, budget=8 * MB, pushes=3)
    assert c.windows == nwin
    assert (ks, cs) == (ek, ec)


def _suffix_map(files, prefix):
    return {name[len(prefix):]: data for name, data in files.items()}


@pytest.mark.parametrize("k", [25, 12, 31, 33, 55])
def test_kmer_set_files_bit_identical(oracle, k):
    rng = random.Random(50 + k)
    reads = make_reads(rng, 400, 150, 30000)
    fq = "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads))
    exp, nwin = oracle.build_kmer_set([(oracle.FASTQ, "r.fq", fq)], k, out="ks")
    with g.Context(k, g.MODE_KMER_SET, hbm_budget=256 * MB) as ctx:
        ctx.push_host("\n".join(reads) + "\n")
        c = ctx.finish()
        got = ctx.emit()
    assert c.windows == nwin
    exp = _suffix_map(exp, "ks")
    assert sorted(got) == sorted(exp)
    for name in exp:
        assert got[name] == exp[name], name


@pytest.mark.parametrize("k", [15, 27, 30, 31, 55])
def test_graph_files_bit_identical(oracle, k):
    rng = random.Random(90 + k)
    # a tiny genome so that counts exceed 255 and (for one case) 65535
    reads = make_reads(rng, 3000, 150, 400, n_rate=0.0005)
    if k == 15:
        core = "".join(rng.choice("ACGT") for _ in range(40))
        reads += [core] * 70000
    fa = "".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads))
    exp, nwin = oracle.build_graph([(oracle.FASTA, "r.fa", fa)], k, out="gr")
    with g.Context(k, g.MODE_GRAPH, hbm_budget=512 * MB) as ctx:
        ctx.push_host("\n".join(reads) + "\n")
        c = ctx.finish()
        got = ctx.emit()
    assert c.windows == nwin
    exp = _suffix_map(exp, "gr")
    assert sorted(got) == sorted(exp)
    for name in exp:
        assert got[name] == exp[name], name
    if k == 15:
        assert len(exp["-counts.ord2"]) > 0      # counts > 65535 exercised


def test_reference_test_cases(oracle):
    """The inputs of the reference's own tests (testGossCmdBuildGraph.cc:114-179)."""
    with g.Context(27, g.MODE_GRAPH, hbm_budget=64 * MB) as ctx:
        ctx.push_host("AAAAAAAAAAAAAAAAAAAAAAAAAAAA\n")
        c = ctx.finish()
        ks, cs = ctx.result()
    assert c.distinct == 2 and list(cs) == [1, 1]
    assert ks[1] == oracle.revcomp(ks[0], 28)
    with g.Context(15, g.MODE_GRAPH, hbm_budget=64 * MB) as ctx:
        ctx.push_host("NACTTTTGATGCAATGTCAAATTCTCCNCGTCATTCGCAACTGAATACAAGNGAATTTGGAAGGAGAATNTGGTA\n")
        c = ctx.finish()
    assert c.distinct == 42


def test_empty_and_degenerate_inputs(oracle):
    for payload in ["", "\n\n\n", "ACGT\nACG\n", "NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN\n"]:
        with g.Context(25, g.MODE_KMER_SET, hbm_budget=64 * MB) as ctx:
            ctx.push_host(payload)
            c = ctx.finish()
            got = ctx.emit()
        assert c.distinct == 0 and c.windows == 0
        exp = _suffix_map(oracle.write_kmer_set([], 25, 0, out="ks"), "ks")
        assert sorted(got) == sorted(exp)
        for name in exp:
            assert got[name] == exp[name], name


def _sparse_case(oracle, positions, N, M, words):
    import torch
    exp = _suffix_map(oracle.write_sparse_array(positions, N, M, base="sa"), "sa")
    flat = []
    for p in positions:
        flat.append(p & 0xFFFFFFFFFFFFFFFF)
        if words == 2:
            flat.append(p >> 64)
    # torch has no uint64 arithmetic but can carry the bit patterns
    t = torch.tensor([x - (1 << 64) if x >= (1 << 63) else x for x in flat] or [0], dtype=torch.int64, device="cuda")
    with g.Context(25, g.MODE_KMER_SET, hbm_budget=256 * MB) as ctx:
        got = ctx.emit_sparse_array(t.data_ptr(), words, len(positions), N, M)
    assert sorted(got) == sorted(exp)
    for name in exp:
        assert got[name] == exp[name], name


def test_sparse_array_dense_select_block_kinds(oracle):
    """Densities that produce small, intermediate and large DenseSelect blocks
    (the shapes testDenseArray.cc:26-617 exercises)."""
    rng = random.Random(3)
    # uniform, dense: small blocks + last partial block
    pos = sorted(rng.sample(range(1 << 26), 40000))
    _sparse_case(oracle, pos, 1 << 26, len(pos), 1)
    # deliberately bad estimate M -> small D -> sparse bitmap: intermediate / large blocks
    pos = sorted(rng.sample(range(1 << 40), 30000))
    _sparse_case(oracle, pos, 1 << 40, 1 << 22, 1)
    # clustered: dense clumps separated by huge gaps (mix of block kinds in d0 and d1)
    pos = set()
    for c in range(6):
        base = rng.randrange(1 << 44)
        for _ in range(9000):
            pos.add(base + rng.randrange(1 << (8 + 3 * c)))
    pos = sorted(pos)
    _sparse_case(oracle, pos, 1 << 46, len(pos), 1)
    _sparse_case(oracle, pos, 1 << 46, 1 << 28, 1)


def test_sparse_array_wide_universes(oracle):
    """72- and 100-bit universes (testSparseArray.cc:27-307)."""
    rng = random.Random(4)
    for bits, n in [(72, 5000), (100, 20000), (126, 300)]:
        pos = sorted({rng.getrandbits(bits) for _ in range(n)})
        _sparse_case(oracle, pos, 1 << bits, len(pos), 2)
    _sparse_case(oracle, [], 1 << 100, 0, 2)
    _sparse_case(oracle, [5], 1 << 20, 1, 1)


def test_push_device_misaligned(oracle):
    import torch
    rng = random.Random(11)
    reads = make_reads(rng, 500, 150, 5000)
    payload = ("\n".join(reads) + "\n").encode()
    ek, ec, nwin = oracle_counts(oracle, reads, 25, 0)
    for shift in (0, 1, 7, 13):
        buf = torch.zeros(len(payload) + 64, dtype=torch.uint8, device="cuda")
        buf[shift:shift + len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).cuda()
        with g.Context(25, g.MODE_KMER_SET, hbm_budget=128 * MB) as ctx:
            ctx.push_device(buf.data_ptr() + shift, len(payload))
            c = ctx.finish()
            ks, cs = ctx.result()
        assert c.windows == nwin
        assert (ks, [int(x) for x in cs]) == (ek, ec)


def test_synth_generator_matches_host():
    import torch
    n, L, G = 2000, 150, 100000
    host = g.synth_reads_host(n, L, G, seed=1, first_read=5)
    buf = torch.zeros(n * (L + 1), dtype=torch.uint8, device="cuda")
    with g.Context(25, hbm_budget=64 * MB) as ctx:
        ctx.synth_reads(buf.data_ptr(), n, L, G, seed=1, first_read=5)
    assert bytes(buf.cpu().numpy().tobytes()) == host
    assert host.count(b"N") == len([r for r in range(5, 5 + n) if r % 97 == 96])


def test_goss_cli_end_to_end(oracle, tmp_path):
    """The goss executable: FASTQ (+ .gz) / FASTA / line inputs -> files on disk, byte for byte
    what the oracle's restatement of GossCmdBuildKmerSet / GossCmdBuildGraph writes."""
    import gzip
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    goss = os.path.join(root, "gossamer_amd", "goss")
    rng = random.Random(77)
    reads = make_reads(rng, 600, (30, 150), 8000, lower=True)
    fq = "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads[:300]))
    fa = "".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads[300:500]))
    ln = "\n".join(reads[500:]) + "\n"
    (tmp_path / "a.fq").write_text(fq)
    with gzip.open(tmp_path / "a.fq.gz", "wb") as f:
        f.write(fq.encode())
    (tmp_path / "b.fa").write_text(fa)
    (tmp_path / "c.txt").write_text(ln)
    inputs = [(oracle.LINE, "c.txt", ln), (oracle.FASTA, "b.fa", fa), (oracle.FASTQ, "a.fq", fq)]
    for cmd, k, obuild, base, fqname in (("build-kmer-set", 25, oracle.build_kmer_set, "ks", "a.fq"),
                                         ("build-graph", 27, oracle.build_graph, "gr", "a.fq.gz"),
                                         ("build-graph", 55, oracle.build_graph, "g55", "a.fq")):
        exp, nwin = obuild(inputs, k, out=base)
        out = tmp_path / base
        p = subprocess.run([goss, cmd, "-k", str(k), "-i", str(tmp_path / fqname), "-I", str(tmp_path / "b.fa"),
                            "--line-in", str(tmp_path / "c.txt"), "-O", str(out), "--hbm-budget", "1", "-v"],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        assert b"total build time" in p.stderr
        got = {n: (tmp_path / n).read_bytes() for n in os.listdir(tmp_path) if n.startswith(base + ".") or n.startswith(base + "-")}
        assert sorted(got) == sorted(exp)
        for name in exp:
            assert got[name] == exp[name], name
    # an input without reads is an error, like the reference's KmerizingAdapter
    (tmp_path / "empty.fq").write_text("")
    p = subprocess.run([goss, "build-kmer-set", "-k", "25", "-i", str(tmp_path / "empty.fq"), "-O", str(tmp_path / "e"), "--hbm-budget", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 1 and p.stderr.decode() == "error performing build-kmer-set:\nNo valid reads."


def test_medium_scale_files_vs_oracle(oracle):
    """300,000 synthetic 150 bp reads (37.7 M k-mers, ~30x coverage of a 1.5 Mbp genome): large
    enough that the partition passes run thousands of tiles (unstable fast ranking, look-back
    chain, LDS hash segments) yet small enough for the oracle."""
    reads = g.synth_reads_host(300000, 150, 1500000, seed=3)
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], 25, out="ks")
    exp = _suffix_map(exp, "ks")
    for path in (0, 1):
        with g.Context(25, g.MODE_KMER_SET, hbm_budget=8 << 30) as ctx:
            ctx.set_path(path)
            ctx.push_host(reads)
            c = ctx.finish()
            got = ctx.emit()
        assert c.windows == nwin
        assert sorted(got) == sorted(exp)
        for name in exp:
            assert got[name] == exp[name], (path, name)


def test_large_scale_properties():
    """8 M reads (1.0 G k-mers) generated on the device: the segment-hash path and the LSD-only
    path must agree exactly; counts must add up to the number of windows; keys must be strictly
    increasing (size-independent properties at a size the oracle cannot reach)."""
    import torch
    from gossamer_amd import dist as gd
    n, L, G = 8_000_000, 150, 8_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    res = []
    for path in (0, 1):
        ctx = g.Context(25, g.MODE_KMER_SET, hbm_budget=40 << 30)
        if path == 0:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=5)
        ctx.set_path(path)
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        kp, cp, m = ctx.result_ptrs()
        keys = gd.device_view(kp, m, torch.int64, "cuda").clone()
        counts = gd.device_view(cp, m, torch.int32, "cuda").clone()
        assert m == c.distinct and c.keys == c.windows
        assert int(counts.to(torch.int64).sum().item()) == c.windows
        assert bool((keys[1:] > keys[:-1]).all().item())
        res.append((keys, counts, c.windows))
        ctx.close()
    assert res[0][2] == res[1][2]
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    # windows = reads * 126 minus the windows an 'N' removes
    assert res[0][2] <= n * (L - 25 + 1) and res[0][2] > n * (L - 25 + 1) * 0.99


@pytest.fixture(scope="module")
def nccl_world1():
    """one RCCL process group for the whole module (setting one up takes tens of seconds)"""
    import os
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("k,graph", [(25, False), (40, False), (27, True), (55, True)])
def test_distributed_path_world1(oracle, nccl_world1, k, graph):
    """The multi-GPU code path (exchange over RCCL, merge of received runs, gather to rank 0,
    emit) with a single rank: must give exactly the single-GPU files -- k-mer sets and graphs,
    one- and two-word keys."""
    import os
    import torch
    import torch.distributed as dist
    from gossamer_amd import dist as gd
    reads = g.synth_reads_host(20000, 150, 100000, seed=9)
    if graph:
        exp, nwin = oracle.build_graph([(oracle.LINE, "reads", reads)], k, out="ob")
    else:
        exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], k, out="ob")
    exp = _suffix_map(exp, "ob")
    buf = torch.frombuffer(bytearray(reads), dtype=torch.uint8).cuda()
    with g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=1 << 30) as ctx:
        r = gd.count_distributed(ctx, buf.data_ptr(), buf.numel(), 2 * (k + 1 if graph else k), torch.device("cuda", 0))
        got = gd.assemble_files([ctx.files()])
        # the first form of the path (ranges gathered on rank 0, which builds everything) gives the same files
        gd.count_distributed(ctx, buf.data_ptr(), buf.numel(), 2 * (k + 1 if graph else k), torch.device("cuda", 0), emission="root")
        assert ctx.files() == got
    assert r["windows"] == nwin
    assert sorted(got) == sorted(exp)
    for name in exp:
        assert got[name] == exp[name], name


@pytest.mark.parametrize("k,graph", [(25, False), (27, True)])
def test_record_exchange_world1(oracle, nccl_world1, k, graph):
    """The exchange before counting (super-k-mer records routed by minimizer, all-to-all over RCCL, count of what was
    received) with a single rank: exactly the single-GPU files."""
    import torch
    from gossamer_amd import dist as gd
    reads = g.synth_reads_host(20000, 150, 100000, seed=9)
    exp, nwin = (oracle.build_graph if graph else oracle.build_kmer_set)([(oracle.LINE, "reads", reads)], k, out="ob")
    exp = _suffix_map(exp, "ob")
    buf = torch.frombuffer(bytearray(reads), dtype=torch.uint8).cuda()
    with g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=1 << 30) as ctx:
        r = gd.count_distributed(ctx, buf.data_ptr(), buf.numel(), 2 * (k + 1 if graph else k), torch.device("cuda", 0), exchange="records")
        got = gd.assemble_files([ctx.files()])
        # in pieces: asynchronous all-to-alls on RCCL's stream beside the library's kernels
        r4 = gd.count_distributed(ctx, buf.data_ptr(), buf.numel(), 2 * (k + 1 if graph else k), torch.device("cuda", 0), exchange="records",
                                  record_pieces=4)
        assert r4["windows"] == nwin and gd.assemble_files([ctx.files()]) == got
    assert r["windows"] == nwin
    assert sorted(got) == sorted(exp)
    for name in exp:
        assert got[name] == exp[name], name


@pytest.mark.parametrize("k,graph", [(25, False), (27, True), (55, True)])
def test_one_rank_takes_the_load_of_an_eight_rank_build(oracle, nccl_world1, k, graph):
    """C3's eight-way cut with one process: GOSS_DIST_ROUTE_PARTS=8 makes the rank cut its reads into records for
    EIGHT destinations (the routing kernel's eight-part form: minimizer -> part, a buffer per part) and count all
    eight parts itself -- every window once, whatever part it went to -- then the range exchange and the distributed
    emission as in a build over ranks.  Files equal to the oracle's build of the reads."""
    import os
    import torch
    from gossamer_amd import dist as gd
    reads = g.synth_reads_host(20000, 150, 100000, seed=13)
    exp, nwin = (oracle.build_graph if graph else oracle.build_kmer_set)([(oracle.LINE, "reads", reads)], k, out="ob")
    exp = _suffix_map(exp, "ob")
    buf = torch.frombuffer(bytearray(reads), dtype=torch.uint8).cuda()
    os.environ["GOSS_DIST_ROUTE_PARTS"] = "8"
    try:
        with g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=1 << 30) as ctx:
            r = gd.count_distributed(ctx, buf.data_ptr(), buf.numel(), 2 * (k + 1 if graph else k), torch.device("cuda", 0), exchange="records")
            got = gd.assemble_files([ctx.files()])
            sizes = gd._ROUTE_SIZES[(buf.data_ptr(), buf.numel(), 8, ctx.k, ctx.mode)]
    finally:
        os.environ.pop("GOSS_DIST_ROUTE_PARTS", None)
    assert len(sizes) == 8 and all(n > 0 for n in sizes), sizes          # (eight parts were cut, none of them empty)
    assert r["windows"] == nwin
    assert sorted(got) == sorted(exp)
    for name in exp:
        assert got[name] == exp[name], name


def test_all_to_all_segments_above_one_gib(nccl_world1):
    """RCCL 2.26 drops the second half of an all-to-all segment above 1 GiB without an error (found on this box: one rank
    sending to itself).  gossamer_amd.dist moves every segment in rounds of 512 MiB: 1.5 GiB must arrive whole."""
    import torch
    from gossamer_amd import dist as gd
    n = 3 << 29
    src = torch.randint(0, 251, (n,), dtype=torch.uint8, device="cuda")
    big = torch.empty(n + 4096, dtype=torch.uint8, device="cuda")
    big[100:100 + n] = src
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    gd.all_to_all_views([out], [big[100:100 + n]], n)
    torch.cuda.synchronize()
    assert torch.equal(out, src)
    # int64 views (keys) of the same size
    out64 = torch.zeros(n // 8, dtype=torch.int64, device="cuda")
    gd.all_to_all_views([out64], [src.view(torch.int64)], n)
    torch.cuda.synchronize()
    assert torch.equal(out64, src.view(torch.int64))


@pytest.mark.parametrize("k", [25, 45])
def test_distributed_set_algebra_world1(oracle, nccl_world1, k):
    """BASELINE config C5 through the multi-GPU code path with one rank: two (three) k-mer sets
    counted and range-partitioned, intersected / subtracted range by range, assembled on rank 0 --
    files equal to the oracle's restatement of intersect-kmer-sets / subtract-kmer-set."""
    import torch
    from gossamer_amd import dist as gd
    dev = torch.device("cuda", 0)
    texts = [g.synth_reads_host(6000, 150, 400000, seed=71, first_read=f) for f in (0, 3000)] + [b"ACGTACGT\n"]
    files, names = {}, []
    for i, t in enumerate(texts):
        f, _ = oracle.build_kmer_set([(oracle.LINE, "reads", t)], k, out="s%d" % i)
        files.update(f)
        names.append("s%d" % i)
    bufs = [torch.frombuffer(bytearray(t), dtype=torch.uint8).cuda() for t in texts]
    ins = [(b.data_ptr(), b.numel()) for b in bufs]
    with g.Context(k, g.MODE_KMER_SET, hbm_budget=1 << 30) as ctx:
        # all inputs empty: undefined in the reference (the oracle refuses); here the empty set
        r = gd.set_algebra_distributed(ctx, [ins[2]], 2 * k, "intersect", dev)
        assert r["M"] == 0 and r["sizes"] == [0] and struct.unpack("<QQQ", ctx.files()[".header"])[2] == 0
        for sel, op in (((0, 1), "intersect"), ((0, 2, 1), "intersect"), ((0, 1), "subtract"),
                        ((1, 0), "subtract"), ((0, 2), "subtract"), ((2, 0), "subtract"), ((0, 0), "subtract")):
            if op == "intersect":
                exp = oracle.intersect_kmer_sets(files, [names[j] for j in sel], "out")
            else:
                exp = oracle.subtract_kmer_set(files, names[sel[0]], names[sel[1]], "out")
            r = gd.set_algebra_distributed(ctx, [ins[j] for j in sel], 2 * k, op, dev)
            got = gd.assemble_files([ctx.files()])
            exp = _suffix_map(exp, "out")
            assert sorted(got) == sorted(exp), (sel, op)
            for name in exp:
                assert got[name] == exp[name], (sel, op, name)
            assert r["M"] == struct.unpack("<QQQ", got[".header"])[2]


def test_goss_merge_commands(oracle, tmp_path):
    """goss merge-kmer-sets / merge-graphs (GossCmdMerge.tcc:151-326): objects built by the product
    are decoded on the device, merged, and written with the estimate M = sum of the input counts;
    every output file must equal the oracle's restatement of the reference merge, for one- and
    two-word keys, a full merge and --max-merge 2 (intermediate objects change the estimate)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    goss = os.path.join(root, "gossamer_amd", "goss")
    rng = random.Random(41)
    parts = []
    for p in range(4):
        reads = make_reads(rng, 400, (60, 150), 20000 if p < 3 else 500)
        txt = "\n".join(reads) + "\n"
        (tmp_path / ("p%d.txt" % p)).write_text(txt)
        parts.append(txt)

    def run(args):
        p = subprocess.run([goss] + args + ["--hbm-budget", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        return p

    def disk(base):
        return {n: (tmp_path / n).read_bytes() for n in os.listdir(tmp_path)
                if n.startswith(base + ".") or n.startswith(base + "-")}

    for cmd_b, cmd_m, kind, k in (("build-kmer-set", "merge-kmer-sets", 0, 25), ("build-graph", "merge-graphs", 1, 27),
                                  ("build-kmer-set", "merge-kmer-sets", 0, 51), ("build-graph", "merge-graphs", 1, 55)):
        tag = "%s%d" % ("gr" if kind else "ks", k)
        files = {}
        names = []
        for i in range(4):
            base = "%s_%d" % (tag, i)
            p = run([cmd_b, "-k", str(k), "--line-in", str(tmp_path / ("p%d.txt" % i)), "-O", str(tmp_path / base)])
            assert p.returncode == 0, p.stderr.decode()
            files.update(disk(base))
            names.append(base)
        (tmp_path / "list.txt").write_text("".join(str(tmp_path / n) + "\n" for n in names[2:]))
        for mm in (8, 2):
            out = "%s_m%d" % (tag, mm)
            exp = oracle.merge(files, names, kind, out, max_merge=mm)
            args = [cmd_m, "-G", str(tmp_path / names[0]), "-G", str(tmp_path / names[1]),
                    "--graphs-in", str(tmp_path / "list.txt"), "-O", str(tmp_path / out), "--max-merge", str(mm), "-v"]
            p = run(args)
            assert p.returncode == 0, p.stderr.decode()
            got = disk(out)
            assert sorted(got) == sorted(exp), (cmd_m, k, mm)
            for name in exp:
                assert got[name] == exp[name], (cmd_m, k, mm, name)
            if mm == 2:
                # --tmp-dir given: the partial merges are written there as runs, read back and removed by the command
                # (GossCmdMerge.tcc:176-208's temporary objects) -- same files, nothing left behind
                tdir = tmp_path / ("tmp_" + tag)
                tdir.mkdir()
                for n in got:
                    os.remove(tmp_path / n)
                p = run(args + ["--tmp-dir", str(tdir)])
                assert p.returncode == 0, p.stderr.decode()
                assert p.stderr.decode().count("written to " + str(tdir)) == 2, p.stderr.decode()
                assert disk(out) == got and os.listdir(tdir) == []
    # a single input is a re-encode with M = its own count
    exp = oracle.merge(files, names[:1], 1, "one")
    p = run(["merge-graphs", "-G", str(tmp_path / names[0]), "-O", str(tmp_path / "one")])
    assert p.returncode == 0, p.stderr.decode()
    got = disk("one")
    assert sorted(got) == sorted(exp) and all(got[n] == exp[n] for n in exp)
    # error texts
    p = run(["merge-graphs", "-O", str(tmp_path / "x")])
    assert p.returncode == 1
    assert "At least one input graph must be supplied either using --graph-in or --graphs-in." in p.stderr.decode()
    p = run(["merge-graphs", "-G", str(tmp_path / "gr27_0"), "-G", str(tmp_path / "gr55_0"), "-O", str(tmp_path / "x")])
    assert p.returncode == 1 and "must have the same kmer-size" in p.stderr.decode()


def test_goss_graph_to_kmer_set(oracle, tmp_path):
    """goss graph-to-kmer-set (GossCmdGraphToKmerSet.cc:30-59): graphs built by the product, their
    normal edges selected on the device and written as a k-mer set of k + 1 sized with the edge
    count -- every file equal to the oracle's, for one-word edges, edges that need two words as
    k-mers of k + 1 = 32, and two-word edges."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    goss = os.path.join(root, "gossamer_amd", "goss")
    rng = random.Random(47)
    reads = make_reads(rng, 600, (60, 150), 30000)
    reads.append("ACGT" * 30)
    (tmp_path / "r.txt").write_text("\n".join(reads) + "\n")

    def run(args):
        return subprocess.run([goss] + args + ["--hbm-budget", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)

    def disk(base):
        return {n: (tmp_path / n).read_bytes() for n in os.listdir(tmp_path) if n.startswith(base + ".") or n.startswith(base + "-")}

    for k in (27, 31, 55):
        g_name, s_name = "g%d" % k, "s%d" % k
        p = run(["build-graph", "-k", str(k), "--line-in", str(tmp_path / "r.txt"), "-O", str(tmp_path / g_name)])
        assert p.returncode == 0, p.stderr.decode()
        exp = oracle.graph_to_kmer_set(disk(g_name), g_name, s_name)
        p = run(["graph-to-kmer-set", "-G", str(tmp_path / g_name), "-O", str(tmp_path / s_name), "-v"])
        assert p.returncode == 0, p.stderr.decode()
        assert "building (k+1)-mer set" in p.stderr.decode()
        got = disk(s_name)
        assert sorted(got) == sorted(exp), k
        for name in exp:
            assert got[name] == exp[name], (k, name)
    p = run(["graph-to-kmer-set", "-O", str(tmp_path / "bad")])
    assert p.returncode == 1 and "mandatory option graph-in was not given." in p.stderr.decode()
    p = run(["graph-to-kmer-set", "-G", str(tmp_path / "nope"), "-O", str(tmp_path / "bad")])
    assert p.returncode == 1


def test_goss_set_algebra_commands(oracle, tmp_path):
    """goss intersect-kmer-sets / subtract-kmer-set / merge-and-annotate-kmer-sets on sets built by
    the product, every output file compared with the oracle's restatement of the reference loops
    (GossCmdIntersectKmerSets.cc, GossCmdSubtractKmerSet.cc, GossCmdMergeAndAnnotateKmerSets.cc);
    one- and two-word keys, an empty input, identical inputs."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    goss = os.path.join(root, "gossamer_amd", "goss")
    rng = random.Random(43)
    genome = "".join(rng.choice("ACGT") for _ in range(30000))
    for p in range(3):
        lo = 5000 * p
        reads = [genome[s:s + 120] for s in (rng.randrange(lo, lo + 18000) for _ in range(1500))]
        (tmp_path / ("p%d.txt" % p)).write_text("\n".join(reads) + "\n")
    (tmp_path / "p3.txt").write_text("ACGTACGT\n")          # too short for any k used: an empty set

    def run(args):
        return subprocess.run([goss] + args + ["--hbm-budget", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)

    def disk(base):
        return {n: (tmp_path / n).read_bytes() for n in os.listdir(tmp_path) if n.startswith(base + ".")}

    def same(base, exp, what):
        got = disk(base)
        assert sorted(got) == sorted(exp), what
        for name in exp:
            assert got[name] == exp[name], (what, name)

    for k in (25, 45):
        files = {}
        names = []
        for i in range(4):
            base = "s%d_%d" % (k, i)
            p = run(["build-kmer-set", "-k", str(k), "--line-in", str(tmp_path / ("p%d.txt" % i)), "-O", str(tmp_path / base)])
            assert p.returncode == 0, p.stderr.decode()
            files.update(disk(base))
            names.append(base)
        P = [str(tmp_path / n) for n in names]
        # intersection of three sets, and with the empty set in the list (skipped by the reference)
        for sel, tag in (((0, 1, 2), "x3"), ((0, 3, 1), "xe"), ((2,), "x1")):
            out = "i%d_%s" % (k, tag)
            exp = oracle.intersect_kmer_sets(files, [names[j] for j in sel], out)
            args = ["intersect-kmer-sets", "-O", str(tmp_path / out)]
            for j in sel:
                args += ["-G", P[j]]
            p = run(args)
            assert p.returncode == 0, p.stderr.decode()
            same(out, exp, (k, tag))
        # differences
        for a, b, tag in ((0, 1, "ab"), (1, 0, "ba"), (0, 3, "ae"), (0, 0, "aa"), (3, 0, "ea")):
            out = "d%d_%s" % (k, tag)
            exp = oracle.subtract_kmer_set(files, names[a], names[b], out)
            p = run(["subtract-kmer-set", "-G", P[a], "-G", P[b], "-O", str(tmp_path / out)])
            assert p.returncode == 0, p.stderr.decode()
            same(out, exp, (k, tag))
        # annotated union
        out = "u%d" % k
        exp, stats = oracle.merge_and_annotate(files, names[0], names[1], out)
        p = run(["merge-and-annotate-kmer-sets", "-G", P[0], "-G", P[1], "-O", str(tmp_path / out)])
        assert p.returncode == 0, p.stderr.decode()
        assert p.stdout.decode() == "%d\t%d\t%d\n" % stats
        same(out, exp, (k, "annotate"))
        p = run(["merge-and-annotate-kmer-sets", "-G", P[0], "-G", P[3], "-O", str(tmp_path / "bad")])
        assert p.returncode == 1 and p.stderr.decode() == "caught unknown exception\n"
    p = run(["subtract-kmer-set", "-G", P[0], "-O", str(tmp_path / "bad")])
    assert p.returncode == 1 and "Exactly two input k-mer sets required!" in p.stderr.decode()
    p = run(["merge-and-annotate-kmer-sets", "-G", P[0], "-O", str(tmp_path / "bad")])
    assert p.returncode == 1 and "mandatory option graph-in must be supplied exactly twice." in p.stderr.decode()


@pytest.mark.parametrize("mode", [0, 1])
def test_every_k(oracle, mode):
    """Every kmer-size the commands accept (build-kmer-set 1..63, build-graph 1..62): keys and
    counts against the oracle on a small ragged input -- every key width, both word counts, every
    byte class of the hash, the 31/32 and 30/31 boundaries."""
    rng = random.Random(77 + mode)
    reads = make_reads(rng, 220, (3, 140), 1500, lower=True)
    for k in range(1, 64 - mode):
        length = k + 1 if mode == 1 else k
        ek, ec, nwin = oracle_counts(oracle, reads, length, mode)
        ks, cs, c = gpu_counts(reads, k, mode)
        assert c.windows == nwin, k
        assert ks == ek, k
        assert cs == ec, k
