"""Several contexts in one process (goss_gpu_group_exchange / goss_gpu_group_emit -- what `goss --devices`
drives): every context counts a share of the reads, the ranges are exchanged device to device, every
context emits its span and context 0 the index.  One MI355X is enough: the contexts share cuda:0.  The
assembled files must equal the oracle's single build of all reads."""
import struct

import pytest

import gossamer_amd as g
from gossamer_amd import dist as gd

pytestmark = pytest.mark.gpu


def _suffix_map(files, prefix):
    return {k[len(prefix):]: v for k, v in files.items()}


def _split_reads(text, parts):
    lines = text.split(b"\n")[:-1]
    per = (len(lines) + parts - 1) // parts
    return [b"".join(l + b"\n" for l in lines[i * per:(i + 1) * per]) for i in range(parts)]


def _group_build(shards, k, mode):
    ctxs = [g.Context(k, mode, hbm_budget=768 << 20) for _ in shards]
    try:
        windows = 0
        for c, s in zip(ctxs, shards):
            if s:
                c.push_host(s)
            windows += c.finish().windows
        sizes = g.group_exchange(ctxs, sample_per_context=256)
        assert sizes == [c.result_ptrs()[2] for c in ctxs]
        g.group_emit(ctxs)
        per_ctx = [c.files() for c in ctxs]
        for other in per_ctx[1:]:
            assert all(".low-bits" in n or n == "-counts.ord0" or n.startswith(".part.") for n in other), sorted(other)
        return gd.assemble_files(per_ctx), sizes, windows
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("kind,k", [("kmer", 25), ("kmer", 45), ("graph", 27), ("graph", 55)])
@pytest.mark.parametrize("parts", [2, 3])
def test_group_of_contexts_builds_the_oracles_object(oracle, kind, k, parts):
    reads = g.synth_reads_host(18000, 150, 120000, seed=29)
    build = oracle.build_graph if kind == "graph" else oracle.build_kmer_set
    exp, nwin = build([(oracle.LINE, "reads", reads)], k, out="ob")
    exp = _suffix_map(exp, "ob")
    M = struct.unpack("<8Q", exp[("-edges" if kind == "graph" else ".kmers") + ".header"])[7]
    got, sizes, windows = _group_build(_split_reads(reads, parts), k, g.MODE_GRAPH if kind == "graph" else g.MODE_KMER_SET)
    assert windows == nwin and sum(sizes) == M
    assert max(sizes) <= 1.3 * M / parts + 64, sizes
    assert sorted(got) == sorted(exp)
    for name in exp:
        assert got[name] == exp[name], name


def test_group_with_an_empty_member_and_a_single_member(oracle):
    reads = g.synth_reads_host(6000, 150, 50000, seed=31)
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], 21, out="ob")
    exp = _suffix_map(exp, "ob")
    for shards in ([reads, b""], [reads]):
        got, sizes, windows = _group_build(shards, 21, g.MODE_KMER_SET)
        assert windows == nwin
        assert sorted(got) == sorted(exp)
        for name in exp:
            assert got[name] == exp[name], name


def test_group_refuses_mixed_contexts():
    with g.Context(21, g.MODE_KMER_SET, hbm_budget=256 << 20) as a, g.Context(23, g.MODE_KMER_SET, hbm_budget=256 << 20) as b:
        a.push_host(b"ACGTACGTACGTACGTACGTACGTACGTA\n")
        b.push_host(b"ACGTACGTACGTACGTACGTACGTACGTA\n")
        a.finish()
        b.finish()
        with pytest.raises(g.GossGpuError):
            g.group_exchange([a, b])
        with pytest.raises(g.GossGpuError):
            g.group_exchange([a, a])


def test_goss_devices_option(oracle, tmp_path):
    """`goss build-kmer-set / build-graph --devices 0,0[,0]`: one context per listed device (here the same GPU
    several times), batches dealt round the contexts by feeder threads, ranges exchanged, every context's slices
    written behind one another -- files byte for byte the oracle's; FASTQ parsed in parallel chunks whose buffers
    are handed over to the feeders (GOSS_PARSE_CHUNK makes the chunks small), plus FASTA and line input."""
    import os
    import random
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    goss = os.path.join(root, "gossamer_amd", "goss")
    rng = random.Random(5)
    genome = "".join(rng.choice("ACGT") for _ in range(40000))
    reads = []
    for _ in range(9000):
        L = rng.randint(40, 150)
        p = rng.randint(0, len(genome) - L)
        reads.append(genome[p:p + L])
    fq = "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads[:8000]))
    fa = "".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads[8000:8500]))
    ln = "\n".join(reads[8500:]) + "\n"
    (tmp_path / "a.fq").write_text(fq)
    (tmp_path / "b.fa").write_text(fa)
    (tmp_path / "c.txt").write_text(ln)
    inputs = [(oracle.LINE, "c.txt", ln), (oracle.FASTA, "b.fa", fa), (oracle.FASTQ, "a.fq", fq)]
    env = dict(os.environ, GOSS_PARSE_CHUNK="65536")
    for cmd, k, obuild, base, devices in (("build-kmer-set", 25, oracle.build_kmer_set, "ks", "0,0"),
                                          ("build-graph", 27, oracle.build_graph, "gr", "0,0,0"),
                                          ("build-graph", 55, oracle.build_graph, "g55", "0,0")):
        exp, nwin = obuild(inputs, k, out=base)
        out = tmp_path / base
        p = subprocess.run([goss, cmd, "-k", str(k), "-i", str(tmp_path / "a.fq"), "-I", str(tmp_path / "b.fa"),
                            "--line-in", str(tmp_path / "c.txt"), "-O", str(out), "--hbm-budget", "1", "-T", "4", "-v",
                            "--devices", devices],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
        assert p.returncode == 0, p.stderr.decode()
        assert ("counted on %d devices" % len(devices.split(","))).encode() in p.stderr
        assert ("k-mer windows: %d," % nwin).encode() in p.stderr
        got = {n: (tmp_path / n).read_bytes() for n in os.listdir(tmp_path) if n.startswith(base + ".") or n.startswith(base + "-")}
        assert sorted(got) == sorted(exp)
        for name in exp:
            assert got[name] == exp[name], name
    p = subprocess.run([goss, "build-kmer-set", "-k", "25", "-i", str(tmp_path / "a.fq"), "-O", str(tmp_path / "x"), "--devices", "0,x"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode != 0 and b"--devices" in p.stderr
