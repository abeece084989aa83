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
