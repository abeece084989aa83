"""goss_gpu_push_keys_{host,device}: the raw-key entry of the boundary -- what the templated
GossCmdBuildKmerSet::operator()(cxt, KmerSrc&) (GossCmdBuildKmerSet.hh:23-30, .tcc:246-256; electus: ElectApp.cc:183-215)
does with the k-mers a caller cuts out itself: normalise every one, count.  The k-mers come from the oracle's
go_kmerize (GossReadBaseString.hh:52-188), un-normalised, shuffled."""
import random

import numpy as np
import pytest

import gossamer_amd as g
from test_gpu_parity import make_reads, oracle_counts

pytestmark = pytest.mark.gpu
MB = 1 << 20


def raw_kmers(oracle, reads, k):
    out = []
    for r in reads:
        out += oracle.kmerize(r, k)
    return out


def as_words(keys, words):
    if words == 1:
        return np.array(keys, dtype=np.uint64)
    a = np.empty((len(keys), 2), dtype=np.uint64)
    a[:, 0] = [x & ((1 << 64) - 1) for x in keys]
    a[:, 1] = [x >> 64 for x in keys]
    return a


@pytest.mark.parametrize("k", [11, 25, 31, 32, 47, 63])
def test_raw_kmers_are_normalised_and_counted(oracle, k):
    rng = random.Random(4200 + k)
    reads = make_reads(rng, 400, (max(5, k - 3), 150), 4000, lower=True)
    ek, ec, nwin = oracle_counts(oracle, reads, k, 0)
    raw = raw_kmers(oracle, reads, k)
    assert len(raw) == nwin
    rng.shuffle(raw)
    exp_files, _ = oracle.build_kmer_set([(oracle.LINE, "r", "\n".join(reads) + "\n")], k, out="ks")
    exp_files = {n[2:]: d for n, d in exp_files.items()}
    for how in ("host", "device", "pieces"):
        with g.Context(k, g.MODE_KMER_SET, hbm_budget=256 * MB) as ctx:
            a = as_words(raw, ctx.key_words)
            if how == "host":
                ctx.push_keys_host(a)
            elif how == "device":
                import torch
                t = torch.from_numpy(a.view(np.int64)).cuda()
                ctx.push_keys_device(t.data_ptr(), len(raw))
            else:          # several pushes, the runs merged at finish
                third = len(raw) // 3
                ctx.push_keys_host(a[:third])
                ctx.push_keys_host(a[third:2 * third])
                ctx.push_keys_host(a[2 * third:])
            c = ctx.finish()
            ks, cs = ctx.result()
            assert c.windows == nwin and c.keys == nwin and c.distinct == len(ek)
            assert ks == ek and [int(x) for x in cs] == ec, how
            assert ctx.emit() == exp_files, how


def test_raw_graph_keys_are_counted_as_given(oracle):
    """Graph mode: the caller pushes what ReverseComplementAdapter yields (both strands of every (k+1)-mer)."""
    rng = random.Random(77)
    for k in (27, 55):
        reads = make_reads(rng, 300, (k, 150), 3000)
        keys, _, nwin = oracle.collect([(oracle.LINE, "r", "\n".join(reads) + "\n")], k + 1, 1)
        ek, ec, _ = oracle_counts(oracle, reads, k + 1, 1)
        rng.shuffle(keys)
        exp_files, _ = oracle.build_graph([(oracle.LINE, "r", "\n".join(reads) + "\n")], k, out="gr")
        exp_files = {n[2:]: d for n, d in exp_files.items()}
        with g.Context(k, g.MODE_GRAPH, hbm_budget=256 * MB) as ctx:
            ctx.push_keys_host(as_words(keys, ctx.key_words))
            c = ctx.finish()
            ks, cs = ctx.result()
            assert c.keys == 2 * nwin and c.windows == nwin
            assert ks == ek and [int(x) for x in cs] == ec
            assert ctx.emit() == exp_files


def test_keys_and_bases_mix_and_wide_keys_are_refused(oracle):
    k = 25
    rng = random.Random(5)
    reads = make_reads(rng, 300, 120, 2500)
    ek, ec, nwin = oracle_counts(oracle, reads, k, 0)
    half = len(reads) // 2
    raw = raw_kmers(oracle, reads[half:], k)
    with g.Context(k, g.MODE_KMER_SET, hbm_budget=256 * MB) as ctx:
        ctx.push_host("\n".join(reads[:half]) + "\n")
        with pytest.raises(g.GossGpuError) as e:
            ctx.push_keys_host(np.array([raw[0], 1 << 50, raw[1]], dtype=np.uint64))     # not a 25-mer
        assert e.value.status == -1
        ctx.push_keys_host(as_words(raw, 1))
        c = ctx.finish()
        ks, cs = ctx.result()
        assert c.windows == nwin
        assert ks == ek and [int(x) for x in cs] == ec


@pytest.mark.parametrize("k,mode", [(25, 0), (45, 0), (27, 1)])
def test_raw_keys_records_and_bases_in_one_build(oracle, k, mode):
    """The three entries of the boundary mixed in one context: a third of the reads as raw un-normalised k-mers
    (goss_gpu_push_keys_host: the templated operator()(cxt, KmerSrc&), GossCmdBuildKmerSet.tcc:246-256), a third as
    super-k-mer records routed for three parts and pushed part by part (goss_gpu_push_records_device), a third as bases.
    The result is the oracle's build of all reads (graph mode: the raw keys are the (k+1)-mers and their reverse
    complements, as ReverseComplementAdapter.hh:34-55 yields them)."""
    import torch
    rng = random.Random(777 + k)
    reads = make_reads(rng, 900, (max(8, k - 3), 170 if k < 40 else 220), 6000, lower=True)
    length = k + 1 if mode else k
    text = lambda rs: ("\n".join(rs) + "\n").encode()          # noqa: E731
    build = oracle.build_graph if mode else oracle.build_kmer_set
    exp, nwin = build([(oracle.LINE, "r", text(reads))], k, out="ks")
    exp = {n[2:]: d for n, d in exp.items()}
    a, b, c3 = reads[:300], reads[300:600], reads[600:]
    raw = raw_kmers(oracle, a, length)
    if mode:
        raw = raw + [oracle.revcomp(x, length) for x in raw]
    rng.shuffle(raw)
    rb = g.binding.record_bytes(k, mode)
    with g.Context(k, mode, hbm_budget=256 * MB) as ctx:
        ctx.push_keys_host(as_words(raw, ctx.key_words))
        tb = text(b)
        bases = torch.frombuffer(bytearray(tb), dtype=torch.uint8).cuda()
        need = [1, 1, 1]
        for attempt in range(2):
            first = [sum(need[:p]) for p in range(3)]
            buf = torch.empty(sum(need) * rb, dtype=torch.uint8, device="cuda")
            recs, wins, ok = ctx.route_records(bases.data_ptr(), bases.numel(), 3, buf.data_ptr(), first, need)
            need = recs
        assert ok
        for p in range(3):
            if recs[p]:
                ctx.push_records(buf.data_ptr() + first[p] * rb, recs[p], wins[p])
        ctx.push_host(text(c3))
        cts = ctx.finish()
        got = ctx.emit()
    assert cts.windows == nwin, (cts.windows, nwin)
    assert got == exp
