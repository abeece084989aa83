"""BASELINE config C2 at full size (k = 25, 100 M x 150 bp reads, 12.58 G k-mers) on one GPU:
size-independent properties where the oracle cannot go.

* the default pipeline (fused extraction, two-level partition by atomic cursors) and the unfused
  one (dense extraction, two partition passes with look-back) are independent code paths down to
  the counting kernel: they must give identical keys and counts;
* counts add up to the number of windows, keys are strictly increasing;
* the emitted KmerSet passes its own index check: select / rank / access of all 10^8 elements
  through the -d1 / -d0 DenseSelect images (goss_gpu_check_index), and its headers are consistent.
"""
import os
import struct

import pytest

import gossamer_amd as g

pytestmark = pytest.mark.gpu


def test_c2_full_size_properties():
    import torch
    from gossamer_amd import dist as gd
    free_b, total_b = torch.cuda.mem_get_info(0)
    if total_b < 250 * (1 << 30):
        pytest.skip("needs the 288 GB of an MI355X")
    n, L, G = 100_000_000, 150, 100_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    budget = int((free_b - buf.numel()) * 0.94)          # bench.py's share: one chunk
    res = []
    for env in ({}, {"GOSS_GPU_NO_FUSED": "1"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = g.Context(25, g.MODE_KMER_SET, hbm_budget=budget)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=1)
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        fused = ctx.stat("fused_msd_chunks")
        assert fused == 0 if env else fused >= 1
        kp, cp, m = ctx.result_ptrs()
        keys = gd.device_view(kp, m, torch.int64, "cuda").clone()
        counts = gd.device_view(cp, m, torch.int32, "cuda").clone()
        assert m == c.distinct and c.keys == c.windows
        if not res:
            assert bool((keys[1:] > keys[:-1]).all().item())
            assert int(counts.to(torch.int64).sum().item()) == c.windows
            # windows: every read has 126, minus the ones an 'N' removes (one N in every 97th read)
            assert n * 126 * 0.99 < c.windows <= n * 126
            files = ctx.emit()
            hdr = struct.unpack("<QQQ", files[".header"])
            assert hdr == (2011101701, 25, m)
            sa = struct.unpack("<8Q", files[".kmers.header"])
            assert sa[0] == 2012030501 and sa[7] == m and sa[5] == 4 ** 25 and sa[6] == 0
            assert sa[1] == 23 and sa[2] == 24                      # D, quantizedD of SURVEY section 8(a13) for M ~ 10^8
            rep = ctx.check_index({k[len(".kmers"):]: v for k, v in files.items() if k.startswith(".kmers")})
            assert rep["select"] == 0 and rep["rank"] == 0 and rep["access"] == 0 and rep["failures"] == 0, rep
        res.append((keys, counts, c.windows))
        ctx.close()
    assert res[0][2] == res[1][2]
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_c2_with_one_percent_errors_full_size_properties():
    """C2's reads with 1 % of the bases substituted at random (what sequencers produce: every error makes up to k new
    k-mers, 2.0e9 distinct 25-mers instead of 1.0e8).  The default pipeline then takes forms the clean reads never see --
    the first level computes gossamer's canonical form itself, ten bits at the second level, 16384-slot tables of 32-bit
    remainders (round 5; a third level inside the segments and 4096-slot tables before) -- and must give, key for key and count for count, what the 8-byte
    forms behind a first level of strand representatives give (three partition digits, the canonical re-ordering of
    2e9 pairs): no kernel in common between the counting stages.  Counts add up to the windows, keys increase."""
    import torch
    from gossamer_amd import dist as gd
    free_b, total_b = torch.cuda.mem_get_info(0)
    if total_b < 250 * (1 << 30):
        pytest.skip("needs the 288 GB of an MI355X")
    n, L, G = 100_000_000, 150, 100_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    budget = int((free_b - buf.numel() - (60 << 30)) * 0.94)          # (the two results, 24 GB each, live beside the arena)
    res = []
    for env in ({}, {"GOSS_GPU_NO_REM32": "1", "GOSS_GPU_CANON_L1": "0"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = g.Context(25, g.MODE_KMER_SET, hbm_budget=budget)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=1)
            torch.cuda.synchronize()
            gen = torch.Generator(device="cuda")
            gen.manual_seed(7)
            lut = torch.tensor([ord(ch) for ch in "ACGT"], dtype=torch.uint8, device="cuda")
            step = 1 << 28
            for at in range(0, buf.numel(), step):
                v = buf[at:at + step]
                hit = (torch.rand(v.numel(), device="cuda", generator=gen) < 0.01) & (v != 10)
                sub = lut[torch.randint(0, 4, (v.numel(),), device="cuda", generator=gen)]
                v[hit] = sub[hit]
                del hit, sub
            torch.cuda.synchronize()
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        if not env:
            assert ctx.stat("rem32_chunks") >= 1 and ctx.stat("canon_chunks") >= 1 and ctx.stat("rem32_bits") == 10 and ctx.stat("rem32_split") == 0, \
                {s: ctx.stat(s) for s in ("rem32_chunks", "canon_chunks", "rem32_split", "rem32_bits", "fused_chunks")}
        else:
            assert ctx.stat("rem32_chunks") == 0 and ctx.stat("canon_chunks") == 0
        kp, cp, m = ctx.result_ptrs()
        keys = gd.device_view(kp, m, torch.int64, "cuda").clone()
        counts = gd.device_view(cp, m, torch.int32, "cuda").clone()
        assert m == c.distinct and c.keys == c.windows and 1.8e9 < m < 2.3e9
        if not res:
            assert bool((keys[1:] > keys[:-1]).all().item())
            assert int(counts.to(torch.int64).sum().item()) == c.windows
        res.append((keys, counts, c.windows))
        ctx.close()
    assert res[0][2] == res[1][2]
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_c4_full_size_properties():
    """BASELINE config C4 at full size: build-graph k = 55 on 200 M x 150 bp reads (19 G windows,
    38 G 112-bit keys, several chunks + merge).  The edge set of a graph built from reads is
    closed under reverse complement with equal multiplicities -- checked for all 2e8 edges by the
    device-side lint pass -- multiplicities add up to twice the number of windows, edges are
    strictly increasing, and the emitted edge SparseArray passes its own select / rank check."""
    import torch
    from gossamer_amd import dist as gd
    free_b, total_b = torch.cuda.mem_get_info(0)
    if total_b < 250 * (1 << 30):
        pytest.skip("needs the 288 GB of an MI355X")
    n, L, G = 200_000_000, 150, 100_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    budget = int((free_b - buf.numel()) * 0.90)
    with g.Context(55, g.MODE_GRAPH, hbm_budget=budget) as ctx:
        ctx.synth_reads(buf.data_ptr(), n, L, G, seed=1)
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        assert c.key_words == 2 and c.keys == 2 * c.windows
        assert n * 95 * 0.98 < c.windows <= n * 95
        kp, cp, m = ctx.result_ptrs()
        assert m == c.distinct and m % 2 == 0          # edges come in reverse-complement pairs (a palindromic 56-mer has probability 4^-28)
        counts = gd.device_view(cp, m, torch.int32, "cuda")
        assert int(counts.to(torch.int64).sum().item()) == c.keys
        rep = ctx.lint()
        assert rep == {"missing_rc": 0, "count_mismatch": 0, "zero_count": 0, "order_violation": 0}, rep
        files = ctx.emit()
        assert struct.unpack("<QQQ", files[".header"]) == (2011101014, 55, 0)
        sa = struct.unpack("<8Q", files["-edges.header"])
        assert sa[7] == m and sa[1] == 84 and sa[2] == 88            # D = 84, qD = 88
        irep = ctx.check_index({k[len("-edges"):]: v for k, v in files.items() if k.startswith("-edges")})
        assert irep["select"] == 0 and irep["rank"] == 0 and irep["access"] == 0 and irep["failures"] == 0, irep
        hist = files["-counts-hist.txt"].decode().split("\n")
        assert sum(int(l.split("\t")[1]) for l in hist if l) == m


def test_c5_full_size_properties():
    """BASELINE config C5 at full size on one GPU: two k-mer sets of 50 M x 150 bp reads each (genomes
    sharing half of their sequence), intersect / subtract through the library's set algebra on weighted
    runs (what the commands and the multi-GPU path do range by range).  Size-independent properties:
    |A n B| + |A \\ B| = |A|, |A n B| + |B \\ A| = |B|, commutativity of the intersection (same keys), the
    three parts are disjoint and their union is the merge, and the emitted intersection passes its own
    index check."""
    import torch
    from gossamer_amd import dist as gd
    free_b, total_b = torch.cuda.mem_get_info(0)
    if total_b < 250 * (1 << 30):
        pytest.skip("needs the 288 GB of an MI355X")
    n, L, G = 50_000_000, 150, 100_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    budget = int((free_b - buf.numel()) * 0.80)
    sets = []
    with g.Context(25, g.MODE_KMER_SET, hbm_budget=budget) as ctx:
        # two "genomes" of 200 Mbp that share one half: A = reads of sequence 1 and of sequence 3, B = reads of
        # sequence 1 (other reads) and of sequence 2 -- the generator's sequences of different seeds are unrelated
        half = n // 2
        for seeds, first in (((1, 3), 0), ((1, 2), half)):
            ctx.reset()
            ctx.synth_reads(buf.data_ptr(), half, L, G, seed=seeds[0], first_read=first)
            ctx.synth_reads(buf.data_ptr() + half * (L + 1), n - half, L, G, seed=seeds[1], first_read=0)
            torch.cuda.synchronize()
            ctx.push_device(buf.data_ptr(), buf.numel())
            c = ctx.finish()
            kp, cp, m = ctx.result_ptrs()
            sets.append(gd.device_view(kp, m, torch.int64, "cuda").clone())
            assert bool((sets[-1][1:] > sets[-1][:-1]).all().item())
        a, b = sets
        na, nb = int(a.numel()), int(b.numel())
        assert 190_000_000 < na < 200_000_001 and 190_000_000 < nb < 200_000_001

        def algebra(x, wx, y, wy, lo, hi):
            ctx.reset()
            ones = torch.full((max(x.numel(), y.numel()),), 1, dtype=torch.int32, device="cuda")
            twos = ones * 2
            torch.cuda.synchronize()
            ctx.push_run(x.data_ptr(), (ones if wx == 1 else twos).data_ptr(), x.numel())
            ctx.push_run(y.data_ptr(), (ones if wy == 1 else twos).data_ptr(), y.numel())
            ctx.finish()
            ctx.select_counts(lo, hi)
            kp, cp, m = ctx.result_ptrs()
            return gd.device_view(kp, m, torch.int64, "cuda").clone()

        ab = algebra(a, 1, b, 1, 2, 2)            # intersect: count == number of sets
        ba = algebra(b, 1, a, 1, 2, 2)
        a_b = algebra(a, 1, b, 2, 1, 1)           # subtract: weights 1 and 2, keep count == 1
        b_a = algebra(b, 1, a, 2, 1, 1)
        union = algebra(a, 1, b, 1, 1, 2)
        assert torch.equal(ab, ba)
        assert 90_000_000 < ab.numel() < 100_000_001          # the shared sequence
        assert ab.numel() + a_b.numel() == na and ab.numel() + b_a.numel() == nb
        assert union.numel() == na + nb - ab.numel()
        # the intersection is inside both sets (membership by binary search), the differences are not in the other
        assert bool((a[torch.searchsorted(a, ab).clamp(max=na - 1)] == ab).all().item())
        assert bool((b[torch.searchsorted(b, ab).clamp(max=nb - 1)] == ab).all().item())
        assert not bool((b[torch.searchsorted(b, a_b).clamp(max=nb - 1)] == a_b).any().item())
        # emit the intersection and let the device reader walk its own index
        ctx.reset()
        ones = torch.ones(ab.numel(), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.push_run(ab.data_ptr(), ones.data_ptr(), ab.numel())
        ctx.finish()
        files = ctx.emit()
        assert struct.unpack("<QQQ", files[".header"]) == (2011101701, 25, ab.numel())
        rep = ctx.check_index({k[len(".kmers"):]: v for k, v in files.items() if k.startswith(".kmers")})
        assert rep["select"] == 0 and rep["rank"] == 0 and rep["access"] == 0 and rep["failures"] == 0, rep



@pytest.mark.parametrize("k,mode,n", [(25, 0, 20_000_000), (55, 1, 15_000_000)])
def test_reads_with_homopolymer_tails_at_scale(k, mode, n):
    """What a random genome never holds and real reads hold everywhere: poly-A / poly-T stretches (tails of 30 to 80
    bases on a sixth of the reads, planted on the device).  Their keys are the all-zeros and all-ones patterns that pads
    and empty-slot markers are made of -- round 6's fuzz runs found two places where a key of all ones was taken for a
    pad (the strand-pair expansion's sort, the two-word counting table's ordering).  At a size the fused kernels take by
    themselves: the default pipeline and the unfused one (look-back partition passes, no kernel in common before the
    counting stage; for graphs both strands extracted instead of strand pairs) must agree key for key and count for
    count, counts add up to the adapter's key stream, and the keys of the stretches are there: A..A and T..T with equal
    counts (graph: an edge and its reverse complement; k-mer set: one canonical form of the two)."""
    import torch
    from gossamer_amd import dist as gd
    free_b, total_b = torch.cuda.mem_get_info(0)
    if total_b < 250 * (1 << 30):
        pytest.skip("needs the 288 GB of an MI355X")
    L, G = 150, 20_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    budget = min(int((free_b - buf.numel()) * 0.9), 160 << 30)
    res = []
    length = k + 1 if mode else k
    for env in ({}, {"GOSS_GPU_NO_FUSED": "1"}):
        old = {a: os.environ.get(a) for a in env}
        os.environ.update(env)
        try:
            ctx = g.Context(k, mode, hbm_budget=budget)
        finally:
            for a, v in old.items():
                if v is None:
                    del os.environ[a]
                else:
                    os.environ[a] = v
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=3)
            torch.cuda.synchronize()
            # tails: read r (every sixth) ends in t(r) = 30 .. 80 copies of 'A' (even r / 6) or 'T' (odd)
            rows = buf.view(n, L + 1)
            gen = torch.Generator(device="cuda")
            gen.manual_seed(11)
            step = 6_000_000
            col = torch.arange(L, device="cuda")
            for at in range(0, n, step):
                part = rows[at:at + step:6]
                m = part.shape[0]
                tlen = torch.randint(30, 81, (m, 1), device="cuda", generator=gen)
                letter = torch.where((torch.arange(m, device="cuda") % 2 == 0).view(m, 1), torch.tensor(65, device="cuda"), torch.tensor(84, device="cuda")).to(torch.uint8)
                mask = col.view(1, L) >= (L - tlen)
                body = part[:, :L]
                body[mask] = letter.expand(m, L)[mask]
                del tlen, letter, mask, body
            torch.cuda.synchronize()
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        if not env:
            # (the fused path took the chunk, and the few segments that hold the stretches' families -- more distinct keys
            # than a table takes -- were counted by sort instead of sending the whole chunk up the ladder of forms)
            assert ctx.stat("fused_chunks") >= 1 and ctx.stat("overflow_units") >= 1, {a: ctx.stat(a) for a in ("fused_chunks", "overflow_units", "segment_retries")}
        else:
            assert ctx.stat("fused_chunks") == 0
        kp, cp, m = ctx.result_ptrs()
        words = c.key_words
        keys = gd.key_view(kp, m, words, "cuda").clone()
        counts = gd.device_view(cp, m, torch.int32, "cuda").clone()
        assert m == c.distinct
        if not res:
            assert int(counts.to(torch.int64).sum().item()) == c.keys == c.windows * (2 if mode else 1)
            # the stretches' own keys: A..A = 0 is the first key, T..T = all ones of 2 * length bits the last
            first = keys[0]
            last = keys[-1]
            bits = 2 * length
            if words == 1:
                a_key, t_key = int(first.item()), int(last.item()) & ((1 << 64) - 1)
            else:
                lo0, hi0 = (int(x) & ((1 << 64) - 1) for x in first.tolist())
                lo1, hi1 = (int(x) & ((1 << 64) - 1) for x in last.tolist())
                a_key, t_key = lo0 | (hi0 << 64), lo1 | (hi1 << 64)
            has_a, has_t = a_key == 0, t_key == (1 << bits) - 1
            if mode:
                assert has_a and has_t, (hex(a_key), hex(t_key))
                assert int(counts[0].item()) == int(counts[-1].item()) > 1_000_000
            else:
                # one canonical form of the pair (A..A, T..T) -- whichever the FNV order picks -- holds both strands' windows
                assert has_a != has_t, (hex(a_key), hex(t_key))
                assert int((counts[0] if has_a else counts[-1]).item()) > 1_000_000
        res.append((keys, counts, c.windows))
        ctx.close()
    assert res[0][2] == res[1][2]
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
