"""The 32-bit-remainder form of the second partition level and of the counting stage (kernels_partition.hpp:
subpart32_kernel, kernels_count.hpp: seg_hash_reduce32_kernel; role: BackyardHash.cc:115-242 insert + count,
BlendedSort.hh:68-167 order).  One-word keys whose bits below a 17-bit prefix fit 32 -- k <= 24, and k = 25 k-mer sets,
whose strand representative has one bit that is always clear -- are written as 4-byte remainders into 131 072
sub-regions and counted there.  Files / (key, count) lists against the oracle and against the 8-byte form."""
import os
import random

import pytest

import gossamer_amd as g

pytestmark = pytest.mark.gpu


class env:
    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


STATS = ("fused_chunks", "fused_msd_chunks", "rem32_chunks", "segment_retries", "big_table_chunks", "fused_overflows", "rec_chunks")


def build(reads, k, mode, budget=8 << 30):
    with g.Context(k, mode, hbm_budget=budget) as ctx:
        ctx.push_host(reads)
        c = ctx.finish()
        files = ctx.emit()
        st = {n: ctx.stat(n) for n in STATS}
    return c, files, st


@pytest.mark.parametrize("k,mode", [(25, 0), (21, 0), (24, 0), (13, 0), (17, 0), (24, 1), (20, 1), (15, 1), (26, 0), (27, 0), (25, 1)])
def test_rem32_form_matches_the_oracle(oracle, k, mode):
    """300 k reads of 150 bp (45 M window starts: the fused path takes them by itself), ~30x coverage.  k = 25: 33 bits
    below a 17-bit prefix, the clear bit 24 squeezed out; graph k = 24: 33 bits and no clear bit -- ten bits at the second
    level leave 32; k = 26, 27 and graph k = 25: no form fits, the 8-byte form takes them; the others: narrower
    remainders."""
    reads = g.synth_reads_host(300_000, 150, 1_500_000, seed=100 + k + mode)
    build_o = oracle.build_graph if mode else oracle.build_kmer_set
    exp, nwin = build_o([(oracle.LINE, "reads", reads)], k, out="o")
    exp = {n[1:]: d for n, d in exp.items()}
    c, got, st = build(reads, k, mode)
    length = k + 1 if mode else k
    # some number of second-level bits (9 .. 12) leaves a remainder of 8 .. 32 bits (33 at 9 bits for an odd k-mer set: squeezed)
    fits = any(8 <= 2 * length - 8 - b2 and 2 * length - 8 - b2 - (1 if (not mode and length % 2 == 1 and 2 * length - 8 - b2 == 33) else 0) <= 32
               for b2 in (9, 10))
    assert c.windows == nwin
    assert st["fused_chunks"] == 1 and st["fused_msd_chunks"] == 1 and st["segment_retries"] == 0, st
    assert st["rem32_chunks"] == (1 if fits else 0), st
    assert got == exp
    if fits:
        with env(GOSS_GPU_NO_REM32=1):
            c2, got2, st2 = build(reads, k, mode)
        assert st2["rem32_chunks"] == 0 and st2["fused_msd_chunks"] == 1 and got2 == exp
        # between the two levels: remainder + digit, twelve keys to a granule (round 5) -- with the LDS layout capped at
        # what a tile without carried keys needs, so that every other tile sends its carried granules off short (the
        # path keys that spread never take); and the 8-byte keys of rounds 3-4
        for e in ({"GOSS_GPU_NARROW_CAPG": 576}, {"GOSS_GPU_NARROW": 0}):
            with env(**e):
                c3, got3, st3 = build(reads, k, mode)
            assert c3.windows == nwin and st3["rem32_chunks"] == 1 and st3["segment_retries"] == 0, (e, st3)
            assert got3 == exp, e


def test_rem32_small_inputs_every_k(oracle):
    """Small ragged reads with non-bases forced down the fused path (exact sub-region sizes): every k the form serves,
    k-mer sets and graphs, keys and counts against the oracle."""
    rng = random.Random(321)
    genome = "".join(rng.choice("ACGT") for _ in range(30000))
    reads = []
    for _ in range(40000):
        L = rng.randint(40, 150)
        p = rng.randint(0, len(genome) - L)
        r = genome[p:p + L]
        if rng.random() < 0.05:
            q = rng.randrange(L)
            r = r[:q] + "N" + r[q + 1:]
        if rng.random() < 0.5:
            r = r.lower()
        reads.append(r)
    txt = "\n".join(reads) + "\n"
    with env(GOSS_GPU_FUSED_MIN=0):
        for mode in (0, 1):
            for k in range(13, 26 if mode == 0 else 24):
                length = k + 1 if mode else k
                ek, ec, _, nwin = oracle.count([(oracle.LINE, "r", txt)], length, mode)
                for slots in (2048, 4096):
                    with env(GOSS_GPU_REM32_SLOTS=slots):
                        with g.Context(k, mode, hbm_budget=2 << 30) as ctx:
                            ctx.push_host(txt)
                            c = ctx.finish()
                            assert ctx.stat("fused_chunks") == 1 and ctx.stat("rem32_chunks") == 1, (k, mode)
                            gk, gc = ctx.result()
                    assert c.windows == nwin, (k, mode)
                    assert gk == ek, (k, mode, slots)
                    assert [int(x) for x in gc] == ec, (k, mode, slots)


def test_rem32_tables_overflow_in_turn():
    """2.2e8 distinct 25-mers with the estimate halved on purpose: 840 per 17-bit segment expected, 1 680 there -- the
    2048-slot tables (1 536) overflow and the counting alone is redone in the 4096-slot tables (the remainders are still
    in their sub-regions); with the right estimate the 4096-slot tables are taken at once.  Same keys and counts as
    the 8-byte form every time.  (Round 5's tables of four-slot buckets take the small table only up to 400 expected
    keys per segment -- GOSS_GPU_R32_SMALL_MAX puts the old threshold back so that the ladder is still walked.)"""
    import torch
    from gossamer_amd import dist as gd
    n, L, G = 6_000_000, 150, 230_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    res = []
    for e, want in (({"GOSS_GPU_NO_REM32": 1}, (0, None)), ({}, (1, 0)), ({"GOSS_GPU_EST_SCALE": 0.5, "GOSS_GPU_R32_SMALL_MAX": 1152}, (1, 1))):
        with env(GOSS_GPU_CANON_L1=0, **e):          # (strand representatives whatever the estimate says: the 9-bit form and its two tables)
            ctx = g.Context(25, g.MODE_KMER_SET, hbm_budget=24 << 30)
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=55)
            torch.cuda.synchronize()
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        assert ctx.stat("fused_chunks") == 1 and ctx.stat("rem32_chunks") == want[0], (e, ctx.stat("rem32_chunks"))
        if want[1] is not None:
            assert ctx.stat("segment_retries") == want[1], (e, ctx.stat("segment_retries"))
        kp, cp, m = ctx.result_ptrs()
        res.append((gd.device_view(kp, m, torch.int64, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone(), c.windows))
        ctx.close()
    for other in res[1:]:
        assert res[0][2] == other[2]
        assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1])
    assert int(res[0][1].to(torch.int64).sum().item()) == res[0][2]


def test_rem32_gives_way_to_the_8_byte_form(oracle):
    """Reads = one fixed 16-base prefix + 9 random bases, 1.2 M of them (the fused path wants a million keys): every
    forward 25-mer lies under ONE 17-bit prefix and is its own strand representative (262 144 distinct ones) -- both
    32-bit tables overflow, the chunk is redone in the 8-byte form, whose own ladder ends in the full sort (the way of
    rounds 1-5, still there behind the new one); by default the one segment is counted by sort and the chunk keeps its
    form.  Files against the oracle."""
    rng = random.Random(12)
    prefix = "ACGTTGCAAGCTGAGG"          # (base 12, the middle of the 25-mer, is a G: the forward strand is the representative)
    reads = [prefix + "".join(rng.choice("ACGT") for _ in range(9)) for _ in range(1_200_000)]
    txt = ("\n".join(reads) + "\n").encode()
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", txt)], 25, out="o")
    exp = {n[1:]: d for n, d in exp.items()}
    # (the ladder of rounds 1-5: GOSS_GPU_OVERFLOW_BY_SORT=0 -- it still stands behind round 6's way, below)
    with env(GOSS_GPU_FUSED_MIN=0, GOSS_GPU_EST_SCALE=0.05, GOSS_GPU_OVERFLOW_BY_SORT=0):
        c, got, st = build(txt, 25, 0, budget=2 << 30)
    assert c.windows == nwin == 1_200_000
    assert st["rem32_chunks"] == 0 and st["segment_retries"] >= 2, st
    assert got == exp
    # round 6: the one segment whose table overflowed is counted by sort, the chunk stays in the 32-bit form
    with env(GOSS_GPU_FUSED_MIN=0, GOSS_GPU_EST_SCALE=0.05):
        c, got, st = build(txt, 25, 0, budget=2 << 30)
    assert c.windows == nwin and st["rem32_chunks"] == 1 and st["segment_retries"] == 0, st
    assert got == exp


def test_rem32_skewed_low_bits(oracle):
    """Keys of one segment that agree on the bits the ordering pass bins on: hundreds of entries in one bin, the bitonic
    fallback of seg_hash_reduce32_kernel.  One fixed prefix + 5 random bases (1 024 distinct forward keys, fewer than
    a table takes), 1.1 M reads; k = 25 (squeezed remainders) and graph k = 20."""
    rng = random.Random(8)
    for k, mode, prefix in ((25, 0, "ACGTTGCAAGCTGAGGCATC"), (20, 1, "ACGTTGCAAGCTTAGG")):
        tails = ["".join(rng.choice("ACGT") for _ in range(5)) for _ in range(1_100_000)]
        txt = ("\n".join(prefix + t for t in tails) + "\n").encode()
        build_o = oracle.build_graph if mode else oracle.build_kmer_set
        exp, nwin = build_o([(oracle.LINE, "reads", txt)], k, out="o")
        exp = {n[1:]: d for n, d in exp.items()}
        with env(GOSS_GPU_FUSED_MIN=0):
            c, got, st = build(txt, k, mode, budget=2 << 30)
        assert c.windows == nwin == 1_100_000
        assert st["rem32_chunks"] == 1 and st["segment_retries"] == 0, st
        assert got == exp


@pytest.mark.parametrize("bits,split", [(10, 0), (9, 1), (9, 4), (10, 4)])          # (two and three third-level bits: tests/fuzz_parity.py draws them)
def test_rem32_wider_second_level_and_third_level(oracle, bits, split):
    """Ten bits at the second level, and a third level inside the segments (1 .. 4 bits: what reads with many distinct
    k-mers get by themselves), forced on the same reads: k = 25 (squeezed at 9 bits, 32-bit remainders at 10), k = 21,
    graph k = 24 (needs 10), graph k = 20; both tables.  Files against the oracle."""
    for k, mode in ((25, 0), (21, 0), (24, 1), (20, 1)):
        if bits == 9 and (k, mode) == (24, 1):
            continue
        reads = g.synth_reads_host(300_000, 150, 1_500_000, seed=300 + k + mode)
        build_o = oracle.build_graph if mode else oracle.build_kmer_set
        exp, nwin = build_o([(oracle.LINE, "reads", reads)], k, out="o")
        exp = {n[1:]: d for n, d in exp.items()}
        for slots in (2048, 4096):
            with env(GOSS_GPU_REM32_BITS=bits, GOSS_GPU_REM32_SPLIT=split, GOSS_GPU_REM32_SLOTS=slots):
                with g.Context(k, mode, hbm_budget=8 << 30) as ctx:
                    ctx.push_host(reads)
                    c = ctx.finish()
                    got = ctx.emit()
                    st = {n: ctx.stat(n) for n in STATS + ("rem32_bits", "rem32_split")}
            assert c.windows == nwin
            assert st["rem32_chunks"] == 1 and st["rem32_bits"] == bits and st["rem32_split"] == split and st["segment_retries"] == 0, st
            assert got == exp, (k, mode, slots)


def test_rem32_big_tables_before_a_third_level():
    """Reads with many distinct k-mers (sequencing errors): where ten bits at the second level and the 4096-slot table do
    not hold a segment's distinct keys, the tables of 8192 and 16384 slots (512 / 1024 threads a workgroup) are taken
    before a third partition level (the sizing itself is what tests/test_gpu_fullsize.py's C2 with 1 % errors goes
    through: 2e9 distinct keys, ten bits, no third level).  Here the kernels: 2.2e8 distinct 25-mers in 2^17 and 2^18
    segments with either table forced, and the largest tables on small inputs of every shape; same keys and counts as
    the 8-byte form, no retry."""
    import torch
    from gossamer_amd import dist as gd
    n, L, G = 6_000_000, 150, 230_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    res = []
    for e, want in (({"GOSS_GPU_NO_REM32": 1}, None), ({"GOSS_GPU_REM32_BITS": 10, "GOSS_GPU_REM32_SLOTS": 8192}, (10, 0)),
                    ({"GOSS_GPU_REM32_SLOTS": 8192}, (9, 0)), ({"GOSS_GPU_REM32_BITS": 10, "GOSS_GPU_REM32_SLOTS": 16384}, (10, 0))):
        with env(GOSS_GPU_CANON_L1=0, **e):
            ctx = g.Context(25, g.MODE_KMER_SET, hbm_budget=24 << 30)
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=55)
            torch.cuda.synchronize()
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        if want:
            assert (ctx.stat("rem32_chunks"), ctx.stat("rem32_bits"), ctx.stat("rem32_split"), ctx.stat("segment_retries")) == (1,) + want + (0,), e
        kp, cp, m = ctx.result_ptrs()
        res.append((gd.device_view(kp, m, torch.int64, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone(), c.windows))
        ctx.close()
    for other in res[1:]:
        assert res[0][2] == other[2]
        assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1])
    del buf, res
    for k, mode in ((25, 0), (21, 0), (24, 1)):
        reads = g.synth_reads_host(200_000, 150, 1_000_000, seed=900 + k + mode)
        want = None
        for e in ({"GOSS_GPU_NO_REM32": 1}, {"GOSS_GPU_REM32_SLOTS": 8192}, {"GOSS_GPU_REM32_SLOTS": 16384}):
            with env(GOSS_GPU_FUSED_MIN=0, **e):
                with g.Context(k, mode, hbm_budget=8 << 30) as ctx:
                    ctx.push_host(reads)
                    ctx.finish()
                    got = ctx.emit()
                    assert ctx.stat("rem32_chunks") == (0 if "GOSS_GPU_NO_REM32" in e else 1), (k, mode, e)
            want = want or got
            assert got == want, (k, mode, e)


def test_rem32_takes_a_third_level_when_the_tables_overflow():
    """2.2e8 distinct 25-mers with the estimate at a quarter and the small table forced: 420 per 17-bit segment expected,
    1 680 there -- the 2048-slot tables overflow and the chunk is redone with every segment split in two (840 each).  With
    the right estimate and free choice: the 4096-slot tables at once; at 0.6 of the estimate ... the same.  Same keys
    and counts as the 8-byte form."""
    import torch
    from gossamer_amd import dist as gd
    n, L, G = 6_000_000, 150, 230_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    res = []
    for e, split in (({"GOSS_GPU_NO_REM32": 1}, 0), ({"GOSS_GPU_EST_SCALE": 0.25, "GOSS_GPU_REM32_SLOTS": 2048}, 1), ({"GOSS_GPU_REM32_SPLIT": 2}, 2)):
        with env(GOSS_GPU_CANON_L1=0, **e):
            ctx = g.Context(25, g.MODE_KMER_SET, hbm_budget=24 << 30)
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=55)
            torch.cuda.synchronize()
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        assert ctx.stat("fused_chunks") == 1 and ctx.stat("rem32_split") == split, (e, ctx.stat("rem32_split"), ctx.stat("segment_retries"))
        assert ctx.stat("rem32_chunks") == (1 if split else 0)
        kp, cp, m = ctx.result_ptrs()
        res.append((gd.device_view(kp, m, torch.int64, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone(), c.windows))
        ctx.close()
    for other in res[1:]:
        assert res[0][2] == other[2]
        assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1])


@pytest.mark.parametrize("k", [25, 22, 17, 31, 27])
def test_first_level_computes_the_canonical_form_itself(oracle, k):
    """GOSS_GPU_CANON_L1=2: the fused first level stores gossamer's canonical form (the strand with the smaller FNV-1a
    hash, RankSelect.hh:126-140) instead of the strand representative -- what chunks with more than 10 % distinct keys
    get by themselves, whose re-ordering after counting would cost more than hashing every window.  The run needs no
    canonical re-ordering (rep_chunks == 0); k = 25 then takes ten second-level bits (no bit to squeeze out).  Files
    against the oracle; k = 31 and 27: the 8-byte forms behind the same first level."""
    reads = g.synth_reads_host(300_000, 150, 1_500_000, seed=500 + k)
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], k, out="o")
    exp = {n[1:]: d for n, d in exp.items()}
    with env(GOSS_GPU_CANON_L1=2):
        with g.Context(k, 0, hbm_budget=8 << 30) as ctx:
            ctx.push_host(reads)
            c = ctx.finish()
            got = ctx.emit()
            st = {n: ctx.stat(n) for n in STATS + ("rem32_bits", "canon_chunks", "rep_chunks")}
    assert c.windows == nwin
    assert st["fused_chunks"] == 1 and st["canon_chunks"] == 1 and st["rep_chunks"] == 0, st
    if k == 25:
        assert st["rem32_chunks"] == 1 and st["rem32_bits"] == 10, st
    assert got == exp
