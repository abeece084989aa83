"""The extraction that partitions (extract1_part_kernel + gapped first look-back pass): same files
as the oracle when it runs, when a bucket region overflows and the chunk is redone unfused, and
when it declines the input."""
import os
import random

import pytest

import gossamer_amd as g

pytestmark = pytest.mark.gpu


def _suffix_map(files, prefix):
    return {k[len(prefix):]: v for k, v in files.items()}


def _build(reads, k, env=None, budget=8 << 30, mode=None):
    old = {}
    for name, val in (env or {}).items():
        old[name] = os.environ.get(name)
        os.environ[name] = val
    try:
        with g.Context(k, g.MODE_KMER_SET if mode is None else mode, hbm_budget=budget) as ctx:
            ctx.push_host(reads)
            c = ctx.finish()
            files = ctx.emit()
            stats = {n: ctx.stat(n) for n in ("fused_chunks", "fused_msd_chunks", "fused_overflows", "segment_retries",
                                              "lookback_failures", "overflow_units")}
        return c, files, stats
    finally:
        for name, val in old.items():
            if val is None:
                del os.environ[name]
            else:
                os.environ[name] = val


def _same(got, exp):
    assert sorted(got) == sorted(exp)
    for name in exp:
        assert got[name] == exp[name], name


def test_fused_path_runs_and_matches_oracle(oracle):
    reads = g.synth_reads_host(300000, 150, 1500000, seed=11)          # 45 M window starts, ~30x coverage
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], 25, out="ks")
    exp = _suffix_map(exp, "ks")
    c, got, st = _build(reads, 25)
    assert st["fused_chunks"] == 1 and st["fused_overflows"] == 0 and st["lookback_failures"] == 0
    assert c.windows == nwin
    _same(got, exp)
    # regions made too small on purpose: the chunk must be redone by the unfused kernels
    c, got, st = _build(reads, 25, env={"GOSS_GPU_FUSED_CAPSCALE": "0.6"})
    assert st["fused_chunks"] == 0 and st["fused_overflows"] == 1
    assert c.windows == nwin
    _same(got, exp)
    # switched off
    c, got, st = _build(reads, 25, env={"GOSS_GPU_NO_FUSED": "1"})
    assert st["fused_chunks"] == 0 and st["fused_overflows"] == 0
    _same(got, exp)
    # the 64-bit form of the window arithmetic (the 32-bit form is what the other runs took: k = 25, msd)
    c, got, st = _build(reads, 25, env={"GOSS_GPU_NO_FAST32": "1"})
    assert st["fused_chunks"] == 1 and st["fused_overflows"] == 0
    assert c.windows == nwin
    _same(got, exp)


def test_chunk_runs_stay_in_representative_space_until_they_must_not(oracle):
    """Several fused chunks: their runs hold strand representatives and are merged as such; gossamer's
    canonical form and order are applied once, on the merged run, at finish (by 16-bit or 20-bit groups).  A run
    handed in from outside (canonical keys) forces the mapping before the merge.  Files / (key, count) lists
    against the oracle."""
    import torch
    reads = g.synth_reads_host(300000, 150, 1500000, seed=23)          # 45 M window starts, 1.5 M distinct 25-mers
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], 25, out="ks")
    exp = _suffix_map(exp, "ks")
    for bits in ("16", "20"):
        old = {n: os.environ.get(n) for n in ("GOSS_GPU_FUSED_MIN", "GOSS_GPU_ORDER_BITS", "GOSS_GPU_CANON_L1")}
        os.environ["GOSS_GPU_FUSED_MIN"] = "0"
        os.environ["GOSS_GPU_ORDER_BITS"] = bits
        os.environ["GOSS_GPU_CANON_L1"] = "0"          # (small chunks: more than 10 % of a chunk's keys are distinct, which would make the first level hash)
        try:
            with g.Context(25, g.MODE_KMER_SET, hbm_budget=400 << 20) as ctx:
                ctx.push_host(reads)
                c = ctx.finish()
                assert ctx.stat("rep_chunks") >= 3 and ctx.stat("fused_chunks") == ctx.stat("rep_chunks"), ctx.stat("rep_chunks")
                assert c.windows == nwin
                _same(ctx.emit(), exp)
                keys, counts = ctx.result()
            # the same build with a canonical run pushed between the chunks' runs and the merge
            extra = {keys[0]: 5, keys[len(keys) // 2]: 7, keys[-1] + 1: 9}
            ek = torch.tensor(sorted(extra), dtype=torch.int64, device="cuda")
            ec = torch.tensor([extra[x] for x in sorted(extra)], dtype=torch.int32, device="cuda")
            with g.Context(25, g.MODE_KMER_SET, hbm_budget=400 << 20) as ctx:
                ctx.push_host(reads)
                ctx.push_run(ek.data_ptr(), ec.data_ptr(), len(extra))
                ctx.finish()
                keys2, counts2 = ctx.result()
            want = dict(zip(keys, (int(x) for x in counts)))
            for x, n in extra.items():
                want[x] = want.get(x, 0) + n
            assert keys2 == sorted(want)
            assert [int(x) for x in counts2] == [want[x] for x in keys2]
        finally:
            for n, v in old.items():
                if v is None:
                    os.environ.pop(n, None)
                else:
                    os.environ[n] = v


@pytest.mark.parametrize("k", [11, 15, 19, 23, 27, 31, 35, 47, 13, 26])
def test_graph_strand_pairs_keep_the_all_t_edge_beside_palindromes(oracle, k):
    """Graph builds count one strand of every edge and write the other afterwards (graph_expand_kernel); a palindromic
    edge has no other strand and leaves a pad in its slot, and the pads are sorted behind the keys and cut off.  Where
    the edge's 2 (k + 1) bits fill whole bytes -- k + 1 = 12, 16, 20, 24, 28, 32, 36, 48 -- the edge T..T has every digit a pad
    has: a sort on the key's digits alone left them in input order and the cut dropped T..T for a pad whenever a
    palindrome stood in front of it (found by tests/fuzz_parity.py in round 6: reads of a real genome hold poly-A
    everywhere).  Reads with poly-A, poly-T and palindromic stretches through the fused path; files against the oracle
    (ReverseComplementAdapter.hh:20-93: both strands of every edge)."""
    rng = random.Random(7 * k)
    L = k + 1
    genome = "".join(rng.choice("ACGT") for _ in range(30000))
    half = "".join(rng.choice("ACGT") for _ in range(L // 2))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    pal = half + "".join(comp[c] for c in reversed(half))          # (even L: its own reverse complement)
    special = ["A" * (L + 20), "T" * (L + 7), "ACGT" * 40, pal + "ACGTTGCA", "AT" * 60, "C" * (L + 3), "G" * (L + 1)]
    reads = []
    for i in range(30000):
        n = rng.randint(max(40, L), 150)
        p = rng.randint(0, len(genome) - n)
        reads.append(genome[p:p + n])
        if i % 600 == 0:
            reads.extend(rng.sample(special, 3))
    txt = ("\n".join(reads) + "\n").encode()
    exp, nwin = oracle.build_graph([(oracle.LINE, "r", txt)], k, out="gr")
    exp = _suffix_map(exp, "gr")
    old = os.environ.get("GOSS_GPU_FUSED_MIN")
    os.environ["GOSS_GPU_FUSED_MIN"] = "0"
    try:
        with g.Context(k, g.MODE_GRAPH, hbm_budget=2 << 30) as ctx:
            ctx.push_host(txt)
            c = ctx.finish()
            assert ctx.stat("fused_chunks") >= 1 and ctx.stat("rep_chunks") >= 1, k          # (the strand-pair form was taken)
            assert c.windows == nwin
            _same(ctx.emit(), exp)
    finally:
        if old is None:
            os.environ.pop("GOSS_GPU_FUSED_MIN", None)
        else:
            os.environ["GOSS_GPU_FUSED_MIN"] = old


def test_two_word_keys_that_end_in_48_ts(oracle):
    """The counting table of two-word keys (seg_hash_reduce96_body: 96-bit remainders; also the merge of the chunks' runs
    through the same table) orders its occupied slots before it hands them out; clustered remainders go through a bitonic
    network padded with entries of all ones and count 0 -- the very words of a key whose low 96 bits are all ones, an
    edge that ends in 48 T's.  Reads of a real genome hold such stretches everywhere (poly-A tails); round 6's fuzz run
    found the pad handed out in the key's place: the key with the count 0.  Reads with a few long homopolymers -- the
    segments A..A / T..T then hold the edges A^j T^(56-j) and every T..T + Y where a stretch ends: the cluster -- and 0.2 %
    errors, k = 55 graph, counted in two chunks whose runs are merged through the table; files against the oracle.
    (The library without the tie-break returns seven such edges with the count 0 on these reads.)"""
    rng = random.Random(55)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    genome = list("".join(rng.choice("ACGT") for _ in range(60000)))
    for _ in range(6):
        n = rng.randint(60, 140)
        piece = rng.choice("AT") * n if rng.random() < 0.7 else "A" * (n // 2) + "T" * (n - n // 2)
        at = rng.randint(0, len(genome) - n)
        genome[at:at + n] = piece
    genome = "".join(genome)
    reads = []
    for _ in range(120000):
        p = rng.randint(0, len(genome) - 101)
        r = list(genome[p:p + rng.randint(100, 101)])
        for i in range(len(r)):
            if rng.random() < 0.002:
                r[i] = rng.choice("ACGT")
        r = "".join(r)
        reads.append(r if rng.random() < 0.5 else "".join(comp[c] for c in reversed(r)))
    txt = ("\n".join(reads) + "\n").encode()
    exp, nwin = oracle.build_graph([(oracle.LINE, "r", txt)], 55, out="gr")
    exp = _suffix_map(exp, "gr")
    env = {"GOSS_GPU_FUSED_MIN": "0", "GOSS_GPU_NO_FUSED": "1"}
    old = {n: os.environ.get(n) for n in env}
    os.environ.update(env)
    try:
        with g.Context(55, g.MODE_GRAPH, hbm_budget=1 << 30) as ctx:
            ctx.push_host(txt)
            c = ctx.finish()
            assert ctx.stat("hash_merges") >= 1, {n: ctx.stat(n) for n in ("hash_merges", "seg_merges", "runs")}
            assert c.windows == nwin
            _same(ctx.emit(), exp)
    finally:
        for n, v in old.items():
            if v is None:
                os.environ.pop(n, None)
            else:
                os.environ[n] = v


@pytest.mark.parametrize("hint", [False, True])
def test_chunks_of_one_build_choose_their_key_space_for_the_build(oracle, hint):
    """A build counted in several chunks merges their runs before it re-orders them, so the key space its chunks count in
    is chosen for the WHOLE build: every chunk of these reads holds all 1.5 M k-mers of the genome -- a tenth of its own
    windows, which alone would read "canonical forms in the first level" -- and a thirtieth of the build's.  With the
    caller's word on the input's size (goss_gpu_expect_bases) and without it (more input is known to follow the first
    chunks): all chunks in representative space, one re-ordering at finish, files the oracle's.  And the other way
    round: reads whose k-mers occur once stay in canonical space whatever follows."""
    reads = g.synth_reads_host(300000, 150, 1500000, seed=23)          # 45 M window starts, 1.5 M distinct 25-mers
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], 25, out="ks")
    exp = _suffix_map(exp, "ks")
    old = os.environ.get("GOSS_GPU_FUSED_MIN")
    os.environ["GOSS_GPU_FUSED_MIN"] = "0"
    try:
        with g.Context(25, g.MODE_KMER_SET, hbm_budget=400 << 20) as ctx:
            if hint:
                ctx.expect_bases(len(reads))
            ctx.push_host(reads)
            c = ctx.finish()
            st = {n: ctx.stat(n) for n in ("rep_chunks", "canon_chunks", "fused_chunks")}
            assert st["rep_chunks"] >= 3 and st["fused_chunks"] == st["rep_chunks"] and st["canon_chunks"] == 0, st
            assert c.windows == nwin
            _same(ctx.emit(), exp)
        # a genome read once (every k-mer of multiplicity one): canonical forms, with or without more to follow
        unique = g.synth_reads_host(60000, 150, 40_000_000, seed=29)
        exp1, nwin1 = oracle.build_kmer_set([(oracle.LINE, "reads", unique)], 25, out="ks")
        with g.Context(25, g.MODE_KMER_SET, hbm_budget=400 << 20) as ctx:
            if hint:
                ctx.expect_bases(len(unique))
            ctx.push_host(unique)
            c = ctx.finish()
            assert ctx.stat("rep_chunks") == 0, {n: ctx.stat(n) for n in ("rep_chunks", "canon_chunks", "fused_chunks")}
            assert c.windows == nwin1
            _same(ctx.emit(), _suffix_map(exp1, "ks"))
    finally:
        if old is None:
            os.environ.pop("GOSS_GPU_FUSED_MIN", None)
        else:
            os.environ["GOSS_GPU_FUSED_MIN"] = old


def test_reads_with_sequencing_errors_need_no_retry(oracle):
    """Reads with 1 % substituted bases: the k-mers of the genome occur ~30 times, the ~25 k-mers around every
    error once -- 5 times more distinct keys than the genome has.  An estimate that takes all keys for equally
    frequent (the birthday estimate on a sample) sees the genome only and sends the chunk through counting
    tables that overflow; the spectrum estimate (singletons beyond what the frequent keys explain) must pick a
    form that holds them at the first attempt.  Files against the oracle."""
    import random
    base = g.synth_reads_host(300000, 150, 1500000, seed=41).decode()
    rng = random.Random(41)
    out = []
    for line in base.split("\n")[:-1]:
        b = list(line)
        for i in range(len(b)):
            if rng.random() < 0.01:
                b[i] = rng.choice("ACGT")
        out.append("".join(b))
    reads = ("\n".join(out) + "\n").encode()
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], 25, out="ks")
    exp = _suffix_map(exp, "ks")
    import struct
    M = struct.unpack("<8Q", exp[".kmers.header"])[7]
    assert M > 4 * 1500000
    c, got, st = _build(reads, 25)
    assert c.windows == nwin and c.distinct == M
    assert st["segment_retries"] == 0 and st["fused_overflows"] == 0, st
    _same(got, exp)


@pytest.mark.parametrize("k", [16, 31])
def test_fused_path_other_k(oracle, k):
    """Shortest key width with a fused path worth taking (32 bits) and the longest one-word key (62 bits)."""
    reads = g.synth_reads_host(250000, 150, 1000000, seed=12)
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], k, out="ks")
    c, got, st = _build(reads, k)
    assert st["fused_chunks"] == 1
    assert c.windows == nwin
    _same(got, _suffix_map(exp, "ks"))


@pytest.mark.parametrize("k,env,form", [(27, {}, "msd"), (24, {"GOSS_GPU_NO_MSD": "1"}, "lsd")])
def test_fused_path_graph_mode(oracle, k, env, form):
    """build-graph with one-word keys (k <= 30): both strands of every (k+1)-mer go through the
    extraction that partitions; files against the oracle."""
    reads = g.synth_reads_host(260000, 150, 1200000, seed=13)
    exp, nwin = oracle.build_graph([(oracle.LINE, "reads", reads)], k, out="gr")
    c, got, st = _build(reads, k, env=env, mode=g.MODE_GRAPH)
    assert st["fused_chunks"] == 1 and st["fused_overflows"] == 0
    assert st["fused_msd_chunks"] == (1 if form == "msd" else 0)
    assert c.windows == nwin and c.keys == 2 * nwin
    _same(got, _suffix_map(exp, "gr"))


def test_fused_path_contiguous_sequence(oracle):
    """Long sequences (almost every byte starts a window): the bucket regions need the larger key
    buffer; 40 copies of one 1 Mbp sequence."""
    import random
    rng = random.Random(5)
    seq = "".join(rng.choice("ACGT") for _ in range(1_000_000))
    reads = ("\n".join([seq] * 40) + "\n").encode()
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], 27, out="ks")
    c, got, st = _build(reads, 27)
    assert st["fused_chunks"] == 1
    assert c.windows == nwin
    _same(got, _suffix_map(exp, "ks"))


def test_two_level_form_with_sampled_regions():
    """4.6 M reads (695 M window starts): the regions and sub-regions come from a real sample
    (160 M window starts in 64 slices), not from the whole chunk.  The two-level form, the
    one-level form and the unfused sequence must give the same keys and counts."""
    import torch
    from gossamer_amd import dist as gd
    n, L, G = 4_600_000, 150, 5_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    res = []
    for env, want in (({}, "msd"), ({"GOSS_GPU_NO_MSD": "1"}, "lsd"), ({"GOSS_GPU_NO_FUSED": "1"}, "plain")):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = g.Context(25, g.MODE_KMER_SET, hbm_budget=24 << 30)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=21)
            # skew: the last fifth of the input is one read repeated 920 000 times (126 k-mers with
            # huge counts, all at the end of the chunk)
            one = buf[: L + 1].clone()
            buf.view(n, L + 1)[n - n // 5:] = one
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        assert ctx.stat("fused_overflows") == 0
        assert ctx.stat("fused_chunks") == (0 if want == "plain" else 1)
        assert ctx.stat("fused_msd_chunks") == (1 if want == "msd" else 0)
        kp, cp, m = ctx.result_ptrs()
        res.append((gd.device_view(kp, m, torch.int64, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone(), c.windows))
        ctx.close()
    for other in res[1:]:
        assert other[2] == res[0][2]
        assert torch.equal(other[0], res[0][0]) and torch.equal(other[1], res[0][1])
    assert int(res[0][1].to(torch.int64).sum().item()) == res[0][2]


def test_key_buffers_sized_by_valid_windows():
    """Chunks above 640 M window starts get key buffers for the estimated number of valid windows
    (a 64 MB sample of non-base bytes).  Plain reads: the estimate holds and the chunk is counted
    in the smaller buffers.  Reads with long runs of N: a run removes far fewer windows than
    25 per byte, the estimate is far too low, the fused path notices from its own sample and the
    chunk is redone in full-size buffers.  Both must equal the result with the sizing off."""
    import torch
    from gossamer_amd import dist as gd
    n, L, G = 4_600_000, 150, 5_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    for with_n_runs in (False, True):
        res = []
        for env in ({}, {"GOSS_GPU_NO_VALID_SIZING": "1"}):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                ctx = g.Context(25, g.MODE_KMER_SET, hbm_budget=24 << 30)
            finally:
                for k, v in old.items():
                    if v is None:
                        del os.environ[k]
                    else:
                        os.environ[k] = v
            if not res:
                ctx.synth_reads(buf.data_ptr(), n, L, G, seed=33)
                if with_n_runs:
                    # every 1 MiB block starts with 300 KiB of N
                    blocks = buf[: (buf.numel() >> 20) << 20].view(-1, 1 << 20)
                    blocks[:, : 300 << 10] = ord("N")
                torch.cuda.synchronize()
            ctx.push_device(buf.data_ptr(), buf.numel())
            c = ctx.finish()
            assert ctx.stat("fused_chunks") == 1
            if env:
                assert ctx.stat("valid_sized_chunks") == 0 and ctx.stat("valid_resizes") == 0
            elif with_n_runs:
                assert ctx.stat("valid_sized_chunks") == 0 and ctx.stat("valid_resizes") == 1
            else:
                assert ctx.stat("valid_sized_chunks") == 1 and ctx.stat("valid_resizes") == 0
            kp, cp, m = ctx.result_ptrs()
            res.append((gd.device_view(kp, m, torch.int64, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone(), c.windows))
            ctx.close()
        assert res[0][2] == res[1][2]
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        assert int(res[0][1].to(torch.int64).sum().item()) == res[0][2]


def test_valid_window_estimate_too_low_and_no_room_for_full_buffers():
    """Runs of N make the estimate of the valid windows far too low; the arena (9 GB) holds the
    small buffers of one big chunk but not full-size ones: the chunk must be redone in pieces --
    same keys and counts as with a roomy arena and the sizing off."""
    import torch
    from gossamer_amd import dist as gd
    n, L, G = 4_600_000, 150, 5_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    res = []
    for env, budget in (({}, 9 << 30), ({"GOSS_GPU_NO_VALID_SIZING": "1"}, 24 << 30)):
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
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=34)
            blocks = buf[: (buf.numel() >> 20) << 20].view(-1, 1 << 20)
            blocks[:, : 300 << 10] = ord("N")
            torch.cuda.synchronize()
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        if not env:
            assert ctx.stat("valid_resizes") >= 1 and ctx.stat("valid_sized_chunks") == 0
        kp, cp, m = ctx.result_ptrs()
        res.append((gd.device_view(kp, m, torch.int64, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone(), c.windows))
        ctx.close()
    assert res[0][2] == res[1][2]
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert int(res[0][1].to(torch.int64).sum().item()) == res[0][2]


def test_big_counting_table_two_level_form():
    """1.6e8 distinct k-mers in one chunk: more than 65 536 segments hold in the 4096-slot table
    (3/4 of 3072 each), fewer than the 8192-slot table of seg_hash_reduce_big_kernel takes -- the
    two-level form with the big table must give the keys and counts of the three-digit form.
    (The library estimates the key population, 1.72e8, from the repeats in a 4 M-key sample; repeats
    come in clusters of overlapping reads, so the estimate scatters by ~3 %: the genome size sits
    12 % inside both limits, 65 536 x 2304 and a third of the 5.8e8 keys.)"""
    import torch
    from gossamer_amd import dist as gd
    n, L, G = 4_600_000, 150, 172_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    res = []
    # (and with 2 / 4 workgroups sharing every segment, each counting one value of the next 1 / 2 key bits)
    # (the 8-byte forms: the 32-bit-remainder form, which would take this chunk, has its own tests in test_gpu_rem32.py)
    for env in ({}, {"GOSS_GPU_NO_BIG_TABLE": "1"}, {"GOSS_GPU_BIG_ROUNDS_MIN": "1"}, {"GOSS_GPU_BIG_ROUNDS_MIN": "2"}):
        env = dict(env, GOSS_GPU_NO_REM32="1")
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = g.Context(25, g.MODE_KMER_SET, hbm_budget=24 << 30)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=55)
            torch.cuda.synchronize()
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        assert ctx.stat("fused_chunks") == 1 and ctx.stat("segment_retries") == 0
        plain = "GOSS_GPU_NO_BIG_TABLE" in env
        assert ctx.stat("big_table_chunks") == (0 if plain else 1)
        assert ctx.stat("fused_msd_chunks") == (0 if plain else 1)
        kp, cp, m = ctx.result_ptrs()
        assert 150_000_000 < m < 172_000_000
        res.append((gd.device_view(kp, m, torch.int64, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone(), c.windows))
        ctx.close()
    for other in res[1:]:
        assert res[0][2] == other[2]
        assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1])
    assert bool((res[0][0][1:] > res[0][0][:-1]).all().item())
    assert int(res[0][1].to(torch.int64).sum().item()) == res[0][2]


def test_counting_table_too_small_is_replaced_without_redoing_the_chunk():
    """The distinct-key estimate halved on purpose (GOSS_GPU_EST_SCALE): the fused path picks the 4096-slot
    tables for 2.2e8 distinct k-mers (3 350 per segment, the table takes 3 072), they overflow, and only the
    counting is redone with the 8192-slot table (the keys are still in their sub-regions) -- one retry, still
    one fused chunk, same keys and counts."""
    import torch
    from gossamer_amd import dist as gd
    n, L, G = 6_000_000, 150, 230_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    res = []
    for env in ({}, {"GOSS_GPU_EST_SCALE": "0.5"}):
        scaled = bool(env)
        env = dict(env, GOSS_GPU_NO_REM32="1")          # (the 8-byte forms; test_gpu_rem32.py has the 32-bit form's ladder)
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = g.Context(25, g.MODE_KMER_SET, hbm_budget=24 << 30)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=55)
            torch.cuda.synchronize()
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        assert ctx.stat("fused_chunks") == 1 and ctx.stat("fused_msd_chunks") == 1 and ctx.stat("big_table_chunks") == 1
        assert ctx.stat("segment_retries") == (1 if scaled else 0)
        kp, cp, m = ctx.result_ptrs()
        res.append((gd.device_view(kp, m, torch.int64, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone(), c.windows))
        ctx.close()
    assert res[0][2] == res[1][2]
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_big_counting_table_two_word_keys():
    """k = 45, 1.1e8 distinct two-word k-mers in one chunk: more than 65 536 segments of the
    2048-slot table take with margin (1152 each), fewer than seg_hash_reduce2_big_kernel's 4096
    slots do -- the two-level form with the big table (also with two workgroups sharing every
    segment) against the three-digit form."""
    import torch
    from gossamer_amd import dist as gd
    n, L, G = 4_600_000, 150, 110_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    res = []
    for env in ({"GOSS_GPU_NO_TABLE96": "1"}, {"GOSS_GPU_NO_BIG_TABLE": "1"}, {"GOSS_GPU_BIG_ROUNDS_MIN": "1"}, {}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = g.Context(45, g.MODE_KMER_SET, hbm_budget=40 << 30)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=56)
            torch.cuda.synchronize()
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        assert c.key_words == 2
        assert ctx.stat("fused_chunks") == 1 and ctx.stat("segment_retries") == 0
        plain = "GOSS_GPU_NO_BIG_TABLE" in env
        assert ctx.stat("big_table_chunks") == (0 if plain else 1)
        assert ctx.stat("fused_msd_chunks") == (0 if plain else 1)
        assert ctx.stat("table96_chunks") == (0 if env else 1)       # k = 45: 90 - 16 = 74 remainder bits, 16-byte slots
        kp, cp, m = ctx.result_ptrs()
        assert 100_000_000 < m < 110_000_000
        res.append((gd.key_view(kp, m, 2, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone(), c.windows))
        ctx.close()
    for other in res[1:]:
        assert res[0][2] == other[2]
        assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1])
    assert int(res[0][1].to(torch.int64).sum().item()) == res[0][2]


def test_wide_counting_table_two_word_keys():
    """k = 45, ~1.8e8 distinct two-word k-mers in one chunk (2 800 per 16-bit segment): too many for the
    4096-slot table with margin, few enough for the 6144-slot one -- one workgroup and one read per
    segment (seg_hash_reduce2_wide_kernel) against two workgroups sharing every segment of the 4096-slot
    table: same keys and counts."""
    import torch
    from gossamer_amd import dist as gd
    free_b, total_b = torch.cuda.mem_get_info(0)
    if total_b < 100 * (1 << 30):
        pytest.skip("needs ~50 GB of HBM")
    n, L, G = 6_000_000, 150, 190_000_000
    buf = torch.empty(n * (L + 1), dtype=torch.uint8, device="cuda")
    res = []
    for env in ({"GOSS_GPU_NO_TABLE96": "1"}, {"GOSS_GPU_NO_TABLE96": "1", "GOSS_GPU_NO_WIDE_TABLE": "1"}, {}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ctx = g.Context(45, g.MODE_KMER_SET, hbm_budget=48 << 30)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        if not res:
            ctx.synth_reads(buf.data_ptr(), n, L, G, seed=57)
            torch.cuda.synchronize()
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        assert c.key_words == 2
        assert ctx.stat("fused_chunks") == 1 and ctx.stat("segment_retries") == 0
        assert ctx.stat("big_table_chunks") == 1 and ctx.stat("wide_table_chunks") == (1 if len(env) == 1 else 0)
        assert ctx.stat("table96_chunks") == (0 if env else 1)
        kp, cp, m = ctx.result_ptrs()
        assert 170_000_000 < m < 190_000_000
        res.append((gd.key_view(kp, m, 2, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone(), c.windows))
        ctx.close()
    for other in res[1:]:
        assert res[0][2] == other[2]
        assert torch.equal(res[0][0], other[0]) and torch.equal(res[0][1], other[1])
    assert int(res[0][1].to(torch.int64).sum().item()) == res[0][2]


def test_segment_sort_skewed_low_bits(oracle):
    """Keys of one segment that also agree on the ten bits below the segment bits overflow the
    bucket sort's insertion-sort limit: the bitonic fallback of seg_hash_reduce_kernel must give
    the same order.  Reads = one fixed 18-base prefix + 7 random bases, k = 25."""
    import random
    rng = random.Random(8)
    prefix = "ACGTTGCAAGCTTAGGCA"
    reads = [prefix + "".join(rng.choice("ACGT") for _ in range(7)) for _ in range(3000)]
    reads += [rng.choice(reads) for _ in range(3000)]                  # repeats: counts above one
    txt = "\n".join(reads) + "\n"
    for k, build in ((25, oracle.build_kmer_set),):
        exp, nwin = build([(oracle.LINE, "reads", txt)], k, out="ks")
        c, got, st = _build(txt.encode(), k)
        assert c.windows == nwin == 6000
        _same(got, _suffix_map(exp, "ks"))
    # the same through build-graph (both strands, counts kept)
    exp, nwin = oracle.build_graph([(oracle.LINE, "reads", txt)], 24, out="gr")
    with g.Context(24, g.MODE_GRAPH, hbm_budget=1 << 30) as ctx:
        ctx.push_host(txt.encode())
        c = ctx.finish()
        got = ctx.emit()
    assert c.windows == nwin
    _same(got, _suffix_map(exp, "gr"))
    # two-word keys: one fixed 38-base prefix + 7 random bases, k = 45 (the canonical forms of
    # 3000 windows agree on far more than the bits the bucket sort of seg_hash_reduce2_kernel bins
    # on: its bitonic fallback), k-mer set and graph
    prefix2 = "ACGTTGCAAGCTTAGGCATTGACCGTAAGCTTGACAGT"
    reads = [prefix2 + "".join(rng.choice("ACGT") for _ in range(7)) for _ in range(3000)]
    reads += [rng.choice(reads) for _ in range(3000)]
    txt = "\n".join(reads) + "\n"
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", txt)], 45, out="ks")
    c, got, st = _build(txt.encode(), 45)
    assert c.windows == nwin == 6000 and c.key_words == 2
    _same(got, _suffix_map(exp, "ks"))
    exp, nwin = oracle.build_graph([(oracle.LINE, "reads", txt)], 44, out="gr")
    with g.Context(44, g.MODE_GRAPH, hbm_budget=1 << 30) as ctx:
        ctx.push_host(txt.encode())
        c = ctx.finish()
        got = ctx.emit()
    assert c.windows == nwin and c.key_words == 2
    _same(got, _suffix_map(exp, "gr"))


@pytest.mark.parametrize("by_sort", [True, False])
def test_one_segment_with_more_keys_than_any_table(oracle, by_sort):
    """One fixed prefix, 9 random bases behind it: ~2x 20 000 distinct keys of which half share
    their top 32 bits -- the 4096-slot table overflows with 16, 20 and 24 segment bits.  Since round 6 the segment
    whose table overflowed is counted by itself, by sort, and the others keep what their tables counted (skew: a few
    giant segments are what homopolymer stretches make of real reads); GOSS_GPU_OVERFLOW_BY_SORT=0: the ladder of
    rounds 1-5 -- the whole chunk again with more segment bits (2^24 workgroups: the launch HIP used to refuse),
    then the full radix sort -- which still stands behind the new way.  One- and two-word keys, k-mer set and graph."""
    import random
    rng = random.Random(12)
    env = {} if by_sort else {"GOSS_GPU_OVERFLOW_BY_SORT": "0"}
    for k, prefix in ((25, "ACGTTGCAAGCTTAGG"), (45, "ACGTTGCAAGCTTAGGCATTGACCGTAAGCTTGACA")):
        reads = [prefix + "".join(rng.choice("ACGT") for _ in range(9)) for _ in range(20000)]
        reads += [rng.choice(reads) for _ in range(10000)]
        txt = "\n".join(reads) + "\n"
        exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", txt)], k, out="ks")
        c, got, st = _build(txt.encode(), k, env=env)
        assert c.windows == nwin == 30000
        if by_sort:
            assert st["overflow_units"] >= 1 and st["segment_retries"] == 0, st
        else:
            assert st["segment_retries"] >= 3 and st["overflow_units"] == 0, st
        _same(got, _suffix_map(exp, "ks"))
        exp, nwin = oracle.build_graph([(oracle.LINE, "reads", txt)], k - 1, out="gr")
        c, got, st = _build(txt.encode(), k - 1, env=env, budget=2 << 30, mode=g.MODE_GRAPH)
        if by_sort:
            assert st["overflow_units"] >= 1 and st["segment_retries"] == 0, st
        else:
            assert st["segment_retries"] >= 3, st
        assert c.windows == nwin
        _same(got, _suffix_map(exp, "gr"))


def test_runs_of_two_word_keys_merged_through_the_counting_table(oracle):
    """build-graph k = 55 in five pushes (five runs of mostly the same 112-bit edges): the runs are merged by
    hash inserts into the 96-bit-remainder table, segment by segment (seg_hash_merge96_kernel) -- files equal
    to the oracle's; the same with the general merge (GOSS_GPU_NO_TABLE96), and k = 62 (126-bit keys: too wide
    for the remainder table, general merge)."""
    reads = g.synth_reads_host(6000, 150, 60000, seed=44)
    lines = reads.split(b"\n")[:-1]
    pieces = [b"".join(l + b"\n" for l in lines[i::5]) for i in range(5)]
    for k, env, want in ((55, {"GOSS_GPU_HASH_MERGE_MIN": "1"}, 1), (55, {"GOSS_GPU_NO_TABLE96": "1"}, 0), (62, {"GOSS_GPU_HASH_MERGE_MIN": "1"}, 0)):
        exp, nwin = oracle.build_graph([(oracle.LINE, "reads", reads)], k, out="gr")
        old = {n: os.environ.get(n) for n in env}
        os.environ.update(env)
        try:
            import torch
            from gossamer_amd import dist as gd
            with g.Context(k, g.MODE_GRAPH, hbm_budget=1 << 30) as ctx:
                runs, windows = [], 0
                for p in pieces:                       # every piece counted on its own: five sorted runs
                    ctx.reset()
                    ctx.push_host(p)
                    c = ctx.finish()
                    windows += c.windows
                    kp, cp, m = ctx.result_ptrs()
                    runs.append((gd.key_view(kp, m, 2, "cuda").clone(), gd.device_view(cp, m, torch.int32, "cuda").clone()))
                ctx.reset()
                for keys, counts in runs:
                    ctx.push_run(keys.data_ptr(), counts.data_ptr(), keys.shape[0])
                ctx.finish()
                assert ctx.stat("runs") == 1 and ctx.stat("hash_merges") == want, (k, env, ctx.stat("hash_merges"))
                got = ctx.emit()
        finally:
            for n, v in old.items():
                if v is None:
                    del os.environ[n]
                else:
                    os.environ[n] = v
        assert windows == nwin
        _same(got, _suffix_map(exp, "gr"))


def test_segments_of_24_bits_succeed_after_16_and_20_overflow(oracle):
    """One-word keys that need the third width of the segment stage and get through it: build-graph
    k = 27 on 28-base reads (one edge each) that share their first 10 bases (20 key bits), take all
    16 values of the next two bases and 3 000 distinct tails under each -- 48 000 distinct forward
    edges in one 16-bit and one 20-bit segment (the 4096-slot table holds 3 072), 3 000 in each of
    16 segments of 24 bits.  Exactly two retries, the result from the counting kernel (no full sort),
    files equal to the oracle's."""
    import random
    rng = random.Random(77)
    prefix = "GATTACAGAT"
    reads = []
    for a in "ACGT":
        for b in "ACGT":
            tails = set()
            while len(tails) < 3000:
                tails.add("".join(rng.choice("ACGT") for _ in range(16)))
            for t in tails:
                reads += [prefix + a + b + t] * 4
    rng.shuffle(reads)
    txt = "\n".join(reads) + "\n"
    exp, nwin = oracle.build_graph([(oracle.LINE, "reads", txt)], 27, out="gr")
    assert nwin == len(reads) == 192000
    # (the ladder itself: since round 6 a segment whose table overflows is counted by sort first -- switched off here)
    old = os.environ.get("GOSS_GPU_OVERFLOW_BY_SORT")
    os.environ["GOSS_GPU_OVERFLOW_BY_SORT"] = "0"
    try:
        with g.Context(27, g.MODE_GRAPH, hbm_budget=1 << 30) as ctx:
            ctx.push_host(txt.encode())
            c = ctx.finish()
            assert c.windows == nwin and c.distinct == 2 * 48000
            assert ctx.stat("segment_retries") == 2, ctx.stat("segment_retries")
            got = ctx.emit()
    finally:
        if old is None:
            os.environ.pop("GOSS_GPU_OVERFLOW_BY_SORT", None)
        else:
            os.environ["GOSS_GPU_OVERFLOW_BY_SORT"] = old
    _same(got, _suffix_map(exp, "gr"))
    with g.Context(27, g.MODE_GRAPH, hbm_budget=1 << 30) as ctx:
        ctx.push_host(txt.encode())
        c = ctx.finish()
        assert c.windows == nwin and c.distinct == 2 * 48000
        assert ctx.stat("segment_retries") == 0 and ctx.stat("overflow_units") >= 1
        _same(ctx.emit(), _suffix_map(exp, "gr"))


def test_fused_path_declines_unique_input(oracle):
    """No duplication (every k-mer once): the sample says so and the plain sequence runs."""
    import numpy as np
    rng = np.random.default_rng(6)
    reads = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 40_000_000)].tobytes() + b"\n"
    with g.Context(25, g.MODE_KMER_SET, hbm_budget=8 << 30) as ctx:
        ctx.push_host(reads)
        c = ctx.finish()
        assert ctx.stat("fused_chunks") == 0
        assert c.windows == 40_000_000 - 24
        assert c.distinct > 0.98 * c.windows


def test_extreme_skew_one_kmer_five_billion_times():
    """4.6 G 'A's: one canonical 25-mer, 4.6e9 times.  Every key lands in one bucket, one
    sub-region and one segment; the segment is too long for 32-bit slot counts, so the chunk goes
    down to the full sort, whose run length saturates -- a k-mer set stores no counts and must
    come out with exactly one k-mer.  A graph with the same shape (2 x 1e8 windows) keeps exact
    counts."""
    import torch
    n = 4_600_000_000
    buf = torch.full((n + 1,), 65, dtype=torch.uint8, device="cuda")
    buf[-1] = 10
    with g.Context(25, g.MODE_KMER_SET, hbm_budget=100 << 30) as ctx:
        ctx.push_device(buf.data_ptr(), buf.numel())
        c = ctx.finish()
        assert c.windows == n - 24 and c.distinct == 1
        keys, counts = ctx.result()
        assert keys == [0] and counts[0] == 0xFFFFFFFF
        files = ctx.emit()
        import struct
        assert struct.unpack("<QQQ", files[".header"]) == (2011101701, 25, 1)
    m = 100_000_000
    with g.Context(27, g.MODE_GRAPH, hbm_budget=16 << 30) as ctx:
        ctx.push_device(buf.data_ptr(), m + 1)          # the byte after the m A's is another A: m + 1 - 28 + 1 windows
        c = ctx.finish()
        keys, counts = ctx.result()
        assert c.distinct == 2 and keys == [0, 4 ** 28 - 1]
        assert counts[0] == counts[1] == m + 1 - 27
        assert ctx.big_counts() == {}
    # the whole string as a graph: both edges occur n - 27 = 4 599 999 973 times, more than a u32 holds.  The
    # reference keeps the u64 count as the histogram key and stores it narrowed to 32 bits in the
    # VariableByteArray (Graph.hh:101-106, VariableByteArray.hh:72,81); first-principles expectation, both
    # with the string counted in chunks (the counts meet 2^32 in a merge) and in as few as fit
    for budget in (48 << 30, 200 << 30):
        with g.Context(27, g.MODE_GRAPH, hbm_budget=budget) as ctx:
            ctx.push_device(buf.data_ptr(), buf.numel())
            c = ctx.finish()
            want = n - 27
            assert c.windows == want and c.distinct == 2 and c.keys == 2 * want
            keys, counts = ctx.result()
            assert keys == [0, 4 ** 28 - 1]
            assert [int(x) for x in counts] == [want & 0xFFFFFFFF] * 2
            assert ctx.big_counts() == {0: want, 4 ** 28 - 1: want}
            files = ctx.emit()
            assert files["-counts-hist.txt"] == b"%d\t2\n" % want
            # the VariableByteArray holds the narrowed value: ord0 = low byte, ord1 = next byte, ord2 = the high half
            v = want & 0xFFFFFFFF
            assert files["-counts.ord0"] == bytes([v & 0xFF] * 2)
            assert files["-counts.ord1"] == bytes([(v >> 8) & 0xFF] * 2)
            assert files["-counts.ord2"] == struct.pack("<HH", v >> 16, v >> 16)


@pytest.mark.parametrize("k,mode,form", [(13, "kmer", "msd"), (16, "kmer", "lsd"), (25, "kmer", "msd"), (31, "kmer", "lsd"),
                                         (12, "graph", "msd"), (24, "graph", "lsd"), (30, "graph", "msd"),
                                         (32, "kmer", "msd"), (45, "kmer", "lsd"), (63, "kmer", "msd"),
                                         (31, "graph", "lsd"), (55, "graph", "msd"), (62, "graph", "lsd")])
def test_fused_kernels_on_ragged_reads(oracle, k, mode, form):
    """The fused kernels on reads that are nothing like the synthetic set: lengths 20..180 (some
    shorter than k), 2 % non-ACGT bytes of several kinds, lower case, heavy duplication.
    GOSS_GPU_FUSED_MIN=0 sends this small input (about 4.5 M window starts) down the fused path;
    files against the oracle."""
    import random
    rng = random.Random(1000 + k)
    genome = "".join(rng.choice("ACGT") for _ in range(30000))
    reads = []
    bad = 0.02 if k < 32 else 0.004             # long windows survive fewer non-bases
    for _ in range(45000 if k < 32 else 70000):
        L = rng.randint(20, 180)
        p = rng.randint(0, len(genome) - L)
        r = list(genome[p:p + L])
        for i in range(L):
            x = rng.random()
            if x < bad:
                r[i] = rng.choice("NnRY.-*")
            elif x < 0.3:
                r[i] = r[i].lower()
        reads.append("".join(r))
    txt = "\n".join(reads) + "\n"
    env = {"GOSS_GPU_FUSED_MIN": "0"}
    if form == "lsd":
        env["GOSS_GPU_NO_MSD"] = "1"
    if mode == "kmer":
        exp, nwin = oracle.build_kmer_set([(oracle.LINE, "r", txt)], k, out="o")
        c, got, st = _build(txt.encode(), k, env=env, budget=1 << 30)
    else:
        exp, nwin = oracle.build_graph([(oracle.LINE, "r", txt)], k, out="o")
        c, got, st = _build(txt.encode(), k, env=env, budget=1 << 30, mode=g.MODE_GRAPH)
    assert st["fused_chunks"] == 1 and st["fused_overflows"] == 0, st
    assert st["fused_msd_chunks"] == (1 if form == "msd" else 0)
    assert c.windows == nwin
    _same(got, _suffix_map(exp, "o"))


def test_arena_grows_for_inputs_without_duplication():
    """60 M random bases (every 25-mer once): the sorted runs do not shrink, so a 1 GB arena cannot
    hold the merge; with a budget limit the context maps a larger arena, moves its runs and its
    staging buffer over, and ends with the same result as a context that had room from the start."""
    import numpy as np
    import torch
    from gossamer_amd import dist as gd
    rng = np.random.default_rng(16)
    reads = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 60_000_000)].tobytes() + b"\n"
    res = []
    for budget, limit in ((1 << 30, 24 << 30), (16 << 30, 0)):
        with g.Context(25, g.MODE_KMER_SET, hbm_budget=budget) as ctx:
            if limit:
                ctx.set_budget_limit(limit)
            for i in range(0, len(reads), 8 << 20):                # host pushes through the staging buffer
                ctx.push_host(reads[i:i + (8 << 20)])
            c = ctx.finish()
            grows = ctx.stat("arena_grows")
            assert (grows >= 1) == bool(limit)
            kp, cp, m = ctx.result_ptrs()
            keys = gd.device_view(kp, m, torch.int64, "cuda").clone()
            files = ctx.emit()
            res.append((keys, c.windows, c.distinct, files))
    assert res[0][1] != 0
    # pushes of 8 MB cut windows at their borders (windows never span two pushes): both contexts
    # were fed the same pieces, so everything must agree
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][3] == res[1][3]
    # without a limit the small arena must fail loudly, not silently drop data
    with g.Context(25, g.MODE_KMER_SET, hbm_budget=1 << 30) as ctx:
        with pytest.raises(g.GossGpuError):
            for i in range(0, len(reads), 8 << 20):
                ctx.push_host(reads[i:i + (8 << 20)])
            ctx.finish()
            ctx.emit()


@pytest.mark.parametrize("k,mode", [(25, "kmer"), (27, "graph"), (45, "kmer")])
def test_fused_path_misaligned_device_buffer(k, mode):
    """push_bases_device with a pointer that is not 16-byte aligned (the kernels load 16-byte
    vectors from the aligned address below it): same result as the aligned copy."""
    import torch
    from gossamer_amd import dist as gd
    reads = g.synth_reads_host(300000, 150, 1500000, seed=17)
    base = torch.frombuffer(bytearray(b"\n" * 16 + reads), dtype=torch.uint8).cuda()
    res = []
    for off in (16, 3, 9):
        view = base[off:]
        if off != 16:
            view = base[off: off + len(reads)]
            view.copy_(base[16: 16 + len(reads)].clone())
        torch.cuda.synchronize()          # the context runs on its own stream: the bytes must be there
        m = g.MODE_GRAPH if mode == "graph" else g.MODE_KMER_SET
        with g.Context(k, m, hbm_budget=12 << 30) as ctx:
            ctx.push_device(view.data_ptr(), len(reads))
            c = ctx.finish()
            assert ctx.stat("fused_chunks") == 1
            kp, cp, n = ctx.result_ptrs()
            w = c.key_words
            res.append((gd.device_view(kp, n * w, torch.int64, "cuda").clone(), gd.device_view(cp, n, torch.int32, "cuda").clone(), c.windows))
        base[16: 16 + len(reads)].copy_(torch.frombuffer(bytearray(reads), dtype=torch.uint8).cuda())
    for other in res[1:]:
        assert other[2] == res[0][2]
        assert torch.equal(other[0], res[0][0]) and torch.equal(other[1], res[0][1])


@pytest.mark.parametrize("mode", ["kmer", "graph"])
def test_fused_path_many_k(oracle, mode):
    """The fused kernels over the key widths they serve (24..126 bits): keys and counts against the
    oracle, small input forced down the fused path."""
    import random
    rng = random.Random(4242)
    genome = "".join(rng.choice("ACGT") for _ in range(20000))
    reads = []
    for _ in range(30000):
        L = rng.randint(70, 150)
        p = rng.randint(0, len(genome) - L)
        r = genome[p:p + L]
        if rng.random() < 0.05:
            q = rng.randrange(L)
            r = r[:q] + "N" + r[q + 1:]
        reads.append(r)
    txt = "\n".join(reads) + "\n"
    old = os.environ.get("GOSS_GPU_FUSED_MIN")
    os.environ["GOSS_GPU_FUSED_MIN"] = "0"
    try:
        top = 63 if mode == "kmer" else 62
        ks_to_try = sorted(set(list(range(12, top + 1, 3)) + [30, 31, 32, 33, top]))
        for k in ks_to_try:
            length = k if mode == "kmer" else k + 1
            ek, ec, nreads, nwin = oracle.count([(oracle.LINE, "r", txt)], length, 0 if mode == "kmer" else 1)
            with g.Context(k, g.MODE_KMER_SET if mode == "kmer" else g.MODE_GRAPH, hbm_budget=2 << 30) as ctx:
                ctx.push_host(txt)
                c = ctx.finish()
                assert ctx.stat("fused_chunks") == 1, k
                gk, gc = ctx.result()
            assert c.windows == nwin, k
            assert gk == ek, k
            assert [int(x) for x in gc] == ec, k
    finally:
        if old is None:
            del os.environ["GOSS_GPU_FUSED_MIN"]
        else:
            os.environ["GOSS_GPU_FUSED_MIN"] = old
