"""Super-k-mer records (include/goss_gpu.h: goss_gpu_route_records_device / goss_gpu_push_records_device): the exchange
BEFORE counting of a multi-GPU build.  What must hold (kernels_route.hpp): the records of all parts together carry
every valid window of the reads exactly once (KmerizingAdapter.hh:20-86 / ReverseComplementAdapter.hh:20-93 key stream),
and all copies of a key -- either strand -- are in ONE part.  Checked against the oracle's count of the same reads."""
import os
import random

import pytest
import torch

import gossamer_amd as g
from test_gpu_parity import make_reads, oracle_counts

pytestmark = pytest.mark.gpu
MB = 1 << 20
REC = g.binding.RECORD_BYTES


def route(ctx, text, nparts, caps=None):
    """-> (records tensor, part_first, part_records, part_windows)"""
    dev = torch.device("cuda", 0)
    bases = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    if caps is None:
        caps = [len(text) + 64] * nparts          # one record per window is the worst case (short k: a window is its own minimizer)
    first = [sum(caps[:p]) for p in range(nparts)]
    buf = torch.empty(max(1, sum(caps)) * g.binding.record_bytes(ctx.k, ctx.mode), dtype=torch.uint8, device=dev)
    recs, wins, ok = ctx.route_records(bases.data_ptr(), bases.numel(), nparts, buf.data_ptr(), first, caps)
    return buf, first, recs, wins, ok


def count_parts(k, mode, buf, first, recs, wins, parts, budget=256 * MB):
    rb = g.binding.record_bytes(k, mode)
    with g.Context(k, mode, hbm_budget=budget) as ctx:
        for p in parts:
            if recs[p]:
                ctx.push_records(buf.data_ptr() + first[p] * rb, recs[p], wins[p])
        c = ctx.finish()
        ks, cs = ctx.result()
        files = ctx.emit()
        stats = {s: ctx.stat(s) for s in ("fused_chunks", "rec_chunks")}
    return ks, [int(x) for x in cs], c, files, stats


@pytest.mark.parametrize("k,mode", [(25, 0), (11, 0), (15, 0), (19, 0), (21, 0), (31, 0), (7, 0), (3, 0), (24, 1), (15, 1), (30, 1), (9, 1),
                                    # two-word keys: 20-byte records, the minimizer of the window's central 31 / 30 bases
                                    (32, 0), (33, 0), (45, 0), (63, 0), (62, 0), (31, 1), (32, 1), (55, 1), (62, 1), (44, 1)])
@pytest.mark.parametrize("nparts", [1, 3, 8])
def test_records_of_all_parts_hold_every_window_once(oracle, k, mode, nparts):
    rng = random.Random(900 + 7 * k + mode + nparts)
    reads = make_reads(rng, 500, (max(4, k - 2), 170 if k < 40 else 230), 5000, lower=True)
    text = ("\n".join(reads) + "\n").encode()
    length = k + 1 if mode else k
    ek, ec, nwin = oracle_counts(oracle, reads, length, mode)
    build = oracle.build_graph if mode else oracle.build_kmer_set
    exp, _ = build([(oracle.LINE, "r", text)], k, out="o")
    exp = {n[1:]: d for n, d in exp.items()}
    with g.Context(k, mode, hbm_budget=64 * MB) as rctx:
        buf, first, recs, wins, ok = route(rctx, text, nparts)
    assert ok and sum(wins) == nwin
    for fused in (False, True):          # the plain record kernel, and the record form of the fused extraction
        if fused:
            os.environ["GOSS_GPU_FUSED_MIN"] = "0"
        try:
            ks, cs, c, files, stats = count_parts(k, mode, buf, first, recs, wins, range(nparts))
        finally:
            os.environ.pop("GOSS_GPU_FUSED_MIN", None)
        assert c.windows == nwin and c.keys == nwin * (2 if mode else 1)
        assert ks == ek and cs == ec, (fused, stats)
        assert files == exp
    # every key in exactly one part (graph mode: an edge and its reverse complement together), counts final there
    seen = {}
    for p in range(nparts):
        ks, cs, c, _, _ = count_parts(k, mode, buf, first, recs, wins, [p])
        assert c.windows == wins[p]
        for key, n in zip(ks, cs):
            assert key not in seen, "key counted in two parts"
            seen[key] = n
        if mode:
            have = set(ks)
            assert all(oracle.revcomp(key, length) in have for key in ks[:2000])
    assert seen == dict(zip(ek, ec))


def test_small_part_buffers_are_reported_and_a_second_call_fits(oracle):
    k = 25
    rng = random.Random(5150)
    reads = make_reads(rng, 800, 150, 20000)
    text = ("\n".join(reads) + "\n").encode()
    ek, ec, nwin = oracle_counts(oracle, reads, k, 0)
    with g.Context(k, 0, hbm_budget=64 * MB) as rctx:
        _, _, need, wins, ok = route(rctx, text, 4, caps=[10, 10, 10, 10])
        assert not ok and all(n > 10 for n in need) and sum(wins) == nwin
        buf, first, recs, wins, ok = route(rctx, text, 4, caps=need)          # exactly what was asked for
        assert ok and recs == need
    ks, cs, c, _, _ = count_parts(k, 0, buf, first, recs, wins, range(4))
    assert ks == ek and cs == ec


def test_misaligned_bases_both_key_widths():
    """A base string that does not start on a 16-byte boundary (the pieces of a pipelined exchange start anywhere): a
    routing thread's 64 bases then come from five vectors -- a record may run 16 windows into the next thread's."""
    rng = random.Random(77)
    reads = make_reads(rng, 1500, (90, 160), 6000)
    text = ("\n".join(reads) + "\n").encode()
    dev = torch.device("cuda", 0)
    for k, mode in ((21, 0), (25, 0), (31, 0), (27, 1), (30, 1), (45, 0), (55, 1), (63, 0), (32, 0)):
        REC = g.binding.record_bytes(k, mode)
        with g.Context(k, mode, hbm_budget=64 * MB) as ctx:
            ctx.push_host(text)
            ctx.finish()
            want = ctx.result()
        for off in (1, 6, 7, 13, 15):
            t = torch.zeros(len(text) + 32, dtype=torch.uint8, device=dev)
            t[off:off + len(text)] = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
            cap = len(text) // 4
            buf = torch.empty(cap * 2 * REC, dtype=torch.uint8, device=dev)
            with g.Context(k, mode, hbm_budget=64 * MB) as ctx:
                recs, wins, ok = ctx.route_records(t.data_ptr() + off, len(text), 2, buf.data_ptr(), [0, cap], [cap, cap])
                assert ok
                for p in range(2):
                    ctx.push_records(buf.data_ptr() + p * cap * REC, recs[p], wins[p])
                ctx.finish()
                got = ctx.result()
            assert got[0] == want[0] and list(got[1]) == list(want[1]), (k, mode, off)


def test_records_at_a_size_the_fused_path_takes_by_itself(oracle):
    """400 k reads of 150 bp: ~9 M records = 140 M window slots, above the fused path's minimum."""
    k = 25
    reads = g.synth_reads_host(400_000, 150, 2_000_000, seed=3)
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "r", reads)], k, out="o")
    exp = {n[1:]: d for n, d in exp.items()}
    with g.Context(k, 0, hbm_budget=64 * MB) as rctx:
        _, _, need, _, _ = route(rctx, reads, 8, caps=[1] * 8)
        buf, first, recs, wins, ok = route(rctx, reads, 8, caps=need)          # exact sizes: the parts lie back to back
    assert ok and sum(wins) == nwin and recs == need
    # the routing workgroups take room in blocks and fill what they do not use with pads (third word 1 << 27, no
    # window): counted as records, skipped by the counting side
    w2 = buf.view(torch.int32).view(-1, 3)[:, 2].to(torch.int64) & 0xFFFFFFFF
    pads = (w2 >> 27) == 1
    npads = int(pads.sum())
    assert 0 < npads <= 8 * 1280 * 512, npads
    assert int(((w2 >> 28) + 1)[~pads].sum()) == nwin
    # one push of everything, as a rank pushes the segments it received from all ranks
    recs, wins, first = [sum(recs)], [sum(wins)], [0]
    ks, cs, c, files, stats = count_parts(k, 0, buf, first, recs, wins, [0], budget=6 << 30)
    assert c.windows == nwin
    assert stats["rec_chunks"] >= 1, stats
    assert files == exp
    print("windows per record: %.2f, record bytes per window: %.2f" % (nwin / sum(recs), 12.0 * sum(recs) / nwin))
    # one and three parts: a tile's records of a part outnumber the block a workgroup takes ahead (512 slots at one
    # part: ~550 records per tile), so tiles are split between the rest of a block and new room, several blocks at once
    for nparts in (1, 3):
        with g.Context(k, 0, hbm_budget=64 * MB) as rctx:
            _, _, need, _, _ = route(rctx, reads, nparts, caps=[1] * nparts)
            buf, first, recs, wins, ok = route(rctx, reads, nparts, caps=need)
        assert ok and sum(wins) == nwin and recs == need
        w2 = buf.view(torch.int32).view(-1, 3)[:, 2].to(torch.int64) & 0xFFFFFFFF
        pads = (w2 >> 27) == 1
        assert int(((w2 >> 28) + 1)[~pads].sum()) == nwin
        ks, cs, c, files, stats = count_parts(k, 0, buf, [0], [sum(recs)], [sum(wins)], [0], budget=6 << 30)
        assert c.windows == nwin and files == exp


@pytest.mark.parametrize("nparts", [64, 256])
def test_many_parts_small_blocks(oracle, nparts):
    """64 and 256 parts (the most the boundary takes): the routing workgroups take room in blocks of 64 / 32 slots, a tile
    brings a handful of records per part; everything pushed in one go equals the oracle's build."""
    k = 21
    reads = g.synth_reads_host(120_000, 150, 700_000, seed=9)
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "r", reads)], k, out="o")
    exp = {n[1:]: d for n, d in exp.items()}
    with g.Context(k, 0, hbm_budget=64 * MB) as rctx:
        _, _, need, _, _ = route(rctx, reads, nparts, caps=[1] * nparts)
        buf, first, recs, wins, ok = route(rctx, reads, nparts, caps=need)
    assert ok and sum(wins) == nwin and recs == need
    assert max(wins) < 3 * nwin / nparts          # (no part far above its share)
    ks, cs, c, files, stats = count_parts(k, 0, buf, [0], [sum(recs)], [sum(wins)], [0], budget=2 << 30)
    assert c.windows == nwin and files == exp


@pytest.mark.parametrize("k,mode,env", [(27, 1, {}), (24, 1, {}), (30, 1, {}), (22, 0, {}), (16, 0, {}), (25, 0, {"GOSS_GPU_NO_MSD": "1"}),
                                        (27, 1, {"GOSS_GPU_NO_MSD": "1"}), (31, 0, {}), (21, 0, {"GOSS_GPU_NO_REM32": "1"}),
                                        # two-word keys: the record form of extract2_part_kernel (20-byte records; graphs as strand pairs)
                                        (45, 0, {}), (55, 1, {}), (31, 1, {}), (63, 0, {}), (32, 0, {}), (62, 1, {}), (45, 0, {"GOSS_GPU_NO_MSD": "1"}),
                                        (27, 1, {"GOSS_GPU_NO_GRAPH_REP": "1"})])
def test_fused_record_kernel_variants_at_its_own_size(oracle, k, mode, env):
    """The record form of the fused extraction in every variant it is compiled in: graph mode (a thread takes 8
    windows, so a record of up to 16 holds the first window of TWO threads), even k (strand_rep instead of the middle
    base), the one-level form (digit histograms in the kernel, GOSS_GPU_NO_MSD) -- at a size the fused path takes by
    itself (300 k reads: ~100 M window slots), against the oracle's files.  rec_chunks says the kernel ran."""
    reads = g.synth_reads_host(300_000, 150, 1_500_000, seed=40 + k + mode)
    build = oracle.build_graph if mode else oracle.build_kmer_set
    exp, nwin = build([(oracle.LINE, "r", reads)], k, out="o")
    exp = {n[1:]: d for n, d in exp.items()}
    with g.Context(k, mode, hbm_budget=64 * MB) as rctx:
        _, _, need, _, _ = route(rctx, reads, 4, caps=[1] * 4)
        buf, first, recs, wins, ok = route(rctx, reads, 4, caps=need)
    assert ok and sum(wins) == nwin and recs == need
    old = {n: os.environ.get(n) for n in env}
    os.environ.update(env)
    try:
        # (two-word graph keys: room for the whole input as ONE chunk -- a chunk below the fused path's minimum is counted unfused)
        ks, cs, c, files, stats = count_parts(k, mode, buf, [0], [sum(recs)], [sum(wins)], [0], budget=(16 if k > 30 and mode else 6) << 30)
    finally:
        for n, v in old.items():
            if v is None:
                os.environ.pop(n, None)
            else:
                os.environ[n] = v
    assert c.windows == nwin and c.keys == nwin * (2 if mode else 1)
    assert stats["rec_chunks"] >= 1, stats
    assert files == exp
