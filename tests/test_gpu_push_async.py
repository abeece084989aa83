"""goss_gpu_push_bases_host_async / goss_gpu_push_packed_host(_async) / goss_gpu_flush: the same key stream as the
synchronous byte form (include/goss_gpu.h), whatever the size and number of the pushes; every lent buffer comes back."""
import random

import pytest

import gossamer_amd as g
from test_gpu_parity import make_reads, oracle_counts

pytestmark = pytest.mark.gpu
MB = 1 << 20


def batches(rng, reads, nb):
    per = (len(reads) + nb - 1) // nb
    return [("\n".join(reads[i:i + per]) + rng.choice(["\n", "\n", "", "N"])).encode() for i in range(0, len(reads), per)]


@pytest.mark.parametrize("k,mode", [(25, 0), (31, 1), (45, 0), (55, 1)])
def test_async_and_packed_pushes_count_what_the_synchronous_push_counts(oracle, k, mode):
    rng = random.Random(31 * k + mode)
    reads = make_reads(rng, 3000, (max(5, k - 3), 160), 30000, lower=True)
    length = k + 1 if mode else k
    for nb in (1, 7, 200):
        bs = batches(random.Random(nb), reads, nb)
        # a batch that does not end with a separator must not join the next one: the oracle sees them apart
        ek, ec, nwin = oracle_counts(oracle, [b.decode() for b in bs], length, mode)
        for how in ("sync", "async", "packed", "packed-async", "mixed"):
            released = []
            with g.Context(k, mode, hbm_budget=(64 if nb == 200 else 256) * MB) as ctx:
                for i, b in enumerate(bs):
                    h = how if how != "mixed" else ("sync", "async", "packed", "packed-async")[i % 4]
                    if h == "sync":
                        ctx.push_host(b)
                    elif h == "async":
                        ctx.push_host_async(b, on_release=lambda i=i: released.append(i))
                    else:
                        ctx.push_packed_host(b, async_=(h == "packed-async"))
                if how == "async" and nb == 7:
                    ctx.flush()
                    assert sorted(released) == list(range(len(bs)))
                c = ctx.finish()
                ks, cs = ctx.result()
                assert not getattr(ctx, "_lent", None), "a buffer was not handed back by finish"
            if how == "async":
                assert sorted(released) == list(range(len(bs)))
            assert c.windows == nwin, (how, nb)
            assert ks == ek and [int(x) for x in cs] == ec, (how, nb)


def test_packed_push_larger_than_the_staging_buffer(oracle):
    """48 MB of positions into a context whose staging buffer holds ~10 MB: unpacked and counted piece by piece, no
    window lost where the staging buffer is counted in between."""
    k = 21
    reads = g.synth_reads_host(320_000, 150, 400_000, seed=5)
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "r", reads)], k, out="o")
    exp = {n[1:]: d for n, d in exp.items()}
    with g.Context(k, 0, hbm_budget=256 * MB) as ctx:
        ctx.push_packed_host(reads)
        c = ctx.finish()
        assert c.windows == nwin
        assert ctx.emit() == exp


def test_host_pushes_beside_an_arena_that_takes_nearly_all_memory(oracle):
    """The two staging buffers of host pushes are device memory beside the arena: an arena that was given all but
    a few GB of the device leaves less than 1/24 of itself twice -- the buffers shrink to what is there."""
    import torch
    free_b, _ = torch.cuda.mem_get_info(0)
    k = 25
    rng = random.Random(2)
    reads = make_reads(rng, 2000, 150, 20000)
    ek, ec, nwin = oracle_counts(oracle, reads, k, 0)
    with g.Context(k, 0, hbm_budget=int(free_b * 0.9)) as ctx:
        for i in range(0, len(reads), 500):
            ctx.push_host_async("\n".join(reads[i:i + 500]) + "\n")
        c = ctx.finish()
        ks, cs = ctx.result()
    assert c.windows == nwin and ks == ek and [int(x) for x in cs] == ec


@pytest.mark.parametrize("k,mode", [(25, 0), (27, 1), (55, 1)])
def test_packed_pushes_are_read_as_they_are_by_the_fused_kernels(oracle, k, mode):
    """Packed bases stay packed in HBM (round 5): at a size the fused path takes by itself its kernels -- the sample's, the
    first level's -- read the staged groups of 2-bit codes and flags, nothing is unpacked to bytes
    (packed_fused_chunks / packed_unpacked_chunks), and the files are the oracle's.  Pushed in pieces that do not end on
    group boundaries, alternately with and without waiting for the copy."""
    reads = g.synth_reads_host(300_000, 150, 1_500_000, seed=900 + k)
    build_o = oracle.build_graph if mode else oracle.build_kmer_set
    exp, nwin = build_o([(oracle.LINE, "reads", reads)], k, out="o")
    exp = {n[1:]: d for n, d in exp.items()}
    lines = reads.split(b"\n")[:-1]
    with g.Context(k, mode, hbm_budget=24 << 30) as ctx:          # (room for one chunk of two-word keys: the fused path's)
        for i in range(0, len(lines), 77_777):
            ctx.push_packed_host(b"".join(l + b"\n" for l in lines[i:i + 77_777]), async_=(i % 2 == 0))
        c = ctx.finish()
        got = ctx.emit()
        st = {n: ctx.stat(n) for n in ("packed_fused_chunks", "packed_unpacked_chunks", "fused_chunks")}
    assert c.windows == nwin
    assert st["packed_fused_chunks"] >= 1 and st["packed_unpacked_chunks"] == 0 and st["fused_chunks"] == st["packed_fused_chunks"], st
    assert got == exp


@pytest.mark.parametrize("mis", [0, 1, 5, 15])
def test_pack_bases_device_is_the_host_packer(mis):
    """goss_gpu_pack_bases_device (the per-base encoder of GossReadBaseString.hh:133-188 as a kernel): codes and flags of
    a byte string that starts anywhere equal the numpy restatement of the host parser's packer -- lower case, N, line
    ends and arbitrary bytes flagged, the positions behind the string's end in its last group flagged."""
    import numpy as np
    import torch
    rng = random.Random(17 + mis)
    for n in (1, 15, 16, 17, 4096 + 3, 100_001):
        text = bytes(rng.choice(b"ACGTacgtNn\n\r-\x00\xff" if rng.random() < 0.2 else b"ACGT") for _ in range(n))
        codes, bad = g.binding.pack_bases(text)
        buf = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        buf[mis:mis + n] = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        groups = (n + 15) // 16
        dc = torch.full((groups + 1,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
        db = torch.full((groups + 1,), 0x5A5A, dtype=torch.int16, device="cuda")
        with g.Context(25, 0, hbm_budget=64 * MB) as ctx:
            ctx.pack_bases_device(buf.data_ptr() + mis, n, dc.data_ptr(), db.data_ptr())
        gc = dc.cpu().numpy().view(np.uint32)
        gb = db.cpu().numpy().view(np.uint16)
        assert gc[groups] == 0x5A5A5A5A and gb[groups] == 0x5A5A          # (nothing written behind the last group)
        # (codes of flagged positions are not specified: compared where the position is a base)
        mask = np.zeros(groups, dtype=np.uint32)
        for j in range(16):
            mask |= ((~bad.astype(np.uint32) >> j) & 1) * (3 << (2 * j))
        assert np.array_equal(gb[:groups], bad), (n, mis)
        assert np.array_equal(gc[:groups] & mask, codes & mask), (n, mis)


@pytest.mark.parametrize("k,mode,nreads", [(25, 0, 4000), (27, 1, 4000), (55, 1, 4000), (25, 0, 400_000), (31, 1, 400_000), (45, 0, 400_000)])
def test_packed_bases_resident_in_hbm(oracle, k, mode, nreads):
    """goss_gpu_push_packed_device: packed bases that already lie in HBM -- packed there by goss_gpu_pack_bases_device
    -- are counted where they lie; files equal to the oracle's build of the bytes.  Small inputs take the plain kernels
    (the chunk is unpacked for them); of 400 000 reads in two pushes the second -- 300 000 reads, 45 M window starts -- is a
    chunk the fused path takes, whose kernels read the groups as they are."""
    import torch
    reads = g.synth_reads_host(nreads, 150, 5 * nreads, seed=77 + k)
    exp, nwin = (oracle.build_graph if mode else oracle.build_kmer_set)([(oracle.LINE, "reads", reads)], k, out="o")
    exp = {n[1:]: d for n, d in exp.items()}
    lines = reads.split(b"\n")[:-1]
    cut = len(lines) // 4
    parts = [b"".join(l + b"\n" for l in lines[:cut]), b"".join(l + b"\n" for l in lines[cut:])]
    with g.Context(k, mode, hbm_budget=24 << 30) as ctx:
        held = []
        for text in parts:
            buf = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
            groups = (len(text) + 15) // 16
            dc = torch.empty(groups, dtype=torch.int32, device="cuda")
            db = torch.empty(groups, dtype=torch.int16, device="cuda")
            ctx.pack_bases_device(buf.data_ptr(), len(text), dc.data_ptr(), db.data_ptr())
            del buf
            ctx.push_packed_device(dc.data_ptr(), db.data_ptr(), len(text))
            held.append((dc, db))
        c = ctx.finish()
        got = ctx.emit()
        st = {n: ctx.stat(n) for n in ("packed_fused_chunks", "packed_unpacked_chunks", "fused_chunks")}
    assert c.windows == nwin
    if nreads >= 400_000:
        assert st["packed_fused_chunks"] >= 1 and st["fused_chunks"] == st["packed_fused_chunks"], st
    assert got == exp
