"""world_size-2 test of the multi-GPU exchange logic on CPU (gloo): range partition of the
sorted distinct keys, all-to-all(v) of (key,count) runs, per-range merge, all-gather of the
range sizes, assembly on rank 0.  The per-rank counting and the merge of received runs are
done by the oracle here; on GPUs the same functions of gossamer_amd.dist run over RCCL with
libgossgpu.so doing the counting (bench.py --gpus N)."""
import os
import random
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, k, shards, expect_keys, expect_counts, q, mode="uniform"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    from gossamer_amd import dist as gd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # local count of this rank's share of the reads
        keys, nreads, nwin = o.collect([(o.LINE, "r", shards[rank])], k, 0)
        d = {}
        for x in keys:
            d[x] = d.get(x, 0) + 1
        ks = sorted(d)
        two = 2 * k > 62                          # two-word keys travel as [m, 2] = (lo, hi)

        def to_tensor(vals):
            if not two:
                return torch.tensor(vals, dtype=torch.int64)
            return torch.tensor([[gd._as_i64(v), gd._as_i64(v >> 64)] for v in vals], dtype=torch.int64).reshape(len(vals), 2)

        def to_ints(t):
            if not two:
                return [int(x) for x in t.tolist()]
            return [((h & (2 ** 64 - 1)) << 64) | (l & (2 ** 64 - 1)) for l, h in t.tolist()]

        kt = to_tensor(ks)
        ct = torch.tensor([d[x] for x in ks], dtype=torch.int32)
        splitters = gd.uniform_splitters(2 * k, world) if mode == "uniform" else gd.sampled_splitters(kt, world)
        rk, rc, recv = gd.exchange_runs(kt, ct, splitters)
        assert len(recv) == world and sum(recv) == rk.shape[0]
        # every received key belongs to this rank's range, and each run is sorted
        cuts = to_ints(splitters)
        lo = 0 if rank == 0 else cuts[rank - 1]
        hi = (1 << (2 * k)) if rank == world - 1 else cuts[rank]
        got = to_ints(rk)
        assert all(lo <= x < hi for x in got)
        off = 0
        for n in recv:
            seg = got[off:off + n]
            assert seg == sorted(seg)
            off += n
        # merge the runs (sum equal keys)
        m = {}
        for x, c in zip(got, rc.tolist()):
            m[x] = m.get(x, 0) + c
        mk = sorted(m)
        mkt = to_tensor(mk)
        mct = torch.tensor([m[x] for x in mk], dtype=torch.int32)
        ms, M, offset = gd.gather_counts(len(mk), "cpu")
        assert M == len(expect_keys) and offset == sum(ms[:rank])
        ak, ac = gd.gather_ranges_to_root(mkt, mct, ms)
        if rank == 0:
            assert to_ints(ak) == expect_keys
            assert ac.tolist() == expect_counts
        else:
            assert ak.numel() == 0
        q.put((rank, "ok", nwin))
    except Exception as e:            # surface the failure in the parent
        q.put((rank, "fail: %r" % (e,), 0))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,mode", [(13, "uniform"), (40, "uniform"), (13, "sampled"), (40, "sampled")])
def test_range_partition_exchange_world2(k, mode):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    rng = random.Random(31)
    genome = "".join(rng.choice("ACGT") for _ in range(3000))
    reads = [genome[p:p + 60] for p in (rng.randrange(0, 2940) for _ in range(400))]
    shards = ["\n".join(reads[:200]) + "\n", "\n".join(reads[200:]) + "\n"]
    keys, _, nwin = o.collect([(o.LINE, "r", "\n".join(reads) + "\n")], k, 0)
    d = {}
    for x in keys:
        d[x] = d.get(x, 0) + 1
    ek = sorted(d)
    ec = [d[x] for x in ek]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + random.randrange(2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, k, shards, ek, ec, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[1] for r in results) == ["ok", "ok"], results
    assert sum(r[2] for r in results) == nwin


def _spawn(target, world, args, timeout=600):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + random.randrange(2000)
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[1] for r in results) == ["ok"] * world, results
    return results


def _reads(seed, n, genome_len=3000, L=60):
    rng = random.Random(seed)
    genome = "".join(rng.choice("ACGT") for _ in range(genome_len))
    return [genome[p:p + L] for p in (rng.randrange(0, genome_len - L) for _ in range(n))]


@pytest.mark.parametrize("k,mode", [(13, "sampled"), (40, "uniform"), (25, "sampled")])
def test_range_partition_exchange_world8(k, mode):
    """BASELINE config C3's split -- EIGHT ranks -- on the CPU: no process of this test opens a GPU (eight rank
    processes on one card are more than a box of the pool allows, tests/gpu_procs.py).  Same checks as with two ranks:
    every received key inside the rank's range, runs sorted, rank 0's gathered set = the oracle's count of all reads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    reads = _reads(37, 800)
    shards = ["\n".join(reads[i::8]) + "\n" for i in range(8)]
    keys, _, nwin = o.collect([(o.LINE, "r", "\n".join(reads) + "\n")], k, 0)
    d = {}
    for x in keys:
        d[x] = d.get(x, 0) + 1
    ek = sorted(d)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + random.randrange(2000)
    procs = [ctx.Process(target=_worker, args=(r, 8, port, k, shards, ek, [d[x] for x in ek], q, mode)) for r in range(8)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[1] for r in results) == ["ok"] * 8, results
    assert sum(r[2] for r in results) == nwin


def _algebra_worker(rank, world, port, k, shard_sets, op, expect, q):
    """set_algebra_distributed's range logic with the oracle counting: every set counted per rank, range-partitioned
    with the splitters sampled from the FIRST set (the same cuts for every set, so that no further exchange is
    needed), combined range by range, the ranges gathered on rank 0"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    from gossamer_amd import dist as gd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        two = 2 * k > 62

        def to_tensor(vals):
            if not two:
                return torch.tensor(vals, dtype=torch.int64)
            return torch.tensor([[gd._as_i64(v), gd._as_i64(v >> 64)] for v in vals], dtype=torch.int64).reshape(len(vals), 2)

        def to_ints(t):
            if not two:
                return [int(x) for x in t.tolist()]
            return [((h & (2 ** 64 - 1)) << 64) | (l & (2 ** 64 - 1)) for l, h in t.tolist()]

        splitters, ranges, sizes = None, [], []
        for shards in shard_sets:
            keys, _, _ = o.collect([(o.LINE, "r", shards[rank])], k, 0)
            ks = sorted(set(keys))
            kt = to_tensor(ks)
            ct = torch.ones(len(ks), dtype=torch.int32)
            if splitters is None:
                splitters = gd.sampled_splitters(kt, world)
            rk, rc, recv = gd.exchange_runs(kt, ct, splitters)
            mine = sorted(set(to_ints(rk)))
            cuts = to_ints(splitters)
            lo = 0 if rank == 0 else cuts[rank - 1]
            hi = (1 << (2 * k)) if rank == world - 1 else cuts[rank]
            assert all(lo <= x < hi for x in mine)
            ranges.append(mine)
            sizes.append(gd.gather_counts(len(mine), "cpu")[1])
        if op == "intersect":
            res = sorted(set(ranges[0]).intersection(*[set(r) for r in ranges[1:]]))
        else:
            res = sorted(set(ranges[0]) - set(ranges[1]))
        ms, M, first = gd.gather_counts(len(res), "cpu")
        assert first == sum(ms[:rank]) and M == len(expect["keys"])
        assert sizes == expect["sizes"], (sizes, expect["sizes"])
        ak, ac = gd.gather_ranges_to_root(to_tensor(res), torch.ones(len(res), dtype=torch.int32), ms)
        if rank == 0:
            assert to_ints(ak) == expect["keys"]
        q.put((rank, "ok", len(res)))
    except Exception as e:
        q.put((rank, "fail: %r" % (e,), 0))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k", [25, 40])
def test_set_algebra_ranges_world8(k):
    """BASELINE config C5's split over EIGHT ranks on the CPU: two read sets that share half their genome, counted by
    the oracle, cut at common splitters, intersected / subtracted range by range -- rank 0's gathered result = the
    set operation on the oracle's two whole sets.  No GPU in any process."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    rng = random.Random(9)
    shared = "".join(rng.choice("ACGT") for _ in range(1500))
    genomes = [shared + "".join(rng.choice("ACGT") for _ in range(1500)) for _ in range(2)]
    texts = []
    for gi, gen in enumerate(genomes):
        r = random.Random(100 + gi)
        texts.append([gen[p:p + 60] for p in (r.randrange(0, len(gen) - 60) for _ in range(600))])
    whole = []
    for t in texts:
        keys, _, _ = o.collect([(o.LINE, "r", "\n".join(t) + "\n")], k, 0)
        whole.append(set(keys))
    shard_sets = [["\n".join(t[i::8]) + "\n" for i in range(8)] for t in texts]
    for op in ("intersect", "subtract"):
        want = sorted(whole[0] & whole[1]) if op == "intersect" else sorted(whole[0] - whole[1])
        assert want
        results = _spawn(_algebra_worker, 8, (k, shard_sets, op, {"keys": want, "sizes": [len(w) for w in whole]}))
        assert sum(r[2] for r in results) == len(want)


def test_assemble_files_of_eight_ranks():
    """assemble_files: the slices (low-bits columns, ord0) of eight ranks in rank order, everything else from rank 0,
    the .part.* transport files dropped; a rank without a slice (an empty range) contributes nothing."""
    sys.path.insert(0, ROOT)
    from gossamer_amd import dist as gd
    rng = random.Random(3)
    whole = {"-edges.low-bits.lwr": bytes(rng.randrange(256) for _ in range(4000)), "-edges.low-bits.upr": bytes(rng.randrange(256) for _ in range(1000)),
             "-counts.ord0": bytes(rng.randrange(256) for _ in range(1000)), "-edges.high-bits": b"H" * 300, "-edges.header": b"h" * 64,
             "-counts.ord1p.low-bits": b"p" * 7, ".header": b"x" * 24}
    cuts = sorted(rng.randrange(0, 1001) for _ in range(6))
    cuts = [0] + cuts[:3] + [cuts[3]] + cuts[3:] + [1000]          # (one empty range in the middle)
    per_rank = []
    for r in range(8):
        a, b = cuts[r], cuts[r + 1]
        f = {"-edges.low-bits.lwr": whole["-edges.low-bits.lwr"][4 * a:4 * b], "-edges.low-bits.upr": whole["-edges.low-bits.upr"][a:b],
             "-counts.ord0": whole["-counts.ord0"][a:b], ".part.span": b"s" * (r + 1)}
        if a == b:
            f = {".part.span": b""}
        if r == 0:
            f.update({n: d for n, d in whole.items() if n not in f})
        per_rank.append(f)
    assert gd.assemble_files(per_rank) == whole


def test_uniform_splitters_and_split_sizes():
    sys.path.insert(0, ROOT)
    from gossamer_amd import dist as gd
    s = gd.uniform_splitters(50, 8)
    assert s.tolist() == [(1 << 50) * p // 8 for p in range(1, 8)]
    keys = torch.tensor([0, 5, (1 << 47) - 1, 1 << 47, (1 << 49), (1 << 50) - 1], dtype=torch.int64)
    assert gd.split_sizes(keys, s) == [3, 1, 0, 0, 1, 0, 0, 1]
    assert gd.split_sizes(keys, gd.uniform_splitters(50, 1)) == [6]
    assert gd.split_sizes(torch.empty(0, dtype=torch.int64), s) == [0] * 8


def test_two_word_splitters_and_split_sizes():
    """Two-word keys are (lo, hi) pairs of unsigned words: the order must survive the int64
    container (sign bits flipped for the comparison), also for the longest keys (128 bits)."""
    sys.path.insert(0, ROOT)
    from gossamer_amd import dist as gd
    rng = random.Random(5)
    for bits in (64, 80, 112, 128):
        for parts in (1, 2, 8):
            s = gd.uniform_splitters(bits, parts)
            assert tuple(s.shape) == (parts - 1, 2)
            cuts = [((h & (2 ** 64 - 1)) << 64) | (l & (2 ** 64 - 1)) for l, h in s.tolist()]
            assert cuts == [(1 << bits) * p // parts for p in range(1, parts)]
            vals = sorted(set([0, (1 << bits) - 1] + cuts + [c - 1 for c in cuts if c] +
                              [rng.randrange(1 << bits) for _ in range(200)] +
                              [(rng.randrange(1 << (bits - 64)) << 64) | v for v in (0, 2 ** 63, 2 ** 64 - 1) for _ in range(5)]))
            t = torch.tensor([[gd._as_i64(v), gd._as_i64(v >> 64)] for v in vals], dtype=torch.int64).reshape(len(vals), 2)
            edges = [0] + cuts + [1 << bits]
            want = [sum(1 for v in vals if edges[i] <= v < edges[i + 1]) for i in range(parts)]
            assert gd.split_sizes(t, s) == want, (bits, parts)
    assert gd.split_sizes(torch.empty((0, 2), dtype=torch.int64), gd.uniform_splitters(100, 4)) == [0] * 4
