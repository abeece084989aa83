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


def _worker(rank, world, port, k, shards, expect_keys, expect_counts, q):
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
        kt = torch.tensor(ks, dtype=torch.int64)
        ct = torch.tensor([d[x] for x in ks], dtype=torch.int32)
        splitters = gd.uniform_splitters(2 * k, world)
        rk, rc, recv = gd.exchange_runs(kt, ct, splitters)
        assert len(recv) == world and sum(recv) == rk.numel()
        # every received key belongs to this rank's range, and each run is sorted
        lo = 0 if rank == 0 else int(splitters[rank - 1])
        hi = (1 << (2 * k)) if rank == world - 1 else int(splitters[rank])
        assert all(lo <= int(x) < hi for x in rk.tolist())
        off = 0
        for n in recv:
            seg = rk[off:off + n].tolist()
            assert seg == sorted(seg)
            off += n
        # merge the runs (sum equal keys)
        m = {}
        for x, c in zip(rk.tolist(), rc.tolist()):
            m[x] = m.get(x, 0) + c
        mk = sorted(m)
        mkt = torch.tensor(mk, dtype=torch.int64)
        mct = torch.tensor([m[x] for x in mk], dtype=torch.int32)
        ms, M, offset = gd.gather_counts(len(mk), "cpu")
        assert M == len(expect_keys) and offset == sum(ms[:rank])
        ak, ac = gd.gather_ranges_to_root(mkt, mct, ms)
        if rank == 0:
            assert ak.tolist() == expect_keys
            assert ac.tolist() == expect_counts
        else:
            assert ak.numel() == 0
        q.put((rank, "ok", nwin))
    except Exception as e:            # surface the failure in the parent
        q.put((rank, "fail: %r" % (e,), 0))
        raise
    finally:
        dist.destroy_process_group()


def test_range_partition_exchange_world2():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    rng = random.Random(31)
    k = 13
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
    procs = [ctx.Process(target=_worker, args=(r, 2, port, k, shards, ek, ec, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[1] for r in results) == ["ok", "ok"], results
    assert sum(r[2] for r in results) == nwin


def test_uniform_splitters_and_split_sizes():
    sys.path.insert(0, ROOT)
    from gossamer_amd import dist as gd
    s = gd.uniform_splitters(50, 8)
    assert s.tolist() == [(1 << 50) * p // 8 for p in range(1, 8)]
    keys = torch.tensor([0, 5, (1 << 47) - 1, 1 << 47, (1 << 49), (1 << 50) - 1], dtype=torch.int64)
    assert gd.split_sizes(keys, s) == [3, 1, 0, 0, 1, 0, 0, 1]
    assert gd.split_sizes(keys, gd.uniform_splitters(50, 1)) == [6]
    assert gd.split_sizes(torch.empty(0, dtype=torch.int64), s) == [0] * 8
