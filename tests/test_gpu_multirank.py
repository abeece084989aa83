"""The multi-GPU path with more than one rank and libgossgpu.so doing the work of every rank.

One MI355X is enough: N processes (2 and 4: tests/gpu_procs.py has the bound) each create a Context on cuda:0 with a small
budget, count their shard of the reads with the HIP kernels, exchange (key,count) runs over a
gloo group (keys staged through host memory, the code path of gossamer_amd.dist is otherwise
the one RCCL runs), merge their range on the device, and the object is assembled and emitted.
Rank 0's files must equal the oracle's single-process build of ALL reads -- BASELINE configs C3
(k-mer set over ranks) and C5 (two sets + intersect / subtract over ranks) at test scale, plus
graphs and two-word keys.  Fresh processes are spawned; nothing re-executes a process that has
touched the GPU.
"""
import os
import random
import struct
import sys

import pytest

import gossamer_amd as g
import gpu_procs

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _suffix_map(files, prefix):
    return {name[len(prefix):]: data for name, data in files.items()}


def _worker(rank, world, port, cases, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import gossamer_amd as gg
    from gossamer_amd import dist as gd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    done = []
    try:
        for case in cases:
            kind, k, name = case["kind"], case["k"], case["name"]
            graph = kind == "graph"
            key_bits = 2 * (k + 1 if graph else k)
            # (the sizes of a piece of records through a gloo group beside the data's group, as under RCCL)
            os.environ["GOSS_DIST_META_GROUP"] = "1" if case.get("meta_group") else "0"
            with gg.Context(k, gg.MODE_GRAPH if graph else gg.MODE_KMER_SET, device=0, hbm_budget=768 << 20) as ctx:
                if kind in ("kmer", "graph"):
                    buf = torch.frombuffer(bytearray(case["shards"][rank]), dtype=torch.uint8).to(dev)
                    r = gd.count_distributed(ctx, buf.data_ptr(), buf.numel(), key_bits, dev, splitters=case.get("splitters", "sampled"),
                                             exchange=case.get("exchange", "counted"), record_pieces=case.get("pieces", 0))
                    assert sum(r["ranges"]) == r["M"]
                    if case.get("balanced"):
                        # sampled splitters: no range far from M / world, whatever the key distribution
                        assert max(r["ranges"]) <= 1.3 * r["M"] / world + 64, r["ranges"]
                    windows = torch.tensor([r["windows"]], dtype=torch.int64)
                    dist.all_reduce(windows)
                    assert int(windows.item()) == case["windows"], (name, int(windows.item()), case["windows"])
                else:
                    bufs = [torch.frombuffer(bytearray(s[rank]), dtype=torch.uint8).to(dev) for s in case["sets"]]
                    r = gd.set_algebra_distributed(ctx, [(b.data_ptr(), b.numel()) for b in bufs], key_bits, kind, dev,
                                                   exchange=case.get("exchange", "counted"))
                    assert r["sizes"] == case["sizes"], (name, r["sizes"], case["sizes"])
                # every rank holds its span of the object: the test plays the per-rank writers
                mine = ctx.files()
                parts = [None] * world
                dist.all_gather_object(parts, mine)
                if rank == 0:
                    assert any(n.startswith(".part.") for n in mine)
                    for other in parts[1:]:
                        assert all(".low-bits" in n or n == "-counts.ord0" or n.startswith(".part.") for n in other), sorted(other)
                    got = gd.assemble_files(parts)
                    exp = case["expect"]
                    assert sorted(got) == sorted(exp), (name, sorted(got), sorted(exp))
                    for f in exp:
                        assert got[f] == exp[f], (name, f)
                    assert r["M"] == case["M"]
            dist.barrier()
            done.append(name)
        q.put((rank, "ok", done))
    except Exception as e:            # surface the failure in the parent
        q.put((rank, "fail after %s: %r" % (done, e), done))
        raise
    finally:
        dist.destroy_process_group()


def _split_reads(text, world):
    """line-aligned shards of a read file (one read per line)"""
    lines = text.split(b"\n")[:-1]
    per = (len(lines) + world - 1) // world
    return [b"".join(l + b"\n" for l in lines[i * per:(i + 1) * per]) for i in range(world)]


def _run(world, cases):
    import torch.multiprocessing as mp
    gpu_procs.check(world)          # (before anything is started: the pool ends a run with more than six holders of the card)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + random.randrange(2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert sorted(r[1] for r in results) == ["ok"] * world, results


def _build_cases(oracle, world):
    cases = []
    reads = g.synth_reads_host(24000, 150, 150000, seed=17)
    for kind, k in (("kmer", 25), ("kmer", 45), ("graph", 27), ("graph", 55)):
        build = oracle.build_graph if kind == "graph" else oracle.build_kmer_set
        exp, nwin = build([(oracle.LINE, "reads", reads)], k, out="ob")
        exp = _suffix_map(exp, "ob")
        M = struct.unpack("<8Q", exp[("-edges" if kind == "graph" else ".kmers") + ".header"])[7]
        cases.append({"kind": kind, "k": k, "name": "%s k=%d" % (kind, k), "shards": _split_reads(reads, world),
                      "expect": exp, "windows": nwin, "M": M, "balanced": True})
    # uniform splitters, and a skewed key distribution (reads of a low-complexity genome: most
    # k-mers start with A) that uniform cuts would leave unbalanced
    cases.append(dict(cases[0], name="kmer k=25 uniform splitters", splitters="uniform", balanced=False))
    rng = random.Random(5)
    genome = "".join(rng.choice("AAAAAAAC" if (i // 40) % 2 == 0 else "ACGT") for i in range(60000))
    skew = "".join(genome[p:p + 120] + "\n" for p in (rng.randrange(0, len(genome) - 120) for _ in range(8000))).encode()
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", skew)], 21, out="ob")
    exp = _suffix_map(exp, "ob")
    cases.append({"kind": "kmer", "k": 21, "name": "kmer k=21 skewed", "shards": _split_reads(skew, world), "expect": exp,
                  "windows": nwin, "M": struct.unpack("<8Q", exp[".kmers.header"])[7], "balanced": True})
    # the exchange BEFORE counting (super-k-mer records routed by minimizer): one-word keys take it, the two-word
    # cases fall back to the exchange of counted runs by themselves
    for c in list(cases):
        # (four rank processes: the record form of the one-word cases and of the two-word graph; two ranks: of all)
        if world > 2 and c["name"] in ("kmer k=45", "kmer k=25 uniform splitters", "kmer k=21 skewed"):
            continue
        cases.append(dict(c, name=c["name"] + ", records", exchange="records"))
    # ... in three pieces: the all-to-all of one piece overlaps the routing of the next and the counting of the one before
    cases.append(dict(cases[0], name="kmer k=25, records in 3 pieces", exchange="records", pieces=3))
    cases.append(dict(cases[2], name="graph k=27, records in 3 pieces", exchange="records", pieces=3))
    cases.append(dict(cases[0], name="kmer k=25, records in 5 pieces, sizes through a side group", exchange="records", pieces=5, meta_group=True))
    return cases


@pytest.mark.parametrize("world", [2, 4])
def test_hip_path_with_several_ranks(oracle, world):
    """C3 at test scale: files of rank 0 == oracle build of all reads (k-mer sets k = 25 / 45, graphs
    k = 27 / 55), every rank counting with the HIP library."""
    _run(world, _build_cases(oracle, world))


@pytest.mark.parametrize("world", [2, 4])
def test_set_algebra_with_several_ranks(oracle, world):
    """C5 at test scale over 2 and 4 rank processes: two k-mer sets counted and range-partitioned with common
    splitters, intersected / subtracted range by range, assembled on rank 0.  (BASELINE's eight-way split: the same
    range logic over eight gloo ranks on the CPU, tests/test_dist_gloo.py, and eight contexts in one process,
    tests/test_gpu_group.py -- eight rank processes on one card are more than a box of the pool allows.)"""
    texts = [g.synth_reads_host(6000, 150, 400000, seed=71, first_read=f) for f in (0, 3000)]
    cases = []
    # (four rank processes: one-word keys with both exchanges, two-word keys once -- every case costs a round of four
    # processes' pushes, exchanges and emissions, and the suite has 600 of the driver's 900 seconds; two ranks: all)
    for k in (25, 45):
        files, names, sizes = {}, [], []
        for i, t in enumerate(texts):
            f, _ = oracle.build_kmer_set([(oracle.LINE, "reads", t)], k, out="s%d" % i)
            files.update(f)
            names.append("s%d" % i)
            sizes.append(struct.unpack("<QQQ", f["s%d.header" % i])[2])
        shards = [_split_reads(t, world) for t in texts]
        for sel, op in (((0, 1), "intersect"), ((0, 1), "subtract"), ((1, 0), "subtract")):
            if world > 2 and ((k == 45 and op != "subtract") or (k == 25 and sel == (1, 0))):
                continue
            if op == "intersect":
                exp = oracle.intersect_kmer_sets(files, [names[j] for j in sel], "out")
            else:
                exp = oracle.subtract_kmer_set(files, names[sel[0]], names[sel[1]], "out")
            exp = _suffix_map(exp, "out")
            cases.append({"kind": op, "k": k, "name": "%s %s k=%d" % (op, sel, k), "sets": [shards[j] for j in sel],
                          "sizes": [sizes[j] for j in sel], "expect": exp, "M": struct.unpack("<QQQ", exp[".header"])[2]})
            if k == 25 and (world == 2 or op == "intersect"):
                cases.append(dict(cases[-1], name=cases[-1]["name"] + ", records", exchange="records"))
    _run(world, cases)
