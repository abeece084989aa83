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
        # DenseSelect blocks per range (round 5): the ranges built the blocks inside their own ones / zeros, the assembling
        # context only those that straddle two ranges (at most one per range boundary and sense, and the arrays' last)
        built, assembled = ctxs[0].stat("ds_blocks_from_ranges"), ctxs[0].stat("ds_blocks_assembled")
        if sum(sizes) >= 8192 * 4 * len(ctxs):
            assert built > 0 and assembled <= 2 * len(ctxs) + 2, (built, assembled, sizes)
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


def _record_group_build(shards, k, mode, rounds=2, transport=0, stage_cap=None):
    """the exchange before counting: deferred contexts, the shards pushed in `rounds` portions with a
    goss_gpu_group_route_exchange behind each, then finish / range exchange / emission as above"""
    import os
    old = os.environ.get("GOSS_GPU_STAGE_CAP")
    if stage_cap:
        os.environ["GOSS_GPU_STAGE_CAP"] = str(stage_cap)
    ctxs = [g.Context(k, mode, hbm_budget=768 << 20) for _ in shards]
    try:
        for c in ctxs:
            c.set_deferred(True)
        stats = []
        pieces = [[b"".join(l + b"\n" for l in s.split(b"\n")[:-1][r::rounds]) for r in range(rounds)] for s in shards]
        for r in range(rounds):
            for c, ps in zip(ctxs, pieces):
                if ps[r]:
                    c.push_host(ps[r])
            stats.append(g.group_route_exchange(ctxs, transport))
        windows = sum(c.finish().windows for c in ctxs)
        rec_chunks = [c.stat("rec_chunks") + c.stat("fused_chunks") for c in ctxs]
        # the members' counted sets are disjoint: no key in two of them
        seen = set()
        for c in ctxs:
            ks, _ = c.result()
            assert not (seen & set(ks))
            seen |= set(ks)
        sizes = g.group_exchange(ctxs, sample_per_context=256)
        g.group_emit(ctxs)
        return gd.assemble_files([c.files() for c in ctxs]), sizes, windows, stats, rec_chunks
    finally:
        for c in ctxs:
            c.close()
        if old is None:
            os.environ.pop("GOSS_GPU_STAGE_CAP", None)
        else:
            os.environ["GOSS_GPU_STAGE_CAP"] = old


@pytest.mark.parametrize("kind,k,parts", [("kmer", 25, 4), ("graph", 27, 4), ("kmer", 21, 3), ("graph", 30, 2), ("kmer", 31, 8),
                                          ("kmer", 45, 4), ("graph", 55, 4), ("graph", 31, 3), ("kmer", 63, 2),
                                          # BASELINE's eight-way shapes (C3: k = 25; C4's key width over eight members) in one process
                                          ("kmer", 25, 8), ("graph", 55, 8)])
def test_group_route_exchange_builds_the_oracles_object(oracle, kind, k, parts):
    """goss_gpu_group_route_exchange (what `goss --devices` with four and more devices drives): the members' reads cut
    into records routed by minimizer, part p counted by member p -- here all on cuda:0, so the parts travel by peer
    copies of the one device.  Files equal to the oracle's single build of all reads."""
    reads = g.synth_reads_host(18000, 150, 120000, seed=31)
    build = oracle.build_graph if kind == "graph" else oracle.build_kmer_set
    exp, nwin = build([(oracle.LINE, "reads", reads)], k, out="ob")
    exp = _suffix_map(exp, "ob")
    got, sizes, windows, stats, _ = _record_group_build(_split_reads(reads, parts), k, g.MODE_GRAPH if kind == "graph" else g.MODE_KMER_SET)
    assert windows == nwin == sum(st["windows"] for st in stats)
    assert all(st["transport"] == 2 and st["records"] > 0 for st in stats), stats          # (one device several times: no communicators)
    assert sorted(got) == sorted(exp)
    for name in exp:
        assert got[name] == exp[name], name


@pytest.mark.parametrize("k", [25, 45])
def test_set_algebra_over_eight_members(oracle, k):
    """BASELINE config C5's eight-way split in ONE process (eight rank processes on one card are more than a box of the
    pool allows): both read sets go through goss_gpu_group_route_exchange over the same eight members -- a key's member
    is a function of its minimizer, so member p holds class p of set A and class p of set B and combines them alone
    (weighted runs + goss_gpu_select_counts, as the single-GPU commands and set_algebra_distributed do) -- then the
    range exchange of the results and the distributed emission.  Files equal to the oracle's intersect-kmer-sets /
    subtract-kmer-set (GossCmdIntersectKmerSets.cc:29-128, GossCmdSubtractKmerSet.cc:32-85)."""
    import torch
    parts = 8
    texts = [g.synth_reads_host(6000, 150, 400000, seed=71, first_read=f) for f in (0, 3000)]
    files, names = {}, []
    for i, t in enumerate(texts):
        f, _ = oracle.build_kmer_set([(oracle.LINE, "reads", t)], k, out="s%d" % i)
        files.update(f)
        names.append("s%d" % i)
    words = 2 if 2 * k > 62 else 1
    classes = []          # classes[set][member] = the member's sorted keys of that set (a tensor on the device)
    for t in texts:
        ctxs = [g.Context(k, g.MODE_KMER_SET, hbm_budget=512 << 20) for _ in range(parts)]
        try:
            for c in ctxs:
                c.set_deferred(True)
            for c, s in zip(ctxs, _split_reads(t, parts)):
                c.push_host(s)
            g.group_route_exchange(ctxs)
            mine = []
            for c in ctxs:
                c.finish()
                kp, _, m = c.result_ptrs()
                mine.append(gd.key_view(kp, m, words, "cuda").clone())
            classes.append(mine)
        finally:
            for c in ctxs:
                c.close()
    assert sum(x.shape[0] for x in classes[0]) == struct.unpack("<QQQ", files["s0.header"])[2]
    for sel, op in (((0, 1), "intersect"), ((0, 1), "subtract"), ((1, 0), "subtract")):
        if op == "intersect":
            exp = oracle.intersect_kmer_sets(files, [names[j] for j in sel], "out")
            weights, keep = (1, 1), 2
        else:
            exp = oracle.subtract_kmer_set(files, names[sel[0]], names[sel[1]], "out")
            weights, keep = (1, 2), 1
        exp = _suffix_map(exp, "out")
        ctxs = [g.Context(k, g.MODE_KMER_SET, hbm_budget=512 << 20) for _ in range(parts)]
        try:
            for p, c in enumerate(ctxs):
                held = []
                for j, w in zip(sel, weights):
                    keys = classes[j][p]
                    if keys.shape[0]:
                        held.append((keys, torch.full((keys.shape[0],), w, dtype=torch.int32, device="cuda")))
                torch.cuda.synchronize()
                for keys, w in held:
                    c.push_run(keys.data_ptr(), w.data_ptr(), keys.shape[0])
                c.finish()
                c.select_counts(keep, keep)
            sizes = g.group_exchange(ctxs, sample_per_context=256)
            g.group_emit(ctxs)
            got = gd.assemble_files([c.files() for c in ctxs])
        finally:
            for c in ctxs:
                c.close()
        assert sum(sizes) == struct.unpack("<QQQ", exp[".header"])[2] > 0, (sel, op)
        assert sorted(got) == sorted(exp), (sel, op)
        for name in exp:
            assert got[name] == exp[name], (sel, op, name)


@pytest.mark.parametrize("kind,k", [("kmer", 25), ("kmer", 45), ("graph", 27)])
def test_group_emit_with_empty_ranges(oracle, kind, k):
    """Ranges that are EMPTY when the object is emitted (a set operation whose result has nothing in a range, splitters
    that leave the ranges behind the last key without one): an empty range builds no DenseSelect block and the tail of
    "-d0" -- the zeros behind the last key, thousands of them -- belongs to the last NON-empty range alone (round 5
    let every empty range behind it build that tail again: "a block built twice").  And an object with no key at all
    over several members: every block comes from the assembler."""
    reads = g.synth_reads_host(9000, 150, 60000, seed=41)
    graph = kind == "graph"
    exp, _ = (oracle.build_graph if graph else oracle.build_kmer_set)([(oracle.LINE, "reads", reads)], k, out="ob")
    exp = _suffix_map(exp, "ob")
    mode = g.MODE_GRAPH if graph else g.MODE_KMER_SET
    for layout in ("full,empty,empty", "empty,full,empty", "empty,empty,full,empty"):
        ctxs = [g.Context(k, mode, hbm_budget=512 << 20) for _ in layout.split(",")]
        try:
            for c, what in zip(ctxs, layout.split(",")):
                if what == "full":
                    c.push_host(reads)
                c.finish()
            g.group_emit(ctxs)
            got = gd.assemble_files([c.files() for c in ctxs])
        finally:
            for c in ctxs:
                c.close()
        assert sorted(got) == sorted(exp), layout
        for name in exp:
            assert got[name] == exp[name], (layout, name)


def test_group_emit_of_an_empty_object(oracle):
    """intersect of two disjoint k-mer sets over several members: no member holds a key, total = 0 -- the oracle's
    empty KmerSet files (every DenseSelect block built by the assembling member; round 5: every member built block 0)."""
    a = g.synth_reads_host(300, 150, 20000, seed=3)
    b = g.synth_reads_host(300, 150, 20000, seed=4)
    files = {}
    for name, t in (("a", a), ("b", b)):
        f, _ = oracle.build_kmer_set([(oracle.LINE, "reads", t)], 25, out=name)
        files.update(f)
    exp = _suffix_map(oracle.intersect_kmer_sets(files, ["a", "b"], "out"), "out")
    assert struct.unpack("<QQQ", exp[".header"])[2] == 0
    for members in (1, 2, 5):
        ctxs = [g.Context(25, g.MODE_KMER_SET, hbm_budget=256 << 20) for _ in range(members)]
        try:
            for c in ctxs:
                c.finish()
            g.group_emit(ctxs)
            got = gd.assemble_files([c.files() for c in ctxs])
        finally:
            for c in ctxs:
                c.close()
        assert sorted(got) == sorted(exp), members
        for name in exp:
            assert got[name] == exp[name], (members, name)


def test_group_emit_without_threads_to_be_had(oracle, tmp_path):
    """goss_gpu_group_emit starts a host thread per member; when none can be started (RLIMIT_NPROC reached: std::thread's
    constructor throws std::system_error) nothing may cross the C boundary or be left joinable -- the members' parts are
    built on the caller's thread, same files.  In a child process, the limit lowered after the contexts were made.  (A
    privileged user is not bound by the limit: the call then simply runs with its threads.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    reads = g.synth_reads_host(6000, 150, 50000, seed=43)
    exp, _ = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], 25, out="ob")
    exp = _suffix_map(exp, "ob")
    (tmp_path / "reads.txt").write_bytes(reads)
    script = """
import os, resource, sys
sys.path.insert(0, %r)
import gossamer_amd as g
from gossamer_amd import dist as gd
reads = open(%r, "rb").read()
lines = reads.split(b"\\n")[:-1]
ctxs = [g.Context(25, g.MODE_KMER_SET, hbm_budget=256 << 20) for _ in range(3)]
for i, c in enumerate(ctxs):
    c.push_host(b"".join(l + b"\\n" for l in lines[i::3]))
    c.finish()
g.group_exchange(ctxs, sample_per_context=256)
soft, hard = resource.getrlimit(resource.RLIMIT_NPROC)
resource.setrlimit(resource.RLIMIT_NPROC, (1, hard))
g.group_emit(ctxs)
resource.setrlimit(resource.RLIMIT_NPROC, (soft, hard))
for name, data in gd.assemble_files([c.files() for c in ctxs]).items():
    open(os.path.join(%r, "out" + name), "wb").write(data)
""" % (root, str(tmp_path / "reads.txt"), str(tmp_path))
    p = subprocess.run([sys.executable, "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    got = {n[3:]: (tmp_path / n).read_bytes() for n in os.listdir(tmp_path) if n.startswith("out")}
    assert sorted(got) == sorted(exp)
    for name in exp:
        assert got[name] == exp[name], name


@pytest.mark.parametrize("packed", [False, True])
def test_deferred_context_reports_a_full_staging_buffer(oracle, packed):
    """A deferred context refuses the push that does not fit (nothing of it taken) and says how much fits; what is
    staged when it is finished without an exchange is counted locally -- same result.  packed: the 2-bit form of the
    push, whose first piece would fit into what is left of the buffer (>= 64 KB) while the whole push does not -- a
    refusal after that piece was staged would count its windows twice when the caller pushes the batch again."""
    import os
    reads = g.synth_reads_host(4000, 150, 50000, seed=5)
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], 25, out="o")
    os.environ["GOSS_GPU_STAGE_CAP"] = str(256 << 10)
    try:
        with g.Context(25, 0, hbm_budget=512 << 20) as ctx:
            ctx.set_deferred(True)
            room, cap = ctx.stage_room()
            assert cap == 256 << 10 and 0 < room <= cap
            lines = reads.split(b"\n")[:-1]
            chunk = b"".join(l + b"\n" for l in lines[:1000])          # 151 000 bytes
            push = ctx.push_packed_host if packed else ctx.push_host
            push(chunk)
            room2, _ = ctx.stage_room()
            if packed:
                assert room - len(chunk) - 64 <= room2 <= room - len(chunk) and room2 >= 65536          # (whole groups + a group of separators)
            else:
                assert room2 == room - len(chunk)          # (the chunk ends with its own separator)
            with pytest.raises(g.GossGpuError) as e:
                push(chunk)
            assert e.value.status == -9
            assert ctx.stage_room()[0] == room2
            st = g.group_route_exchange([ctx])
            assert 990 * 126 < st["windows"] <= 1000 * 126 and ctx.stage_room()[0] == room          # (a read in 97 holds an N)
            for i in range(1, 4):
                push(b"".join(l + b"\n" for l in lines[1000 * i:1000 * (i + 1)]))
                if i < 3:
                    g.group_route_exchange([ctx])
            c = ctx.finish()          # (the last thousand reads are still staged: counted here)
            assert c.windows == nwin
            got = ctx.emit()
        assert {n: d for n, d in got.items()} == {n[1:]: d for n, d in exp.items()}
    finally:
        os.environ.pop("GOSS_GPU_STAGE_CAP", None)


def test_group_route_exchange_over_rccl_with_one_member(oracle):
    """The RCCL transport (librccl.so loaded at run time, ncclCommInitAll, ncclSend / ncclRecv inside one group call) on
    the one GPU a test box has: a group of one member sends its single part to itself.  Two rounds; files against the
    oracle.  (More than one RCCL rank needs more than one GPU: not here.)"""
    reads = g.synth_reads_host(12000, 150, 90000, seed=77)
    exp, nwin = oracle.build_kmer_set([(oracle.LINE, "reads", reads)], 25, out="ob")
    exp = _suffix_map(exp, "ob")
    got, sizes, windows, stats, _ = _record_group_build([reads], 25, g.MODE_KMER_SET, rounds=2, transport=1)
    assert windows == nwin and all(st["transport"] == 1 for st in stats), stats
    for name in exp:
        assert got[name] == exp[name], name


def test_goss_devices_option(oracle, tmp_path):
    """`goss build-kmer-set / build-graph --devices 0,0[,0]`: one context per listed device (here the same GPU
    several times), batches dealt round the contexts by feeder threads, ranges exchanged, every context's slices
    written behind one another -- files byte for byte the oracle's; FASTQ parsed in parallel chunks whose buffers
    are handed over to the feeders (GOSS_PARSE_CHUNK makes the chunks small), plus FASTA and line input."""
    import os
    import random
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    goss = os.path.join(root, "gossamer_amd", "goss")
    rng = random.Random(5)
    genome = "".join(rng.choice("ACGT") for _ in range(40000))
    reads = []
    for _ in range(9000):
        L = rng.randint(40, 150)
        p = rng.randint(0, len(genome) - L)
        reads.append(genome[p:p + L])
    fq = "".join("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r)) for i, r in enumerate(reads[:8000]))
    fa = "".join(">r%d\n%s\n" % (i, r) for i, r in enumerate(reads[8000:8500]))
    ln = "\n".join(reads[8500:]) + "\n"
    (tmp_path / "a.fq").write_text(fq)
    (tmp_path / "b.fa").write_text(fa)
    (tmp_path / "c.txt").write_text(ln)
    inputs = [(oracle.LINE, "c.txt", ln), (oracle.FASTA, "b.fa", fa), (oracle.FASTQ, "a.fq", fq)]
    env = dict(os.environ, GOSS_PARSE_CHUNK="65536")
    # (four devices and more: the exchange before counting -- records routed by minimizer, 12 bytes for one-word keys
    # and 20 for two-word keys; a staging buffer of 256 KB makes the 1 MB of reads take several exchange rounds; fewer
    # devices: the counted ranges are exchanged)
    for cmd, k, obuild, base, devices, records in (("build-kmer-set", 25, oracle.build_kmer_set, "ks", "0,0", False),
                                                   ("build-graph", 27, oracle.build_graph, "gr", "0,0,0", False),
                                                   ("build-graph", 55, oracle.build_graph, "g55", "0,0", False),
                                                   ("build-kmer-set", 25, oracle.build_kmer_set, "ks4", "0,0,0,0", True),
                                                   ("build-graph", 27, oracle.build_graph, "gr4", "0,0,0,0", True),
                                                   ("build-graph", 55, oracle.build_graph, "g554", "0,0,0,0", True)):
        exp, nwin = obuild(inputs, k, out=base)
        out = tmp_path / base
        p = subprocess.run([goss, cmd, "-k", str(k), "-i", str(tmp_path / "a.fq"), "-I", str(tmp_path / "b.fa"),
                            "--line-in", str(tmp_path / "c.txt"), "-O", str(out), "--hbm-budget", "1", "-T", "4", "-v",
                            "--devices", devices],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=dict(env, GOSS_GPU_STAGE_CAP=str(256 << 10)))
        assert p.returncode == 0, p.stderr.decode()
        assert ("counted on %d devices" % len(devices.split(","))).encode() in p.stderr
        assert ("k-mer windows: %d," % nwin).encode() in p.stderr
        assert (b"records routed by minimizer" in p.stderr) == records, p.stderr.decode()
        if records:
            import re
            m = re.search(rb"\((\d+) windows\) exchanged in (\d+) round\(s\) over peer copies", p.stderr)
            assert m and int(m.group(1)) == nwin and int(m.group(2)) >= 2, p.stderr.decode()
        got = {n: (tmp_path / n).read_bytes() for n in os.listdir(tmp_path) if n.startswith(base + ".") or n.startswith(base + "-")}
        assert sorted(got) == sorted(exp)
        for name in exp:
            assert got[name] == exp[name], name
    p = subprocess.run([goss, "build-kmer-set", "-k", "25", "-i", str(tmp_path / "a.fq"), "-O", str(tmp_path / "x"), "--devices", "0,x"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode != 0 and b"--devices" in p.stderr
