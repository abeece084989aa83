#!/usr/bin/env python3
"""Randomised parity run on the GPU box: many (k, mode, read shape, error rate, budget) combinations, every one
through the product (fused path forced on where the input is small) and through the oracle; the object files
must be byte-identical.  Complements the fixed cases of tests/: a place to catch what a particular tile count,
key width or table form breaks.  usage: python tests/fuzz_parity.py [cases] [seed] [groups]
("groups": half of the cases count on 2 or 3 contexts of the process and go through the group exchange / emission)"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # (this file lives in tests/: it uses the oracle)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gossamer_amd as g  # noqa: E402
import oracle_lib as oracle  # noqa: E402


def make_reads(rng, nreads, genome_len, lmin, lmax, err, nrate, lower, read_rng=None):
    genome = "".join(rng.choice("ACGT") for _ in range(genome_len))
    # (round 6) what a random genome almost never holds and a real one holds everywhere: homopolymers, short repeats and
    # reverse-complement palindromes, planted into the genome -- the edge T..T beside palindromes is how this script
    # found the strand-pair expansion's pad bug
    if rng.random() < 0.6:
        comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
        g2 = list(genome)
        for _ in range(rng.choice([3, 20, 200])):
            n = rng.randint(8, min(140, max(9, genome_len // 4)))
            kind = rng.randrange(5)
            if kind == 0:
                piece = rng.choice("ACGT") * n
            elif kind == 1:
                piece = (rng.choice(["AT", "CG", "AC", "ACGT", "TTAGGG"]) * n)[:n]
            elif kind == 2:
                half = "".join(rng.choice("ACGT") for _ in range(n // 2))
                piece = half + "".join(comp[c] for c in reversed(half))
            elif kind == 3:
                piece = "A" * (n // 2) + "T" * (n - n // 2)
            else:
                piece = "T" * (n // 2) + "A" * (n - n // 2)
            at = rng.randint(0, max(0, genome_len - len(piece)))
            g2[at:at + len(piece)] = piece
        genome = "".join(g2)[:genome_len]
    if read_rng is not None:          # (fuzz_setops.py: several read sets over ONE genome)
        rng = read_rng
    out = []
    for _ in range(nreads):
        L = rng.randint(lmin, lmax)
        p = rng.randint(0, genome_len - L)
        r = genome[p:p + L]
        if err or nrate or lower:
            b = list(r)
            for i in range(L):
                x = rng.random()
                if x < err:
                    b[i] = rng.choice("ACGT")
                elif x < err + nrate:
                    b[i] = rng.choice("NnRY.-")
                elif lower and rng.random() < 0.2:
                    b[i] = b[i].lower()
            r = "".join(b)
        out.append(r)
    return ("\n".join(out) + "\n").encode()


GROUPS = False       # argument "groups": some cases count on 2 or 3 contexts and exchange
LARGE = False        # argument "large": 0.4 to 1 M reads per case (several chunks, fused path without forcing)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    global GROUPS
    GROUPS = len(sys.argv) > 3 and "groups" in sys.argv[3:]
    global LARGE
    LARGE = "large" in sys.argv[3:]
    rng = random.Random(seed)
    oracle.lib()
    bad = 0
    t00 = time.time()
    for case in range(cases):
        graph = rng.random() < 0.4
        k = rng.choice([11, 13, 15, 16, 21, 24, 25, 27, 30, 31, 32, 33, 40, 45, 55, 62 if graph else 63])
        if graph:
            k = min(k, 62)
        nreads = rng.choice([400000, 1000000]) if LARGE else rng.choice([20000, 60000, 150000])
        lmin, lmax = rng.choice([(150, 150), (30, 200), (k, k + 5), (100, 101)])
        lmin = max(lmin, 1)
        genome_len = rng.choice([5000, 60000, 400000, 2000000])
        genome_len = max(genome_len, lmax + 1)
        err = rng.choice([0.0, 0.0, 0.002, 0.02])
        nrate = rng.choice([0.0, 0.001, 0.01])
        reads = make_reads(rng, nreads, genome_len, lmin, lmax, err, nrate, rng.random() < 0.3)
        env = {} if LARGE and rng.random() < 0.5 else {"GOSS_GPU_FUSED_MIN": "0"}
        if rng.random() < 0.3:
            env["GOSS_GPU_NO_MSD"] = "1"
        if rng.random() < 0.15:
            env["GOSS_GPU_NO_FUSED"] = "1"
        if rng.random() < 0.2:
            env["GOSS_GPU_EST_SCALE"] = rng.choice(["0.3", "3.0"])
        if rng.random() < 0.15:
            env["GOSS_GPU_OVERFLOW_BY_SORT"] = "0"          # (the ladder of forms instead of the overflowed segments counted by sort)
        if rng.random() < 0.2:
            env["GOSS_GPU_ORDER_BITS"] = rng.choice(["16", "17", "20"])
        # the 32-bit-remainder forms (second-level bits, third level, table) and the first level's key space
        if rng.random() < 0.15:
            env["GOSS_GPU_NO_REM32"] = "1"
        if rng.random() < 0.25:
            env["GOSS_GPU_REM32_BITS"] = "10"
        if rng.random() < 0.3:
            env["GOSS_GPU_REM32_SPLIT"] = rng.choice(["1", "2", "3", "4"])
        if rng.random() < 0.2:
            env["GOSS_GPU_REM32_SLOTS"] = rng.choice(["2048", "4096"])
        if rng.random() < 0.25:
            env["GOSS_GPU_CANON_L1"] = rng.choice(["0", "2"])
        budget = rng.choice([512 << 20, 2 << 30, 8 << 30]) if LARGE else rng.choice([96 << 20, 256 << 20, 1 << 30, 4 << 30])
        t0 = time.time()
        build = oracle.build_graph if graph else oracle.build_kmer_set
        exp, nwin = build([(oracle.LINE, "r", reads)], k, out="o")
        exp = {n[1:]: b for n, b in exp.items()}
        old = {n: os.environ.get(n) for n in env}
        os.environ.update(env)
        parts = rng.choice([1, 1, 2, 3]) if GROUPS else 1
        try:
            try:
                # one-word keys, some cases: the reads go through the routing kernel (super-k-mer records for 1 .. 8
                # parts) and the context counts the records of all parts -- the exchange before counting, on one GPU
                via_records = parts == 1 and rng.random() < 0.4
                rb = g.binding.record_bytes(k, 1 if graph else 0)          # (12-byte records for one-word keys, 20-byte for two-word keys)
                devices = GROUPS and parts > 1 and rng.random() < 0.25
                if devices:
                    # the command line on several "devices" (this GPU several times): from four on the exchange before counting
                    import subprocess
                    import tempfile
                    d = tempfile.mkdtemp(prefix="goss_fuzz_")
                    try:
                        with open(os.path.join(d, "r.txt"), "wb") as f:
                            f.write(reads)
                        ndev = rng.choice([2, 4, 5])
                        e2 = dict(os.environ, GOSS_GPU_STAGE_CAP=str(rng.choice([1 << 20, 8 << 20, 1 << 30])), GOSS_PARSE_CHUNK="262144")
                        p = subprocess.run([os.path.join(ROOT, "gossamer_amd", "goss"), "build-graph" if graph else "build-kmer-set", "-k", str(k),
                                            "--line-in", os.path.join(d, "r.txt"), "-O", os.path.join(d, "o"), "--hbm-budget", "1", "-T", "4", "-v",
                                            "--devices", ",".join(["0"] * ndev)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=e2)
                        if p.returncode != 0 and b"out of memory (HBM budget)" in p.stderr:
                            # (--hbm-budget 1: a legitimate refusal, as for the library's own -3 below)
                            raise g.GossGpuError(-3, "out of memory (HBM budget)", "goss --devices")
                        if p.returncode != 0:
                            raise RuntimeError("goss --devices failed: " + p.stderr.decode(errors="replace")[-400:])
                        got = {n[1:]: open(os.path.join(d, n), "rb").read() for n in os.listdir(d) if n.startswith("o.") or n.startswith("o-")}
                        import re
                        m = re.search(rb"k-mer windows: (\d+),", p.stderr)
                        windows = int(m.group(1)) if m else -1
                        stats = {"devices": ndev, "records": b"records routed by minimizer" in p.stderr}
                        if ndev >= 4 and not stats["records"]:
                            raise RuntimeError("four devices and no exchange before counting")
                    finally:
                        import shutil
                        shutil.rmtree(d, ignore_errors=True)
                elif via_records:
                    import torch
                    nparts = rng.choice([1, 2, 3, 5, 8])
                    dev = torch.device("cuda", 0)
                    with g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=budget) as ctx:
                        bases = torch.frombuffer(bytearray(reads), dtype=torch.uint8).to(dev)
                        need = [1] * nparts
                        for attempt in range(2):
                            first = [sum(need[:p]) for p in range(nparts)]
                            buf = torch.empty(sum(need) * rb, dtype=torch.uint8, device=dev)
                            recs, wins, okr = ctx.route_records(bases.data_ptr(), bases.numel(), nparts, buf.data_ptr(), first, need)
                            need = recs
                        assert okr
                        if rng.random() < 0.5:
                            ctx.push_records(buf.data_ptr(), sum(recs), sum(wins))          # (exact sizes: the parts lie back to back)
                        else:
                            for p in range(nparts):
                                if recs[p]:
                                    ctx.push_records(buf.data_ptr() + first[p] * rb, recs[p], wins[p])
                        c = ctx.finish()
                        got = ctx.emit()
                        windows = c.windows
                        stats = {"records": sum(recs), "parts": nparts, "rec_chunks": ctx.stat("rec_chunks"), "fused_chunks": ctx.stat("fused_chunks")}
                elif parts == 1:
                    with g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=budget) as ctx:
                        # (round 6) what the caller says it will push -- nothing, the truth, far too little, far too much: a
                        # hint for the key space the chunks count in, never for the result; and how the bases arrive: one
                        # host push, several, packed on the host, or packed on the device and counted where they lie
                        hint = rng.choice([None, None, 1.0, 0.05, 30.0])
                        if hint is not None:
                            ctx.expect_bases(int(len(reads) * hint))
                        how = rng.choice(["host", "host", "pieces", "packed-host", "packed-device"])
                        if how == "host":
                            ctx.push_host(reads)
                        elif how == "packed-device":
                            import torch
                            buf = torch.frombuffer(bytearray(reads), dtype=torch.uint8).cuda()
                            groups = (len(reads) + 15) // 16
                            dc = torch.empty(groups, dtype=torch.int32, device="cuda")
                            db = torch.empty(groups, dtype=torch.int16, device="cuda")
                            ctx.pack_bases_device(buf.data_ptr(), len(reads), dc.data_ptr(), db.data_ptr())
                            del buf
                            ctx.push_packed_device(dc.data_ptr(), db.data_ptr(), len(reads))
                        else:
                            lines = reads.split(b"\n")[:-1]
                            npieces = rng.choice([2, 3, 7])
                            per = (len(lines) + npieces - 1) // npieces
                            for i in range(0, len(lines), per):
                                piece = b"".join(x + b"\n" for x in lines[i:i + per])
                                if how == "packed-host":
                                    ctx.push_packed_host(piece, async_=rng.random() < 0.5)
                                else:
                                    ctx.push_host(piece)
                        c = ctx.finish()
                        got = ctx.emit()
                        windows = c.windows
                        stats = {s: ctx.stat(s) for s in ("fused_chunks", "fused_msd_chunks", "big_table_chunks", "segment_retries", "rep_chunks", "canon_chunks", "runs")}
                        stats.update(hint=hint, how=how)
                else:
                    # several contexts of this process, each counting a share of the reads (goss_gpu_group_exchange / _emit);
                    # half of these cases exchange BEFORE counting: deferred contexts, the shards pushed in 1 .. 3 portions
                    # with a goss_gpu_group_route_exchange behind each
                    from gossamer_amd import dist as gd
                    lines = reads.split(b"\n")[:-1]
                    per = (len(lines) + parts - 1) // parts
                    shards = [b"".join(x + b"\n" for x in lines[i * per:(i + 1) * per]) for i in range(parts)]
                    ctxs = [g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=max(budget, 256 << 20)) for _ in shards]
                    before = rng.random() < 0.5
                    try:
                        windows = 0
                        if before:
                            rounds = rng.choice([1, 2, 3])
                            for cx in ctxs:
                                cx.set_deferred(True)
                            for r in range(rounds):
                                for cx, sh in zip(ctxs, shards):
                                    piece = b"".join(x + b"\n" for x in sh.split(b"\n")[:-1][r::rounds])
                                    if piece:
                                        cx.push_host(piece)
                                if r + 1 < rounds or rng.random() < 0.7:          # (what is still staged at finish is counted locally)
                                    g.group_route_exchange(ctxs)
                        for cx, sh in zip(ctxs, shards):
                            if sh and not before:
                                cx.push_host(sh)
                            windows += cx.finish().windows
                        g.group_exchange(ctxs, sample_per_context=rng.choice([0, 64, 1000]))
                        g.group_emit(ctxs)
                        got = gd.assemble_files([cx.files() for cx in ctxs])
                        stats = {"contexts": parts, "exchange": "before counting" if before else "counted"}
                    finally:
                        for cx in ctxs:
                            cx.close()
                ok = windows == nwin and sorted(got) == sorted(exp) and all(got[n] == exp[n] for n in exp)
                why = "" if ok else "windows %d/%d, files differ: %s" % (windows, nwin, [n for n in exp if got.get(n) != exp[n]][:4])
            except g.GossGpuError as e:
                # a budget too small for the input is a legitimate refusal -- and so is a deferred context's "this push does
                # not fit the staging buffer" (-9: the caller of goss_gpu_group_route_exchange cuts its pushes by
                # goss_gpu_stage_room; this script pushes whole shards) -- anything else is not
                ok = e.status == -3 or (e.status == -9 and "deferred" in str(e))
                why = "refused: %s" % e
                stats = {}
        finally:
            for n, v in old.items():
                if v is None:
                    os.environ.pop(n, None)
                else:
                    os.environ[n] = v
        bad += 0 if ok else 1
        if not ok and os.environ.get("FUZZ_DUMP_DIR"):
            # (a failing case's input and settings, for a replay outside the random sequence: tools/dbg/fuzz_replay.py)
            import json
            dd = os.environ["FUZZ_DUMP_DIR"]
            os.makedirs(dd, exist_ok=True)
            with open(os.path.join(dd, "case%d.reads" % case), "wb") as f:
                f.write(reads)
            with open(os.path.join(dd, "case%d.json" % case), "w") as f:
                json.dump({"graph": graph, "k": k, "budget": budget, "env": env, "stats": {a: (b if isinstance(b, (int, float, str, type(None))) else str(b)) for a, b in stats.items()}, "why": why}, f)
        print("%s case %d: %s k=%d reads=%d len=%d..%d genome=%d err=%g n=%g budget=%dM env=%s windows=%d %s %.1fs %s"
              % ("ok  " if ok else "FAIL", case, "graph" if graph else "kmer", k, nreads, lmin, lmax, genome_len, err, nrate, budget >> 20,
                 {a: b for a, b in env.items() if a != "GOSS_GPU_FUSED_MIN"}, nwin, stats, time.time() - t0, why), flush=True)
    print("fuzz: %d cases, %d failed, %.0f s" % (cases, bad, time.time() - t00))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
