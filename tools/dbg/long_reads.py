#!/usr/bin/env python3
"""Sequences far longer than a read (a chromosome on one line, FASTA wrapped at 70 columns, a FASTQ record of 5 Mbp
among short ones): library with small arenas and the command line, files against the oracle's.  GPU box."""
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gossamer_amd as g  # noqa: E402
import oracle_lib as oracle  # noqa: E402

GOSS = os.path.join(ROOT, "gossamer_amd", "goss")
rng = random.Random(3)
bad = 0


def seq(n, nrate=0.0):
    s = "".join(rng.choice("ACGT") for _ in range(n))
    if nrate:
        b = list(s)
        for _ in range(int(n * nrate)):
            b[rng.randrange(n)] = "N"
        s = "".join(b)
    return s


def check(tag, got, exp):
    global bad
    ok = sorted(got) == sorted(exp) and all(got[n] == exp[n] for n in exp)
    bad += 0 if ok else 1
    print("%s %s%s" % ("ok  " if ok else "FAIL", tag, "" if ok else " " + str([n for n in exp if got.get(n) != exp[n]][:4])), flush=True)


chrom = seq(30_000_000, 1e-5)
line = (chrom + "\n" + seq(1000) + "\n" + chrom[5_000_000:9_000_000] + "\n").encode()
# (arenas a few times the counted set: 34 M distinct k-mers, twice as many edges -- the set itself must fit)
for k, graph, budget in ((25, False, 1536 << 20), (25, False, 4 << 30), (27, True, 3 << 30), (45, False, 2 << 30), (55, True, 5 << 30)):
    build = oracle.build_graph if graph else oracle.build_kmer_set
    exp, nwin = build([(oracle.LINE, "r", line)], k, out="o")
    exp = {n[1:]: b for n, b in exp.items()}
    for how in ("host", "pieces"):
        try:
            with g.Context(k, g.MODE_GRAPH if graph else g.MODE_KMER_SET, hbm_budget=budget) as ctx:
                if how == "host":
                    ctx.push_host(line)
                else:
                    # the caller cuts at line ends only: pieces of one line each
                    for piece in line.split(b"\n")[:-1]:
                        ctx.push_host(piece + b"\n")
                c = ctx.finish()
                got = ctx.emit()
                st = {s: ctx.stat(s) for s in ("fused_chunks", "runs")}
            check("library k=%d %s budget=%dM %s windows %d/%d %s" % (k, "graph" if graph else "kmer", budget >> 20, how, c.windows, nwin, st), got, exp)
            if c.windows != nwin:
                bad += 1
        except g.GossGpuError as e:
            print("%s library k=%d budget=%dM %s refused: %s" % ("ok  " if e.status == -3 else "FAIL", k, budget >> 20, how, e), flush=True)
            bad += 0 if e.status == -3 else 1

d = tempfile.mkdtemp(prefix="goss_long_")
fa = os.path.join(d, "c.fa")
with open(fa, "w") as f:
    for name, s in (("chr1", chrom[:20_000_000]), ("chr2", seq(3_000_000)), ("empty", ""), ("chr3", chrom[100:5_000_100])):
        f.write(">%s\n" % name)
        for i in range(0, len(s), 70):
            f.write(s[i:i + 70] + "\n")
fq = os.path.join(d, "u.fq")
with open(fq, "w") as f:
    for i in range(3000):
        n = 5_000_000 if i in (7, 1500) else rng.randint(100, 20000)
        s = chrom[i * 1000:i * 1000 + n]
        f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
for k, cmd, kind in ((25, "build-kmer-set", 0), (31, "build-graph", 1)):
    build = oracle.build_graph if kind else oracle.build_kmer_set
    exp, nwin = build([(oracle.FASTA, "c.fa", open(fa).read()), (oracle.FASTQ, "u.fq", open(fq).read())], k, out="o")
    for T in ("1", "8"):
        for extra in ([], ["--hbm-budget", "3"]):
            out = os.path.join(d, "o")
            for n in os.listdir(d):
                if n.startswith("o.") or n.startswith("o-"):
                    os.remove(os.path.join(d, n))
            p = subprocess.run([GOSS, cmd, "-k", str(k), "-I", fa, "-i", fq, "-O", out, "-T", T] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            if p.returncode and b"out of memory (HBM budget)" in p.stderr:
                # (the counted set, its merge and the staging buffers must fit the budget: a refusal, as in tests/fuzz_parity.py)
                print("ok   goss %s -T %s %s refused: HBM budget" % (cmd, T, extra), flush=True)
                continue
            if p.returncode:
                print("FAIL goss %s -T %s %s: %s" % (cmd, T, extra, p.stderr.decode()[-300:]), flush=True)
                bad += 1
                continue
            got = {n: open(os.path.join(d, n), "rb").read() for n in os.listdir(d) if n.startswith("o.") or n.startswith("o-")}
            check("goss %s k=%d -T %s %s" % (cmd, k, T, extra), got, exp)
import shutil
shutil.rmtree(d, ignore_errors=True)
print("long reads:", bad, "failed")
sys.exit(1 if bad else 0)
