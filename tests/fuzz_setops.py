#!/usr/bin/env python3
"""Randomised parity run of the commands either side of the build on the GPU box: objects built by
`goss build-kmer-set / build-graph` from random overlapping read sets (checked against the oracle's builds), then
merge-kmer-sets / merge-graphs (random --max-merge, with and without --tmp-dir), intersect-kmer-sets, subtract-kmer-set,
merge-and-annotate-kmer-sets, graph-to-kmer-set, dump / restore -- every output file against the oracle's restatement
of the reference's loops (GossCmdMerge.tcc:151-326, GossCmdIntersectKmerSets.cc, GossCmdSubtractKmerSet.cc,
GossCmdMergeAndAnnotateKmerSets.cc, GossCmdGraphToKmerSet.cc:30-59).  usage: python tests/fuzz_setops.py [cases] [seed]"""
import os
import random
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # (this file lives in tests/: it uses the oracle)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle  # noqa: E402
from fuzz_parity import make_reads  # noqa: E402

GOSS = os.path.join(ROOT, "gossamer_amd", "goss")


def run(args, env=None):
    return subprocess.run([GOSS] + args + ["--hbm-budget", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)


def disk(d, base):
    return {n: open(os.path.join(d, n), "rb").read() for n in os.listdir(d) if n.startswith(base + ".") or n.startswith(base + "-")}


LAST = {}          # the two file sets of the last comparison (kept beside the inputs when a case fails)


def differs(got, exp):
    LAST["got"], LAST["exp"] = got, exp
    if sorted(got) != sorted(exp):
        return "file names: %s / %s" % (sorted(set(got) - set(exp))[:4], sorted(set(exp) - set(got))[:4])
    bad = [n for n in exp if got[n] != exp[n]]
    return "files differ: %s" % bad[:4] if bad else ""


def one_case(rng, d):
    graph = rng.random() < 0.45
    k = rng.choice([11, 15, 16, 21, 25, 27, 31, 32, 33, 45, 55, 62])
    kind = 1 if graph else 0
    nsets = rng.choice([2, 2, 3, 4, 6])
    # one genome's worth of features per case: the sets overlap where their reads do
    seed = rng.randrange(1 << 30)
    genome_len = rng.choice([3000, 40000, 300000])
    log = []
    files, names = {}, []
    for i in range(nsets):
        shape = rng.random()
        if shape < 0.12:
            reads = b"ACGTACG\n"                      # too short for any k here: an empty object
        else:
            r2 = random.Random(seed if rng.random() < 0.7 else seed + i + 1)      # mostly the same genome, sometimes another
            n = rng.choice([300, 3000, 30000])
            reads = make_reads(r2, n, max(genome_len, 260), *rng.choice([(150, 150), (40, 250), (k + 1, k + 8)]), rng.choice([0.0, 0.0, 0.01]),
                               rng.choice([0.0, 0.002]), rng.random() < 0.2, read_rng=random.Random(rng.randrange(1 << 30)))
            if shape > 0.9 and names:
                reads = open(os.path.join(d, "p0.txt"), "rb").read()          # the same set twice
        with open(os.path.join(d, "p%d.txt" % i), "wb") as f:
            f.write(reads)
        base = "in%d" % i
        env = dict(os.environ)
        if rng.random() < 0.5:
            env["GOSS_GPU_FUSED_MIN"] = "0"
        p = run(["build-graph" if graph else "build-kmer-set", "-k", str(k), "--line-in", os.path.join(d, "p%d.txt" % i), "-O", os.path.join(d, base)], env)
        if p.returncode != 0:
            return "build %d failed: %s" % (i, p.stderr.decode(errors="replace")[-300:]), log
        build = oracle.build_graph if graph else oracle.build_kmer_set
        exp, _ = build([(oracle.LINE, "r", reads)], k, out=base)
        got = disk(d, base)
        why = differs(got, exp)
        if why:
            return "build %d: %s" % (i, why), log
        files.update(got)
        names.append(base)
    P = [os.path.join(d, n) for n in names]
    log.append("%s k=%d sets=%d" % ("graph" if graph else "kmer", k, nsets))

    # merges
    for rep in range(2):
        sel = [rng.randrange(nsets) for _ in range(rng.randint(1, min(nsets + 1, 6)))] if rng.random() < 0.3 else list(range(nsets))
        mm = rng.choice([2, 3, 8])
        out = "m%d" % rep
        exp = oracle.merge(files, [names[j] for j in sel], kind, out, max_merge=mm)
        args = ["merge-graphs" if graph else "merge-kmer-sets", "-O", os.path.join(d, out), "--max-merge", str(mm)]
        if rng.random() < 0.4 and len(sel) > 1:
            with open(os.path.join(d, "list.txt"), "w") as f:
                f.write("".join(P[j] + "\n" for j in sel[1:]))
            args += ["-G", P[sel[0]], "--graphs-in", os.path.join(d, "list.txt")]
        else:
            for j in sel:
                args += ["-G", P[j]]
        tdir = None
        if rng.random() < 0.4:
            tdir = os.path.join(d, "tmp%d" % rep)
            os.mkdir(tdir)
            args += ["--tmp-dir", tdir]
        p = run(args)
        log.append("merge %s mm=%d%s" % (sel, mm, " tmp" if tdir else ""))
        if p.returncode != 0:
            return "merge failed: %s" % p.stderr.decode(errors="replace")[-300:], log
        why = differs(disk(d, out), exp)
        if why:
            return "merge %s max-merge %d: %s" % (sel, mm, why), log
        if tdir and os.listdir(tdir):
            return "merge left files in --tmp-dir: %s" % os.listdir(tdir)[:3], log
        # the merged object read back: dump equal to the oracle's dump of its own merge
        if rng.random() < 0.5:
            txt = os.path.join(d, out + ".txt")
            p = run(["dump-graph" if graph else "dump-kmer-set", "-G", os.path.join(d, out), "-o", txt])
            if p.returncode != 0:
                return "dump failed: %s" % p.stderr.decode(errors="replace")[-300:], log
            if open(txt, "rb").read() != oracle.dump(exp, out, kind):
                return "dump of the merge differs", log
            if graph and rng.random() < 0.6:
                p = run(["restore-graph", "-f", txt, "-O", os.path.join(d, out + "r")])
                if p.returncode != 0:
                    return "restore failed: %s" % p.stderr.decode(errors="replace")[-300:], log
                why = differs(disk(d, out + "r"), oracle.restore_graph(open(txt, "rb").read(), out + "r"))
                if why:
                    return "restore: " + why, log
                log.append("dump+restore")

    if graph:
        j = rng.randrange(nsets)
        try:
            exp = oracle.graph_to_kmer_set(files, names[j], "g2k")
        except oracle.OracleError as e:
            exp = None
            log.append("graph-to-kmer-set refused by the oracle: %s" % e)
        if exp is not None:
            p = run(["graph-to-kmer-set", "-G", P[j], "-O", os.path.join(d, "g2k")])
            if p.returncode != 0:
                return "graph-to-kmer-set failed: %s" % p.stderr.decode(errors="replace")[-300:], log
            why = differs(disk(d, "g2k"), exp)
            if why:
                return "graph-to-kmer-set: " + why, log
        return "", log

    # set algebra (k-mer sets)
    for rep in range(2):
        sel = [rng.randrange(nsets) for _ in range(rng.randint(1, 4))]
        out = "i%d" % rep
        try:
            exp = oracle.intersect_kmer_sets(files, [names[j] for j in sel], out)
        except oracle.OracleError as e:          # (every input empty: undefined in the reference)
            log.append("intersect %s refused by the oracle: %s" % (sel, e))
            continue
        args = ["intersect-kmer-sets", "-O", os.path.join(d, out)]
        for j in sel:
            args += ["-G", P[j]]
        p = run(args)
        log.append("intersect %s" % sel)
        if p.returncode != 0:
            return "intersect failed: %s" % p.stderr.decode(errors="replace")[-300:], log
        why = differs(disk(d, out), exp)
        if why:
            return "intersect %s: %s" % (sel, why), log
    for rep in range(2):
        a, b = rng.randrange(nsets), rng.randrange(nsets)
        out = "d%d" % rep
        exp = oracle.subtract_kmer_set(files, names[a], names[b], out)
        p = run(["subtract-kmer-set", "-G", P[a], "-G", P[b], "-O", os.path.join(d, out)])
        log.append("subtract %d %d" % (a, b))
        if p.returncode != 0:
            return "subtract failed: %s" % p.stderr.decode(errors="replace")[-300:], log
        why = differs(disk(d, out), exp)
        if why:
            return "subtract %d %d: %s" % (a, b, why), log
    a, b = rng.randrange(nsets), rng.randrange(nsets)
    try:
        exp, stats = oracle.merge_and_annotate(files, names[a], names[b], "u")
    except oracle.OracleError as e:
        exp = None
        log.append("annotate %d %d refused by the oracle: %s" % (a, b, e))
    p = run(["merge-and-annotate-kmer-sets", "-G", P[a], "-G", P[b], "-O", os.path.join(d, "u")])
    if exp is None:
        if p.returncode == 0:
            return "annotate %d %d: the oracle refuses, the command does not" % (a, b), log
    else:
        log.append("annotate %d %d" % (a, b))
        if p.returncode != 0:
            return "annotate failed: %s" % p.stderr.decode(errors="replace")[-300:], log
        if p.stdout.decode() != "%d\t%d\t%d\n" % stats:
            return "annotate %d %d: stdout %r, expected %r" % (a, b, p.stdout.decode(), stats), log
        why = differs({n: v for n, v in disk(d, "u").items()}, exp)
        if why:
            return "annotate %d %d: %s" % (a, b, why), log
    return "", log


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    oracle.lib()
    bad = 0
    t00 = time.time()
    keep = os.environ.get("FUZZ_DUMP_DIR")
    for case in range(cases):
        d = tempfile.mkdtemp(prefix="goss_setops_")
        t0 = time.time()
        try:
            why, log = one_case(rng, d)
        except subprocess.TimeoutExpired as e:
            why, log = "timed out: %s" % e, []
        except oracle.OracleError as e:          # (a case the oracle has no answer for is not a case)
            why, log = "", ["skipped, the oracle refuses: %s" % e]
        print("%s case %d: %s %.1fs %s" % ("FAIL" if why else "ok  ", case, "; ".join(log), time.time() - t0, why), flush=True)
        if why:
            bad += 1
            if keep:
                os.makedirs(keep, exist_ok=True)
                for n in os.listdir(d):
                    if n.startswith("p") and n.endswith(".txt"):
                        shutil.copy(os.path.join(d, n), os.path.join(keep, "setops%d_%s" % (case, n)))
                for side in ("got", "exp"):
                    for n, v in LAST.get(side, {}).items():
                        with open(os.path.join(keep, "setops%d_%s_%s" % (case, side, n)), "wb") as f:
                            f.write(v)
        shutil.rmtree(d, ignore_errors=True)
    print("setops fuzz: seed %d, %d cases, %d failed, %.0f s" % (seed, cases, bad, time.time() - t00))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
