#!/usr/bin/env python3
"""Replay of a failing case of tests/fuzz_setops.py from its dumped inputs (tools/dbg/data/setops<case>_p<i>.txt):
    python tools/dbg/setops_replay.py <case> <k> merge|g2k [max-merge]
builds the graphs with goss, runs the command, compares with the oracle and prints where the decoded contents differ."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle  # noqa: E402

GOSS = os.path.join(ROOT, "gossamer_amd", "goss")
DATA = os.path.join(ROOT, "tools", "dbg", "data")


def run(args):
    p = subprocess.run([GOSS] + args + ["--hbm-budget", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if p.returncode:
        print("FAILED", args, p.stderr.decode()[-400:])
    return p


def disk(d, base):
    return {n: open(os.path.join(d, n), "rb").read() for n in os.listdir(d) if n.startswith(base + ".") or n.startswith(base + "-")}


def compare(tag, got, exp, d, base, kind):
    bad = [n for n in sorted(exp) if got.get(n) != exp[n]]
    print(tag, "files differing:", bad, "missing:", sorted(set(exp) - set(got)), "extra:", sorted(set(got) - set(exp)))
    if not bad:
        return
    for n in bad[:6]:
        a, b = got.get(n, b""), exp[n]
        first = next((i for i in range(min(len(a), len(b))) if a[i] != b[i]), min(len(a), len(b)))
        print("   %s: %d / %d bytes, first difference at byte %d" % (n, len(a), len(b), first))
    txt = os.path.join(d, base + ".dump")
    run(["dump-graph" if kind else "dump-kmer-set", "-G", os.path.join(d, base), "-o", txt])
    mine = open(txt, "rb").read().split(b"\n")
    ref = oracle.dump(exp, base, kind).split(b"\n")
    print("   dump lines %d / %d" % (len(mine), len(ref)))
    sm, sr = set(mine), set(ref)
    print("   only in the product's: %d, only in the oracle's: %d" % (len(sm - sr), len(sr - sm)))
    for x in sorted(sm - sr)[:8]:
        print("     product:", x[:120])
    for x in sorted(sr - sm)[:8]:
        print("     oracle: ", x[:120])
    if sm == sr and mine != ref:
        at = next(i for i in range(min(len(mine), len(ref))) if mine[i] != ref[i])
        print("   same lines, another order; first at line", at, mine[at][:100], ref[at][:100])


def main():
    case, k, op = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    mm = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    d = tempfile.mkdtemp(prefix="setops_replay_")
    files, names = {}, []
    i = 0
    while os.path.exists(os.path.join(DATA, "setops%s_p%d.txt" % (case, i))):
        src = os.path.join(DATA, "setops%s_p%d.txt" % (case, i))
        base = "in%d" % i
        run(["build-graph", "-k", str(k), "--line-in", src, "-O", os.path.join(d, base)])
        exp, _ = oracle.build_graph([(oracle.LINE, "r", open(src, "rb").read())], k, out=base)
        got = disk(d, base)
        print("build", i, "equal" if got == exp else "DIFFERS", "edges", int.from_bytes(got[base + ".header"][16:24], "little") if base + ".header" in got else "?")
        files.update(got)
        names.append(base)
        i += 1
    if op == "merge":
        exp = oracle.merge(files, names, 1, "m", max_merge=mm)
        args = ["merge-graphs", "-O", os.path.join(d, "m"), "--max-merge", str(mm), "-v"]
        for n in names:
            args += ["-G", os.path.join(d, n)]
        p = run(args)
        print(p.stderr.decode()[-1500:])
        compare("merge", disk(d, "m"), exp, d, "m", 1)
    else:
        for j, n in enumerate(names):
            try:
                exp = oracle.graph_to_kmer_set(files, n, "g2k%d" % j)
            except oracle.OracleError as e:
                print("g2k", j, "refused by the oracle:", e)
                continue
            p = run(["graph-to-kmer-set", "-G", os.path.join(d, n), "-O", os.path.join(d, "g2k%d" % j), "-v"])
            print(p.stderr.decode()[-600:])
            compare("g2k %d" % j, disk(d, "g2k%d" % j), exp, d, "g2k%d" % j, 0)


if __name__ == "__main__":
    main()
