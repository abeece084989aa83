#!/usr/bin/env python3
"""CPU-only fuzz of the host's FASTQ / FASTA framing (goss dump-bases; no GPU): random files -- wrapped records,
quality lines that start with '@' or '+', \\r\\n, empty reads, no trailing newline, and now and then a defect --
framed serially, by the parallel framer with random thread counts / chunk sizes / slack / mapping, and by the oracle's
restatement of the reference's parsers (FastqParser.hh:78-176, FastaParser.hh:50-110).  All three must agree: the
same bases in the same order, or the same message with the same line number.

    python tools/dbg/parser_fuzz.py [cases] [seed]
"""
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle  # noqa: E402  (the checker)

GOSS = os.environ.get("GOSS_BIN") or os.path.join(ROOT, "gossamer_amd", "goss")          # (GOSS_BIN: a sanitizer build of the host)


def make_fastq(rng):
    n = rng.choice([1, 2, 7, 50, 400, 3000])
    alphabet = rng.choice(["ACGT", "ACGTN", "ACGTNacgtn", "ACGT.RY"])
    eol_all = rng.choice(["\n", "\n", "\r\n", None])
    max_len = rng.choice([5, 40, 150, 600])
    p_wrap = rng.choice([0.0, 0.02, 0.5])
    p_nasty = rng.choice([0.0, 0.05, 0.9])
    p_empty = rng.choice([0.0, 0.0, 0.05])
    out = []
    for i in range(n):
        eol = eol_all if eol_all is not None else rng.choice(["\n", "\r\n"])
        ln = 0 if rng.random() < p_empty else rng.randint(1, max_len)
        s = "".join(rng.choice(alphabet) for _ in range(ln))
        qa = "@+I5#" if rng.random() < p_nasty else "I5#;"
        q = "".join(rng.choice(qa) for _ in range(ln))
        title = "r%d %s" % (i, "".join(rng.choice("@+ :/x") for _ in range(rng.randint(0, 6))))
        same = rng.random() < 0.3
        if ln > 2 and rng.random() < p_wrap:
            pieces = rng.randint(2, min(5, ln))
            cuts = sorted(rng.sample(range(1, ln), pieces - 1))
            sp = [s[a:b] for a, b in zip([0] + cuts, cuts + [ln])]
            cuts = sorted(rng.sample(range(1, ln), pieces - 1)) if rng.random() < 0.5 else cuts
            qp = [q[a:b] for a, b in zip([0] + cuts, cuts + [ln])]
            out.append("@%s%s%s%s+%s%s%s%s" % (title, eol, eol.join(sp), eol, title if same else "", eol, eol.join(qp), eol))
        else:
            out.append("@%s%s%s%s+%s%s%s%s" % (title, eol, s, eol, title if same else "", eol, q, eol))
    text = "".join(out)
    r = rng.random()
    if r < 0.15:
        text = text.rstrip("\r\n")
    elif r < 0.25:
        text += rng.choice(["\n", "\n\n", "\r\n"])
    # defects
    r = rng.random()
    if r < 0.12 and len(text) > 10:
        at = rng.randrange(len(text))
        text = text[:at] + rng.choice(["@", "+", "\n", "", "\n@x\n", "-"]) + text[at + rng.randint(0, 3):]
    elif r < 0.16:
        text = text[:rng.randrange(len(text))] if text else text
    return text


def make_fasta(rng):
    n = rng.choice([1, 3, 40, 600])
    eol = rng.choice(["\n", "\r\n"])
    out = []
    for i in range(n):
        lines = rng.randint(0, 4)
        out.append(">s%d d%s" % (i, eol))
        for _ in range(lines):
            out.append("".join(rng.choice("ACGTNacgt") for _ in range(rng.randint(0, 80))) + eol)
    text = "".join(out)
    if rng.random() < 0.2:
        text = text.rstrip("\r\n")
    if rng.random() < 0.08 and len(text) > 4:
        at = rng.randrange(len(text))
        text = text[:at] + rng.choice([">", "\n>", "\n\n"]) + text[at:]
    return text


def run(args, env=None):
    p = subprocess.run([GOSS] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    return p.returncode, p.stdout, p.stderr.decode(errors="replace")


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    failed = 0
    errors_seen = 0
    tmp = tempfile.mkdtemp(prefix="parserfuzz")
    for case in range(cases):
        fasta = rng.random() < 0.2
        text = make_fasta(rng) if fasta else make_fastq(rng)
        path = os.path.join(tmp, "c%d.%s" % (case, "fa" if fasta else "fq"))
        with open(path, "w", newline="") as f:
            f.write(text)
        flag = "-I" if fasta else "-i"
        kind = oracle.FASTA if fasta else oracle.FASTQ
        k = rng.choice([3, 11, 25])
        try:
            exp = oracle.collect([(kind, path, text)], k, 0)
            exp_err = None
        except oracle.OracleError as e:
            exp, exp_err = None, str(e)
        rc, out, err = run(["dump-bases", flag, path])
        why = None
        if exp_err is not None:
            errors_seen += 1
            # the oracle reports "<name>: <message>" or the bare message; the CLI wraps it (App.cc:357-363)
            msg = exp_err.split(": ", 1)[-1] if ": " in exp_err else exp_err
            if rc != 1 or msg not in err:
                why = "oracle refuses (%r), serial: rc %d %r" % (exp_err, rc, err[:200])
        else:
            if rc != 0:
                if not (exp[1] == 0 and "No valid reads." in err):
                    why = "serial refuses what the oracle takes: rc %d %r" % (rc, err[:200])
            else:
                got = oracle.collect([(oracle.LINE, "dump", out.decode())], k, 0)
                if got[0] != exp[0] or got[1] != exp[1] or got[2] != exp[2]:
                    why = "serial framing differs from the oracle's (%d/%d reads, %d/%d windows)" % (got[1], exp[1], got[2], exp[2])
        if why is None and not fasta:
            for _ in range(3):
                env = dict(os.environ, GOSS_PARSE_CHUNK=str(rng.choice([64, 100, 257, 1024, 4096, 65536])))
                if rng.random() < 0.4:
                    env["GOSS_PARSE_SLACK"] = str(rng.choice([1, 7, 64, 300]))
                if rng.random() < 0.5:
                    env["GOSS_PARSE_MMAP"] = rng.choice(["0", "1"])
                T = str(rng.choice([2, 3, 4, 8]))
                prc, pout, perr = run(["dump-bases", "-T", T, flag, path], env)
                # (what reached stdout before a defect was met is not compared: a build that meets one writes nothing)
                if prc != rc or perr != err or (rc == 0 and pout != out):
                    why = "parallel (T %s, %s) differs from serial: rc %d/%d, %d/%d bytes, %r / %r" % (
                        T, {k_: v for k_, v in env.items() if k_.startswith("GOSS_PARSE")}, prc, rc, len(pout), len(out), perr[:160], err[:160])
                    break
        if why:
            failed += 1
            print("FAIL case %d (%s, %d bytes): %s" % (case, path, len(text), why), flush=True)
        else:
            os.unlink(path)
    if not failed:
        shutil.rmtree(tmp, ignore_errors=True)
    print("parser fuzz: seed %d, %d cases (%d refused by both alike), %d failed%s" % (seed, cases, errors_seen, failed, "; failing files kept in " + tmp if failed else ""))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
