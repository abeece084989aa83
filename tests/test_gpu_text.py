"""goss dump-kmer-set / dump-graph / restore-graph / lint-graph on the GPU against the oracle."""
import os
import random
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOSS = os.path.join(ROOT, "gossamer_amd", "goss")


def run(args, **kw):
    return subprocess.run([GOSS] + args + ["--hbm-budget", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, **kw)


def disk(tmp_path, base):
    return {n: (tmp_path / n).read_bytes() for n in os.listdir(tmp_path) if n.startswith(base + ".") or n.startswith(base + "-")}


def property_lines(files, base, k):
    """Graph::stat().print(out, 1) of the reference (Graph.hh:588-603, SparseArray.cc:142-162, DenseArray.cc:94-132,
    WordyBitVector.hh:265-270, IntegerArray.cc:111-116 / 182-187, VariableByteArray.hh:249-266, Properties.hh:72-85) from
    the object's files: properties, then sub-trees, both in map order; one space per level."""
    import struct

    def u64s(b, n):
        return struct.unpack("<%dQ" % n, b[:8 * n])

    def dense(name):
        f = files[name]
        h = u64s(f, 16)
        return {"storage": 128 + len(f), "invertSense": h[1] & 1,
                "index": {"entries": h[8], "size": h[9]}, "smallBlocks": {"entries": h[10], "size": h[11]},
                "intermediateBlocks": {"entries": h[12], "size": h[13]}, "largeBlocks": {"entries": h[14], "size": h[15]}}

    def sparse(b):
        h = u64s(files[b + ".header"], 8)
        hb = len(files[b + ".high-bits"]) // 8 * 8
        lb = sum(len(v) for n, v in files.items() if n.startswith(b + ".low-bits"))
        t = {"high-bits": {"storage": hb}, "low-bits": {"storage": lb}, "D0": dense(b + "-d0"), "D1": dense(b + "-d1"),
             "size": h[5] | (h[6] << 64), "count": h[7]}
        t["storage"] = 64 + hb + lb + t["D0"]["storage"] + t["D1"]["storage"]
        return t

    o0, o1, o2 = (len(files[base + "-counts.ord%d" % i]) for i in range(3))
    counts = {"ord1-pred": sparse(base + "-counts.ord1p"), "ord2-pred": sparse(base + "-counts.ord2p"), "size": o0}
    counts["storage"] = o0 + counts["ord1-pred"]["storage"] + o1 + counts["ord2-pred"]["storage"] + o2
    g = {"edges": sparse(base + "-edges"), "counts": counts, "K": k}
    g["count"] = g["edges"]["count"]
    g["storage"] = 24 + g["edges"]["storage"] + counts["storage"]
    out = []

    def emit(t, ind):
        for key in sorted(x for x in t if not isinstance(t[x], dict)):
            out.append(" " * ind + key + "\t" + str(t[key]))
        for key in sorted(x for x in t if isinstance(t[x], dict)):
            out.append(" " * ind + key)
            emit(t[key], ind + 1)
    emit(g, 1)
    return out


def test_dump_restore_lint(oracle, tmp_path):
    rng = random.Random(47)
    genome = "".join(rng.choice("ACGT") for _ in range(20000))
    reads = [genome[s:s + 130] for s in (rng.randrange(0, 19870) for _ in range(4000))]
    reads += ["A" * 150] * 300                         # a multiplicity above 255: exercises ord1
    txt = "\n".join(reads) + "\n"
    (tmp_path / "r.txt").write_text(txt)
    for k in (25, 40):
        ks, gr = "ks%d" % k, "gr%d" % k
        assert run(["build-kmer-set", "-k", str(k), "--line-in", str(tmp_path / "r.txt"), "-O", str(tmp_path / ks)]).returncode == 0
        assert run(["build-graph", "-k", str(k), "--line-in", str(tmp_path / "r.txt"), "-O", str(tmp_path / gr)]).returncode == 0
        # dump to stdout and to a file
        p = run(["dump-kmer-set", "-G", str(tmp_path / ks)])
        assert p.returncode == 0, p.stderr.decode()
        assert p.stdout == oracle.dump(disk(tmp_path, ks), ks, 0)
        p = run(["dump-graph", "-G", str(tmp_path / gr), "-o", str(tmp_path / "g.txt")])
        assert p.returncode == 0, p.stderr.decode()
        text = (tmp_path / "g.txt").read_bytes()
        assert text == oracle.dump(disk(tmp_path, gr), gr, 1)
        assert ("A" * (k + 1) + "\t%d\n" % (300 * (150 - k))).encode() in text
        # restore: byte-identical to the oracle's restore, and to the original graph
        back = "back%d" % k
        p = run(["restore-graph", "-f", str(tmp_path / "g.txt"), "-O", str(tmp_path / back)])
        assert p.returncode == 0, p.stderr.decode()
        exp = oracle.restore_graph(text, back)
        got = disk(tmp_path, back)
        assert sorted(got) == sorted(exp)
        for name in exp:
            assert got[name] == exp[name], (k, name)
        orig = disk(tmp_path, gr)
        for name in orig:
            assert got[back + name[len(gr):]] == orig[name], (k, name)
        # restore from stdin, asymmetric flag set
        flagged = text.replace(b"\t0\n", b"\t1\n", 1)
        p = run(["restore-graph", "-O", str(tmp_path / "asym")], input=flagged)
        assert p.returncode == 0, p.stderr.decode()
        exp = oracle.restore_graph(flagged, "asym")
        got = disk(tmp_path, "asym")
        assert all(got[n] == exp[n] for n in exp) and sorted(got) == sorted(exp)
        # lint: a graph built from reads is symmetric by construction
        p = run(["lint-graph", "-G", str(tmp_path / gr), "-v"])
        assert p.returncode == 0, p.stderr.decode()
        err = p.stderr.decode()
        assert "Pass 1: Checking counts are sane." in err and "Pass 2: Checking traversal is sane." in err
        assert "warning" not in err
        # --dump-properties: the reference's PropertyTree of the graph, line for line (one empty line behind it)
        p = run(["lint-graph", "-G", str(tmp_path / gr), "--dump-properties", "-v"])
        assert p.returncode == 0, p.stderr.decode()
        msgs = [ln.split("\tinfo\t", 1)[1] if "\tinfo\t" in ln else None for ln in p.stderr.decode().split("\n")]
        at = msgs.index("Graph properties:")
        want = property_lines(disk(tmp_path, gr), gr, k)
        got = [m if m is not None else "" for m in msgs[at + 1:at + 2 + len(want)]]
        assert got == want + [""], "\n".join(got)
        assert msgs[at + 2 + len(want)] == "Pass 1: Checking counts are sane."
        # damage the select index of a copy (the second block of -edges-d1 starts one position late):
        # pass 2 evaluates the object's own select / rank on the device and must notice
        import shutil
        import struct
        for n in os.listdir(tmp_path):
            if n.startswith(gr + ".") or n.startswith(gr + "-"):
                shutil.copy(tmp_path / n, tmp_path / ("dmg" + n[len(gr):]))
        b = bytearray((tmp_path / "dmg-edges-d1").read_bytes())
        rank_off = struct.unpack("<Q", b[24:32])[0]
        v = struct.unpack("<Q", b[rank_off + 8: rank_off + 16])[0]
        b[rank_off + 8: rank_off + 16] = struct.pack("<Q", v + 1)
        (tmp_path / "dmg-edges-d1").write_bytes(bytes(b))
        p = run(["lint-graph", "-G", str(tmp_path / "dmg")])
        assert p.returncode == 0
        assert "iterator and select conflict." in p.stderr.decode()
        # break the symmetry: drop one edge / change one count in the text, restore, lint
        lines = text.split(b"\n")
        comp = {65: "T", 67: "G", 71: "C", 84: "A"}

        def revcomp(seq):
            return "".join(comp[b] for b in reversed(seq))

        victim = next(i for i in range(2, len(lines) - 2)
                      if lines[i] and revcomp(lines[i].split(b"\t")[0]).encode() != lines[i].split(b"\t")[0]
                      and revcomp(lines[i + 1].split(b"\t")[0]).encode() != lines[i + 1].split(b"\t")[0])
        broken = b"\n".join(lines[:victim] + lines[victim + 1:])
        p = run(["restore-graph", "-O", str(tmp_path / "brk")], input=broken)
        assert p.returncode == 0, p.stderr.decode()
        p = run(["lint-graph", "-G", str(tmp_path / "brk")])
        err = p.stderr.decode()
        assert p.returncode == 0
        assert err.count("No reverse complement for the following edge exists:") == 1
        seq, c = lines[victim].split(b"\t")
        rc = revcomp(seq)
        assert ("  fwd edge    %s %s" % (rc, c.decode())) in err
        seq2, c2 = lines[victim + 1].split(b"\t")
        bumped = b"\n".join(lines[:victim + 1] + [seq2 + b"\t" + str(int(c2) + 1).encode()] + lines[victim + 2:])
        p = run(["restore-graph", "-O", str(tmp_path / "bmp")], input=bumped)
        assert p.returncode == 0, p.stderr.decode()
        p = run(["lint-graph", "-G", str(tmp_path / "bmp")])
        err = p.stderr.decode()
        # the edge and its reverse complement both report the mismatch
        assert err.count("counts on fwd and rev edges are not equal:") == 2
    # errors
    p = run(["dump-graph"])
    assert p.returncode == 1 and "mandatory option graph-in was not given." in p.stderr.decode()
    p = run(["restore-graph", "-O", str(tmp_path / "bad")], input=b"#v\n25\t1\t0\nACGT\t1\n")
    assert p.returncode == 1 and "sequence ACGT has wrong length" in p.stderr.decode()


def test_several_gz_inputs_side_by_side(oracle, tmp_path):
    """Compressed FASTQ files (.gz and .bz2) are decompressed and framed by one worker each (-T > 1): the objects must
    be the ones a serial pass over the same files gives, and a framing error in one of them must
    still surface as the reference's message."""
    import gzip
    rng = random.Random(53)
    genome = "".join(rng.choice("ACGT") for _ in range(5000))
    inputs = []
    args = []
    for i in range(4):
        reads = [genome[s:s + 100] for s in (rng.randrange(0, 4900) for _ in range(800))]
        fq = "".join("@r%d\n%s\n+\n%s\n" % (j, r, "I" * len(r)) for j, r in enumerate(reads))
        name = ("p%d.fq.gz" % i if i < 2 else "p2.fq.bz2") if i < 3 else "p3.fq"
        if name.endswith(".gz"):
            with gzip.open(tmp_path / name, "wb") as f:
                f.write(fq.encode())
        elif name.endswith(".bz2"):
            import bz2
            (tmp_path / name).write_bytes(bz2.compress(fq.encode()))
        else:
            (tmp_path / name).write_text(fq)
        inputs.append((oracle.FASTQ, name, fq))
        args += ["-i", str(tmp_path / name)]
    for cmd, k, build, base in (("build-kmer-set", 25, oracle.build_kmer_set, "ks"), ("build-graph", 31, oracle.build_graph, "gr")):
        exp, nwin = build(inputs, k, out=base)
        p = run([cmd, "-k", str(k), "-T", "4", "-O", str(tmp_path / base)] + args)
        assert p.returncode == 0, p.stderr.decode()
        got = disk(tmp_path, base)
        assert sorted(got) == sorted(exp)
        for name in exp:
            assert got[name] == exp[name], (cmd, name)
    with gzip.open(tmp_path / "bad.fq.gz", "wb") as f:
        f.write(b"@r0\nACGT\n+\nIII\n")                       # quality shorter than the sequence
    p = run(["build-kmer-set", "-k", "25", "-T", "4", "-O", str(tmp_path / "x"), "-i", str(tmp_path / "p0.fq.gz"),
             "-i", str(tmp_path / "bad.fq.gz")])
    assert p.returncode == 1 and "bad.fq.gz" in p.stderr.decode()
